#!/bin/bash
# parquet tests, then a kernel trace of the snappy scan (per-launch durations of pq_inflate_kernel: index / blocks / finish)
out=gpurun_out/${1:-r3z}; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_parquet.py -x -q > $out/t.txt 2>&1; rc=$?; echo "parquet tests rc=$rc"; tail -3 $out/t.txt | cut -c1-600
[ $rc -ne 0 ] && exit $rc
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for shape in sample compressible; do
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $out/prof_$shape -o scan -- python3 bench/micro/parquet_scan.py 8000000 snappy $shape > $out/run_$shape.txt 2>&1 || exit 1
grep "chq scan" $out/run_$shape.txt
done
