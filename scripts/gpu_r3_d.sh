#!/bin/bash
out=gpurun_out/${1:-r3d}; mkdir -p $out
timeout -k 10 300 python -m pytest tests/test_gpu_parquet.py -x -q > $out/pq_tests.txt 2>&1; echo "parquet tests rc=$?"; tail -2 $out/pq_tests.txt
for a in "snappy sample" "snappy compressible"; do
  timeout -k 10 200 python bench/micro/parquet_scan.py 20000000 $a > $out/pq_$(echo $a | tr ' ' '_').txt 2>&1; grep "chq scan" $out/pq_$(echo $a | tr ' ' '_').txt
done
bash scripts/gpu_c3regs.sh $(basename $out)
CHQ_TIMING=1 timeout -k 10 200 python bench_configs.py --steps 5 --only "12 500 x 10k-row" > $out/refgroup_timing.txt 2>&1
grep "chq timing" $out/refgroup_timing.txt | sort | uniq -c | sort -rn | head -5; grep "chq timing" $out/refgroup_timing.txt | tail -12
