#!/bin/bash
out=gpurun_out/${1:-r3h}; mkdir -p $out
timeout -k 10 300 python -m pytest tests/test_gpu_parquet.py tests/test_gpu_parity.py -x -q > $out/tests.txt 2>&1; echo "tests rc=$?"; tail -3 $out/tests.txt
for a in "none sample" "snappy sample" "snappy compressible"; do
  timeout -k 10 200 python bench/micro/parquet_scan.py 20000000 $a > $out/pq_$(echo $a | tr ' ' '_').txt 2>&1; grep "chq scan" $out/pq_$(echo $a | tr ' ' '_').txt
done
timeout -k 10 260 python -m tests.fuzz_long 180 3201 > $out/fuzz.txt 2>&1; echo "fuzz rc=$?"; tail -2 $out/fuzz.txt | cut -c1-400
