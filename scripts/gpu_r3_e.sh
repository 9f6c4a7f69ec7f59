#!/bin/bash
out=gpurun_out/${1:-r3e}; mkdir -p $out
timeout -k 10 300 python -m pytest tests/test_gpu_parquet.py -x -q > $out/pq_tests.txt 2>&1; echo "parquet tests rc=$?"; tail -2 $out/pq_tests.txt
for a in "snappy sample" "snappy compressible"; do
  timeout -k 10 200 python bench/micro/parquet_scan.py 20000000 $a > $out/pq_$(echo $a | tr ' ' '_').txt 2>&1; grep "chq scan\|pyarrow read" $out/pq_$(echo $a | tr ' ' '_').txt
done
