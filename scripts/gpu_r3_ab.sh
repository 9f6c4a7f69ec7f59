#!/bin/bash
# round 3 A/B on one box: previous commit's library (bench/ab/libchq_head.so) vs the working tree's + the Parquet scan rates
out=gpurun_out/${1:-r3ab}; mkdir -p $out
bash scripts/gpu_ab.sh > $out/ab.txt 2>&1; tail -40 $out/ab.txt
for a in "none sample" "snappy sample" "snappy compressible" "none compressible"; do
  timeout -k 10 200 python bench/micro/parquet_scan.py 20000000 $a > $out/pq_$(echo $a | tr ' ' '_').txt 2>&1; tail -4 $out/pq_$(echo $a | tr ' ' '_').txt
done
