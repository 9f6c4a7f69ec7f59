#!/bin/bash
# round-3 evidence, part C: the bench line, every config, the Parquet scan rates
cd "$(dirname "$0")/.."
OUT=gpurun_out/profiles_r3; mkdir -p $OUT
timeout -k 10 600 python bench.py > $OUT/bench_line.json 2> $OUT/bench_line.err; echo "bench rc=$?"; cut -c1-600 $OUT/bench_line.json
timeout -k 10 300 python bench.py --config 5 --steps 5 --warmup 2 > $OUT/bench_config5_line.json 2> $OUT/bench_config5.err; echo "bench config5 rc=$?"; cut -c1-600 $OUT/bench_config5_line.json
timeout -k 10 900 python bench_configs.py --out $OUT/configs.json > $OUT/configs.log 2>&1; echo "configs rc=$?"; tail -3 $OUT/configs.log | cut -c1-300
{ for args in "20000000 none sample" "20000000 snappy sample" "20000000 snappy compressible"; do echo "== parquet_scan.py $args"; timeout -k 10 200 python bench/micro/parquet_scan.py $args 2>&1 | grep -v amdgpu.ids | tail -9; done; } > $OUT/parquet_scan.txt; cat $OUT/parquet_scan.txt | cut -c1-200
