#!/bin/bash
out=gpurun_out/${1:-r3q}; mkdir -p $out
timeout -k 10 400 python -m pytest tests/test_gpu_parquet.py -x -q > $out/tests.txt 2>&1; rc=$?; echo "parquet tests rc=$rc"; tail -12 $out/tests.txt | cut -c1-400
[ $rc -ne 0 ] && exit $rc
timeout -k 10 200 python bench_configs.py --only config2b --steps 5 --opt tile_kind=1 > $out/c2b_tile1.txt 2>&1; grep -o '"filter_kernel_ms": [0-9.]*' $out/c2b_tile1.txt
bash scripts/gpu_profiles_r3b.sh
