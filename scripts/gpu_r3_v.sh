#!/bin/bash
# experiment: tile kind 3 (512 threads x 16 rows, two workgroups per CU) against kind 0 on config 2 / 2b / 2c / 3
out=gpurun_out/${1:-r3v}; mkdir -p $out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -k "every_kernel or fuzz_filter" > $out/t.txt 2>&1; echo "tests rc=$?"; tail -2 $out/t.txt
for k in 0 3; do
  for c in "config2 value2>10" config2b config2c compound; do
    timeout -k 10 200 python bench_configs.py --only "$c" --steps 5 --no-select --opt tile_kind=$k > $out/k${k}.txt 2>&1
    echo "tile_kind=$k $c: $(grep -o '"filter_kernel_ms": [0-9.]*' $out/k${k}.txt | head -1)"
  done
done
