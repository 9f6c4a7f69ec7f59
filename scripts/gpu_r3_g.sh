#!/bin/bash
out=gpurun_out/${1:-r3g}; mkdir -p $out
timeout -k 10 300 python -m pytest tests/test_gpu_regs.py tests/test_gpu_config3.py -x -q > $out/regs_tests.txt 2>&1; echo "regs tests rc=$?"; tail -5 $out/regs_tests.txt
for r in 1 0; do
  timeout -k 10 200 python bench_configs.py --steps 7 --only "config3 compound" --no-select --opt regs=$r > $out/config3_regs$r.txt 2>&1
  echo "regs=$r $(grep -o '"filter_kernel_ms": [0-9.]*' $out/config3_regs$r.txt)"
done
