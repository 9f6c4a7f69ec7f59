#!/bin/bash
# new type coverage (temporal / decimal compares, Float16, Utf8 -> Boolean) + the nullable reference-schema group case
out=gpurun_out/${1:-r3l}; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_group.py tests/test_gpu_fused.py tests/test_gpu_operators.py -x -q > $out/tests.txt 2>&1; rc=$?; echo "tests rc=$rc"; tail -25 $out/tests.txt | cut -c1-400
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python bench_configs.py --steps 5 --only "12 500 x 10k-row" > $out/refgroup.txt 2>&1; echo "bench_configs rc=$?"
grep -o '"name": "[^"]*"\|"group_call_ms": [0-9.]*\|"c_call_ms": [0-9.]*\|"coalesced_call_ms": [0-9.]*\|"filter_kernel_ms": [0-9.]*' $out/refgroup.txt | head -40
