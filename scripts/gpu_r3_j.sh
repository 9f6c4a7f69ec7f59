#!/bin/bash
out=gpurun_out/${1:-r3j}; mkdir -p $out
timeout -k 10 400 python -m pytest tests/test_gpu_group.py tests/test_gpu_utf8_fold.py tests/test_gpu_operators.py tests/test_gpu_threads.py tests/test_gpu_scale.py -x -q > $out/tests.txt 2>&1; echo "tests rc=$?"; tail -3 $out/tests.txt
CHQ_TIMING=1 timeout -k 10 200 python bench_configs.py --steps 5 --only "12 500 x 10k-row" > $out/refgroup_timing.txt 2>&1
grep "chq timing" $out/refgroup_timing.txt | tail -8 | cut -c1-330
grep -o '"group_call_ms": [0-9.]*\|"c_call_ms": [0-9.]*\|"coalesced_call_ms": [0-9.]*' $out/refgroup_timing.txt
timeout -k 10 200 python bench_configs.py --steps 5 --only "10k-row batches" > $out/group2.txt 2>&1
grep -o '"c_call_ms": [0-9.]*\|"coalesced_call_ms": [0-9.]*\|"group_kernel_ms": [0-9.]*' $out/group2.txt
