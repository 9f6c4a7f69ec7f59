#!/bin/bash
out=gpurun_out/${1:-r3k}; mkdir -p $out
timeout -k 10 500 python -m pytest tests/test_gpu_group.py tests/test_gpu_utf8_fold.py tests/test_gpu_operators.py tests/test_gpu_threads.py tests/test_gpu_parity.py tests/test_c_example.py -x -q > $out/tests.txt 2>&1; echo "tests rc=$?"; tail -12 $out/tests.txt | cut -c1-300
CHQ_TIMING=1 timeout -k 10 200 python bench_configs.py --steps 5 --only "12 500 x 10k-row" > $out/refgroup_timing.txt 2>&1
grep "chq timing" $out/refgroup_timing.txt | tail -6 | cut -c1-330
grep -o '"group_call_ms": [0-9.]*\|"c_call_ms": [0-9.]*\|"coalesced_call_ms": [0-9.]*' $out/refgroup_timing.txt
