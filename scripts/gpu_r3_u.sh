#!/bin/bash
out=gpurun_out/${1:-r3u}; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_group.py tests/test_gpu_fused.py tests/test_gpu_operators.py tests/test_gpu_utf8_fold.py tests/test_gpu_large_host.py tests/test_gpu_threads.py -x -q > $out/tests.txt 2>&1; rc=$?; echo "tests rc=$rc"; tail -6 $out/tests.txt | cut -c1-400
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python bench/micro/nullable_projection.py 400000000 2>&1 | tail -2
timeout -k 10 300 python bench_configs.py --steps 5 --only "12 500 x 10k-row" > $out/refgroup.txt 2>&1; grep -o '"c_call_ms": [0-9.]*\|"coalesced_call_ms": [0-9.]*' $out/refgroup.txt
timeout -k 10 260 python -m tests.fuzz_long 200 91 > $out/fuzz_200s.txt 2>&1; echo "fuzz rc=$?"; tail -1 $out/fuzz_200s.txt | cut -c1-400
