#!/bin/bash
out=gpurun_out/${1:-r3m}; mkdir -p $out
timeout -k 10 300 python -m pytest tests/test_gpu_group.py -x -q > $out/tests.txt 2>&1; rc=$?; echo "tests rc=$rc"; tail -5 $out/tests.txt | cut -c1-300
[ $rc -ne 0 ] && exit $rc
CHQ_TIMING=1 timeout -k 10 300 python bench_configs.py --steps 5 --only "12 500 x 10k-row" > $out/refgroup.txt 2>&1; echo "bench_configs rc=$?"
grep "chq timing" $out/refgroup.txt | tail -4 | cut -c1-400
grep -o '"group_call_ms": [0-9.]*\|"c_call_ms": [0-9.]*\|"coalesced_call_ms": [0-9.]*' $out/refgroup.txt
