#!/bin/bash
out=gpurun_out/${1:-r3s}; mkdir -p $out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/gpu_suite.txt 2>&1; rc=$?; echo "gpu suite rc=$rc"; tail -8 $out/gpu_suite.txt | cut -c1-400
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python bench/micro/nullable_projection.py 400000000 2>&1 | tail -2
timeout -k 10 260 python -m tests.fuzz_long 200 77 > $out/fuzz_200s.txt 2>&1; echo "fuzz rc=$?"; tail -2 $out/fuzz_200s.txt | cut -c1-500
