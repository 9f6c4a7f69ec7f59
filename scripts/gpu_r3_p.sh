#!/bin/bash
out=gpurun_out/${1:-r3p}; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_c_example.py -x -q > $out/tests.txt 2>&1; rc=$?; echo "tests rc=$rc"; tail -6 $out/tests.txt | cut -c1-400
[ $rc -ne 0 ] && exit $rc
timeout -k 10 400 python -m tests.fuzz_long 330 ${2:-31} > $out/fuzz.txt 2>&1; echo "fuzz rc=$?"; tail -4 $out/fuzz.txt | cut -c1-600
