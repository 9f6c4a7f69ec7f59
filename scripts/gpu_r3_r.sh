#!/bin/bash
out=gpurun_out/${1:-r3r}; mkdir -p $out
timeout -k 10 500 python -m pytest tests/test_gpu_parquet.py -x -q > $out/tests.txt 2>&1; rc=$?; echo "parquet tests rc=$rc"; tail -8 $out/tests.txt | cut -c1-400
[ $rc -ne 0 ] && exit $rc
timeout -k 10 200 python bench/micro/parquet_scan.py 20000000 none sample 2>&1 | tail -4
timeout -k 10 900 python -m pytest tests/test_gpu_scale.py -x -q -k "rehearsed" > $out/rehearse.txt 2>&1; rc=$?; echo "rehearsal rc=$rc"; tail -15 $out/rehearse.txt | cut -c1-600
