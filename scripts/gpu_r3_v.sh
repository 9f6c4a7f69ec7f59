#!/bin/bash
out=gpurun_out/${1:-r3v}; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_parquet.py -x -q > $out/t.txt 2>&1; rc=$?; echo "parquet tests rc=$rc"; tail -8 $out/t.txt | cut -c1-600
