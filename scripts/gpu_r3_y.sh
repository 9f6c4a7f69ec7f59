#!/bin/bash
# the batched snappy parser: parquet tests, then the scan bench on both compressed shapes
out=gpurun_out/${1:-r3y}; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_parquet.py -x -q > $out/t.txt 2>&1; rc=$?; echo "parquet tests rc=$rc"; tail -8 $out/t.txt | cut -c1-600
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python bench/micro/parquet_scan.py 20000000 snappy > $out/scan_snappy.txt 2>&1 && grep "chq scan\|pyarrow read" $out/scan_snappy.txt &&
timeout -k 10 300 python bench/micro/parquet_scan.py 20000000 snappy compressible > $out/scan_compressible.txt 2>&1 && grep "chq scan\|pyarrow read" $out/scan_compressible.txt
