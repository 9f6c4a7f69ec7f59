#!/bin/bash
# final evidence of the round on the final build: the GPU suite, smoke(), a long fuzz run, the Parquet scan rates
out=gpurun_out/${1:-r3final}; mkdir -p $out
timeout -k 10 900 python -m pytest tests -m gpu -q > $out/gpu_suite.txt 2>&1; echo "gpu suite rc=$?"; tail -3 $out/gpu_suite.txt | cut -c1-300
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.txt 2>&1; echo "smoke rc=$?"; tail -2 $out/smoke.txt | cut -c1-300
{ for args in "20000000 none sample" "20000000 snappy sample" "20000000 snappy compressible"; do echo "== parquet_scan.py $args"; timeout -k 10 200 python bench/micro/parquet_scan.py $args 2>&1 | grep -v amdgpu.ids | tail -9; done; } > $out/parquet_scan.txt; grep -c "pyarrow read_table" $out/parquet_scan.txt
timeout -k 10 ${2:-620} python -m tests.fuzz_long ${3:-600} ${4:-303} > $out/fuzz_a.txt 2>&1; echo "fuzz rc=$?"; tail -1 $out/fuzz_a.txt | cut -c1-500
