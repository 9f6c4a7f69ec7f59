#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
python -m pytest tests/test_gpu_group.py tests/test_gpu_operators.py -q -m gpu 2>&1 | tail -8
rm -f gpurun_out/exp9_bench.log
timeout -k 10 600 python bench_configs.py --steps 3 --only "refgroup" >> gpurun_out/exp9_bench.log 2>&1 || echo "FAILED rc=$?" >> gpurun_out/exp9_bench.log
timeout -k 10 300 python bench_configs.py --steps 3 --only "config5" >> gpurun_out/exp9_bench.log 2>&1
timeout -k 10 300 python bench_configs.py --steps 3 --only "10k-row" >> gpurun_out/exp9_bench.log 2>&1
grep -E "^\{|FAILED|Error|error" gpurun_out/exp9_bench.log | cut -c1-900
