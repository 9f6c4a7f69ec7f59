#!/bin/bash
# round-2 experiment 1: correctness of the restructured filter kernel, then config 2 / 3 timings per tile kind / stash count
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m pytest tests -q -m gpu --deselect tests/test_gpu_scale.py --deselect tests/test_gpu_config3.py::test_config3_full_size_properties > gpurun_out/exp1_tests.log 2>&1
echo "tests rc=$?" | tee -a gpurun_out/exp1_tests.log
tail -5 gpurun_out/exp1_tests.log
for opts in "" "--opt stash=0" ; do
  echo "== config2 $opts" >> gpurun_out/exp1_bench.log
  timeout -k 10 300 python bench_configs.py --only "config2 value2>10" --steps 5 $opts >> gpurun_out/exp1_bench.log 2>&1
done
for opts in "--opt tile_kind=0 --opt stash=1" "--opt tile_kind=0" "--opt tile_kind=3" "--opt tile_kind=3 --opt stash=2" "--opt tile_kind=3 --opt stash=0"; do
  echo "== config3 $opts" >> gpurun_out/exp1_bench.log
  timeout -k 10 300 python bench_configs.py --only "config3 compound" --steps 5 $opts >> gpurun_out/exp1_bench.log 2>&1
done
grep -E "^==|filter_kernel_ms" gpurun_out/exp1_bench.log | python -c "
import sys,json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('=='): print(l); continue
    try:
        j=json.loads(l); print('   ', j['case'], 'kernel_ms', round(j['filter_kernel_ms'],3), 'GBps', round(j['fused_kernel_GBps']), 'frac', round(j['fused_kernel_frac_of_8TBps'],3), 'one_pass', j.get('filter_project_one_pass',{}).get('kernel_ms'))
    except Exception as e: print('?', l[:200])
"
