#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m pytest tests -q -m gpu --deselect tests/test_gpu_scale.py --deselect tests/test_gpu_config3.py::test_config3_full_size_properties 2>&1 | tail -4
rm -f gpurun_out/exp6_bench.log
run() { echo "== $*" >> gpurun_out/exp6_bench.log; timeout -k 10 300 python bench_configs.py --steps 5 "$@" >> gpurun_out/exp6_bench.log 2>&1 || echo "FAILED rc=$?" >> gpurun_out/exp6_bench.log; }
run --only "config2 value2>10"
run --only "config3 compound"
run --only "config3 compound" --opt tile_kind=0
run --only "config3 compound" --opt tile_kind=3 --opt kflags=0
run --only "config3 compound" --opt tile_kind=3 --opt stash=2
run --only "config3b"
run --only "10k-row"
grep -E "^==|filter_kernel_ms|group_kernel_ms|FAILED|differs" gpurun_out/exp6_bench.log | python -c "
import sys,json
for l in sys.stdin:
    l=l.strip()
    if not l.startswith('{'): print(l); continue
    try:
        j=json.loads(l)
        if 'group_kernel_ms' in j: print('   ', j['case'], 'group_kernel_ms', round(j['group_kernel_ms'],3), 'frac', round(j['group_kernel_frac_of_8TBps'],3), 'c_call_ms', round(j['c_call_ms'],2), 'coalesced_ms', round(j['coalesced_call_ms'],2))
        else: print('   ', j['case'], 'kernel_ms', round(j['filter_kernel_ms'],3), 'GBps', round(j['fused_kernel_GBps']), 'frac', round(j['fused_kernel_frac_of_8TBps'],3), 'proj', j.get('project_wall_ms'), 'onepass', j.get('filter_project_one_pass',{}).get('kernel_ms'))
    except Exception as e: print('?', l[:100])
"
