#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_group.py tests/test_gpu_fused.py -q -m gpu 2>&1 | tail -3
rm -f gpurun_out/exp3_bench.log
run() { echo "== $*" >> gpurun_out/exp3_bench.log; timeout -k 10 300 python bench_configs.py --steps 5 "$@" >> gpurun_out/exp3_bench.log 2>&1 || echo "FAILED rc=$?" >> gpurun_out/exp3_bench.log; }
run --only "config2 value2>10"
run --only "config2 value2>10" --opt kflags=1
run --only "config2 value2>10" --opt kflags=2
run --only "config2b"
run --only "config3 compound" --opt tile_kind=0
run --only "config3 compound" --opt tile_kind=3
run --only "10k-row"
grep -E "^==|filter_kernel_ms|group_kernel_ms|FAILED" gpurun_out/exp3_bench.log | python -c "
import sys,json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('==') or l.startswith('FAILED'): print(l); continue
    try:
        j=json.loads(l)
        if 'group_kernel_ms' in j: print('   ', j['case'], 'group_kernel_ms', round(j['group_kernel_ms'],3), 'frac', round(j['group_kernel_frac_of_8TBps'],3), 'c_call_ms', round(j['c_call_ms'],2), 'coalesced_ms', round(j['coalesced_call_ms'],2))
        else: print('   ', j['case'], 'kernel_ms', round(j['filter_kernel_ms'],3), 'GBps', round(j['fused_kernel_GBps']), 'frac', round(j['fused_kernel_frac_of_8TBps'],3))
    except Exception as e: print('?', l[:200])
"
bash scripts/gpu_pmc_c3.sh
