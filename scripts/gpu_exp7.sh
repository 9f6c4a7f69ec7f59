#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fused.py tests/test_gpu_group.py -q -m gpu 2>&1 | tail -2
rm -f gpurun_out/exp7_bench.log
run() { echo "== $*" >> gpurun_out/exp7_bench.log; timeout -k 10 300 python bench_configs.py --steps 5 "$@" >> gpurun_out/exp7_bench.log 2>&1 || echo "FAILED rc=$?" >> gpurun_out/exp7_bench.log; }
run --only "config2 value2>10"
run --only "config3 compound" --opt tile_kind=0
run --only "config3 compound" --opt tile_kind=3
run --only "config3b"
run --only "10k-row"
grep -E "^==|filter_kernel_ms|group_kernel_ms|FAILED|differs" gpurun_out/exp7_bench.log | python -c "
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
for KIND in 0 3; do
OUT=gpurun_out/pmc_full_k$KIND; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 250 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH --kernel-trace -d $OUT/w -o r -- python3 bench_configs.py --only "config3 compound" --no-select --steps 1 --opt tile_kind=$KIND > $OUT/w.log 2>&1 || echo "pass failed"
python3 scripts/rocpd_summary.py $OUT/w "filter_fused_kernel<1024" | python3 -c "
import sys,json,re
j=json.load(sys.stdin)
for name,v in j.items():
    if re.search(r'<\d+, \d+, false, 0, false', name): print('kind $KIND', v['calls'], round(v['avg_ms'],3), {k:int(x*1.5) for k,x in v['counters'].items()})
"
done
