#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for KIND in 0 3; do
OUT=gpurun_out/pmc_full_k$KIND; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 250 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH --kernel-trace -d $OUT/w -o r -- python3 bench_configs.py --only "config3 compound" --no-select --steps 1 --opt tile_kind=$KIND "$@" > $OUT/w.log 2>&1 || echo "pass failed"
python3 scripts/rocpd_summary.py $OUT/w "filter_fused_kernel<1024" | python3 -c "
import sys,json,re
j=json.load(sys.stdin)
for name,v in j.items():
    if re.search(r'<\d+, \d+, false, 0, false', name):
        tiles={'0':61035,'3':122070}['$KIND']; wt=4096*tiles/256
        print('kind $KIND', v['calls'], round(v['avg_ms'],3), {k:round(x*1.5/wt,1) for k,x in v['counters'].items()})
"
done
