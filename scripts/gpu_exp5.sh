#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
rm -f gpurun_out/exp5_bench.log
run() { echo "== $*" >> gpurun_out/exp5_bench.log; timeout -k 10 300 python bench_configs.py --steps 5 "$@" >> gpurun_out/exp5_bench.log 2>&1 || echo "FAILED rc=$?" >> gpurun_out/exp5_bench.log; }
run --only "config3 compound" --opt tile_kind=0 --opt kflags=8 --opt fuse=0
run --only "config3 compound" --opt tile_kind=3 --opt kflags=8 --opt fuse=0
run --only "config3 compound" --opt tile_kind=3 --opt kflags=8 --opt stash=0
run --only "config3 compound" --opt tile_kind=3 --opt kflags=9
grep -E "^==|filter_kernel_ms|group_kernel_ms|FAILED|differs" gpurun_out/exp5_bench.log | python -c "
import sys,json
for l in sys.stdin:
    l=l.strip()
    if not l.startswith('{'): print(l); continue
    try:
        j=json.loads(l)
        print('   ', j['case'], 'kernel_ms', round(j['filter_kernel_ms'],3), 'GBps', round(j['fused_kernel_GBps']), 'frac', round(j['fused_kernel_frac_of_8TBps'],3))
    except Exception as e: print('?', l[:100])
"
