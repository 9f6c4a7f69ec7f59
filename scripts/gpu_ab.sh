#!/bin/bash
# A/B: the previous commit's library against the working tree's, same box, alternating
set -o pipefail
cd "$(dirname "$0")/.."
rm -f gpurun_out/ab.log
for rep in 1 2; do
for which in head new; do
  if [ $which = head ]; then export CHQ_LIB_PATH=$PWD/bench/ab/libchq_head.so; else unset CHQ_LIB_PATH; fi
  for c in ${AB_CASES:-"config2 value2>10" "config3 compound" "100k-row" "10k-row" "12 500 x 10k-row" "config5-shape"}; do
    echo "== $which | $c" >> gpurun_out/ab.log
    timeout -k 10 300 python bench_configs.py --steps 7 --only "$c" >> gpurun_out/ab.log 2>&1 || echo "FAILED" >> gpurun_out/ab.log
  done
done
done
grep -E "^==|kernel_ms|FAILED" gpurun_out/ab.log | python -c "
import sys,json
for l in sys.stdin:
    l=l.strip()
    if not l.startswith('{'): print(l); continue
    try:
        j=json.loads(l)
        if 'group_kernel_ms' in j: print('   group_kernel_ms', round(j['group_kernel_ms'],3), 'c_call_ms', round(j['c_call_ms'],2), 'coalesced_ms', round(j['coalesced_call_ms'],2))
        else: print('   kernel_ms', round(j['filter_kernel_ms'],3), 'wall', round(j['filter_wall_ms'],3), 'proj', j.get('project_wall_ms') and round(j.get('project_wall_ms'),2), 'onepass', j.get('filter_project_one_pass',{}).get('kernel_ms'))
    except Exception as e: print('?', l[:100])
"
