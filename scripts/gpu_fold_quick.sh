#!/bin/bash
# quick check of the Utf8 fold: its parity tests, then config 5 / reference-schema group timings
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_utf8_fold.py -x -q > gpurun_out/foldq_tests.log 2>&1
rc=$?
tail -3 gpurun_out/foldq_tests.log
[ $rc = 0 ] || exit $rc
rm -f gpurun_out/foldq.log
for c in "config5" "refgroup id%2=0, 12 500" $EXTRA_CASES; do
  timeout -k 10 300 python bench_configs.py --steps 9 --only "$c" $BENCH_OPTS >> gpurun_out/foldq.log 2>&1 || echo "FAILED $c" >> gpurun_out/foldq.log
done
grep -E "^\{|FAILED" gpurun_out/foldq.log | python3 -c "
import sys, json
for l in sys.stdin:
    try: j = json.loads(l)
    except Exception: print(l[:200]); continue
    if 'coalesced_call_ms' in j: print(j['case'][:40], 'coalesced_ms', round(j['coalesced_call_ms'], 3), 'group_call_ms', round(j['group_call_ms'], 2))
    else: print(j['case'][:40], 'kernel_ms', round(j['filter_kernel_ms'], 3), 'wall_ms', round(j['filter_wall_ms'], 3), 'launches', j['launches'], 'whole_frac', round(j['whole_filter_frac_of_8TBps'], 3))
"
