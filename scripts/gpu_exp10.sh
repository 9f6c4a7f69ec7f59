#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
python -m pytest tests/test_gpu_parity.py tests/test_gpu_group.py -q -m gpu 2>&1 | tail -2
rm -f gpurun_out/exp10.log
for rep in 1 2; do
for which in head new new1; do
  unset CHQ_LIB_PATH; OPT=""
  if [ $which = head ]; then export CHQ_LIB_PATH=$PWD/bench/ab/libchq_head.so; fi
  if [ $which = new1 ]; then OPT="--opt utf8_variant=1"; fi
  for c in "config5" "config4 wide" "config4b" "config4d"; do
    echo "== $which | $c" >> gpurun_out/exp10.log
    timeout -k 10 300 python bench_configs.py --steps 5 --only "$c" $OPT >> gpurun_out/exp10.log 2>&1 || echo "FAILED" >> gpurun_out/exp10.log
  done
done
done
grep -E "^==|kernel_ms|FAILED" gpurun_out/exp10.log | python -c "
import sys,json
for l in sys.stdin:
    l=l.strip()
    if not l.startswith('{'): print(l); continue
    try:
        j=json.loads(l)
        print('   kernel_ms', round(j['filter_kernel_ms'],3), 'wall', round(j['filter_wall_ms'],3), 'whole GBps', round(j['whole_filter_GBps']), 'frac', round(j['whole_filter_frac_of_8TBps'],3))
    except Exception as e: print('?', l[:100])
"
