#!/bin/bash
# Utf8 fold: parity first, then fold on / off on the Utf8 configs (same library, option fold_utf8)
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_utf8_fold.py tests/test_gpu_parity.py tests/test_gpu_group.py -x -q > gpurun_out/fold_tests.log 2>&1
rc=$?
tail -5 gpurun_out/fold_tests.log
[ $rc = 0 ] || exit $rc
rm -f gpurun_out/fold_ab.log
for fold in 0 1; do
  for c in "config5" "config4b" "config4c" "refgroup id%2=0, 12 500"; do
    echo "== fold=$fold | $c" >> gpurun_out/fold_ab.log
    timeout -k 10 300 python bench_configs.py --steps 7 --only "$c" --opt fold_utf8=$fold >> gpurun_out/fold_ab.log 2>&1 || echo "FAILED" >> gpurun_out/fold_ab.log
  done
done
grep -E "^==|^\{|FAILED" gpurun_out/fold_ab.log | cut -c1-600
