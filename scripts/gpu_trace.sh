#!/bin/bash
# kernel trace of one bench_configs case: per-kernel average durations (rocprofv3 --kernel-trace)
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
CASE="$1"; NAME="$2"; shift 2
OUT=gpurun_out/trace_$NAME; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 280 rocprofv3 --kernel-trace -d $OUT/t -o r -- python3 bench_configs.py --only "$CASE" --steps 3 "$@" > $OUT/run.log 2>&1 || echo "trace failed rc=$?"
python3 scripts/rocpd_summary.py $OUT/t | python3 -c "
import sys,json
j=json.load(sys.stdin)
rows=sorted(j.items(), key=lambda kv: -(kv[1]['avg_ms'] or 0)*kv[1]['calls'])
for name,v in rows[:14]: print(f\"{v['calls']:5d} calls  avg {v['avg_ms']:.4f} ms  {name[:110]}\")
" | tee $OUT/summary.txt
