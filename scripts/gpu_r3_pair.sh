#!/bin/bash
# the bench line and the kernel trace of the same command on ONE box (box-to-box spread is ~3 %)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/profiles_r3; mkdir -p $OUT
timeout -k 10 600 python bench.py > $OUT/bench_line.json 2> $OUT/bench_line.err; echo "bench rc=$?"
BENCH="python3 bench.py --no-cpu-baseline --no-extra --steps 10 --warmup 2 --validate-rows 0"
timeout -k 10 280 rocprofv3 --kernel-trace -d $OUT/bench_trace -o r -- $BENCH > $OUT/bench_trace.log 2>&1 || echo "trace failed"
python3 scripts/rocpd_summary.py $OUT/bench_trace chq:: > $OUT/bench_kernel_stats.json
rm -rf $OUT/bench_trace
python3 - <<PY
import json
j = json.load(open("$OUT/bench_line.json")); k = json.load(open("$OUT/bench_kernel_stats.json"))
name = [n for n in k if "false, 0, false" in n][0]
print("bench.py kernel_ms", j["roofline"]["kernel_ms"], "frac", j["roofline"]["frac"], "| trace avg", k[name]["avg_ms"], "median", k[name]["median_ms"], "calls", k[name]["calls"])
PY
