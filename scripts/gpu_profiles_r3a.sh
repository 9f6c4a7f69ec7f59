#!/bin/bash
# round-3 profile evidence, part A: headline (trace + FETCH_SIZE + WRITE_SIZE passes), config 3, config 2b counters.
# Every rocprofv3 pass is its own process; PMC passes are combined with --kernel-trace only.
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/profiles_r3; mkdir -p $OUT
HASH=$(python3 -c "import bench; print(bench.kernel_source_hash())")
sumj() { python3 scripts/rocpd_summary.py "$1" "$2"; }
pass() { local dir=$1; shift; local flags=$1; shift; timeout -k 10 280 rocprofv3 $flags --kernel-trace -d $OUT/$dir -o r -- "$@" > $OUT/$dir.log 2>&1 || echo "$dir failed"; echo "$dir done"; }
BENCH="python3 bench.py --no-cpu-baseline --no-extra --steps 10 --warmup 2 --validate-rows 0"
pass bench_trace "" $BENCH
sumj $OUT/bench_trace chq:: > $OUT/bench_kernel_stats.json
pass bench_fetch "--pmc FETCH_SIZE" $BENCH
pass bench_write "--pmc WRITE_SIZE" $BENCH
python3 - <<PY
import json, sys
sys.path.insert(0, "scripts")
from rocpd_summary import summarise
pick = "filter_fused_kernel<1024, 16, false, 0, false"
f = summarise("$OUT/bench_fetch", pick); w = summarise("$OUT/bench_write", pick); k = summarise("$OUT/bench_trace", pick)
name = next(iter(f))
fetch_kb = f[name]["counters"]["FETCH_SIZE"]; write_kb = w[next(iter(w))]["counters"]["WRITE_SIZE"]
out = {"kernel": name, "kernel_source_sha256": "$HASH", "rows": 1000000000, "calls": f[name]["calls"],
       "FETCH_SIZE_KB_per_launch": fetch_kb, "WRITE_SIZE_KB_per_launch": write_kb,
       "fetch_bytes_corrected": fetch_kb * 1024 * 2, "write_bytes": write_kb * 1024,
       "correction": "gfx950: FETCH_SIZE reports half the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM section): x2",
       "kernel_avg_ms_trace_pass": k[next(iter(k))]["avg_ms"], "kernel_median_ms_trace_pass": k[next(iter(k))]["median_ms"], "command": "$BENCH"}
json.dump(out, open("$OUT/bench_pmc_hbm.json", "w"), indent=1)
print(json.dumps(out))
PY
for cfg in "c3|compound|--no-select|32.5e9" "c2b|config2b||13.2e9"; do
  IFS='|' read -r tag only extra alg <<< "$cfg"
  CMD="python3 bench_configs.py --only $only $extra --steps 3"
  pass ${tag}_trace "" $CMD
  pass ${tag}_fetch "--pmc FETCH_SIZE" $CMD
  pass ${tag}_write "--pmc WRITE_SIZE" $CMD
  pass ${tag}_sq "--pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" $CMD
  python3 - <<PY
import json, sys
sys.path.insert(0, "scripts")
from rocpd_summary import summarise
pick = "filter_fused_kernel<1024, 16, false, 0, false"
f = summarise("$OUT/${tag}_fetch", pick); w = summarise("$OUT/${tag}_write", pick); k = summarise("$OUT/${tag}_trace", pick); q = summarise("$OUT/${tag}_sq", pick)
name = next(iter(f))
calls = f[name]["calls"]; big = calls - 1   # the extra dispatch is the 2 M-row validation prefix (0.2 % of the work)
fetch = f[name]["counters"]["FETCH_SIZE"] * calls / big; write = w[next(iter(w))]["counters"]["WRITE_SIZE"] * calls / big
out = {"kernel": name, "kernel_source_sha256": "$HASH", "dispatches_per_pass": calls,
       "note": "per-launch figures = pass total / %d full-size launches (the extra dispatch is the 2 M-row validation prefix)" % big,
       "fetch_bytes_corrected": fetch * 1024 * 2, "write_bytes": write * 1024, "traffic_bytes": fetch * 2048 + write * 1024,
       "algorithmic_bytes": $alg, "kernel_median_ms": k[next(iter(k))]["median_ms"], "kernel_durations_ms": k[next(iter(k))].get("durations_ms"),
       "sq_per_launch": {c: v * calls / big for c, v in q[next(iter(q))]["counters"].items()}, "command": "$CMD"}
json.dump(out, open("$OUT/${tag}_pmc.json", "w"), indent=1)
print(json.dumps(out))
PY
  sumj $OUT/${tag}_trace chq:: > $OUT/${tag}_kernel_stats.json
done
rm -rf $OUT/*/r_results.db $OUT/*_trace $OUT/*_fetch $OUT/*_write $OUT/*_sq 2>/dev/null
ls $OUT
