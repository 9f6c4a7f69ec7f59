#!/bin/bash
# round-2 profile evidence: kernel traces and PMC passes (separate passes, never combined with other trace domains)
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/profiles_r2; rm -rf $OUT; mkdir -p $OUT
HASH=$(python3 -c "import bench; print(bench.kernel_source_hash())")
sumj() { python3 scripts/rocpd_summary.py "$1" "$2"; }
BENCH="python3 bench.py --no-cpu-baseline --no-extra --steps 10 --warmup 2 --validate-rows 0"
# 1. headline: kernel trace, then FETCH_SIZE and WRITE_SIZE in their own passes
timeout -k 10 280 rocprofv3 --kernel-trace -d $OUT/bench_trace -o r -- $BENCH > $OUT/bench_trace.log 2>&1 || echo "bench trace failed"
sumj $OUT/bench_trace chq:: > $OUT/bench_kernel_stats.json
timeout -k 10 280 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/bench_fetch -o r -- $BENCH > $OUT/bench_fetch.log 2>&1 || echo "bench fetch failed"
timeout -k 10 280 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/bench_write -o r -- $BENCH > $OUT/bench_write.log 2>&1 || echo "bench write failed"
python3 - <<PY
import json, sys
sys.path.insert(0, "scripts")
from rocpd_summary import summarise
f = summarise("$OUT/bench_fetch", "filter_fused_kernel<1024, 16, false, 0, false")
w = summarise("$OUT/bench_write", "filter_fused_kernel<1024, 16, false, 0, false")
k = summarise("$OUT/bench_trace", "filter_fused_kernel<1024, 16, false, 0, false")
name = next(iter(f))
fetch_kb = f[name]["counters"]["FETCH_SIZE"]; write_kb = w[next(iter(w))]["counters"]["WRITE_SIZE"]
out = {"kernel": name, "kernel_source_sha256": "$HASH", "calls": f[name]["calls"],
       "FETCH_SIZE_KB_per_launch": fetch_kb, "WRITE_SIZE_KB_per_launch": write_kb,
       "fetch_bytes_corrected": fetch_kb * 1024 * 2, "write_bytes": write_kb * 1024,
       "correction": "gfx950: FETCH_SIZE reports half the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM section): x2",
       "kernel_avg_ms_trace_pass": k[next(iter(k))]["avg_ms"], "command": "$BENCH"}
json.dump(out, open("$OUT/bench_pmc_hbm.json", "w"), indent=1)
print(json.dumps(out))
PY
# 2. config 3
C3='python3 bench_configs.py --only compound --no-select --steps 3'
timeout -k 10 280 rocprofv3 --kernel-trace -d $OUT/c3_trace -o r -- $C3 > $OUT/c3_trace.log 2>&1 || echo "c3 trace failed"
timeout -k 10 280 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/c3_fetch -o r -- $C3 > $OUT/c3_fetch.log 2>&1 || echo "c3 fetch failed"
timeout -k 10 280 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/c3_write -o r -- $C3 > $OUT/c3_write.log 2>&1 || echo "c3 write failed"
timeout -k 10 280 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY --kernel-trace -d $OUT/c3_sq -o r -- $C3 > $OUT/c3_sq.log 2>&1 || echo "c3 sq failed"
python3 - <<PY
import json, sys
sys.path.insert(0, "scripts")
from rocpd_summary import summarise
pick = "filter_fused_kernel<1024, 16, false, 0, false"
f = summarise("$OUT/c3_fetch", pick); w = summarise("$OUT/c3_write", pick); k = summarise("$OUT/c3_trace", pick); q = summarise("$OUT/c3_sq", pick)
name = next(iter(f))
# 4 dispatches per pass: 3 x 1e9 rows + the 2 M-row validation launch (0.2 % of the work): per-launch = sum / 3
calls = f[name]["calls"]; big = calls - 1
fetch = f[name]["counters"]["FETCH_SIZE"] * calls / big; write = w[next(iter(w))]["counters"]["WRITE_SIZE"] * calls / big
out = {"kernel": name, "kernel_source_sha256": "$HASH", "dispatches_per_pass": calls, "note": "per-launch figures = pass total / %d full-size launches (the extra dispatch is the 2 M-row validation prefix)" % big,
       "fetch_bytes_corrected": fetch * 1024 * 2, "write_bytes": write * 1024, "traffic_bytes": fetch * 2048 + write * 1024,
       "algorithmic_bytes": 32.5e9, "kernel_median_ms": k[next(iter(k))]["median_ms"], "kernel_durations_ms": k[next(iter(k))].get("durations_ms"),
       "sq_per_launch": {c: v * calls / big for c, v in q[next(iter(q))]["counters"].items()}, "command": "$C3"}
json.dump(out, open("$OUT/config3_pmc.json", "w"), indent=1)
print(json.dumps(out))
PY
sumj $OUT/c3_trace chq:: > $OUT/config3_kernel_stats.json
# 3. reference-schema group, Utf8 configs
timeout -k 10 280 rocprofv3 --kernel-trace -d $OUT/grp_trace -o r -- python3 bench_configs.py --only "refgroup id%2=0, 12 500" --steps 3 > $OUT/grp_trace.log 2>&1 || echo "grp trace failed"
sumj $OUT/grp_trace chq:: > $OUT/refgroup_kernel_stats.json
for c in config5 "config4 wide"; do
  n=$(echo $c | tr -d ' ')
  timeout -k 10 280 rocprofv3 --kernel-trace -d $OUT/${n}_trace -o r -- python3 bench_configs.py --only "$c" --steps 3 > $OUT/${n}_trace.log 2>&1 || echo "$n trace failed"
  sumj $OUT/${n}_trace chq:: > $OUT/${n}_kernel_stats.json
  timeout -k 10 280 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY --kernel-trace -d $OUT/${n}_sq -o r -- python3 bench_configs.py --only "$c" --steps 3 > $OUT/${n}_sq.log 2>&1 || echo "$n sq failed"
  sumj $OUT/${n}_sq chq:: > $OUT/${n}_sq.json
done
rm -rf $OUT/*/r_results.db $OUT/*_trace $OUT/*_fetch $OUT/*_write $OUT/*_sq 2>/dev/null
ls -la $OUT
