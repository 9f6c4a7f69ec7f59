#!/bin/bash
# HBM traffic of the one-launch group kernel with Utf8 columns (reference schema, 12 500 device batches)
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/pmc_refgroup; rm -rf $OUT; mkdir -p $OUT
CMD='python3 bench_configs.py --only "refgroup id%2=0, 12 500" --steps 3'
timeout -k 10 280 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/f -o r -- python3 bench_configs.py --only "refgroup id%2=0, 12 500" --steps 3 > $OUT/f.log 2>&1 || echo "fetch failed"
timeout -k 10 280 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/w -o r -- python3 bench_configs.py --only "refgroup id%2=0, 12 500" --steps 3 > $OUT/w.log 2>&1 || echo "write failed"
python3 - <<PY
import json, sys
sys.path.insert(0, "scripts")
from rocpd_summary import summarise
pick = "filter_fused_kernel<1024, 16, false, 0, true, 1, true, 2>"
f = summarise("$OUT/f", pick); w = summarise("$OUT/w", pick)
name = next(iter(f))
out = {"kernel": name, "what": "12 500 x 10 000-row device batches of (Int32, Utf8(8), Float32), WHERE id % 2 = 0: the group kernel reads the batches as they lie",
       "dispatches": f[name]["calls"], "median_ms": f[name]["median_ms"],
       "fetch_bytes_corrected_per_launch": f[name]["counters"]["FETCH_SIZE"] * 1024 * 2, "write_bytes_per_launch": w[next(iter(w))]["counters"]["WRITE_SIZE"] * 1024,
       "algorithmic_bytes": {"read": 125e6 * (4 + 4 + 8) + 62.5e6 * 8, "written": 62.5e6 * (4 + 4 + 4 + 8)},
       "note": "offsets are read by both phases (8 B/row); every 128-byte line of the string bytes is touched although half the rows are selected (8 B/row fetched for 4 B/row counted)"}
json.dump(out, open("$OUT/refgroup_pmc.json", "w"), indent=1)
print(json.dumps(out))
PY
