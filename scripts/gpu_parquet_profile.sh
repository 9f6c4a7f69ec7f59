#!/bin/bash
# kernel trace of the Parquet scan micro-benchmark
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/parquet_prof; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 400 rocprofv3 --kernel-trace --memory-copy-trace -d $OUT/t -o r -- python3 bench/micro/parquet_scan.py ${1:-20000000} ${2:-none} ${3:-sample} > $OUT/run.log 2>&1 || echo "trace failed"
tail -4 $OUT/run.log
python3 scripts/rocpd_summary.py $OUT/t pq_ > $OUT/kernels.json
python3 - <<PY
import json, sqlite3, glob
d = json.load(open("$OUT/kernels.json"))
tot = 0
for k, v in sorted(d.items(), key=lambda kv: -kv[1]["avg_ms"] * kv[1]["calls"]):
    print(f"{k[:60]:60s} calls {v['calls']:5d} avg {v['avg_ms']*1e3:9.1f} us total {v['avg_ms']*v['calls']:8.2f} ms")
    tot += v["avg_ms"] * v["calls"]
print("all pq kernels:", round(tot, 2), "ms over 5 scans (4 timed + parity)")
db = sqlite3.connect(glob.glob("$OUT/t/**/*.db", recursive=True)[0])
try:
    rows = db.execute("select name, count(*), sum(size), sum(duration) from memory_copies group by name").fetchall()
    for r in rows: print("memcpy", r[0], "n", r[1], "MB", round((r[2] or 0)/1e6,1), "ms", round((r[3] or 0)/1e6,2))
except Exception as e:
    print("no memory copy table:", e)
PY
