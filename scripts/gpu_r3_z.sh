#!/bin/bash
# kernel trace of the snappy scan (per-launch durations of pq_inflate_kernel: index / blocks / finish)
out=gpurun_out/${1:-r3z}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $out/prof -o scan -- python3 bench/micro/parquet_scan.py 8000000 snappy > $out/run.txt 2>&1
echo rc=$?
python3 - $out <<'PY'
import csv, glob, sys, collections
out=sys.argv[1]
f=glob.glob(out+"/prof/**/*kernel_trace.csv", recursive=True)
rows=list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
agg=collections.defaultdict(list)
for r in rows: agg[r["Kernel_Name"][:60]].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6)
for k,v in sorted(agg.items(), key=lambda kv:-sum(kv[1]))[:12]:
    print(f"{k:60s} n={len(v):5d} total={sum(v):9.2f} ms max={max(v):8.3f}")
inf=[(int(r["Start_Timestamp"]),(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6,r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size","")) for r in rows if "inflate" in r["Kernel_Name"]]
print("inflate launches (ms, grid):", [(round(d,2),g) for _,d,g in inf[:12]])
PY
