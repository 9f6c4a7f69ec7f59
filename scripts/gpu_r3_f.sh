#!/bin/bash
# per-launch durations of the inflate kernel on ONE row group (1 Mi rows of the sample shape: three launches) + the 20 M-row scans
out=gpurun_out/${1:-r3f}; mkdir -p $out; export TMPDIR=/tmp
rm -rf $out/t; timeout -k 10 300 rocprofv3 --kernel-trace -d $out/t -o r -- python3 bench/micro/parquet_scan.py 1048576 snappy sample > $out/run1.log 2>&1
python3 - <<PY
import sqlite3, glob
db = sqlite3.connect(glob.glob("$out/t/**/*.db", recursive=True)[0])
rows = db.execute("select name, duration from kernels where name like '%inflate%' order by start").fetchall()
print("inflate launches (us):", [round(r[1] / 1e3) for r in rows][:12])
PY
for a in "snappy sample" "snappy compressible"; do
  timeout -k 10 200 python bench/micro/parquet_scan.py 20000000 $a > $out/pq_$(echo $a | tr ' ' '_').txt 2>&1; grep "chq scan" $out/pq_$(echo $a | tr ' ' '_').txt
done
bash scripts/gpu_c3regs.sh $(basename $out) | grep -v "^$"
