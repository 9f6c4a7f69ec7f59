#!/bin/bash
out=gpurun_out/${1:-r3n}; mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $out/prof -o nulls -- python3 bench_configs.py --steps 5 --only "with nulls" > $out/run.txt 2>&1; echo "rc=$?"
f=$(find $out/prof -name "*kernel_stats.csv" | head -1)
head -12 "$f" | cut -c1-220
