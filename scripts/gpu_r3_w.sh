#!/bin/bash
out=gpurun_out/${1:-r3w}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 400 rocprofv3 --kernel-trace -d $out/prof -o pages -- python3 bench/micro/snappy_pages.py > $out/run.txt 2>&1 || exit 1
grep -v "^W2026\|^E2026" $out/run.txt
