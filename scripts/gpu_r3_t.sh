#!/bin/bash
out=gpurun_out/${1:-r3t}; mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_gpu_threads.py -x -q 2>&1 | tail -3
timeout -k 10 280 rocprofv3 --kernel-trace -d $out/np -o r -- python3 bench/micro/nullable_projection.py 400000000 > $out/np.log 2>&1; tail -2 $out/np.log
python3 scripts/rocpd_summary.py $out/np chq:: > $out/nullable_kernel_stats.json
python3 -c "
import json
d=json.load(open('$out/nullable_kernel_stats.json'))
for k,v in sorted(d.items(), key=lambda kv:-kv[1]['avg_ms']*kv[1]['calls'])[:8]: print(k[:100], v['calls'], round(v['avg_ms'],4))
"
rm -rf $out/np
