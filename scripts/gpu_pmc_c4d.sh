#!/bin/bash
# how many bytes does the Utf8-predicate kernel fetch? (config 4d: value1 >= 'n' on 20 M 100-byte strings)
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/pmc_c4d
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 280 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/f -o r -- python3 bench_configs.py --only "config4d" --steps 3 > $OUT/f.log 2>&1 || echo "fetch pass failed"
timeout -k 10 280 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace -d $OUT/s -o r -- python3 bench_configs.py --only "config4d" --steps 3 > $OUT/s.log 2>&1 || echo "sq pass failed"
python3 scripts/rocpd_summary.py $OUT/f chq:: > $OUT/fetch.json
python3 scripts/rocpd_summary.py $OUT/s filter_fused > $OUT/sq.json
cat $OUT/fetch.json | python3 -c "
import json,sys
d=json.load(sys.stdin)
for k,v in d.items(): print(k[:100], v['calls'], 'median_ms', round(v['median_ms'],4), 'FETCH_KB', v['counters'].get('FETCH_SIZE'))
"
cat $OUT/sq.json
