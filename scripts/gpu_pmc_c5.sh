#!/bin/bash
# PMC passes of the config-5 shape (Utf8 filtered inside the main kernel): instruction mix, wait cycles, HBM bytes
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/pmc_c5
rm -rf $OUT; mkdir -p $OUT
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INST_CYCLES_SALU" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 280 rocprofv3 --pmc $set --kernel-trace -d $OUT/s$i -o r -- python3 bench_configs.py --only "config5" --steps 2 > $OUT/s$i.log 2>&1 || echo "pass s$i failed rc=$?"
done
for i in 1 2 3 4; do echo "== set $i"; python3 scripts/rocpd_summary.py $OUT/s$i filter_fused_kernel; done > $OUT/summary.txt 2>&1
tail -c 5000 $OUT/summary.txt
