#!/bin/bash
# PMC passes of config 3 per tile kind: instruction mix and wait cycles of the filter kernel
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/pmc_c3
rm -rf $OUT; mkdir -p $OUT
for kind in 0 3; do
  i=0
  for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH" \
             "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INST_CYCLES_SALU" \
             "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    timeout -k 10 280 rocprofv3 --pmc $set --kernel-trace -d $OUT/k${kind}_s$i -o r -- python3 bench_configs.py --only "config3 compound" --steps 2 --opt tile_kind=$kind --opt kflags=1 > $OUT/k${kind}_s$i.log 2>&1 || echo "pass k$kind s$i failed rc=$?"
  done
  python3 scripts/rocpd_summary.py $OUT filter_fused > /dev/null
done
for kind in 0 3; do for i in 1 2 3 4; do echo "== kind $kind set $i"; python3 scripts/rocpd_summary.py $OUT/k${kind}_s$i filter_fused_kernel; done; done > $OUT/summary.txt 2>&1
tail -c 6000 $OUT/summary.txt
