#!/bin/bash
out=gpurun_out/${1:-r3w}; mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 280 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_SMEM --kernel-trace -d $out/p -o r -- python3 bench/micro/parquet_scan.py 20000000 snappy sample > $out/log.txt 2>&1
python3 scripts/rocpd_summary.py $out/p pq_inflate > $out/inflate_sq.json; cat $out/inflate_sq.json | head -30
rm -rf $out/p
