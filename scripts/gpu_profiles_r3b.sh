#!/bin/bash
# round-3 profile evidence, part B: kernel traces of the group calls (with and without nulls), the config-5 shard, the Utf8
# configs and the SNAPPY Parquet scan
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/profiles_r3; mkdir -p $OUT
sumj() { python3 scripts/rocpd_summary.py "$1" "$2"; }
pass() { local dir=$1; shift; local flags=$1; shift; timeout -k 10 280 rocprofv3 $flags --kernel-trace -d $OUT/$dir -o r -- "$@" > $OUT/$dir.log 2>&1 || echo "$dir failed"; echo "$dir done"; }
pass grp_trace "" python3 bench_configs.py --only "refgroup id%2=0, 12 500" --steps 3
sumj $OUT/grp_trace chq:: > $OUT/refgroup_kernel_stats.json
pass grpn_trace "" python3 bench_configs.py --only "refgroup with nulls" --steps 3
sumj $OUT/grpn_trace chq:: > $OUT/refgroup_nulls_kernel_stats.json
pass c5shard_trace "" python3 bench.py --config 5 --steps 5 --warmup 2 --no-cpu-baseline --no-extra --validate-rows 0
sumj $OUT/c5shard_trace chq:: > $OUT/config5_shard_kernel_stats.json
for c in config5 "config4 wide"; do
  n=$(echo $c | tr -d ' ')
  pass ${n}_trace "" python3 bench_configs.py --only "$c" --steps 3
  sumj $OUT/${n}_trace chq:: > $OUT/${n}_kernel_stats.json
done
pass pq_snappy_trace "" python3 bench/micro/parquet_scan.py 20000000 snappy sample
sumj $OUT/pq_snappy_trace pq_ > $OUT/parquet_snappy_kernel_stats.json
rm -rf $OUT/*/r_results.db $OUT/*_trace 2>/dev/null
ls $OUT
