#!/bin/bash
# instruction cost model of the interpreter: PMC instruction counts of the filter kernel for growing predicates on the
# config-3 table (5 x 4-byte columns, 1e9 rows), tile kind given as $1
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
KIND=${1:-0}
OUT=gpurun_out/pmc_terms_k$KIND
rm -rf $OUT; mkdir -p $OUT
i=0
for where in "e > 1.0" "e > c" "b + e > c" "a + b > c" "a + b > c and d < 5.0" "a + b > c and d < 5.0 or e > 1.0"; do
  i=$((i+1))
  timeout -k 10 250 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH --kernel-trace -d $OUT/w$i -o r -- python3 bench_configs.py --only "config3 compound" --no-select --steps 1 --opt tile_kind=$KIND --where "$where" > $OUT/w$i.log 2>&1 || echo "pass $i failed"
  echo "== $where" >> $OUT/summary.txt
  python3 scripts/rocpd_summary.py $OUT/w$i "filter_fused_kernel<1024" >> $OUT/summary.txt
done
python3 - <<'PY'
import re,json,sys,os
kind=os.environ.get('KIND','0')
PY
cat $OUT/summary.txt | python3 -c "
import sys,re,json
txt=sys.stdin.read()
parts=re.split(r'== (.*)\n', txt)
for i in range(1,len(parts),2):
    where, body = parts[i], parts[i+1]
    j=json.loads(body)
    for name,v in j.items():
        if ', false, ' in name.split('<')[1][:40] and re.search(r'<\d+, \d+, false, 0, false', name):
            c=v['counters']; calls=v['calls']
            print(where, '| calls',calls,'avg_ms',round(v['avg_ms'],3), {k:int(x) for k,x in c.items()})
"
