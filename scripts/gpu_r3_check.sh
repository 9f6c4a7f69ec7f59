#!/bin/bash
# round 3: GPU suite + a short fuzz + the bench line on the current build; everything lands under gpurun_out/$1
out=gpurun_out/${1:-r3}
mkdir -p $out
timeout -k 10 500 python -m pytest tests -m gpu -x -q > $out/pytest.txt 2>&1; echo "pytest rc=$?" | tee -a $out/pytest.txt
tail -3 $out/pytest.txt
timeout -k 10 ${FUZZ_LIMIT:-330} python -m tests.fuzz_long ${FUZZ_SECS:-240} ${FUZZ_SEED:-3101} > $out/fuzz.txt 2>&1; echo "fuzz rc=$?" | tee -a $out/fuzz.txt
tail -4 $out/fuzz.txt
timeout -k 10 300 python bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"
python - <<PY
import json
j=json.loads(open("$out/bench.json").read().strip().splitlines()[-1])
print("headline", j["value"], j["ms_per_step"], j["roofline"]["frac"], j["roofline"]["kernel_ms"])
for k,v in j.get("extra",{}).items(): print(k, {a:(round(b,4) if isinstance(b,float) else b) for a,b in v.items() if a!="workload"})
PY
