#!/bin/bash
out=gpurun_out/${1:-r3x}; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_utf8_fold.py tests/test_gpu_scale.py tests/test_gpu_parity.py -x -q > $out/t.txt 2>&1; rc=$?; echo "tests rc=$rc"; tail -8 $out/t.txt | cut -c1-400
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python bench.py --config 5 --steps 5 --warmup 2 2>/dev/null | python3 -c "import json,sys; j=json.load(sys.stdin); print('config5 shard', j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline']['kernel_ms'])"
timeout -k 10 300 python bench.py --config 5 --steps 5 --warmup 2 --opt uniform_utf8_rows=0 2>/dev/null | python3 -c "import json,sys; j=json.load(sys.stdin); print('config5 shard, specialisation off', j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline']['kernel_ms'])"
timeout -k 10 300 python bench_configs.py --only config5 --steps 5 2>&1 | grep -o '"filter_kernel_ms": [0-9.]*\|"call_ms": [0-9.]*' | head -3
