#!/bin/bash
out=gpurun_out/${1:-r3suite}; mkdir -p $out
timeout -k 10 900 python -m pytest tests -m gpu -q > $out/gpu_suite.txt 2>&1; echo "gpu suite rc=$?"; tail -3 $out/gpu_suite.txt | cut -c1-300
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.txt 2>&1; echo "smoke rc=$?"; tail -2 $out/smoke.txt | cut -c1-300
