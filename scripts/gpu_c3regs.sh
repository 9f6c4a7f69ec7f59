#!/bin/bash
# round 3, VERDICT item 4: the register-resident config-3 experiment (bench/micro/config3_regs.hip) next to the product kernel
out=gpurun_out/${1:-r3c3}; mkdir -p $out
[ -x bench/micro/config3_regs ] || /opt/rocm/bin/hipcc -O3 -std=c++17 -ffp-contract=off -Wno-unused-function --offload-arch=gfx950 -Ichapterhouseqe_amd/csrc bench/micro/config3_regs.hip -o bench/micro/config3_regs
timeout -k 10 120 ./bench/micro/config3_regs 1000000000 > $out/config3_regs.txt 2>&1; echo "rc=$?" >> $out/config3_regs.txt
cat $out/config3_regs.txt
timeout -k 10 200 python bench_configs.py --steps 7 --only "config3 compound" --no-select > $out/config3_product.txt 2>&1
grep -o '"filter_kernel_ms": [0-9.]*' $out/config3_product.txt
