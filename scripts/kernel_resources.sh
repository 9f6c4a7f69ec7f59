#!/bin/bash
# VGPR / SGPR / LDS / scratch of every instantiation whose name contains $1 (default: filter_fused), from the built object
cd "$(dirname "$0")/.."
want=${1:-filter_fused}
tmp=$(mktemp -d)
for o in chapterhouseqe_amd/lib/kernels_*.o; do
  cp $o $tmp/k.o
  (cd $tmp && /opt/rocm/lib/llvm/bin/llvm-objdump --offloading k.o > /dev/null 2>&1)
  /opt/rocm/lib/llvm/bin/llvm-readelf --notes $tmp/k.o.0.hipv4-amdgcn-amd-amdhsa--gfx950
done | grep -E "^\s+\.name:|\.vgpr_count|\.private_segment_fixed_size|\.sgpr_count|\.vgpr_spill_count|\.group_segment_fixed_size" | paste - - - - - - | grep "$want" | sed -e 's/_ZN3chq//' | awk '{printf "%s lds=%s scratch=%s sgpr=%s vgpr=%s spill=%s\n",$4,$2,$6,$8,$10,$12}' | sort
rm -rf $tmp
