#!/bin/bash
# Development helper: VGPR / SGPR / scratch / LDS of the kernels in an object file or shared library (gfx950 code-object metadata).
#   tools/kernel_resources.sh build/hip/cg_k_big.o [name-filter]
set -e
f=$(realpath "$1"); pat=${2:-.}
t=$(mktemp -d); cd $t
cp "$f" in.bin
/opt/rocm/lib/llvm/bin/llvm-objdump --offloading in.bin > /dev/null
for co in *amdgcn*; do
  /opt/rocm/lib/llvm/bin/llvm-readelf --notes "$co" | awk -v pat="$pat" '
    /\.name:/ {name=$2} /\.vgpr_count:/ {v=$2} /\.sgpr_count:/ {s=$2} /\.private_segment_fixed_size:/ {p=$2} /\.agpr_count:/ {a=$2}
    /\.vgpr_spill_count:/ {sp=$2}
    /\.wavefront_size:/ { if (name ~ pat) printf "%-90s vgpr %3d agpr %3d sgpr %3d scratch %4d B spills %d\n", name, v, a, s, p, sp }'
done
rm -rf $t
