#!/bin/bash
# Development helper: recompile only the named translation units (default: the (2,16,16) units) and relink
# libcoulombgas_hip.so from build/hip/*.o.  The full, dependency-checked build is `python -m coulombgas_amd.build`.
set -e
cd "$(dirname "$0")/.."
units=${@:-cg_k_sampler_a cg_k_derivs_a}
pids=()
for u in $units; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC ${CG_EXTRA_FLAGS} -c -o build/hip/$u.o coulombgas_amd/csrc/$u.hip &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o coulombgas_amd/lib/libcoulombgas_hip.so build/hip/cg_k_sampler_a.o build/hip/cg_k_sampler_b.o build/hip/cg_k_derivs_a.o build/hip/cg_k_derivs_b.o build/hip/cg_k_big.o build/hip/cg_k_van.o build/hip/cg_hip.o build/hip/cg_k_generic.o -ldl
echo "relinked"
