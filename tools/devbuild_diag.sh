#!/bin/bash
# Development helper: recompile the named units (default cg_k_derivs_a) of the -DCG_STAMPS diagnostic library and relink
# coulombgas_amd/lib/diag/libcg_stamps.so from build/diag_cg_stamps/*.o (built once by `python -m coulombgas_amd.build --diag cg_stamps
# -DCG_STAMPS -DCG_ONLY_2_16_16`).
set -e
cd "$(dirname "$0")/.."
units=${@:-cg_k_derivs_a}
D=build/diag_cg_stamps
pids=()
for u in $units; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -DCG_STAMPS -DCG_ONLY_2_16_16 ${CG_EXTRA_FLAGS} -c -o $D/$u.o coulombgas_amd/csrc/$u.hip &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o coulombgas_amd/lib/diag/libcg_stamps.so $D/cg_k_sampler_a.o $D/cg_k_sampler_b.o $D/cg_k_derivs_a.o $D/cg_k_derivs_b.o $D/cg_hip.o $D/cg_k_generic.o -ldl
echo "relinked diag"
