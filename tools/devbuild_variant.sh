#!/bin/bash
# Development helper: an A/B variant of the product library -- the named units recompiled with extra flags, everything else from
# build/hip/*.o -- as coulombgas_amd/lib/diag/lib<NAME>.so (select it with COULOMBGAS_HIP_LIB).
#   tools/devbuild_variant.sh NAME "-DFLAG ..." unit [unit ...]
set -e
cd "$(dirname "$0")/.."
name=$1; flags=$2; shift 2
mkdir -p build/var_$name coulombgas_amd/lib/diag
objs=""
for u in cg_k_sampler_a cg_k_sampler_b cg_k_derivs_a cg_k_derivs_b cg_k_big cg_k_van cg_hip cg_k_generic; do
  if [[ " $* " == *" $u "* ]]; then
    /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC $flags -c -o build/var_$name/$u.o coulombgas_amd/csrc/$u.hip &
    objs="$objs build/var_$name/$u.o"
  else
    objs="$objs build/hip/$u.o"
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o coulombgas_amd/lib/diag/lib$name.so $objs -ldl
echo "built coulombgas_amd/lib/diag/lib$name.so"
