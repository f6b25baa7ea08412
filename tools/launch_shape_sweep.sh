#!/bin/bash
# Tuning run: launch shapes of the chunked derivative kernels (threads per workgroup, LDS budget, workgroups per CU) at n = 29 / 49 / 57
# through the CG_LAP_* / CG_VJP_* overrides of csrc/cg_k_derivs.inc.   bash tools/launch_shape_sweep.sh > gpurun_out/shape_sweep.txt
# (result of round 3: profiles/r03e_launch_shape_sweep.txt -- 256 threads x 2 workgroups per CU for k_grad_lap2 at N <= 64, else 512 x 1)
run() {  # n B  NT LDS_KB PER_CU
  echo "== n=$1 B=$2  NT=$3 LDS_KB=$4 PER_CU=$5"
  CG_LAP_NT=$3 CG_LAP_LDS_KB=$4 CG_LAP_PER_CU=$5 CG_VJP_NT=$3 CG_VJP_LDS_KB=$4 CG_VJP_PER_CU=$5 timeout -k 10 120 python tools/deriv_timing.py $1 $2 5 2>&1 | grep -v "^ *$"
}
for cfg in "512 156 1" "512 156 2" "512 78 2" "256 156 2" "256 78 2" "256 52 3" "256 39 4" "256 78 3"; do run 29 2048 $cfg; done
for cfg in "512 156 1" "512 78 2" "512 52 2" "256 78 2" "256 52 3"; do run 57 512 $cfg; done
for cfg in "512 156 1" "512 78 2" "256 78 2"; do run 49 512 $cfg; done
