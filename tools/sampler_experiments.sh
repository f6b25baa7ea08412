#!/bin/bash
# Diagnostic (GPU box): timing experiments on k_mcmc (garbage numbers in the variants, same instruction mix elsewhere): how much
# could a cached sigmoid (exp_sigma: the Jacobian pass gets it for free), free LUs (exp_nolu, n <= 16 only) or free MFMA dense phases
# (exp_nodense) gain AT MOST?   Variants: tools/devbuild_variant.sh exp_sigma "-DCG_EXP_FREE_SIGMA" cg_k_sampler_a   etc.
for cfg in "13 8192 25" "29 2048 25" "57 512 49"; do
  set -- $cfg
  for v in "" exp_sigma exp_nolu exp_nodense; do
    lib=""; [ -n "$v" ] && lib="COULOMBGAS_HIP_LIB=coulombgas_amd/lib/diag/lib$v.so"
    env $lib python3 bench.py --n $1 --batch $2 --Emax $3 --no-cpu-baseline --no-energy-check --no-update-extras --steps 10 --warmup 3 | python3 -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('n=%-3s B=%-5s %-12s %7.3f M walker-steps/s, k_mcmc %7.3f ms, accept %.3f' % ('$1', '$2', '${v:-product}', j['value']/1e6, j['roofline']['kernel_avg_ms'], j['accept_rate']))"
  done
done
