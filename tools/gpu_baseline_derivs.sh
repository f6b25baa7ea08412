#!/bin/bash
# Diagnostic (GPU box): kernel stats, counters and phase stamps of the derivative kernels at the three published sizes.
# usage: tools/gpu_baseline_derivs.sh TAG     -> gpurun_out/TAG/*
set -e
TAG=${1:-base}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for cfg in "13 8192" "29 2048" "49 512" "57 512"; do
  set -- $cfg
  python3 tools/deriv_timing.py $1 $2 5 >> $OUT/deriv_timing.txt 2>&1
done
cat $OUT/deriv_timing.txt
for cfg in "13 8192" "29 2048" "49 512" "57 512"; do
  set -- $cfg
  rocprofv3 --kernel-trace --stats -d $OUT/stats_n$1 -- python3 tools/deriv_timing.py $1 $2 3 > $OUT/stats_n$1.log 2>&1
  python3 tools/profile_summary.py stats $OUT/stats_n$1 > $OUT/kernel_stats_derivs_n$1_B$2.csv
  for pmc in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_MFMA SQ_BUSY_CYCLES"; do
    nm=$(echo $pmc | cut -d' ' -f1 | tr 'A-Z' 'a-z')
    rocprofv3 --pmc $pmc -d $OUT/pmc_${nm}_n$1 -- python3 tools/deriv_timing.py $1 $2 2 > $OUT/pmc_${nm}_n$1.log 2>&1
    python3 tools/profile_summary.py pmc $OUT/pmc_${nm}_n$1 k_grad_lap k_param_vjp k_scores k_gradlap_big k_gradlap_scores_big > $OUT/pmc_${nm}_derivs_n$1_B$2.txt
  done
  echo "done n=$1" 
done
if [ -f coulombgas_amd/lib/diag/libcg_stamps.so ]; then
  for cfg in "13 8192" "29 2048" "49 512" "57 512"; do
    set -- $cfg
    if [ "$1" = "13" ]; then COULOMBGAS_HIP_LIB=coulombgas_amd/lib/diag/libcg_stamps.so python3 tools/stamps_gradlap.py $1 $2 2 > $OUT/stamps_gradlap_n$1.txt 2>&1 || true
    else COULOMBGAS_HIP_LIB=coulombgas_amd/lib/diag/libcg_stamps.so python3 tools/stamps_big.py $1 $2 > $OUT/stamps_big_n$1.txt 2>&1 || true; fi
  done
fi
for d in $OUT/stats_n13 $OUT/stats_n29 $OUT/stats_n57 $OUT/pmc_*_n13 $OUT/pmc_*_n29 $OUT/pmc_*_n57; do [ -d "$d" ] && rm -rf "$d"; done; true
ls $OUT
