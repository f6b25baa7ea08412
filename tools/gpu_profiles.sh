#!/bin/bash
# GPU box: everything profiles/ holds for one state of the tree (run at the end of a round):  tools/gpu_profiles.sh TAG -> gpurun_out/TAG/
#   bench line (plain and under rocprofv3), kernel stats of the bench run, FETCH / WRITE / SQ counter passes of k_mcmc and the traffic
#   file bench.py reads, the same for the derivative kernels at n = 13 / 29 / 57 (tools/gpu_baseline_derivs.sh), kernel stats of SR and
#   hybrid epochs, the damped solve at P = 5907 (kernel stats + a stretch of the two-stream timeline), bench lines + sampler stamps at
#   n = 29 / 57.
#   tools/gpu_profiles.sh TAG 1 : the first half (bench + its counters, derivative kernels);  TAG 2 : the rest;  no stage: everything
#   (one gpurun call is limited to 20 minutes)
set -e
TAG=${1:-prof}
STAGE=${2:-0}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
if [ "$STAGE" != "2" ]; then
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
echo "bench done: $(cut -c1-160 $OUT/bench.json)"
rocprofv3 --kernel-trace --stats -d $OUT/st_bench -- python3 bench.py --no-cpu-baseline --no-energy-check --no-update-extras > $OUT/bench_under_rocprof.json 2> $OUT/st_bench.log
python3 tools/profile_summary.py stats $OUT/st_bench > $OUT/kernel_stats_bench_n13_B8192.csv
for pmc in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_MFMA SQ_BUSY_CYCLES"; do
  nm=$(echo $pmc | cut -d' ' -f1 | tr 'A-Z' 'a-z')
  rocprofv3 --pmc $pmc -d $OUT/pmc_bench_$nm -- python3 bench.py --no-cpu-baseline --no-energy-check --no-update-extras --steps 3 --warmup 1 > $OUT/pmc_bench_$nm.log 2>&1
  python3 tools/profile_summary.py pmc $OUT/pmc_bench_$nm k_mcmc > $OUT/pmc_${nm}_k_mcmc_n13_B8192.txt
done
echo "bench counters done"
bash tools/gpu_baseline_derivs.sh $TAG/derivs > $OUT/derivs.log 2>&1
python3 tools/make_traffic_derivs.py $OUT/derivs $OUT/traffic_derivs.json "$TAG" > /dev/null
echo "derivative kernels done"
fi
if [ "$STAGE" = "1" ]; then ls $OUT; exit 0; fi
rocprofv3 --kernel-trace --stats -d $OUT/st_sr -- python3 tools/epoch_timing.py 13 8192 > $OUT/sr_epoch_rows.txt 2> $OUT/st_sr.log
python3 tools/profile_summary.py stats $OUT/st_sr > $OUT/kernel_stats_sr_epoch_n13_B8192.csv
rocprofv3 --kernel-trace --stats -d $OUT/st_hyb -- python3 tools/epoch_breakdown.py 13 8192 --van > $OUT/hybrid_epoch_breakdown.txt 2> $OUT/st_hyb.log
python3 tools/profile_summary.py stats $OUT/st_hyb > $OUT/kernel_stats_hybrid_epoch_n13_B8192.csv
for cfg in "13 8192" "29 2048" "49 512" "57 512"; do set -- $cfg; python3 tools/epoch_breakdown.py $1 $2 > $OUT/epoch_breakdown_n$1.txt 2>&1; done
( echo "finite-temperature epochs (both parameter sets trained) at the per-GPU shapes of BASELINE configs 4 / 5 (tools/epoch_breakdown.py N B --van)"
  for cfg in "29 2048" "57 512"; do set -- $cfg; echo; echo "== n=$1 B=$2"; python3 tools/epoch_breakdown.py $1 $2 --van 2>&1 | tail -n 8; done ) > $OUT/hybrid_epoch_breakdown_n29_n57.txt
for cfg in "13 8192" "29 2048" "57 512"; do set -- $cfg; python3 tools/van_timing.py $1 $2 >> $OUT/van_timing.txt 2>&1; done
python3 tools/fisher_timing.py > $OUT/fisher_timing.txt 2>&1
echo "epochs done"
python3 tools/solve_timing.py 333 1074 5907 > $OUT/solve_timing.txt 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/st_solve -- python3 tools/solve_timing.py 5907 > /dev/null 2> $OUT/st_solve.log
python3 tools/profile_summary.py stats $OUT/st_solve > $OUT/kernel_stats_spd_solve_P5907.csv
python3 tools/profile_summary.py timeline $OUT/st_solve chol 40 > $OUT/timeline_spd_solve_P5907.txt
echo "solve done"
python3 bench.py --n 29 --batch 2048 --Emax 25 --no-update-extras > $OUT/bench_n29.json 2> $OUT/bench_n29.err
python3 bench.py --n 57 --batch 512 --Emax 49 --no-update-extras > $OUT/bench_n57.json 2> $OUT/bench_n57.err
rocprofv3 --kernel-trace --stats -d $OUT/st_n29 -- python3 bench.py --n 29 --batch 2048 --Emax 25 --no-cpu-baseline --no-energy-check --no-update-extras > /dev/null 2> $OUT/st_n29.log
python3 tools/profile_summary.py stats $OUT/st_n29 > $OUT/kernel_stats_bench_n29_B2048.csv
rocprofv3 --kernel-trace --stats -d $OUT/st_n57 -- python3 bench.py --n 57 --batch 512 --Emax 49 --no-cpu-baseline --no-energy-check --no-update-extras > /dev/null 2> $OUT/st_n57.log
python3 tools/profile_summary.py stats $OUT/st_n57 > $OUT/kernel_stats_bench_n57_B512.csv
if [ -f coulombgas_amd/lib/diag/libcg_stamps.so ]; then
  for cfg in "13 8192" "29 2048" "57 512"; do set -- $cfg
    COULOMBGAS_HIP_LIB=coulombgas_amd/lib/diag/libcg_stamps.so python3 tools/stamps.py $1 $2 > $OUT/stamps_k_mcmc_n$1.txt 2>&1 || true
    [ "$1" = "13" ] && (COULOMBGAS_HIP_LIB=coulombgas_amd/lib/diag/libcg_stamps.so python3 tools/stamps_scores.py 13 8192 > $OUT/stamps_k_scores_n13.txt 2>&1 || true)
  done
fi
for d in $OUT/st_bench $OUT/st_solve $OUT/st_sr $OUT/st_hyb $OUT/st_n29 $OUT/st_n57 $OUT/pmc_bench_*; do [ -d "$d" ] && rm -rf "$d"; done
ls $OUT
