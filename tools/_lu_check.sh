set -e
timeout -k 10 200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -x -q -m gpu -k "n29 or n57 or golden or large or blocked or lu" 2>&1 | tail -3
export COULOMBGAS_HIP_LIB=coulombgas_amd/lib/diag/libcg_stamps.so
timeout -k 10 100 python tools/stamps.py 29 2048 | grep -E "kernel|real LU|slater matrix|complex LU|dual LU|total"
timeout -k 10 100 python tools/stamps.py 57 512 | grep -E "kernel|real LU|slater matrix|complex LU|dual LU|total"
