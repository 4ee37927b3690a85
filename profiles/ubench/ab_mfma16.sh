cd $GRAFT_REPO_ROOT
for cfg in "0 0" "1 0" "0 1" "1 1"; do set -- $cfg; echo "== MFMA16=$1 SETPRIO=$2"; GLF_MFMA16=$1 GLF_SETPRIO=$2 python3 profiles/ubench/gemm_probe.py f16x3 8 2>&1 | grep "nt M"; done
GLF_MFMA16=1 timeout 300 python -m pytest tests/test_gpu_ops.py -q -x -k "gemm or conv2d" 2>&1 | tail -3
