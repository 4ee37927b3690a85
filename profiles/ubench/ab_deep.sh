cd $GRAFT_REPO_ROOT
for d in 0 1 0 1; do echo "== DEEP=$d"; GLF_DEEP=$d python3 profiles/ubench/gemm_probe.py f16x3 8 2>&1 | grep "nt M"; done
GLF_DEEP=1 timeout 400 python -m pytest tests/test_gpu_ops.py -q -x -k "gemm or conv2d" 2>&1 | tail -3
