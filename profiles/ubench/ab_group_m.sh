cd $GRAFT_REPO_ROOT
for g in 0 2 4 8 16; do echo "== GROUP_M=$g"; GLF_GROUP_M=$g python3 profiles/ubench/gemm_probe.py f16x3 8 2>&1 | grep "nt M"; done
GLF_GROUP_M=4 timeout 300 python -m pytest tests/test_gpu_ops.py -q -x -k "gemm or conv2d" 2>&1 | tail -3
