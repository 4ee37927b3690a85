export PYTHONUNBUFFERED=1
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -u -m pytest -q -x --timeout 300 tests/test_gpu_ops.py -k "conv or gemm or presplit" 2>&1 | tail -2
PROBE_CONV=1 python3 profiles/ubench/gemm_probe.py f16x3 10 2>&1 | grep conv3x3
timeout -k 10 300 python bench.py --steps 8 --warmup 3 --no-config3 --no-exact-f32 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c100-200
