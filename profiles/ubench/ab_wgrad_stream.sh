export PYTHONUNBUFFERED=1
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -u -m pytest -q -x --timeout 300 tests/test_gpu_ops.py -k "conv" 2>&1 | tail -2
GLF_WGRAD_STREAM=1 timeout -k 10 600 python -u -m pytest -q -x --timeout 300 tests/test_gpu_ops.py tests/test_gpu_model.py -k "conv or bottleneck or kinkfree" 2>&1 | tail -2
for a in 0 1 0 1; do echo "== GLF_WGRAD_STREAM=$a"; GLF_WGRAD_STREAM=$a timeout -k 10 300 python bench.py --steps 8 --warmup 3 --no-config3 --no-exact-f32 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c100-200; done
