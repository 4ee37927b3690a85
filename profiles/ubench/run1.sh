export PYTHONUNBUFFERED=1
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -u -m pytest -q -x --timeout 300 tests/test_gpu_ops.py tests/test_gpu_model.py -k "presplit or packed or tpavi or fusion or attention or softmax or kinkfree or bottleneck" > gpurun_out/t1.log 2>&1; tail -4 gpurun_out/t1.log
timeout -k 10 300 python bench.py --steps 8 --warmup 3 --no-config3 --no-exact-f32 --no-cpu-baseline 2>&1 | tail -1 | cut -c100-200
