export PYTHONUNBUFFERED=1
cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests/test_gpu_engine.py tests/test_gpu_model.py tests/test_gpu_ops.py tests/test_gpu_optim.py -q -m gpu > gpurun_out/t_rest.log 2>&1
grep -v "Exception ignored\|Traceback\|AttributeError\|^  File" gpurun_out/t_rest.log | tail -n 40
