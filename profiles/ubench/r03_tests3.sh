export PYTHONUNBUFFERED=1
cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests/test_gpu_engine.py tests/test_gpu_model.py tests/test_gpu_ops.py -q -m gpu -k "graph or variants_train or temporal or retained" > gpurun_out/t_sub.log 2>&1
grep -v "Exception ignored\|Traceback\|AttributeError\|^  File" gpurun_out/t_sub.log | grep -n "AssertionError: (\|closest to its gate\|passed\|failed" | tail -n 30
