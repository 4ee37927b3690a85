export PYTHONUNBUFFERED=1
cd $GRAFT_REPO_ROOT
timeout -k 10 200 python profiles/ubench/trace_amax.py > gpurun_out/trace_amax.log 2>&1 || tail -5 gpurun_out/trace_amax.log
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/t_all.log 2>&1
tail -n 15 gpurun_out/t_all.log
