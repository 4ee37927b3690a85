export PYTHONUNBUFFERED=1
cd $GRAFT_REPO_ROOT
timeout -k 10 1150 python -u -m pytest -x -q --timeout 400 tests -m gpu > gpurun_out/full_gpu.log 2>&1; tail -4 gpurun_out/full_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
