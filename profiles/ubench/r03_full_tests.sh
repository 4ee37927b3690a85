export PYTHONUNBUFFERED=1
cd $GRAFT_REPO_ROOT
timeout -k 10 1150 python -m pytest tests -q -m gpu --durations=30 > gpurun_out/t_full.log 2>&1
grep -v "Exception ignored\|Traceback\|AttributeError\|^  File" gpurun_out/t_full.log | grep "AssertionError: (\|^FAILED\|passed\|failed\|^E  \|s call\|s setup" | tail -n 50
