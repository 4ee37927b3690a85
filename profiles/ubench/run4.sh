export PYTHONUNBUFFERED=1
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; tail -1 gpurun_out/bench_default.json
