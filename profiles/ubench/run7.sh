export PYTHONUNBUFFERED=1
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python bench.py --steps 8 --warmup 3 --no-config3 --no-exact-f32 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], 'host', d['host_enqueue_ms_per_step'], d['roofline']['frac'], d['roofline']['all_contractions'])"
timeout -k 10 1100 python -u -m pytest -x -q --timeout 400 tests -m gpu > gpurun_out/full_gpu.log 2>&1; tail -5 gpurun_out/full_gpu.log
