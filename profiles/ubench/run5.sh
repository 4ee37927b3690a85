export PYTHONUNBUFFERED=1
cd $GRAFT_REPO_ROOT
for st in 1 0; do GLF_STREAMS=$st timeout -k 10 300 python bench.py --steps 8 --warmup 3 --no-config3 --no-exact-f32 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('streams=$st', d['ms_per_step'], 'host', d['host_enqueue_ms_per_step'])"; done
