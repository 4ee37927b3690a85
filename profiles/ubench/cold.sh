export PYTHONUNBUFFERED=1
cd $GRAFT_REPO_ROOT
for i in 1 2 3 4 5 6; do timeout -k 10 300 python bench.py --no-config3 --no-exact-f32 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('run $i', d['ms_per_step'], 'host', d['host_enqueue_ms_per_step'])"; done
