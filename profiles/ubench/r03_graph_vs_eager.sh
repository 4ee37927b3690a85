# same box: the graph-replayed step against the eagerly issued one, alternating, at the default --steps/--warmup (5/2) and at 20/5
export PYTHONUNBUFFERED=1
cd $GRAFT_REPO_ROOT
for sw in "5 2" "20 5" "5 2"; do
  set -- $sw
  for mode in "--graph" ""; do
    out=$(timeout -k 10 300 python bench.py --steps $1 --warmup $2 --no-exact-f32 --no-config3 --no-cpu-baseline --no-other-mode $mode 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['host_enqueue_ms_per_step'], d['host_enqueue_ms_first_step_idle_queue'])") || { echo "[$mode] FAILED"; exit 1; }
    echo "[steps $1 warmup $2 ${mode:-eager}] ms/step, host ms/step, host first step: $out"
  done
done
