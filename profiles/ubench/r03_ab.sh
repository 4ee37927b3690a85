# A/B of env switches on the C2 step (eagerly issued by default; AB_ARGS="--graph" times the hipGraph replay).
# Usage: r03_ab.sh "VAR=val ..." "VAR2=val ..." ...   (first: baseline "")
export PYTHONUNBUFFERED=1
cd $GRAFT_REPO_ROOT
for cfg in "$@"; do
  out=$(env $cfg timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-exact-f32 --no-config3 --no-cpu-baseline --no-other-mode $AB_ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('graph_replay_ms_per_step') or d.get('eager_ms_per_step'), d['peak_mem_gb'])") || { echo "[$cfg] FAILED"; exit 1; }
  echo "[$cfg] ms/step other-mode mem: $out"
done
