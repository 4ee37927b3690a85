# A/B of env switches on the graph-replayed C2 step.  Usage: r03_ab.sh "VAR=val ..." "VAR2=val ..." ...   (first: baseline "")
export PYTHONUNBUFFERED=1
cd $GRAFT_REPO_ROOT
for cfg in "$@"; do
  out=$(env $cfg timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-exact-f32 --no-config3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('eager_ms_per_step'), d['peak_mem_gb'])") || { echo "[$cfg] FAILED"; exit 1; }
  echo "[$cfg] ms/step eager mem: $out"
done
