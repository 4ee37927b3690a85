export PYTHONUNBUFFERED=1
cd $GRAFT_REPO_ROOT
for a in 0 1 0 1; do echo "== GLF_GRAD_JOIN=$a"; GLF_GRAD_JOIN=$a timeout -k 10 300 python bench.py --steps 8 --warmup 3 --no-config3 --no-exact-f32 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c100-200; done
