export PYTHONUNBUFFERED=1
cd $GRAFT_REPO_ROOT
for mc in 0 512 1024 2048 4096; do echo "== GLF_PRESPLIT_MIN_COLS=$mc"; GLF_PRESPLIT_MIN_COLS=$mc timeout -k 10 300 python bench.py --steps 8 --warmup 3 --no-config3 2>&1 | tail -1 | cut -c100-200; done
