# same-box A/B of library builds on the C2 step: ab_libs_bench.sh <lib1> <lib2> ...  (paths relative to the repo root; each twice, interleaved)
export PYTHONUNBUFFERED=1
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for l in "$@"; do echo "== $l"; GLF_LIB_PATH=$GRAFT_REPO_ROOT/$l timeout -k 10 300 python bench.py --steps 8 --warmup 3 --no-config3 --no-exact-f32 --no-cpu-baseline 2>&1 | tail -1 | cut -c100-200; done; done
for l in "$@"; do echo "== $l"; for pk in "" b ab; do GLF_LIB_PATH=$GRAFT_REPO_ROOT/$l PROBE_SHAPES=0,2,4,5 PROBE_PACK=$pk timeout -k 10 200 python -u profiles/ubench/gemm_probe.py f16x3 5 2>&1 | grep " M="; done; done
