# A/B of alternative library builds on the GEMM probe: ab_lib.sh <lib1> <lib2> ...   (paths relative to the repo root)
cd $GRAFT_REPO_ROOT
for l in "$@"; do echo "== $l"; GLF_LIB_PATH=$GRAFT_REPO_ROOT/$l python3 profiles/ubench/gemm_probe.py f16x3 8 2>&1 | grep " M="; done
