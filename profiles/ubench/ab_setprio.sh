cd $GRAFT_REPO_ROOT
for m in 0 1 0 1; do echo "== SETPRIO=$m"; GLF_SETPRIO=$m PROBE_SHAPES=0,1,3,6 PROBE_PACK=ab python3 profiles/ubench/gemm_probe.py f16x3 8 2>&1 | grep " M="; done
