cd $GRAFT_REPO_ROOT
for g in 0 2 4 8 0 4; do echo "== GROUP_M=$g"; for pk in ab b; do GLF_GROUP_M=$g PROBE_SHAPES=0,1,2,3 PROBE_PACK=$pk python3 profiles/ubench/gemm_probe.py f16x3 8 2>&1 | grep "nt M"; done; done
