cd $GRAFT_REPO_ROOT
for m in 0 1 0 1; do echo "== MFMA16_PRESPLIT=$m"; PROBE_NOCHECK=1 GLF_MFMA16_PRESPLIT=$m PROBE_SHAPES=0,1,2,3,5 PROBE_PACK=ab python3 profiles/ubench/gemm_probe.py f16x3 8 2>&1 | grep "nt M"; done
