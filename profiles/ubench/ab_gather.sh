export PYTHONUNBUFFERED=1
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -u -m pytest -q -x --timeout 300 tests/test_gpu_ops.py -k "conv or gemm or presplit" 2>&1 | tail -2
for l in gl-fusion_amd/lib/libglfusion_prev.so gl-fusion_amd/lib/libglfusion_hip.so gl-fusion_amd/lib/libglfusion_prev.so gl-fusion_amd/lib/libglfusion_hip.so; do echo "== $l"; GLF_LIB_PATH=$GRAFT_REPO_ROOT/$l PROBE_CONV=1 python3 profiles/ubench/gemm_probe.py f16x3 10 2>&1 | grep conv3x3; done
