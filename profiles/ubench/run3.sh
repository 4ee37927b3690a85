export PYTHONUNBUFFERED=1
cd $GRAFT_REPO_ROOT
GLF_LIB_PATH=$GRAFT_REPO_ROOT/gl-fusion_amd/lib/libglfusion_stamps2.so STAMPS_MODE=2 timeout -k 10 120 python3 profiles/ubench/stamps.py f16x3 ab 2>&1 | grep -v amdgpu.ids | head -4
for l in gl-fusion_amd/lib/libglfusion_prev.so gl-fusion_amd/lib/libglfusion_hip.so; do echo "== $l"; for pk in b ab; do GLF_LIB_PATH=$GRAFT_REPO_ROOT/$l PROBE_SHAPES=0,2,4,5 PROBE_PACK=$pk timeout -k 10 200 python -u profiles/ubench/gemm_probe.py f16x3 5 2>&1 | grep " M="; done; done
timeout -k 10 600 python -u -m pytest -q -x --timeout 300 tests/test_gpu_ops.py -k "presplit or packed or gemm or conv or column" > gpurun_out/t3.log 2>&1; tail -2 gpurun_out/t3.log
for rep in 1 2; do for l in gl-fusion_amd/lib/libglfusion_prev.so gl-fusion_amd/lib/libglfusion_hip.so; do echo "== $l"; GLF_LIB_PATH=$GRAFT_REPO_ROOT/$l timeout -k 10 300 python bench.py --steps 8 --warmup 3 --no-config3 --no-exact-f32 --no-cpu-baseline 2>&1 | tail -1 | cut -c100-200; done; done
