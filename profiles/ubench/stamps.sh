export PYTHONUNBUFFERED=1
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -u -m pytest -q -x --timeout 300 tests/test_gpu_ops.py -k "conv or gemm" 2>&1 | tail -2
export GLF_LIB_PATH=$GRAFT_REPO_ROOT/gl-fusion_amd/lib/libglfusion_stamps2.so
for cv in 256,256,1 512,512,2 2048,256,12; do echo "conv $cv"; STAMPS_CONV=$cv STAMPS_MODE=2 timeout -k 10 120 python3 profiles/ubench/stamps.py f16x3 ab 2>&1 | grep -v amdgpu.ids | head -2 | tail -1; done
