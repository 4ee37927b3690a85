export PYTHONUNBUFFERED=1
cd $GRAFT_REPO_ROOT
export GLF_LIB_PATH=$GRAFT_REPO_ROOT/gl-fusion_amd/lib/libglfusion_stamps2.so
for cfg in "f16x3 ab" "f16x3 -"; do set -- $cfg; STAMPS_TN=1 STAMPS_MODE=2 timeout -k 10 120 python3 profiles/ubench/stamps.py $1 ${2#-} 2>&1 | grep -v amdgpu.ids | head -4; done
STAMPS_MODE=2 timeout -k 10 120 python3 profiles/ubench/stamps.py f16x3 ab 2>&1 | grep -v amdgpu.ids | head -3
