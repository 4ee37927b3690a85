export PYTHONUNBUFFERED=1
cd $GRAFT_REPO_ROOT
timeout -k 10 400 python3 profiles/ubench/config5.py 32 f16x3 2>&1 | grep -v "amdgpu.ids\|UserWarning\|Consider\|loss {" | tail -5
