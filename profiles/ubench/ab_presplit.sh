set -e
export PYTHONUNBUFFERED=1
cd $GRAFT_REPO_ROOT
for pk in "" ab; do PROBE_PACK=$pk timeout -k 10 200 python -u profiles/ubench/gemm_probe.py f16x3 5; done > gpurun_out/presplit_probe.log 2>&1
timeout -k 10 700 python -u -m pytest -q -x --timeout 300 tests/test_gpu_ops.py -k "presplit or packed or gemm or conv or column" > gpurun_out/presplit_tests.log 2>&1 || true
tail -5 gpurun_out/presplit_tests.log
GLF_PRESPLIT=0 timeout -k 10 300 python bench.py --steps 8 --warmup 3 --no-config3 > gpurun_out/presplit_bench0.log 2>&1
GLF_PRESPLIT=1 timeout -k 10 300 python bench.py --steps 8 --warmup 3 --no-config3 > gpurun_out/presplit_bench1.log 2>&1
grep -v amdgpu.ids gpurun_out/presplit_probe.log
tail -1 gpurun_out/presplit_bench0.log | cut -c1-300
tail -1 gpurun_out/presplit_bench1.log | cut -c1-300
