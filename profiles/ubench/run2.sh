export PYTHONUNBUFFERED=1
cd $GRAFT_REPO_ROOT
for pk in "" b ab; do PROBE_SHAPES=0,1,2,3,4,5 PROBE_PACK=$pk timeout -k 10 200 python -u profiles/ubench/gemm_probe.py f16x3 5 2>&1 | grep " M="; done
timeout -k 10 600 python -u -m pytest -q -x --timeout 300 tests/test_gpu_ops.py -k "presplit or packed or gemm or conv or column" > gpurun_out/t2.log 2>&1; tail -3 gpurun_out/t2.log
timeout -k 10 300 python bench.py --steps 8 --warmup 3 --no-config3 --no-exact-f32 --no-cpu-baseline 2>&1 | tail -1 | cut -c100-200
