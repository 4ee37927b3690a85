timeout -k 10 800 python -m pytest tests/test_gpu_ops.py tests/test_gpu_s16.py tests/test_gpu_model.py -x -q > gpurun_out/t13.log 2>&1; echo rc=$?; tail -3 gpurun_out/t13.log
b() { timeout -k 10 200 python bench.py --precision $1 --steps 6 --warmup 2 --no-exact-f32 --no-config3 --no-bf16 --no-cpu-baseline --no-other-mode --no-fusion-block 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$2', d['ms_per_step'])"; }
b bf16 "bf16"
b f16x3 "f16x3"
b bf16 "bf16"
b f16x3 "f16x3"
timeout -k 10 120 python profiles/ubench/s16_bn_probe.py 2>&1 | grep -v amdgpu.ids
