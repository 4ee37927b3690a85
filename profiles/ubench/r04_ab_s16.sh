# A/B of 16-bit-storage switches on one box: bench.py --precision bf16 (ms per step)
b() { timeout -k 10 200 python bench.py --precision $1 --steps 6 --warmup 2 --no-exact-f32 --no-config3 --no-bf16 --no-cpu-baseline --no-other-mode --no-fusion-block 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$2', d['ms_per_step'])"; }
b bf16 "bf16 default"
GLF_S16_WGRAD_STREAM=1 b bf16 "bf16 wgrad side stream"
b bf16 "bf16 default"
GLF_S16_WGRAD_STREAM=1 b bf16 "bf16 wgrad side stream"
