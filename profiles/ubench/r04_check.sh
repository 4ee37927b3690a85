# the whole GPU suite, then the step time of both storage modes (ms per step, two alternations)
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/gputests.log 2>&1; echo rc=$?; tail -3 gpurun_out/gputests.log
timeout -k 10 400 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo smoke rc=$?; tail -4 gpurun_out/smoke.log
b() { timeout -k 10 200 python bench.py --precision $1 --steps 6 --warmup 2 --no-exact-f32 --no-config3 --no-bf16 --no-cpu-baseline --no-other-mode --no-fusion-block 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$2', d['ms_per_step'])"; }
b bf16 "bf16"
b f16x3 "f16x3"
b bf16 "bf16"
b f16x3 "f16x3"
