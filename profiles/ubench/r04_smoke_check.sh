# smoke() twice, then the whole GPU suite, then the exact leg's step time
for k in 1 2; do timeout -k 10 400 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -E "smoke OK|AssertionError" | cut -c1-520; done
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/gputests.log 2>&1; echo rc=$?; tail -3 gpurun_out/gputests.log
timeout -k 10 300 python bench.py --precision f32 --steps 3 --warmup 1 --no-exact-f32 --no-config3 --no-bf16 --no-cpu-baseline --no-other-mode --no-fusion-block 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('exact f32 ms/step', d['ms_per_step'])"
