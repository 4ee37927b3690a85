# eager vs hipGraph replay of the 16-bit-storage step on one box (ms per step)
b() { timeout -k 10 300 python bench.py --precision bf16 --steps 6 --warmup 2 --no-exact-f32 --no-config3 --no-bf16 --no-cpu-baseline --no-other-mode --no-fusion-block $1 2> gpurun_out/graph_bf16_$2.err | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$2', d['ms_per_step'], d.get('host_enqueue_ms_first_step_idle_queue'), d.get('launch','')[:40])"; tail -2 gpurun_out/graph_bf16_$2.err; }
b "" eager
b "--graph" graph
b "" eager
b "--graph" graph
