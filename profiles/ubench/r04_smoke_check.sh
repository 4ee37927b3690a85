# smoke() under different chain lengths of the exact-precision fusion-block contractions, and the exact leg's step time
for kc in 256 512 1024 0; do
  echo "== GLF_EXACT_KCHUNK=$kc"
  GLF_EXACT_KCHUNK=$kc timeout -k 10 400 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -E "smoke OK \[f32\]|AssertionError" | cut -c1-700
  GLF_EXACT_KCHUNK=$kc timeout -k 10 300 python bench.py --precision f32 --steps 3 --warmup 1 --no-exact-f32 --no-config3 --no-bf16 --no-cpu-baseline --no-other-mode --no-fusion-block 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('exact f32 ms/step', d['ms_per_step'])"
done
