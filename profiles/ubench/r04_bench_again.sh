timeout -k 10 900 python bench.py > gpurun_out/bench_default2.json 2> gpurun_out/bench_default2.err; echo rc=$?
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_default2.json').read().strip().splitlines()[-1])
print("main", d["ms_per_step"], "exact", d["exact_f32"]["ms_per_step"], "f16", d["config3_f16"]["ms_per_step"], "bf16", d["config3_bf16"]["ms_per_step"])
PY
grep -v amdgpu gpurun_out/bench_default2.err | tail -5
