export PYTHONUNBUFFERED=1
cd $GRAFT_REPO_ROOT
GLF_BENCH_DUMP=$GRAFT_REPO_ROOT/gpurun_out/per_shape timeout -k 10 400 python bench.py --steps 4 --warmup 2 --no-config3 --no-cpu-baseline --no-exact-f32 $1 > gpurun_out/bench_dump.json 2> gpurun_out/bench_dump.err || tail -5 gpurun_out/bench_dump.err
python - <<'PY'
import csv, json
d = json.load(open("gpurun_out/bench_dump.json"))
r = d["roofline"]
print("ms/step", d["ms_per_step"], "frac", r["frac"], "avg_launch_ms", r["avg_launch_ms"], "all frac", r["all_contractions"]["mfma_frac_of_peak"], "contr s/step", r["all_contractions"]["s_per_step"])
rows = list(csv.DictReader(open("gpurun_out/per_shape.f16x3.csv")))
tot = sum(float(x["ms_per_step"]) for x in rows)
print("sum of per-shape ms/step", round(tot, 1))
for x in rows[:45]:
    print(f"{x['kernel']:26s} M={x['M']:>7} N={x['N']:>5} K={x['K']:>6} taps={x['kept_taps']} b={x['batch']:>2} split={x['split']:>3} dil={x['dil']:>2} n={x['launches_per_step']:>3} ms={float(x['ms_per_step']):6.2f} avg={float(x['avg_ms']):.4f} execTF={float(x['executed_TF']):6.1f}")
PY
