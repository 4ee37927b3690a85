# round-3 working profile: the graph-replayed step, then rocprofv3 kernel stats of the eagerly issued step on ONE stream
export PYTHONUNBUFFERED=1
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 400 python bench.py --steps 10 --warmup 3 --no-config3 --no-exact-f32 --no-cpu-baseline --no-other-mode > gpurun_out/bench_main.json 2> gpurun_out/bench_main.err || { tail -5 gpurun_out/bench_main.err; exit 1; }
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_main.json"))
r = d["roofline"]
print("ms/step", d["ms_per_step"], "eager", d.get("eager_ms_per_step"), "host", d["host_enqueue_ms_per_step"], "host idle", d.get("host_enqueue_ms_first_step_idle_queue"),
      "frac", r["frac"], "all", r["all_contractions"]["mfma_frac_of_peak"], "contr s/step", r["all_contractions"]["s_per_step"], "mem", d["peak_mem_gb"])
PY
cd /tmp && export TMPDIR=/tmp
GLF_STREAMS=0 timeout -k 10 500 rocprofv3 --kernel-trace --stats -d /tmp/prof -o p --output-format csv -- python3 $R/bench.py --no-graph --steps 4 --warmup 2 --no-config3 --no-cpu-baseline --no-other-mode --no-exact-f32 > /tmp/prof.log 2>&1 || { tail -5 /tmp/prof.log; exit 1; }
f=$(find /tmp/prof -name "*kernel_stats.csv" | head -1)
cp $f $R/gpurun_out/one_stream_kernel_stats.csv
python3 $R/profiles/ubench/kstats_groups.py $R/gpurun_out/one_stream_kernel_stats.csv 7
