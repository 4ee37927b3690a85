# the round's judged artefacts: default bench line, kernel stats of the same command (eager step, streams on), one-stream
# kernel stats, per-shape contraction table, HBM traffic passes (FETCH_SIZE / WRITE_SIZE, separate), timeline of the step
export PYTHONUNBUFFERED=1
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err || { tail -5 gpurun_out/bench_default.err; exit 1; }
tail -1 gpurun_out/bench_default.json | cut -c1-300
GLF_BENCH_DUMP=$R/gpurun_out/per_shape timeout -k 10 400 python bench.py --steps 4 --warmup 2 --no-config3 --no-cpu-baseline --no-other-mode > gpurun_out/bench_dump.json 2>/dev/null || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d /tmp/ks1 -o p --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-config3 --no-cpu-baseline --no-exact-f32 --no-other-mode > /tmp/ks1.log 2>&1 || { tail -3 /tmp/ks1.log; exit 1; }
cp $(find /tmp/ks1 -name "*kernel_stats.csv" | head -1) $R/gpurun_out/ks_graph_streams.csv
# steps 2 .. 5 of the command's 2 warm-up + 5 timed steps: four steps inside the timed region (the trace goes on with bench.py's one-stream steps)
python3 $R/profiles/ubench/timeline.py $(find /tmp/ks1 -name "*kernel_trace.csv" | head -1) 4 2 | tee $R/gpurun_out/timeline_graph.txt
GLF_STREAMS=0 timeout -k 10 500 rocprofv3 --kernel-trace --stats -d /tmp/ks0 -o p --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-config3 --no-cpu-baseline --no-exact-f32 --no-other-mode > /tmp/ks0.log 2>&1 || { tail -3 /tmp/ks0.log; exit 1; }
cp $(find /tmp/ks0 -name "*kernel_stats.csv" | head -1) $R/gpurun_out/ks_one_stream.csv
python3 $R/profiles/ubench/kstats_groups.py $R/gpurun_out/ks_one_stream.csv 11 | tee $R/gpurun_out/ks_one_stream_groups.txt
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 500 rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_$c -o p --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-config3 --no-cpu-baseline --no-exact-f32 --no-other-mode > /tmp/pmc_$c.log 2>&1 || { tail -3 /tmp/pmc_$c.log; exit 1; }
done
python3 $R/profiles/summarize_pmc.py $(find /tmp/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find /tmp/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1) > $R/gpurun_out/pmc_hbm_traffic.csv
head -12 $R/gpurun_out/pmc_hbm_traffic.csv; tail -1 $R/gpurun_out/pmc_hbm_traffic.csv
