export PYTHONUNBUFFERED=1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
GLF_STREAMS=0 timeout -k 10 500 rocprofv3 --kernel-trace --stats -d /tmp/prof -o p --output-format csv -- python3 $R/bench.py --steps 4 --warmup 2 --no-config3 --no-cpu-baseline --no-exact-f32 > /tmp/prof.log 2>&1 || tail -5 /tmp/prof.log
f=$(find /tmp/prof -name "*kernel_stats.csv" | head -1)
cp $f $R/gpurun_out/step_kernel_stats.csv
head -40 $f | cut -c1-200
python3 - <<'PY'
import csv, os, re
rows = list(csv.DictReader(open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/step_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms (all launches):", tot / 1e6)
PY
