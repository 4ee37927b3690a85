export PYTHONUNBUFFERED=1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace -d /tmp/tl -o p --output-format csv -- python3 $R/bench.py --steps 4 --warmup 2 --no-config3 --no-cpu-baseline --no-other-mode --no-exact-f32 > /tmp/tl.log 2>&1 || { tail -5 /tmp/tl.log; exit 1; }
f=$(find /tmp/tl -name "*kernel_trace.csv" | head -1)
head -2 $f | cut -c1-400
python3 $R/profiles/ubench/timeline.py $f 3
