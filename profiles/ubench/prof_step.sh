export PYTHONUNBUFFERED=1
R=$GRAFT_REPO_ROOT
cd $R
GLF_STREAMS=0 timeout -k 10 300 python bench.py --steps 8 --warmup 3 --no-config3 --no-exact-f32 --no-cpu-baseline 2>&1 | tail -1 | cut -c100-200
cd /tmp && export TMPDIR=/tmp
GLF_STREAMS=0 timeout -k 10 500 rocprofv3 --kernel-trace --stats -d /tmp/prof -o p --output-format csv -- python3 $R/bench.py --steps 4 --warmup 2 --no-config3 --no-cpu-baseline --no-exact-f32 > /tmp/prof.log 2>&1 || tail -5 /tmp/prof.log
f=$(find /tmp/prof -name "*kernel_stats.csv" | head -1)
cp $f $R/gpurun_out/step_kernel_stats.csv
tail -1 /tmp/prof.log | cut -c100-200
