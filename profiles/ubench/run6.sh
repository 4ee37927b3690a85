export PYTHONUNBUFFERED=1
cd $GRAFT_REPO_ROOT
GLF_BENCH_CPROFILE=gpurun_out/host_profile timeout -k 10 300 python bench.py --steps 4 --warmup 2 --no-config3 --no-exact-f32 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-50
head -70 gpurun_out/host_profile.f16x3.txt
