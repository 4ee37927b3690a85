# A/B on the graph step AND on the one-stream contraction total (gap-free event timing)
export PYTHONUNBUFFERED=1
cd $GRAFT_REPO_ROOT
for cfg in "$@"; do
  env $cfg timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-exact-f32 --no-config3 --no-cpu-baseline 2>/dev/null > gpurun_out/ab_tmp.json || { echo "[$cfg] FAILED"; exit 1; }
  python -c "
import json
d = json.load(open('gpurun_out/ab_tmp.json')); r = d['roofline']
print('[$cfg] ms/step', d['ms_per_step'], 'contractions one-stream s/step', r['all_contractions']['s_per_step'], 'dominant frac', r['frac'], 'all frac', r['all_contractions']['mfma_frac_of_peak'])"
done
