# SQ counters of the isolated large GEMMs (NT shape 0, TN shape 6), both operands pre-split
export PYTHONUNBUFFERED=1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_LDS" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum"; do
  i=$((i+1))
  PROBE_NOCHECK=1 PROBE_SHAPES=0,6 PROBE_PACK=ab timeout -k 10 300 rocprofv3 --pmc $grp --kernel-trace -d /tmp/sq$i -o p --output-format csv -- python3 $R/profiles/ubench/gemm_probe.py f16x3 3 > /tmp/sq$i.log 2>&1 || tail -3 /tmp/sq$i.log
  f=$(find /tmp/sq$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 $R/profiles/summarize_counters.py $f | grep -E "^kernel|gemm_rows|gemm_tn" | tee -a $R/gpurun_out/pmc_sq_presplit.csv
done
