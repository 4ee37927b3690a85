# L2 / L1 behaviour of the dominant NT contraction: row-stride experiment + counter passes (one pass per counter group)
export PYTHONUNBUFFERED=1
R=$GRAFT_REPO_ROOT
cd $R
for pad in 0 32 64; do PROBE_SHAPES=0,1,3 PROBE_PACK=ab PROBE_LDPAD=$pad timeout -k 10 200 python -u profiles/ubench/gemm_probe.py f16x3 5 2>&1 | grep " M="; done
for pad in 0 32; do PROBE_SHAPES=0,1,3 PROBE_LDPAD=$pad timeout -k 10 200 python -u profiles/ubench/gemm_probe.py f16 5 2>&1 | grep " M="; done
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
           "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_TCR_TCP_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_BUSY_sum TCC_CYCLE_sum" \
           "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  PROBE_SHAPES=0 PROBE_PACK=ab timeout -k 10 300 rocprofv3 --pmc $grp --kernel-trace -d /tmp/pmc$i -o p --output-format csv -- python3 $R/profiles/ubench/gemm_probe.py f16x3 2 > /tmp/pmc$i.log 2>&1 || { tail -5 /tmp/pmc$i.log; }
  f=$(find /tmp/pmc$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 $R/profiles/summarize_counters.py $f | grep -E "^kernel|gemm_rows" | tee -a $R/gpurun_out/pmc_l2_summary.csv
done
