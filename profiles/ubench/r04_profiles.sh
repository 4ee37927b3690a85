#!/bin/bash
# Round-4 judged artefacts, collected in one gpurun call and stamped (profiles/stamp.py: git hash + hash of the kernel sources; bench.py
# refuses a PMC-derived figure whose stamp does not match the sources it runs on).  Usage on the GPU box:
#   GLF_GIT_HASH=<hash> bash profiles/ubench/r04_profiles.sh [part ...]     parts: ks shapes pmc busy bench (default: all)
# Every rocprofv3 pass has the program itself after `--`; --pmc passes carry --kernel-trace only, one counter group per pass.
export PYTHONUNBUFFERED=1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04
mkdir -p $O
PARTS=${@:-ks shapes pmc busy bench}
COMMON="--no-config3 --no-cpu-baseline --no-exact-f32 --no-other-mode --no-bf16 --no-fusion-block"
cd /tmp && export TMPDIR=/tmp
has() { [[ " $PARTS " == *" $1 "* ]]; }

if has ks; then
  for prec in f16x3 bf16; do
    # (a) the bench's own launch mode (eager step, side streams on): the kernel durations the bench line's roofline is checked against
    timeout -k 10 400 rocprofv3 --kernel-trace --stats -d /tmp/ks1_$prec -o p --output-format csv -- python3 $R/bench.py --precision $prec --steps 5 --warmup 2 $COMMON > /tmp/ks1_$prec.log 2>&1 || { tail -3 /tmp/ks1_$prec.log; exit 1; }
    cp $(find /tmp/ks1_$prec -name "*kernel_stats.csv" | head -1) $O/r04_bench_c2_${prec}_kernel_stats.csv
    # where the GPU's time goes during a step of the timed mode: steps 2 .. 5 of the command's 2 warm-up + 5 timed steps
    python3 $R/profiles/ubench/timeline.py $(find /tmp/ks1_$prec -name "*kernel_trace.csv" | head -1) 4 2 > $O/r04_timeline_step_$prec.txt
    head -4 $O/r04_timeline_step_$prec.txt
    # (b) one stream: per-kernel time without overlap (the denominators of the per-kernel tables in DESIGN.md)
    GLF_STREAMS=0 timeout -k 10 400 rocprofv3 --kernel-trace --stats -d /tmp/ks0_$prec -o p --output-format csv -- python3 $R/bench.py --precision $prec --steps 5 --warmup 2 $COMMON > /tmp/ks0_$prec.log 2>&1 || { tail -3 /tmp/ks0_$prec.log; exit 1; }
    cp $(find /tmp/ks0_$prec -name "*kernel_stats.csv" | head -1) $O/r04_bench_c2_${prec}_one_stream_kernel_stats.csv
    python3 $R/profiles/ubench/kstats_groups.py $O/r04_bench_c2_${prec}_one_stream_kernel_stats.csv 11 > $O/r04_bench_c2_${prec}_one_stream_groups.txt
    tail -4 $O/r04_bench_c2_${prec}_one_stream_groups.txt
  done
fi

if has shapes; then
  # per-shape contraction tables (event pairs around every contraction launch of one-stream steps, outside the timed region)
  for prec in f16x3 bf16; do
    GLF_BENCH_DUMP=$O/r04_bench_c2_per_shape timeout -k 10 300 python3 $R/bench.py --precision $prec --steps 4 --warmup 2 $COMMON > /tmp/ps_$prec.log 2>&1 || { tail -3 /tmp/ps_$prec.log; exit 1; }
  done
  ls $O | grep per_shape
fi

if has pmc; then
  for prec in ${PMC_PRECS:-f16x3 bf16}; do
    for c in FETCH_SIZE WRITE_SIZE; do
      timeout -k 10 400 rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_${prec}_$c -o p --output-format csv -- python3 $R/bench.py --precision $prec --steps 2 --warmup 1 $COMMON > /tmp/pmc_${prec}_$c.log 2>&1 || { tail -3 /tmp/pmc_${prec}_$c.log; exit 1; }
    done
    python3 $R/profiles/summarize_pmc.py $(find /tmp/pmc_${prec}_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find /tmp/pmc_${prec}_WRITE_SIZE -name "*counter_collection.csv" | head -1) > $O/r04_bench_c2_${prec}_pmc_hbm_traffic.csv
    head -6 $O/r04_bench_c2_${prec}_pmc_hbm_traffic.csv; tail -2 $O/r04_bench_c2_${prec}_pmc_hbm_traffic.csv
    python3 $R/profiles/stamp.py $O/r04_bench_c2_${prec}_pmc_hbm_traffic.csv
  done
fi

if has busy; then
  # MFMA-busy: (a) every kernel of the one-stream step, (b) the fusion block alone (north_star's ">= 40 % on the fusion block")
  for prec in f16x3 bf16; do
    GLF_STREAMS=0 timeout -k 10 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d /tmp/mb_$prec -o p --output-format csv -- python3 $R/bench.py --precision $prec --steps 2 --warmup 1 $COMMON > /tmp/mb_$prec.log 2>&1 || { tail -3 /tmp/mb_$prec.log; exit 1; }
    python3 $R/profiles/ubench/mfma_busy.py $(find /tmp/mb_$prec -name "*counter_collection.csv" | head -1) "one-stream eager bench steps of --precision $prec (GLF_STREAMS=0), all launches of the run" > $O/r04_step_mfma_busy_$prec.txt
    tail -1 $O/r04_step_mfma_busy_$prec.txt
    timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d /tmp/fb_$prec -o p --output-format csv -- python3 $R/bench.py --precision $prec --fusion-block-only > /tmp/fb_$prec.log 2>&1 || { tail -3 /tmp/fb_$prec.log; exit 1; }
    python3 $R/profiles/ubench/mfma_busy.py $(find /tmp/fb_$prec -name "*counter_collection.csv" | head -1) "bench.py --precision $prec --fusion-block-only (7 forward + backward passes of one TPAVIModule at the C2 shape)" --json $O/r04_fusion_block_${prec}_mfma_busy.json > $O/r04_fusion_block_${prec}_mfma_busy.txt
    tail -1 $O/r04_fusion_block_${prec}_mfma_busy.txt
    python3 $R/profiles/stamp.py $O/r04_fusion_block_${prec}_mfma_busy.json
  done
fi

if has bench; then
  cd $R
  # the bench line quotes the stamped PMC figures: put this call's passes where bench.py looks for them (profiles/), as they will be committed
  cp $O/r04_bench_c2_*_pmc_hbm_traffic.* $O/r04_fusion_block_*_mfma_busy.* $R/profiles/ 2>/dev/null
  timeout -k 10 900 python bench.py > $O/r04_bench_default.json 2> $O/bench_default.err || { tail -5 $O/bench_default.err; exit 1; }
  tail -1 $O/r04_bench_default.json | cut -c1-400
fi
