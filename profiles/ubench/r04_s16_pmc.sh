#!/bin/bash
# PMC counters of the two 16-bit contraction kernels on the M = 150 528 projection shapes (one counter group per pass)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for kind in nt tn; do
  for grp in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU"; do
    tag=$(echo $grp | cut -d' ' -f1)
    timeout -k 10 120 rocprofv3 --pmc $grp --kernel-trace -d $R/gpurun_out/pmc_s16_${kind}_$tag -o p -- python3 $R/profiles/ubench/s16_gemm_probe.py --one $kind > $R/gpurun_out/pmc_s16_${kind}_$tag.log 2>&1 || echo "pass failed: $kind $tag"
  done
done
python3 - <<'PY'
import glob, os, sqlite3, re
R = os.environ["GRAFT_REPO_ROOT"]
out = open(os.path.join(R, "gpurun_out", "s16_pmc_summary.txt"), "w")
for d in sorted(glob.glob(os.path.join(R, "gpurun_out", "pmc_s16_*"))):
    if not os.path.isdir(d): continue
    for db in glob.glob(os.path.join(d, "*.db")):
        con = sqlite3.connect(db); cur = con.cursor()
        tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
        view = "counters_collection" if "counters_collection" in tabs else None
        if not view:
            out.write(f"{d}: no counters view; tables {tabs[:12]}\n"); continue
        cols = [r[1] for r in cur.execute(f"pragma table_info({view})")]
        rows = list(cur.execute(f"select * from {view}"))
        agg = {}
        for r in rows:
            dct = dict(zip(cols, r))
            k = (re.sub(r"\(.*", "", str(dct.get("kernel_name", dct.get("name", "?"))))[:60], dct.get("counter_name"))
            a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += float(dct.get("value", 0) or 0)
        for (kn, cn), (n, v) in sorted(agg.items()):
            if "s16" in kn: out.write(f"{os.path.basename(d)} {kn} {cn} n={n} sum={v:.4g} per_launch={v / n:.4g}\n")
out.close()
PY
