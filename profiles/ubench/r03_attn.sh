export PYTHONUNBUFFERED=1
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 300 python profiles/ubench/attn_softmax_only.py 5 2>&1 | tee gpurun_out/attn_softmax_timing.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/at -o p --output-format csv -- python3 $R/profiles/ubench/attn_softmax_only.py 3 > /tmp/at.log 2>&1 || tail -3 /tmp/at.log
cp $(find /tmp/at -name "*kernel_stats.csv" | head -1) $R/gpurun_out/attn_softmax_kernel_stats.csv
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d /tmp/atp -o p --output-format csv -- python3 $R/profiles/ubench/attn_softmax_only.py 2 > /tmp/atp.log 2>&1 || tail -3 /tmp/atp.log
f=$(find /tmp/atp -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY' | tee $R/gpurun_out/attn_softmax_pmc.txt
import collections, csv, sys
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:70]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    agg[k]["_n"] += 1.0 / 3
for k, c in agg.items():
    if "attn" not in k: continue
    busy, sq, gui = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0), c.get("SQ_BUSY_CYCLES", 0), c.get("GRBM_GUI_ACTIVE", 0)
    # SQ_VALU_MFMA_BUSY_CYCLES counts per SIMD (x4 per CU, 256 CUs); GRBM_GUI_ACTIVE per dispatch wall clock
    print(f"{k}: launches {c['_n']:.0f}  MFMA busy / (GUI active x 1024 SIMDs) = {busy / max(gui * 1024, 1):.3f}   (raw busy {busy:.3e}, gui {gui:.3e})")
PY
head -12 $R/gpurun_out/attn_softmax_kernel_stats.csv | cut -c1-160
