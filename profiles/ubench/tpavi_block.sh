export PYTHONUNBUFFERED=1
R=$GRAFT_REPO_ROOT
cd $R
python3 profiles/ubench/tpavi_only.py f16x3 5 2>&1 | grep -v amdgpu.ids
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d /tmp/tp -o p --output-format csv -- python3 $R/profiles/ubench/tpavi_only.py f16x3 3 > /tmp/tp.log 2>&1 || tail -3 /tmp/tp.log
python3 - <<'PY'
import csv, glob, collections, re
f = glob.glob('/tmp/tp/**/*counter_collection.csv', recursive=True)[0]
busy = collections.defaultdict(float); act = collections.defaultdict(float)
for r in csv.DictReader(open(f)):
    k = re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", ""))
    (busy if r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES" else act)[k] += float(r["Counter_Value"])
tb, ta = sum(busy.values()), sum(act.values())
gemm = [k for k in act if k.startswith("gemm_")]
print("whole block MFMA busy fraction: %.4f" % (tb / (ta / 8 * 1024)))
print("contraction kernels only:       %.4f" % (sum(busy[k] for k in gemm) / (sum(act[k] for k in gemm) / 8 * 1024)))
for k in sorted(act, key=lambda k: -act[k])[:10]:
    print("  %-70s share of GPU-active %.3f  MFMA busy %.3f" % (k[:70], act[k] / ta, busy[k] / (act[k] / 8 * 1024)))
PY
