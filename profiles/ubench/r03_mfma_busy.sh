# MFMA-busy of every contraction kernel of the C2 step (final round-3 code): rocprofv3 --pmc in its own pass, --kernel-trace only.
export PYTHONUNBUFFERED=1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
GLF_STREAMS=0 timeout -k 10 500 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d /tmp/mb -o p --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-config3 --no-cpu-baseline --no-exact-f32 --no-other-mode > /tmp/mb.log 2>&1 || { tail -3 /tmp/mb.log; exit 1; }
f=$(find /tmp/mb -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY' | tee $R/gpurun_out/mfma_busy_step.txt
import collections, csv, re, sys
agg = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"\s*([A-Za-z_0-9:]+(?:<[^>]*>)?)", k)
    k = m.group(1) if m else k[:60]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    disp[k].add(r["Dispatch_Id"])
print("# one-stream eager bench steps (GLF_STREAMS=0), all launches of the run; MFMA-busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)")
print(f"{'kernel':58s} {'launches':>8s} {'MFMA-busy':>10s} {'share of GUI-active cycles':>28s}")
tot_gui = sum(c.get("GRBM_GUI_ACTIVE", 0.0) for c in agg.values())
rows = []
for k, c in agg.items():
    busy, gui = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), c.get("GRBM_GUI_ACTIVE", 0.0)
    if gui <= 0: continue
    rows.append((gui, k, len(disp[k]), busy / (gui / 8.0 * 1024.0)))
for gui, k, n, frac in sorted(rows, reverse=True)[:24]:
    print(f"{k[:58]:58s} {n:8d} {frac:10.3f} {gui / tot_gui:28.3f}")
gb = sum(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) for k, c in agg.items() if "gemm_" in k)
gg = sum(c.get("GRBM_GUI_ACTIVE", 0.0) for k, c in agg.items() if "gemm_" in k)
print(f"all contraction kernels: MFMA-busy {gb / (gg / 8.0 * 1024.0):.3f} over {gg / tot_gui:.3f} of the GUI-active cycles; whole run: {sum(c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) for c in agg.values()) / (tot_gui / 8.0 * 1024.0):.3f}")
PY
