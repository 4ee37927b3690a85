"""MFMA-busy per kernel from a rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv pass:
python mfma_busy.py <counter_collection.csv> "<what was run>" [--json out.json]
MFMA-busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)  (MI355X_MICROARCH.md, MFMA utilisation)."""
import collections
import csv
import json
import re
import sys


def main():
    path, what = sys.argv[1], sys.argv[2]
    out_json = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        m = re.match(r"\s*([A-Za-z_0-9:]+(?:<[^>]*>)?)", k)
        k = m.group(1) if m else k[:60]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k].add(r["Dispatch_Id"])
    print(f"# {what}; MFMA-busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)")
    print(f"{'kernel':58s} {'launches':>8s} {'MFMA-busy':>10s} {'share of GUI-active cycles':>28s}")
    tot_gui = sum(c.get("GRBM_GUI_ACTIVE", 0.0) for c in agg.values())
    rows = []
    for k, c in agg.items():
        busy, gui = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), c.get("GRBM_GUI_ACTIVE", 0.0)
        if gui > 0:
            rows.append((gui, k, len(disp[k]), busy / (gui / 8.0 * 1024.0)))
    for gui, k, n, frac in sorted(rows, reverse=True)[:28]:
        print(f"{k[:58]:58s} {n:8d} {frac:10.3f} {gui / tot_gui:28.3f}")
    is_contr = lambda k: "gemm_" in k or "s16_rows" in k or "s16_tn_kernel" in k
    gb = sum(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) for k, c in agg.items() if is_contr(k))
    gg = sum(c.get("GRBM_GUI_ACTIVE", 0.0) for k, c in agg.items() if is_contr(k))
    # the model's own kernels: torch's fill / copy / random kernels of the set-up are not part of the block
    own = lambda k: not (k.startswith("at::") or k.startswith("__amd") or "spin_kernel" in k)
    ob = sum(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) for k, c in agg.items() if own(k))
    og = sum(c.get("GRBM_GUI_ACTIVE", 0.0) for k, c in agg.items() if own(k))
    contr = gb / (gg / 8.0 * 1024.0) if gg else 0.0
    allk = ob / (og / 8.0 * 1024.0) if og else 0.0
    print(f"all contraction kernels: MFMA-busy {contr:.3f} over {gg / tot_gui:.3f} of the GUI-active cycles; all library kernels: {allk:.3f}; "
          f"whole run: {sum(c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) for c in agg.values()) / (tot_gui / 8.0 * 1024.0):.3f}")
    if out_json:
        json.dump({"value": round(allk, 4), "contraction_kernels": round(contr, 4), "contraction_share_of_cycles": round(gg / og, 4) if og else None,
                   "counter": "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024)", "over": "every libglfusion kernel of the run", "what": what},
                  open(out_json, "w"), indent=1)


if __name__ == "__main__":
    main()
