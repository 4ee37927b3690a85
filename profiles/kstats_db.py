"""Per-kernel totals from a rocprofv3 results .db (kernel trace): python profiles/kstats_db.py <results.db> [steps] -> CSV on stdout
(name, calls, total_ms, avg_us, pct), names shortened to the kernel's own name + leading template arguments."""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    name = re.sub(r"\(.*$", "", name)
    return name[:110]


def main():
    db = sqlite3.connect(sys.argv[1])
    steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    ni, si, ei = cols.index("name"), cols.index("start"), cols.index("end")
    agg = {}
    for r in cur.execute("select * from kernels"):
        a = agg.setdefault(short(r[ni]), [0, 0.0])
        a[0] += 1
        a[1] += (r[ei] - r[si]) / 1e6
    tot = sum(a[1] for a in agg.values())
    print("kernel,calls,total_ms,ms_per_step,avg_us,pct")
    for k, (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f'"{k}",{n},{ms:.3f},{ms / steps:.3f},{ms / n * 1e3:.2f},{100 * ms / tot:.2f}')
    print(f'"TOTAL",{sum(a[0] for a in agg.values())},{tot:.3f},{tot / steps:.3f},,100')


if __name__ == "__main__":
    main()
