"""Per-kernel PMC sums from rocprofv3 --pmc results .db files: python profiles/pmc_db.py <db> [...]  (pmc_events view: one row per
dispatch, counter and hardware instance; values are summed over instances and averaged over the launches of a kernel)."""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return re.sub(r"\(.*$", "", name)[:80]


def main():
    for path in sys.argv[1:]:
        con = sqlite3.connect(path)
        cur = con.cursor()
        cols = [r[1] for r in cur.execute("pragma table_info(pmc_events)")]
        ni, di, ci, vi, du = cols.index("name"), cols.index("dispatch_id"), cols.index("counter_name"), cols.index("counter_value"), cols.index("duration")
        agg, launches, dur = {}, {}, {}
        for r in cur.execute("select * from pmc_events"):
            k = short(r[ni])
            agg[(k, r[ci])] = agg.get((k, r[ci]), 0.0) + float(r[vi])
            launches.setdefault(k, set()).add(r[di])
            dur[(k, r[di])] = r[du]
        for (k, c), v in sorted(agg.items()):
            n = len(launches[k])
            avg_us = sum(d for (kk, _), d in dur.items() if kk == k) / n / 1e3
            print(f"{path.split('/')[-2]}: {k} launches {n} avg_us {avg_us:.1f} {c} per_launch {v / n:.6g}")


if __name__ == "__main__":
    main()
