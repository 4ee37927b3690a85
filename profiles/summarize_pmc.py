#!/usr/bin/env python3
"""Fold rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE counter_collection.csv files into a per-kernel table.
Units: the counters are in KiB.  gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports
exactly half the bytes of a wide coalesced read => doubled here; WRITE_SIZE is exact for 16-B stores."""
import collections
import csv
import re
import sys


def kname(s: str) -> str:
    s = s.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"\s*([A-Za-z_0-9:]+(?:<[^>]*>)?)", s)
    return m.group(1) if m else s[:60]


def agg(path):
    d = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        k = kname(r["Kernel_Name"])
        d[k][0] += 1
        d[k][1] += float(r["Counter_Value"])
    return d


def main(fetch_csv, write_csv):
    f, w = agg(fetch_csv), agg(write_csv)
    print("kernel,launches,fetch_GB_corrected(x2),write_GB,total_GB,avg_MB_per_launch")
    tot = 0.0
    for k in sorted(f, key=lambda k: -(2 * f[k][1] + w.get(k, [0, 0])[1])):
        fe = 2 * f[k][1] * 1024 / 1e9
        wr = w.get(k, [0, 0.0])[1] * 1024 / 1e9
        tot += fe + wr
        print(f"{k},{f[k][0]},{fe:.3f},{wr:.3f},{fe + wr:.3f},{(fe + wr) * 1e3 / max(f[k][0], 1):.2f}")
    print(f"TOTAL,,,,{tot:.3f},")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
