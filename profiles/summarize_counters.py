#!/usr/bin/env python3
"""Fold a rocprofv3 --pmc counter_collection.csv into one row per kernel: launches and the sum of every counter.
Usage: summarize_counters.py <counter_collection.csv>"""
import collections
import csv
import re
import sys


def kname(s: str) -> str:
    s = s.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"\s*([A-Za-z_0-9:]+(?:<[^>]*>)?)", s)
    return m.group(1) if m else s[:60]


def main(path):
    d = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.defaultdict(set)
    names = []
    for r in csv.DictReader(open(path)):
        k = kname(r["Kernel_Name"])
        c = r["Counter_Name"]
        if c not in names:
            names.append(c)
        d[k][c] += float(r["Counter_Value"])
        n[k].add(r["Dispatch_Id"])
    print("kernel,launches," + ",".join(names))
    for k in sorted(d, key=lambda k: -d[k].get(names[0], 0.0)):
        print(f"{k},{len(n[k])}," + ",".join(f"{d[k].get(c, 0.0):.0f}" for c in names))


if __name__ == "__main__":
    main(sys.argv[1])
