#!/usr/bin/env python3
"""Group a rocprofv3 kernel_stats.csv by kernel family: ms per step for `steps` profiled steps (warm-up + profiling + timed steps of
the command all count).  Usage: kstats_groups.py kernel_stats.csv steps"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
groups = [("NT contractions", r"gemm_rows|s16_rows_kernel|gemm_skinny"), ("TN contractions (wgrad)", r"gemm_tn|s16_tn_kernel"), ("tn_reduce", r"tn_reduce"),
          ("BN bwd reduce", r"colreduce_kernel<.*OpBnBwd|bnbwd_reduce"), ("BN bwd apply", r"bn_bwd_apply|bnbwd_apply"), ("BN apply", r"bn_apply_kernel|bn_apply_sums"),
          ("BN stats / finalize", r"colreduce_kernel<.*OpStats|bn_stats_finalize|sum_finalize|bnbwd_finalize"),
          ("other column reduces", r"colreduce"), ("add_n (fan-in)", r"add_n"), ("split_packed", r"split_packed"),
          ("weights refresh", r"weights_refresh"), ("bn_res_ln", r"bn_res_ln"), ("stem", r"stem"), ("maxpool", r"maxpool"),
          ("at::native", r"at::native|elementwise_kernel|vectorized"), ("amax", r"amax_kernel")]
tot = {g: [0.0, 0] for g, _ in groups}
rest, rest_n, allt = {}, 0, 0.0
for r in rows:
    name, t, n = r["Name"], float(r["TotalDurationNs"]) / 1e6, int(r["Calls"])
    if re.search(r"spin_kernel|probe_mfma", name):       # measurement aids of bench.py (queue preload, MFMA peak probe): not step work
        aids = aids + t if "aids" in dir() else t
        continue
    allt += t
    for g, pat in groups:
        if re.search(pat, name):
            tot[g][0] += t; tot[g][1] += n
            break
    else:
        k = re.sub(r"\(anonymous namespace\)::|void ", "", name)[:50]
        rest[k] = rest.get(k, 0.0) + t
print(f"total kernel time {allt / steps:.1f} ms/step over {steps:g} steps (bench.py's spin / probe kernels left out: {(aids if 'aids' in dir() else 0.0) / steps:.1f} ms/step)")
for g, _ in groups:
    if tot[g][1]:
        print(f"  {g:28s} {tot[g][0] / steps:8.2f} ms/step  {tot[g][1] / steps:8.1f} launches/step")
for k, t in sorted(rest.items(), key=lambda kv: -kv[1])[:12]:
    print(f"  {k:50s} {t / steps:8.2f} ms/step")
