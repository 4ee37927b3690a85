#!/usr/bin/env python3
"""What the GPU is doing over one step, from a rocprofv3 --kernel-trace CSV: for the LAST `nsteps` steps of the trace
(delimited by counter_add_kernel launches, the first kernel of every step, eager or replayed) -- wall time per step, time with at least
one contraction kernel resident, time with only streaming kernels resident, idle time, and the mean number of concurrent
kernels.  Usage: timeline.py kernel_trace.csv [nsteps [index of the first step]]"""
import csv
import re
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
marks = [s for s, e, n in rows if "counter_add_kernel" in n]
if len(marks) < nsteps + 1:
    print("not enough step markers:", len(marks)); sys.exit(1)
first = int(sys.argv[3]) if len(sys.argv) > 3 else len(marks) - nsteps - 1       # index of the first step's marker (default: the last nsteps)
if first + nsteps >= len(marks):
    print("not enough step markers after", first, ":", len(marks)); sys.exit(1)
lo, hi = marks[first], marks[first + nsteps]
sel = [(s, e, n) for s, e, n in rows if s >= lo and s < hi]
is_gemm = lambda n: bool(re.search(r"gemm_rows|gemm_tn_|attn_|s16_rows_kernel|s16_tn_kernel", n))
ev = []
for s, e, n in sel:
    g = 1 if is_gemm(n) else 0
    ev.append((s, 1, g)); ev.append((e, -1, -g))
ev.sort()
# which kernels are resident while NO contraction is: sweep again with names
import collections
short = lambda n: re.sub(r"\(anonymous namespace\)::|void |<.*|\(.*", "", n)[:40]
ev2 = []
for s, e, n in sel:
    g = 1 if is_gemm(n) else 0
    ev2.append((s, 1, g, n)); ev2.append((e, -1, -g, n))
ev2.sort(key=lambda x: (x[0], x[1]))
t_prev, act, actg = lo, 0, 0
t_gemm = t_stream = t_idle = 0
area = 0
resident = collections.Counter()
exposed = collections.Counter()
for t, d, g, n in ev2:
    dt = t - t_prev
    if dt > 0:
        if actg > 0: t_gemm += dt
        elif act > 0:
            t_stream += dt
            k = sum(resident.values())
            for name, c in resident.items():
                if c > 0:
                    exposed[short(name)] += dt * c / k
        else: t_idle += dt
        area += dt * act
    act += d; actg += g; t_prev = t
    resident[n] += d
    if resident[n] <= 0:
        del resident[n]
wall = hi - lo
ksum = sum(e - s for s, e, n in sel)
gsum = sum(e - s for s, e, n in sel if is_gemm(n))
print(f"{nsteps} steps: wall {wall / nsteps / 1e6:.1f} ms/step; kernels {len(sel) / nsteps:.0f}/step; sum of kernel durations {ksum / nsteps / 1e6:.1f} ms/step "
      f"(contractions {gsum / nsteps / 1e6:.1f})")
print(f"  >=1 contraction resident: {t_gemm / nsteps / 1e6:.1f} ms/step; only streaming kernels resident: {t_stream / nsteps / 1e6:.1f}; "
      f"nothing resident: {t_idle / nsteps / 1e6:.1f}; mean concurrent kernels {area / wall:.2f}")
print("  streaming-only time by resident kernel (ms/step):")
for k, v in exposed.most_common(14):
    print(f"    {k:42s} {v / nsteps / 1e6:6.2f}")
