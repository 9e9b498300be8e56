#!/usr/bin/env python3
"""Per-kernel timeline of ONE replay of a captured step from a rocprofv3 --kernel-trace CSV:
   python tools/graph_timeline2.py <kernel_trace.csv> [marker-kernel-substring] [replay index from the end]
Splits the trace at every launch of the marker kernel (default: the first kernel of a step, k_pack_batched_amax), takes one
replay, and prints kernel, duration, gap to the previous kernel's end, plus the totals (sum of kernels, sum of gaps)."""
import csv, sys, collections

path = sys.argv[1]
marker = sys.argv[2] if len(sys.argv) > 2 else "k_pack_batched_amax"
back = int(sys.argv[3]) if len(sys.argv) > 3 else 2
rows = []
with open(path) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
starts = [i for i, r in enumerate(rows) if marker in r[2]]
if len(starts) < back + 1:
    print("marker", marker, "seen", len(starts), "times"); sys.exit(1)
lo, hi = starts[-back - 1], starts[-back]
seg = rows[lo:hi]
prev_end = seg[0][0]
tot_k = tot_g = 0
agg = collections.OrderedDict()
for s, e, n in seg:
    short = n.replace("void srk::", "").replace("srk::", "").split("(")[0][:70]
    gap = s - prev_end
    print("%8.2f us  gap %6.2f  %s" % ((e - s) / 1e3, gap / 1e3, short))
    tot_k += e - s
    tot_g += max(gap, 0)
    a = agg.setdefault(short, [0, 0.0])
    a[0] += 1; a[1] += (e - s) / 1e3
    prev_end = max(prev_end, e)
print("---- %d kernels, kernel time %.1f us, gaps %.1f us, span %.1f us" % (len(seg), tot_k / 1e3, tot_g / 1e3, (prev_end - seg[0][0]) / 1e3))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%4d x %8.2f us = %8.1f  %s" % (v[0], v[1] / v[0], v[1], k))
