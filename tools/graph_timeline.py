#!/usr/bin/env python3
"""Per-kernel timeline of ONE replay of a captured step from a rocprofv3 kernel trace.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -o shard -- python tools/shard_step.py 16 20
    python tools/graph_timeline.py gpurun_out/tl [period_hint]

Takes the last complete repetition of the kernel-name sequence (the replays are identical) and prints start offset,
duration and the gap to the previous kernel's end, plus totals: kernel time, gaps, span."""
import csv, glob, os, sys

d = sys.argv[1]
f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = []
with open(f) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
names = [r[2] for r in rows]
# period = distance between the last two occurrences of the rarest kernel of the tail
tail = names[-4000:]
anchor = "k_adam" if any("k_adam" in n for n in tail) else ("k_sgd" if any("k_sgd" in n for n in tail) else tail[-1])
idx = [i for i, n in enumerate(names) if anchor in n]
per = idx[-1] - idx[-2]
rep = rows[idx[-2] + 1: idx[-1] + 1]
t0 = rep[0][0]
prev_end = t0
ksum = gsum = 0
short = lambda n: n.replace("void ", "").replace("srk::", "").split("(")[0][:60]
agg = {}
for s, e, n in rep:
    gap = s - prev_end
    print("%9.2f us  dur %8.2f  gap %6.2f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap / 1e3, short(n)))
    ksum += e - s
    gsum += max(gap, 0)
    a = agg.setdefault(short(n), [0, 0])
    a[0] += 1
    a[1] += e - s
    prev_end = max(prev_end, e)
print("--- %d kernels per replay; kernel time %.1f us, gaps %.1f us, span %.1f us" % (per, ksum / 1e3, gsum / 1e3, (prev_end - t0) / 1e3))
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%4d x %8.2f us = %8.1f us  %s" % (c, t / c / 1e3, t / 1e3, n))
