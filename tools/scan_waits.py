#!/usr/bin/env python3
"""Scan compiled kernels (hipcc -S output) for s_waitcnt vmcnt(0) INSIDE loops that also issue vector stores or loads:
the signature of a wait the compiler could not count (a load first used inside the loop, memory operations under a
branch, a run-time switch around them) -- every iteration then waits for the write acknowledgements of its own stores.
   python tools/scan_waits.py /tmp/asm/*.s"""
import re, sys
for path in sys.argv[1:]:
    lines = open(path).read().split("\n")
    funcs, cur = [], None
    for i, ln in enumerate(lines):
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            cur = [m.group(1), i, None]; funcs.append(cur)
        if cur and "s_endpgm" in ln:
            cur[2] = i
    for name, a, b in funcs:
        if b is None: continue
        body = lines[a:b + 1]
        labels = {}
        for i, ln in enumerate(body):
            m = re.match(r"^(\.LBB\d+_\d+):", ln)
            if m: labels[m.group(1)] = i
        loops = []
        for i, ln in enumerate(body):
            m = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)", ln)
            if m:
                t = m.group(1) or m.group(2)
                if t in labels and labels[t] < i: loops.append((labels[t], i))
        out = []
        for lo, hi in loops:
            seg = body[lo:hi + 1]
            w0 = [lo + k for k, l in enumerate(seg) if re.search(r"s_waitcnt.*vmcnt\(0\)", l)]
            st = sum(1 for l in seg if re.search(r"\b(buffer|global|flat)_store", l))
            ld = sum(1 for l in seg if re.search(r"\b(buffer|global|flat)_load", l))
            mf = sum(1 for l in seg if "v_mfma" in l)
            if w0 and (st or ld) and hi - lo > 40:
                out.append("   loop @%d..%d (%d lines): vmcnt(0) x%d at %s, stores %d, loads %d, mfma %d" % (lo, hi, hi - lo, len(w0), w0[:6], st, ld, mf))
        if out:
            print(path.split("/")[-1], name)
            print("\n".join(out))
