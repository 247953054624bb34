"""Kernel time against the FRAME INDEX from a rocprofv3 --kernel-trace CSV of a bench.py run (VERDICT r04 #1a: the rate
halves over 200 steps -- which launches grow?).  A step is delimited by its gs_icp_finish_batch_kernel launch; per bin of
frames: mean duration of every kernel of the step, and of ICP half-iteration launch h (h = 2 x iteration + half) by
variant.

    python tools/frame_profile.py <kernel_trace.csv> [bin edges, default 5,25,55,65,95,105,195,205]
"""
import csv
import re
import sys
from collections import defaultdict

import numpy as np

path = sys.argv[1]
edges = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "5,25,55,65,95,105,195,205").split(",")]
bins = list(zip(edges[0::2], edges[1::2]))
rows = []
for r in csv.DictReader(open(path)):
    n = r["Kernel_Name"]
    if n.startswith("gs_") or "gs_icp" in n:
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n))
rows.sort()


def short(n):
    m = re.search(r"(gs_\w+)(<[^>]*>)?", n)
    return m.group(1) + (m.group(2) or "")


# steps: everything up to and including the merge_append launch that follows a finish launch
steps, cur, seen_finish = [], [], False
for s, e, n in rows:
    cur.append((s, e, short(n)))
    if "gs_icp_finish" in n:
        seen_finish = True
    if seen_finish and "gs_mu_merge_append" in n:
        steps.append(cur)
        cur, seen_finish = [], False
print("# %s: %d steps with a solve" % (path.split("/")[-1], len(steps)))
for lo, hi in bins:
    sel = steps[lo:hi]
    if not sel:
        continue
    print("\n## steps %d..%d (%d steps; index 0 = first step with an ICP solve, i.e. frame 1 of the run)" % (lo, min(hi, len(steps)) - 1, len(sel)))
    per_kernel = defaultdict(list)
    half = defaultdict(list)
    span, gaps = [], []
    for st in sel:
        span.append((st[-1][1] - st[0][0]) / 1e3)
        busy = sum(e - s for s, e, _ in st) / 1e3
        gaps.append(span[-1] - busy)
        acc = defaultdict(float)
        h = 0
        for s, e, n in st:
            acc[n] += (e - s) / 1e3
            if "gs_icp_half" in n:
                half[h].append(((e - s) / 1e3, n.split("<")[1].rstrip(">")))
                h += 1
        for k, v in acc.items():
            per_kernel[k].append(v)
    print("step span (first kernel start -> merge end) %.1f us, of which gaps between kernels %.1f us" % (np.mean(span), np.mean(gaps)))
    tot = {k: np.mean(v) * len(v) / len(sel) for k, v in per_kernel.items()}
    for k in sorted(tot, key=lambda k: -tot[k]):
        print("  %-52s %9.1f us per step" % (k, tot[k]))
    icp = sum(v for k, v in tot.items() if "gs_icp_half" in k)
    print("  ICP half-iteration launches together %.1f us; everything else %.1f us" % (icp, sum(tot.values()) - icp))
    print("  launch h: variant <FULL, G, FAR, LMODE>: mean / max us")
    for h in sorted(half):
        d = np.array([x[0] for x in half[h]])
        print("    %2d  %-22s %7.2f %7.2f" % (h, half[h][0][1], d.mean(), d.max()))
