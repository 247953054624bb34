"""Per-launch anatomy of the ICP half-iteration chain from a rocprofv3 --kernel-trace CSV (the *_kernel_trace.csv of a
bench.py run): mean duration of launch h of a solve (h = 2 x iteration + half) and mean launch PERIOD (start of launch h
to start of launch h + 1: duration + kernel boundary), over all solves of the trace.

    python tools/icp_launch_profile.py <kernel_trace.csv> [warm-up solves to skip]
"""
import csv, re, sys
import numpy as np
path = sys.argv[1]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rows = []
for r in csv.DictReader(open(path)):
    n = r["Kernel_Name"]
    if "gs_icp_half_batch_kernel" in n or "gs_icp_finish_batch_kernel" in n or "gs_icp_far_build" in n:
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n))
rows.sort()
solves, cur = [], None
for s, e, n in rows:
    if "finish" in n:
        if cur is not None:
            cur.append((s, e, n)); solves.append(cur); cur = None
        continue
    if "far_build" in n:
        continue
    m = re.search(r"<(true|false), (\d), (true|false)(?:, (\d))?>", n)
    full, lm = m.group(1) == "true", int(m.group(4) or 0)
    if cur is None:
        cur = []
    cur.append((s, e, n))
solves = [s for s in solves if len(s) == len(solves[0])][skip:]
nh = len(solves[0]) - 1
dur = np.array([[(e - s) / 1e3 for s, e, _ in sv[:nh]] for sv in solves])
per = np.array([[(sv[h + 1][0] - sv[h][0]) / 1e3 for h in range(nh)] for sv in solves])
name = lambda n: re.search(r"<[^>]*>", n).group(0)
print("# %s: %d solves of %d half-iteration launches; us" % (path.split("/")[-1], len(solves), nh))
print("# launch  variant                      duration mean / max     period mean (start to next start)")
for h in range(nh):
    print("  %2d  %-28s %8.2f %8.2f %12.2f" % (h, name(solves[0][h][2]), dur[:, h].mean(), dur[:, h].max(), per[:, h].mean()))
print("# per solve: sum of durations %.1f us, first start to finish start %.1f us; mean launch period %.2f us" % (
    dur.sum(1).mean(), per.sum(1).mean(), per.mean()))
