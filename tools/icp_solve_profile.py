"""ICP time per frame from a rocprofv3 --kernel-trace CSV of a bench.py run, whatever the engine (launch per half-iteration or
the persistent per-XCD solve): first half-iteration launch of a solve -> end of its finish launch, and the duration of the
persistent launch when there is one.

    python tools/icp_solve_profile.py <kernel_trace.csv>
"""
import csv, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "gs_icp_" in n:
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n))
rows.sort()
first, persist, out = None, None, []
for s, e, n in rows:
    if first is None and "half_batch" in n:
        first = s
    if "persist" in n:
        persist = (e - s) / 1e3
    if "finish" in n and first is not None:
        out.append(((e - first) / 1e3, persist))
        first, persist = None, None
print("# frame: ICP us (first half-iteration launch -> end of the finish launch)  [persistent launch us]")
for i, (t, p) in enumerate(out):
    print("%3d %8.1f %s" % (i + 1, t, "" if p is None else "%8.1f" % p))
tt = [t for t, _ in out]
print("# mean %.1f us over %d solves; mean of the last 20: %.1f" % (sum(tt) / len(tt), len(tt), sum(tt[-20:]) / len(tt[-20:])))
