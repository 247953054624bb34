"""Table of tools/calib/run.sh: per calibration launch the true bytes, FETCH_SIZE / WRITE_SIZE as rocprofv3 reports them (KB
in its unit) and the duration -> counter-per-true-byte factors and GB/s."""
import csv, glob, sys
O = sys.argv[1]
true = [l.split() for l in open(O + "/true_bytes.txt") if not l.startswith("#")]
def per_dispatch(d, col):
    f = glob.glob(O + "/" + d + "/**/*" + col + ".csv", recursive=True)[0]
    return list(csv.DictReader(open(f)))
tr = [r for r in per_dispatch("trace", "kernel_trace") if not r["Kernel_Name"].startswith("__amd")]
tr.sort(key=lambda r: int(r["Start_Timestamp"]))
def counters(d, name):
    rows = [r for r in per_dispatch(d, "counter_collection") if r["Counter_Name"] == name and not r["Kernel_Name"].startswith("__amd")]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    return [float(r["Counter_Value"]) for r in rows]
fe, wr = counters("fetch", "FETCH_SIZE"), counters("write", "WRITE_SIZE")
n = len(true)
reps = len(tr) // n
print("# rocprofv3 FETCH_SIZE / WRITE_SIZE (reported in KB) against the TRUE bytes of calibration kernels, MI355X; mean over %d repetitions (the first is cold)" % reps)
print("# %-14s %11s %10s %10s | %10s %8s | %10s %8s | %8s %9s" % ("kernel", "rows", "read MB", "write MB", "FETCH MB", "/ true", "WRITE MB", "/ true", "us", "true GB/s"))
for k in range(n):
    name, rows, rb, wb = true[k][0], int(true[k][1]), float(true[k][2]), float(true[k][3])
    idx = [r * n + k for r in range(1, reps)] or [k]
    f = sum(fe[i] for i in idx) / len(idx) * 1024.0
    w = sum(wr[i] for i in idx) / len(idx) * 1024.0
    us = sum((int(tr[i]["End_Timestamp"]) - int(tr[i]["Start_Timestamp"])) / 1e3 for i in idx) / len(idx)
    assert name in tr[idx[0]]["Kernel_Name"], (name, tr[idx[0]]["Kernel_Name"])
    print("  %-14s %11d %10.1f %10.1f | %10.1f %8s | %10.1f %8s | %8.1f %9.0f" % (
        name, rows, rb / 1e6, wb / 1e6, f / 1e6, ("%.3f" % (f / rb)) if rb else "-", w / 1e6, ("%.3f" % (w / wb)) if wb else "-",
        us, (rb + wb) / us / 1e3))
