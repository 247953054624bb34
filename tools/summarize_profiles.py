"""Turns the outputs of tools/collect_profiles.sh into the committed summaries:
  profiles/<tag>_kernel_stats.csv     rocprofv3 --kernel-trace --stats table (verbatim)
  profiles/<tag>_pmc_hbm_traffic.txt  per kernel: calls, mean duration, HBM bytes per launch from the
                                      FETCH_SIZE / WRITE_SIZE passes, corrected as MI355X_MICROARCH.md says."""
import csv, collections, glob, os, shutil, sys
tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "gpurun_out")
def one(pattern):
    hits = glob.glob(os.path.join(out, pattern), recursive=True)
    assert hits, pattern
    return hits[0]
stats = one("%s_trace/**/*kernel_stats.csv" % tag)
shutil.copy(stats, os.path.join(root, "profiles", "%s_kernel_stats.csv" % tag))
dur = {}
for r in csv.DictReader(open(stats)):
    dur[r["Name"]] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3)
def pmc(kind):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(one("%s_pmc_%s/**/*counter_collection.csv" % (tag, kind)))):
        a = acc[r["Kernel_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
    return {k: v[0] / v[1] for k, v in acc.items()}
fetch, write = pmc("fetch"), pmc("write")
short = lambda n: n.split("(")[0][:62]
lines = ["# per kernel: calls and mean duration (rocprofv3 --kernel-trace --stats), HBM traffic per launch from separate",
         "# rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE passes of `python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline-pass --no-secondary` (B = 8 sequences on one GPU);",
         "# FETCH_SIZE (KB) doubled per MI355X_MICROARCH.md (gfx950 reports 1/2 of wide coalesced reads; narrow gathers are then",
         "# over-estimated), WRITE_SIZE (KB) as reported.  GB/s = (2*fetch + write) / mean duration.",
         "%-64s %6s %9s %12s %12s %9s" % ("kernel", "calls", "avg_us", "fetchKB(x2)", "writeKB", "GB/s")]
for name, (calls, us) in sorted(dur.items(), key=lambda kv: -kv[1][0] * kv[1][1]):
    f, w = 2.0 * fetch.get(name, 0.0), write.get(name, 0.0)
    lines.append("%-64s %6d %9.1f %12.1f %12.1f %9.1f" % (short(name), calls, us, f, w, (f + w) * 1024 / (us * 1e-6) / 1e9))
open(os.path.join(root, "profiles", "%s_pmc_hbm_traffic.txt" % tag), "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:14]))
