"""L2 hit rate and HBM fetch of the ICP half-iteration launches against the FRAME INDEX, from rocprofv3 --pmc passes
(the first halves of a long run slow down without failing lists; is it the per-sequence working
set outgrowing the 4 MB L2 of its XCD?).

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/l2 -o bench -- \
        python $ROOT/bench.py --batch 8 --steps 60 --warmup 5 --no-cpu-baseline --no-roofline-pass --no-secondary
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o bench -- (the same command)
    python tools/l2_probe.py $OUT/l2 $OUT/fetch > profiles/rNN_icp_l2_by_frame.txt

(counters in their own passes, no trace flags next to --pmc).  The dispatches of one frame are consecutive: 2 x numiters
half-iteration launches per solve, so dispatch k of the kernel belongs to frame k // 40 of the run.
"""
import csv
import glob
import sys
from collections import defaultdict


def load(d):
    rows = defaultdict(lambda: defaultdict(float))   # dispatch id -> counter -> value
    names = {}
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "gs_icp_half_batch_kernel" not in r["Kernel_Name"]:
                continue
            k = int(r["Dispatch_Id"])
            rows[k][r["Counter_Name"]] += float(r["Counter_Value"])
            names[k] = r["Kernel_Name"]
    order = sorted(rows)
    return [(names[k], rows[k]) for k in order]


def variant(name):   # <FULL, G, FAR, LMODE>
    a = name.split("<", 1)[1].split(">", 1)[0].replace(" ", "").split(",")
    return ("first half" if a[0] == "true" else "look-ahead") + (", lists" if a[3] == "2" else ", mode " + a[3])


per_solve = int(sys.argv[3]) if len(sys.argv) > 3 else 40
l2 = load(sys.argv[1])
fetch = load(sys.argv[2]) if len(sys.argv) > 2 else []
print("# frame (of the profiled run): per list-checking launch of that frame's solves, mean: L2 hit rate, misses, HBM fetch")
nfr = len(l2) // per_solve
for fr in range(nfr):
    for half in sorted(set(variant(n) for n, _ in l2[fr * per_solve:(fr + 1) * per_solve])):
        sel = [c for n, c in l2[fr * per_solve:(fr + 1) * per_solve] if variant(n) == half]
        if not sel:
            continue
        hit = sum(c["TCC_HIT_sum"] for c in sel) / len(sel)
        mis = sum(c["TCC_MISS_sum"] for c in sel) / len(sel)
        line = "frame %3d %-18s launches %2d  hit rate %.3f  misses %9.0f" % (fr, half, len(sel), hit / max(hit + mis, 1.0), mis)
        if fetch:
            fs = [c for n, c in fetch[fr * per_solve:(fr + 1) * per_solve] if variant(n) == half]
            if fs:   # FETCH_SIZE is in KB on this stack; x2 on MI355X per the guide
                line += "  fetch %.2f MB" % (2.0 * sum(c["FETCH_SIZE"] for c in fs) / len(fs) / 1024.0)
        print(line)
