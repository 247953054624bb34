#!/bin/bash
# closing evidence of a round (TAG=r06_a bash tools/evidence.sh on the GPU box): kernel stats + HBM traffic (tools/collect_profiles.sh), per-launch durations of the ICP
# chain at B = 8 and B = 1, SQ counters and L2 hit rates of the half-iteration kernels -- all of the CLOSING kernel mix
cd "$GRAFT_REPO_ROOT" || exit 1
ROOT=$GRAFT_REPO_ROOT
TAG=${TAG:-r06_a}
O=$ROOT/gpurun_out
bash tools/collect_profiles.sh $TAG > $O/${TAG}_collect.log 2>&1; tail -3 $O/${TAG}_collect.log
B="timeout 170 python $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline-pass --no-secondary"
cd /tmp && export TMPDIR=/tmp
f=$(find $O/${TAG}_trace -name '*kernel_trace.csv' | head -1)
python $ROOT/tools/icp_launch_profile.py $f 3 > $O/${TAG}_icp_launches_b8.txt 2>&1; tail -2 $O/${TAG}_icp_launches_b8.txt
rocprofv3 --kernel-trace --output-format csv -d $O/${TAG}_trace_b1 -o bench -- $B --batch 1 > $O/${TAG}_trace_b1.log 2>&1
f=$(find $O/${TAG}_trace_b1 -name '*kernel_trace.csv' | head -1)
python $ROOT/tools/icp_launch_profile.py $f 3 > $O/${TAG}_icp_launches_b1.txt 2>&1; tail -2 $O/${TAG}_icp_launches_b1.txt
rm -rf $O/${TAG}_trace_b1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --output-format csv -d $O/${TAG}_pmc_sq1 -o bench -- $B > $O/${TAG}_pmc_sq1.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES --output-format csv -d $O/${TAG}_pmc_sq2 -o bench -- $B > $O/${TAG}_pmc_sq2.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/${TAG}_pmc_l2 -o bench -- $B > $O/${TAG}_pmc_l2.log 2>&1
python - <<PY
import csv, collections, glob, re
out = "$O"; tag = "$TAG"
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for d in ("sq1", "sq2", "l2"):
    for f in glob.glob("%s/%s_pmc_%s/**/*counter_collection.csv" % (out, tag, d), recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if not (n.startswith("gs_") or "gs_icp" in n):
                continue
            k = re.sub(r"\(.*", "", n).replace("void ", "")
            a = acc[k][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"]); a[1] += 1
names = ["SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_ANY", "SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT", "TCC_HIT_sum", "TCC_MISS_sum"]
lines = ["# SQ and L2 counters of every kernel of a step, mean per dispatch (three rocprofv3 --pmc passes of",
         "# \`python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline-pass --no-secondary\`, 8 sequences of 640x480 per launch; closing state)",
         "%-52s %6s " % ("kernel", "calls") + " ".join("%20s" % n for n in names) + "   L2 hit rate"]
for k in sorted(acc, key=lambda k: -max(v[1] for v in acc[k].values())):
    calls = max(v[1] for v in acc[k].values())
    h, m = acc[k]["TCC_HIT_sum"], acc[k]["TCC_MISS_sum"]
    hr = (h[0] / max(h[0] + m[0], 1.0)) if h[1] else float("nan")
    lines.append("%-52s %6d " % (k[:52], calls) + " ".join("%20.0f" % (acc[k][n][0] / acc[k][n][1]) if acc[k][n][1] else "%20s" % "-" for n in names) + "   %.3f" % hr)
open("%s/%s_sq_l2_counters.txt" % (out, tag), "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:8]))
PY
rm -rf $O/${TAG}_pmc_sq1 $O/${TAG}_pmc_sq2 $O/${TAG}_pmc_l2
