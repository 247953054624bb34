#!/bin/bash
# round 4, GPU call: the evidence behind DESIGN.md section 4 -- kernel stats + HBM traffic (tools/collect_profiles.sh),
# per-launch durations of the ICP chain, SQ counters of the half-iteration kernels, in-kernel timelines.
cd "$GRAFT_REPO_ROOT" || exit 1
ROOT=$GRAFT_REPO_ROOT
TAG=${R4_TAG:-r04_a}
O=$ROOT/gpurun_out
bash tools/collect_profiles.sh $TAG > $O/${TAG}_collect.log 2>&1; tail -3 $O/${TAG}_collect.log
B="timeout 170 python $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline-pass --no-secondary"
cd /tmp && export TMPDIR=/tmp
for b in 8 1; do
  f=$(find $O/${TAG}_trace -name '*kernel_trace.csv' | head -1)
  if [ "$b" = "1" ]; then
    rocprofv3 --kernel-trace --output-format csv -d $O/${TAG}_trace_b1 -o bench -- $B --batch 1 > $O/${TAG}_trace_b1.log 2>&1
    f=$(find $O/${TAG}_trace_b1 -name '*kernel_trace.csv' | head -1)
  fi
  python $ROOT/tools/icp_launch_profile.py $f 3 > $O/${TAG}_icp_launches_b$b.txt 2>&1; tail -3 $O/${TAG}_icp_launches_b$b.txt
done
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --output-format csv -d $O/${TAG}_pmc_sq1 -o bench -- $B > $O/${TAG}_pmc_sq1.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES --output-format csv -d $O/${TAG}_pmc_sq2 -o bench -- $B > $O/${TAG}_pmc_sq2.log 2>&1
python - <<PY
import csv, collections, glob, re
out = "$O"; tag = "$TAG"
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for d in ("sq1", "sq2"):
    for f in glob.glob("%s/%s_pmc_%s/**/*counter_collection.csv" % (out, tag, d), recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if "gs_icp_half_batch_kernel" not in n:
                continue
            a = acc[re.search(r"<[^>]*>", n).group(0)][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"]); a[1] += 1
names = ["SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_ANY", "SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT"]
lines = ["# SQ counters of gs_icp_half_batch_kernel<first half, lanes, far lists, list mode>, mean per dispatch (two rocprofv3 --pmc passes of",
         "# \`python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline-pass --no-secondary\`, 8 sequences of 640x480 per launch)",
         "%-26s %8s " % ("variant", "calls") + " ".join("%20s" % n for n in names)]
for k in sorted(acc):
    calls = max(v[1] for v in acc[k].values())
    lines.append("%-26s %8d " % (k, calls) + " ".join("%20.0f" % (acc[k][n][0] / acc[k][n][1]) if acc[k][n][1] else "%20s" % "-" for n in names))
open("%s/%s_icp_sq.txt" % (out, tag), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
cd $ROOT
GRADSLAM_HIP_BUILD_FLAGS=-DGS_ICP_TIMELINE python -m gradslam_amd.csrc.build > $O/${TAG}_build_tl.log 2>&1 || tail -20 $O/${TAG}_build_tl.log
: > $O/${TAG}_icp_timeline.txt
for b in 8 1; do
  for l in 1 0; do
    echo "######## B=$b candidate lists=$l (12 frames, last iteration of the last frame's solve)" >> $O/${TAG}_icp_timeline.txt
    GRADSLAM_HIP_ICP_LISTS=$l GRADSLAM_HIP_ICP_TIMELINE=$O/tl_tmp.txt timeout 200 python tools/icp_timeline.py $b 12 2>&1 | grep -v amdgpu.ids | tail -14 >> $O/${TAG}_icp_timeline.txt
  done
done
rm -f $O/tl_tmp.txt $O/tl_tmp.txt.next
rm -rf $O/${TAG}_trace_b1
tail -60 $O/${TAG}_icp_timeline.txt | head -30
