#!/bin/bash
# kernel-trace totals of the 1296x968 workload (first N frames), with and without the far-query candidate lists
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r3c5prof; mkdir -p $O
N=${N:-260}
for far in ${FARS:-1 0}; do
  B="timeout 300 python $GRAFT_REPO_ROOT/bench.py --workload c5 --steps $N --warmup 3 --no-cpu-baseline --no-roofline-pass"
  (cd /tmp && export TMPDIR=/tmp && GRADSLAM_HIP_ICP_FAR=$far rocprofv3 --kernel-trace --stats --output-format csv -d $O/far$far -o bench -- $B > $O/far$far.log 2>&1)
  echo "== far=$far"; grep -o '"value": [0-9.]*' $O/far$far.log | head -1
  python - <<PY
import csv, glob
f = glob.glob("$O/far$far/**/*kernel_stats.csv", recursive=True)
for row in list(csv.DictReader(open(f[0])))[:12]:
    print("%-64s %6s  avg %8.1f us  total %8.2f ms" % (row["Name"][:64], row["Calls"], float(row["AverageNs"]) / 1e3, float(row["TotalDurationNs"]) / 1e6))
PY
done
