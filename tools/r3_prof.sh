#!/bin/bash
# kernel-trace stats of the default bench + plain bench lines (B = 8, B = 1)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r3prof; mkdir -p $O
if [ "${PYTEST:-0}" = "1" ]; then
  timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
fi
for b in 8 1; do
  timeout 300 python bench.py --batch $b --no-cpu-baseline --no-secondary > $O/b$b.json 2> $O/b$b.err
  python - <<PY
import json
try:
    d = json.load(open("$O/b$b.json")); r = d.get("roofline") or {}
    print("B=$b", round(d["value"], 1), "f/s", round(d["ms_per_step"], 4), "ms/step  icp us/launch", round(r.get("avg_launch_us", 0), 2),
          "sha", d["config"]["poses_sha"], "enq", round(d["config"]["host_enqueue_ms_per_step"], 3),
          "groups", {k: round(v, 4) for k, v in d["roofline_hbm"]["gpu_ms_per_step_by_group"].items()})
except Exception as e:
    print("B=$b FAILED", e); print(open("$O/b$b.err").read()[-1500:])
PY
done
B="timeout 170 python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline-pass --no-secondary"
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- $B > $O/trace.log 2>&1)
python - <<PY
import csv, glob
f = glob.glob("$O/trace/**/*kernel_stats.csv", recursive=True)
for row in list(csv.DictReader(open(f[0])))[:22]:
    print("%-60s %5s %9.1f us" % (row["Name"][:60], row["Calls"], float(row["AverageNs"]) / 1e3))
PY
