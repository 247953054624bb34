#!/bin/bash
# round 3: fused launches of the map passes + candidate lists of far queries: parity suite, then A/B bench lines
# (expected fingerprints: B=8 8237d47a2b9fd695, B=1 bb535eb0c4a55979, c5 x 200 frames e7768e388490025c)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3fuse; mkdir -p $O
if [ "${PYTEST:-1}" = "1" ]; then
  timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest.log
fi
line() {  # file label
  python - <<PY
import json
try:
    d = json.load(open("$1"))
    r = d.get("roofline") or {}
    print("$2", round(d["value"], 1), "f/s", round(d["ms_per_step"], 4), "ms/step  icp us/launch", round(r.get("avg_launch_us", 0), 2),
          "sha", d["config"]["poses_sha"], "enq", round(d["config"]["host_enqueue_ms_per_step"], 3),
          "groups", {k: round(v, 4) for k, v in d["roofline_hbm"]["gpu_ms_per_step_by_group"].items()})
    if d.get("segments"):
        print("   segments", [round(s["ms_per_frame"], 3) for s in d["segments"]])
except Exception as e:
    print("$2 FAILED", e); print(open("$1".replace(".json", ".err")).read()[-1500:])
PY
}
for b in 8 1; do
  timeout 300 python bench.py --batch $b --no-cpu-baseline --no-secondary > $O/b$b.json 2> $O/b$b.err; line $O/b$b.json "default B=$b"
done
GRADSLAM_HIP_ICP_FAR=0 timeout 300 python bench.py --batch 8 --no-cpu-baseline --no-secondary > $O/b8_nofar.json 2> $O/b8_nofar.err; line $O/b8_nofar.json "no-far B=8"
GRADSLAM_HIP_FASTPATH=0 timeout 300 python bench.py --batch 8 --no-cpu-baseline --no-secondary > $O/b8_generic.json 2> $O/b8_generic.err; line $O/b8_generic.json "generic-path B=8"
for far in 1 0; do
  GRADSLAM_HIP_ICP_FAR=$far timeout 400 python bench.py --workload c5 --steps ${C5_STEPS:-200} --warmup 3 --no-cpu-baseline > $O/c5_far$far.json 2> $O/c5_far$far.err; line $O/c5_far$far.json "c5 far=$far"
done
