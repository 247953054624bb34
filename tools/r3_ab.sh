#!/bin/bash
# round 3, GPU call 1: tile engine parity + A/B against the row-unit engine + timelines
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3c2; mkdir -p $O
echo skip-pytest > $O/pytest.log
tail -25 $O/pytest.log
for eng in ${ENGINES:-tile}; do
  for b in 8 1; do
    GRADSLAM_HIP_ICP_ENGINE=$eng timeout 300 python bench.py --batch $b --no-cpu-baseline > $O/bench_${eng}_b$b.json 2> $O/bench_${eng}_b$b.err
    python - <<PY
import json
try:
    d = json.load(open("$O/bench_${eng}_b$b.json"))
    print("$eng B=$b", round(d["value"], 1), "f/s", round(d["ms_per_step"], 4), "ms/step  icp us/launch", round(d["roofline"]["avg_launch_us"], 2),
          "sha", d["config"]["poses_sha"], "ate_ref", d["config"]["ate_vs_reference_golden"], "enq", round(d["config"]["host_enqueue_ms_per_step"], 3),
          "groups", {k: round(v, 4) for k, v in d["roofline_hbm"]["gpu_ms_per_step_by_group"].items()})
except Exception as e:
    print("$eng B=$b FAILED", e)
PY
  done
done
GRADSLAM_HIP_BUILD_FLAGS=-DGS_ICP_TIMELINE python -m gradslam_amd.csrc.build > $O/build_tl.log 2>&1
for b in 8 1; do
  GRADSLAM_HIP_ICP_TIMELINE=$O/tl_b$b.txt timeout 200 python tools/icp_tile_timeline.py $b 2>&1 | tail -34
done
GRADSLAM_HIP_ICP_TIMELINE=$O/tl_c5.txt timeout 200 python tools/icp_tile_timeline.py 1 968 1296 2>&1 | tail -34
GRADSLAM_HIP_ICP_TIMELINE=$O/tl_b8_f12.txt timeout 200 python tools/icp_tile_timeline.py 8 480 640 12 2>&1 | tail -34
