#!/bin/bash
# timelines of every launch of a solve (debugging build)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3tl; mkdir -p $O
for b in ${BENCH_B:-8 1}; do
  timeout 300 python bench.py --batch $b --no-cpu-baseline > $O/bench_b$b.json 2> $O/bench_b$b.err
  python - <<PY
import json
try:
    d = json.load(open("$O/bench_b$b.json"))
    print("B=$b", round(d["value"], 1), "f/s", round(d["ms_per_step"], 4), "ms/step  icp us/launch", round(d["roofline"]["avg_launch_us"], 2),
          "sha", d["config"]["poses_sha"], "ate_ref", d["config"]["ate_vs_reference_golden"]["value_m"], "enq", round(d["config"]["host_enqueue_ms_per_step"], 3),
          "groups", {k: round(v, 4) for k, v in d["roofline_hbm"]["gpu_ms_per_step_by_group"].items()})
except Exception as e:
    print("B=$b FAILED", e)
PY
done
GRADSLAM_HIP_BUILD_FLAGS=-DGS_ICP_TIMELINE python -m gradslam_amd.csrc.build > $O/build_tl.log 2>&1 || tail -20 $O/build_tl.log
GRADSLAM_HIP_ICP_TIMELINE=$O/tl_b8_f12.txt TL_LAUNCHES=${TL_LAUNCHES:-0,1,2,3,4,39} timeout 200 python tools/icp_tile_timeline.py 8 480 640 12 2>&1 | grep -v amdgpu.ids | tail -40
GRADSLAM_HIP_ICP_FORCE_SCAN=1 GRADSLAM_HIP_ICP_TIMELINE=$O/tl_b8_f12_force.txt TL_LAUNCHES=20,39 timeout 200 python tools/icp_tile_timeline.py 8 480 640 12 2>&1 | grep -v amdgpu.ids | tail -14
GRADSLAM_HIP_ICP_TIMELINE=$O/tl_b1_f12.txt TL_LAUNCHES=${TL_LAUNCHES1:-0,1,39} timeout 200 python tools/icp_tile_timeline.py 1 480 640 12 2>&1 | grep -v amdgpu.ids | tail -20
