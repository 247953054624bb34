#!/bin/bash
# round 3: the 1296x968 growing-map workload under both ICP engines (rows = default, tile = LDS slabs + lists)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3c5; mkdir -p $O
STEPS=${STEPS:-200}
for eng in rows tile; do
  GRADSLAM_HIP_ICP_ENGINE=$eng timeout 400 python bench.py --workload c5 --steps $STEPS --warmup 3 --no-cpu-baseline > $O/c5_$eng.json 2> $O/c5_$eng.err
  python - <<PY
import json
try:
    d = json.load(open("$O/c5_$eng.json"))
    c = d["config"]
    print("$eng", round(d["value"], 1), "f/s", round(d["ms_per_step"], 4), "ms/frame  icp us/launch", round(d["roofline"]["avg_launch_us"], 2),
          "sha", c["poses_sha"], "groups", {k: round(v, 4) for k, v in d["roofline_hbm"]["gpu_ms_per_step_by_group"].items()})
    print("   segments", [(s["map_bound_end"], round(s["ms_per_frame"], 3)) for s in (d.get("segments") or [])])
except Exception as e:
    print("$eng FAILED", e)
    import subprocess; print(open("$O/c5_$eng.err").read()[-1500:])
PY
done
