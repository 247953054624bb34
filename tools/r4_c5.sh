#!/bin/bash
# round 4, GPU call: 1296x968 workload (BASELINE configs[4]) with far-candidate lists (default there) vs ordinary lists
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${R4_TAG:-r4c5}; mkdir -p $O
for cfg in ${R4_CFGS:-"far" "lists" "none"}; do
  case $cfg in
    far) envs="GRADSLAM_HIP_ICP_FAR=1";;
    lists) envs="GRADSLAM_HIP_ICP_FAR=0 GRADSLAM_HIP_ICP_LISTS=1";;
    none) envs="GRADSLAM_HIP_ICP_FAR=0 GRADSLAM_HIP_ICP_LISTS=0";;
  esac
  env $envs timeout 900 python bench.py --workload c5 --steps ${R4_STEPS:-200} --no-cpu-baseline --no-roofline-pass > $O/c5_$cfg.json 2> $O/c5_$cfg.err
  python - $O/c5_$cfg.json $cfg <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print(sys.argv[2], round(d["value"], 1), "f/s", round(d["ms_per_step"], 3), "ms/frame sha", d["config"]["poses_sha"],
          "segments", [round(s["ms_per_frame"], 2) for s in (d["config"].get("segments") or [])])
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done
