#!/bin/bash
# round 4, GPU call: full GPU suite, then A/B of the candidate lists of ordinary queries on the benchmark workload
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${R4_TAG:-r4a}; mkdir -p $O
if [ "${R4_TESTS:-1}" = "1" ]; then
  timeout 1500 python -m pytest tests -m gpu -x -q ${R4_PYTEST_ARGS:-} > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest.log
fi
line() {
  python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    s = d.get("secondary") or {}
    b1 = (s.get("b1_640x480") or {})
    print(sys.argv[2], round(d["value"], 1), "f/s", round(d["ms_per_step"], 4), "ms/step  icp us/launch", round(d["roofline"]["avg_launch_us"], 2),
          "sha", d["config"]["poses_sha"], "ate_ref", d["config"]["ate_vs_reference_golden"]["value_m"], "enq", round(d["config"]["host_enqueue_ms_per_step"], 3),
          "groups", {k: round(v, 4) for k, v in d["roofline_hbm"]["gpu_ms_per_step_by_group"].items()},
          "b1", round(b1.get("value", 0), 1), (b1.get("roofline") or {}).get("avg_launch_us"))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
for cfg in ${R4_CFGS:-"L1" "L0" "L1K"}; do
  case $cfg in
    L1) envs="GRADSLAM_HIP_ICP_LISTS=1";;
    L0) envs="GRADSLAM_HIP_ICP_LISTS=0";;
    L1K) envs="GRADSLAM_HIP_ICP_LISTS=1 HIP_FORCE_DEV_KERNARG=1";;
    L0K) envs="GRADSLAM_HIP_ICP_LISTS=0 HIP_FORCE_DEV_KERNARG=1";;
    F*) envs="GRADSLAM_HIP_ICP_LISTS=1 GRADSLAM_HIP_ICP_LISTS_FROM=${cfg#F}";;
    L1K0) envs="GRADSLAM_HIP_ICP_LISTS=1 HIP_FORCE_DEV_KERNARG=0";;
  esac
  for b in ${R4_B:-8 1}; do
    env $envs timeout 300 python bench.py --batch $b --no-cpu-baseline --no-secondary > $O/bench_${cfg}_b$b.json 2> $O/bench_${cfg}_b$b.err
    line $O/bench_${cfg}_b$b.json "$cfg B=$b"
  done
done
if [ "${R4_STATS:-1}" = "1" ]; then
  timeout 300 python tools/list_stats_probe.py 8 12 2>&1 | grep -v amdgpu.ids | tail -30
fi
