#!/bin/bash
# fast path (one foreign call per frame, graph replay) vs generic path, both ICP engines
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3host; mkdir -p $O
if [ "${PYTEST:-1}" = "1" ]; then
  timeout 900 python -m pytest tests -m gpu -q -x --deselect tests/test_hip_batch.py::test_two_ranks_sharing_one_gpu_rehearsal > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
fi
run() {  # name, env...
  name=$1; shift
  for b in ${BENCH_B:-8 1}; do
    env "$@" timeout 300 python bench.py --batch $b --no-cpu-baseline ${BENCH_ARGS:-} > $O/bench_${name}_b$b.json 2> $O/bench_${name}_b$b.err
    python - <<PY
import json
try:
    d = json.load(open("$O/bench_${name}_b$b.json"))
    r = d.get("roofline") or {}
    print("$name B=$b", round(d["value"], 1), "f/s", round(d["ms_per_step"], 4), "ms/step  icp us/launch", round(r.get("avg_launch_us", 0), 2),
          "sha", d["config"]["poses_sha"], "ate_ref", (d["config"]["ate_vs_reference_golden"] or {}).get("value_m"), "enq", round(d["config"]["host_enqueue_ms_per_step"], 3))
except Exception as e:
    print("$name B=$b FAILED", e); print(open("$O/bench_${name}_b$b.err").read()[-1500:])
PY
  done
}
for cfg in ${CONFIGS:-tile_fast rows_fast tile_generic rows_generic tile_fast_nograph}; do
  case $cfg in
    tile_fast) run $cfg GRADSLAM_HIP_ICP_ENGINE=tile ;;
    rows_fast) run $cfg GRADSLAM_HIP_ICP_ENGINE=rows ;;
    tile_generic) run $cfg GRADSLAM_HIP_ICP_ENGINE=tile GRADSLAM_HIP_FASTPATH=0 ;;
    rows_generic) run $cfg GRADSLAM_HIP_ICP_ENGINE=rows GRADSLAM_HIP_FASTPATH=0 ;;
    tile_fast_nograph) run $cfg GRADSLAM_HIP_ICP_ENGINE=tile GRADSLAM_HIP_GRAPH=0 ;;
  esac
done
