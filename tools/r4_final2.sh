#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r4final2; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout 230 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - $O/bench.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
s = d.get("secondary") or {}
print(round(d["value"], 1), d["unit"], round(d["ms_per_step"], 4), "ms/step sha", d["config"]["poses_sha"], "icp us", round(d["roofline"]["avg_launch_us"], 2))
for k, v in s.items():
    print("  ", k, json.dumps(v)[:200])
PY
