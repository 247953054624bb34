#!/bin/bash
# round 4, closing GPU call: GPU suite, smoke, the default bench line, kernel stats of the benchmark step
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/${R4_TAG:-r4final}; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu.ids | tail -2
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - $O/bench.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
s = d.get("secondary") or {}
print(round(d["value"], 1), d["unit"], round(d["ms_per_step"], 4), "ms/step sha", d["config"]["poses_sha"], "icp us", round(d["roofline"]["avg_launch_us"], 2),
      "cpu", d["cpu_baseline"]["value"], "groups", {k: round(v, 4) for k, v in d["roofline_hbm"]["gpu_ms_per_step_by_group"].items()})
for k, v in s.items():
    print("  ", k, json.dumps(v)[:260])
PY
ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- timeout 170 python $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline-pass --no-secondary > $O/trace.log 2>&1
cp $(find $O/trace -name '*kernel_stats.csv' | head -1) $O/kernel_stats.csv; rm -rf $O/trace
