#!/bin/bash
# round-3 evidence: rocprof kernel stats + HBM counters of the bench command, the default bench line, the c5 line
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-r03_b}
O=$GRAFT_REPO_ROOT/gpurun_out
bash tools/collect_profiles.sh $TAG > $O/${TAG}_collect.log 2>&1; tail -3 $O/${TAG}_collect.log
timeout 600 python bench.py > $O/${TAG}_bench_line.json 2> $O/${TAG}_bench_line.err; echo "bench rc=$?"
python - <<PY
import json
d = json.load(open("$O/${TAG}_bench_line.json"))
print({k: d[k] for k in ("metric", "value", "unit", "ms_per_step", "n_gpus", "steps")})
print("roofline", d["roofline"]); print("cpu", {k: d["cpu_baseline"][k] for k in ("value", "unit", "cores", "kind")})
s = d["secondary"]
for k, v in s.items():
    print(k, json.dumps(v)[:700])
print("cfg", {k: d["config"][k] for k in ("poses_sha", "host_enqueue_ms_per_step", "host_wait_ms_per_step", "ms_per_step_first_quartile", "ms_per_step_last_quartile", "ate_vs_reference_goldens_by_seed_rank0")})
PY
timeout 500 python bench.py --workload c5 --steps ${C5_STEPS:-500} --warmup 5 --no-cpu-baseline > $O/${TAG}_c5_bench_line.json 2> $O/${TAG}_c5_bench_line.err; echo "c5 rc=$?"
python - <<PY
import json
d = json.load(open("$O/${TAG}_c5_bench_line.json"))
print("c5", round(d["value"], 1), "f/s", round(d["ms_per_step"], 4), "ms/frame", d["config"]["poses_sha"], [round(s["ms_per_frame"], 2) for s in d["segments"]])
PY
