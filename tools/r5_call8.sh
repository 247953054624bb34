#!/bin/bash
# round 5, GPU call 8: whole GPU suite (deterministic backward, tie settling in the per-pixel pass), 1296x968 with far lists
# vs ordinary + wide lists, the default bench line with its secondary measurements
cd "$GRAFT_REPO_ROOT" || exit 1
ROOT=$GRAFT_REPO_ROOT
O=$ROOT/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests -x -q -m gpu > $O/e8_pytest.log 2>&1; tail -4 $O/e8_pytest.log
sha() { python -c "import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], d['config']['poses_sha'], d['config']['map_surfels_end_rank0'][:2])" $1; }
C5="python $ROOT/bench.py --workload c5 --steps 150 --warmup 5 --no-cpu-baseline --no-roofline-pass --no-secondary"
timeout 600 $C5 > $O/e8_c5_far.json 2> $O/e8_c5_far.err; sha $O/e8_c5_far.json
GRADSLAM_HIP_ICP_FAR=0 timeout 600 $C5 > $O/e8_c5_wide.json 2> $O/e8_c5_wide.err; sha $O/e8_c5_wide.json
/usr/bin/time -v timeout 900 python $ROOT/bench.py > $O/e8_bench_default.json 2> $O/e8_bench_default.err
grep -E "Elapsed|Maximum resident" $O/e8_bench_default.err; python - <<PY
import json
d=json.loads(open("$O/e8_bench_default.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["config"]["poses_sha"], d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], d["cpu_baseline"].get("total_s"))
s=d["secondary"]
for k,v in s.items():
    if isinstance(v, dict): print(k, {a:b for a,b in v.items() if a in ("frames_per_s","ms_per_step","frames_per_s_streamed","frames_per_s_resident","ratio","forward_ms","forward_taped_plus_backward_ms","driver_forward_ms","driver_forward_plus_backward_ms","ms_per_frame","error")})
print("seconds", s.get("seconds"))
PY
