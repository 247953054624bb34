#!/bin/bash
# round 5, GPU call 9: far lists opt-in -> tests that use them, the default bench line (timed), 1296x968 over 500 frames,
# closing evidence (tools/r5_final_evidence.sh)
cd "$GRAFT_REPO_ROOT" || exit 1
ROOT=$GRAFT_REPO_ROOT
O=$ROOT/gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_batch.py -x -q -m gpu -k "far or 1296 or wide or candidate" > $O/e9_pytest.log 2>&1; tail -3 $O/e9_pytest.log
t0=$(date +%s)
timeout 900 python $ROOT/bench.py > $O/e9_bench_default.json 2> $O/e9_bench_default.err
echo "default bench wall seconds: $(( $(date +%s) - t0 ))"
python - <<PY
import json
d=json.loads(open("$O/e9_bench_default.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["config"]["poses_sha"], d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], d["cpu_baseline"].get("total_s"))
print("roofline", d["roofline"]["frac"], d["roofline"]["avg_launch_us"], d["roofline_hbm"]["non_icp_gpu_ms_per_step"], d["roofline_hbm"]["hbm_frac_whole_step"])
s=d["secondary"]
for k,v in s.items():
    if isinstance(v, dict): print(k, {a:b for a,b in v.items() if a in ("frames_per_s","ms_per_step","frames_per_s_streamed","frames_per_s_resident","ratio","forward_ms","forward_taped_plus_backward_ms","driver_forward_ms","driver_forward_plus_backward_ms","ms_per_frame","error")})
print("secondary seconds", s.get("seconds"))
PY
sha() { python -c "import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], d['config']['poses_sha'], d['config']['map_surfels_end_rank0'][:2]); print(d.get('segments'))" $1; }
timeout 900 python $ROOT/bench.py --workload c5 --steps 500 --warmup 5 --no-cpu-baseline --no-roofline-pass --no-secondary > $O/e9_c5_500.json 2> $O/e9_c5_500.err; sha $O/e9_c5_500.json
bash tools/r5_final_evidence.sh
