#!/bin/bash
# round 5, GPU call 13: the default bench line of the closing state (bench.py edited after call 11: two timed reference frames)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
t0=$(date +%s); timeout 900 python bench.py > $O/e13_bench.json 2> $O/e13_bench.err; echo "wall $(( $(date +%s) - t0 )) s"
python - <<PY
import json
d=json.loads(open("$O/e13_bench.json").read().strip().splitlines()[-1]); c=d["config"]
print("value %.0f ms/step %.3f host_enqueue %.3f sha %s" % (d["value"], d["ms_per_step"], c["host_enqueue_ms_per_step"], c["poses_sha"]))
cb=d["cpu_baseline"]; print(cb["kind"], cb["value"], cb["cores"], cb["seconds_per_frame"], cb["port"]["value"])
print(c["ate_vs_reference_live_m"], d["roofline"]["frac"], d["roofline"]["avg_launch_us"])
s=d["secondary"]
for k in ("b1_640x480","stream_b8_640x480","steady_b8_200_steps","c5_1296x968"):
    print("   ", k, {a:round(b,3) for a,b in s[k].items() if isinstance(b,float) and ("ms_per" in a or "frames_per_s" in a)})
PY
GRADSLAM_DIST_BACKEND=gloo GRADSLAM_BENCH_SHARE_GPU=1 timeout 300 python bench.py --gpus 2 --steps 3 --warmup 2 --no-secondary --no-cpu-baseline --no-roofline-pass 2> $O/e13_g2.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('2 ranks (shared GPU rehearsal):', d['value'], d['n_ranks_seen_by_backend'], [r['device_index'] for r in d['ranks']['per_rank']])"
