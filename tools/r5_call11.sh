#!/bin/bash
# round 5, GPU call 11: poses in an allocation of their own (no 69 MB block pinned per kept pose) -- whole GPU suite, smoke,
# the default bench line twice, the long run
cd "$GRAFT_REPO_ROOT" || exit 1
ROOT=$GRAFT_REPO_ROOT
O=$ROOT/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests -x -q -m gpu > $O/e11_pytest.log 2>&1; tail -3 $O/e11_pytest.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3
show() { python -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d['config']
print(sys.argv[1].split('/')[-1], 'value %.0f ms/step %.3f host_enqueue %.3f sha %s' % (d['value'], d['ms_per_step'], c['host_enqueue_ms_per_step'], c['poses_sha']))
s=d.get('secondary') or {}
for k in ('b1_640x480','stream_b8_640x480','steady_b8_200_steps','c5_1296x968'):
    if k in s: print('   ', k, {a:round(b,3) for a,b in s[k].items() if isinstance(b,float) and ('ms_per' in a or 'frames_per_s' in a or 'host' in a)})
cb=d.get('cpu_baseline')
if cb: print('    cpu_baseline', cb.get('kind'), cb.get('value'), cb.get('cores'), cb.get('total_s'))
" $1; }
t0=$(date +%s); timeout 900 python $ROOT/bench.py > $O/e11_bench_a.json 2> $O/e11_bench_a.err; echo "wall $(( $(date +%s) - t0 )) s"; show $O/e11_bench_a.json
timeout 400 python $ROOT/bench.py --no-cpu-baseline --no-secondary > $O/e11_bench_b.json 2> $O/e11_bench_b.err; show $O/e11_bench_b.json
timeout 400 python $ROOT/bench.py --steps 205 --warmup 5 --no-cpu-baseline --no-roofline-pass --no-secondary > $O/e11_long.json 2> $O/e11_long.err; show $O/e11_long.json
