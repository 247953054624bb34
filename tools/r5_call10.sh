#!/bin/bash
# round 5, GPU call 10: is the slow default bench line (host enqueue 2.3 ms per step) reproducible, and what does it depend on?
cd "$GRAFT_REPO_ROOT" || exit 1
ROOT=$GRAFT_REPO_ROOT
O=$ROOT/gpurun_out; mkdir -p $O
show() { python -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d['config']
print(sys.argv[1].split('/')[-1], 'value %.0f ms/step %.3f host_enqueue %.3f first/last quartile %.3f %.3f' % (d['value'], d['ms_per_step'], c['host_enqueue_ms_per_step'], c['ms_per_step_first_quartile'], c['ms_per_step_last_quartile']))
s=d.get('secondary') or {}
for k in ('b1_640x480','stream_b8_640x480','steady_b8_200_steps'):
    if k in s: print('   ', k, {a:round(b,3) for a,b in s[k].items() if isinstance(b,float) and ('ms_per_step' in a or 'frames_per_s' in a or 'host' in a)})
" $1; }
uptime
timeout 900 python $ROOT/bench.py --no-cpu-baseline > $O/e10_a.json 2> $O/e10_a.err; show $O/e10_a.json
uptime
timeout 400 python $ROOT/bench.py --no-cpu-baseline --no-secondary > $O/e10_b.json 2> $O/e10_b.err; show $O/e10_b.json
timeout 900 python $ROOT/bench.py --no-cpu-baseline --no-roofline-pass > $O/e10_c.json 2> $O/e10_c.err; show $O/e10_c.json
uptime
