#!/bin/bash
# round 5, GPU call 12 (closing): the per-GPU shard rates of N = 2 / 4 / 8 (4 / 2 / 1 sequences per GPU), the whole GPU suite once more
cd "$GRAFT_REPO_ROOT" || exit 1
ROOT=$GRAFT_REPO_ROOT
O=$ROOT/gpurun_out; mkdir -p $O
show() { python -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d['config']
print(sys.argv[1].split('/')[-1], 'value %.0f ms/step %.3f host_enqueue %.3f sha %s' % (d['value'], d['ms_per_step'], c['host_enqueue_ms_per_step'], c['poses_sha']))
" $1; }
for b in 4 2 1; do
  timeout 400 python $ROOT/bench.py --batch $b --no-cpu-baseline --no-roofline-pass --no-secondary > $O/e12_b$b.json 2> $O/e12_b$b.err; show $O/e12_b$b.json
done
timeout 900 python -m pytest tests -x -q -m gpu > $O/e12_pytest.log 2>&1; tail -3 $O/e12_pytest.log
