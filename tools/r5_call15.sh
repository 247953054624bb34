#!/bin/bash
# round 5, GPU call 15: with wide lists in place, do the lists pay from iteration 0 on (the block-wide pass of a far point
# then runs twice per solve instead of four times)?  GRADSLAM_HIP_ICP_LISTS_FROM=0 against the default (1)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
show() { python -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d['config']
print(sys.argv[1].split('/')[-1], 'value %.0f ms/step %.4f sha %s' % (d['value'], d['ms_per_step'], c['poses_sha']))
" $1; }
B="python bench.py --no-cpu-baseline --no-roofline-pass --no-secondary"
for rep in 1 2; do
  for lf in 1 0; do
    GRADSLAM_HIP_ICP_LISTS_FROM=$lf timeout 400 $B --steps 205 --warmup 5 > $O/e15_long_lf${lf}_$rep.json 2> /dev/null; show $O/e15_long_lf${lf}_$rep.json
  done
done
for lf in 1 0; do
  GRADSLAM_HIP_ICP_LISTS_FROM=$lf timeout 400 $B > $O/e15_short_lf$lf.json 2> /dev/null; show $O/e15_short_lf$lf.json
  GRADSLAM_HIP_ICP_LISTS_FROM=$lf timeout 400 $B --batch 1 > $O/e15_b1_lf$lf.json 2> /dev/null; show $O/e15_b1_lf$lf.json
done
