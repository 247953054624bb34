#!/bin/bash
# round 5, GPU call 16: many failing lists re-searched by 8-lane groups with two gathers in flight (96 points per round)
# instead of 16-lane groups (48 per round): list tests, then A/B against the previous build
cd "$GRAFT_REPO_ROOT" || exit 1
ROOT=$GRAFT_REPO_ROOT
O=$ROOT/gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_batch.py -x -q -m gpu -k "candidate or wide or reproducible or engines" > $O/e16_pytest.log 2>&1; tail -2 $O/e16_pytest.log
show() { python -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d['config']
print(sys.argv[1].split('/')[-1], 'value %.0f ms/step %.4f sha %s' % (d['value'], d['ms_per_step'], c['poses_sha']))
" $1; }
B="python bench.py --no-cpu-baseline --no-roofline-pass --no-secondary"
for rep in 1 2; do
  for v in new prev; do
    L=""; [ $v = prev ] && L=$ROOT/gradslam_amd/csrc/libgradslam_hip_prev.so
    GRADSLAM_HIP_LIB=$L timeout 400 $B --steps 205 --warmup 5 > $O/e16_long_${v}_$rep.json 2> /dev/null; show $O/e16_long_${v}_$rep.json
  done
done
for v in new prev; do
  L=""; [ $v = prev ] && L=$ROOT/gradslam_amd/csrc/libgradslam_hip_prev.so
  GRADSLAM_HIP_LIB=$L timeout 400 $B > $O/e16_short_$v.json 2> /dev/null; show $O/e16_short_$v.json
done
