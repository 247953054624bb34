#!/bin/bash
# round 5, GPU call 14 (VERDICT r04 #2b: MEASURE the scalar stage in float32): an experimental build in which the 6x6 solve,
# both SE(3) exponentials and the gradLM update of the half-iteration prologue run in float32 (the float64 row sums stay)
# next to the product build -- launch durations and ATE against the reference goldens.  Not a product path.
cd "$GRAFT_REPO_ROOT" || exit 1
ROOT=$GRAFT_REPO_ROOT
O=$ROOT/gpurun_out; mkdir -p $O
show() { python -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d['config']
a=c['ate_vs_reference_goldens_by_seed_rank0']
print(sys.argv[1].split('/')[-1], 'value %.0f ms/step %.4f launch %.2f us sha %s' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], c['poses_sha']), 'ATE vs reference goldens max %.2e' % max(v['value_m'] for v in a.values()), 'ATE vs oracle', c.get('ate_vs_oracle_m'))
" $1; }
for rep in 1 2; do
  for v in main f32exp; do
    L=""; [ $v = f32exp ] && L=$ROOT/gradslam_amd/csrc/libgradslam_hip_f32exp.so
    GRADSLAM_HIP_LIB=$L timeout 400 python bench.py --no-secondary --no-cpu-baseline > $O/e14_b8_${v}_$rep.json 2> $O/e14_b8_${v}_$rep.err; show $O/e14_b8_${v}_$rep.json
  done
done
for v in main f32exp; do
  L=""; [ $v = f32exp ] && L=$ROOT/gradslam_amd/csrc/libgradslam_hip_f32exp.so
  GRADSLAM_HIP_LIB=$L timeout 400 python bench.py --batch 1 --no-secondary --no-cpu-baseline > $O/e14_b1_$v.json 2> $O/e14_b1_$v.err; show $O/e14_b1_$v.json
done
