#!/bin/bash
# round 5, GPU call 2: late-frame ICP launches by sequence (tools/r5_late_probe.py)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout 400 python tools/r5_late_probe.py 80 seeds > $O/e2_seeds.txt 2>&1; tail -9 $O/e2_seeds.txt
for fr in 62 90; do
  for it in 19 5; do
    GRADSLAM_HIP_LIB=$GRAFT_REPO_ROOT/gradslam_amd/csrc/libgradslam_hip_tl.so GRADSLAM_HIP_ICP_TIMELINE_IT=$it GRADSLAM_HIP_ICP_TIMELINE=/tmp/tl_$fr.txt \
      timeout 300 python tools/r5_late_probe.py $fr tl 2>&1 | grep -v amdgpu.ids > $O/e2_tl_f${fr}_it$it.txt
    cat $O/e2_tl_f${fr}_it$it.txt
  done
done
