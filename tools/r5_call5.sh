#!/bin/bash
# round 5, GPU call 5: where does a look-ahead launch with failing lists spend its time now?
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
for it in 5 13; do
  GRADSLAM_HIP_LIB=$GRAFT_REPO_ROOT/gradslam_amd/csrc/libgradslam_hip_tl.so GRADSLAM_HIP_ICP_TIMELINE_IT=$it GRADSLAM_HIP_ICP_TIMELINE=/tmp/tl_90.txt \
    timeout 300 python tools/r5_late_probe.py 96 tl 2>&1 | grep -v amdgpu.ids > $O/e5_tl_f96_it$it.txt
  cut -c1-260 $O/e5_tl_f96_it$it.txt
done
