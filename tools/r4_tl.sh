#!/bin/bash
# round 4, GPU call: in-kernel timelines of the last iteration's two launches, lists on / off
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${R4_TAG:-r4t}; mkdir -p $O
GRADSLAM_HIP_BUILD_FLAGS=-DGS_ICP_TIMELINE python -m gradslam_amd.csrc.build > $O/build_tl.log 2>&1 || tail -20 $O/build_tl.log
for b in ${R4_B:-8 1}; do
  for l in ${R4_L:-1 0}; do
    echo "######## B=$b lists=$l"
    GRADSLAM_HIP_ICP_LISTS=$l GRADSLAM_HIP_ICP_TIMELINE_IT=${R4_IT:-19} GRADSLAM_HIP_ICP_TIMELINE=$O/tl_b${b}_l$l.txt timeout 200 python tools/icp_timeline.py $b ${R4_FRAMES:-12} 2>&1 | grep -v amdgpu.ids | tail -${R4_TAIL:-14}
  done
done
