#!/bin/bash
# round 4, GPU call: per-launch durations of the ICP chain (rocprofv3 kernel trace) + in-kernel phase timeline
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/${R4_TAG:-r4p}; mkdir -p $O
ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for b in ${R4_B:-8 1}; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_b$b -o bench -- timeout 170 python $ROOT/bench.py --batch $b --steps 10 --warmup 3 --no-cpu-baseline --no-roofline-pass --no-secondary > $O/trace_b$b.log 2>&1
  f=$(find $O/trace_b$b -name '*kernel_trace.csv' | head -1)
  python $ROOT/tools/icp_launch_profile.py $f 3 > $O/launches_b$b.txt 2>&1; cat $O/launches_b$b.txt
  cp $(find $O/trace_b$b -name '*kernel_stats.csv' | head -1) $O/kernel_stats_b$b.csv
  rm -rf $O/trace_b$b
done
cd $ROOT
if [ "${R4_TL:-1}" = "1" ]; then
  GRADSLAM_HIP_BUILD_FLAGS=-DGS_ICP_TIMELINE python -m gradslam_amd.csrc.build > $O/build_tl.log 2>&1 || tail -20 $O/build_tl.log
  for b in ${R4_B:-8 1}; do
    GRADSLAM_HIP_ICP_TIMELINE=$O/tl_b$b.txt timeout 200 python tools/icp_timeline.py $b 2>&1 | grep -v amdgpu.ids | head -12
  done
fi
