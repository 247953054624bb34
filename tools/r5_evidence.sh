#!/bin/bash
# round 5, GPU call 1 (VERDICT r04 #1a/#1b): what grows with the frame index?  Kernel traces of long runs binned by
# frame (tools/frame_profile.py), L2 hit rates (tools/l2_probe.py), and the never-run two-level-list draft as a second
# build of the library (GRADSLAM_HIP_LIB) next to main: parity first (poses_sha, list tests), then the same traces.
cd "$GRAFT_REPO_ROOT" || exit 1
ROOT=$GRAFT_REPO_ROOT
O=$ROOT/gpurun_out
mkdir -p $O
DRAFT=$ROOT/gradslam_amd/csrc/libgradslam_hip_draft2l.so
B="python $ROOT/bench.py --steps 205 --warmup 5 --no-cpu-baseline --no-roofline-pass --no-secondary"
sha() { python -c "import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], d['config']['poses_sha'], d['config']['map_surfels_end_rank0'][:2])" $1; }
timeout 400 $B > $O/e1_main_long.json 2> $O/e1_main_long.err; sha $O/e1_main_long.json
GRADSLAM_HIP_LIB=$DRAFT timeout 400 $B > $O/e1_draft_long.json 2> $O/e1_draft_long.err; sha $O/e1_draft_long.json
GRADSLAM_HIP_LIB=$DRAFT timeout 400 python -m pytest tests/test_hip_batch.py -x -q -m gpu -k "candidate_lists or reproducible or engines" > $O/e1_draft_pytest.log 2>&1; tail -3 $O/e1_draft_pytest.log
cd /tmp && export TMPDIR=/tmp
for v in main draft; do
  L=""; [ $v = draft ] && L=$DRAFT
  GRADSLAM_HIP_LIB=$L timeout 500 rocprofv3 --kernel-trace --output-format csv -d $O/e1_trace_b8_$v -o bench -- $B > $O/e1_trace_b8_$v.log 2>&1
  f=$(find $O/e1_trace_b8_$v -name '*kernel_trace.csv' | head -1)
  python $ROOT/tools/frame_profile.py $f > $O/e1_frames_b8_$v.txt 2>&1
  grep -A3 "^## steps" $O/e1_frames_b8_$v.txt | grep "step span"
  rm -rf $O/e1_trace_b8_$v
done
timeout 500 rocprofv3 --kernel-trace --output-format csv -d $O/e1_trace_b1 -o bench -- $B --batch 1 > $O/e1_trace_b1.log 2>&1
f=$(find $O/e1_trace_b1 -name '*kernel_trace.csv' | head -1)
python $ROOT/tools/frame_profile.py $f > $O/e1_frames_b1_main.txt 2>&1
rm -rf $O/e1_trace_b1
B2="python $ROOT/bench.py --steps 105 --warmup 5 --no-cpu-baseline --no-roofline-pass --no-secondary"
timeout 500 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/e1_l2 -o bench -- $B2 > $O/e1_l2.log 2>&1
python $ROOT/tools/l2_probe.py $O/e1_l2 > $O/e1_icp_l2_vs_frame.txt 2>&1
tail -4 $O/e1_icp_l2_vs_frame.txt
rm -rf $O/e1_l2
