#!/bin/bash
# round 5, GPU call 3: wide lists of hard queries -- the whole GPU suite, the long run, the frame-binned kernel trace, the
# late-frame timeline again, and one default bench line (new roofline accounting, live reference CPU baseline)
cd "$GRAFT_REPO_ROOT" || exit 1
ROOT=$GRAFT_REPO_ROOT
O=$ROOT/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests -x -q -m gpu > $O/e3_pytest.log 2>&1; tail -4 $O/e3_pytest.log
B="python $ROOT/bench.py --steps 205 --warmup 5 --no-cpu-baseline --no-roofline-pass --no-secondary"
sha() { python -c "import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], d['config']['poses_sha'], d['config']['map_surfels_end_rank0'][:2])" $1; }
timeout 400 $B > $O/e3_long.json 2> $O/e3_long.err; sha $O/e3_long.json
GRADSLAM_HIP_ICP_WIDE=0 timeout 400 $B > $O/e3_long_nowide.json 2> $O/e3_long_nowide.err; sha $O/e3_long_nowide.json
timeout 600 python $ROOT/bench.py --no-secondary > $O/e3_bench.json 2> $O/e3_bench.err; tail -c 3000 $O/e3_bench.json; tail -3 $O/e3_bench.err
cd /tmp && export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --output-format csv -d $O/e3_trace_b8 -o bench -- $B > $O/e3_trace_b8.log 2>&1
f=$(find $O/e3_trace_b8 -name '*kernel_trace.csv' | head -1)
python $ROOT/tools/frame_profile.py $f > $O/e3_frames_b8.txt 2>&1
grep -A3 "^## steps" $O/e3_frames_b8.txt | grep "step span"
rm -rf $O/e3_trace_b8
cd $ROOT
GRADSLAM_HIP_LIB=$ROOT/gradslam_amd/csrc/libgradslam_hip_tl.so GRADSLAM_HIP_ICP_TIMELINE_IT=19 GRADSLAM_HIP_ICP_TIMELINE=/tmp/tl_90.txt \
  timeout 300 python tools/r5_late_probe.py 90 tl 2>&1 | grep -v amdgpu.ids > $O/e3_tl_f90_it19.txt
cut -c1-250 $O/e3_tl_f90_it19.txt
