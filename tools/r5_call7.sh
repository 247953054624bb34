#!/bin/bash
# round 5, GPU call 7: winner selection folded into the key pass and the merge (no pick pass over the map): the whole GPU suite (incl. the long-horizon goldens,
# recorded), long run + frame-binned trace, default-length run, B = 1 run
cd "$GRAFT_REPO_ROOT" || exit 1
ROOT=$GRAFT_REPO_ROOT
O=$ROOT/gpurun_out; mkdir -p $O
GRADSLAM_TEST_RECORD=$O timeout 900 python -m pytest tests -x -q -m gpu > $O/e7_pytest.log 2>&1; tail -4 $O/e7_pytest.log
B="python $ROOT/bench.py --no-cpu-baseline --no-roofline-pass --no-secondary"
sha() { python -c "import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], d['config']['poses_sha'], d['config']['map_surfels_end_rank0'][:2])" $1; }
timeout 400 $B --steps 205 --warmup 5 > $O/e7_long.json 2> $O/e7_long.err; sha $O/e7_long.json
timeout 400 $B > $O/e7_short.json 2> $O/e7_short.err; sha $O/e7_short.json
timeout 400 $B > $O/e7_short2.json 2> $O/e7_short2.err; sha $O/e7_short2.json
timeout 400 $B --batch 1 > $O/e7_b1.json 2> $O/e7_b1.err; sha $O/e7_b1.json
timeout 400 $B --batch 1 --steps 205 --warmup 5 > $O/e7_b1_long.json 2> $O/e7_b1_long.err; sha $O/e7_b1_long.json
cd /tmp && export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --output-format csv -d $O/e7_trace_b8 -o bench -- $B --steps 205 --warmup 5 > $O/e7_trace_b8.log 2>&1
f=$(find $O/e7_trace_b8 -name '*kernel_trace.csv' | head -1)
python $ROOT/tools/frame_profile.py $f > $O/e7_frames_b8.txt 2>&1
grep -A3 "^## steps" $O/e7_frames_b8.txt | grep "step span"
rm -rf $O/e7_trace_b8
