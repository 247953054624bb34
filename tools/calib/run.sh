#!/bin/bash
# On the GPU box: three rocprofv3 passes of the calibration binary (durations, FETCH_SIZE, WRITE_SIZE), then the table.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
ROOT=$PWD; O=$ROOT/gpurun_out/calib; mkdir -p $O
BIN=$ROOT/tools/calib/counter_calibration
cd /tmp && export TMPDIR=/tmp
$BIN 3 > $O/true_bytes.txt
rocprofv3 --kernel-trace --output-format csv -d $O/trace -o c -- $BIN 3 > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o c -- $BIN 3 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o c -- $BIN 3 > /dev/null 2>&1
python $ROOT/tools/calib/summarize.py $O > $ROOT/gpurun_out/counter_calibration.txt; cat $ROOT/gpurun_out/counter_calibration.txt
rm -rf $O/trace $O/fetch $O/write
