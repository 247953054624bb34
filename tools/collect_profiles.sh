#!/bin/bash
# Run on the GPU box (through gpurun): kernel-trace stats and the two HBM-traffic PMC passes of the bench
# command, each in its own rocprofv3 run (PMC passes carry no trace options).  Outputs under gpurun_out/$1_*.
# Usage: tools/collect_profiles.sh <tag>     then, locally:  python tools/summarize_profiles.py <tag>
set -u
TAG=${1:-r02}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out
B="timeout 170 python $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline-pass --no-secondary"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace -o bench -- $B > $OUT/${TAG}_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_pmc_fetch -o bench -- $B > $OUT/${TAG}_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_pmc_write -o bench -- $B > $OUT/${TAG}_pmc_write.log 2>&1
ls $OUT/${TAG}_trace $OUT/${TAG}_pmc_fetch $OUT/${TAG}_pmc_write
grep -h -o '"value": [0-9.]*' $OUT/${TAG}_trace.log $OUT/${TAG}_pmc_fetch.log $OUT/${TAG}_pmc_write.log
