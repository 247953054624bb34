#!/bin/bash
# PMC counters of the ICP half-iteration kernels for one variant (run on the GPU box through gpurun).
# Usage: tools/pmc_icp.sh <tag> [env assignments...]   -> gpurun_out/<tag>_pmc{1,2}/ + gpurun_out/<tag>_pmc.txt
set -u
TAG=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out
B="python $ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline-pass"
cd /tmp && export TMPDIR=/tmp
env "$@" rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU \
  --kernel-include-regex "gs_icp_half" --output-format csv -d $OUT/${TAG}_pmc1 -o p -- $B > $OUT/${TAG}_pmc1.log 2>&1
env "$@" rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM \
  --kernel-include-regex "gs_icp_half" --output-format csv -d $OUT/${TAG}_pmc2 -o p -- $B > $OUT/${TAG}_pmc2.log 2>&1
env "$@" rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE \
  --kernel-include-regex "gs_icp_half" --output-format csv -d $OUT/${TAG}_pmc3 -o p -- $B > $OUT/${TAG}_pmc3.log 2>&1
python - <<PY
import csv, glob, collections
out = open("$OUT/${TAG}_pmc.txt", "w")
for k in (1, 2, 3):
    files = glob.glob("$OUT/${TAG}_pmc%d/**/*counter_collection.csv" % k, recursive=True)
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in files:
        for r in csv.DictReader(open(f)):
            a = acc[(r["Kernel_Name"].split("(")[0][:60], r["Counter_Name"])]
            a[0] += float(r["Counter_Value"]); a[1] += 1
    for (kn, cn), (v, n) in sorted(acc.items()):
        out.write("%-62s %-24s %14.1f per launch (%d launches)\n" % (kn, cn, v / n, n))
out.close()
print(open("$OUT/${TAG}_pmc.txt").read())
PY
