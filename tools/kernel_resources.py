"""Register / LDS / scratch report of the kernels of one source (hipcc -Rpass-analysis=kernel-resource-usage), one line per
kernel.  Runs without a GPU (cross-compilation).

    python tools/kernel_resources.py gs_icp_loop.hip [substring of the kernel name ...]
"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HERE = os.path.join(ROOT, "gradslam_amd", "csrc")
sys.path.insert(0, ROOT)
from gradslam_amd.csrc.build import FLAGS, _hipcc   # noqa: E402
src = sys.argv[1]
pats = sys.argv[2:]
r = subprocess.run([_hipcc()] + FLAGS + ["-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(HERE, src), "-o", "/dev/null"],
                   capture_output=True, text=True)
if r.returncode != 0:
    sys.exit(r.stderr[-4000:])
cur, rows = None, []
for line in r.stderr.splitlines():
    m = re.search(r"remark: (?:\S+: )?\s*(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (\S+)", line)
    if not m:
        continue
    k, v = m.group(1), m.group(2)
    if k == "Function Name":
        cur = {"name": subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip() or v}
        rows.append(cur)
    elif cur is not None:
        cur[k] = v
print("# %s   flags: %s" % (src, " ".join(FLAGS)))
print("# VGPR SGPR scratch occupancy(waves/SIMD) sgpr_spill vgpr_spill LDS(bytes)  kernel")
for c in rows:
    name = re.sub(r"\(.*", "", c["name"])
    if pats and not any(p in name for p in pats):
        continue
    print("%4s %4s %7s %9s %10s %10s %9s  %s" % (c.get("VGPRs"), c.get("TotalSGPRs"), c.get("ScratchSize [bytes/lane]"),
          c.get("Occupancy [waves/SIMD]"), c.get("SGPRs Spill"), c.get("VGPRs Spill"), c.get("LDS Size [bytes/block]"), name))
