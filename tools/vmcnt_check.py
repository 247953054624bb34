"""Disassembly check of the hand-counted memory waits in the list-checking half-iteration kernels.

The LMODE 2 prologue of gs_icp_loop.hip issues the copy of the previous state with a hand-written `global_load_dwordx4`
and waits for it with a hand-written `s_waitcnt vmcnt(N)` (N = the gathers that may stay in flight).  Loads return in
order, so the wait covers the state copy only if AT LEAST N vector-memory instructions are issued between the two; if a
toolchain or flag change sank loads past the wait, the state would be read stale without any error.  This tool
cross-compiles the source to ISA (no GPU needed) and, per list-checking kernel, counts the vector-memory instructions
between the hand-issued load and every hand-written vmcnt wait that follows it before the first barrier.

    python tools/vmcnt_check.py            # one line per kernel; exit status 1 if a wait is not covered
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gradslam_amd.csrc.build import FLAGS, _hipcc   # noqa: E402

src = os.path.join(ROOT, "gradslam_amd", "csrc", "gs_icp_loop.hip")
with tempfile.TemporaryDirectory() as td:
    out = os.path.join(td, "loop.s")
    flags = [f for f in FLAGS if f != "-fPIC"]
    r = subprocess.run([_hipcc()] + flags + ["--cuda-device-only", "-S", "-o", out, src], capture_output=True, text=True)
    if r.returncode != 0:
        sys.exit(r.stderr[-3000:])
    lines = open(out).read().splitlines()
VMEM = re.compile(r"^\s*(global_load|global_store|global_atomic|buffer_load|buffer_store|buffer_atomic|scratch_load|scratch_store|flat_load|flat_store)")
bad = 0
i = 0
print("# kernel: vector-memory instructions between the hand-issued state load and each hand-written `s_waitcnt vmcnt(N)` before the first barrier")
while i < len(lines):
    m = re.match(r"^(_Z24gs_icp_half_batch_kernelILb([01])ELi(\d)ELb0ELi2EE\w*):", lines[i])
    if not m:
        i += 1
        continue
    name = "gs_icp_half_batch_kernel<%s, %s, false, 2>" % ("true" if m.group(2) == "1" else "false", m.group(3))
    j, in_app, issued, count, waits = i + 1, False, False, 0, []
    while j < len(lines) and not lines[j].startswith("\t.end_amdhsa_kernel") and "s_endpgm" not in lines[j]:
        ln = lines[j]
        if "#ASMSTART" in ln or ";APP" in ln:
            in_app = True
        elif "#ASMEND" in ln or ";NO_APP" in ln:
            in_app = False
        elif in_app and "global_load_dwordx4" in ln and not issued:
            issued = True
        elif issued and in_app and re.search(r"s_waitcnt vmcnt\((\d+)\)", ln):
            waits.append((int(re.search(r"vmcnt\((\d+)\)", ln).group(1)), count))
        elif issued and VMEM.match(ln):
            count += 1
        elif issued and "s_barrier" in ln:
            break
        j += 1
    ok = issued and waits and all(c >= n for n, c in waits)
    bad += 0 if ok else 1
    print("%-46s %s  %s" % (name, "ok " if ok else "BAD", ", ".join("vmcnt(%d) behind %d" % w for w in waits) or "no hand-written wait found"))
    i = j
sys.exit(1 if bad else 0)
