"""Build-time invariants of the ICP half-iteration kernels (no GPU needed: hipcc cross-compiles and reports the register
allocation).  A kernel of the 2 x numiters dependent launches of a solve that touches its private segment costs every
launch microseconds, and a register spill in the prologue of the list variants is a store in the
memory queue behind which every wait for a load drains the gathers in flight -- so ScratchSize 0 is checked, not hoped
for."""
import os
import re
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_half_iteration_kernels_use_no_scratch_and_fit_two_blocks_per_cu():
    r = subprocess.run([sys.executable, os.path.join(REPO, "tools", "kernel_resources.py"), "gs_icp_loop.hip", "gs_icp_half_batch_kernel"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    rows = [ln.split(None, 7) for ln in r.stdout.splitlines() if "gs_icp_half_batch_kernel" in ln]
    # 2 halves x (3 lane counts x 3 list modes + 2 far-list variants + the 8-entry list builder of the persistent solve, 2 lanes)
    assert len(rows) == 24, r.stdout
    for vgpr, sgpr, scratch, occ, sspill, vspill, lds, name in rows:
        assert int(scratch) == 0 and int(vspill) == 0, (name, scratch, vspill)
        assert int(vgpr) <= 80 and int(occ) >= 6, (name, vgpr, occ)          # 2 blocks of 12 waves per CU
        assert int(lds) <= 80 * 1024, (name, lds)                             # ... and of the 160 KB of LDS
    assert re.search(r"<true, 2, false, 2>", r.stdout) and re.search(r"<false, 8, false, 1>", r.stdout)
    assert re.search(r"<false, 2, false, 3>", r.stdout)


def test_persistent_solve_fits_one_block_of_sixteen_waves_per_cu():
    """The opt-in persistent per-XCD solve (csrc/gs_icp_persist.h): one block of 1024 threads per CU needs <= 128 VGPRs and
    <= 160 KB of LDS, or its blocks are not co-resident and its barriers never complete (every wait is bounded, but the
    result would be a NaN pose)."""
    r = subprocess.run([sys.executable, os.path.join(REPO, "tools", "kernel_resources.py"), "gs_icp_loop.hip", "gs_icp_persist_kernel"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    rows = [ln.split(None, 7) for ln in r.stdout.splitlines() if "gs_icp_persist_kernel" in ln]
    assert len(rows) == 1, r.stdout
    vgpr, sgpr, scratch, occ, sspill, vspill, lds, name = rows[0]
    assert int(vgpr) <= 128 and int(occ) >= 4 and int(lds) <= 160 * 1024, rows[0]


def test_hand_counted_memory_waits_of_the_list_checking_kernels_are_covered():
    """the LMODE 2 prologue waits for its hand-issued state load by COUNT (`s_waitcnt vmcnt(N)`); the
    wait covers the load only while at least N vector-memory instructions are issued between the two.  tools/vmcnt_check.py
    reads that off the ISA of every list-checking variant (a toolchain or flag change that sank loads past the wait would
    make every block read a stale state, silently)."""
    r = subprocess.run([sys.executable, os.path.join(REPO, "tools", "vmcnt_check.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr[-2000:]
    rows = [ln for ln in r.stdout.splitlines() if ln.startswith("gs_icp_half_batch_kernel")]
    assert len(rows) == 6 and all(" ok " in ln and "vmcnt(" in ln for ln in rows), r.stdout
