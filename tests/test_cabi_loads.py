"""CPU-only: the C-ABI library builds for gfx950, loads, and exports every symbol that
include/gradslam_hip.h declares (no compute calls: there is no GPU here)."""
import os
import re

import pytest

from gradslam_amd import _C

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    hdr = open(os.path.join(REPO, "include", "gradslam_hip.h")).read()
    return set(re.findall(r"^GS_API [a-z0-9_ \*]+?\b(gs_[a-z0-9_]+)\(", hdr, flags=re.M))


def test_library_exports_every_declared_symbol():
    if not os.path.exists(_C.LIB_PATH):
        from gradslam_amd.csrc import build
        build.build()
    lib = _C.lib()
    decl = declared_symbols()
    assert len(decl) >= 26
    for name in decl:
        assert hasattr(lib, name), name
    assert decl == set(_C.EXPORTS), decl ^ set(_C.EXPORTS)
    assert lib.gs_abi_version() == _C.ABI_VERSION
    assert lib.gs_scratch_bytes(1000, 64 * 64) > 0
    assert lib.gs_icp_scratch_bytes(256, 256) > 0


def test_host_argument_validation_without_gpu():
    """GS_ERR_INVALID paths return before any HIP call, so they can be exercised on the CPU."""
    lib = _C.lib()
    assert lib.gs_frame_maps_f32(None, None, 4, 4, 0.72, None, None, None, None, None) == 1
    assert b"NULL" in lib.gs_last_error()
    assert lib.gs_knn1_f32(None, 0, None, 0, None, None, None, None) == 1
    assert lib.gs_solve_normal_eq_f32(None, None, None, 5, 9, 1e-8, None, None) == 1


def test_cpu_tensors_are_rejected_loudly():
    import torch
    from gradslam_amd import ops
    with pytest.raises(_C.HipExtensionError, match="no CPU fallback"):
        ops.frame_maps(torch.ones(4, 4), torch.eye(4))
