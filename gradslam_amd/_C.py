"""ctypes binding of libgradslam_hip.so (include/gradslam_hip.h).

The HIP library is the product: there is NO CPU or PyTorch fallback.  `lib()` raises if the
shared object is missing, and every wrapper raises if it is handed a tensor that does not live
on a HIP device.  torch is used only for device memory, streams and dtype plumbing.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# GRADSLAM_HIP_LIB: another build of the same library (A/B runs of an experimental build on one GPU box); it must export
# the same symbols and ABI version, and there is still no fallback if it is missing
LIB_PATH = os.environ.get("GRADSLAM_HIP_LIB") or os.path.join(_HERE, "csrc", "libgradslam_hip.so")
if os.environ.get("GRADSLAM_HIP_LIB"):   # (a bench line must not come from an experimental build unnoticed)
    import warnings
    warnings.warn("GRADSLAM_HIP_LIB overrides the product library: %s" % LIB_PATH, RuntimeWarning)
ABI_VERSION = 2

_lib = None

_vp, _i64, _i32, _f = C.c_void_p, C.c_int64, C.c_int, C.c_float


class IcpParams(C.Structure):
    _fields_ = [("mode", C.c_int), ("numiters", C.c_int), ("damp", C.c_float),
                ("dist_thresh", C.c_float), ("lambda_max", C.c_float), ("B", C.c_float),
                ("B2", C.c_float), ("nu", C.c_float)]


class MapView(C.Structure):
    """gs_map_view: one sequence's capacity-backed surfel store as the batched entry points see it."""
    _fields_ = [("points", C.c_void_p), ("normals", C.c_void_p), ("colors", C.c_void_p), ("ccounts", C.c_void_p),
                ("capacity", C.c_int64), ("n_bound", C.c_int64), ("n_dev", C.c_void_p)]


class LocalizeSeq(C.Structure):
    _fields_ = [("vertex", C.c_void_p), ("depth", C.c_void_p), ("K16", C.c_void_p), ("prev_pose16", C.c_void_p),
                ("map", MapView), ("out_pose16", C.c_void_p), ("scratch", C.c_void_p)]


class UpdateSeq(C.Structure):
    _fields_ = [("map", MapView), ("vertex", C.c_void_p), ("normal", C.c_void_p), ("depth", C.c_void_p),
                ("rgb", C.c_void_p), ("alpha", C.c_void_p), ("pose16", C.c_void_p), ("K16", C.c_void_p),
                ("gvertex", C.c_void_p), ("gnormal", C.c_void_p), ("best_pix", C.c_void_p),
                ("new_count_out", C.c_void_p), ("scratch", C.c_void_p)]


class StepSeq(C.Structure):
    """gs_step_seq: one sequence's part of gs_pointfusion_step_batch_f32."""
    _fields_ = [("depth", C.c_void_p), ("rgb", C.c_void_p), ("K16", C.c_void_p), ("prev_pose16", C.c_void_p),
                ("out_pose16", C.c_void_p), ("vertex", C.c_void_p), ("normal", C.c_void_p), ("alpha", C.c_void_p),
                ("gvertex", C.c_void_p), ("gnormal", C.c_void_p), ("best_pix", C.c_void_p),
                ("new_count_out", C.c_void_p), ("map", MapView), ("loc_scratch", C.c_void_p),
                ("upd_scratch", C.c_void_p)]


# name -> argtypes (return type is int unless listed in _RESTYPE)
_PROTOS = {
    "gs_abi_version": [],
    "gs_last_error": [],
    "gs_scratch_bytes": [_i64, _i64],
    "gs_profile_begin": [_i32],
    "gs_profile_end": [],
    "gs_profile_read": [_i32, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double)],
    "gs_frame_maps_f32": [_vp, _vp, _i32, _i32, _f, _vp, _vp, _vp, _vp, _vp],
    "gs_global_maps_f32": [_vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp],
    "gs_alpha_f32": [_vp, _i64, _f, _f, _vp, _vp],
    "gs_alpha_backward_f32": [_vp, _i64, _f, _f, _vp, _vp, _vp, _vp],
    "gs_downsample_frame_f32": [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp],
    "gs_project_map_f32": [_vp, _i64, _vp, _vp, _i32, _i32, _vp, _vp],
    "gs_active_table_i64": [_vp, _i64, _i32, _i64, _vp, _vp, _vp, _vp],
    "gs_select_targets_f32": [_vp, _i64, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp],
    "gs_downsample_table_f32": [_vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "gs_knn1_f32": [_vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp],
    "gs_knn1_grid_scratch_bytes": [_i64, _i64],
    "gs_knn1_grid_f32": [_vp, _i64, _vp, _i64, _vp, _vp, _vp, C.POINTER(C.c_int64), _vp],
    "gs_gauss_newton_rows_f32": [_vp, _i64, _vp, _vp, _i64, _f, _vp, _vp, _vp, _vp, _vp, _vp],
    "gs_solve_normal_eq_f32": [_vp, _vp, _vp, _i64, _i32, _f, _vp, _vp],
    "gs_se3_exp_f32": [_vp, _vp, _vp],
    "gs_se3_exp_backward_f32": [_vp, _vp, _vp, _vp],
    "gs_transform_points_f32": [_vp, _i64, _vp, _vp, _vp],
    "gs_icp_scratch_bytes": [_i64, _i64],
    "gs_icp_f32": [_vp, _i64, _vp, _vp, _i64, _vp, _vp, C.POINTER(IcpParams), _vp, _vp, _vp, _vp],
    "gs_icp_trace_f32": [_vp, _i32, _vp, _vp],
    "gs_icp_dc_f32": [_vp, _i64, _vp, _vp, _vp, _i64, _vp, _vp, _vp, C.POINTER(IcpParams), _vp, _vp, _vp, _vp],
    "gs_frame_maps_backward_f32": [_vp, _vp, _i32, _i32, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "gs_frame_maps_backward_kbar_scratch_bytes": [_i32, _i32],
    "gs_global_maps_backward_f32": [_vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp],
    "gs_downsample_frame_backward_f32": [_vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp],
    "gs_icp_tape_bytes": [_i64, _i32],
    "gs_icp_tape_f32": [_vp, _i64, _vp, _vp, _i64, _vp, _vp, C.POINTER(IcpParams), _vp, _vp, _vp, _vp, _vp],
    "gs_icp_backward_scratch_bytes": [_i64, _i64],
    "gs_icp_backward_f32": [_vp, _vp, _i64, _vp, _vp, _i64, _vp, C.POINTER(IcpParams), _vp, _vp, _vp, _vp, _vp, _vp,
                            _vp],
    "gs_similar_rows_f32": [_vp, _i64, _vp, _vp, _vp, _vp, _i32, _f, _f, _vp, _vp],
    "gs_best_unique_rows_f32": [_vp, _i64, _vp, _vp, _vp, _i32, _i32, _i64, _vp, _vp, _vp, _vp, _vp],
    "gs_associate_f32": [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _f, _f, _vp, _vp, _vp, _vp],
    "gs_best_table_i64": [_vp, _i32, _i32, _i64, _vp, _vp, _vp, _vp],
    "gs_rows_to_best_pix": [_vp, _i64, _i32, _i32, _vp, _vp],
    "gs_fuse_append_f32": [_vp, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32,
                           _vp, _vp, _vp],
    "gs_append_valid_f32": [_vp, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp],
    "gs_ingest_depth_u16_f32": [_vp, _i32, _i32, _vp, _i32, _i32, C.c_double, _vp],
    "gs_ingest_color_u8_f32": [_vp, _i32, _i32, _vp, _i32, _i32, _i32, _vp],
    "gs_ingest_frames_native_f32": [_vp, _vp, _i64, _i32, _i32, C.c_double, _i32, _vp, _vp, _vp],
    "gs_host_device_pointer": [_vp, C.POINTER(C.c_void_p)],
    "gs_global_maps_pose_backward_scratch_bytes": [_i32, _i32],
    "gs_global_maps_pose_backward_f32": [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp],
    "gs_fuse_append_backward_f32": [_vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp,
                                    _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "gs_update_map_scratch_bytes": [_i64, _i32, _i32],
    "gs_update_map_fusion_dc_f32": [_vp, _vp, _vp, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32,
                                    _f, _f, _i32, _vp, _vp, _vp, _vp, _vp, _vp],
    "gs_icp_map_dc_f32": [_vp, _i64, _vp, _vp, _vp, _vp, _i64, _vp, _i32, _i32, _vp, _vp, C.POINTER(IcpParams), _vp, _vp,
                          _vp],
    "gs_lattice_source_f32": [_vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp],
    "gs_relative_pose_f32": [_vp, _vp, _i64, _vp, _vp],
    "gs_project_map_dc_f32": [_vp, _i64, _vp, _vp, _vp, _i32, _i32, _vp, _vp],
    "gs_select_targets_dc_f32": [_vp, _i64, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp],
    "gs_associate_dc_f32": [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _f, _f, _vp, _vp, _vp, _vp],
    "gs_fuse_append_dc_f32": [_vp, _vp, _vp, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32,
                              _vp, _vp, _vp],
    "gs_append_valid_dc_f32": [_vp, _vp, _vp, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp,
                               _vp],
    "gs_frame_maps_batch_f32": [_vp, _i64, _i64, _vp, _i32, _i32, _i32, _i32, _f, _vp, _vp, _vp, _vp],
    "gs_localize_scratch_bytes": [_i32, _i32, _i32, _i64],
    "gs_localize_batch_f32": [C.POINTER(LocalizeSeq), _i32, _i32, _i32, _i32, C.POINTER(IcpParams), _vp],
    "gs_update_map_fusion_batch_f32": [C.POINTER(UpdateSeq), _i32, _i32, _i32, _f, _f, _i32, _vp],
    "gs_project_points_f32": [_vp, _i32, _i64, _vp, _i64, _vp, _vp],
    "gs_unproject_points_f32": [_vp, _i32, _i64, _vp, _i64, _vp, _vp, _vp],
    "gs_lie_small_f32": [_i32, _vp, _vp, _vp],
    "gs_localize_far_stats_i64": [_vp, _i32, _i32, _i32, _i64, C.POINTER(C.c_int64), _vp],
    "gs_localize_list_stats_i64": [_vp, _i32, _i32, _i32, _i64, C.POINTER(C.c_int64), _vp],
    "gs_pointfusion_step_batch_f32": [C.POINTER(StepSeq), _i32, _i32, _i32, _i32, C.POINTER(IcpParams), _f, _f, _f, _i32, _vp],
}
_RESTYPE = {"gs_last_error": C.c_char_p, "gs_scratch_bytes": _i64, "gs_icp_scratch_bytes": _i64,
            "gs_knn1_grid_scratch_bytes": _i64, "gs_update_map_scratch_bytes": _i64, "gs_global_maps_pose_backward_scratch_bytes": _i64, "gs_icp_tape_bytes": _i64, "gs_icp_backward_scratch_bytes": _i64,
            "gs_localize_scratch_bytes": _i64, "gs_frame_maps_backward_kbar_scratch_bytes": _i64}
EXPORTS = tuple(_PROTOS)


class HipExtensionError(RuntimeError):
    pass


def lib():
    """Loads libgradslam_hip.so (once).  Fails loudly: there is no fallback path."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipExtensionError(
                "gradslam_amd: %s is missing. Build it with `python -m gradslam_amd.csrc.build` "
                "(hipcc --offload-arch=gfx950). There is no CPU / PyTorch fallback." % LIB_PATH)
        handle = C.CDLL(LIB_PATH)
        for name, argtypes in _PROTOS.items():
            fn = getattr(handle, name)  # AttributeError if the .so does not export it
            fn.argtypes = argtypes
            fn.restype = _RESTYPE.get(name, C.c_int)
        if handle.gs_abi_version() != ABI_VERSION:
            raise HipExtensionError("libgradslam_hip.so ABI %d != expected %d; rebuild"
                                    % (handle.gs_abi_version(), ABI_VERSION))
        _lib = handle
    return _lib


def check(status, what):
    if status != 0:
        raise HipExtensionError("%s failed (code %d): %s" % (what, status, lib().gs_last_error().decode()))


def require_device(*tensors):
    """Every tensor handed to the HIP library must be a contiguous tensor on a HIP device."""
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise HipExtensionError(
                "gradslam_amd kernels run on the GPU only (got a %s tensor). Move the data to a HIP "
                "device; there is no CPU fallback." % t.device)
        if not t.is_contiguous():
            raise HipExtensionError("internal error: non-contiguous tensor passed to the HIP library")
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise HipExtensionError("tensors live on different devices: %s vs %s" % (dev, t.device))
    return dev


def ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def stream(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class Workspace:
    """Scratch buffers per (device, stream), grown on demand and reused every frame (the library never allocates).
    Keyed by the stream the calls are enqueued on: two chains of calls on different streams never share scratch."""
    _instances = {}

    def __init__(self, device):
        self.device = device
        self._bufs = {}

    @classmethod
    def get(cls, device):
        device = torch.device(device)
        if device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        key = (device, torch.cuda.current_stream(device).cuda_stream)
        ws = cls._instances.get(key)
        if ws is None:
            ws = cls._instances[key] = Workspace(device)
        return ws

    def bytes(self, name, nbytes):
        buf = self._bufs.get(name)
        if buf is None or buf.numel() < nbytes:
            nbytes = int(nbytes * 1.5) + 4096
            buf = self._bufs[name] = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        return buf

    def scratch(self, n_map, n_pix):
        return self.bytes("scratch", lib().gs_scratch_bytes(int(n_map), int(n_pix)))

    def tensor(self, name, shape, dtype):
        n = 1
        for s in shape:
            n *= int(s)
        item = torch.empty((), dtype=dtype).element_size()
        buf = self.bytes(name, max(n, 1) * item)
        return buf[: n * item].view(dtype).view(*shape)
