"""The in-place PointFusion frame loop with ONE foreign call per frame.

`ICPSLAM.step` -> `_localize` -> `_map` (slam/icpslam.py:140-264, slam/pointfusion.py:107-112) costs the reference
a few hundred tensor operations per frame; the generic path of this package still spends ~0.9 ms of Python per frame
on container bookkeeping around its three batched entry points, which is as long as the GPU needs for the frame
itself (8 sequences of 640x480).  A `StepPlan` keeps everything that does not change between frames (the descriptor
array of gs_pointfusion_step_batch_f32, scratch, solver parameters, two device count buffers) and per frame only
rewrites the pointers that move: frame slices, pose buffers, count bounds.  The kernels and their order are those of
the generic path (same C functions), so the results are bit-identical
(tests/test_hip_batch.py::test_fast_path_equals_generic_path).

Used by `PointFusion.step(..., inplace=True)` when nothing is on the autograd tape and the map already holds surfels
with device-side counts; everything else takes the generic path."""
import os

import torch

from .. import _C, ops
from ..structures.pointclouds import _ATTRS
from .fusionutils import RENORMALIZE_UNMATCHED

__all__ = ["try_step"]

f32 = torch.float32
FASTPATH = os.environ.get("GRADSLAM_HIP_FASTPATH", "1") != "0"   # 0: always the generic path (A/B runs)
# 1: the step also writes the live frame's GLOBAL vertex / normal maps (one more per-pixel pass per frame).  Default 0:
# the update computes them where it uses them and the container computes them when somebody asks (the same bits).
STEP_GLOBAL_MAPS = os.environ.get("GRADSLAM_HIP_STEP_GLOBAL_MAPS", "0") == "1"


def _kwargs_key(kw):
    """hashable, comparable form of the solver keywords.  A tensor-valued keyword (dist_thresh) is keyed by identity and
    version -- storage, shape, in-place modification counter -- NOT by value: reading a device tensor's value is a blocking
    device-to-host copy, and this key is formed on every step of a loop that is meant to run without a host / device
    synchronisation.  The value is converted once, when a new StepPlan is built."""
    def norm(v):
        if torch.is_tensor(v):
            # (id, not data_ptr: the plan keeps the tensors of its key alive -- StepPlan.kw_tensors -- so neither the object nor
            # its storage address can be reused by another tensor while the plan exists, ADVICE r05.  A write that bypasses the
            # version counter -- t.data.fill_(x), a foreign kernel -- is not seen: change a tensor threshold by assignment or
            # by an in-place torch operation.)
            return ("tensor", id(v), int(v._version), tuple(v.shape), str(v.dtype), str(v.device))
        return v
    return tuple(sorted((k, norm(v)) for k, v in kw.items()))


def _dense_frames(t, inner):
    """(B, 1, H, W, C) float32 tensor whose (H, W, C) frames are contiguous and equally strided over b -> (ptr, stride
    in elements) or None."""
    if t.dtype != f32 or t.stride()[2:] != inner:
        return None
    B = t.shape[0]
    sb = t.stride(0) if B > 1 else inner[0] * t.shape[2]
    if B > 1 and sb < inner[0] * t.shape[2]:
        return None
    return t.data_ptr(), sb


class StepPlan(object):
    def __init__(self, slam, B, H, W, device):
        self.B, self.H, self.W, self.P, self.device = B, H, W, H * W, device
        self.lib = _C.lib()
        self.seqs = (_C.StepSeq * B)()
        prov = slam.odomprov
        kw = prov._kwargs()
        self.kw_tensors = [v for v in kw.values() if torch.is_tensor(v)]   # (kept alive: see _kwargs_key)
        self.prm = _C.IcpParams(int(prov._mode), int(kw.get("numiters", 20)), float(kw.get("damp", 1e-8)),
                                ops._thresh(kw.get("dist_thresh")), float(kw.get("lambda_max", 2.0)),
                                float(kw.get("B", 1.0)), float(kw.get("B2", 1.0)), float(kw.get("nu", 200.0)))
        self.key = (float(slam.sigma), float(slam.dist_th), float(slam.dot_th), int(slam.dsratio), _kwargs_key(kw),
                    int(prov._mode))
        self.sigma = float(slam.sigma)
        self.tss = ops.two_sigma_sq(slam.sigma)
        self.dist_th, self.dot_th, self.ds = float(slam.dist_th), float(slam.dot_th), int(slam.dsratio)
        self.caps = [-1] * B          # capacity the descriptors / scratch of sequence b were set up for
        self.bufs = [None] * B        # (points, normals, colors, features) tensors behind the descriptors
        self.scratch = [None] * B     # (localize, update) scratch tensors
        self.stream = _C.stream(device)
        self.stream_id = torch.cuda.current_stream(device).cuda_stream

    def _bind_map(self, pc, b):
        """descriptor fields that only change when the buffers of sequence b are reallocated"""
        bufs = tuple(pc._buf[k][b] for k in _ATTRS)
        cap = int(bufs[0].shape[0])
        if any(t.shape[0] != cap for t in bufs):
            return False
        q = self.seqs[b]
        m = q.map
        m.points, m.normals, m.colors, m.ccounts = (t.data_ptr() for t in bufs)
        m.capacity = cap
        if cap != self.caps[b]:
            ws = _C.Workspace.get(self.device)
            loc = ws.bytes("localize%d" % b, self.lib.gs_localize_scratch_bytes(self.H, self.W, self.ds, cap))
            upd = ws.bytes("map_update%d" % b, self.lib.gs_update_map_scratch_bytes(cap, self.H, self.W))
            self.scratch[b] = (loc, upd)
            q.loc_scratch, q.upd_scratch = loc.data_ptr(), upd.data_ptr()
            self.caps[b] = cap
        self.bufs[b] = bufs
        return True

    def run(self, pc, live, prev):
        B, P, H, W = self.B, self.P, self.H, self.W
        dev = self.device
        d = _dense_frames(live._depth_image, (W, 1, 1))
        r = _dense_frames(live._rgb_image, (3 * W, 3, 1))
        K, pp = live._intrinsics, prev._poses
        if d is None or r is None or K.dtype != f32 or pp.dtype != f32 or not K.is_contiguous() or not pp.is_contiguous():
            return None
        grp = pc._dcount[0].group
        grp.poll()
        bounds = grp.bounds
        # room for this frame's appends (geometric growth; the descriptors follow the new buffers)
        for b in range(B):
            bufs = self.bufs[b]
            # (every attribute by identity: a setter of normals / colors / features swaps that buffer alone, and another
            # map stepped with the same slam object brings four other buffers)
            if (bufs is None or any(bufs[i] is not pc._buf[k][b] for i, k in enumerate(_ATTRS)) or
                    bounds[b] + P > self.caps[b]):
                if bounds[b] + P > int(pc._buf["points"][b].shape[0]):
                    pc._reserve(b, P, pc.RESERVE_FRAMES)
                    bounds = grp.bounds      # (_reserve may have replaced the bounds by the exact counts)
                if not self._bind_map(pc, b):
                    return None
        # outputs of this frame: one allocation, carved into the local / global maps and the poses
        # (no global maps: the step computes them where the update uses them and never writes them; the container
        # computes them from the local maps and the pose when somebody asks -- the same bits)
        gm = STEP_GLOBAL_MAPS
        out = torch.empty(B * (13 if gm else 7) * P, dtype=f32, device=dev)
        # The poses live in an allocation of their own (64 B per sequence): a caller that keeps the recovered poses of every
        # frame -- every SLAM loop does, slam/icpslam.py:134-137 -- would otherwise pin the 69 MB of maps they were carved
        # from (8 sequences of 640x480), the caching allocator could never hand a block back, and every step would pay a
        # fresh hipMalloc of that size on the host thread: usually ~0.1 ms, but 2 ms on a bad day of a shared host, which
        # made one bench run in six host-bound at half the rate (round 5, profiles/r05_bench_line_slow_host.json).
        poses_out = torch.empty(B * 16, dtype=f32, device=dev)
        best = torch.empty((B, P), dtype=torch.int32, device=dev)
        o = out.data_ptr()
        v0, n0, a0, po0 = o, o + 12 * P * B, o + 24 * P * B, poses_out.data_ptr()
        gv0, gn0 = o + 28 * P * B, o + 28 * P * B + 12 * P * B
        # the new counts go to a buffer the MAP owns (two per count group, alternating): a plan serves whatever map it is
        # handed, and a map's live device count must not be a word some other map's next frame writes
        pair = getattr(grp, "_step_counts", None)
        if pair is None or pair[0].shape[0] != B or pair[0].device != dev:
            pair = grp._step_counts = [torch.zeros(B, dtype=torch.int64, device=dev) for _ in range(2)]
        cnt_new = pair[0] if pair[0].data_ptr() != grp.dev.data_ptr() else pair[1]
        d0, ds_ = d
        r0, rs_ = r
        k0, p0, c0, n_dev0, b0 = K.data_ptr(), pp.data_ptr(), cnt_new.data_ptr(), grp.dev.data_ptr(), best.data_ptr()
        seqs = self.seqs
        for b in range(B):
            q = seqs[b]
            q.depth, q.rgb = d0 + 4 * ds_ * b, r0 + 4 * rs_ * b
            q.K16, q.prev_pose16, q.out_pose16 = k0 + 64 * b, p0 + 64 * b, po0 + 64 * b
            q.vertex, q.normal, q.alpha = v0 + 12 * P * b, n0 + 12 * P * b, a0 + 4 * P * b
            q.gvertex, q.gnormal = (gv0 + 12 * P * b, gn0 + 12 * P * b) if gm else (None, None)
            q.best_pix = b0 + 4 * P * b
            q.new_count_out = c0 + 8 * b
            q.map.n_bound = bounds[b]
            q.map.n_dev = n_dev0 + 8 * b
        _C.check(self.lib.gs_pointfusion_step_batch_f32(seqs, B, H, W, self.ds, self.prm, self.tss, self.dist_th,
                                                        self.dot_th, 1 if RENORMALIZE_UNMATCHED else 0, self.stream), "gs_pointfusion_step_batch_f32")
        # hand the results to the containers (the caches the generic path fills)
        n3 = 3 * P * B
        live._vertex_map = out[:n3].view(B, 1, H, W, 3)
        live._normal_map = out[n3:2 * n3].view(B, 1, H, W, 3)
        live._alpha_cache = (self.sigma, out[2 * n3:2 * n3 + P * B].view(B, 1, H, W, 1))
        live._poses = poses_out.view(B, 1, 4, 4)
        if gm:
            g = 7 * P * B
            live._global_vertex_map = out[g:g + n3].view(B, 1, H, W, 3)
            live._global_normal_map = out[g + n3:g + 2 * n3].view(B, 1, H, W, 3)
        else:
            live._global_vertex_map = live._global_normal_map = None    # (computed on demand under the new pose)
        grp.advance(cnt_new, P)
        pc._padded_cache.clear()
        pc.equisized = True if B == 1 else None
        return pc, live._poses


def try_step(slam, pointclouds, live_frame, prev_frame):
    """PointFusion.step(pointclouds, live_frame, prev_frame, inplace=True) through a StepPlan, or None when the call
    is not the plain SLAM-loop case (the caller then takes the generic path)."""
    if not ops.DEVICE_COUNTS or not FASTPATH or prev_frame is None or slam.odom not in ("icp", "gradicp"):
        return None
    B = len(pointclouds._n_host)
    dc = pointclouds._dcount
    if B == 0 or len(dc) != B:
        return None
    g0 = dc[0].group
    if len(g0.bounds) != B or any(dc[b].group is not g0 or dc[b].index != b for b in range(B)) or min(g0.bounds) <= 0:
        return None
    if live_frame._channels_first or prev_frame._channels_first or live_frame._L != 1 or live_frame._B != B:
        return None
    if prev_frame._poses is None or live_frame._intrinsics.data_ptr() != prev_frame._intrinsics.data_ptr():
        return None
    if tuple(prev_frame._poses.shape) != (B, 1, 4, 4):   # (poses are addressed as base + 64 b)
        return None
    dev = live_frame.device
    if dev.type != "cuda" or pointclouds.device != dev or prev_frame.device != dev:
        return None
    bufs = pointclouds._buf
    if any(bufs[k] is None for k in _ATTRS) or bufs["points"][0].dtype != f32 or bufs["features"][0].shape[-1] != 1:
        return None
    if torch.is_grad_enabled() and (live_frame._depth_image.requires_grad or live_frame._rgb_image.requires_grad or
                                    prev_frame._poses.requires_grad or
                                    any(t.requires_grad for k in _ATTRS for t in bufs[k])):
        return None
    from ..odometry.gradicp import GradICPOdometryProvider
    from ..odometry.icp import ICPOdometryProvider
    if type(slam.odomprov) not in (ICPOdometryProvider, GradICPOdometryProvider):
        return None
    H, W = live_frame.h, live_frame.w
    plan = getattr(slam, "_step_plan", None)
    key = (float(slam.sigma), float(slam.dist_th), float(slam.dot_th), int(slam.dsratio),
           _kwargs_key(slam.odomprov._kwargs()), int(slam.odomprov._mode))
    if (plan is None or (plan.B, plan.H, plan.W, plan.device) != (B, H, W, dev) or plan.key != key or
            plan.stream_id != torch.cuda.current_stream(dev).cuda_stream):
        plan = slam._step_plan = StepPlan(slam, B, H, W, dev)
    return plan.run(pointclouds, live_frame, prev_frame)
