"""Streaming ingest (datasets/tum.py:352-434, datasets/datautils.py:73-118 do this per item on the
host): raw sensor frames -- uint16 depth, uint8 colour, as decoded from the PNGs -- wait in PINNED host memory, time-major;
while the GPU works on frame t, frame t + 1 crosses PCIe on a side stream (one hipMemcpyAsync per modality for the whole
batch, i.e. the SDMA engines: no compute unit involved) and becomes the float32 images the SLAM step consumes by ONE
launch (gs_ingest_frames_native_f32) on the same stream.  The compute stream only waits for an event.  A ring of RING
device buffers makes it unnecessary for the side stream to wait for anything on the device: the host checks an event that
is RING - 3 steps old before it reuses a slot (it has always fired: the frame loop keeps the host ~6 steps ahead at
most).  That matters: a hipMemcpyAsync enqueued BEHIND a device-side wait blocks the enqueuing thread until the wait
resolves (0.3 - 0.5 ms per copy, measured: the host became the bottleneck of a 1.07 ms step).
Measured at 8 sequences of 640x480 (tools/stream_probe.py, 12.3 MB of raw pixels per step, 49 GB/s when copied alone):
1.111 ms per step streamed against 1.070 resident = 96 %.  zero_copy=True lets the conversion kernel read the pinned
frames itself (the allocation is mapped into the device's address space): no copy engine, but a kernel parked on PCIe loads
for 0.3 ms next to the ICP chain, whose blocks want every SIMD's register file -- 88 %.

    st = FrameStreamer(depth_u16, color_u8, intrinsics, first_poses, scale_div=5000.0, device="cuda")
    for t in range(len(st)):
        live = st.frame(t)          # RGBDImages (B, 1, H, W, .), frame t + 1 already on its way
        pc, live.poses = slam.step(pc, live, prev, inplace=True); prev = live

PyTorch is used for what it is here for: pinned / device memory, streams, events."""
import torch

from .. import ops
from ..structures.rgbdimages import RGBDImages

__all__ = ["FrameStreamer", "quantize_sequences"]


def quantize_sequences(seqs, scale_div=5000.0):
    """synthetic float sequences (datasets/synthetic.py) -> what a sensor would have stored: depth uint16 (metres x
    scale_div, rounded; TUM's png_depth_scale), colour uint8.  Returns pinned, TIME-MAJOR (L, B, H, W) uint16 and
    (L, B, H, W, 3) uint8: the frame of all sequences at time t is one contiguous block, i.e. one copy per modality."""
    import numpy as np
    d = np.stack([np.clip(np.rint(s["depths"][..., 0].astype(np.float64) * scale_div), 0, 65535).astype(np.uint16) for s in seqs], 1)
    c = np.stack([np.clip(np.floor(s["colors"]), 0, 255).astype(np.uint8) for s in seqs], 1)
    d, c = np.ascontiguousarray(d), np.ascontiguousarray(c)
    return torch.from_numpy(d.view(np.int16)).view(torch.uint16).pin_memory(), torch.from_numpy(c).pin_memory()


class FrameStreamer(object):
    RING = 10   # device buffers per modality (10 x 61 MB at 8 sequences of 640x480)

    def __init__(self, depth_u16, color_u8, intrinsics, first_poses, scale_div, device="cuda", normalize_color=False,
                 zero_copy=False):
        """depth_u16 (L, B, H, W) uint16 and color_u8 (L, B, H, W, 3) uint8, time-major and contiguous in pinned host
        memory; intrinsics (B, 1, 4, 4) and first_poses (B, 1, 4, 4) on the device (every frame carries them: the SLAM
        step reads the first frame's pose and overwrites the others)."""
        if not (depth_u16.is_pinned() and color_u8.is_pinned()):
            raise ValueError("FrameStreamer: the raw frames must live in pinned host memory (tensor.pin_memory())")
        if depth_u16.dtype not in (torch.uint16, torch.int16) or color_u8.dtype != torch.uint8:
            raise TypeError("FrameStreamer: depth must be uint16 and colour uint8")
        L, B, H, W = depth_u16.shape
        if tuple(color_u8.shape) != (L, B, H, W, 3) or not depth_u16.is_contiguous() or not color_u8.is_contiguous():
            raise ValueError("FrameStreamer: contiguous time-major frames expected: depth (L, B, H, W), colour (L, B, H, W, 3)")
        self.device = torch.device(device)
        self.B, self.L, self.H, self.W = B, L, H, W
        self.depth_u16, self.color_u8 = depth_u16, color_u8
        self.K, self.P0 = intrinsics, first_poses
        self.scale_div, self.normalize = float(scale_div), bool(normalize_color)
        dev = self.device
        R = self.RING
        self.copy_stream = torch.cuda.Stream(device=dev)
        # zero-copy needs the pinned frames mapped into the device's address space (hipHostMalloc memory is)
        self.zero_copy = bool(zero_copy) and ops.host_device_pointer(depth_u16) is not None and \
            ops.host_device_pointer(color_u8) is not None
        self.raw_d = self.raw_c = None
        if not self.zero_copy:
            self.raw_d = [torch.empty((B, H, W), dtype=depth_u16.dtype, device=dev) for _ in range(R)]
            self.raw_c = [torch.empty((B, H, W, 3), dtype=torch.uint8, device=dev) for _ in range(R)]
        self.f_d = [torch.empty((B, 1, H, W, 1), dtype=torch.float32, device=dev) for _ in range(R)]
        self.f_c = [torch.empty((B, 1, H, W, 3), dtype=torch.float32, device=dev) for _ in range(R)]
        self.ready = [torch.cuda.Event() for _ in range(R)]     # copy + conversion of the slot's frame done
        self.done = {}                                            # frame t -> event: every step up to t has been enqueued
        self.in_slot = [-1] * R
        self._prefetch(0)

    def __len__(self):
        return self.L

    def _prefetch(self, t):
        if t >= self.L:
            return
        k = t % self.RING
        if self.in_slot[k] == t:
            return
        cs = self.copy_stream
        # The slot held frame t - RING, which steps t - RING (as the live frame) and t - RING + 1 (as the previous one)
        # read: both must have finished.  Zero-copy: the side stream waits on the device.  Copy path: the HOST waits --
        # an event RING - 3 frames old has long fired (the frame loop keeps the host ~6 steps ahead of the device at
        # most), whereas a hipMemcpyAsync enqueued behind a device-side wait blocks the enqueuing thread until the wait
        # resolves (measured: 0.3 - 0.5 ms per copy).
        ev = self.done.get(t - self.RING + 1)
        if ev is None and self.in_slot[k] >= 0:
            # frames asked for out of order (or skipped): no event of the steps that read this slot is on record, so
            # everything enqueued on the frame loop's stream so far stands in for them (conservative, always safe)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
        if ev is not None:
            if self.zero_copy:
                cs.wait_event(ev)
            else:
                ev.synchronize()
        with torch.cuda.stream(cs):
            if self.zero_copy:   # the conversion kernel reads the pinned host frames itself
                ops.ingest_frames_native(self.depth_u16[t], self.color_u8[t], self.f_d[k], self.f_c[k], self.scale_div,
                                         self.normalize, on_stream=cs)
            else:                # one contiguous block per modality: two copies per frame of the batch, then the conversion
                self.raw_d[k].copy_(self.depth_u16[t], non_blocking=True)
                self.raw_c[k].copy_(self.color_u8[t], non_blocking=True)
                ops.ingest_frames_native(self.raw_d[k], self.raw_c[k], self.f_d[k], self.f_c[k], self.scale_div,
                                         self.normalize, on_stream=cs)
            self.ready[k].record(cs)
        self.in_slot[k] = t

    def close(self):
        """Waits for the copies in flight (the ring buffers are used on the copy stream: they must not go back to the
        allocator while a transfer still writes them)."""
        if getattr(self, "copy_stream", None) is not None:
            self.copy_stream.synchronize()

    def __del__(self):
        try:
            self.close()
        except Exception:   # noqa: BLE001  (interpreter shutdown)
            pass

    def frame(self, t):
        """RGBDImages of frame t for all sequences (float32 on the device); starts the transfer of frame t + 1.
        The tensors of the returned frame live in a ring slot: they stay valid until frame t + RING - 2 is asked for
        (a SLAM loop reads frame t in steps t and t + 1 only)."""
        self._prefetch(t)
        k = t % self.RING
        cur = torch.cuda.current_stream(self.device)
        if t > 0 and (t - 1) not in self.done:   # everything enqueued so far = the steps up to t - 1
            ev = torch.cuda.Event()
            ev.record(cur)
            self.done[t - 1] = ev
            self.done.pop(t - 1 - 2 * self.RING, None)
        cur.wait_event(self.ready[k])
        self._prefetch(t + 1)
        return RGBDImages(self.f_c[k], self.f_d[k], self.K, self.P0)
