"""Where does the streamed frame loop lose time against the resident one?  (tools; GPU)
    python tools/stream_probe.py [B] [steps]
Variants: resident | streamed | streamed without the host->device copies (raw frames already in HBM) | copies only."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gradslam_amd as gs
from gradslam_amd import ops
from gradslam_amd.datasets.synthetic import make_sequence
from gradslam_amd.datasets.streaming import FrameStreamer, quantize_sequences
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
Wm, H, W = 5, 480, 640
dev = torch.device("cuda")
seqs = [make_sequence(Wm + K, H, W, seed=b) for b in range(B)]
d16, c8 = quantize_sequences(seqs, 5000.0)
Ks = torch.from_numpy(np.stack([s["intrinsics"] for s in seqs])).to(dev)
P0 = torch.from_numpy(np.stack([s["poses"][:1] for s in seqs])).to(dev)

def loop(get, label):
    slam = gs.slam.PointFusion(odom="gradicp", device=dev)
    pc, prev = gs.Pointclouds(device=dev), None
    for t in range(Wm):
        live = get(t); pc, _ = slam.step(pc, live, prev, inplace=True); prev = live
    torch.cuda.synchronize()
    t0 = time.perf_counter(); th = 0.0
    for t in range(Wm, Wm + K):
        a = time.perf_counter()
        live = get(t)
        th += time.perf_counter() - a
        pc, _ = slam.step(pc, live, prev, inplace=True); prev = live
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%-34s %.3f ms/step   host loop %.3f ms/step (get(): %.3f)   %.0f frames/s" % (
        label, (t2 - t0) / K * 1e3, (t1 - t0) / K * 1e3, th / K * 1e3, B * K / (t2 - t0)))

dres = torch.empty((Wm + K, B, H, W, 1), dtype=torch.float32, device=dev)
cres = torch.empty((Wm + K, B, H, W, 3), dtype=torch.float32, device=dev)
ops.ingest_frames_native(d16.to(dev), c8.to(dev), dres, cres, 5000.0)
fr = gs.RGBDImages(cres.transpose(0, 1), dres.transpose(0, 1), Ks, P0.repeat(1, Wm + K, 1, 1))
loop(lambda t: fr[:, t], "resident")
st = FrameStreamer(d16, c8, Ks, P0, 5000.0, device=dev, zero_copy=True)
loop(st.frame, "streamed (zero-copy=%s)" % st.zero_copy)
st = FrameStreamer(d16, c8, Ks, P0, 5000.0, device=dev, zero_copy=False)
loop(st.frame, "streamed (hipMemcpyAsync)")

class NoCopy(FrameStreamer):   # raw frames already on the device: conversion + events only
    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.dd, self.cc = self.depth_u16.to(self.device), self.color_u8.to(self.device)
        self.depth_u16, self.color_u8 = self.dd, self.cc
        self.in_slot = [-1] * self.RING
        self._prefetch(0)
st2 = NoCopy(d16, c8, Ks, P0, 5000.0, device=dev, zero_copy=False)
loop(st2.frame, "streamed, raw frames in HBM")
# the copies alone
cs = torch.cuda.Stream()
rd = torch.empty((B, H, W), dtype=d16.dtype, device=dev); rc = torch.empty((B, H, W, 3), dtype=torch.uint8, device=dev)
torch.cuda.synchronize(); t0 = time.perf_counter()
with torch.cuda.stream(cs):
    for t in range(Wm + K):
        rd.copy_(d16[t], non_blocking=True); rc.copy_(c8[t], non_blocking=True)
te = time.perf_counter()
torch.cuda.synchronize(); t1 = time.perf_counter()
mb = (Wm + K) * B * H * W * 5 / 1e6
print("host->device copies alone: %.3f ms per frame of the batch (enqueue %.3f), %.1f GB/s" % ((t1 - t0) / (Wm + K) * 1e3, (te - t0) / (Wm + K) * 1e3, mb / 1e3 / (t1 - t0)))
