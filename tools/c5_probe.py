"""Debugging aid: per frame of a 1296x968 sequence, what the last full ICP half-iteration did (needs a library built
with GRADSLAM_HIP_BUILD_FLAGS=-DGS_ICP_TIMELINE): queries left open by the 2x2x2 stage, queries that fell through to the
block brute-force scan, duration of the launch.
    GRADSLAM_HIP_ICP_TIMELINE=/tmp/tl.txt python tools/c5_probe.py [frames]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import gradslam_amd as gs  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 60
path = os.environ["GRADSLAM_HIP_ICP_TIMELINE"]
seqs = bench.make_sequences([0], L, 968, 1296)
frames = bench.frames_on_device(gs, seqs, torch.device("cuda", 0))
slam = gs.slam.PointFusion(odom="gradicp", device="cuda")
pc, prev = gs.Pointclouds(device="cuda"), None
for f in range(L):
    live = frames[:, f]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pc, _ = slam.step(pc, live, prev, inplace=True)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3
    prev = live
    if f == 0:
        continue
    rows = np.loadtxt(path, dtype=np.uint64, ndmin=2)
    start, end = rows[:, 1].astype(np.float64), rows[:, 2].astype(np.float64)
    live_b = end > start
    dur = (end[live_b].max() - start[live_b].min()) / 100.0
    nun = int((rows[:, 8] & 0xffffffff).astype(np.int64).sum())
    opened = int(rows[:, 3].astype(np.int64).sum())
    print("frame %3d  step %.2f ms (with dump)  map %8d  launch %.1f us  open after 2x2x2 %6d  brute-force queries %5d  %s"
          % (f, ms, pc._count_of(0)[0], dur, opened, nun, open(path).readline().strip()[:28]), flush=True)
