"""Per frame of the 1296x968 sequence: step time and what the candidate lists of far ICP queries did
(gs_localize_far_stats_i64: source points handed to the list builder, how many still prove on their list at the last
search).      python tools/c5_far_probe.py [frames] [first_frame_printed]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import gradslam_amd as gs  # noqa: E402
from gradslam_amd import ops  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 60
first = int(sys.argv[2]) if len(sys.argv) > 2 else 1
H, W = 968, 1296
dev = torch.device("cuda", 0)
seqs = bench.make_sequences([0], L, H, W)
frames = bench.frames_on_device(gs, seqs, dev)
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
    far, listed, far2, built = ops.localize_far_stats(dev, 0, H, W, 4, pc._buf["points"][0].shape[0])
    if f >= first and (ms > 1.9 * (1 + f / 400.0) or f % 10 == 0 or os.environ.get("GRADSLAM_HIP_DEBUG_GRID")):
        print("frame %3d  step %.2f ms  map %8d  far %5d (second pass %5d)  lists that fit %5d  proven at the end %5d"
              % (f, ms, pc._count_of(0)[0], far, far2, built, listed), flush=True)
