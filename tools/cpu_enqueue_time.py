"""How long does the HOST need to enqueue one PointFusion frame (no waiting for the GPU)?
Compares the enqueue time with the end-to-end time per frame: the larger of the two bounds fps."""
import sys, time, numpy as np, torch
sys.path.insert(0, ".")
import gradslam_amd as gs
from gradslam_amd.datasets.synthetic import make_sequence

L, W0 = 43, 3
seq = make_sequence(L, 480, 640, seed=0)
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
poses = seq["poses"].copy(); poses[1:] = poses[:1]
frames = gs.RGBDImages(T(seq["colors"][None]), T(seq["depths"][None]), T(seq["intrinsics"][None]), T(poses[None]))
slam = gs.slam.PointFusion(odom="gradicp", device="cuda")
for rep in range(2):
    pc, prev = gs.Pointclouds(device="cuda"), None
    for s in range(L):
        if s == W0:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        live = frames[:, s]
        pc, _ = slam.step(pc, live, prev, inplace=True); prev = live
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    n = L - W0
    print("rep %d: host enqueue %.3f ms/frame, end-to-end %.3f ms/frame" % (rep, (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3))
import cProfile, pstats
pc, prev = gs.Pointclouds(device="cuda"), None
pr = cProfile.Profile()
for s in range(L):
    if s == W0: pr.enable()
    live = frames[:, s]
    pc, _ = slam.step(pc, live, prev, inplace=True); prev = live
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
