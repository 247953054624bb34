"""Probe: the 8 benchmark sequences of one GPU as C independent chains of calls (8 / C sequences each) on C HIP
streams instead of one batch on one stream.  A chain's latency-bound ICP launches can then overlap the other
chains' HBM-bound map passes.  python tools/two_chain_probe.py --chains 1 2 4 [--steps 20 --warmup 5]"""
import argparse
import gc
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import gradslam_amd as gs  # noqa: E402


def run(seqs, chains, K, Wm, device, stagger):
    B = len(seqs)
    per = B // chains
    L = Wm + K
    frames = [bench.frames_on_device(gs, seqs[c * per:(c + 1) * per], device) for c in range(chains)]
    streams = [torch.cuda.Stream(device) for _ in range(chains)] if chains > 1 else [torch.cuda.current_stream(device)]
    slam = gs.slam.PointFusion(odom="gradicp", device=device)
    pcs = [gs.Pointclouds(device=device) for _ in range(chains)]
    prevs = [None] * chains
    poses = [[] for _ in range(chains)]
    torch.cuda.synchronize(device)

    def steps(first, last):
        for s in range(first, last):
            for c in range(chains):
                with torch.cuda.stream(streams[c]):
                    live = frames[c][:, s]
                    pcs[c], pose = slam.step(pcs[c], live, prevs[c], inplace=True)
                    poses[c].append(pose[:, 0])
                    prevs[c] = live

    steps(0, Wm)
    torch.cuda.synchronize(device)
    if stagger and chains > 1:
        # chain c starts c / chains of a step late: its ICP phase then faces the other chains' map passes
        for c in range(1, chains):
            with torch.cuda.stream(streams[c]):
                torch.cuda._sleep(int(stagger * c * 2.4e6 / chains))  # ~cycles at 2.4 GHz per ms
    t0 = time.perf_counter()
    steps(Wm, L)
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize(device)
    dt = time.perf_counter() - t0
    P = torch.cat([torch.stack(p, 1) for p in poses]).cpu().numpy()
    n_map = [int(p.shape[0]) for pc in pcs for p in pc.points_list]
    return B * K / dt, dt / K * 1e3, t_enq / K * 1e3, P, n_map


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chains", type=int, nargs="+", default=[1, 2, 4])
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--stagger-ms", type=float, default=0.0)
    ap.add_argument("--repeat", type=int, default=2)
    a = ap.parse_args()
    device = torch.device("cuda", 0)
    seqs = bench.make_sequences(list(range(a.batch)), a.warmup + a.steps, 480, 640)
    gc.collect()
    gc.freeze()
    ref = None
    for c in a.chains:
        for r in range(a.repeat):
            fps, ms, enq, P, n_map = run(seqs, c, a.steps, a.warmup, device, a.stagger_ms)
            same = None if ref is None else bool(np.array_equal(P, ref[0]) and n_map == ref[1])
            if ref is None:
                ref = (P, n_map)
            print("chains %d  run %d: %8.1f frames/s  %.3f ms/step  host enqueue %.3f ms/step  identical to first: %s"
                  % (c, r, fps, ms, enq, same), flush=True)


if __name__ == "__main__":
    main()
