"""Runs gradslam's OWN CPU path (the staged reference, oracle/stage_reference.py) on the host cores and times it:

    python -m oracle.run_reference --frames 3 --height 480 --width 640 --seed 0 --out /tmp/ref.json

PointFusion(odom).step of the unmodified reference (slam/icpslam.py:140-178, slam/pointfusion.py:16-112) on the seeded
synthetic sequence bench.py uses, one step per frame: frame 0 initialises the map, frame 1 is the warm-up of the
localisation path, the remaining frames are timed.  Writes the per-frame seconds and the recovered poses as JSON.
Runs in its own process (bench.py's `cpu_baseline` leg starts it): the reference and its shims never share an
interpreter with the product.  TEST INFRASTRUCTURE ONLY.
"""
import argparse
import json
import os
import sys
import time
import warnings

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=3)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--odom", default="gradicp")
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--scene", default="wave")
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    ref_parent = os.path.join(HERE, "_ref")
    ref_zip = os.path.join(ref_parent, "gradslam_ref.zip")
    if not os.path.isfile(ref_zip):
        raise SystemExit("oracle/_ref/gradslam_ref.zip is not staged (python -m oracle.stage_reference in the build container)")
    for p in (REPO, ref_zip, os.path.join(HERE, "shims")):   # (the archive is imported in place: zipimport)
        sys.path.insert(0, p)
    warnings.filterwarnings("ignore")
    import importlib.util

    import numpy as np
    import torch
    host = len(os.sched_getaffinity(0))
    # PyTorch's intra-op pool on ALL cores of a 256-core host makes the reference's thousands of small tensor operations
    # per frame 30x slower than on 8 cores (measured on the GPU box: 87 s per frame against 2.7 s): the pool is capped at
    # 32 threads (what is used is what is reported); the OpenMP brute-force KNN stand-in keeps every core
    cores = a.threads or min(host, 32)
    torch.set_num_threads(cores)
    os.environ.setdefault("OMP_NUM_THREADS", str(host))
    import gradslam
    assert os.path.realpath(gradslam.__file__).startswith(os.path.realpath(ref_parent)), gradslam.__file__
    from gradslam.slam.pointfusion import PointFusion
    from gradslam.structures.pointclouds import Pointclouds
    from gradslam.structures.rgbdimages import RGBDImages
    spec = importlib.util.spec_from_file_location("_gs_synthetic", os.path.join(REPO, "gradslam_amd", "datasets", "synthetic.py"))
    syn = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(syn)
    L = a.frames
    s = syn.make_sequence(L, a.height, a.width, seed=a.seed, scene=a.scene)
    T = torch.from_numpy
    poses = T(s["poses"][None]).clone()
    poses[:, 1:] = poses[:, :1]
    frames = RGBDImages(T(s["colors"][None]), T(s["depths"][None]), T(s["intrinsics"][None]), poses)
    slam = PointFusion(odom=a.odom)
    pc, prev, secs, rec, counts = Pointclouds(), None, [], [], []
    with torch.no_grad():
        for f in range(L):
            live = frames[:, f]
            t0 = time.perf_counter()
            pc, live.poses = slam.step(pc, live, prev, inplace=True)
            secs.append(time.perf_counter() - t0)
            prev = live
            rec.append(live.poses[0, 0].numpy().tolist())
            counts.append(int(pc.points_list[0].shape[0]))
    timed = secs[2:] if L > 2 else secs[1:]
    out = {"seconds_per_frame": secs, "frames_timed": len(timed), "frames_per_s": len(timed) / sum(timed) if timed else None,
           "cores": cores, "host_cores": host, "torch": torch.__version__, "gradslam_version": getattr(gradslam, "__version__", "?"),
           "poses": rec, "counts": counts, "depth_sum": float(s["depths"].astype(np.float64).sum())}
    with open(a.out, "w") as f:
        json.dump(out, f)


if __name__ == "__main__":
    main()
