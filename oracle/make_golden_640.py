"""End-to-end golden of the REAL reference at the benchmarked resolution (BASELINE configs[1]): runs
gradslam.slam.PointFusion(odom="gradicp") of /root/reference (oracle/refimport.py + shims, OpenMP brute-force
stand-in for chamferdist's knn_points) on the seeded synthetic 640x480 sequence bench.py uses (seed 0) and records
the recovered poses, the surfel count after every frame, per-frame float64 checksums of the fused points and
the wall time of every frame on this container's host cores.

    python -m oracle.make_golden_640 [--frames 12]

Build-container only (minutes of CPU).  Outputs (committed, travel to the GPU box):
  tests/golden/pf640.npz             poses (L,4,4), counts (L,), point / normal / colour / ccount sums per frame,
                                     a checksum of the input depths (the inputs are regenerated from the seed)
  tests/golden/cpu_ref_timing.json   seconds per frame of the unmodified reference on `cores` host cores
    python -m oracle.make_golden_640 --slam icpslam --odom icp --frames 8 --tag icpslam640
  tests/golden/icpslam640.npz        the same for ICPSLAM (hard-LM ICP odometry, aggregate mapping)
    python -m oracle.make_golden_640 --odom gt --frames 8 --tag pf640_gt
    python -m oracle.make_golden_640 --scene facets --frames 60 --tag facets640_l60
  tests/golden/facets640_l60.npz     the "facets" scene (inclined planes + a ridge): the reference's solves converge on it, so
                                     BASELINE's ATE <= 1e-4 m can be asserted over the whole horizon (round 6)
  tests/golden/pf640_gt.npz          PointFusion with GROUND-TRUTH odometry: the fusion path (K5 / K6) across frames with
                                     no ICP in the loop (+ sha256 of the first map's tables, which are exact)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, REPO)
from oracle import refimport  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=12)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--tag", default="pf640")
    ap.add_argument("--slam", default="pointfusion", choices=["pointfusion", "icpslam"])
    ap.add_argument("--odom", default="gradicp", choices=["gradicp", "icp", "gt"])
    ap.add_argument("--scene", default="wave", choices=["wave", "facets"],
                    help="gradslam_amd.datasets.synthetic scene: the benchmark's smooth height field, or inclined planes + a "
                         "ridge (the scene on which the reference's 20 iterations converge: long-horizon parity)")
    ap.add_argument("--threads", type=int, default=0, help="intra-op threads of the reference (default: all cores)")
    ap.add_argument("--hole-frac", type=float, default=0.05, help="fraction of depth pixels zeroed per frame")
    args = ap.parse_args()
    refimport.import_reference()
    import torch
    from gradslam.slam.icpslam import ICPSLAM
    from gradslam.slam.pointfusion import PointFusion
    from gradslam.structures.pointclouds import Pointclouds
    from gradslam.structures.rgbdimages import RGBDImages
    from gradslam_amd.datasets.synthetic import make_sequence

    cores = len(os.sched_getaffinity(0))
    torch.set_num_threads(args.threads or cores)
    L, H, W = args.frames, args.height, args.width
    s = make_sequence(L, H, W, seed=args.seed, scene=args.scene, hole_frac=args.hole_frac)
    T = torch.from_numpy
    poses = T(s["poses"][None]).clone()
    if args.odom != "gt":   # (ground-truth odometry reads the frames' own poses: slam/icpslam.py:231-236)
        poses[:, 1:] = poses[:, :1]
    frames = RGBDImages(T(s["colors"][None]), T(s["depths"][None]), T(s["intrinsics"][None]), poses)
    slam = (PointFusion if args.slam == "pointfusion" else ICPSLAM)(odom=args.odom)
    pc = Pointclouds()
    prev = None
    rec = np.zeros((L, 4, 4), np.float32)
    counts = np.zeros(L, np.int64)
    sums = {k: np.zeros((L, c), np.float64) for k, c in (("points", 3), ("normals", 3), ("colors", 3), ("ccounts", 1))}
    secs = np.zeros(L)
    import hashlib
    sha0 = {}
    with torch.no_grad():
        for f in range(L):   # slam/icpslam.py:124-137, one step() per frame so that every frame can be recorded
            live = frames[:, f]
            t0 = time.perf_counter()
            pc, live.poses = slam.step(pc, live, prev, inplace=True)
            secs[f] = time.perf_counter() - t0
            prev = live
            rec[f] = live.poses[0, 0].numpy()
            counts[f] = pc.points_list[0].shape[0]
            if f == 0:   # the first map is exact (no fusion yet): fingerprints of its tables
                for k, lst in (("points", pc.points_list), ("normals", pc.normals_list), ("colors", pc.colors_list)):
                    sha0[k] = hashlib.sha256(lst[0].numpy().tobytes()).hexdigest()
            for k, lst in (("points", pc.points_list), ("normals", pc.normals_list), ("colors", pc.colors_list),
                           ("ccounts", pc.features_list)):
                if lst is not None:   # ICPSLAM's aggregate map carries no confidence counts
                    sums[k][f] = lst[0].double().sum(0).numpy()
            print("frame %2d  %.2f s  %d surfels" % (f, secs[f], counts[f]), flush=True)
    np.savez_compressed(os.path.join(OUT, args.tag + ".npz"), poses=rec, counts=counts, gt_poses=s["poses"],
                        depth_sum=np.float64(s["depths"].astype(np.float64).sum()),
                        color_sum=np.float64(s["colors"].astype(np.float64).sum()),
                        seed=np.int64(args.seed), H=np.int64(H), W=np.int64(W), scene=np.array(args.scene), hole_frac=np.float64(args.hole_frac),
                        last_points=pc.points_list[0][-4096:].numpy(),
                        sha_frame0=np.array([sha0.get(k, "") for k in ("points", "normals", "colors")]),
                        **{"sum_" + k: v for k, v in sums.items()})
    timing = {"what": "unmodified gradslam v0.1.0 %s(odom='%s').step on CPU (torch %s), synthetic %dx%d "
                      "sequence seed %d; chamferdist.knn_points replaced by an OpenMP brute-force stand-in "
                      "(oracle/shims), everything else is the reference's own PyTorch code"
                      % ("PointFusion" if args.slam == "pointfusion" else "ICPSLAM", args.odom, torch.__version__, W, H,
                         args.seed),
              "cores": cores, "torch_threads": args.threads or cores, "frames": L, "seconds_per_frame": [float(x) for x in secs],
              "frames_per_s_steady": float((L - 2) / secs[2:].sum()) if L > 3 else None,
              "machine": "build container (not the GPU box's host)"}
    with open(os.path.join(OUT, "cpu_ref_timing" + ("" if args.tag == "pf640" else "_" + args.tag) + ".json"), "w") as f:
        json.dump(timing, f, indent=1)
    print(json.dumps(timing))


if __name__ == "__main__":
    main()
