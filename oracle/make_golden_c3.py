"""Config C3 at the benchmarked size: the REAL reference's autograd through point_to_plane_gradICP
(odometry/icputils.py:479-545, 20 iterations) between two 640x480 frames on the ds = 4 lattice (~18k x ~18k points):
gradients of <W, T> with respect to the source points, the target points and the target normals.

    python -m oracle.make_golden_c3
    python -m oracle.make_golden_c3 --driver     (the whole chain through the SLAM driver, see driver())

Build-container only (a minute of CPU).  Output (committed): tests/golden/c3_grad640.npz -- the transform, every 8th
row of the three gradients, their float64 column sums and norms, and checksums of the inputs (the inputs are
regenerated from the seed on the GPU box: the frame maps and the down-sampler are bit-exact, tests/test_hip_api.py)."""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, REPO)
from oracle import refimport  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")
SEED, STRIDE = 8, 8


def driver():
    """BASELINE configs[2] / SURVEY C3 exactly as stated: depth.requires_grad_(); PointFusion(odom="gradicp") over two
    640x480 frames; recovered_poses.sum().backward(); depth.grad of BOTH frames (frame 1 reaches pose 1 through the ICP
    source, frame 0 through the map it was fused into: targets and normals).  -> tests/golden/c3_driver640.npz: every
    8th image row of the two gradient images, their float64 sums, norms and supports."""
    refimport.import_reference()
    import torch
    from gradslam.slam.pointfusion import PointFusion
    from gradslam.structures.rgbdimages import RGBDImages
    from gradslam_amd.datasets.synthetic import make_sequence

    torch.set_num_threads(len(os.sched_getaffinity(0)))
    T = torch.from_numpy
    s = make_sequence(2, 480, 640, seed=SEED)
    depth = T(s["depths"][None]).clone().requires_grad_(True)
    poses = T(s["poses"][None]).clone()
    poses[:, 1:] = poses[:, :1]
    frames = RGBDImages(T(s["colors"][None]), depth, T(s["intrinsics"][None]), poses)
    t0 = time.perf_counter()
    _, rp = PointFusion(odom="gradicp")(frames)
    rp.sum().backward()
    secs = time.perf_counter() - t0
    gr = depth.grad[0, :, :, :, 0].numpy()   # (2, H, W)
    g = dict(seed=np.int64(SEED), stride=np.int64(STRIDE), seconds=np.float64(secs), poses=rp.detach().numpy()[0],
             depth_sum=np.float64(s["depths"].astype(np.float64).sum()),
             grad_rows=gr[:, ::STRIDE].copy(), grad_sum=gr.astype(np.float64).sum((1, 2)),
             grad_norm=np.sqrt((gr.astype(np.float64) ** 2).sum((1, 2))), grad_absmax=np.abs(gr).max((1, 2)),
             support=(gr != 0).sum((1, 2)).astype(np.int64))
    np.savez_compressed(os.path.join(OUT, "c3_driver640.npz"), **g)
    print("PointFusion(gradicp) 2 x 640x480 forward + backward of the reference: %.1f s" % secs)
    print({k: (v.shape if hasattr(v, "shape") and v.ndim else v) for k, v in g.items() if k != "grad_rows"})


def main():
    if "--driver" in sys.argv:
        return driver()
    refimport.import_reference()
    import torch
    from gradslam.odometry import icputils
    from gradslam.structures.rgbdimages import RGBDImages
    from gradslam_amd.datasets.synthetic import make_sequence

    torch.set_num_threads(len(os.sched_getaffinity(0)))
    T = torch.from_numpy
    s = make_sequence(2, 480, 640, seed=SEED)
    fr = RGBDImages(T(s["colors"][None]), T(s["depths"][None]), T(s["intrinsics"][None]), T(s["poses"][None, :1].repeat(2, 1)))
    tgt_pc = icputils.downsample_rgbdimages(fr[:, 0], 4)
    src_pc = icputils.downsample_rgbdimages(fr[:, 1], 4)
    src, tgt, tn = src_pc.points_list[0], tgt_pc.points_list[0], tgt_pc.normals_list[0]
    W = np.random.default_rng(2).standard_normal((4, 4)).astype(np.float32)
    leaf = [t.clone().requires_grad_(True) for t in (src, tgt, tn)]
    t0 = time.perf_counter()
    Tg, _ = icputils.point_to_plane_gradICP(leaf[0][None], leaf[1][None], leaf[2][None], torch.eye(4), numiters=20)
    (Tg * T(W)).sum().backward()
    secs = time.perf_counter() - t0
    g = dict(W=W, T=Tg.detach().numpy(), seed=np.int64(SEED), stride=np.int64(STRIDE), seconds=np.float64(secs),
             n_src=np.int64(src.shape[0]), n_tgt=np.int64(tgt.shape[0]))
    for name, t, x in zip(("src", "tgt", "tn"), leaf, (src, tgt, tn)):
        gr = t.grad.numpy()
        g["in_sum_" + name] = x.double().sum(0).numpy()
        g["grad_" + name] = gr[::STRIDE].copy()
        g["grad_sum_" + name] = gr.astype(np.float64).sum(0)
        g["grad_norm_" + name] = np.float64(np.linalg.norm(gr.astype(np.float64)))
    np.savez_compressed(os.path.join(OUT, "c3_grad640.npz"), **g)
    print("src %d tgt %d  forward + backward of the reference: %.1f s" % (src.shape[0], tgt.shape[0], secs))
    print({k: (v.shape if hasattr(v, "shape") and v.ndim else v) for k, v in g.items()})


if __name__ == "__main__":
    main()
