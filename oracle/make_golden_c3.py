"""Config C3 at the benchmarked size: the REAL reference's autograd through point_to_plane_gradICP
(odometry/icputils.py:479-545, 20 iterations) between two 640x480 frames on the ds = 4 lattice (~18k x ~18k points):
gradients of <W, T> with respect to the source points, the target points and the target normals.

    python -m oracle.make_golden_c3

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


def main():
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
