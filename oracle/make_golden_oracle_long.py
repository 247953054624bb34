"""Trajectory of the ORACLE (oracle/slam.py: the C restatement, oracle/gs_oracle.c) over a long horizon, recorded next to the
reference's golden of the same sequence: the GPU test asserts that the HIP path reproduces it BIT FOR BIT over the whole
horizon (tests/test_hip_batch.py::test_pointfusion_long_horizon_vs_reference_golden), which is the statement that stays
well-posed where the reference no longer reproduces itself (tests/golden/reference_sensitivity_*.json).

    python -m oracle.make_golden_oracle_long --tag pf640_l60                      # reads tests/golden/pf640_l60.npz for the inputs
    python -m oracle.make_golden_oracle_long --tag facets640_l60

Build-container only (minutes of CPU).  Output: tests/golden/<tag>_oracle.npz (poses (L,4,4) f32, counts (L,), point sums).
TEST INFRASTRUCTURE ONLY."""
import argparse
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, REPO)
from gradslam_amd.datasets.synthetic import make_sequence  # noqa: E402
from oracle import slam as oslam  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", required=True)
    a = ap.parse_args()
    g = np.load(os.path.join(REPO, "tests", "golden", a.tag + ".npz"))
    L, H, W, seed = int(g["poses"].shape[0]), int(g["H"]), int(g["W"]), int(g["seed"])
    scene = str(g["scene"]) if "scene" in g.files else "wave"
    hole = float(g["hole_frac"]) if "hole_frac" in g.files else 0.05
    s = make_sequence(L, H, W, seed=seed, scene=scene, hole_frac=hole)
    assert abs(float(s["depths"].astype(np.float64).sum()) - float(g["depth_sum"])) < 1e-6 * float(g["depth_sum"])
    poses = s["poses"].copy()
    poses[1:] = poses[:1]
    counts, sums = [], []
    t0 = time.time()

    def rec(f, m, p):
        counts.append(len(m))
        sums.append(m.points.astype(np.float64).sum(0))
        print("frame %2d  %d surfels  %.0f s" % (f, len(m), time.time() - t0), flush=True)

    _, rp = oslam.run_sequence(s["colors"], s["depths"], s["intrinsics"][0], poses, per_frame=rec)
    np.savez_compressed(os.path.join(REPO, "tests", "golden", a.tag + "_oracle.npz"), poses=rp.astype(np.float32),
                        counts=np.asarray(counts, np.int64), sum_points=np.asarray(sums), seed=np.int64(seed), H=np.int64(H), W=np.int64(W),
                        scene=np.array(scene), hole_frac=np.float64(hole))
    d = np.linalg.norm(rp[:, :3, 3] - g["poses"][:, :3, 3], axis=1)
    print("max translation difference to the reference golden: %.3e m at frame %d" % (d.max(), d.argmax()))


if __name__ == "__main__":
    main()
