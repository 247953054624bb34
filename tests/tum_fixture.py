"""A tiny TUM-format dataset written to a temp directory (two sequences, un-synchronised colour / depth /
pose stamps, a comment header, a dropped pose line).  Shared by oracle/make_golden_extra.py (which runs
the reference's TUM loader on it) and the dataset tests (which run ours on the same bytes)."""
import os

import numpy as np

H, W = 24, 32


def write(root, n_frames=9, seed=3):
    from PIL import Image
    rng = np.random.default_rng(seed)
    for si, name in enumerate(("rgbd_dataset_freiburg1_alpha", "rgbd_dataset_freiburg2_beta")):
        d = os.path.join(root, name)
        os.makedirs(os.path.join(d, "rgb"), exist_ok=True)
        os.makedirs(os.path.join(d, "depth"), exist_ok=True)
        t0 = 1305031100.0 + 50 * si
        rgb_lines, depth_lines, gt_lines = ["# color images", "# timestamp filename"], ["# depth maps"], ["# ground truth"]
        for k in range(n_frames):
            tr = t0 + 0.0333 * k + 0.002 * rng.random()
            td = tr + 0.004 + 0.003 * rng.random()
            rgb = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
            dep = rng.integers(0, 40000, (H, W), dtype=np.uint16)
            dep[rng.random((H, W)) < 0.1] = 0
            fr, fd = "rgb/%.6f.png" % tr, "depth/%.6f.png" % td
            Image.fromarray(rgb).save(os.path.join(d, fr))
            Image.fromarray(dep).save(os.path.join(d, fd))
            rgb_lines.append("%.6f %s" % (tr, fr))
            depth_lines.append("%.6f %s" % (td, fd))
        for k in range(4 * n_frames):   # poses at 4x the frame rate
            tp = t0 - 0.01 + 0.0333 / 4 * k
            ang = 0.02 * k + 0.3 * si
            q = np.array([0.1 * np.sin(ang), np.sin(ang / 2) * 0.7, 0.05, np.cos(ang / 2)])
            q /= np.linalg.norm(q)
            if k == 5:
                q[:] = 0          # the loaders drop all-zero quaternions
            gt_lines.append("%.4f %.4f %.4f %.4f %.4f %.4f %.4f %.4f" % (tp, 0.01 * k, -0.02 * k, 0.5 + 0.001 * k, *q))
        for fname, lines in (("rgb.txt", rgb_lines), ("depth.txt", depth_lines), ("groundtruth.txt", gt_lines)):
            with open(os.path.join(d, fname), "w") as f:
                f.write("\n".join(lines) + "\n")
        open(os.path.join(d, "accelerometer.txt"), "w").write("# unused\n")
    return root


CASES = {
    "default": dict(seqlen=3, height=H, width=W),
    "strided": dict(seqlen=2, dilation=1, stride=2, start=1, end=8, height=H, width=W, normalize_color=True,
                    channels_first=True, sequences=("rgbd_dataset_freiburg2_beta",)),
}


# ------------------------------------------------------------------ ICL-NUIM format
def write_icl(root, n_frames=8, seed=11):
    from PIL import Image
    rng = np.random.default_rng(seed)
    for num in (0, 2):
        name = "living_room_traj%d_frei_png" % num
        d = os.path.join(root, name)
        os.makedirs(os.path.join(d, "rgb"), exist_ok=True)
        os.makedirs(os.path.join(d, "depth"), exist_ok=True)
        assoc, sim = [], []
        for k in range(n_frames):
            Image.fromarray(rng.integers(0, 256, (H, W, 3), dtype=np.uint8)).save(os.path.join(d, "rgb/%d.png" % k))
            dep = rng.integers(0, 30000, (H, W), dtype=np.uint16)
            dep[rng.random((H, W)) < 0.05] = 0
            Image.fromarray(dep).save(os.path.join(d, "depth/%d.png" % k))
            assoc.append("%d depth/%d.png %d rgb/%d.png" % (k, k, k, k))
            a = 0.03 * k + 0.2 * num
            R = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
            t = np.array([0.02 * k, -0.01 * k, 0.3 + 0.005 * k * k])
            for r in range(3):
                sim.append("%.6f %.6f %.6f %.6f" % (R[r, 0], R[r, 1], R[r, 2], t[r]))
            sim.append("")
        with open(os.path.join(d, "associations.txt"), "w") as f:
            f.write("\n".join(assoc) + "\n")
        with open(os.path.join(d, "livingRoom%dn.gt.sim" % num), "w") as f:
            f.write("\n".join(sim) + "\n")
    return root


ICL_CASES = {
    "default": dict(seqlen=3, height=H, width=W),
    "strided": dict(seqlen=2, dilation=2, stride=1, start=1, end=7, height=H, width=W, normalize_color=True,
                    channels_first=True, trajectories=("living_room_traj2_frei_png",)),
}


# ------------------------------------------------------------------ ScanNet format
def write_scannet(root, n_frames=6, seed=17):
    """two scenes, one sequence each: `scans/<scene>/{color,depth,pose,label-filt,intrinsic}` + the per-sequence
    metadata files in `seqmeta/` (the layout datasets/scannet.py:130-168 parses)."""
    from PIL import Image
    rng = np.random.default_rng(seed)
    base, meta = os.path.join(root, "scans"), os.path.join(root, "seqmeta")
    os.makedirs(meta, exist_ok=True)
    for si, scene in enumerate(("scene0000_00", "scene0001_00")):
        d = os.path.join(base, scene)
        for sub in ("color", "depth", "pose", "label-filt", "intrinsic"):
            os.makedirs(os.path.join(d, sub), exist_ok=True)
        K = np.array([[577.6 + si, 0, 318.9, 0], [0, 578.7, 242.7 - si, 0], [0, 0, 1, 0], [0, 0, 0, 1]])
        np.savetxt(os.path.join(d, "intrinsic", "intrinsic_depth.txt"), K)
        np.savetxt(os.path.join(d, "intrinsic", "intrinsic_color.txt"), K * 2)
        lines = []
        for k in range(n_frames):
            Image.fromarray(rng.integers(0, 256, (H, W, 3), dtype=np.uint8)).save(os.path.join(d, "color/%d.png" % k))
            dep = rng.integers(0, 9000, (H, W), dtype=np.uint16)
            dep[rng.random((H, W)) < 0.07] = 0
            Image.fromarray(dep).save(os.path.join(d, "depth/%d.png" % k))
            Image.fromarray(rng.integers(0, 41, (H, W), dtype=np.uint8)).save(os.path.join(d, "label-filt/%d.png" % k))
            a = 0.04 * k + 0.1 * si
            T = np.eye(4)
            T[:3, :3] = [[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]]
            T[:3, 3] = [0.03 * k, 0.5 - 0.01 * k, 1.0 + 0.002 * k * k]
            np.savetxt(os.path.join(d, "pose/%d.txt" % k), T)
            rel = lambda sub, f: "%s/%s/%s" % (scene, sub, f)  # noqa: E731
            lines.append(" ".join(["color", rel("color", "%d.png" % k), "depth", rel("depth", "%d.png" % k), "pose",
                                   rel("pose", "%d.txt" % k), "label-filt", rel("label-filt", "%d.png" % k), "label",
                                   "-", "instance-filt", "-", "instance", "-", "intrinsic_depth",
                                   rel("intrinsic", "intrinsic_depth.txt"), "intrinsic_color",
                                   rel("intrinsic", "intrinsic_color.txt")]))
        with open(os.path.join(meta, "%s-seq_0.txt" % scene), "w") as f:
            f.write("\n".join(lines) + "\n")
    return base, meta


SCANNET_CASES = {
    "default": dict(scenes=None, height=H, width=W),
    "nyu40_cf": dict(scenes=("scene0001_00",), start=1, end=5, height=H, width=W, seg_classes="nyu40",
                     channels_first=True, normalize_color=True),
}
