"""Seeded synthetic RGB-D sequences: the workload of bench.py and the parity
tests.  Pure numpy so the same bytes can be fed to the HIP path, the oracle and (in the
build container) the reference.  Intrinsics follow the TUM convention the reference's loader
uses (datasets/tum.py:338-346: fx=fy=525, scaled with the image width)."""
import numpy as np

__all__ = ["make_sequence", "tum_intrinsics", "gt_pose", "path_parameter", "SCENES"]


def tum_intrinsics(H, W):
    K = np.eye(4, dtype=np.float32)
    K[0, 0] = K[1, 1] = 525.0 * W / 640.0
    K[0, 2] = (W - 1) / 2.0
    K[1, 2] = (H - 1) / 2.0
    return K


PATH_TURN = 150   # frames after which the camera path turns back (and forth: period 2 * PATH_TURN)


def path_parameter(s):
    """The camera sweeps out for PATH_TURN frames, then retraces its path, and so on: long sequences stay within
    0.3 rad / 0.75 m of the start (the ray casting below assumes a camera that looks roughly along z) and revisit
    mapped surface, so the map saturates instead of growing for ever.  Identity for the first PATH_TURN frames."""
    p = s % (2 * PATH_TURN)
    return p if p <= PATH_TURN else 2 * PATH_TURN - p


def gt_pose(s, yaw_per_frame=0.002, tx_per_frame=0.005):
    """Camera-to-world pose of frame s: small yaw about y plus x translation."""
    s = path_parameter(s)
    a = yaw_per_frame * s
    T = np.eye(4, dtype=np.float64)
    T[0, 0] = np.cos(a); T[0, 2] = np.sin(a)
    T[2, 0] = -np.sin(a); T[2, 2] = np.cos(a)
    T[0, 3] = tx_per_frame * s
    return T.astype(np.float32)


def _scene_depth_at(x_w, y_w):
    """Height field z = f(x, y) of a static smooth scene in WORLD coordinates ("wave": the benchmark workload)."""
    return 2.0 + 0.3 * np.sin(3.0 * x_w + 0.4) * np.cos(2.5 * y_w) + 0.2 * x_w


# "facets": three mutually inclined planes meeting in an apex behind the image centre (a concave corner seen from inside)
# plus a ridge of triangular profile across two of them.  Point-to-plane residuals on planes do not depend on where
# along its plane a point is matched, three independent normals pin the translation and the spread of the points the
# rotation, so the reference's 20 Gauss-Newton iterations CONVERGE on it (on "wave" they are cut off while the solve
# still slides along the smooth surface by millimetres): the scene on which the long horizon is compared with the
# reference at BASELINE's 1e-4 m (tests/golden/facets640_l60.npz).
_FACET_SLOPE = 0.45
_FACET_DIRS = tuple((np.cos(a), np.sin(a)) for a in (np.pi / 2 + 0.3, np.pi / 2 + 0.3 + 2 * np.pi / 3,
                                                     np.pi / 2 + 0.3 + 4 * np.pi / 3))


def _facets_depth_at(x_w, y_w):
    x = x_w - 0.15
    y = y_w + 0.05
    p = None
    for cx, cy in _FACET_DIRS:
        q = _FACET_SLOPE * (cx * x + cy * y)
        p = q if p is None else np.maximum(p, q)
    u = 0.8 * x_w - 0.6 * y_w - 0.35   # signed distance from the ridge line
    ridge = 0.07 * np.maximum(0.0, 1.0 - np.abs(u) / 0.12)
    return 2.45 - p - ridge


SCENES = {"wave": _scene_depth_at, "facets": _facets_depth_at}


def make_sequence(L, H, W, seed=0, hole_frac=0.05, yaw_per_frame=0.002, tx_per_frame=0.005, first=0, scene="wave"):
    """Returns dict(colors (L,H,W,3) f32 in [0,255), depths (L,H,W,1) f32 metres with
    `hole_frac` pixels zeroed, intrinsics (1,4,4), poses (L,4,4) ground truth).

    Depth is the ray-cast (fixed-point iteration) of a static height-field scene seen from the
    moving camera, so consecutive frames are geometrically consistent and ICP converges to the
    ground-truth motion.

    first > 0: frames first .. first + L - 1 of the same camera path with their own random stream (holes, colours),
    so that a long sequence can be generated in parallel chunks (bench.py --workload c5); first = 0 is the sequence the
    parity tests and goldens use.
    scene: "wave" (the benchmark's smooth height field) or "facets" (inclined planes + a ridge, see above)."""
    scene_fn = SCENES[scene]
    rng = np.random.default_rng(seed if first == 0 else [seed, first])
    K = tum_intrinsics(H, W)
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    u, v = np.meshgrid(np.arange(W, dtype=np.float64), np.arange(H, dtype=np.float64))
    rx = (u - cx) / fx
    ry = (v - cy) / fy
    depths = np.empty((L, H, W, 1), np.float32)
    poses = np.empty((L, 4, 4), np.float32)
    phase = 0.1 * seed
    for s in range(L):
        T = gt_pose(first + s, yaw_per_frame, tx_per_frame).astype(np.float64)
        poses[s] = T.astype(np.float32)
        R, t = T[:3, :3], T[:3, 3]
        d = np.full((H, W), 2.0)
        for _ in range(30):  # fixed point: world z of the ray point must equal the height field
            pc = np.stack([rx * d, ry * d, d], -1)
            pw = pc @ R.T + t
            zw = scene_fn(pw[..., 0] + phase, pw[..., 1])
            # move along the ray so that world-z matches (R is close to identity)
            d = d + (zw - pw[..., 2]) / R[2, 2]
        holes = rng.random((H, W)) < hole_frac
        d = np.where(holes, 0.0, d)
        depths[s, ..., 0] = d.astype(np.float32)
    colors = (rng.random((L, H, W, 3)) * 255.0).astype(np.float32)
    return {"colors": colors, "depths": depths, "intrinsics": K[None], "poses": poses}
