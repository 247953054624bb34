"""CPU study behind the candidate lists of the ICP half-iteration kernels.

Replays one gradICP solve of the benchmark workload with the ORACLE (its per-iteration trace gives every query position of
the 2 x numiters searches), then simulates the list scheme on those positions: a list = every target within R of the
position q0 it was built at (at most M slots), a later search from q is exact on the list alone when
sqrt(best list distance) + |q - q0| < 0.9999 R.  Prints, per launch, how many queries fail that proof and how many
blocks of 384 queries contain a failing one (a launch is as slow as its slowest block).

    python tools/icp_list_sim.py [seed] [frame] [margin_cells] [M] [gt|gradicp] [cube_lists 0|1]
"""
import os
import sys

import numpy as np
from scipy.spatial import cKDTree

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as o, slam as osl   # noqa: E402
from gradslam_amd.datasets.synthetic import make_sequence   # noqa: E402

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
frame = int(sys.argv[2]) if len(sys.argv) > 2 else 5
margin_cells = float(sys.argv[3]) if len(sys.argv) > 3 else 0.25
M = int(sys.argv[4]) if len(sys.argv) > 4 else 4
ODOM = sys.argv[5] if len(sys.argv) > 5 else "gt"
CUBE_LISTS = int(sys.argv[6]) if len(sys.argv) > 6 else 0
H, W, ds = 480, 640, 4

seq = make_sequence(frame + 1, H, W, seed=seed)
K = seq["intrinsics"][0]
m, poses = osl.run_sequence(seq["colors"][:frame], seq["depths"][:frame], K, seq["poses"][:frame], odom=ODOM)
depth = seq["depths"][frame].reshape(H, W)
prev_pose = poses[frame - 1]
v, n, a, _ = o.frame_maps(depth, K, 0.6)
gv, gn = o.global_maps(v, n, depth, prev_pose)
src, _, _ = o.downsample_frame(gv, gn, seq["colors"][frame], depth, ds)
pix = o.project_map(m.points, prev_pose, K, H, W)
tgt, tgtn, _ = o.select_targets(pix, W, ds, m.points, m.normals)
T, idx, trace = o.icp(src, tgt, tgtn, init=None, compose=prev_pose, mode=1, numiters=20, return_trace=True)
print("map %d surfels, src %d, tgt %d" % (len(m), len(src), len(tgt)))

# query positions of the 40 searches
pos = []
A = src.astype(np.float64)
for k in range(20):
    xi, sig = trace[k, 4:10], trace[k, 3]
    Tr = o.se3_exp(xi).astype(np.float64)
    Ts = o.se3_exp((np.float32(sig) * xi).astype(np.float32)).astype(np.float64)
    pos.append(A)                                    # first half of iteration k
    pos.append(A @ Tr[:3, :3].T + Tr[:3, 3])          # look-ahead
    A = A @ Ts[:3, :3].T + Ts[:3, 3]
print("|xi_t| per iteration (mm):", " ".join("%.3f" % (1e3 * np.linalg.norm(trace[k, 4:7])) for k in range(20)))
print("mean displacement between consecutive searches (mm):",
      " ".join("%.3f" % (1e3 * np.linalg.norm(pos[h + 1] - pos[h], axis=1).mean()) for h in range(39)))

# the grid of gs_knn.hip:grid_from_bbox
lo, hi = tgt.min(0).astype(np.float64), tgt.max(0).astype(np.float64)
e = (hi - lo) + 1e-6
c = 1.5 * np.sqrt((e[0] * e[1] + e[1] * e[2] + e[0] * e[2]) / len(tgt))
nxyz = (e / c).astype(int) + 1
print("cell edge %.2f mm, grid %s" % (1e3 * c, nxyz))
tree = cKDTree(tgt.astype(np.float64))


def face_bound(q):
    """(amin - 0.001) * c of grid_search_stage0: distance to the nearest face of the 2x2x2 block that has cells behind"""
    p = np.clip(q, lo, hi)
    f = (p - lo) / c
    cell = np.minimum(np.maximum(f.astype(int), 0), nxyz - 1)
    fr = f - cell
    x0 = np.where(fr < 0.5, cell - 1, cell)
    big = 3e38
    lo_d = np.where(x0 >= 1, fr + (cell - x0), big)
    hi_d = np.where(x0 + 2 < nxyz, (x0 + 2 - cell) - fr, big)
    amin = np.minimum(lo_d, hi_d).min(1)
    return (amin - 0.001) * c


def build(q):
    """gl_build_block: the widest of the nested radii d1 + margin / 2^k (capped by the block-face bound) whose targets
    fit the M slots"""
    dd, ii = tree.query(q, k=M + 1)
    d1 = dd[:, 0]
    rb = face_bound(q)
    open_ = d1 > rb                     # stage 0 cannot prove it: cube scans, no list
    R = np.zeros(len(q))
    for k in (3, 2, 1, 0):              # the widest that fits wins
        Rk = np.minimum(d1 + margin_cells * c / (1 << k), rb)
        fits = dd[:, M] >= Rk           # at most M targets within Rk
        R = np.where(fits, Rk, R)
    lists = np.where(dd[:, :M] < R[:, None], ii[:, :M], -1)
    R = np.where(open_, 0.0, R)
    return lists, R


NQ = 365   # valid queries of a 384-slot block (95 % of the lattice has depth)
nblk = (len(src) + NQ - 1) // NQ
q0 = None
for h in range(40):
    q = pos[h]
    if h == 0:
        print("launch  0: plain search (no lists yet)")
        continue
    if h == 1:
        lists, R = build(q)
        q0 = q.copy()
        nl = (lists >= 0).sum(1)
        print("launch  1: lists built for all; open %d (%.2f %%), entries per list mean %.2f max %d, R mean %.2f mm" % (
            (R == 0).sum(), 100.0 * (R == 0).mean(), nl[R > 0].mean(), nl.max(), 1e3 * R[R > 0].mean()))
        continue
    tl = np.where(lists >= 0, lists, 0)
    dl = np.linalg.norm(tgt[tl].astype(np.float64) - q[:, None, :], axis=2)
    dl = np.where(lists >= 0, dl, np.inf)
    bd = dl.min(1)
    delta = np.linalg.norm(q - q0, axis=1)
    ok = (R > 0) & (bd + delta < 0.9999 * R)
    # check exactness of the claim
    d1, i1 = tree.query(q, k=1)
    best = np.take_along_axis(tl, dl.argmin(1)[:, None], 1)[:, 0]
    wrong = ok & (np.abs(bd - d1) > 1e-12)
    fail = ~ok
    was_open = R == 0
    nb_fail = len(np.unique(np.nonzero(fail)[0] // NQ))
    nb_new = len(np.unique(np.nonzero(fail & ~was_open)[0] // NQ))
    print("launch %2d: fail %5d (%.3f %%) of which open before %5d; blocks with a failing query %3d / %d (with a NEW failure %3d); "
          "wrong %d" % (h, fail.sum(), 100.0 * fail.mean(), (fail & was_open).sum(), nb_fail, nblk, nb_new, wrong.sum()))
    if fail.any():   # rebuild the failing ones where they are now
        l2, R2 = build(q[fail])
        lists[fail] = l2
        R[fail] = R2
        q0[fail] = q[fail]
