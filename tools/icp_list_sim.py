"""CPU study of the candidate lists of the ICP half-iteration kernels (DESIGN.md sections 4 and 7).

Replays one gradICP solve of the benchmark workload with the ORACLE (bit-identical to the HIP path; its per-iteration
trace gives every query position of the 2 x numiters searches) and simulates the list scheme on those positions:
a list = the M nearest targets of the position q0 it was built at, R = distance of the (M+1)-th nearest (the scan that
builds a list covers more than that), a later search from q is exact on the list alone when
    sqrt(best list distance) + |q - q0| < 0.9999 R;
a list that gives no proof is rebuilt where the point is now.  Printed per launch: lists without a proof for M = 4 (what
2 lanes per point keep today), M = 8, and for a two-level list (4 slots checked first, 4 more and their radius fetched
only when the first four give no proof).

    python tools/icp_list_sim.py [seed] [frame] [gt|gradicp]      (the replay is cached under /tmp)
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
ODOM = sys.argv[3] if len(sys.argv) > 3 else "gradicp"
H, W, ds = 480, 640, 4
cache = "/tmp/icp_list_sim_s%d_f%d_%s.npz" % (seed, frame, ODOM)

if os.path.exists(cache):
    z = np.load(cache)
    src, tgt, trace, n_map = z["src"], z["tgt"], z["trace"], int(z["n_map"])
else:
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
    n_map = len(m)
    np.savez(cache, src=src, tgt=tgt, trace=trace, n_map=n_map)
print("seed %d frame %d (%s): map %d surfels, %d source points, %d targets" % (seed, frame, ODOM, n_map, len(src), len(tgt)))

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
tree = cKDTree(tgt.astype(np.float64))
dd, _ = tree.query(pos[0], k=9)
print("distance of the 1st / 5th / 9th nearest target (mm, median over the source points): %.2f / %.2f / %.2f" %
      tuple(1e3 * np.median(dd[:, j]) for j in (0, 4, 8)))


def simulate(levels):
    """levels = (4,), (8,) or (4, 8): slots checked first, then (optionally) the longer list"""
    Mmax = max(levels)
    out = []
    q0 = nn = None
    for h in range(1, 40):
        q = pos[h]
        if h == 1:   # the building launch
            nn_d, nn_i = tree.query(q, k=Mmax + 1)
            q0 = q.copy()
            continue
        delta = np.linalg.norm(q - q0, axis=1)
        proved = np.zeros(len(q), bool)
        first = None
        for M in levels:
            dl = np.linalg.norm(tgt[nn_i[:, :M]].astype(np.float64) - q[:, None, :], axis=2).min(1)
            ok = dl + delta < 0.9999 * nn_d[:, M]
            proved |= ok
            if first is None:
                first = ok.copy()
        fail = ~proved
        out.append((h, int((~first).sum()), int(fail.sum())))
        if fail.any():   # rebuilt where the point is now
            d2, i2 = tree.query(q[fail], k=Mmax + 1)
            nn_d[fail], nn_i[fail] = d2, i2
            q0[fail] = q[fail]
    return out


r4, r8, r48 = simulate((4,)), simulate((8,)), simulate((4, 8))
print("launch: lists without a proof, M = 4 | M = 8 | two levels 4 + 4: first level fails -> second level fails too")
for (h, _, f4), (_, _, f8), (_, a48, f48) in zip(r4, r8, r48):
    print("  %2d (%s): %6d | %6d | %6d -> %6d" % (h, "first half" if h % 2 == 0 else "look-ahead", f4, f8, a48, f48))
for name, r, col in (("M = 4", r4, 2), ("M = 8", r8, 2), ("two levels, re-scans", r48, 2), ("two levels, second fetches", r48, 1)):
    print("%s: total over the solve %d, launches with more than 50: %d" % (name, sum(x[col] for x in r), sum(x[col] > 50 for x in r)))
