"""GPU parity tests proper: every C-ABI entry point (through gradslam_amd.ops) against the CPU
oracle on the same inputs, plus the golden vectors recorded from the real reference.

Bars: index tables / masks / counts BIT-EXACT; frame maps, alpha and fused
surfels bit-exact against the oracle (same operation sequence by construction); ICP transforms
within 1e-6 of the oracle and 2e-5 of the reference (float64-accumulated normal equations vs
the reference's float32 sgemm)."""
import math

import numpy as np
import pytest
import torch

from gradslam_amd.datasets.synthetic import make_sequence
from oracle import oracle as o

pytestmark = pytest.mark.gpu

DIST_TH, DOT_TH, SIGMA = 0.05, math.cos(20 * math.pi / 180), 0.6


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from gradslam_amd import ops as _ops
    return _ops


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    return t.detach().cpu().numpy()


def same_bits(a, b):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    if a.dtype == np.float32:
        # +0 / -0 compare equal on purpose; NaNs are not expected
        assert np.array_equal(a, b), "mismatches: %d of %d (max abs %g)" % (
            (a != b).sum(), a.size, np.abs(a.astype(np.float64) - b.astype(np.float64)).max())
    else:
        assert np.array_equal(a, b), "mismatches: %d of %d" % ((a != b).sum(), a.size)


# ------------------------------------------------------------------------------------ K1
@pytest.mark.parametrize("case", ["msrd", "synth640", "tiny", "neg_fy_ragged"])
def test_frame_and_global_maps(ops, golden, case):
    if case == "msrd":
        g = golden("msrd_b0")
        depth, K, pose = g["depths"][1, ..., 0], g["intrinsics"], g["poses"][1]
    elif case == "synth640":
        s = make_sequence(2, 480, 640, seed=5)
        depth, K, pose = s["depths"][1, ..., 0], s["intrinsics"][0], s["poses"][1]
    elif case == "tiny":
        depth = np.array([[1.0, 0.0], [2.0, 3.0]], np.float32)
        K, pose = np.eye(4, dtype=np.float32), np.eye(4, dtype=np.float32)
        K[0, 0] = K[1, 1] = 2.0
    else:  # odd sizes that do not divide the 64x8 tile, negative fy, invalid borders
        rng = np.random.default_rng(3)
        depth = (rng.random((67, 131)) * 3).astype(np.float32)
        depth[rng.random(depth.shape) < 0.2] = 0
        depth[:, -1] = 0
        depth[-1, :5] = -1.0
        K = np.array([[120.3, 0, 65.2, 0], [0, -120.0, 33.1, 0], [0, 0, 1, 0], [0, 0, 0, 1]], np.float32)
        pose = o.se3_exp(np.array([0.1, -0.2, 0.05, 0.02, -0.01, 0.03], np.float32))
    v, n, a, valid = ops.frame_maps(dev(depth), dev(K), SIGMA)
    gv, gn = ops.global_maps(v, n, dev(depth), dev(pose))
    ov, on, oa, ovalid = o.frame_maps(depth, K, SIGMA)
    ogv, ogn = o.global_maps(ov, on, depth, pose)
    same_bits(host(v), ov)
    same_bits(host(n), on)
    same_bits(host(a), oa)
    assert np.array_equal(host(valid), ovalid)
    same_bits(host(gv), ogv)
    same_bits(host(gn), ogn)
    if case == "msrd":  # and against the reference's own golden maps
        same_bits(host(v), g["vertex_map"][1])
        same_bits(host(n), g["normal_map"][1])
        same_bits(host(gv), g["global_vertex_map"][1])
        same_bits(host(gn), g["global_normal_map"][1])
        np.testing.assert_allclose(host(a), g["alpha"][1], rtol=2e-7)


def test_global_maps_without_pose_is_a_copy(ops):
    s = make_sequence(1, 32, 48, seed=1)
    depth = dev(s["depths"][0, ..., 0])
    v, n, _, _ = ops.frame_maps(depth, dev(s["intrinsics"][0]))
    gv, gn = ops.global_maps(v, n, depth, None)
    assert torch.equal(gv, v) and torch.equal(gn, n)


def test_alpha_of_points(ops):
    rng = np.random.default_rng(0)
    pts = (rng.standard_normal((1000, 3)) * 2).astype(np.float32)
    pts[0] = 0
    pts[1] = 100.0
    same_bits(host(ops.alpha_of_points(dev(pts), SIGMA)), o.alpha(pts, SIGMA))


# ------------------------------------------------------------------------------------ K2/K5/K6
def _fusion_case(name, golden):
    if name == "msrd":
        g = golden("msrd_b0")
        return dict(P=g["map0_points"], N=g["map0_normals"], C=g["map0_colors"], F=g["map0_ccounts"],
                    depth=g["depths"][1, ..., 0], rgb=g["colors"][1], K=g["intrinsics"], pose=g["poses"][1], g=g)
    H, W = (480, 640) if name == "synth640" else (96, 128)
    s = make_sequence(2, H, W, seed=11)
    K = s["intrinsics"][0]
    d0 = s["depths"][0, ..., 0]
    v, n, a, _ = o.frame_maps(d0, K, SIGMA)
    gv, gn = o.global_maps(v, n, d0, s["poses"][0])
    e3, e1 = np.zeros((0, 3), np.float32), np.zeros((0, 1), np.float32)
    P, N, C, F = o.fuse_append(e3, e3, e3, e1, np.full(H * W, -1, np.int32), gv, gn, s["colors"][0], a, d0)
    return dict(P=P, N=N, C=C, F=F, depth=s["depths"][1, ..., 0], rgb=s["colors"][1], K=K, pose=s["poses"][1], g=None)


@pytest.mark.parametrize("name", ["msrd", "synth128", "synth640"])
def test_association_and_fusion_bit_exact(ops, golden, name):
    c = _fusion_case(name, golden)
    H, W = c["depth"].shape
    P, N, C, F = (dev(c[k]) for k in "PNCF")
    depth, rgb, K, pose = dev(c["depth"]), dev(c["rgb"]), dev(c["K"]), dev(c["pose"])
    v, n, a, _ = ops.frame_maps(depth, K, SIGMA)
    gv, gn = ops.global_maps(v, n, depth, pose)
    ov, on, oa, _ = o.frame_maps(c["depth"], c["K"], SIGMA)
    ogv, ogn = o.global_maps(ov, on, c["depth"], c["pose"])

    pix = ops.project_map(P, pose, K, H, W)
    opix = o.project_map(c["P"], c["pose"], c["K"], H, W)
    same_bits(host(pix), opix)
    act = ops.active_table(pix, W)
    oact = o.active_table(opix, W)
    same_bits(host(act), oact)
    mask = ops.similar_rows(act, P, N, gv, gn, DIST_TH, DOT_TH)
    omask = o.similar_rows(oact, c["P"], c["N"], ogv, ogn, DIST_TH, DOT_TH)
    same_bits(host(mask), omask)
    uq, best_from_rows = ops.best_unique_rows(act[mask], P, F, gv)
    ouq = o.best_unique_rows(oact[omask], c["P"], c["F"], ogv)
    same_bits(host(uq), ouq)
    best, sim = ops.associate(pix, P, N, F, gv, gn, DIST_TH, DOT_TH, want_similar=True)
    obest, osim = o.associate(opix, c["P"], c["N"], c["F"], ogv, ogn, DIST_TH, DOT_TH)
    same_bits(host(best), obest)
    same_bits(host(best_from_rows), obest)
    same_bits(host(sim), osim)
    same_bits(host(ops.best_table(best, H, W)), ouq)
    same_bits(host(ops.rows_to_best_pix(uq, H, W)), obest)

    # K2 down-samplers
    tp, tn, _ = ops.select_targets(pix, W, 4, P, N)
    otp, otn, _ = o.select_targets(opix, W, 4, c["P"], c["N"])
    same_bits(host(tp), otp)
    same_bits(host(tn), otn)
    tp2, tn2, _ = ops.downsample_table(act, 4, P, N)
    same_bits(host(tp2), otp)
    fp, fn, fc = ops.downsample_frame(gv, gn, rgb, depth, 4)
    ofp, ofn, ofc = o.downsample_frame(ogv, ogn, c["rgb"], c["depth"], 4)
    same_bits(host(fp), ofp)
    same_bits(host(fn), ofn)
    same_bits(host(fc), ofc)

    # K6 fuse + append, parity mode and fast mode
    for renorm in (True, False):
        n0 = c["P"].shape[0]
        cap = n0 + H * W
        bufs = [torch.zeros((cap, k), dtype=torch.float32, device="cuda") for k in (3, 3, 3, 1)]
        for b_, src in zip(bufs, (P, N, C, F)):
            b_[:n0] = src.reshape(n0, -1)
        cnt = ops.fuse_append_(bufs[0], bufs[1], bufs[2], bufs[3], n0, best, gv, gn, rgb, a, depth, renorm)
        oP, oN, oC, oF = o.fuse_append(c["P"], c["N"], c["C"], c["F"], obest, ogv, ogn, c["rgb"], oa, c["depth"], renorm)
        assert cnt == oP.shape[0]
        for b_, ref in zip(bufs, (oP, oN, oC, oF)):
            same_bits(host(b_[:cnt]), ref)
    if c["g"] is not None:  # tables against the reference's own output
        g = c["g"]
        same_bits(host(act), g["active"])
        same_bits(host(mask), g["similar_mask"])
        same_bits(host(uq), g["unique"])


def test_empty_map_and_all_invalid_frame(ops):
    H, W = 16, 24
    depth = np.zeros((H, W), np.float32)
    K = np.eye(4, dtype=np.float32)
    K[0, 0] = K[1, 1] = 20
    d = dev(depth)
    v, n, a, valid = ops.frame_maps(d, dev(K))
    assert not host(valid).any() and not host(v).any()
    gv, gn = ops.global_maps(v, n, d, dev(np.eye(4, dtype=np.float32)))
    e3 = torch.zeros((0, 3), device="cuda")
    pix = ops.project_map(e3, dev(np.eye(4, dtype=np.float32)), dev(K), H, W)
    assert pix.numel() == 0
    best = ops.associate(pix, e3, e3, torch.zeros((0, 1), device="cuda"), gv, gn, DIST_TH, DOT_TH)
    assert (host(best) == -1).all()
    bufs = [torch.zeros((H * W, k), device="cuda") for k in (3, 3, 3, 1)]
    rgb = torch.zeros((H, W, 3), device="cuda")
    assert ops.fuse_append_(*bufs, 0, best, gv, gn, rgb, a, d) == 0       # nothing valid to append
    depth[3, 5] = 1.5
    d = dev(depth)
    v, n, a, valid = ops.frame_maps(d, dev(K))
    gv, gn = ops.global_maps(v, n, d, dev(np.eye(4, dtype=np.float32)))
    assert ops.fuse_append_(*bufs, 0, best, gv, gn, rgb, a, d) == 1
    same_bits(host(bufs[0][:1]), host(gv)[3, 5][None])
    pts, _, _ = ops.downsample_frame(gv, gn, rgb, d, 4)
    assert pts.shape[0] == 0  # (3,5) is off the ds=4 lattice


def test_store_overflow_is_reported(ops):
    s = make_sequence(1, 32, 32, seed=2)
    d = dev(s["depths"][0, ..., 0])
    v, n, a, _ = ops.frame_maps(d, dev(s["intrinsics"][0]))
    gv, gn = ops.global_maps(v, n, d, dev(s["poses"][0]))
    bufs = [torch.zeros((10, k), device="cuda") for k in (3, 3, 3, 1)]
    best = torch.full((32 * 32,), -1, dtype=torch.int32, device="cuda")
    from gradslam_amd._C import HipExtensionError
    with pytest.raises(HipExtensionError, match="overflow"):
        ops.fuse_append_(*bufs, 0, best, gv, gn, dev(s["colors"][0]), a, d)
    assert torch.isfinite(bufs[0]).all()


# ------------------------------------------------------------------------------------ K3
@pytest.mark.parametrize("ns,nt", [(1, 1), (5, 3), (257, 513), (1024, 512), (5000, 7001), (19200, 23000)])
def test_knn_exact(ops, ns, nt):
    rng = np.random.default_rng(ns * 31 + nt)
    tgt = rng.standard_normal((nt, 3)).astype(np.float32)
    src = (tgt[rng.integers(0, nt, ns)] + 0.01 * rng.standard_normal((ns, 3))).astype(np.float32)
    if nt > 10:  # exact duplicates: ties must resolve to the lowest index
        tgt[nt // 2:nt // 2 + 5] = tgt[3]
        src[0] = tgt[3]
    idx, d2 = ops.knn1(dev(src), dev(tgt))
    oidx, od2 = o.knn1(src, tgt)
    same_bits(host(idx), oidx)
    same_bits(host(d2), od2)


def test_knn_ties_on_a_lattice(ops):
    """Integer lattice: many exactly equidistant targets; the lowest index must win."""
    g = np.stack(np.meshgrid(np.arange(8), np.arange(8), np.arange(8), indexing="ij"), -1).reshape(-1, 3)
    tgt = g.astype(np.float32)
    src = (g[::3] + 0.5).astype(np.float32)
    idx, _ = ops.knn1(dev(src), dev(tgt))
    oidx, _ = o.knn1(src, tgt)
    same_bits(host(idx), oidx)
    d = ((src[:, None] - tgt[None]) ** 2).sum(-1)
    assert np.array_equal(oidx, d.argmin(1))


# ------------------------------------------------------------------------------------ K4
def test_gauss_newton_rows_and_solve(ops, golden):
    g = golden("icp_unit")
    src, tgt, tn = g["src"], g["tgt"], g["tgt_normals"]
    for thr in (None, float(g["gn_thr"])):
        A, b, idx, keep = ops.gauss_newton_rows(dev(src), dev(tgt), dev(tn), thr)
        oA, ob, oidx, okeep = o.gauss_newton_rows(src, tgt, tn, thr)
        same_bits(host(idx), oidx)
        same_bits(host(keep), okeep)
        same_bits(host(A), oA)
        same_bits(host(b), ob)
        x = ops.solve_normal_eq(A, b, 1e-8, keep)
        ox = o.solve_normal_eq(oA, ob, 1e-8, okeep)
        np.testing.assert_allclose(host(x), ox, rtol=1e-6, atol=1e-9)
    same_bits(host(ops.gauss_newton_rows(dev(src), dev(tgt), dev(tn))[0]), g["gn_A"])
    np.testing.assert_allclose(host(ops.solve_normal_eq(dev(g["gn_A"]), dev(g["gn_b"]), 1e-8)), g["solve_x"][:, 0],
                               rtol=2e-3, atol=2e-6)
    # the reference's own KAT (4 unknowns): A x ~ b (tests/odometry/test_icputils.py:18-49)
    x = host(ops.solve_normal_eq(dev(g["kat_A"]), dev(g["kat_b"]), 1e-8))
    np.testing.assert_allclose(x, g["kat_x"][:, 0], rtol=1e-3, atol=1e-5)
    np.testing.assert_allclose(g["kat_A"] @ x, g["kat_b"][:, 0], rtol=1.3e-6 * 50, atol=1e-4)


def test_se3_exp_and_transform(ops, golden):
    g = golden("icp_unit")
    for xi, T in zip(g["se3_xi"], g["se3_T"]):
        t = host(ops.se3_exp(dev(xi)))
        np.testing.assert_allclose(t, o.se3_exp(xi), rtol=0, atol=1e-7)
        np.testing.assert_allclose(t, T, rtol=1e-5, atol=1e-6)
    T = o.se3_exp(g["se3_xi"][1])
    same_bits(host(ops.transform_points(dev(g["src"]), dev(T))), o.transform_points(g["src"], T))


@pytest.mark.parametrize("mode,key", [(0, "icp"), (1, "gradicp")])
@pytest.mark.parametrize("iters", [3, 20])
def test_icp_against_oracle_and_reference(ops, golden, mode, key, iters):
    g = golden("icp_unit")
    src, tgt, tn = g["src"], g["tgt"], g["tgt_normals"]
    T, idx, tr = ops.icp(dev(src), dev(tgt), dev(tn), mode=mode, numiters=iters, return_trace=True)
    oT, oidx, otr = o.icp(src, tgt, tn, mode=mode, numiters=iters, return_trace=True)
    np.testing.assert_allclose(host(T), oT, rtol=0, atol=1e-6)
    assert (host(idx) == oidx).mean() >= 0.999
    np.testing.assert_allclose(host(tr)[:, :2], otr[:, :2], rtol=1e-4, atol=1e-9)   # err, new_err
    np.testing.assert_allclose(host(tr)[:, 4:10], otr[:, 4:10], rtol=1e-3, atol=1e-7)  # xi
    np.testing.assert_allclose(host(T), g["%s%d_T" % (key, iters)], rtol=0, atol=2e-5)
    assert (host(idx) == g["%s%d_idx" % (key, iters)]).mean() > 0.995


def test_icp_compose_and_dist_thresh(ops, golden):
    g = golden("icp_unit")
    src, tgt, tn = g["src"], g["tgt"], g["tgt_normals"]
    comp = o.se3_exp(np.array([0.3, 0.1, -0.2, 0.05, 0.02, -0.04], np.float32))
    init = o.se3_exp(np.array([0.001, 0.0, 0.002, 0.0, 0.001, 0.0], np.float32))
    T = ops.icp(dev(src), dev(tgt), dev(tn), init=dev(init), compose=dev(comp), mode=1, numiters=5, dist_thresh=0.01,
                return_idx=False)
    oT, _ = o.icp(src, tgt, tn, init=init, compose=comp, mode=1, numiters=5, dist_thresh=0.01)
    np.testing.assert_allclose(host(T), oT, rtol=0, atol=1e-6)


def test_icp_full_size_properties(ops):
    """BASELINE size (640x480 / ds4: ~18k x ~18k points): the oracle would take minutes, so check
    size-independent properties: ICP recovers a known small rigid motion and the point-to-plane
    error collapses over the iterations."""
    s = make_sequence(2, 480, 640, seed=4)
    K = s["intrinsics"][0]
    pts = []
    for f in range(2):
        d = dev(s["depths"][f, ..., 0])
        v, n, _, _ = ops.frame_maps(d, dev(K))
        gv, gn = ops.global_maps(v, n, d, dev(s["poses"][0]))  # both under pose 0: frame 1 is misplaced
        p, nn, _ = ops.downsample_frame(gv, gn, None, d, 4)
        pts.append((p, nn))
    (tgt, tn), (src, _) = pts
    T, tr = ops.icp(src, tgt, tn, mode=1, numiters=20, return_idx=False, return_trace=True)
    true_T = np.linalg.inv(s["poses"][0].astype(np.float64)) @ s["poses"][1].astype(np.float64)
    assert np.abs(host(T) - true_T).max() < 2e-3
    tr = host(tr)
    assert tr[-1, 0] < 0.25 * tr[0, 0]  # point-to-plane error (sum of squared residuals) collapses
    # and the hard-LM variant agrees with the soft one on this well-posed problem
    T2 = ops.icp(src, tgt, tn, mode=0, numiters=20, return_idx=False)
    assert np.abs(host(T2) - true_T).max() < 2e-3


def test_icp_with_device_side_counts(ops, golden):
    """gs_icp_dc_f32: point counts read on the device from bound-sized buffers (no host read-back)
    must give exactly the transform of the host-count call on the same points."""
    s = make_sequence(2, 240, 320, seed=12)
    K = dev(s["intrinsics"][0])
    d0, d1 = dev(s["depths"][0, ..., 0]), dev(s["depths"][1, ..., 0])
    pose = dev(s["poses"][0])
    v0, n0, _, _ = ops.frame_maps(d0, K)
    gv0, gn0 = ops.global_maps(v0, n0, d0, pose)
    v1, n1, _, _ = ops.frame_maps(d1, K)
    gv1, gn1 = ops.global_maps(v1, n1, d1, pose)
    tgt, tn, _ = ops.downsample_frame(gv0, gn0, None, d0, 2)
    src, _, _ = ops.downsample_frame(gv1, gn1, None, d1, 2)
    assert tgt.shape[0] > 2048
    T_ref, idx_ref = ops.icp(src, tgt, tn, mode=1, numiters=20)
    # bound-sized buffers with garbage beyond the actual counts
    srcb, _, _, n_src = ops.downsample_frame(gv1, gn1, None, d1, 2, sync=False)
    tgtb, tnb, _, n_tgt = ops.downsample_frame(gv0, gn0, None, d0, 2, sync=False)
    assert srcb.shape[0] > src.shape[0] and int(n_src) == src.shape[0] and int(n_tgt) == tgt.shape[0]
    srcb[src.shape[0]:] = float("nan")
    tgtb[tgt.shape[0]:] = 1e9
    tnb[tgt.shape[0]:] = 7.0
    T, idx = ops.icp(srcb, tgtb, tnb, mode=1, numiters=20, n_src_dev=n_src, n_tgt_dev=n_tgt)
    assert torch.equal(T, T_ref) and torch.equal(idx[: src.shape[0]], idx_ref)
    T0 = ops.icp(srcb, tgtb, tnb, mode=0, numiters=5, n_src_dev=n_src, n_tgt_dev=n_tgt, return_idx=False)
    assert torch.equal(T0, ops.icp(src, tgt, tn, mode=0, numiters=5, return_idx=False))


def test_device_side_map_counts_match_host_counts(ops):
    """PointFusion with the surfel count kept on the device (no read-back per frame) must build
    exactly the map and poses of the run that reads every count back."""
    import gradslam_amd as gs
    s = make_sequence(6, 120, 160, seed=5)
    poses = s["poses"].copy()
    poses[1:] = poses[:1]

    def run(dc):
        old, ops.DEVICE_COUNTS = ops.DEVICE_COUNTS, dc
        try:
            frames = gs.RGBDImages(dev(s["colors"][None]), dev(s["depths"][None]), dev(s["intrinsics"][None]),
                                   dev(poses[None]))
            slam = gs.slam.PointFusion(odom="gradicp", device="cuda")
            pc, prev, out = gs.Pointclouds(device="cuda"), None, []
            for t in range(6):
                live = frames[:, t]
                pc, pose = slam.step(pc, live, prev, inplace=True)
                assert bool(pc._dcount) == dc
                prev = live
                out.append(pose)
            return pc, torch.stack(out)
        finally:
            ops.DEVICE_COUNTS = old

    pc_h, poses_h = run(False)
    pc_d, poses_d = run(True)
    assert pc_d._count_of(0)[0] >= pc_h._n[0]          # the host bound never undercuts the count
    assert torch.equal(poses_d, poses_h)
    for k in ("points_list", "normals_list", "colors_list", "features_list"):
        assert torch.equal(getattr(pc_d, k)[0], getattr(pc_h, k)[0]), k
    assert not pc_d._dcount                             # resolved by the first exact access
    # and the aggregate map (ICPSLAM)
    for dc in (False, True):
        old, ops.DEVICE_COUNTS = ops.DEVICE_COUNTS, dc
        frames = gs.RGBDImages(dev(s["colors"][None]), dev(s["depths"][None]), dev(s["intrinsics"][None]),
                               dev(poses[None]))
        pc, prev = gs.Pointclouds(device="cuda"), None
        slam = gs.slam.ICPSLAM(odom="icp", device="cuda")
        for t in range(3):
            live = frames[:, t]
            pc, _ = slam.step(pc, live, prev, inplace=True)
            prev = live
        ops.DEVICE_COUNTS = old
        if dc:
            assert torch.equal(pc.points_list[0], ref_pts)
        else:
            ref_pts = pc.points_list[0].clone()


def test_icp_large_solve_matches_oracle(ops):
    """A solve with more query rows than FS_REDUCE_ROWS (the path with the extra row-reduction launch and
    several waves of blocks per XCD): transform within 1e-6 of the oracle, neighbours identical."""
    s = make_sequence(2, 600, 800, seed=21)            # ds=2 lattice: ~114k source points = 1190 row units of 96
    K = dev(s["intrinsics"][0])
    pose = dev(s["poses"][0])
    sets = []
    for f in (0, 1):
        d = dev(s["depths"][f, ..., 0])
        v, n, _, _ = ops.frame_maps(d, K)
        gv, gn = ops.global_maps(v, n, d, pose)
        sets.append(ops.downsample_frame(gv, gn, None, d, 2))
    (tgt, tn, _), (src, _, _) = sets
    assert src.shape[0] > 1024 * 96                     # FS_REDUCE_ROWS row units
    T, idx = ops.icp(src, tgt, tn, mode=1, numiters=4)
    To, idxo = o.icp(host(src), host(tgt), host(tn), mode=1, numiters=4)[:2]
    assert np.abs(host(T) - To).max() <= 1e-6
    same_bits(host(idx), idxo)


def test_device_side_counts_batch_and_container_ops(ops):
    """B=2 ragged batch through the read-back-free frame loop, then the Pointclouds API on a map whose
    counts are still on the device: every accessor must resolve them and agree with the host-count run."""
    import gradslam_amd as gs
    sA, sB = make_sequence(4, 96, 128, seed=1), make_sequence(4, 96, 128, seed=2, hole_frac=0.3)
    colors = np.stack([sA["colors"], sB["colors"]])
    depths = np.stack([sA["depths"], sB["depths"]])
    depths[1, 2] = 0.0                       # a frame without a single valid pixel in sequence 1
    K = np.stack([sA["intrinsics"], sB["intrinsics"]])
    poses = np.stack([sA["poses"], sB["poses"]])
    poses[:, 1:] = poses[:, :1]

    def run(dc):
        old, ops.DEVICE_COUNTS = ops.DEVICE_COUNTS, dc
        try:
            frames = gs.RGBDImages(dev(colors), dev(depths), dev(K), dev(poses))
            slam = gs.slam.PointFusion(odom="icp", device="cuda")
            pc, prev = gs.Pointclouds(device="cuda"), None
            for t in range(4):
                live = frames[:, t]
                pc, _ = slam.step(pc, live, prev, inplace=True)
                prev = live
            return pc
        finally:
            ops.DEVICE_COUNTS = old

    ref = run(False)
    n_ref = [p.shape[0] for p in ref.points_list]
    assert n_ref[0] != n_ref[1]
    # each accessor on a fresh device-count map
    pc = run(True)
    assert pc._dcount and len(pc) == 2 and pc._dcount      # len() does not need the counts
    assert pc.num_points_per_pointcloud.tolist() == n_ref and not pc._dcount
    pc = run(True)
    assert torch.equal(pc.points_padded, ref.points_padded) and torch.equal(pc.nonpad_mask, ref.nonpad_mask)
    pc = run(True)
    c = pc.clone()
    assert [p.shape[0] for p in c.points_list] == n_ref and torch.equal(c.features_list[1], ref.features_list[1])
    pc = run(True)
    one = pc[1]
    assert torch.equal(one.points_list[0], ref.points_list[1]) and torch.equal(one.colors_list[0], ref.colors_list[1])
    pc = run(True)
    pc.append_points(ref)
    assert [p.shape[0] for p in pc.points_list] == [2 * n for n in n_ref]
    assert torch.equal(pc.normals_list[0][n_ref[0]:], ref.normals_list[0])
    pc = run(True)
    assert pc.has_points and pc.cpu().points_list[1].shape[0] == n_ref[1]


def test_lattice_source_feeds_icp_like_the_compacted_set(ops):
    """gs_lattice_source_f32: same points as global_maps + downsample_frame (bit-exact), NaN in the empty
    slots; the ICP solve on the un-compacted lattice skips them and equals the solve on the compacted set
    (same neighbours; transform within float64 summation-order noise)."""
    s = make_sequence(2, 240, 320, seed=8, hole_frac=0.2)
    K, pose = dev(s["intrinsics"][0]), dev(s["poses"][0])
    d0, d1 = dev(s["depths"][0, ..., 0]), dev(s["depths"][1, ..., 0])
    v0, n0, _, _ = ops.frame_maps(d0, K)
    gv0, gn0 = ops.global_maps(v0, n0, d0, pose)
    tgt, tn, _ = ops.downsample_frame(gv0, gn0, None, d0, 2)
    v1, n1, _, _ = ops.frame_maps(d1, K)
    gv1, _ = ops.global_maps(v1, n1, d1, pose)
    src, _, _ = ops.downsample_frame(gv1, None, None, d1, 4)
    lat = ops.lattice_source(v1, d1, pose, 4)
    valid = ~torch.isnan(lat[:, 0])
    assert lat.shape[0] == 60 * 80 and int(valid.sum()) == src.shape[0] < lat.shape[0]
    assert torch.equal(lat[valid], src) and bool(torch.isnan(lat[~valid]).all())
    n_tgt = torch.tensor([tgt.shape[0]], dtype=torch.int64, device="cuda")     # device count: grid path
    for mode in (1, 0):
        T_c, idx_c = ops.icp(src, tgt, tn, mode=mode, numiters=10, n_tgt_dev=n_tgt)
        T_l, idx_l = ops.icp(lat, tgt, tn, mode=mode, numiters=10, n_tgt_dev=n_tgt)
        assert torch.equal(idx_l[valid], idx_c) and bool((idx_l[~valid] == -1).all())
        assert float((T_l - T_c).abs().max()) <= 1e-6 and bool(torch.isfinite(T_l).all())


def test_icp_against_the_map_without_gathering_targets(ops):
    """gs_icp_map_dc_f32 (targets = map rows whose projection lies on the lattice, binned straight from the
    map) gives bit for bit the transform of select_targets + icp, with host and with device-side map counts."""
    s = make_sequence(3, 240, 320, seed=4, hole_frac=0.1)
    K, pose = dev(s["intrinsics"][0]), dev(s["poses"][0])
    maps = []
    for f in (0, 1):      # a "map" with duplicates: all valid pixels of two frames
        d = dev(s["depths"][f, ..., 0])
        v, n, _, _ = ops.frame_maps(d, K)
        gv, gn = ops.global_maps(v, n, d, pose)
        maps.append(ops.downsample_frame(gv, gn, None, d, 1)[:2])
    P, N = torch.cat([m[0] for m in maps]), torch.cat([m[1] for m in maps])
    d2 = dev(s["depths"][2, ..., 0])
    v2, _, _, _ = ops.frame_maps(d2, K)
    src = ops.lattice_source(v2, d2, pose, 4)
    pix = ops.project_map(P, pose, K, 240, 320)
    tgt, tn, _ = ops.select_targets(pix, 320, 4, P, N)
    assert 2048 < tgt.shape[0] < P.shape[0] // 4
    n_tgt = torch.tensor([tgt.shape[0]], dtype=torch.int64, device="cuda")
    for mode in (1, 0):
        T_ref = ops.icp(src, tgt, tn, compose=pose, mode=mode, numiters=10, n_tgt_dev=n_tgt, return_idx=False)
        T_map = ops.icp_map(src, P, N, pix, 320, 4, compose=pose, mode=mode, numiters=10)
        assert torch.equal(T_map, T_ref)
        # bound-sized buffers with garbage rows behind the device-side count
        Pb, Nb = torch.cat([P, P[:1000] + 5.0]), torch.cat([N, N[:1000]])
        pixb = torch.cat([pix, torch.zeros(1000, dtype=torch.int32, device="cuda")])   # pixel 0 is on the lattice
        n_map = torch.tensor([P.shape[0]], dtype=torch.int64, device="cuda")
        assert torch.equal(ops.icp_map(src, Pb, Nb, pixb, 320, 4, n_map_dev=n_map, compose=pose, mode=mode, numiters=10), T_ref)
