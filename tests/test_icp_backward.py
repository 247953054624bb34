"""Config C3: gradients through point_to_plane_gradICP.

CPU: the float64 numpy oracle of the backward pass (oracle/icp_backward.py) is pinned against the
reference's OWN autograd gradients (tests/golden/icp_grad.npz, from oracle/make_golden.py).
GPU: the hand-written HIP backward (gs_icp_backward_f32, through torch.autograd) against that
oracle on the same tape semantics, against the golden gradients, and at 640x480 size."""
import numpy as np
import pytest
import torch

from oracle import icp_backward as ib


def rel(a, b):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max() / max(np.abs(b).max(), 1e-30))


CASES = [(1, None, "K1"), (5, None, "K5"), (20, None, "K20"), (5, 1e-4, "K5_thr")]


@pytest.mark.parametrize("K,thr,tag", CASES)
def test_numpy_backward_oracle_matches_reference_autograd(golden, K, thr, tag):
    g, u = golden("icp_grad"), golden("icp_unit")
    T, tape = ib.icp_forward_tape(u["src"], u["tgt"], u["tgt_normals"], numiters=K, dist_thresh=thr)
    np.testing.assert_allclose(T, g[tag + "_T"], atol=2e-5, rtol=0)
    sb, tb, nb, ibar = ib.icp_backward(tape, u["tgt"], u["tgt_normals"], g["W"], u["src"])
    assert rel(sb, g[tag + "_src"]) < 5e-4 and rel(tb, g[tag + "_tgt"]) < 5e-4 and rel(nb, g[tag + "_tn"]) < 5e-4
    assert ibar.shape == (4, 4) and np.all(ibar[3] == 0)


def _c3_inputs_cpu(g):
    """The two ds = 4 clouds of the 640x480 golden, regenerated from the seed with the oracle's (reference-pinned) frame
    maps and down-sampler; their float64 sums must equal the ones recorded with the golden."""
    from gradslam_amd.datasets.synthetic import make_sequence
    from oracle import oracle as o
    s = make_sequence(2, 480, 640, seed=int(g["seed"]))
    pts = []
    for f in range(2):
        d = s["depths"][f, ..., 0]
        v, n = o.frame_maps(d, s["intrinsics"][0])[:2]
        gv, gn = o.global_maps(v, n, d, s["poses"][0])
        pts.append(o.downsample_frame(gv, gn, None, d, 4)[:2])
    (tgt, tn), (src, _) = pts
    for name, x in (("src", src), ("tgt", tgt), ("tn", tn)):
        assert np.abs(x.astype(np.float64).sum(0) - g["in_sum_" + name]).max() < 1e-9, name
    return src, tgt, tn


def test_numpy_backward_oracle_matches_reference_autograd_at_640x480(golden):
    """BASELINE config C3 at the benchmarked size against the REAL reference: autograd through 20 iterations of
    point_to_plane_gradICP on 18 216 x 18 281 points (tests/golden/c3_grad640.npz, oracle/make_golden_c3.py: every 8th
    gradient row + column sums + norms)."""
    g = golden("c3_grad640")
    src, tgt, tn = _c3_inputs_cpu(g)
    T, tape = ib.icp_forward_tape(src, tgt, tn, numiters=20)
    np.testing.assert_allclose(T, g["T"], atol=2e-5, rtol=0)
    sb, tb, nb, _ = ib.icp_backward(tape, tgt, tn, g["W"], src)
    st = int(g["stride"])
    for have, name in ((sb, "src"), (tb, "tgt"), (nb, "tn")):
        assert rel(have[::st], g["grad_" + name]) < 1e-3, name          # measured 1.3e-4 / 7e-5 / 2.9e-4
        assert abs(np.linalg.norm(have) / float(g["grad_norm_" + name]) - 1.0) < 1e-3, name


def test_se3_exp_adjoint_by_finite_differences():
    rng = np.random.default_rng(1)
    # the small-angle branch (|omega| < 1e-6, se3utils.py:89-91) has its own derivative (V = I + w^):
    # probe it with a step that stays inside the branch
    for xi, h in ((rng.standard_normal(6) * 0.3, 1e-6), (np.array([0.1, 0.2, -0.1, 1e-8, 0, 0]), 1e-9),
                  (rng.standard_normal(6) * 2.0, 1e-6)):
        W = rng.standard_normal((4, 4))
        W[3] = 0
        ana = ib.se3_exp_adjoint(xi, W)
        num = np.zeros(6)
        for i in range(6):
            e = np.zeros(6)
            e[i] = h
            num[i] = np.sum(W * (ib.se3_exp(xi + e) - ib.se3_exp(xi - e))) / (2 * h)
        np.testing.assert_allclose(ana, num, rtol=1e-5, atol=1e-6)


# ------------------------------------------------------------------------------------------ GPU
def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.gpu
@pytest.mark.parametrize("K,thr,tag", CASES)
def test_hip_backward_matches_oracle_and_reference(golden, K, thr, tag):
    from gradslam_amd.odometry import icputils
    g, u = golden("icp_grad"), golden("icp_unit")
    leaf = [dev(u[k]).requires_grad_(True) for k in ("src", "tgt", "tgt_normals")]
    init = torch.eye(4, device="cuda", requires_grad=True)
    T, idx = icputils.point_to_plane_gradICP(leaf[0][None], leaf[1][None], leaf[2][None], init, numiters=K,
                                             dist_thresh=thr)
    assert T.requires_grad and not idx.requires_grad
    (T * dev(g["W"])).sum().backward()
    np.testing.assert_allclose(T.detach().cpu().numpy(), g[tag + "_T"], atol=2e-5, rtol=0)
    # oracle backward on the oracle tape (same algorithm, float64 numpy)
    _, tape = ib.icp_forward_tape(u["src"], u["tgt"], u["tgt_normals"], numiters=K, dist_thresh=thr)
    sb, tb, nb, ibar = ib.icp_backward(tape, u["tgt"], u["tgt_normals"], g["W"], u["src"])
    got = [t.grad.cpu().numpy() for t in leaf]
    for have, orc, name in zip(got, (sb, tb, nb), ("src", "tgt", "tn")):
        assert rel(have, orc) < 2e-4, name                      # vs oracle
        assert rel(have, g[tag + "_" + name]) < 5e-4, name       # vs the reference's autograd
    assert rel(init.grad.cpu().numpy(), ibar) < 2e-4


@pytest.mark.gpu
def test_no_grad_path_is_unchanged_and_idx_matches(golden):
    from gradslam_amd import ops
    u = golden("icp_unit")
    src, tgt, tn = (dev(u[k]) for k in ("src", "tgt", "tgt_normals"))
    T0, idx0 = ops.icp(src, tgt, tn, mode=1, numiters=7)
    T1, idx1 = ops.grad_icp(src.clone().requires_grad_(True), tgt, tn, numiters=7)
    assert torch.equal(T0, T1.detach()) and torch.equal(idx0, idx1)


@pytest.mark.gpu
def test_backward_at_640x480_grid_engine():
    """BASELINE config C3 size: ~18k x ~18k points, 20 iterations, grid engine + tape; the HIP
    gradient must agree with the float64 oracle backward evaluated on the oracle's own tape."""
    from gradslam_amd import ops
    from gradslam_amd.datasets.synthetic import make_sequence
    s = make_sequence(2, 480, 640, seed=8)
    K = torch.from_numpy(s["intrinsics"][0]).cuda()
    pts = []
    for f in range(2):
        d = torch.from_numpy(s["depths"][f, ..., 0]).cuda()
        v, n, _, _ = ops.frame_maps(d, K)
        gv, gn = ops.global_maps(v, n, d, torch.from_numpy(s["poses"][0]).cuda())
        pts.append(ops.downsample_frame(gv, gn, None, d, 4)[:2])
    (tgt, tn), (src, _) = pts
    W = torch.from_numpy(np.random.default_rng(2).standard_normal((4, 4)).astype(np.float32)).cuda()
    src_l = src.clone().requires_grad_(True)
    tgt_l = tgt.clone().requires_grad_(True)
    T, _ = ops.grad_icp(src_l, tgt_l, tn, numiters=20)
    (T * W).sum().backward()
    assert torch.isfinite(src_l.grad).all() and torch.isfinite(tgt_l.grad).all() and float(src_l.grad.abs().max()) > 0
    To, tape = ib.icp_forward_tape(src.cpu().numpy(), tgt.cpu().numpy(), tn.cpu().numpy(), numiters=20)
    np.testing.assert_allclose(T.detach().cpu().numpy(), To, atol=1e-6, rtol=0)
    sb, tb, _, _ = ib.icp_backward(tape, tgt.cpu().numpy(), tn.cpu().numpy(), W.cpu().numpy(), src.cpu().numpy())
    assert rel(src_l.grad.cpu().numpy(), sb) < 1e-3
    assert rel(tgt_l.grad.cpu().numpy(), tb) < 1e-3


@pytest.mark.gpu
def test_hip_backward_matches_reference_autograd_at_640x480(golden):
    """Config C3 at the benchmarked size, HIP against the REAL reference's autograd (tests/golden/c3_grad640.npz): the
    clouds come from the HIP frame maps and down-sampler (bit-exact: their sums must match the recorded ones), the
    gradients of <W, T> through 20 gradICP iterations from gs_icp_tape_f32 + gs_icp_backward_f32."""
    from gradslam_amd import ops
    from gradslam_amd.datasets.synthetic import make_sequence
    g = golden("c3_grad640")
    s = make_sequence(2, 480, 640, seed=int(g["seed"]))
    K = torch.from_numpy(s["intrinsics"][0]).cuda()
    pts = []
    for f in range(2):
        d = torch.from_numpy(s["depths"][f, ..., 0]).cuda()
        v, n, _, _ = ops.frame_maps(d, K)
        gv, gn = ops.global_maps(v, n, d, torch.from_numpy(s["poses"][0]).cuda())
        pts.append(ops.downsample_frame(gv, gn, None, d, 4)[:2])
    (tgt, tn), (src, _) = pts
    for name, x in (("src", src), ("tgt", tgt), ("tn", tn)):
        assert np.abs(x.double().sum(0).cpu().numpy() - g["in_sum_" + name]).max() < 1e-9, name
    leaf = [t.clone().requires_grad_(True) for t in (src, tgt, tn)]
    T, _ = ops.grad_icp(leaf[0], leaf[1], leaf[2], numiters=20)
    (T * dev(g["W"])).sum().backward()
    np.testing.assert_allclose(T.detach().cpu().numpy(), g["T"], atol=2e-5, rtol=0)
    st = int(g["stride"])
    for t, name in zip(leaf, ("src", "tgt", "tn")):
        have = t.grad.cpu().numpy()
        assert rel(have[::st], g["grad_" + name]) < 1e-3, (name, rel(have[::st], g["grad_" + name]))
        assert abs(np.linalg.norm(have.astype(np.float64)) / float(g["grad_norm_" + name]) - 1.0) < 1e-3, name


@pytest.mark.gpu
def test_depth_gradients_through_frame_maps(golden):
    """d/d(depth) of <Wv, vertex_map> + <Wn, normal_map> + <Wa, alpha> vs the reference's autograd."""
    import gradslam_amd as gs
    from gradslam_amd.slam import fusionutils as fu
    g = golden("depth_grad")
    d1 = dev(g["depths"][None, 1:2]).requires_grad_(True)
    rgb = torch.zeros((1, 1, 96, 128, 3), device="cuda")
    f1 = gs.RGBDImages(rgb, d1, dev(g["intrinsics"][None, None]), dev(g["poses"][None, :1]))
    alpha = f1._alpha_map(0.6)
    loss = ((f1.vertex_map[0, 0] * dev(g["Wv"])).sum() + (f1.normal_map[0, 0] * dev(g["Wn"])).sum()
            + (alpha[0, 0, ..., 0] * dev(g["Wa"])).sum())
    loss.backward()
    got, ref = d1.grad[0, 0, ..., 0].cpu().numpy(), g["maps_depth_grad"]
    assert np.isfinite(got).all()
    # float32 normal normalisation amplifies rounding on a few near-degenerate pixels: compare robustly
    err = np.abs(got - ref)
    assert np.median(err) < 1e-4 * np.abs(ref).max() and (err < 1e-2 * np.abs(ref).max()).mean() > 0.999
    assert fu.get_alpha(f1.vertex_map, 0.6, dim=4, keepdim=True).shape == alpha.shape


@pytest.mark.gpu
def test_intrinsics_gradients_through_frame_maps(golden):
    """d/dK of <Wv, vertex_map> + <Wn, normal_map> + <Wa, alpha> vs the reference's autograd through
    inverse_intrinsics (tests/golden/intrinsics_grad.npz, a hole-free frame: next to depth holes the reference's own
    float32 and float64 gradients disagree): entries fx, fy, cx, cy; with and without a depth gradient."""
    import gradslam_amd as gs
    g, gk = golden("depth_grad"), golden("intrinsics_grad")
    for depth_grad in (True, False):
        K = dev(gk["intrinsics"][None, None]).requires_grad_(True)
        d1 = dev(gk["depth"][None, None, ..., None]).requires_grad_(depth_grad)
        f1 = gs.RGBDImages(torch.zeros((1, 1, 96, 128, 3), device="cuda"), d1, K, dev(gk["pose"][None, None]))
        alpha = f1._alpha_map(0.6)
        loss = ((f1.vertex_map[0, 0] * dev(g["Wv"])).sum() + (f1.normal_map[0, 0] * dev(g["Wn"])).sum()
                + (alpha[0, 0, ..., 0] * dev(g["Wa"])).sum())
        loss.backward()
        got, ref = K.grad[0, 0].cpu().numpy(), gk["K_grad"]
        assert np.isfinite(got).all() and np.array_equal(got == 0, ref == 0)
        assert np.abs(got - ref).max() <= 1e-3 * np.abs(ref).max(), (got, ref)
        if depth_grad:
            err = np.abs(d1.grad[0, 0, ..., 0].cpu().numpy() - gk["depth_grad"])
            assert err.max() <= 1e-3 * np.abs(gk["depth_grad"]).max()


@pytest.mark.gpu
def test_depth_to_pose_chain_matches_reference_autograd(golden):
    """depth -> vertex -> global vertex -> downsample_rgbdimages -> point_to_plane_gradICP -> <W,T>."""
    import gradslam_amd as gs
    from gradslam_amd.odometry import icputils
    g = golden("depth_grad")
    K, P0 = dev(g["intrinsics"][None, None]), dev(g["poses"][None, :1])
    rgb = torch.zeros((1, 1, 96, 128, 3), device="cuda")
    d1 = dev(g["depths"][None, 1:2]).requires_grad_(True)
    f1 = gs.RGBDImages(rgb, d1, K, P0)
    f0 = gs.RGBDImages(rgb, dev(g["depths"][None, 0:1]), K, P0)
    tg, sr = icputils.downsample_rgbdimages(f0, 4), icputils.downsample_rgbdimages(f1, 4)
    assert sr.points_list[0].requires_grad and not tg.points_list[0].requires_grad
    T, _ = icputils.point_to_plane_gradICP(sr.points_list[0][None], tg.points_list[0][None], tg.normals_list[0][None],
                                           torch.eye(4, device="cuda"), numiters=5)
    np.testing.assert_allclose(T.detach().cpu().numpy(), g["chain_T"], atol=2e-5, rtol=0)
    (T * dev(g["chain_W"])).sum().backward()
    got, ref = d1.grad[0, 0, ..., 0].cpu().numpy(), g["chain_depth_grad"]
    assert (got != 0).sum() == (ref != 0).sum()          # only valid lattice pixels receive gradient
    assert rel(got, ref) < 2e-3


@pytest.mark.gpu
def test_pointfusion_driver_pose_gradient_wrt_live_depth(golden):
    """PointFusion(odom='gradicp')(frames): d <W, pose_1> / d depth_1 through the hand-written backward
    chain equals the reference's autograd (frame 1 reaches pose_1 only through the ICP source)."""
    import gradslam_amd as gs
    g = golden("depth_grad")
    dd = dev(g["slam_depths"][None]).requires_grad_(True)
    pp = dev(g["slam_poses"][None]).clone()
    pp[:, 1:] = pp[:, :1]
    frames = gs.RGBDImages(dev(g["slam_colors"][None]), dd, dev(g["slam_intrinsics"][None, None]), pp)
    _, rp = gs.slam.PointFusion(odom="gradicp", device="cuda")(frames)
    np.testing.assert_allclose(rp[0, 1].detach().cpu().numpy(), g["slam_pose1"], atol=2e-5, rtol=0)
    (rp[0, 1] * dev(g["chain_W"])).sum().backward()
    got, ref = dd.grad[0, 1, ..., 0].cpu().numpy(), g["slam_depth1_grad"]
    assert rel(got, ref) < 5e-3 and (got != 0).sum() == (ref != 0).sum()
    # frame 0 reaches pose_1 only through the map it was fused into (the ICP targets and their normals)
    got0, ref0 = dd.grad[0, 0, ..., 0].cpu().numpy(), golden("slam_grad0")["depth0_grad"]
    err0 = np.abs(got0 - ref0)
    # (the supports agree up to a few pixels whose reference gradient is float32 round-off of an exact zero)
    assert abs(int((got0 != 0).sum()) - int((ref0 != 0).sum())) <= 0.01 * (ref0 != 0).sum()
    assert np.median(err0[ref0 != 0]) < 1e-3 * np.abs(ref0).max()
    assert (err0 < 1e-2 * np.abs(ref0).max()).mean() > 0.999


@pytest.mark.gpu
def test_config_c3_as_stated_at_640x480_vs_reference_autograd(golden):
    """BASELINE configs[2] / SURVEY C3 exactly as stated: depth.requires_grad_(); PointFusion(odom="gradicp") over two
    640x480 frames; recovered_poses.sum().backward(); depth.grad of both frames against the REAL reference's autograd
    (tests/golden/c3_driver640.npz, oracle/make_golden_c3.py --driver: every 8th image row, norms, sums, supports).
    The whole backward chain runs: ICP (20 iterations) -> down-sampler -> global / local frame maps for frame 1; ICP
    targets and normals -> map append -> frame maps for frame 0.  Tolerances as at 96x128 (the per-pixel gradients of
    frame 0 pass through sums over thousands of matches): frame 1 relative error < 5e-3; frame 0 median error < 1e-3 of
    the largest gradient, 99.9 % of the pixels within 1e-2 of it; norms within 1 %."""
    import gradslam_amd as gs
    from gradslam_amd.datasets.synthetic import make_sequence
    g = golden("c3_driver640")
    s = make_sequence(2, 480, 640, seed=int(g["seed"]))
    assert abs(float(s["depths"].astype(np.float64).sum()) - float(g["depth_sum"])) < 1e-6 * float(g["depth_sum"])
    dd = dev(s["depths"][None]).requires_grad_(True)
    pp = dev(s["poses"][None]).clone()
    pp[:, 1:] = pp[:, :1]
    frames = gs.RGBDImages(dev(s["colors"][None]), dd, dev(s["intrinsics"][None]), pp)
    _, rp = gs.slam.PointFusion(odom="gradicp", device="cuda")(frames)
    np.testing.assert_allclose(rp[0].detach().cpu().numpy(), g["poses"], atol=2e-5, rtol=0)
    rp.sum().backward()
    got = dd.grad[0, :, :, :, 0].cpu().numpy()               # (2, H, W)
    st = int(g["stride"])
    ref = g["grad_rows"]
    # frame 1: only the ICP source lattice carries gradient
    assert int((got[1] != 0).sum()) == int(g["support"][1])
    assert rel(got[1][::st], ref[1]) < 5e-3
    # frame 0: through the map (targets and their normals)
    assert abs(int((got[0] != 0).sum()) - int(g["support"][0])) <= 0.01 * int(g["support"][0])
    err0 = np.abs(got[0][::st] - ref[0])
    big = float(g["grad_absmax"][0])
    assert np.median(err0[ref[0] != 0]) < 1e-3 * big
    assert (err0 < 1e-2 * big).mean() > 0.999
    for f in (0, 1):
        nrm = float(np.sqrt((got[f].astype(np.float64) ** 2).sum()))
        assert abs(nrm - float(g["grad_norm"][f])) <= 1e-2 * float(g["grad_norm"][f]), (f, nrm, float(g["grad_norm"][f]))
        assert abs(float(got[f].astype(np.float64).sum()) - float(g["grad_sum"][f])) <= 2e-2 * float(g["grad_norm"][f])


# ------------------------------------------------------------------------------------------ hard-LM ICP (mode 0)
@pytest.mark.parametrize("K,thr,tag", CASES)
def test_numpy_backward_oracle_matches_reference_autograd_hard_lm(golden, K, thr, tag):
    """point_to_plane_ICP: accepted steps carry the gradient, rejected ones and the accept test do not."""
    g, u = golden("icp0_grad"), golden("icp_unit")
    T, tape = ib.icp_forward_tape(u["src"], u["tgt"], u["tgt_normals"], numiters=K, dist_thresh=thr, mode=0)
    np.testing.assert_allclose(T, g[tag + "_T"], atol=2e-5, rtol=0)
    if K >= 5:   # the fixture exercises both branches
        acc = [bool(tape["trace"][k, 1] < tape["trace"][k, 0]) for k in range(K)]
        assert any(acc) and not all(acc)
    sb, tb, nb, _ = ib.icp_backward(tape, u["tgt"], u["tgt_normals"], g["W"], u["src"])
    assert rel(sb, g[tag + "_src"]) < 5e-4 and rel(tb, g[tag + "_tgt"]) < 5e-4 and rel(nb, g[tag + "_tn"]) < 5e-4


@pytest.mark.gpu
@pytest.mark.parametrize("K,thr,tag", CASES)
def test_hip_backward_hard_lm_matches_oracle_and_reference(golden, K, thr, tag):
    from gradslam_amd.odometry import icputils
    g, u = golden("icp0_grad"), golden("icp_unit")
    leaf = [dev(u[k]).requires_grad_(True) for k in ("src", "tgt", "tgt_normals")]
    init = torch.eye(4, device="cuda", requires_grad=True)
    T, idx = icputils.point_to_plane_ICP(leaf[0][None], leaf[1][None], leaf[2][None], init, numiters=K, dist_thresh=thr)
    assert T.requires_grad and not idx.requires_grad
    (T * dev(g["W"])).sum().backward()
    np.testing.assert_allclose(T.detach().cpu().numpy(), g[tag + "_T"], atol=2e-5, rtol=0)
    _, tape = ib.icp_forward_tape(u["src"], u["tgt"], u["tgt_normals"], numiters=K, dist_thresh=thr, mode=0)
    sb, tb, nb, ibar = ib.icp_backward(tape, u["tgt"], u["tgt_normals"], g["W"], u["src"])
    for have, orc, name in zip([t.grad.cpu().numpy() for t in leaf], (sb, tb, nb), ("src", "tgt", "tn")):
        assert rel(have, orc) < 2e-4, name
        assert rel(have, g[tag + "_" + name]) < 5e-4, name
    assert rel(init.grad.cpu().numpy(), ibar) < 2e-4


# ------------------------------------------------------------------------------------------ differentiable mapping
@pytest.mark.gpu
def test_hip_fuse_backward_matches_oracle():
    """gs_fuse_append_backward_f32 (through FuseAppendFunction) against the float64 numpy adjoint on the same
    correspondences: matched, unmatched, appended rows."""
    from gradslam_amd import ops
    from oracle import fusion_backward as fb
    rng = np.random.default_rng(2)
    H, W, n0 = 6, 8, 30
    P = H * W
    old = [rng.standard_normal((n0, 3)).astype(np.float32) for _ in range(3)]
    cc = (rng.random((n0, 1)) + 0.2).astype(np.float32)
    frame = [rng.standard_normal((H, W, 3)).astype(np.float32) for _ in range(3)]
    alpha = rng.random((H, W)).astype(np.float32)
    depth = (rng.random((H, W)) + 0.5).astype(np.float32)
    depth[0, :3] = 0.0
    best = np.full(P, -1, np.int32)
    rows = rng.choice(n0, 10, replace=False)
    pixs = rng.choice(np.arange(8, P), 10, replace=False)
    best[pixs] = rows
    leaves = [dev(a).requires_grad_(True) for a in old + [cc] + frame + [alpha]]
    out = ops.FuseAppendFunction.apply(*leaves, dev(depth), dev(best), True)
    n1 = out[0].shape[0]
    Wt = [rng.standard_normal((n1, c)).astype(np.float32) for c in (3, 3, 3, 1)]
    sum((o_ * dev(w)).sum() for o_, w in zip(out, Wt)).backward()
    pix_of = np.full(n0, -1)
    pix_of[rows] = pixs
    new_pix = np.flatnonzero((depth.reshape(-1) > 0) & (best < 0))
    assert n1 == n0 + len(new_pix)
    o64 = [a.astype(np.float64) for a in old]
    f64 = [a.reshape(P, 3).astype(np.float64) for a in frame]
    ob, cb, fbar, ab = fb.fuse_backward(o64, cc[:, 0].astype(np.float64), f64, alpha.reshape(-1).astype(np.float64),
                                        pix_of, new_pix, [w.astype(np.float64) for w in Wt[:3]],
                                        Wt[3][:, 0].astype(np.float64))
    got = [t.grad.cpu().numpy() for t in leaves]
    for t in range(3):
        assert rel(got[t], ob[t]) < 1e-5 and rel(got[4 + t].reshape(P, 3), fbar[t]) < 1e-5
    assert rel(got[3][:, 0], cb) < 1e-5 and rel(got[7].reshape(-1), ab) < 1e-5


@pytest.mark.gpu
def test_differentiable_mapping_matches_reference_autograd(golden):
    """PointFusion(odom='gt') over 3 frames with depth and rgb on the tape: the fused map equals the reference's
    and d<W, map>/d(depth, rgb) matches the reference's autograd (tests/golden/fusion_grad.npz) -- frame maps,
    global maps, alpha and the fuse all run through their hand-written HIP backward kernels."""
    import gradslam_amd as gs
    g = golden("fusion_grad")
    depth = dev(g["depths"][None]).requires_grad_(True)
    rgb = dev(g["colors"][None]).requires_grad_(True)
    poses = dev(g["poses"][None]).requires_grad_(True)
    frames = gs.RGBDImages(rgb, depth, dev(g["intrinsics"][None]), poses)
    pc, _ = gs.slam.PointFusion(odom="gt", dsratio=4, device="cuda")(frames)
    assert pc.points_list[0].shape[0] == int(g["n"]) and pc.points_list[0].requires_grad
    loss = 0
    for k in ("points", "normals", "colors", "features"):
        t = getattr(pc, k + "_list")[0]
        assert np.abs(t.detach().cpu().numpy() - g["map_" + k]).max() <= 2e-5 * max(1.0, np.abs(g["map_" + k]).max())
        loss = loss + (t * dev(g["W_" + k])).sum()
    loss.backward()
    assert rel(rgb.grad[0].cpu().numpy(), g["rgb_grad"]) < 1e-4
    # the poses reach the map through the global maps (R v + t, R n): 12 sums per frame, last row untouched
    assert rel(poses.grad[0].cpu().numpy(), g["poses_grad"]) < 1e-4 and np.all(poses.grad[0, :, 3].cpu().numpy() == 0)
    # depth: as in test_depth_gradients_through_frame_maps, the reference's float32 normal normalisation amplifies
    # rounding on the few near-degenerate pixels next to depth holes (4 of 3840 here): compare robustly
    got, ref = depth.grad[0].cpu().numpy(), g["depth_grad"]
    err = np.abs(got - ref)
    assert np.isfinite(got).all() and np.median(err) < 1e-4 * np.abs(ref).max()
    assert (err < 1e-2 * np.abs(ref).max()).mean() > 0.999 and (err < 1e-3 * np.abs(ref).max()).mean() > 0.998


@pytest.mark.gpu
def test_differentiable_aggregate_map_matches_reference_autograd(golden):
    """ICPSLAM(odom='gt'): the aggregated map back-propagates to depth like the reference's."""
    import gradslam_amd as gs
    g = golden("fusion_grad")
    depth = dev(g["depths"][None, :2]).requires_grad_(True)
    frames = gs.RGBDImages(dev(g["colors"][None, :2]), depth, dev(g["intrinsics"][None]), dev(g["poses"][None, :2]))
    pc, _ = gs.slam.ICPSLAM(odom="gt", dsratio=4, device="cuda")(frames)
    assert pc.points_list[0].shape[0] == int(g["agg_n"]) and not pc.has_features
    Wa = g["agg_W"]
    ((pc.points_list[0] * dev(Wa)).sum() + (pc.normals_list[0] * dev(Wa[::-1].copy())).sum()).backward()
    got, ref = depth.grad[0].cpu().numpy(), g["agg_depth_grad"]
    err = np.abs(got - ref)
    # median error ~1e-6; the handful of float32-degenerate normals next to depth holes (same pixels as in the
    # fusion test) are the only outliers
    assert np.median(err) < 1e-5 * np.abs(ref).max() and (err < 1e-2 * np.abs(ref).max()).mean() > 0.995
    assert (got != 0).sum() == (ref != 0).sum()


_DET_SCRIPT = r"""
import sys
import numpy as np, torch
sys.path.insert(0, %r)
from gradslam_amd import ops
from gradslam_amd.datasets.synthetic import make_sequence
s = make_sequence(2, 480, 640, seed=0)
K = torch.from_numpy(s["intrinsics"][0]).cuda()
pts = []
for f in range(2):
    d = torch.from_numpy(s["depths"][f, ..., 0]).cuda()
    v, n, _, _ = ops.frame_maps(d, K)
    gv, gn = ops.global_maps(v, n, d, torch.from_numpy(s["poses"][0]).cuda())
    pts.append(ops.downsample_frame(gv, gn, None, d, 4)[:2])
(tgt, tn), (src, _) = pts
tgt, tn = tgt[::3].contiguous(), tn[::3].contiguous()      # three source points per target: shared targets everywhere
W = torch.from_numpy(np.random.default_rng(0).standard_normal((4, 4)).astype(np.float32)).cuda()
runs = []
for r in range(3):
    leaf = [t.clone().requires_grad_(True) for t in (src, tgt, tn)]
    T, _ = ops.grad_icp(leaf[0], leaf[1], leaf[2], numiters=20)
    (T * W).sum().backward()
    runs.append([t.grad.cpu().numpy() for t in leaf])
np.savez(sys.argv[1], **{"r%%d_%%d" %% (r, k): runs[r][k] for r in range(3) for k in range(3)})
"""


@pytest.mark.gpu
def test_deterministic_backward_is_bitwise_reproducible(tmp_path):
    """VERDICT r04 #7b: GRADSLAM_HIP_DETERMINISTIC_BACKWARD=1 replaces the float64 atomics of the ICP backward (the adds
    into a target that several source points share happen in scheduling order) by a stable sort of the (target, source)
    pairs and one thread per target that adds in source order: three runs of taped forward + backward through 20 gradICP
    iterations at 640x480 (6 000 targets for 18 000 source points: every target is shared) give the same bits, and the
    same gradients as the atomic form up to float32 rounding."""
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = {}
    for mode in ("1", "0"):
        out = str(tmp_path / ("det%s.npz" % mode))
        subprocess.run([sys.executable, "-c", _DET_SCRIPT % repo, out], check=True, timeout=900,
                       env=dict(os.environ, GRADSLAM_HIP_DETERMINISTIC_BACKWARD=mode))
        outs[mode] = np.load(out)
    d, a = outs["1"], outs["0"]
    for k in range(3):
        assert np.isfinite(d["r0_%d" % k]).all() and np.abs(d["r0_%d" % k]).max() > 0
        for r in (1, 2):
            assert np.array_equal(d["r0_%d" % k].view(np.int32), d["r%d_%d" % (r, k)].view(np.int32)), (k, r)
        scale = np.abs(a["r0_%d" % k]).max()
        assert np.abs(d["r0_%d" % k] - a["r0_%d" % k]).max() <= 1e-5 * scale, k
