"""GPU tests of the drop-in Python API (gradslam_amd.RGBDImages / Pointclouds / slam.*): the
same calls a gradslam user makes, checked against the golden vectors recorded from the real
reference and against the oracle's frame loop."""
import math

import numpy as np
import pytest
import torch

from gradslam_amd.datasets.synthetic import make_sequence
from oracle import oracle as o
from oracle import slam as oslam
from gradslam_amd.metrics import ate_rmse as ate

pytestmark = pytest.mark.gpu

DIST_TH, DOT_TH, SIGMA = 0.05, math.cos(20 * math.pi / 180), 0.6
T = torch.from_numpy


def host(t):
    return t.detach().cpu().numpy()


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from gradslam_amd import ops as _ops
    return _ops


@pytest.fixture(scope="module")
def gs():
    assert torch.cuda.is_available()
    import gradslam_amd
    return gradslam_amd


def msrd_frames(gs, g, channels_first=False):
    colors, depths = T(g["colors"][None]).cuda(), T(g["depths"][None]).cuda()
    if channels_first:
        colors, depths = colors.permute(0, 1, 4, 2, 3).contiguous(), depths.permute(0, 1, 4, 2, 3).contiguous()
    return gs.RGBDImages(colors, depths, T(g["intrinsics"][None, None]).cuda(), T(g["poses"][None]).cuda(),
                         channels_first=channels_first)


@pytest.mark.parametrize("channels_first", [False, True])
def test_rgbdimages_lazy_maps_match_reference(gs, golden, channels_first):
    g = golden("msrd_b0")
    r = msrd_frames(gs, g, channels_first)
    perm = (lambda t: t.permute(0, 1, 3, 4, 2)) if channels_first else (lambda t: t)
    for prop, key in (("vertex_map", "vertex_map"), ("normal_map", "normal_map"),
                      ("global_vertex_map", "global_vertex_map"), ("global_normal_map", "global_normal_map")):
        assert np.array_equal(host(perm(getattr(r, prop)))[0], g[key]), prop
    assert np.array_equal(host(perm(r.valid_depth_mask))[0], g["depths"] > 0)
    # cache rules: new poses drop only the global maps
    v_before = r.vertex_map
    r.poses = r.poses.clone()
    assert r._global_vertex_map is None and r.vertex_map is v_before
    # slicing keeps cached maps
    sub = r[:, 1]
    assert sub.shape[1] == 1 and np.array_equal(host(perm(sub.vertex_map))[0, 0], g["vertex_map"][1])


def test_fusionutils_tables_match_reference(gs, golden):
    from gradslam_amd.slam import fusionutils as fu
    g = golden("msrd_b0")
    r = msrd_frames(gs, g)
    f0, f1 = r[:, 0], r[:, 1]
    pc0 = fu.update_map_fusion(gs.Pointclouds(device="cuda"), f0, DIST_TH, DOT_TH, SIGMA)
    assert np.array_equal(host(pc0.points_list[0]), g["map0_points"])
    assert np.array_equal(host(pc0.normals_list[0]), g["map0_normals"])
    assert np.array_equal(host(pc0.colors_list[0]), g["map0_colors"])
    np.testing.assert_allclose(host(pc0.features_list[0]), g["map0_ccounts"], rtol=2e-7)
    # use the reference's own map (its ccounts differ from ours by <= 1 ulp through exp) for exact tables
    pc = gs.Pointclouds(points=[T(g["map0_points"]).cuda()], normals=[T(g["map0_normals"]).cuda()],
                        colors=[T(g["map0_colors"]).cuda()], features=[T(g["map0_ccounts"]).cuda()])
    act = fu.find_active_map_points(pc, f1)
    assert act.dtype == torch.int64 and np.array_equal(host(act), g["active"])
    sim, mask = fu.find_similar_map_points(pc, f1, act, DIST_TH, DOT_TH)
    assert np.array_equal(host(mask), g["similar_mask"]) and np.array_equal(host(sim), g["similar"])
    uq = fu.find_best_unique_correspondences(pc, f1, sim)
    assert np.array_equal(host(uq), g["unique"])
    assert np.array_equal(host(fu.find_correspondences(pc, f1, DIST_TH, DOT_TH)), g["unique"])
    # fuse_with_map: alpha of frame 1 comes from our exp -> 1-ulp tolerance on merged values
    fused = fu.fuse_with_map(pc, f1, uq, SIGMA)
    assert fused.points_list[0].shape[0] == g["map1_points"].shape[0]
    np.testing.assert_allclose(host(fused.points_list[0]), g["map1_points"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(host(fused.normals_list[0]), g["map1_normals"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(host(fused.colors_list[0]), g["map1_colors"], rtol=1e-6, atol=1e-4)
    np.testing.assert_allclose(host(fused.features_list[0]), g["map1_ccounts"], rtol=1e-6)
    # reference quirk kept: the input map is merged in place even with inplace=False, but not grown
    assert pc.points_list[0].shape[0] == g["map0_points"].shape[0]
    np.testing.assert_allclose(host(pc.points_list[0]), g["map1_points"][: g["map0_points"].shape[0]], rtol=1e-6,
                               atol=1e-7)
    # down-samplers (odometry/icputils.py)
    from gradslam_amd.odometry import icputils
    pc_ref = gs.Pointclouds(points=[T(g["map0_points"]).cuda()], normals=[T(g["map0_normals"]).cuda()])
    ds = icputils.downsample_pointclouds(pc_ref, act, 4)
    assert np.array_equal(host(ds.points_list[0]), g["ds4_map_points"])
    dsf = icputils.downsample_rgbdimages(f1, 4)
    assert np.array_equal(host(dsf.points_list[0]), g["ds4_frame_points"])
    assert np.array_equal(host(dsf.normals_list[0]), g["ds4_frame_normals"])


def test_fusion_kat_through_the_api(gs, golden):
    """Mirror of tests/slam/test_fusionutils.py:672-750 / :918-986 of the reference."""
    from gradslam_amd.slam import fusionutils as fu
    g = golden("fusion_kat")
    pcs = gs.Pointclouds(points=T(g["points"])[None].cuda(), normals=T(g["normals"])[None].cuda(),
                         colors=T(g["colors"])[None].cuda(), features=T(g["ccounts"])[None].cuda())
    fr = gs.RGBDImages(T(g["rgb"])[None, None].cuda(), T(g["depth"])[None, None].cuda(),
                       T(g["intrinsics"])[None, None].cuda(), T(g["pose"])[None, None].cuda())
    uq = fu.find_best_unique_correspondences(pcs, fr, T(g["rows"]).cuda())
    assert np.array_equal(host(uq), g["unique"])
    fused = fu.fuse_with_map(pcs.clone(), fr, uq, float(g["sigma"]))
    np.testing.assert_allclose(host(fused.points_list[0]), g["fused_points"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(host(fused.features_list[0]), g["fused_ccounts"], rtol=1e-6, atol=1e-9)
    fused0 = fu.fuse_with_map(pcs.clone(), fr, torch.empty((0, 4), dtype=torch.int64, device="cuda"), float(g["sigma"]))
    np.testing.assert_allclose(host(fused0.points_list[0]), g["fused0_points"], rtol=1e-6, atol=1e-6)
    a = fu.get_alpha(T(g["points"]).cuda()[None], float(g["sigma"]), keepdim=True)
    assert a.shape == (1, 8, 1)
    pts = T(g["points"]).cuda()
    close = fu.are_points_close(pts, pts + 0.01, 0.05)
    assert close.all() and not fu.are_points_close(pts, pts + 1.0, 0.05).any()
    nr = T(g["normals"]).cuda()
    assert fu.are_normals_similar(nr, nr, 0.9).all() and not fu.are_normals_similar(nr, -nr, 0.9).any()


def test_icputils_api(gs, golden):
    from gradslam_amd.odometry import icputils
    from gradslam_amd.odometry.icp import GradICPOdometryProvider, ICPOdometryProvider
    g = golden("icp_unit")
    src, tgt, tn = (T(g[k]).cuda() for k in ("src", "tgt", "tgt_normals"))
    A, b, idx = icputils.gauss_newton_solve(src[None], tgt[None], tn[None])
    assert np.array_equal(host(A), g["gn_A"]) and np.array_equal(host(b), g["gn_b"]) and np.array_equal(host(idx), g["gn_idx"])
    A2, b2, idx2 = icputils.gauss_newton_solve(src[None], tgt[None], tn[None], float(g["gn_thr"]))
    assert np.array_equal(host(A2), g["gn_thr_A"]) and np.array_equal(host(idx2), g["gn_thr_idx"])
    x = icputils.solve_linear_system(T(g["kat_A"]).cuda(), T(g["kat_b"]).cuda(), 1e-8)
    assert x.shape == (4, 1)
    np.testing.assert_allclose(g["kat_A"] @ host(x), g["kat_b"], rtol=1e-4, atol=1e-4)
    eye = torch.eye(4, device="cuda")
    Ti, _ = icputils.point_to_plane_ICP(src[None], tgt[None], tn[None], eye, numiters=20)
    Tg, _ = icputils.point_to_plane_gradICP(src[None], tgt[None], tn[None], eye, numiters=20)
    np.testing.assert_allclose(host(Ti), g["icp20_T"], atol=2e-5, rtol=0)
    np.testing.assert_allclose(host(Tg), g["gradicp20_T"], atol=2e-5, rtol=0)
    with pytest.raises(AttributeError):  # the reference's documented-but-broken None default (icputils.py:298)
        icputils.point_to_plane_ICP(src[None], tgt[None], tn[None], None)
    maps_pc, frames_pc = gs.Pointclouds(points=[tgt], normals=[tn]), gs.Pointclouds(points=[src])
    for prov, key in ((ICPOdometryProvider(20), "icp20_T"), (GradICPOdometryProvider(20), "gradicp20_T")):
        out = prov.provide(maps_pc, frames_pc)
        assert out.shape == (1, 1, 4, 4)
        np.testing.assert_allclose(host(out[0, 0]), g[key], atol=2e-5, rtol=0)


@pytest.mark.parametrize("key,cls,odom", [("pf_gradicp", "PointFusion", "gradicp"), ("pf_icp", "PointFusion", "icp"),
                                          ("pf_gt", "PointFusion", "gt"), ("icpslam_gradicp", "ICPSLAM", "gradicp")])
def test_slam_sequences_vs_reference_and_oracle(gs, golden, key, cls, odom):
    """Config C1 (64x64x3): poses within ATE 1e-4 m of the reference, same map size, coordinates
    within 1e-5; and against the oracle's loop on the same inputs."""
    g = golden("synth64")
    poses = g["poses"].copy()
    if odom != "gt":
        poses[1:] = poses[:1]
    frames = gs.RGBDImages(T(g["colors"][None]).cuda(), T(g["depths"][None]).cuda(),
                           T(g["intrinsics"][None, None]).cuda(), T(poses[None]).cuda())
    slam = getattr(gs.slam, cls)(odom=odom, device="cuda")
    pc, rp = slam(frames)
    rp = host(rp)[0]
    assert rp.shape == (3, 4, 4)
    assert ate(rp, g[key + "_poses"]) <= 1e-4
    np.testing.assert_allclose(rp, g[key + "_poses"], rtol=0, atol=2e-5)
    assert pc.points_list[0].shape[0] == g[key + "_points"].shape[0]
    np.testing.assert_allclose(host(pc.points_list[0]), g[key + "_points"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(host(pc.normals_list[0]), g[key + "_normals"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(host(pc.colors_list[0]), g[key + "_colors"], rtol=1e-4, atol=1e-2)
    m, op = oslam.run_sequence(g["colors"], g["depths"], g["intrinsics"], poses,
                               slam="pointfusion" if cls == "PointFusion" else "icpslam", odom=odom)
    np.testing.assert_allclose(rp, op, rtol=0, atol=2e-6)
    assert len(m) == pc.points_list[0].shape[0]
    if odom == "gt":  # no ICP in the loop: the whole map must be bit-identical to the oracle's
        assert np.array_equal(host(pc.points_list[0]), m.points)
        assert np.array_equal(host(pc.features_list[0]), m.ccounts)


def test_batch_of_two_sequences(gs):
    """B=2 on one GPU: sequences are independent; each must equal its own single-sequence run."""
    seqs = [make_sequence(3, 48, 64, seed=s) for s in (21, 22)]
    stack = lambda k: T(np.stack([s[k] for s in seqs])).cuda()  # noqa: E731
    poses = stack("poses")
    poses[:, 1:] = poses[:, :1]
    frames = gs.RGBDImages(stack("colors"), stack("depths"), stack("intrinsics")[:, None][:, :, 0], poses)
    pc, rp = gs.slam.PointFusion(odom="gradicp", device="cuda")(frames)
    assert len(pc) == 2 and rp.shape == (2, 3, 4, 4)
    for b, s in enumerate(seqs):
        p = s["poses"].copy()
        p[1:] = p[:1]
        m, op = oslam.run_sequence(s["colors"], s["depths"], s["intrinsics"][0], p)
        np.testing.assert_allclose(host(rp[b]), op, rtol=0, atol=2e-6)
        assert pc.points_list[b].shape[0] == len(m)
        np.testing.assert_allclose(host(pc.points_list[b]), m.points, rtol=1e-5, atol=1e-5)
    pad = pc.points_padded
    assert pad.shape[0] == 2 and pad.shape[1] == max(pc._n)
    assert bool((pad[pc.nonpad_mask.logical_not()] == 0).all())


def test_sequence_120_vs_reference(gs, golden):
    g = golden("synth120")
    s = make_sequence(4, 120, 160, seed=int(g["colors_seed"]))
    poses = g["poses"].copy()
    poses[1:] = poses[:1]
    frames = gs.RGBDImages(T(s["colors"][None]).cuda(), T(g["depths"][None]).cuda(),
                           T(g["intrinsics"][None, None]).cuda(), T(poses[None]).cuda())
    pc, rp = gs.slam.PointFusion(device="cuda")(frames)
    assert ate(host(rp)[0], g["pf_gradicp_poses"]) <= 1e-4
    assert pc.points_list[0].shape[0] == int(g["pf_gradicp_count"])
    n = pc.points_list[0].shape[0]  # mean coordinate difference below 1e-6 m
    np.testing.assert_allclose(host(pc.points_list[0]).astype(np.float64).sum(0), g["pf_gradicp_points_sum"], rtol=0,
                               atol=1e-6 * n)


def test_step_api_and_growth(gs):
    """ICPSLAM.step as documented (slam/icpslam.py:140-178) + geometric capacity growth."""
    s = make_sequence(6, 48, 64, seed=9)
    frames = gs.RGBDImages(T(s["colors"][None]).cuda(), T(s["depths"][None]).cuda(),
                           T(s["intrinsics"][None]).cuda(), T(s["poses"][None]).cuda())
    slam = gs.slam.PointFusion(odom="gt", device="cuda")
    pc = gs.Pointclouds(device="cuda")
    counts = []
    for t in range(6):
        pc, poses = slam.step(pc, frames[:, t], None, inplace=True)
        assert poses.shape == (1, 1, 4, 4)
        counts.append(pc.points_list[0].shape[0])
    assert all(b >= a for a, b in zip(counts, counts[1:])) and counts[-1] > counts[0]
    pc2 = pc.clone()
    assert torch.equal(pc2.points_padded, pc.points_padded) and pc2.points_list[0].data_ptr() != pc.points_list[0].data_ptr()
    assert pc.cpu().device.type == "cpu"


def test_map_is_not_reallocated_on_a_loose_count_bound(gs):
    """The host only keeps an upper bound of a device-side surfel count (it runs a few frames x H*W rows ahead).  When
    that bound reaches the capacity, `_reserve` first asks for the exact count: buffers with real headroom stay where
    they are (a reallocation copies the map and re-sizes every scratch that follows the capacity); without headroom for
    the frames the host runs ahead they grow."""
    s = make_sequence(5, 48, 64, seed=11)
    poses = s["poses"].copy()
    poses[1:] = poses[:1]
    frames = gs.RGBDImages(T(s["colors"][None]).cuda(), T(s["depths"][None]).cuda(),
                           T(s["intrinsics"][None]).cuda(), T(poses[None]).cuda())
    slam = gs.slam.PointFusion(odom="gradicp", device="cuda")
    pc, prev = gs.Pointclouds(device="cuda"), None
    for t in range(5):
        live = frames[:, t]
        pc, _ = slam.step(pc, live, prev, inplace=True)
        prev = live
    P = 48 * 64
    grp = pc._dcount[0].group            # the counts live on the device in the in-place loop
    exact = grp.resolve()[0]             # (reading points_list would make the host count exact and drop the group)
    buf = pc._buf["points"][0]
    cap = int(buf.shape[0])
    assert exact + 9 * P < cap           # 16 frames of room were reserved up front
    pc._dcount[0].group.bounds = [cap - P // 2]          # a bound that has (wrongly) run up to the capacity
    pc._reserve(0, P, pc.RESERVE_FRAMES)
    assert pc._buf["points"][0] is buf and pc._dcount[0].group.bounds == [exact]
    # no headroom for the frames ahead: grows (and keeps the rows)
    before = buf[:exact].clone()
    pc._dcount[0].group.bounds = [cap]
    pc._reserve(0, cap // 4, pc.RESERVE_FRAMES)
    assert pc._buf["points"][0] is not buf and pc._buf["points"][0].shape[0] >= 2 * cap
    assert torch.equal(pc._buf["points"][0][:exact], before) and pc.points_list[0].shape[0] == exact


def test_scannet_resolution_map_growth(gs):
    """Config C5 shape (1296x968, ScanNet-like): dynamic map growth across capacity doublings with
    gradICP odometry at 78k x ~80k ICP points per frame; poses must track the ground truth."""
    L, H, W = 6, 968, 1296
    s = make_sequence(L, H, W, seed=3)
    poses = s["poses"].copy()
    poses[1:] = poses[:1]
    frames = gs.RGBDImages(T(s["colors"][None]).cuda(), T(s["depths"][None]).cuda(),
                           T(s["intrinsics"][None]).cuda(), T(poses[None]).cuda())
    slam = gs.slam.PointFusion(odom="gradicp", device="cuda")
    pc, prev, counts = gs.Pointclouds(device="cuda"), None, []
    for t in range(L):
        live = frames[:, t]
        pc, _ = slam.step(pc, live, prev, inplace=True)
        prev = live
        counts.append(pc.points_list[0].shape[0])
    assert counts[0] > 1_000_000 and all(b > a for a, b in zip(counts, counts[1:]))
    assert pc._buf["points"][0].shape[0] >= counts[-1]          # capacity-backed store
    assert torch.isfinite(pc.points_list[0]).all() and bool((pc.features_list[0] > 0).all())
    rec = torch.stack([frames[:, 0].poses[0, 0]] + [prev.poses[0, 0]])  # first and last
    assert ate(host(prev.poses[0]), s["poses"][L - 1:L]) < 5e-3
    assert rec.shape == (2, 4, 4)


def test_groundtruth_provider_and_relative_pose(ops, golden):
    """gs_relative_pose_f32 == oracle (bit-exact: same double Gauss-Jordan), and within float32 ulps of the
    reference's GroundTruthOdometryProvider output."""
    import gradslam_amd as gs
    g = golden("gt_odom")
    T1, T2 = torch.from_numpy(g["T1"]).cuda(), torch.from_numpy(g["T2"]).cuda()
    rel = ops.relative_pose(T1, T2)
    assert np.array_equal(rel.cpu().numpy(), o.relative_pose(g["T1"], g["T2"]))
    B = T1.shape[0]
    mk = lambda T: gs.RGBDImages(torch.zeros(B, 1, 4, 4, 3).cuda(), torch.ones(B, 1, 4, 4, 1).cuda(),  # noqa: E731
                                 torch.eye(4).repeat(B, 1, 1, 1).cuda(), T.unsqueeze(1))
    out = gs.odometry.GroundTruthOdometryProvider().provide(mk(T1), mk(T2))
    assert out.shape == (B, 1, 4, 4) and out.is_cuda
    assert np.abs(out.cpu().numpy() - g["rel"]).max() <= 2e-6
    assert torch.equal(gs.geometry.geometryutils.relative_transformation(T1[0], T2[0]), rel[0])


def test_tum_loader_on_device_matches_reference(tmp_path, golden):
    """Our TUM loader (PNG -> pinned staging -> HIP ingest kernels) against the reference loader's items."""
    from gradslam_amd.datasets import TUM
    from tests import tum_fixture as fx
    root = fx.write(str(tmp_path))
    g = golden("tum_items")
    for case, kw in fx.CASES.items():
        ds = TUM(root, **kw)
        gold = {str(g["%s/%d/names" % (case, j)]): j for j in range(int(g[case + "/len"]))}
        for k in range(len(ds)):
            colors, depths, K, poses, transforms, names, stamps = ds[k]
            i = gold[names]   # sequences come in os.listdir order (as in the reference): match items by name
            assert colors.is_cuda and depths.is_cuda and poses.is_cuda
            assert np.array_equal(colors.cpu().numpy(), g["%s/%d/colors" % (case, i)])
            assert np.array_equal(depths.cpu().numpy(), g["%s/%d/depths" % (case, i)])
            assert np.array_equal(K.cpu().numpy(), g["%s/%d/intrinsics" % (case, i)])
            assert np.abs(poses.cpu().numpy() - g["%s/%d/poses" % (case, i)]).max() <= 2e-6
            assert np.abs(transforms.cpu().numpy() - g["%s/%d/transforms" % (case, i)]).max() <= 2e-6
            assert names == str(g["%s/%d/names" % (case, i)]) and stamps == str(g["%s/%d/stamps" % (case, i)])
    # the items feed the SLAM path directly
    import gradslam_amd as gsa
    ds = TUM(root, seqlen=3, height=fx.H, width=fx.W)
    colors, depths, K, poses, *_ = ds[0]
    frames = gsa.RGBDImages(colors[None], depths[None], K[None], poses[None])
    assert frames.vertex_map.shape == (1, 3, fx.H, fx.W, 3)


@pytest.mark.parametrize("size", [(24, 32), (48, 64), (17, 45), (10, 12), (480, 640)])
def test_ingest_kernels_match_oracle(ops, size):
    """gs_ingest_* against the oracle restatement of the OpenCV resize arithmetic, bit-exact, for
    same-size, up- and down-scaling (the resized cases are not pinned against OpenCV itself)."""
    rng = np.random.default_rng(5)
    raw_c = rng.integers(0, 256, (24, 32, 3), dtype=np.uint8)
    raw_d = rng.integers(0, 65535, (24, 32), dtype=np.uint16)
    H, W = size
    for norm in (False, True):
        out = ops.ingest_color(torch.from_numpy(raw_c).cuda(), H, W, norm)
        assert np.array_equal(out.cpu().numpy(), o.ingest_color(raw_c, H, W, norm))
    for div in (5000.0, 1000.0):
        out = ops.ingest_depth(torch.from_numpy(raw_d).cuda(), H, W, div)
        assert np.array_equal(out.cpu().numpy(), o.ingest_depth(raw_d, H, W, div))


def test_icl_loader_on_device_matches_reference(tmp_path, golden):
    from gradslam_amd.datasets import ICL
    from tests import tum_fixture as fx
    root = fx.write_icl(str(tmp_path))
    g = golden("icl_items")
    for case, kw in fx.ICL_CASES.items():
        ds = ICL(root, **kw)
        gold = {str(g["%s/%d/names" % (case, j)]): j for j in range(int(g[case + "/len"]))}
        for k in range(len(ds)):
            colors, depths, K, poses, transforms, names = ds[k]
            i = gold[names]
            assert np.array_equal(colors.cpu().numpy(), g["%s/%d/colors" % (case, i)])
            assert np.array_equal(depths.cpu().numpy(), g["%s/%d/depths" % (case, i)])
            assert np.array_equal(K.cpu().numpy(), g["%s/%d/intrinsics" % (case, i)])
            assert np.abs(poses.cpu().numpy() - g["%s/%d/poses" % (case, i)]).max() <= 2e-6
            assert np.abs(transforms.cpu().numpy() - g["%s/%d/transforms" % (case, i)]).max() <= 2e-6


def test_scannet_loader_on_device_matches_reference(tmp_path, golden):
    """gradslam_amd.datasets.Scannet against the items the reference's own loader returned for the same files
    (tests/golden/scannet_items.npz, oracle/make_golden_extra.py:scannet_items): colours, depths (scale 1000),
    intrinsics, names and labels (both palettes) identical; poses / transforms within 2e-6."""
    from gradslam_amd.datasets import Scannet
    from tests import tum_fixture as fx
    base, meta = fx.write_scannet(str(tmp_path))
    g = golden("scannet_items")
    for case, kw in fx.SCANNET_CASES.items():
        ds = Scannet(base, meta, **kw)
        assert len(ds) == int(g[case + "/len"])
        gold = {str(g["%s/%d/names" % (case, j)]): j for j in range(len(ds))}
        for k in range(len(ds)):
            colors, depths, K, poses, transforms, names, labels = ds[k]
            i = gold[names]
            assert np.array_equal(colors.cpu().numpy(), g["%s/%d/colors" % (case, i)])
            assert np.array_equal(depths.cpu().numpy(), g["%s/%d/depths" % (case, i)])
            assert np.array_equal(K.cpu().numpy(), g["%s/%d/intrinsics" % (case, i)])
            assert np.array_equal(labels.cpu().numpy(), g["%s/%d/labels" % (case, i)])
            assert np.abs(poses.cpu().numpy() - g["%s/%d/poses" % (case, i)]).max() <= 2e-6
            assert np.abs(transforms.cpu().numpy() - g["%s/%d/transforms" % (case, i)]).max() <= 2e-6
    with pytest.raises(ValueError):
        Scannet(base, meta, None, start=3, end=2)
    with pytest.raises(TypeError):
        Scannet(base, meta, 5)


def test_project_unproject_points_vs_reference_golden(golden):
    """geometry.projutils.project_points / unproject_points (HIP kernels gs_project_points_f32 / gs_unproject_points_f32)
    against the REAL reference on every broadcasting case of its docstrings (tests/golden/api_helpers.npz)."""
    from gradslam_amd.geometry import projutils as P
    g = golden("api_helpers")
    d = lambda k: torch.from_numpy(g[k]).cuda()   # noqa: E731
    for c in "abcd":
        out = P.project_points(d("pp_%s_cam" % c), d("pp_%s_proj" % c))
        assert out.shape == g["pp_%s_out" % c].shape
        np.testing.assert_allclose(out.cpu().numpy(), g["pp_%s_out" % c], rtol=2e-6, atol=2e-6, err_msg=c)
    for c in "abc":
        out = P.unproject_points(d("up_%s_pix" % c), d("up_%s_kinv" % c), d("up_%s_depth" % c))
        assert out.shape == g["up_%s_out" % c].shape
        np.testing.assert_allclose(out.cpu().numpy(), g["up_%s_out" % c], rtol=2e-6, atol=2e-5, err_msg=c)
    with pytest.raises(ValueError):
        P.project_points(d("pp_c_cam"), d("pp_c_proj")[:1].repeat(3, 1, 1))
    with pytest.raises(ValueError):
        P.unproject_points(d("up_a_pix"), d("up_a_kinv"), d("up_a_depth")[:5])


def test_so3_se3_helpers_vs_reference_golden(golden):
    from gradslam_amd.geometry import se3utils as S
    g = golden("api_helpers")
    for i, om in enumerate(g["omega"]):
        w = torch.from_numpy(om).cuda()
        assert torch.equal(S.so3_hat(w).cpu(), torch.from_numpy(g["so3_hat"][i]))
        np.testing.assert_allclose(S.so3_exp(w).cpu().numpy(), g["so3_exp"][i], rtol=0, atol=2e-6)
    for i, xi in enumerate(g["xi"]):
        assert torch.equal(S.se3_hat(torch.from_numpy(xi).cuda()).cpu(), torch.from_numpy(g["se3_hat"][i]))


def test_pointclouds_algebra_vs_reference_golden_gpu(golden):
    """offset_ / scale_ / rotate_ / transform_ / pinhole_projection_ and + - * / @ on the GPU (rigid ops and the
    projection through gs_transform_points_f32 / gs_project_points_f32) against the REAL reference."""
    from tests.test_host_api_cpu import check_pointclouds_algebra
    check_pointclouds_algebra(golden, "cuda", 2e-6)


def test_get_alpha_gradients_match_the_reference_formula(gs):
    """Stand-alone fusionutils.get_alpha on the autograd tape (gs_alpha_backward_f32).  The reference's own gradient
    check (tests/slam/test_fusionutils.py:56-75: these six points, sigma = 0.6 as a 0-d tensor, float64 gradcheck) is run
    at float32 tolerance -- the kernels compute in float32 -- and the analytic gradients are compared with PyTorch
    autograd through the reference's formula (slam/fusionutils.py:69-72) in float64."""
    from gradslam_amd.slam import fusionutils as fu
    pts = torch.tensor([[5.0, 5.0, 5.0], [3.0, 3.0, 3.0], [1.0, 2.0, 3.0], [3.0, 2.0, 1.0], [-1.0, 0.0, 1.0], [0.0, 0.0, 0.0]],
                       device="cuda", dtype=torch.float64)
    extra = torch.from_numpy(np.random.default_rng(0).uniform(-1.2, 1.2, (64, 3))).cuda()
    for p0, sg in ((pts, 0.6), (extra, 0.6), (extra, 1.3)):
        p = p0.clone().requires_grad_(True)
        s = torch.tensor(sg, device="cuda", dtype=torch.float64, requires_grad=True)
        w = torch.linspace(0.5, 1.5, p.shape[0], device="cuda", dtype=torch.float64)
        (fu.get_alpha(p, s) * w).sum().backward()
        p2, s2 = p0.clone().requires_grad_(True), torch.tensor(sg, device="cuda", dtype=torch.float64, requires_grad=True)
        ref = torch.clamp(torch.exp(-torch.sum(p2 ** 2, -1) / (2 * (s2 ** 2))), min=1e-7, max=1.01)
        (ref * w).sum().backward()
        assert torch.allclose(p.grad, p2.grad, rtol=2e-6, atol=1e-7), (p.grad - p2.grad).abs().max()
        assert abs(float(s.grad) - float(s2.grad)) <= 2e-6 * abs(float(s2.grad)) + 1e-7
    # dim / keepdim as the SLAM path calls it (slam/fusionutils.py:657), float32 inputs
    v = torch.rand(1, 1, 4, 5, 3, device="cuda", requires_grad=True)
    a = fu.get_alpha(v, 0.6, dim=4, keepdim=True)
    assert a.shape == (1, 1, 4, 5, 1) and a.requires_grad
    a.sum().backward()
    assert torch.allclose(v.grad, -(a.detach() * v.detach()) / 0.36, rtol=1e-5, atol=1e-7)
    # the reference's check itself, with a step and tolerances a float32 forward supports
    p = pts[1:].clone().requires_grad_(True)   # ([5, 5, 5] sits on the clamp: alpha = eps, zero gradient -- checked above)
    s = torch.tensor(0.6, device="cuda", dtype=torch.float64, requires_grad=True)
    assert torch.autograd.gradcheck(fu.get_alpha, (p, s), eps=1e-2, atol=2e-3, rtol=2e-2, nondet_tol=0.0, raise_exception=True)


def test_se3_exp_gradient_matches_autograd_of_the_reference_formula(gs):
    """Stand-alone geometry.se3utils.se3_exp on the autograd tape (gs_se3_exp_backward_f32) against PyTorch autograd
    through the reference's formula (geometry/se3utils.py:77-115) in float64, for a generic twist, a small one and the
    small-angle branch."""
    from gradslam_amd.geometry import se3utils

    def ref_exp(xi):
        v, w = xi[:3], xi[3:]
        z = torch.zeros((), dtype=xi.dtype, device=xi.device)
        wh = torch.stack([torch.stack([z, -w[2], w[1]]), torch.stack([w[2], z, -w[0]]), torch.stack([-w[1], w[0], z])])
        th = w.norm()
        eye = torch.eye(3, dtype=xi.dtype, device=xi.device)
        if float(th) < 1e-6:
            R = V = eye + wh
        else:
            A, B, C = torch.sin(th) / th, (1 - torch.cos(th)) / th ** 2, (th - torch.sin(th)) / th ** 3
            R, V = eye + A * wh + B * wh @ wh, eye + B * wh + C * wh @ wh
        return torch.cat([torch.cat([R, (V @ v)[:, None]], 1), torch.tensor([[0, 0, 0, 1.0]], dtype=xi.dtype, device=xi.device)])

    Wt = torch.from_numpy(np.random.default_rng(1).standard_normal((4, 4))).cuda()
    for vals in ([0.1, -0.2, 0.3, 0.2, 0.1, -0.3], [1e-3, 2e-3, -1e-3, 3e-3, -2e-3, 1e-3], [0.5, 0.1, 0.2, 1e-8, -2e-8, 1e-8]):
        xi = torch.tensor(vals, device="cuda", dtype=torch.float32, requires_grad=True)
        Tm = se3utils.se3_exp(xi.reshape(6, 1))
        assert Tm.requires_grad
        (Tm.double() * Wt).sum().backward()
        x2 = torch.tensor(vals, device="cuda", dtype=torch.float64, requires_grad=True)
        (ref_exp(x2) * Wt).sum().backward()
        assert torch.allclose(xi.grad.double(), x2.grad, rtol=1e-5, atol=1e-6), (xi.grad, x2.grad)


def test_metrics_map_chamfer_uses_exact_nearest_neighbours(gs):
    """gradslam_amd.metrics.map_chamfer (exact 1-NN through gs_knn1_grid_f32) against the oracle's brute force on two
    random clouds, on a shifted copy (every distance = the shift), and on the maps of two PointFusion runs."""
    from gradslam_amd import metrics as M
    rng = np.random.default_rng(5)
    a = rng.uniform(-1, 1, (3000, 3)).astype(np.float32)
    b = rng.uniform(-1, 1, (2500, 3)).astype(np.float32)
    r = M.map_chamfer(a, b)
    _, dab = o.knn1(a, b)
    _, dba = o.knn1(b, a)
    assert abs(r["a_to_b_rms_m"] - math.sqrt(dab.astype(np.float64).mean())) < 1e-7
    assert abs(r["b_to_a_max_m"] - math.sqrt(float(dba.max()))) < 1e-7
    assert abs(r["chamfer_m2"] - (dab.astype(np.float64).mean() + dba.astype(np.float64).mean())) < 1e-9
    grid = np.stack(np.meshgrid(*[np.arange(12, dtype=np.float32) * 0.01] * 3, indexing="ij"), -1).reshape(-1, 3)
    sh = M.map_chamfer(T(grid).cuda(), T(grid + np.float32(1e-3)).cuda())
    assert abs(sh["a_to_b_rms_m"] - math.sqrt(3) * 1e-3) < 1e-6 and abs(sh["b_to_a_max_m"] - math.sqrt(3) * 1e-3) < 1e-6
    s = make_sequence(3, 96, 128, seed=4)
    poses = T(s["poses"][None]).cuda().clone()
    poses[:, 1:] = poses[:, :1]
    fr = gs.RGBDImages(T(s["colors"][None]).cuda(), T(s["depths"][None]).cuda(), T(s["intrinsics"][None]).cuda(), poses)
    pc1, rp1 = gs.slam.PointFusion(odom="gradicp", device="cuda")(fr)
    pc2, rp2 = gs.slam.PointFusion(odom="icp", device="cuda")(fr)
    same = M.map_chamfer(pc1, pc1)
    assert same["chamfer_m2"] == 0.0 and same["points_a"] == pc1.points_list[0].shape[0]
    other = M.map_chamfer(pc1, pc2)
    assert 0.0 < other["a_to_b_rms_m"] < 5e-3 and M.ate_rmse(rp1[0], rp2[0]) < 5e-3
