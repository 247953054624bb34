"""CPU-only tests of the host layer: container logic of Pointclouds / RGBDImages and the
reference's error behaviour (type / shape checks run before any HIP call, with the reference's
messages: tests/slam/test_fusionutils.py:369-398, :543-669 style)."""
import os

import numpy as np
import pytest
import torch

import gradslam_amd as gs
from gradslam_amd._C import HipExtensionError
from gradslam_amd.odometry import icputils
from gradslam_amd.slam import fusionutils as fu
from oracle import oracle as o


def rgbd(B=1, L=2, H=4, W=5, poses=True):
    return gs.RGBDImages(torch.rand(B, L, H, W, 3), torch.rand(B, L, H, W, 1), torch.eye(4).repeat(B, 1, 1, 1),
                         torch.eye(4).repeat(B, L, 1, 1) if poses else None)


# ------------------------------------------------------------------ Pointclouds
def test_pointclouds_list_and_padded_construction():
    p = [torch.rand(5, 3), torch.rand(3, 3)]
    f = [torch.rand(5, 2), torch.rand(3, 2)]
    pc = gs.Pointclouds(points=p, normals=[x + 1 for x in p], features=f)
    assert len(pc) == 2 and pc.has_points and pc.has_normals and not pc.has_colors and pc.has_features
    assert pc.num_points_per_pointcloud.tolist() == [5, 3] and pc.equisized is False
    assert pc.points_padded.shape == (2, 5, 3) and torch.equal(pc.points_padded[1, :3], p[1])
    assert bool((pc.points_padded[1, 3:] == 0).all())
    assert pc.nonpad_mask.tolist() == [[True] * 5, [True] * 3 + [False] * 2]
    assert pc.features_padded.shape == (2, 5, 2)
    pad = gs.Pointclouds(points=torch.rand(2, 4, 3), colors=torch.rand(2, 4, 3))
    assert pad.equisized and [t.shape[0] for t in pad.points_list] == [4, 4]
    one = gs.Pointclouds(points=[p[0]])
    assert one.points_padded.data_ptr() == one.points_list[0].data_ptr()  # zero-copy for one sequence
    empty = gs.Pointclouds()
    assert len(empty) == 0 and not empty.has_points and empty.points_list is None and empty.device.type == "cpu"


def test_pointclouds_constructor_errors():
    with pytest.raises(TypeError, match="Expected points to be of type list or tensor or None"):
        gs.Pointclouds(points=3)
    with pytest.raises(TypeError, match="Expected normals to be of same type as points"):
        gs.Pointclouds(points=[torch.rand(2, 3)], normals=torch.rand(1, 2, 3))
    with pytest.raises(ValueError, match="should be > 0"):
        gs.Pointclouds(points=[])
    with pytest.raises(ValueError, match="ndim of all tensors in points list should be 2"):
        gs.Pointclouds(points=[torch.rand(3)])
    with pytest.raises(ValueError, match="last dim of all tensors in points should have shape 3"):
        gs.Pointclouds(points=[torch.rand(3, 4)])
    with pytest.raises(ValueError, match="normals tensors should have same shape"):
        gs.Pointclouds(points=[torch.rand(3, 3)], normals=[torch.rand(2, 3)])
    with pytest.raises(ValueError, match="number of features per pointcloud has to be equal"):
        gs.Pointclouds(points=[torch.rand(3, 3)], features=[torch.rand(2, 1)])
    with pytest.raises(ValueError, match="points should have ndim=3"):
        gs.Pointclouds(points=torch.rand(3, 3))
    with pytest.raises(ValueError, match="first 2 dims of features tensor"):
        gs.Pointclouds(points=torch.rand(1, 3, 3), features=torch.rand(1, 2, 1))


def test_append_clone_getitem_and_growth():
    a = gs.Pointclouds(points=[torch.rand(4, 3), torch.rand(2, 3)], colors=[torch.rand(4, 3), torch.rand(2, 3)])
    b = gs.Pointclouds(points=[torch.rand(1, 3), torch.rand(6, 3)], colors=[torch.rand(1, 3), torch.rand(6, 3)])
    before = [t.clone() for t in a.points_list]
    c = a.clone()
    a.append_points(b)
    assert a.num_points_per_pointcloud.tolist() == [5, 8]
    assert torch.equal(a.points_list[1][:2], before[1]) and torch.equal(a.points_list[1][2:], b.points_list[1])
    assert c.num_points_per_pointcloud.tolist() == [4, 2]  # the clone is independent
    assert a.points_padded.shape == (2, 8, 3)
    for _ in range(12):  # geometric growth keeps the prefix intact
        a.append_points(b)
    assert a.num_points_per_pointcloud.tolist() == [17, 80] and torch.equal(a.points_list[0][:4], before[0])
    sub = a[1]
    assert len(sub) == 1 and sub.points_list[0].shape[0] == 80
    e = gs.Pointclouds()
    e.append_points(b)
    assert e.num_points_per_pointcloud.tolist() == [1, 6] and e.has_colors
    with pytest.raises(TypeError, match="Append object must be of type gradslam.Pointclouds"):
        a.append_points(torch.rand(2, 3))
    with pytest.raises(ValueError, match="Batch size of pointclouds to append"):
        a.append_points(gs.Pointclouds(points=[torch.rand(1, 3)], colors=[torch.rand(1, 3)]))
    with pytest.raises(ValueError, match="must either both have or not have normals"):
        a.append_points(gs.Pointclouds(points=[torch.rand(1, 3)] * 2, colors=[torch.rand(1, 3)] * 2,
                                       normals=[torch.rand(1, 3)] * 2))


def test_padded_setters_and_rigid_helpers():
    pc = gs.Pointclouds(points=[torch.rand(4, 3), torch.rand(2, 3)], normals=[torch.rand(4, 3), torch.rand(2, 3)])
    new = torch.zeros(2, 4, 3)
    new[0] = 1.0
    new[1, :2] = 2.0
    pc.points_padded = new
    assert torch.equal(pc.points_list[1], torch.full((2, 3), 2.0))
    with pytest.raises(ValueError, match="value must have shape"):
        pc.points_padded = torch.zeros(2, 5, 3)
    T = torch.eye(4)
    T[:3, 3] = torch.tensor([1.0, 2.0, 3.0])
    moved = pc.transform(T)
    assert torch.allclose(moved.points_list[0], torch.ones(4, 3) + T[:3, 3]) and torch.equal(pc.points_list[0], torch.ones(4, 3))
    K = torch.eye(4)
    K[0, 0] = K[1, 1] = 2.0
    proj = gs.Pointclouds(points=[torch.tensor([[1.0, 2.0, 4.0]])]).pinhole_projection(K)
    assert torch.allclose(proj.points_list[0], torch.tensor([[0.5, 1.0, 1.0]]))
    with pytest.raises(ValueError, match="transform should be of shape"):
        pc.transform_(torch.eye(3))


# ------------------------------------------------------------------ RGBDImages
def test_rgbdimages_container():
    r = rgbd(B=2, L=3)
    assert r.shape == (2, 3, 4, 5) and len(r) == 2 and r.has_poses and r.cdim == 4
    s = r[:, 1]
    assert s.shape == (2, 1, 4, 5) and torch.equal(s.depth_image, r.depth_image[:, 1:2]) and s.poses.shape == (2, 1, 4, 4)
    assert r[1].shape == (1, 3, 4, 5) and r[0, 2].shape == (1, 1, 4, 5)
    with pytest.raises(IndexError):
        r[:, 5]
    cf = r.to_channels_first()
    assert cf.channels_first and cf.rgb_image.shape == (2, 3, 3, 4, 5) and cf.cdim == 2
    assert torch.equal(cf.to_channels_last().rgb_image, r.rgb_image)
    assert torch.equal(r.valid_depth_mask, r.depth_image > 0)
    r.poses = torch.eye(4).repeat(2, 3, 1, 1) * 2
    with pytest.raises(ValueError, match="value must have shape"):
        r.poses = torch.eye(4)
    c = r.clone()
    c.depth_image.zero_()
    assert r.depth_image.abs().sum() > 0
    # detach / clone / to (structures/rgbdimages.py:465-525 of the reference): clone copies every tensor, detach shares
    # storage and leaves the tape, to(same device) is the object itself, to(..., copy=True) a deep copy
    g = rgbd(B=1, L=2)
    g.depth_image = g.depth_image.clone().requires_grad_(True)
    d = g.detach()
    assert not d.depth_image.requires_grad and d.depth_image.data_ptr() == g.depth_image.data_ptr()
    assert d.poses.data_ptr() == g.poses.data_ptr() and d.shape == g.shape and d.channels_first == g.channels_first
    k = g.clone()
    assert k.depth_image.requires_grad and k.rgb_image.data_ptr() != g.rgb_image.data_ptr() and torch.equal(k.poses, g.poses)
    assert g.to("cpu") is g and g.cpu() is g
    t = g.to("cpu", copy=True)
    assert t is not g and t.intrinsics.data_ptr() != g.intrinsics.data_ptr() and torch.equal(t.intrinsics, g.intrinsics)


def test_rgbdimages_constructor_errors():
    rgb, d, K = torch.rand(1, 1, 4, 5, 3), torch.rand(1, 1, 4, 5, 1), torch.eye(4).view(1, 1, 4, 4)
    with pytest.raises(TypeError, match="Expected rgb_image to be of type tensor"):
        gs.RGBDImages(None, d, K)
    with pytest.raises(TypeError, match="Expected channels_first to be of type bool"):
        gs.RGBDImages(rgb, d, K, channels_first=1)
    with pytest.raises(ValueError, match="rgb_image should have ndim=5"):
        gs.RGBDImages(rgb[0], d, K)
    with pytest.raises(ValueError, match="Expected rgb_image to have 3 channels"):
        gs.RGBDImages(torch.rand(1, 1, 4, 5, 4), d, K)
    with pytest.raises(ValueError, match="Expected depth_image to have shape"):
        gs.RGBDImages(rgb, torch.rand(1, 1, 4, 4, 1), K)
    with pytest.raises(ValueError, match="Expected intrinsics to have shape"):
        gs.RGBDImages(rgb, d, torch.eye(4).view(1, 4, 4, 1))
    with pytest.raises(ValueError, match="Expected poses to have shape"):
        gs.RGBDImages(rgb, d, K, torch.eye(4).repeat(1, 2, 1, 1))


def test_lazy_maps_need_the_gpu_and_say_so():
    with pytest.raises(HipExtensionError, match="no CPU fallback"):
        rgbd().vertex_map


# ------------------------------------------------------------------ function-level error behaviour
def test_fusionutils_validation_messages():
    pc, fr = gs.Pointclouds(points=[torch.rand(3, 3)]), rgbd(L=1)
    tab = torch.zeros((2, 4), dtype=torch.int64)
    with pytest.raises(TypeError, match="Expected pointclouds to be of type gradslam.Pointclouds"):
        fu.find_active_map_points(torch.rand(3), fr)
    with pytest.raises(TypeError, match="Expected rgbdimages to be of type gradslam.RGBDImages"):
        fu.find_active_map_points(pc, torch.rand(3))
    with pytest.raises(ValueError, match="Expected rgbdimages to have sequence length of 1"):
        fu.find_active_map_points(pc, rgbd(L=2))
    with pytest.raises(ValueError, match="Expected equal batch sizes"):
        fu.find_active_map_points(pc, rgbd(B=2, L=1))
    with pytest.raises(TypeError, match="Expected input pc2im_bnhw to have dtype of torch.int64"):
        fu.find_similar_map_points(pc, fr, tab.int(), 0.1, 0.9)
    with pytest.raises(ValueError, match=r"Expected pc2im_bnhw.shape\[1\] to be 4"):
        fu.find_similar_map_points(pc, fr, tab[:, :3], 0.1, 0.9)
    with pytest.raises(ValueError, match="Pointclouds must have normals for finding similar map points"):
        fu.find_similar_map_points(pc, fr, tab, 0.1, 0.9)
    with pytest.raises(ValueError, match="Pointclouds must have features for finding best unique"):
        fu.find_best_unique_correspondences(pc, fr, tab)
    with pytest.raises(ValueError, match="Pointclouds must have normals for map fusion"):
        fu.fuse_with_map(pc, fr, tab, 0.6)
    with pytest.raises(TypeError, match="Expected input sigma to be of type torch.Tensor or float or int"):
        fu.get_alpha(torch.rand(4, 3), "x")
    with pytest.raises(ValueError, match="Expected length of dim-th"):
        fu.get_alpha(torch.rand(4, 2), 0.6)
    with pytest.raises(ValueError, match="tensor1 and tensor2 should have the same shape"):
        fu.are_points_close(torch.rand(4, 3), torch.rand(5, 3), 0.1)
    # empty inputs return empty tables without touching the GPU (fusionutils.py:237-238, :365-368, :475-476)
    e = gs.Pointclouds()
    assert fu.find_active_map_points(e, fr).shape == (0, 4)
    sim, mask = fu.find_similar_map_points(e, fr, tab, 0.1, 0.9)
    assert sim.shape == (0, 4) and mask.shape == (0,) and mask.dtype == torch.bool
    assert fu.find_best_unique_correspondences(e, fr, tab).dtype == torch.int64


def test_icputils_and_slam_validation_messages():
    a = torch.rand(1, 5, 3)
    with pytest.raises(TypeError, match="Expected A to be of type torch.Tensor"):
        icputils.solve_linear_system(None, torch.rand(3, 1))
    with pytest.raises(ValueError, match=r"b.shape\[1\] should 1"):
        icputils.solve_linear_system(torch.rand(3, 6), torch.rand(3, 2))
    with pytest.raises(ValueError, match="src_pc should have ndim=3"):
        icputils.gauss_newton_solve(a[0], a, a)
    with pytest.raises(ValueError, match=r"tgt_pc.shape\[1\] and tgt_normals.shape\[1\] must be equal"):
        icputils.gauss_newton_solve(a, a, torch.rand(1, 4, 3))
    with pytest.raises(TypeError, match="Expected numiters to be of type int"):
        icputils.point_to_plane_ICP(a, a, a, torch.eye(4), numiters=2.0)
    with pytest.raises(ValueError, match="Expected initial_transform.shape to be"):
        icputils.point_to_plane_gradICP(a, a, a, torch.eye(3))
    with pytest.raises(TypeError, match="Expected lambda_max to be of type float or int"):
        icputils.point_to_plane_gradICP(a, a, a, torch.eye(4), lambda_max="2")
    with pytest.raises(TypeError, match="Expected ds_ratio to be of type int"):
        icputils.downsample_rgbdimages(rgbd(L=1), 2.0)
    with pytest.raises(ValueError, match="odometry method"):
        gs.slam.ICPSLAM(odom="orb")
    with pytest.raises(TypeError, match="Distance threshold must be of type float or int"):
        gs.slam.PointFusion(dist_th="a")
    slam = gs.slam.PointFusion(odom="gradicp")
    assert abs(slam.dot_th - 0.9396926207859084) < 1e-12 and slam.dsratio == 4 and slam.device.type == "cpu"
    with pytest.raises(TypeError, match="Expected frames to be of type gradslam.RGBDImages"):
        slam(torch.rand(3))
    with pytest.raises(ValueError, match="`live_frame` must have poses"):
        slam.step(gs.Pointclouds(), rgbd(L=1, poses=False), None)
    from gradslam_amd.odometry.icp import ICPOdometryProvider
    with pytest.raises(ValueError, match="maps_pointclouds missing normals"):
        ICPOdometryProvider().provide(gs.Pointclouds(points=[a[0]]), gs.Pointclouds(points=[a[0]]))


def test_groundtruth_provider_validation():
    """GroundTruthOdometryProvider rejects bad inputs with the reference's exception types and messages
    (odometry/groundtruth.py:30-72) before anything touches the device."""
    import torch
    import gradslam_amd as gs
    from gradslam_amd.odometry import GroundTruthOdometryProvider, OdometryProvider
    prov = GroundTruthOdometryProvider()
    assert isinstance(prov, OdometryProvider)
    mk = lambda B, L, poses=True: gs.RGBDImages(torch.zeros(B, L, 4, 4, 3), torch.ones(B, L, 4, 4, 1),  # noqa: E731
                                                torch.eye(4).repeat(B, 1, 1, 1),
                                                torch.eye(4).repeat(B, L, 1, 1) if poses else None)
    with pytest.raises(TypeError, match="Expected input 1"):
        prov.provide(torch.eye(4), mk(1, 1))
    with pytest.raises(TypeError, match="Expected input 2"):
        prov.provide(mk(1, 1), None)
    with pytest.raises(ValueError, match="Input 1 .* missing poses"):
        prov.provide(mk(1, 1, False), mk(1, 1))
    with pytest.raises(ValueError, match="Input 2 .* missing poses"):
        prov.provide(mk(1, 1), mk(1, 1, False))
    with pytest.raises(ValueError, match="Sequence length of rgbdimages1 must be 1"):
        prov.provide(mk(1, 2), mk(1, 1))
    with pytest.raises(ValueError, match="Batch size of rgbdimages1 and rgbdimages2 should be equal"):
        prov.provide(mk(2, 1), mk(1, 1))
    with pytest.raises(TypeError, match="trans_01"):
        gs.geometry.geometryutils.relative_transformation(None, torch.eye(4))
    with pytest.raises(ValueError, match="dims must match"):
        gs.geometry.geometryutils.relative_transformation(torch.eye(4), torch.eye(4)[None])


# ------------------------------------------------------------------ dataset loader (host logic)
def _tum(tmp_path):
    from tests import tum_fixture
    return tum_fixture, tum_fixture.write(str(tmp_path))


def test_tum_loader_host_logic_matches_reference(tmp_path, golden):
    """Frame association, sequence extraction, names, time stamps and intrinsics of our TUM loader against
    what the reference's loader returned on the same files (tests/golden/tum_items.npz)."""
    import torch
    from gradslam_amd.datasets import TUM
    fx, root = _tum(tmp_path)
    g = golden("tum_items")
    for case, kw in fx.CASES.items():
        ds = TUM(root, device="cpu", **kw)
        assert len(ds) == int(g[case + "/len"])
        gold = {str(g["%s/%d/names" % (case, j)]): j for j in range(len(ds))}
        assert sorted(gold) == sorted(ds.framenames)
        for k in range(len(ds)):
            i = gold[ds.framenames[k]]   # sequences come in os.listdir order (as in the reference)
            stamps = "\n".join("rgb {} depth {} pose {}".format(*t) for t in ds.timestamps[k])
            assert stamps == str(g["%s/%d/stamps" % (case, i)])
            assert torch.equal(ds.intrinsics, torch.from_numpy(g["%s/%d/intrinsics" % (case, i)]))
            # poses: host quaternion conversion + the oracle's relative pose == reference within float32 ulps
            P = np.stack(ds._homogenPoses(ds.poses[k]))
            rel = o.relative_pose(np.repeat(P[:1], len(P), 0), P)
            assert np.abs(rel - g["%s/%d/poses" % (case, i)]).max() <= 2e-6
            tr = np.concatenate([np.eye(4, dtype=np.float32)[None], o.relative_pose(P[:-1], P[1:])])
            assert np.abs(tr - g["%s/%d/transforms" % (case, i)]).max() <= 2e-6


def test_tum_ingest_oracle_matches_reference_pixels(tmp_path, golden):
    """The oracle restatement of the ingest stage reproduces the reference loader's colour and depth
    tensors exactly (native-size frames; resized frames are not pinned: OpenCV is not available here)."""
    from PIL import Image
    from gradslam_amd.datasets import TUM
    fx, root = _tum(tmp_path)
    g = golden("tum_items")
    for case, kw in fx.CASES.items():
        ds = TUM(root, device="cpu", **kw)
        gold = {str(g["%s/%d/names" % (case, j)]): j for j in range(len(ds))}
        for k in range(len(ds)):
            i = gold[ds.framenames[k]]
            col = np.stack([o.ingest_color(np.asarray(Image.open(p)), fx.H, fx.W, kw.get("normalize_color", False))
                            for p in ds.colorfiles[k]])
            dep = np.stack([o.ingest_depth(np.asarray(Image.open(p)), fx.H, fx.W, 5000.0)[..., None]
                            for p in ds.depthfiles[k]])
            if kw.get("channels_first"):
                col, dep = col.transpose(0, 3, 1, 2), dep.transpose(0, 3, 1, 2)
            assert np.array_equal(col, g["%s/%d/colors" % (case, i)])
            assert np.array_equal(dep, g["%s/%d/depths" % (case, i)])


def test_tum_loader_validation_and_no_cpu_fallback(tmp_path):
    from gradslam_amd import _C
    from gradslam_amd.datasets import TUM
    fx, root = _tum(tmp_path)
    with pytest.raises(TypeError, match='"seqlen" must be int'):
        TUM(root, seqlen=2.0)
    with pytest.raises(ValueError, match='"end" .* must be None or greater than start'):
        TUM(root, start=4, end=2)
    with pytest.raises(ValueError, match="sequences not available in basedir"):
        TUM(root, sequences=("rgbd_dataset_freiburg1_alpha", "rgbd_dataset_freiburg3_missing"))
    with pytest.raises(ValueError, match="Incorrect folder structure in basedir"):
        TUM(root, sequences=("rgbd_dataset_freiburg3_missing",))
    with pytest.raises(TypeError, match='"sequences" should either be path'):
        TUM(root, sequences=["rgbd_dataset_freiburg1_alpha"])
    os.makedirs(os.path.join(root, "not_a_tum_dir"))
    with pytest.raises(ValueError, match="Incorrect folder names"):
        TUM(root)
    os.rmdir(os.path.join(root, "not_a_tum_dir"))
    ds = TUM(root, device="cpu", **fx.CASES["default"])
    with pytest.raises(_C.HipExtensionError):      # the ingest stage is a HIP kernel: no CPU fallback
        ds[0]


def test_icl_loader_host_logic_matches_reference(tmp_path, golden):
    """File discovery, the traj0 special case, sequence extraction, names, intrinsics (fy < 0) and poses of
    our ICL loader against the reference loader's items (tests/golden/icl_items.npz)."""
    import torch
    from PIL import Image
    from gradslam_amd.datasets import ICL
    from tests import tum_fixture as fx
    root = fx.write_icl(str(tmp_path))
    g = golden("icl_items")
    for case, kw in fx.ICL_CASES.items():
        ds = ICL(root, device="cpu", **kw)
        assert len(ds) == int(g[case + "/len"])
        gold = {str(g["%s/%d/names" % (case, j)]): j for j in range(len(ds))}
        assert sorted(gold) == sorted(ds.framenames)
        for k in range(len(ds)):
            i = gold[ds.framenames[k]]
            assert torch.equal(ds.intrinsics, torch.from_numpy(g["%s/%d/intrinsics" % (case, i)]))
            P = np.stack(ds._loadPoses(ds.posemetas[k]["file"], ds.posemetas[k]["line_nums"]))
            rel = o.relative_pose(np.repeat(P[:1], len(P), 0), P)
            assert np.abs(rel - g["%s/%d/poses" % (case, i)]).max() <= 2e-6
            tr = np.concatenate([np.eye(4, dtype=np.float32)[None], o.relative_pose(P[:-1], P[1:])])
            assert np.abs(tr - g["%s/%d/transforms" % (case, i)]).max() <= 2e-6
            col = np.stack([o.ingest_color(np.asarray(Image.open(p)), fx.H, fx.W, kw.get("normalize_color", False))
                            for p in ds.colorfiles[k]])
            dep = np.stack([o.ingest_depth(np.asarray(Image.open(p)), fx.H, fx.W, 5000.0)[..., None]
                            for p in ds.depthfiles[k]])
            if kw.get("channels_first"):
                col, dep = col.transpose(0, 3, 1, 2), dep.transpose(0, 3, 1, 2)
            assert np.array_equal(col, g["%s/%d/colors" % (case, i)])
            assert np.array_equal(dep, g["%s/%d/depths" % (case, i)])
    with pytest.raises(ValueError, match="should only contain trajectory folder names"):
        ICL(root, trajectories=("kitchen",))
    with pytest.raises(TypeError, match="seqlen must be int"):
        ICL(root, seqlen="4")


def test_surfel_store_capacity_policy():
    """_reserve: geometric growth; only the SLAM drivers ask for several frames of room up front."""
    pc = gs.Pointclouds(device="cpu")
    pc._init_empty_batch(1, 1)
    P, N, C, F = pc._reserve(0, 1000)
    assert P.shape == (1024, 3) and F.shape == (1024, 1)          # plain append: no look-ahead
    pc._set_count(0, 1000)
    assert pc._reserve(0, 100)[0].shape[0] == 2048                 # doubling
    pc2 = gs.Pointclouds(device="cpu")
    pc2._init_empty_batch(1, 1)
    assert pc2._reserve(0, 1000, pc2.RESERVE_FRAMES)[0].shape[0] == 16000
    big = gs.Pointclouds(points=[torch.zeros(5000, 3)], normals=[torch.zeros(5000, 3)])
    base = gs.Pointclouds(points=[torch.zeros(10, 3)], normals=[torch.zeros(10, 3)])
    base.append_points(big)
    assert base.points_list[0].shape[0] == 5010 and base._buf["points"][0].shape[0] < 3 * 5010


def test_pointclouds_operators_and_list_setters():
    """reference: structures/pointclouds.py:300-384 (operators), :824-878 (list setters)."""
    import gradslam_amd as gs
    torch.manual_seed(0)
    pts = [torch.rand(5, 3), torch.rand(7, 3)]
    nrm = [torch.rand(5, 3), torch.rand(7, 3)]
    pc = gs.Pointclouds(points=[p.clone() for p in pts], normals=[n.clone() for n in nrm])
    assert torch.allclose((pc + 2.0).points_list[1], pts[1] + 2.0)
    assert torch.allclose((pc - 0.5).points_list[0], pts[0] - 0.5)
    assert torch.allclose((pc * 3).points_list[0], pts[0] * 3)
    assert torch.allclose((pc / 4.0).points_list[1], pts[1] / 4.0)
    assert torch.equal(pc.points_list[0], pts[0])            # out of place
    R = torch.linalg.qr(torch.rand(3, 3))[0]
    rot = pc @ R
    assert torch.allclose(rot.points_list[0], pts[0] @ R, atol=1e-6) and torch.allclose(rot.normals_list[1], nrm[1] @ R, atol=1e-6)
    T = torch.eye(4)
    T[:3, :3], T[:3, 3] = R, torch.tensor([0.1, -0.2, 0.3])
    tr = pc @ T
    assert torch.allclose(tr.points_list[1], pts[1] @ R + T[:3, 3], atol=1e-6)
    with pytest.raises(NotImplementedError):
        pc + "a"
    with pytest.raises(NotImplementedError):
        pc @ 3
    with pytest.raises(ValueError):
        pc @ torch.rand(2, 2)
    new = [torch.rand(5, 3), torch.rand(7, 3)]
    pc.points_list = new
    assert torch.equal(pc.points_list[1], new[1]) and pc.points_list[1].data_ptr() != new[1].data_ptr()
    assert torch.equal(pc.points_padded[1, :7], new[1])
    pc.features_list = [torch.rand(5, 4), torch.rand(7, 4)]
    assert pc.features_list[0].shape == (5, 4)
    with pytest.raises(TypeError):
        pc.points_list = torch.rand(2, 5, 3)
    with pytest.raises(ValueError):
        pc.points_list = [torch.rand(5, 3)]
    with pytest.raises(ValueError):
        pc.normals_list = [torch.rand(5, 3), torch.rand(6, 3)]
    with pytest.raises(ValueError):
        gs.Pointclouds().points_list = [torch.rand(5, 3)]


def test_rgbdimages_custom_pixel_pos_is_rejected_unless_regular():
    import gradslam_amd as gs
    B, L, H, W = 1, 1, 4, 5
    rgb, d, K = torch.rand(B, L, H, W, 3), torch.rand(B, L, H, W, 1), torch.eye(4).view(1, 1, 4, 4)
    v, u = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    grid = torch.stack([u, v, torch.ones_like(u)], -1).view(1, 1, H, W, 3)
    fr = gs.RGBDImages(rgb, d, K, pixel_pos=grid)
    assert torch.equal(fr.pixel_pos, grid)
    assert gs.RGBDImages(rgb, d, K).pixel_pos is None          # None until the vertex map exists (reference)
    with pytest.raises(NotImplementedError):
        gs.RGBDImages(rgb, d, K, pixel_pos=grid + 0.5)


def test_scannet_label_palettes():
    """nyu40 -> scannet20 remap and the colour palettes (reference: datasets/scannet.py:410-531)."""
    from gradslam_amd.datasets.scannet import get_color_encoding, nyu40_to_scannet20
    lab = np.arange(41, dtype=np.uint8).reshape(1, 41)
    out = nyu40_to_scannet20(lab.copy())[0]
    keep = {14: 13, 16: 14, 24: 15, 28: 16, 33: 17, 34: 18, 36: 19, 39: 20}
    for src in range(41):
        want = src if src <= 12 else keep.get(src, 0)
        assert out[src] == want, src
    assert list(get_color_encoding("scannet20"))[13:16] == ["desk", "curtain", "refrigerator"]
    assert len(get_color_encoding("nyu40")) == 41 and len(get_color_encoding("scannet20")) == 21


def test_setters_keep_the_capacity_of_the_store():
    """the list / padded setters used to replace the capacity-backed buffers by exact-size clones; the
    in-place kernels size every attribute by the points' capacity, so the next append would overrun them."""
    pc = gs.Pointclouds(points=[torch.rand(5, 3)], normals=[torch.rand(5, 3)], colors=[torch.rand(5, 3)],
                        features=[torch.rand(5, 1)])
    pc._reserve(0, 1000)
    new_n, new_f = torch.rand(5, 3), torch.rand(5, 1)
    pc.normals_list = [new_n]
    pc.features_list = [new_f]
    pc.colors_padded = torch.rand(1, 5, 3)
    caps = [pc._buf[k][0].shape[0] for k in ("points", "normals", "colors", "features")]
    assert len(set(caps)) == 1 and caps[0] >= 1005
    assert torch.equal(pc.normals_list[0], new_n) and torch.equal(pc.features_list[0], new_f)
    assert all(t.shape[0] >= 1005 for t in pc._reserve(0, 1000))
    # a short attribute buffer (set behind the store's back) is grown by _reserve too
    pc._buf["colors"][0] = pc._buf["colors"][0][:5].clone()
    assert all(t.shape[0] >= 1005 for t in pc._reserve(0, 1000))


def test_pixel_pos_channels_first_standard_grid_is_accepted():
    B, L, H, W = 1, 2, 3, 3
    v, u = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    grid = torch.stack([u, v, torch.ones_like(u)], 0).expand(B, L, 3, H, W).contiguous()
    rgb, depth, K = torch.rand(B, L, 3, H, W), torch.rand(B, L, 1, H, W), torch.eye(4).view(1, 1, 4, 4)
    gs.RGBDImages(rgb, depth, K, channels_first=True, pixel_pos=grid)
    with pytest.raises(NotImplementedError):
        gs.RGBDImages(rgb, depth, K, channels_first=True, pixel_pos=grid + 0.5)


_PC_CASES = {
    "offset_scalar": lambda pc, a: pc.offset_(0.25), "offset_vec": lambda pc, a: pc.offset_(a["off3"]),
    "offset_batch": lambda pc, a: pc.offset_(a["offB"]), "scale_scalar": lambda pc, a: pc.scale_(1.5),
    "scale_vec": lambda pc, a: pc.scale_(a["off3"]), "rotate_1_pre": lambda pc, a: pc.rotate_(a["R1"]),
    "rotate_1_post": lambda pc, a: pc.rotate_(a["R1"], pre_multiplication=False),
    "rotate_B_pre": lambda pc, a: pc.rotate_(a["RB"]), "transform_1_pre": lambda pc, a: pc.transform_(a["T1"]),
    "transform_B_pre": lambda pc, a: pc.transform_(a["TB"]),
    "transform_B_post": lambda pc, a: pc.transform_(a["TB"], pre_multiplication=False),
    "pinhole": lambda pc, a: pc.pinhole_projection_(a["K"]), "op_add": lambda pc, a: pc + 0.25,
    "op_sub": lambda pc, a: pc - a["off3"], "op_mul": lambda pc, a: pc * 1.5, "op_div": lambda pc, a: pc / 2.0,
    "op_matmul_R": lambda pc, a: pc @ a["R1"], "op_matmul_T": lambda pc, a: pc @ a["TB"],
}


def check_pointclouds_algebra(golden, device, tol):
    """Pointclouds.offset_ / scale_ / rotate_ / transform_ / pinhole_projection_ and + - * / @ against the REAL reference
    (tests/golden/api_helpers.npz, oracle/make_golden_api.py); shared by the CPU and the GPU test."""
    g = golden("api_helpers")
    T = lambda k: torch.from_numpy(g[k]).to(device)   # noqa: E731
    args = {k: T("arg_" + k) for k in ("R1", "RB", "T1", "TB", "K", "off3", "offB")}
    for tag, fn in _PC_CASES.items():
        pc = gs.Pointclouds(points=[T("pc_points0"), T("pc_points1")], normals=[T("pc_normals0"), T("pc_normals1")])
        out = fn(pc, args)
        for b in range(2):
            np.testing.assert_allclose(out.points_list[b].cpu().numpy(), g["%s_p%d" % (tag, b)], rtol=tol, atol=tol,
                                       err_msg=tag)
            np.testing.assert_allclose(out.normals_list[b].cpu().numpy(), g["%s_n%d" % (tag, b)], rtol=tol, atol=tol,
                                       err_msg=tag)


def test_pointclouds_algebra_vs_reference_golden_cpu(golden):
    check_pointclouds_algebra(golden, "cpu", 2e-6)


def test_structutils_list_to_padded_and_back():
    """structures/structutils.py:47-124 of the reference (its own tests: tests/structures/test_structutils.py): padding
    to the largest item or to a given size, equisized stacking, cutting back by rows or by (rows, columns), the
    reference's error messages."""
    from gradslam_amd.structures.structutils import list_to_padded, padded_to_list
    g = torch.Generator().manual_seed(3)
    items = [torch.rand(n, c, generator=g) for n, c in ((5, 3), (2, 3), (0, 3), (7, 2))]
    pad = list_to_padded(items)
    assert pad.shape == (4, 7, 3) and pad.dtype == items[0].dtype
    for b, t in enumerate(items):
        assert torch.equal(pad[b, : t.shape[0], : t.shape[1]], t)
        assert float(pad[b, t.shape[0]:].abs().sum()) == 0.0 and float(pad[b, :, t.shape[1]:].abs().sum()) == 0.0
    big = list_to_padded(items, pad_size=(9, 4), pad_value=-1.0)
    assert big.shape == (4, 9, 4) and float(big[2].max()) == -1.0 and torch.equal(big[3, :7, :2], items[3])
    same = [torch.rand(4, 3, generator=g) for _ in range(3)]
    assert torch.equal(list_to_padded(same, equisized=True), torch.stack(same, 0))
    with pytest.raises(ValueError, match="Pad size must contain target size for 1st and 2nd dim"):
        list_to_padded(items, pad_size=(9,))
    with pytest.raises(ValueError, match="Supports only 2-dimensional tensor items"):
        list_to_padded([torch.rand(2, 3, 1)], pad_size=(2, 3))
    back = padded_to_list(pad, [t.shape[0] for t in items])
    assert all(torch.equal(a[:, : b.shape[1]], b) for a, b in zip(back, items))
    back2 = padded_to_list(pad, [tuple(t.shape) for t in items])
    assert all(torch.equal(a, b) for a, b in zip(back2, items))
    assert back2[0].data_ptr() == pad.data_ptr()                      # views, not copies
    assert len(padded_to_list(pad)) == 4 and padded_to_list(pad)[1].shape == (7, 3)
    with pytest.raises(ValueError, match="Supports only 3-dimensional input tensors"):
        padded_to_list(pad[0])
    with pytest.raises(ValueError, match="Split size must be of same length as inputs first dimension"):
        padded_to_list(pad, [1, 2])
    with pytest.raises(ValueError, match="Support only for 2-dimensional unbinded tensor"):
        padded_to_list(pad, [(1, 2, 3)] * 4)


def test_rgbdimages_slice_keeps_the_channels_first_pixel_pos_shape():
    rgb, depth = torch.rand(2, 3, 3, 4, 5), torch.rand(2, 3, 1, 4, 5)
    K = torch.eye(4).repeat(2, 1, 1, 1)
    r = gs.RGBDImages(rgb, depth, K, channels_first=True)
    s = r[:, 1]
    assert s._pixel_pos_shape == (2, 1, 3, 4, 5) and s._rgb_image_shape == (2, 1, 3, 4, 5)
    s2 = r[torch.tensor([1]), 0]     # (a tensor index is not mistaken for "all sequences")
    assert s2._intrinsics.shape == (1, 1, 4, 4) and s2._B == 1
    c = gs.RGBDImages(rgb.permute(0, 1, 3, 4, 2).contiguous(), depth.permute(0, 1, 3, 4, 2).contiguous(), K)
    assert c[:, 2]._pixel_pos_shape == (2, 1, 4, 5, 3)


def test_metrics_package():
    """gradslam_amd.metrics (the reference's package is empty): ATE, RPE, table parity and
    count drift on hand-made cases (map_chamfer needs the GPU: tests/test_hip_api.py)."""
    from gradslam_amd import metrics as M
    L = 6
    a = np.tile(np.eye(4, dtype=np.float32), (L, 1, 1))
    a[:, 0, 3] = 0.1 * np.arange(L)
    b = a.copy()
    assert M.ate_rmse(a, b) == 0.0 and M.ate_rmse(torch.from_numpy(a), b) == 0.0
    b[:, 1, 3] += 3e-3
    assert abs(M.ate_rmse(a, b) - 3e-3) < 1e-9
    r = M.rpe(a, b)
    assert r["pairs"] == L - 1 and r["trans_rmse_m"] < 1e-9 and r["rot_rmse_rad"] == 0.0   # a constant offset has no relative error
    c = a.copy()
    th = 0.01
    for s in range(L):   # yaw grows by `th` per frame
        c[s, :3, :3] = np.array([[np.cos(th * s), 0, np.sin(th * s)], [0, 1, 0], [-np.sin(th * s), 0, np.cos(th * s)]], np.float32)
    r = M.rpe(a, c)
    assert abs(r["rot_rmse_rad"] - th) < 1e-6 and abs(M.rpe(a, c, delta=2)["rot_rmse_rad"] - 2 * th) < 1e-6   # (float32 matrices)
    with pytest.raises(ValueError):
        M.rpe(a, c, delta=L)
    with pytest.raises(ValueError):
        M.ate_rmse(a, b[:3])
    t1 = torch.tensor([[0, 5, 1, 2], [0, 7, 1, 3], [1, 2, 0, 0]])
    assert M.table_parity(t1, t1.clone()) == {"rows_a": 3, "rows_b": 3, "only_in_a": 0, "only_in_b": 0, "identical": True}
    t2 = torch.tensor([[0, 7, 1, 3], [0, 5, 1, 2], [1, 2, 0, 1], [1, 9, 0, 0]])
    p = M.table_parity(t1, t2.numpy())
    assert p == {"rows_a": 3, "rows_b": 4, "only_in_a": 1, "only_in_b": 2, "identical": False}
    assert M.table_parity(t1, t1[[1, 0, 2]])["identical"] is False and M.table_parity(t1, t1[[1, 0, 2]])["only_in_a"] == 0
    d = M.count_drift([10, 20, 33], [10, 21, 30])
    assert d["per_frame"] == [0, 1, 3] and d["max"] == 3 and d["first_frame_with_drift"] == 1 and abs(d["max_relative"] - 0.1) < 1e-12
    assert M.count_drift([1, 2], [1, 2])["first_frame_with_drift"] is None
