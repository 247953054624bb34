"""Round-1 additions to tests/golden (same rules as make_golden.py: the REAL reference, imported from
/root/reference through oracle/refimport.py, on fixed inputs; build-container only).

    python -m oracle.make_golden_extra

  icp0_grad.npz the reference's autograd gradients of the hard-LM point_to_plane_ICP (odometry/icputils.py:235-367)
                on the clouds of icp_unit.npz: d<W,T>/d(src, tgt, normals) for K = 1, 5, 20 and with dist_thresh.
  fusion_grad.npz the reference's autograd through PointFusion(odom="gt") over 3 frames of a 32x40 synthetic
                sequence: d<W, final map (points, normals, colours, confidence counts)>/d(depth, rgb) -- the
                differentiable mapping path (slam/fusionutils.py:653-720 through structures/rgbdimages.py maps).
  slam_grad0.npz the reference's d<W, pose_1>/d depth_0 for the 64x64 two-frame PointFusion(odom="gradicp") run of
                depth_grad.npz: frame 0 reaches pose_1 only through the MAP (the ICP target set).
  gt_odom.npz   GroundTruthOdometryProvider.provide / relative_transformation on seeded poses
                (odometry/groundtruth.py:74-78, geometry/geometryutils.py:413-478).
  icl_items.npz the same for the reference's ICL loader (datasets/icl.py) on tests/tum_fixture.py:write_icl.
  tum_items.npz what the reference's TUM loader (datasets/tum.py) returns for the synthetic TUM-format
                dataset of tests/tum_fixture.py (native-size frames: the cv2 shim only copies), for
                every sequence of two constructor configurations.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, REPO)
from oracle import refimport  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")


def random_poses(rng, n):
    from scipy.spatial.transform import Rotation
    T = np.tile(np.eye(4, dtype=np.float32), (n, 1, 1))
    T[:, :3, :3] = Rotation.from_rotvec(rng.normal(size=(n, 3)) * 0.7).as_matrix().astype(np.float32)
    T[:, :3, 3] = rng.normal(size=(n, 3)).astype(np.float32) * 2
    return T


def main():
    refimport.import_reference()
    import torch
    import gradslam
    from gradslam.geometry.geometryutils import relative_transformation
    from gradslam.odometry.groundtruth import GroundTruthOdometryProvider
    rng = np.random.default_rng(7)
    B = 6
    T1, T2 = random_poses(rng, B), random_poses(rng, B)
    T1[0, :3, :3] *= 1.01   # an imperfect rotation: the general inverse must be used
    mk = lambda T: gradslam.RGBDImages(torch.zeros(B, 1, 4, 4, 3), torch.ones(B, 1, 4, 4, 1),  # noqa: E731
                                       torch.eye(4).repeat(B, 1, 1, 1), torch.from_numpy(T).unsqueeze(1))
    rel = GroundTruthOdometryProvider().provide(mk(T1), mk(T2))
    rel2 = relative_transformation(torch.from_numpy(T1), torch.from_numpy(T2))
    assert torch.equal(rel[:, 0], rel2)
    np.savez_compressed(os.path.join(OUT, "gt_odom.npz"), T1=T1, T2=T2, rel=rel.numpy())
    print("gt_odom.npz", rel.shape)


def tum_items():
    import tempfile
    refimport.import_reference()
    from gradslam.datasets.tum import TUM
    import importlib.util
    spec = importlib.util.spec_from_file_location("tum_fixture", os.path.join(REPO, "tests", "tum_fixture.py"))
    tum_fixture = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tum_fixture)
    out = {}
    with tempfile.TemporaryDirectory() as root:
        tum_fixture.write(root)
        for case, kw in tum_fixture.CASES.items():
            ds = TUM(root, **kw)
            out[case + "/len"] = np.array(len(ds))
            for i in range(len(ds)):
                colors, depths, K, poses, transforms, names, stamps = ds[i]
                for k, v in (("colors", colors), ("depths", depths), ("intrinsics", K), ("poses", poses),
                             ("transforms", transforms)):
                    out["%s/%d/%s" % (case, i, k)] = v.numpy()
                out["%s/%d/names" % (case, i)] = np.array(names)
                out["%s/%d/stamps" % (case, i)] = np.array(stamps)
    np.savez_compressed(os.path.join(OUT, "tum_items.npz"), **out)
    print("tum_items.npz", {k: int(out[k]) for k in out if k.endswith("/len")})
    # the same for the reference's ICL loader (datasets/icl.py)
    from gradslam.datasets.icl import ICL
    out = {}
    with tempfile.TemporaryDirectory() as root:
        tum_fixture.write_icl(root)
        for case, kw in tum_fixture.ICL_CASES.items():
            ds = ICL(root, **kw)
            out[case + "/len"] = np.array(len(ds))
            for i in range(len(ds)):
                colors, depths, K, poses, transforms, names = ds[i]
                for k, v in (("colors", colors), ("depths", depths), ("intrinsics", K), ("poses", poses),
                             ("transforms", transforms)):
                    out["%s/%d/%s" % (case, i, k)] = v.numpy()
                out["%s/%d/names" % (case, i)] = np.array(names)
    np.savez_compressed(os.path.join(OUT, "icl_items.npz"), **out)
    print("icl_items.npz", {k: int(out[k]) for k in out if k.endswith("/len")})


def scannet_items():
    """scannet_items.npz: what the reference's Scannet loader (datasets/scannet.py) returns for
    tests/tum_fixture.py:write_scannet (native-size frames), two constructor configurations."""
    import tempfile
    refimport.import_reference()
    from gradslam.datasets.scannet import Scannet
    import importlib.util
    spec = importlib.util.spec_from_file_location("tum_fixture", os.path.join(REPO, "tests", "tum_fixture.py"))
    fx = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fx)
    out = {}
    with tempfile.TemporaryDirectory() as root:
        base, meta = fx.write_scannet(root)
        for case, kw in fx.SCANNET_CASES.items():
            ds = Scannet(base, meta, **kw)
            out[case + "/len"] = np.array(len(ds))
            for i in range(len(ds)):
                colors, depths, K, poses, transforms, names, labels = ds[i]
                for k, v in (("colors", colors), ("depths", depths), ("intrinsics", K), ("poses", poses),
                             ("transforms", transforms), ("labels", labels)):
                    out["%s/%d/%s" % (case, i, k)] = v.numpy()
                out["%s/%d/names" % (case, i)] = np.array(names)
    np.savez_compressed(os.path.join(OUT, "scannet_items.npz"), **out)
    print("scannet_items.npz", {k: int(out[k]) for k in out if k.endswith("/len")})


def intrinsics_grad():
    """intrinsics_grad.npz: the reference's autograd d/dK of <Wv, vertex_map> + <Wn, normal_map> + <Wa, alpha>
    (structures/rgbdimages.py:662-675 through inverse_intrinsics, geometry/projutils.py:437-449) on a HOLE-FREE 96x128
    frame: next to depth holes the normal normalisation makes the sums ill-conditioned (the reference's own float32
    and float64 autograd then disagree by more than 100 %), so such a frame cannot pin anything."""
    refimport.import_reference()
    import torch
    import gradslam
    from gradslam.slam import fusionutils as fu
    from gradslam_amd.datasets.synthetic import make_sequence
    sq = make_sequence(2, 96, 128, seed=5, hole_frac=0.0)
    g = np.load(os.path.join(OUT, "depth_grad.npz"))
    T = torch.from_numpy
    K = T(sq["intrinsics"][None]).clone().requires_grad_(True)
    d1 = T(sq["depths"][None, 1:2]).clone().requires_grad_(True)
    f1 = gradslam.RGBDImages(torch.zeros(1, 1, 96, 128, 3), d1, K, T(sq["poses"][None, :1]))
    al = fu.get_alpha(f1.vertex_map, dim=4, keepdim=True, sigma=0.6)
    ((f1.vertex_map[0, 0] * T(g["Wv"])).sum() + (f1.normal_map[0, 0] * T(g["Wn"])).sum()
     + (al[0, 0, ..., 0] * T(g["Wa"])).sum()).backward()
    np.savez_compressed(os.path.join(OUT, "intrinsics_grad.npz"), K_grad=K.grad[0, 0].numpy(),
                        depth=sq["depths"][1, ..., 0], intrinsics=sq["intrinsics"][0], pose=sq["poses"][0],
                        depth_grad=d1.grad[0, 0, ..., 0].numpy())
    print("intrinsics_grad.npz", K.grad[0, 0].numpy())


def icp0_grad():
    refimport.import_reference()
    import torch
    from gradslam.odometry import icputils
    gi = np.load(os.path.join(OUT, "icp_unit.npz"))
    W = np.load(os.path.join(OUT, "icp_grad.npz"))["W"]
    out = dict(W=W)
    for K, thr in ((1, None), (5, None), (20, None), (5, 1e-4)):
        leaf = [torch.from_numpy(gi[k]).clone().requires_grad_(True) for k in ("src", "tgt", "tgt_normals")]
        T, _ = icputils.point_to_plane_ICP(leaf[0][None], leaf[1][None], leaf[2][None], torch.eye(4), numiters=K,
                                           dist_thresh=thr)
        (T * torch.from_numpy(W)).sum().backward()
        tag = "K%d%s" % (K, "" if thr is None else "_thr")
        out[tag + "_T"] = T.detach().numpy()
        for name, t in zip(("src", "tgt", "tn"), leaf):
            out[tag + "_" + name] = t.grad.numpy()
    np.savez_compressed(os.path.join(OUT, "icp0_grad.npz"), **out)
    print("icp0_grad.npz", sorted(k for k in out if k.endswith("_T")))


def fusion_grad():
    refimport.import_reference()
    import torch
    import gradslam
    from gradslam_amd.datasets.synthetic import make_sequence
    s = make_sequence(3, 32, 40, seed=3, hole_frac=0.1)
    depth = torch.from_numpy(s["depths"][None]).clone().requires_grad_(True)
    rgb = torch.from_numpy(s["colors"][None]).clone().requires_grad_(True)
    poses = torch.from_numpy(s["poses"][None]).clone().requires_grad_(True)
    frames = gradslam.RGBDImages(rgb, depth, torch.from_numpy(s["intrinsics"][None]), poses)
    slam = gradslam.slam.PointFusion(odom="gt", dsratio=4)
    pc, _ = slam(frames)
    rng = np.random.default_rng(1)
    n = pc.points_list[0].shape[0]
    W = {k: rng.standard_normal((n, c)).astype(np.float32) for k, c in (("points", 3), ("normals", 3), ("colors", 3),
                                                                     ("features", 1))}
    loss = sum((getattr(pc, k + "_list")[0] * torch.from_numpy(W[k])).sum() for k in W)
    loss.backward()
    out = dict(colors=s["colors"], depths=s["depths"], intrinsics=s["intrinsics"], poses=s["poses"],
               depth_grad=depth.grad[0].numpy(), rgb_grad=rgb.grad[0].numpy(), poses_grad=poses.grad[0].numpy(),
               n=np.array(n))
    for k in W:
        out["W_" + k] = W[k]
        out["map_" + k] = getattr(pc, k + "_list")[0].detach().numpy()
    # the same for ICPSLAM's aggregate map (update_map_aggregate, slam/fusionutils.py:725-758): 2 frames
    d2 = torch.from_numpy(s["depths"][None, :2]).clone().requires_grad_(True)
    fr2 = gradslam.RGBDImages(torch.from_numpy(s["colors"][None, :2]), d2, torch.from_numpy(s["intrinsics"][None]),
                              torch.from_numpy(s["poses"][None, :2]))
    pca, _ = gradslam.slam.ICPSLAM(odom="gt", dsratio=4)(fr2)
    na = pca.points_list[0].shape[0]
    Wa = rng.standard_normal((na, 3)).astype(np.float32)
    ((pca.points_list[0] * torch.from_numpy(Wa)).sum() + (pca.normals_list[0] * torch.from_numpy(Wa[::-1].copy())).sum()).backward()
    out["agg_W"], out["agg_n"], out["agg_depth_grad"] = Wa, np.array(na), d2.grad[0].numpy()
    np.savez_compressed(os.path.join(OUT, "fusion_grad.npz"), **out)
    print("fusion_grad.npz: %d surfels, |d depth| max %.3e, |d rgb| max %.3e"
          % (n, np.abs(out["depth_grad"]).max(), np.abs(out["rgb_grad"]).max()))


def slam_grad0():
    refimport.import_reference()
    import torch
    import gradslam
    g = np.load(os.path.join(OUT, "depth_grad.npz"))
    dd = torch.from_numpy(g["slam_depths"][None]).clone().requires_grad_(True)
    pp = torch.from_numpy(g["slam_poses"][None]).clone()
    pp[:, 1:] = pp[:, :1]
    frames = gradslam.RGBDImages(torch.from_numpy(g["slam_colors"][None]), dd,
                                 torch.from_numpy(g["slam_intrinsics"][None, None]), pp)
    _, rp = gradslam.slam.PointFusion(odom="gradicp")(frames)
    (rp[0, 1] * torch.from_numpy(g["chain_W"])).sum().backward()
    assert np.array_equal(dd.grad[0, 1, ..., 0].numpy(), g["slam_depth1_grad"])   # same run as depth_grad.npz (c)
    np.savez_compressed(os.path.join(OUT, "slam_grad0.npz"), depth0_grad=dd.grad[0, 0, ..., 0].numpy().copy())
    print("slam_grad0.npz: |d depth0| max %.3e, nonzero %d" % (np.abs(dd.grad[0, 0]).max(), int((dd.grad[0, 0] != 0).sum())))


if __name__ == "__main__":
    main()
    tum_items()
    scannet_items()
    intrinsics_grad()
    icp0_grad()
    fusion_grad()
    slam_grad0()
