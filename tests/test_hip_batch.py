"""GPU tests of the multi-sequence (batched) frame loop: gs_frame_maps_batch_f32, gs_localize_batch_f32 and
gs_update_map_fusion_batch_f32 run every kernel of a frame for B sequences per launch.  Per sequence the results
must be BIT-IDENTICAL to the one-sequence entry points (same device functions) and to B separate runs of the
driver; at the benchmarked 640x480 resolution the driver is checked end to end against the golden recorded from the
real reference (tests/golden/pf640.npz, oracle/make_golden_640.py) and against the oracle's frame loop."""
import math

import numpy as np
import pytest
import torch

from gradslam_amd.datasets.synthetic import make_sequence
from gradslam_amd.metrics import ate_rmse as ate

pytestmark = pytest.mark.gpu

DIST_TH, DOT_TH, SIGMA = 0.05, math.cos(20 * math.pi / 180), 0.6
T = torch.from_numpy


def host(t):
    return t.detach().cpu().numpy()


def dev(a):
    return T(np.ascontiguousarray(a)).cuda()


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


def frames_of(gs, seqs, L=None):
    stack = lambda k: T(np.stack([s[k][:L] if k != "intrinsics" else s[k] for s in seqs])).cuda()  # noqa: E731
    poses = stack("poses")
    poses[:, 1:] = poses[:, :1]
    return gs.RGBDImages(stack("colors"), stack("depths"), stack("intrinsics"), poses)


def test_frame_maps_batch_equals_single_frames(ops):
    rng = np.random.default_rng(0)
    B, L, H, W = 3, 2, 67, 131   # ragged tile edges
    depth = (1.0 + rng.random((B * L, H, W))).astype(np.float32)
    depth[rng.random(depth.shape) < 0.1] = 0
    K = np.stack([np.array([[100 + 7 * b, 0, 60, 0], [0, -(90 + 3 * b), 30, 0], [0, 0, 1, 0], [0, 0, 0, 1]], np.float32)
                  for b in range(B)])
    stack = dev(depth).view(B, L, H, W)
    v, n, a = ops.frame_maps_batch(stack, dev(K), SIGMA)
    for f in range(B * L):
        v1, n1, a1, _ = ops.frame_maps(dev(depth[f]), dev(K[f // L]), SIGMA)
        assert torch.equal(v[f // L, f % L], v1) and torch.equal(n[f // L, f % L], n1) and torch.equal(a[f // L, f % L], a1)
    v2, n2, a2 = ops.frame_maps_batch(stack, dev(K), None)
    assert a2 is None and torch.equal(v2, v) and torch.equal(n2, n)
    # a one-frame slice of the stack is read in place (sequence stride = L frames)
    v3, n3, a3 = ops.frame_maps_batch(stack[:, 1:2], dev(K), SIGMA)
    assert torch.equal(v3[:, 0], v[:, 1]) and torch.equal(n3[:, 0], n[:, 1]) and torch.equal(a3[:, 0], a[:, 1])


def _build_maps(ops, seeds, H, W, frames=2):
    """per sequence: a surfel map of `frames` fused frames on capacity-backed buffers + the next frame's inputs"""
    out = []
    for seed in seeds:
        s = make_sequence(frames + 1, H, W, seed=seed, hole_frac=0.05 + 0.03 * (seed % 3))
        K = dev(s["intrinsics"][0])
        cap = (frames + 2) * H * W
        bufs = [torch.zeros((cap, k), device="cuda") for k in (3, 3, 3, 1)]
        n_map = 0
        for f in range(frames):
            d, pose = dev(s["depths"][f, ..., 0]), dev(s["poses"][f])
            v, n, a, _ = ops.frame_maps(d, K, SIGMA)
            gv, gn = ops.global_maps(v, n, d, pose)
            pix = ops.project_map(bufs[0][:n_map], pose, K, H, W)
            best = ops.associate(pix, bufs[0][:n_map], bufs[1][:n_map], bufs[3][:n_map], gv, gn, DIST_TH, DOT_TH)
            n_map = ops.fuse_append_(*bufs, n_map, best, gv, gn, dev(s["colors"][f]), a, d)
        out.append(dict(seq=s, K=K, bufs=bufs, n=n_map, prev_pose=dev(s["poses"][frames - 1])))
    return out


@pytest.mark.parametrize("mode", [1, 0])
def test_localize_batch_equals_single_sequence_chain(ops, mode):
    """B = 3 sequences with different maps (sizes differ): gs_localize_batch_f32 == gs_lattice_source_f32 +
    gs_project_map_f32 + gs_icp_map_dc_f32 per sequence, bit for bit; with host and with device-side counts."""
    H, W, ds = 240, 320, 4
    maps = _build_maps(ops, (3, 4, 8), H, W)
    assert len({m["n"] for m in maps}) == 3
    ref, vs, ds_ = [], [], []
    for m in maps:
        d = dev(m["seq"]["depths"][2, ..., 0])
        v, _, _, _ = ops.frame_maps(d, m["K"], SIGMA)
        src = ops.lattice_source(v, d, m["prev_pose"], ds)
        P, N = m["bufs"][0][:m["n"]], m["bufs"][1][:m["n"]]
        pix = ops.project_map(P, m["prev_pose"], m["K"], H, W)
        ref.append(ops.icp_map(src, P, N, pix, W, ds, compose=m["prev_pose"], mode=mode, numiters=10))
        vs.append(v)
        ds_.append(d)
    Ks, poses = torch.stack([m["K"] for m in maps]), torch.stack([m["prev_pose"] for m in maps])
    for device_counts in (False, True):
        mv = []
        for m in maps:
            n_dev = torch.tensor([m["n"]], dtype=torch.int64, device="cuda") if device_counts else None
            bound = m["n"] + (777 if device_counts else 0)   # bound-sized launches, garbage rows behind the count
            mv.append((m["bufs"][0], m["bufs"][1], bound, n_dev))
        out = ops.localize_batch(torch.stack(vs), torch.stack(ds_), Ks, poses, mv, ds, mode=mode, numiters=10)
        for b in range(3):
            assert torch.equal(out[b], ref[b]), (b, device_counts)
    # one sequence alone through the batched entry point
    m = maps[1]
    one = ops.localize_batch(vs[1][None], ds_[1][None], Ks[1:2], poses[1:2], [(m["bufs"][0], m["bufs"][1], m["n"], None)],
                             ds, mode=mode, numiters=10)
    assert torch.equal(one[0], ref[1])


def test_update_map_batch_equals_separate_kernels(ops):
    """gs_update_map_fusion_batch_f32 on 3 sequences == global maps + projection + association + fuse per sequence."""
    H, W = 120, 160
    maps = _build_maps(ops, (5, 6, 7), H, W)
    ref, frames = [], []
    for m in maps:
        s = m["seq"]
        d, pose = dev(s["depths"][2, ..., 0]), dev(s["poses"][2])
        v, n, a, _ = ops.frame_maps(d, m["K"], SIGMA)
        gv, gn = ops.global_maps(v, n, d, pose)
        bufs = [t.clone() for t in m["bufs"]]
        pix = ops.project_map(bufs[0][:m["n"]], pose, m["K"], H, W)
        best = ops.associate(pix, bufs[0][:m["n"]], bufs[1][:m["n"]], bufs[3][:m["n"]], gv, gn, DIST_TH, DOT_TH)
        n1 = ops.fuse_append_(*bufs, m["n"], best, gv, gn, dev(s["colors"][2]), a, d)
        ref.append((bufs, n1, best, gv, gn))
        frames.append((v, n, d, dev(s["colors"][2]), a, pose))
    st = lambda i: torch.stack([f[i] for f in frames])  # noqa: E731
    mv = [(*[t.clone() for t in m["bufs"]], m["n"], None) for m in maps]
    cnt, gv, gn, best = ops.update_map_fusion_batch_(mv, st(0), st(1), st(2), st(3), st(4), st(5),
                                                     torch.stack([m["K"] for m in maps]), DIST_TH, DOT_TH)
    counts = host(cnt)
    for b in range(3):
        bufs, n1, rbest, rgv, rgn = ref[b]
        assert counts[b] == n1
        assert torch.equal(best[b], rbest) and torch.equal(gv[b], rgv) and torch.equal(gn[b], rgn)
        for k in range(4):
            assert torch.equal(mv[b][k][:n1], bufs[k][:n1]), (b, k)


def test_update_map_batch_settles_rows_with_identical_keys(ops):
    """Round 5: the fused update no longer makes a pass over the map to pick the winner of a pixel -- the row that attains
    the pixel's key wins, and two rows with bit-identical keys (same confidence, same distance to the pixel's vertex: the
    LOWER index wins, slam/fusionutils.py:491-536) are noticed by the key pass and settled by a small launch.  Maps in
    which EVERY row has an exact duplicate with a higher index (and, for the third sequence, two): the batched update must
    give what the table-level kernels give (bit-exact against the reference's unique-correspondence table,
    tests/test_hip_parity.py), winners included."""
    H, W = 120, 160
    maps = _build_maps(ops, (5, 6, 7), H, W)
    for i, m in enumerate(maps):     # duplicate every row (twice for the last sequence)
        n, reps = m["n"], (3 if i == 2 else 2)
        big = [torch.zeros((reps * n + 2 * H * W, t.shape[1]), device="cuda") for t in m["bufs"]]
        for t, src in zip(big, m["bufs"]):
            for r in range(reps):
                t[r * n:(r + 1) * n] = src[:n]
        m["bufs"], m["n"] = big, reps * n
    ref, frames = [], []
    for m in maps:
        s = m["seq"]
        d, pose = dev(s["depths"][2, ..., 0]), dev(s["poses"][2])
        v, n, a, _ = ops.frame_maps(d, m["K"], SIGMA)
        gv, gn = ops.global_maps(v, n, d, pose)
        bufs = [t.clone() for t in m["bufs"]]
        pix = ops.project_map(bufs[0][:m["n"]], pose, m["K"], H, W)
        best = ops.associate(pix, bufs[0][:m["n"]], bufs[1][:m["n"]], bufs[3][:m["n"]], gv, gn, DIST_TH, DOT_TH)
        assert int((best >= 0).sum()) > 1000 and int(best.max()) < m["n"] // (3 if m is maps[2] else 2)   # lowest copy wins
        n1 = ops.fuse_append_(*bufs, m["n"], best, gv, gn, dev(s["colors"][2]), a, d)
        ref.append((bufs, n1, best))
        frames.append((v, n, d, dev(s["colors"][2]), a, pose))
    st = lambda i: torch.stack([f[i] for f in frames])  # noqa: E731
    mv = [(*[t.clone() for t in m["bufs"]], m["n"], None) for m in maps]
    cnt, gv, gn, best = ops.update_map_fusion_batch_(mv, st(0), st(1), st(2), st(3), st(4), st(5),
                                                     torch.stack([m["K"] for m in maps]), DIST_TH, DOT_TH)
    counts = host(cnt)
    for b in range(3):
        bufs, n1, rbest = ref[b]
        assert counts[b] == n1
        assert torch.equal(best[b], rbest), b
        for k in range(4):
            assert torch.equal(mv[b][k][:n1], bufs[k][:n1]), (b, k)


def _run_pointfusion(gs, seqs, L, odom="gradicp"):
    frames = frames_of(gs, seqs, L)
    pc, rp = gs.slam.PointFusion(odom=odom, device="cuda")(frames)
    return pc, rp


@pytest.mark.parametrize("odom", ["gradicp", "icp"])
def test_pointfusion_batch8_equals_eight_single_runs(gs, odom):
    """B = 8 sequences in one batch (every kernel serves the 8 sequences at once) vs 8 runs of one sequence each:
    identical poses and identical maps, bit for bit."""
    L, H, W = 5, 96, 128
    seqs = [make_sequence(L, H, W, seed=100 + b, hole_frac=0.03 + 0.01 * b) for b in range(8)]
    pc8, rp8 = _run_pointfusion(gs, seqs, L, odom)
    assert len(pc8) == 8 and rp8.shape == (8, L, 4, 4)
    for b in (0, 3, 7):
        pc1, rp1 = _run_pointfusion(gs, seqs[b:b + 1], L, odom)
        assert torch.equal(rp8[b], rp1[0]), b
        n = pc1.points_list[0].shape[0]
        assert pc8.points_list[b].shape[0] == n
        for a8, a1 in ((pc8.points_list, pc1.points_list), (pc8.normals_list, pc1.normals_list),
                       (pc8.colors_list, pc1.colors_list), (pc8.features_list, pc1.features_list)):
            assert torch.equal(a8[b], a1[0]), b


def test_pointfusion_640x480_vs_reference_golden(gs, golden):
    """BASELINE configs[1] end to end: PointFusion(gradicp) on the seeded 640x480 sequence of bench.py against the
    run of the REAL reference recorded in tests/golden/pf640.npz: pose ATE <= 1e-4 m (BASELINE.json target), the
    same number of surfels after every frame (i.e. identical association / append decisions), and the fused points
    within 1e-5 m on average."""
    g = golden("pf640")
    L, H, W = int(g["poses"].shape[0]), int(g["H"]), int(g["W"])
    s = make_sequence(L, H, W, seed=int(g["seed"]))
    assert abs(float(s["depths"].astype(np.float64).sum()) - float(g["depth_sum"])) < 1e-6 * float(g["depth_sum"])
    frames = frames_of(gs, [s])
    slam = gs.slam.PointFusion(odom="gradicp", device="cuda")
    pc, prev, counts, sums, rec = gs.Pointclouds(device="cuda"), None, [], [], []
    for f in range(L):
        live = frames[:, f]
        pc, pose = slam.step(pc, live, prev, inplace=True)
        prev = live
        rec.append(host(pose[0, 0]))
        counts.append(pc.points_list[0].shape[0])
        sums.append(host(pc.points_list[0].double().sum(0)))
    rec = np.stack(rec)
    assert ate(rec, g["poses"]) <= 1e-4, ate(rec, g["poses"])
    np.testing.assert_allclose(rec, g["poses"], rtol=0, atol=1e-4)
    # association decisions: surfel counts per frame.  Given the same pose AND the same confidence counts the tables are
    # bit-exact (test_hip_parity).  Two things differ from the reference here: (1) the poses, by ~1e-5 (float64 vs
    # float32 normal equations), which flips the similarity / in-frame test of pixels that sit within that distance of
    # a threshold -- the bulk of the measured <= 140 of 307 200 pixels per frame (0.05 %); (2) alpha, by 1 ulp on ~10 % of
    # the pixels (torch's exp is MKL VML on this build and cannot be restated), which alone flips <= 2 decisions in 8
    # frames (test_pointfusion_640x480_ground_truth_odometry_vs_reference_golden: same poses, counts within 2).  The
    # bound is 0.05 % of the map
    diff = np.abs(np.asarray(counts) - g["counts"])
    assert counts[0] == g["counts"][0]   # frame 0: no ICP involved, identical
    assert diff.max() <= 5e-4 * g["counts"][-1], (counts, g["counts"].tolist())
    for f in range(L):
        np.testing.assert_allclose(sums[f], g["sum_points"][f], rtol=0, atol=1e-5 * counts[f] + 4.0 * diff[f] + 1e-3)


def test_pointfusion_640x480_ground_truth_odometry_vs_reference_golden(gs, golden):
    """K5 / K6 across 8 frames at 640x480 with NO ICP in the loop: PointFusion(odom="gt") against the
    REAL reference (tests/golden/pf640_gt.npz) and against the oracle's loop.  HIP == oracle bit for bit (points,
    normals, colours, confidence counts of the final 5.7e5-surfel map).  Against the reference: first map exact
    (sha256), surfel counts identical for the first 4 frames and within 3 afterwards (measured: equal for 6 frames,
    then -1, -2 -- alpha is 1 ulp away from torch's MKL exp on ~10 % of the pixels, which moves one merge decision per
    ~1.5 M), attribute sums per surfel within 2e-5 m / 1e-5 / 2e-3 of 255 / 1e-9."""
    import hashlib
    from oracle import slam as oslam
    g = golden("pf640_gt")
    L, H, W = int(g["poses"].shape[0]), int(g["H"]), int(g["W"])
    s = make_sequence(L, H, W, seed=int(g["seed"]))
    assert abs(float(s["depths"].astype(np.float64).sum()) - float(g["depth_sum"])) < 1e-6 * float(g["depth_sum"])
    frames = gs.RGBDImages(T(s["colors"][None]).cuda(), T(s["depths"][None]).cuda(), T(s["intrinsics"][None]).cuda(),
                           T(s["poses"][None]).cuda())
    slam = gs.slam.PointFusion(odom="gt", device="cuda")
    pc, prev = gs.Pointclouds(device="cuda"), None
    for f in range(L):
        live = frames[:, f]
        pc, pose = slam.step(pc, live, prev, inplace=True)
        prev = live
        assert np.array_equal(host(pose[0, 0]), s["poses"][f])
        n = pc.points_list[0].shape[0]
        if f == 0:
            sha = [hashlib.sha256(host(t[0]).tobytes()).hexdigest() for t in (pc.points_list, pc.normals_list, pc.colors_list)]
            assert sha == [str(x) for x in g["sha_frame0"]]
        dn = abs(n - int(g["counts"][f]))
        assert dn <= (0 if f < 4 else 3), (f, n, int(g["counts"][f]))
        for lst, key, tol, scale in ((pc.points_list, "sum_points", 2e-5, 3.0), (pc.normals_list, "sum_normals", 1e-5, 3.0),
                                     (pc.colors_list, "sum_colors", 2e-3, 255.0), (pc.features_list, "sum_ccounts", 1e-9, 3.0)):
            assert np.abs(host(lst[0].double().sum(0)) - g[key][f]).max() <= tol * n + 4.0 * dn * scale, (f, key)
    m, _ = oslam.run_sequence(s["colors"], s["depths"], s["intrinsics"][0], s["poses"], odom="gt")
    assert len(m) == pc.points_list[0].shape[0]
    for mine, ref in ((pc.points_list, m.points), (pc.normals_list, m.normals), (pc.colors_list, m.colors),
                      (pc.features_list, m.ccounts)):
        assert np.array_equal(host(mine[0]).view(np.int32), np.ascontiguousarray(ref).view(np.int32))


def test_pointfusion_640x480_seeds_1_to_7_batched_vs_reference_goldens(gs, golden):
    """The other sequences of the benchmarked batch: seeds 1..7 tracked as ONE batch (B = 7) against 25-frame runs of the
    REAL reference per seed (tests/golden/pf640_s<seed>.npz, oracle/make_golden_640.py --seed k --frames 25; round 6: the
    goldens cover the benchmark's timed window, frames 5 .. 24): ATE <= 1e-4 m, surfel counts within 0.05 %.  (Seed 0 has
    the 20- and 60-frame goldens; bench.py reports the ATE of every sequence over its warm-up and its timed frames.)"""
    seeds = (1, 2, 3, 4, 5, 6, 7)
    gold = [golden("pf640_s%d" % sd) for sd in seeds]
    L = int(gold[0]["poses"].shape[0])
    seqs = [make_sequence(L, 480, 640, seed=sd) for sd in seeds]
    for s, g in zip(seqs, gold):
        assert abs(float(s["depths"].astype(np.float64).sum()) - float(g["depth_sum"])) < 1e-6 * float(g["depth_sum"])
    frames = frames_of(gs, seqs)
    slam = gs.slam.PointFusion(odom="gradicp", device="cuda")
    pc, prev, rec, counts = gs.Pointclouds(device="cuda"), None, [], []
    for f in range(L):
        live = frames[:, f]
        pc, pose = slam.step(pc, live, prev, inplace=True)
        prev = live
        rec.append(host(pose[:, 0]))
        counts.append([t.shape[0] for t in pc.points_list])
    rec, counts = np.stack(rec, 1), np.asarray(counts)   # (7, L, 4, 4), (L, 7)
    for b, g in enumerate(gold):
        assert ate(rec[b], g["poses"]) <= 1e-4, (b, ate(rec[b], g["poses"]))
        assert counts[0, b] == g["counts"][0]
        assert np.abs(counts[:, b] - g["counts"]).max() <= 5e-4 * g["counts"][-1], (b, counts[:, b], g["counts"].tolist())


def test_icpslam_640x480_vs_reference_golden(gs, golden):
    """ICPSLAM(odom="icp": hard-LM ICP odometry, aggregate mapping) on the same sequence against the REAL reference's
    run (tests/golden/icpslam640.npz): pose ATE <= 1e-4 m; aggregate mapping appends every valid pixel, so the map
    sizes are identical by construction and the point sums differ only through the poses."""
    g = golden("icpslam640")
    L, H, W = int(g["poses"].shape[0]), int(g["H"]), int(g["W"])
    s = make_sequence(L, H, W, seed=int(g["seed"]))
    assert abs(float(s["depths"].astype(np.float64).sum()) - float(g["depth_sum"])) < 1e-6 * float(g["depth_sum"])
    frames = frames_of(gs, [s])
    slam = gs.slam.ICPSLAM(odom="icp", device="cuda")
    pc, prev, rec = gs.Pointclouds(device="cuda"), None, []
    for f in range(L):
        live = frames[:, f]
        pc, pose = slam.step(pc, live, prev, inplace=True)
        prev = live
        rec.append(host(pose[0, 0]))
        n = pc.points_list[0].shape[0]
        assert n == int(g["counts"][f]), (f, n, int(g["counts"][f]))
        # a pose error e moves every point of the frame by <= e * (1 + |p|): 1e-4 per point is generous
        np.testing.assert_allclose(host(pc.points_list[0].double().sum(0)), g["sum_points"][f], rtol=0, atol=1e-4 * n)
    rec = np.stack(rec)
    assert ate(rec, g["poses"]) <= 1e-4, ate(rec, g["poses"])
    np.testing.assert_allclose(rec, g["poses"], rtol=0, atol=1e-4)
    assert not pc.has_features


def test_pointfusion_640x480_vs_oracle(gs):
    """the same config against the oracle's frame loop (float64 normal equations on both sides): poses within 2e-6,
    identical surfel counts, identical points where no ICP rounding enters (frame 0)."""
    from oracle import slam as oslam
    L, H, W = 5, 480, 640
    s = make_sequence(L, H, W, seed=0)
    pc, rp = _run_pointfusion(gs, [s], L)
    poses = s["poses"].copy()
    poses[1:] = poses[:1]
    m, op = oslam.run_sequence(s["colors"], s["depths"], s["intrinsics"][0], poses)
    np.testing.assert_allclose(host(rp[0]), op, rtol=0, atol=2e-6)
    assert ate(host(rp[0]), op) <= 2e-6
    assert pc.points_list[0].shape[0] == len(m)
    np.testing.assert_allclose(host(pc.points_list[0]), m.points, rtol=1e-5, atol=1e-5)


def test_numiters_beyond_64_and_negative_dist_thresh(ops):
    """the reference accepts any iteration count; a negative dist_thresh keeps no pair."""
    s = make_sequence(2, 96, 128, seed=3)
    K, pose = dev(s["intrinsics"][0]), dev(s["poses"][0])
    d0, d1 = dev(s["depths"][0, ..., 0]), dev(s["depths"][1, ..., 0])
    v0, n0, _, _ = ops.frame_maps(d0, K)
    gv0, gn0 = ops.global_maps(v0, n0, d0, pose)
    tgt, tn, _ = ops.downsample_frame(gv0, gn0, None, d0, 2)
    v1, n1, _, _ = ops.frame_maps(d1, K)
    gv1, _ = ops.global_maps(v1, n1, d1, pose)
    src, _, _ = ops.downsample_frame(gv1, None, None, d1, 4)
    T20 = ops.icp(src, tgt, tn, mode=1, numiters=20, return_idx=False)
    T100 = ops.icp(src, tgt, tn, mode=1, numiters=100, return_idx=False)
    assert bool(torch.isfinite(T100).all()) and float((T100 - T20).abs().max()) < 1e-3
    Tneg = ops.icp(src, tgt, tn, mode=0, numiters=5, dist_thresh=-1.0, return_idx=False)
    Tzero = ops.icp(src, tgt, tn, mode=0, numiters=5, dist_thresh=0.0, return_idx=False)
    assert torch.equal(Tneg, Tzero)   # every pair filtered out: A = 0, xi = 0


@pytest.mark.skipif(not torch.cuda.is_available() or torch.cuda.device_count() < 2, reason="needs 2 GPUs")
def test_two_ranks_rccl_gather():
    """bench.py's N > 1 path on real hardware: 2 ranks, 4 sequences, RCCL pose / map gather."""
    import json
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2",
                        "--batch", "4", "--height", "120", "--width", "160", "--no-cpu-baseline", "--no-roofline-pass"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["config"]["sequences_total"] == 4


def test_two_ranks_sharing_one_gpu_rehearsal():
    """The same N > 1 path when only one GPU is there: two self-spawned ranks share cuda:0 and rendezvous over gloo
    (RCCL refuses two ranks on one device).  Exercises the spawn, the sharding, the barriers and max-over-ranks
    timing, and both gathers with device tensors; the per-sequence results are checked against a single-rank run."""
    import json
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--steps", "3", "--warmup", "2", "--batch", "4", "--height", "120", "--width", "160", "--no-cpu-baseline",
              "--no-roofline-pass"]
    env = dict(os.environ, GRADSLAM_DIST_BACKEND="gloo", GRADSLAM_BENCH_SHARE_GPU="1")
    env.pop("WORLD_SIZE", None)
    lines = []
    for n in ("2", "1"):
        r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", n] + common, capture_output=True,
                           text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        lines.append(json.loads(r.stdout.strip().splitlines()[-1]))
    two, one = lines
    assert two["n_gpus"] == 2 and two["config"]["sequences_total"] == 4 and two["config"]["sequences_per_gpu"] == 2
    assert one["n_gpus"] == 1 and one["config"]["sequences_per_gpu"] == 4
    assert two["config"]["poses_sha"] == one["config"]["poses_sha"]          # gathered poses: identical bits
    assert two["config"]["map_surfels_all"] == one["config"]["map_surfels_all"]
    # what every rank reports about itself: its sequences, its own clock, the fingerprint of its poses
    pr = two["ranks"]["per_rank"]
    assert [r["rank"] for r in pr] == [0, 1] and [r["sequences"] for r in pr] == [[0, 1], [2, 3]]
    assert pr[0]["poses_sha"] != pr[1]["poses_sha"] and all(r["elapsed_s"] > 0 for r in pr)
    assert two["ranks"]["elapsed_s_min"] <= two["ranks"]["elapsed_s_mean"] <= two["ranks"]["elapsed_s_max"]
    assert abs(two["ranks"]["elapsed_s_max"] * 1e3 / two["steps"] - two["ms_per_step"]) < 0.5 * two["ms_per_step"]
    assert two["scaling"] == "strong" and one["ranks"]["per_rank"][0]["sequences"] == [0, 1, 2, 3]
    # weak scaling: --batch is per GPU (2 x 2 sequences = the same 4 sequences, the same result)
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--weak"] +
                       [("2" if a == "4" and common[i - 1] == "--batch" else a) for i, a in enumerate(common)],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    weak = json.loads(r.stdout.strip().splitlines()[-1])
    assert weak["scaling"] == "weak" and weak["config"]["sequences_total"] == 4
    assert weak["config"]["poses_sha"] == one["config"]["poses_sha"]


def test_eight_ranks_sharing_one_gpu_rehearsal():
    """The metric's own launch shape -- 8 sequences, one per rank -- rehearsed on ONE GPU (eight self-spawned ranks share
    cuda:0 over gloo; tiny frames): every rank's fingerprint of its poses must be the single-rank run's fingerprint of
    that sequence, the gathered result must be the single-rank result, and rank 0 reports the gather on its own clock."""
    import json
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--steps", "2", "--warmup", "1", "--batch", "8", "--height", "60", "--width", "80", "--no-cpu-baseline",
              "--no-roofline-pass", "--no-secondary"]
    env = dict(os.environ, GRADSLAM_DIST_BACKEND="gloo", GRADSLAM_BENCH_SHARE_GPU="1")
    env.pop("WORLD_SIZE", None)
    lines = []
    for n in ("8", "1"):
        r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", n] + common, capture_output=True,
                           text=True, timeout=900, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        lines.append(json.loads(r.stdout.strip().splitlines()[-1]))
    eight, one = lines
    assert eight["n_gpus"] == 8 and eight["config"]["sequences_per_gpu"] == 1 and eight["config"]["sequences_total"] == 8
    assert eight["config"]["poses_sha"] == one["config"]["poses_sha"]
    assert eight["config"]["map_surfels_all"] == one["config"]["map_surfels_all"]
    pr = eight["ranks"]["per_rank"]
    assert [r["rank"] for r in pr] == list(range(8)) and [r["sequences"] for r in pr] == [[b] for b in range(8)]
    # per-sequence fingerprints: what rank r computed alone == what the one-rank batch computed for sequence r
    assert [r["poses_sha"] for r in pr] == one["config"]["poses_sha_by_sequence"]
    assert eight["config"]["final_gather_ms"] >= 0.0 and eight["config"]["gather_peak_bytes_rank0"] > 0


_AB_SCRIPT = r"""
import sys
import numpy as np, torch
sys.path.insert(0, %r)
import gradslam_amd as gs
from gradslam_amd.datasets.synthetic import make_sequence
B, L, H, W = 3, 5, 120, 160
seqs = [make_sequence(L, H, W, seed=11 + b) for b in range(B)]
st = lambda k: torch.from_numpy(np.stack([s[k] for s in seqs])).cuda()
poses = st("poses"); poses[:, 1:] = poses[:, :1]
frames = gs.RGBDImages(st("colors"), st("depths"), st("intrinsics"), poses)
slam = gs.slam.PointFusion(odom="gradicp", device="cuda")
pc, prev, rec = gs.Pointclouds(device="cuda"), None, []
for i in range(L):
    live = frames[:, i]
    pc, p = slam.step(pc, live, prev, inplace=True)
    prev = live
    rec.append(p[:, 0].cpu().numpy())
np.savez(sys.argv[1], poses=np.stack(rec), n=np.array([int(x.shape[0]) for x in pc.points_list]),
         pts=np.concatenate([x.cpu().numpy() for x in pc.points_list]), nrm=np.concatenate([x.cpu().numpy() for x in pc.normals_list]),
         cc=np.concatenate([x.cpu().numpy() for x in pc.features_list]))
"""


def test_fast_path_equals_generic_path(tmp_path):
    """PointFusion.step(..., inplace=True) goes through slam/_fastpath.py (one foreign call per frame) from the second
    frame on; GRADSLAM_HIP_FASTPATH=0 takes the generic three-call path (_localize + update_map_fusion).  Same kernels
    in the same order: poses, counts, points, normals and confidence counts must be bit-equal.  (Keeps the generic
    path covered: every other in-place test runs the fast path.)"""
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for fast in ("1", "0"):
        out = str(tmp_path / ("fast%s.npz" % fast))
        subprocess.run([sys.executable, "-c", _AB_SCRIPT % repo, out], check=True, timeout=600,
                       env=dict(os.environ, GRADSLAM_HIP_FASTPATH=fast))
        outs.append(np.load(out))
    a, b = outs
    for k in ("poses", "pts", "nrm", "cc"):
        assert np.array_equal(a[k].view(np.int32), b[k].view(np.int32)), k
    assert np.array_equal(a["n"], b["n"])


def test_step_without_materialised_global_maps_gives_the_same_bits(gs, monkeypatch):
    """gs_pointfusion_step_batch_f32 with gvertex = gnormal = NULL (the default of the fused step): the frame-map launch
    initialises the update's per-pixel tables and the projection / merge / append passes transform a pixel where they
    use it, so the global maps are never written.  Poses, map and -- computed on demand by the container -- the global
    maps themselves must equal, bit for bit, the step that materialises them (GRADSLAM_HIP_STEP_GLOBAL_MAPS=1)."""
    from gradslam_amd.slam import _fastpath
    B, L, H, W = 2, 5, 120, 160
    seqs = [make_sequence(L, H, W, seed=41 + b) for b in range(B)]

    def run(materialise):
        monkeypatch.setattr(_fastpath, "STEP_GLOBAL_MAPS", materialise)
        frames = frames_of(gs, seqs)
        slam = gs.slam.PointFusion(odom="gradicp", device="cuda")
        pc, prev, rec = gs.Pointclouds(device="cuda"), None, []
        for i in range(L):
            live = frames[:, i]
            pc, p = slam.step(pc, live, prev, inplace=True)
            if i >= 2:   # (frame 0 has no previous frame, frame 1 finds the counts not yet grouped: generic path)
                assert (live._global_vertex_map is not None) == materialise
            rec += [host(p[:, 0]), host(live.global_vertex_map), host(live.global_normal_map)]
            prev = live
        return rec + [np.concatenate([host(x) for x in getattr(pc, k)]) for k in
                      ("points_list", "normals_list", "colors_list", "features_list")]

    a, b = run(False), run(True)
    for x, y in zip(a, b):
        assert x.shape == y.shape and np.array_equal(x.view(np.int32), y.view(np.int32))


def test_one_slam_object_steps_two_maps_alternately(gs):
    """the device count buffers of the fast path belonged to the plan cached on the slam object, so a second
    map stepped with the same object overwrote the first map's live count.  They belong to the map now: two maps
    stepped alternately with ONE slam object must equal the same maps stepped on their own."""
    L, H, W = 4, 120, 160
    sa, sb = [make_sequence(L, H, W, seed=21)], [make_sequence(L, H, W, seed=22)]
    fa, fb = frames_of(gs, sa), frames_of(gs, sb)

    def alone(frames):
        slam = gs.slam.PointFusion(odom="gradicp", device="cuda")
        pc, prev, rec = gs.Pointclouds(device="cuda"), None, []
        for i in range(L):
            live = frames[:, i]
            pc, p = slam.step(pc, live, prev, inplace=True)
            prev = live
            rec.append(host(p[0, 0]))
        return np.stack(rec), host(pc.points_list[0]), host(pc.features_list[0])

    ra, rb = alone(fa), alone(fb)
    slam = gs.slam.PointFusion(odom="gradicp", device="cuda")
    pcs, prevs, recs = [gs.Pointclouds(device="cuda"), gs.Pointclouds(device="cuda")], [None, None], [[], []]
    for i in range(L):
        for m, frames in enumerate((fa, fb)):
            live = frames[:, i]
            pcs[m], p = slam.step(pcs[m], live, prevs[m], inplace=True)
            prevs[m] = live
            recs[m].append(host(p[0, 0]))
    for m, ref in enumerate((ra, rb)):
        assert np.array_equal(np.stack(recs[m]).view(np.int32), ref[0].view(np.int32))
        assert np.array_equal(host(pcs[m].points_list[0]).view(np.int32), ref[1].view(np.int32))
        assert np.array_equal(host(pcs[m].features_list[0]).view(np.int32), ref[2].view(np.int32))


def test_attribute_setter_between_steps_is_followed_by_the_fast_path(gs):
    """a setter of normals / colors / features gives the map a new buffer for that attribute only; the plan
    compared the points buffer alone and went on writing the old ones.  After `pc.colors_list = ...` between two steps
    the map must hold the new colours where no surfel was merged -- and the run must equal one that takes the generic
    path for that frame (a fresh slam object has no plan yet)."""
    L, H, W = 4, 120, 160
    frames = frames_of(gs, [make_sequence(L, H, W, seed=23)])

    def run(reuse_plan):
        slam = gs.slam.PointFusion(odom="gradicp", device="cuda")
        pc, prev = gs.Pointclouds(device="cuda"), None
        for i in range(L):
            if i == 2:
                pc.colors_list = [c * 0.5 + 1.0 for c in pc.colors_list]
                pc.normals_list = [n.clone() for n in pc.normals_list]
                if not reuse_plan:
                    slam = gs.slam.PointFusion(odom="gradicp", device="cuda")
            live = frames[:, i]
            pc, _ = slam.step(pc, live, prev, inplace=True)
            prev = live
        return host(pc.points_list[0]), host(pc.colors_list[0]), host(pc.normals_list[0])

    a, b = run(True), run(False)
    for x, y in zip(a, b):
        assert np.array_equal(x.view(np.int32), y.view(np.int32))


def test_forward_is_bitwise_reproducible_run_to_run(gs):
    """SURVEY section 5 / VERDICT r03 #10: the same five frames of 8 sequences twice in one process (fresh maps, the same
    slam object, warm allocator the second time): poses and maps bit for bit.  Every reduction on the forward path has
    a fixed order (row units, float64 sums in index order, 64-bit atomicMin keys), so nothing depends on scheduling."""
    B, L, H, W = 8, 5, 240, 320
    frames = frames_of(gs, [make_sequence(L, H, W, seed=40 + b) for b in range(B)])
    slam = gs.slam.PointFusion(odom="gradicp", device="cuda")

    def run():
        pc, prev, rec = gs.Pointclouds(device="cuda"), None, []
        for i in range(L):
            live = frames[:, i]
            pc, p = slam.step(pc, live, prev, inplace=True)
            prev = live
            rec.append(host(p[:, 0]))
        return np.stack(rec), [host(x) for x in pc.points_list], [host(x) for x in pc.features_list]

    a, b = run(), run()
    assert np.array_equal(a[0].view(np.int32), b[0].view(np.int32))
    for x, y in zip(a[1] + a[2], b[1] + b[2]):
        assert x.shape == y.shape and np.array_equal(x.view(np.int32), y.view(np.int32))


def test_streamed_frames_give_the_resident_result(gs):
    """SURVEY section 8 f3 / VERDICT r03 #6: raw uint16 depth + uint8 colour in pinned host memory -> asynchronous copies
    on a copy stream -> gs_ingest_frames_native_f32 -> PointFusion.step, frame t + 1 in flight while step t computes
    (gradslam_amd/datasets/streaming.py).  The result must be bit for bit that of the same (quantised) frames resident
    in HBM, and the one-launch conversion must equal the per-frame entry points."""
    from gradslam_amd import ops
    from gradslam_amd.datasets.streaming import FrameStreamer, quantize_sequences
    B, L, H, W = 3, 6, 120, 160
    seqs = [make_sequence(L, H, W, seed=31 + b) for b in range(B)]
    d16, c8 = quantize_sequences(seqs, 5000.0)
    assert d16.is_pinned() and c8.is_pinned() and tuple(d16.shape) == (L, B, H, W)   # time-major
    # resident reference: the per-frame ingest entry points (the loaders' path)
    dres = torch.stack([torch.stack([ops.ingest_depth(d16[t, b].cuda(), H, W, 5000.0) for t in range(L)]) for b in range(B)])
    cres = torch.stack([torch.stack([ops.ingest_color(c8[t, b].cuda(), H, W) for t in range(L)]) for b in range(B)])
    K = T(np.stack([s["intrinsics"] for s in seqs])).cuda()
    P0 = T(np.stack([s["poses"][:1] for s in seqs])).cuda()
    frames = gs.RGBDImages(cres, dres[..., None], K, P0.repeat(1, L, 1, 1))

    def run(get):
        slam = gs.slam.PointFusion(odom="gradicp", device="cuda")
        pc, prev, rec = gs.Pointclouds(device="cuda"), None, []
        for t in range(L):
            live = get(t)
            pc, p = slam.step(pc, live, prev, inplace=True)
            prev = live
            rec.append(host(p[:, 0]))
        return np.stack(rec), [host(x) for x in pc.points_list], [host(x) for x in pc.colors_list]

    ref = run(lambda t: frames[:, t])
    st = FrameStreamer(d16, c8, K, P0, scale_div=5000.0, device="cuda")
    first = st.frame(0)
    assert torch.equal(first.depth_image, dres[:, :1, ..., None]) and torch.equal(first.rgb_image, cres[:, :1])
    for zero_copy in (False, True):   # hipMemcpyAsync + conversion (default) / the kernel reads the pinned host frames itself
        st2 = FrameStreamer(d16, c8, K, P0, scale_div=5000.0, device="cuda", zero_copy=zero_copy)
        assert st2.zero_copy == zero_copy
        got = run(st2.frame)
        assert np.array_equal(ref[0].view(np.int32), got[0].view(np.int32))
        for a, b in zip(ref[1] + ref[2], got[1] + got[2]):
            assert a.shape == b.shape and np.array_equal(a.view(np.int32), b.view(np.int32))
    with pytest.raises(ValueError, match="pinned host memory"):
        FrameStreamer(d16.clone(), c8, K, P0, 5000.0)


_FAR_SCRIPT = r"""
import sys
import numpy as np, torch
sys.path.insert(0, %r)
import gradslam_amd as gs
from gradslam_amd import ops
from gradslam_amd.datasets.synthetic import make_sequence
L, H, W = 4, 240, 320
s = make_sequence(L, H, W, seed=5)
depths = s["depths"].copy()
depths[0, :, int(0.62 * W):] = 0.0          # the map of frame 0 does not cover the right part of the view:
T = torch.from_numpy                         # the source points there are far from every target
poses = s["poses"].copy(); poses[1:] = poses[:1]
frames = gs.RGBDImages(T(s["colors"][None]).cuda(), T(depths[None]).cuda(), T(s["intrinsics"][None]).cuda(), T(poses[None]).cuda())
slam = gs.slam.PointFusion(odom="gradicp", device="cuda")
pc, prev, rec, stats = gs.Pointclouds(device="cuda"), None, [], []
for i in range(L):
    live = frames[:, i]
    pc, p = slam.step(pc, live, prev, inplace=True)
    prev = live
    rec.append(p[0, 0].cpu().numpy())
    if i == 1:   # the first localisation: the map is frame 0 only
        stats = ops.localize_far_stats(torch.device("cuda", 0), 0, H, W, 4, pc._buf["points"][0].shape[0])
np.savez(sys.argv[1], poses=np.stack(rec), n=np.array([int(pc.points_list[0].shape[0])]), pts=pc.points_list[0].cpu().numpy(),
         far=np.array(stats))
"""


def test_far_candidate_lists_leave_results_identical(tmp_path):
    """Source points far from every target (here: 38 % of the view is missing from the map) get candidate lists after
    the first search of a solve (gs_icp_far_build_kernel) and are then served by 16 gathers instead of a cube scan or
    a pass over all targets (GRADSLAM_HIP_ICP_FAR=1 switches the lists on: opt-in since round 5, when ordinary + wide lists
    became the faster choice at every size).  GRADSLAM_HIP_ICP_FAR=0 runs the same frames without lists: poses and the map must be
    bit-identical, and the lists must actually have been in use.  (This scene is hostile on purpose: half of the far
    points have more than 16 targets within reach or move out of their list's radius, and fall back to the scans.)"""
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for far in ("1", "0"):
        out = str(tmp_path / ("far%s.npz" % far))
        subprocess.run([sys.executable, "-c", _FAR_SCRIPT % repo, out], check=True, timeout=600,
                       env=dict(os.environ, GRADSLAM_HIP_ICP_FAR=far))
        outs.append(np.load(out))
    a, b = outs
    assert a["far"][0] > 300, a["far"]                 # far source points found by the first search ...
    assert a["far"][1] > 50, a["far"]                  # ... some of which still prove on their list at the last search
    assert b["far"][0] == 0
    assert np.array_equal(a["poses"].view(np.int32), b["poses"].view(np.int32))
    assert a["n"][0] == b["n"][0]
    assert np.array_equal(a["pts"].view(np.int32), b["pts"].view(np.int32))


_LIST_SCRIPT = r"""
import sys
import numpy as np, torch
sys.path.insert(0, %r)
import gradslam_amd as gs
from gradslam_amd import ops
from gradslam_amd.datasets.synthetic import make_sequence
B, L, H, W = int(sys.argv[2]), 4, int(sys.argv[3]), int(sys.argv[4])
first = int(sys.argv[5]) if len(sys.argv) > 5 else 0
seqs = [make_sequence(L, H, W, seed=3 + b, first=first) for b in range(B)]
st = lambda k: torch.from_numpy(np.stack([s[k] for s in seqs])).cuda()
poses = st("poses"); poses[:, 1:] = poses[:, :1]
frames = gs.RGBDImages(st("colors"), st("depths"), st("intrinsics"), poses)
slam = gs.slam.PointFusion(odom="gradicp", device="cuda")
pc, prev, rec, stats = gs.Pointclouds(device="cuda"), None, [], []
for i in range(L):
    live = frames[:, i]
    pc, p = slam.step(pc, live, prev, inplace=True)
    prev = live
    rec.append(p[:, 0].cpu().numpy())
    if i >= 1:
        stats.append([ops.localize_list_stats(torch.device("cuda", 0), b, H, W, 4, pc._buf["points"][b].shape[0]) for b in range(B)])
np.savez(sys.argv[1], poses=np.stack(rec), n=np.array([int(x.shape[0]) for x in pc.points_list]),
         pts=np.concatenate([x.cpu().numpy() for x in pc.points_list]), stats=np.array(stats))
"""


@pytest.mark.parametrize("B,H,W", [(1, 480, 640), (8, 480, 640), (3, 240, 320), (2, 67, 131)])
def test_candidate_lists_leave_results_identical(tmp_path, B, H, W):
    """Round 4: the half-iteration kernels keep a candidate list per source point (every target within R of where the
    point was searched from) and, from the third launch of a solve on, try the list before any search
    (gs_knn.h: gl_*; the listed points are fetched while the prologue waits for the partial rows).  A proof on the list
    is exact and everything else is the search that ran before, so GRADSLAM_HIP_ICP_LISTS=0 must give the same bits:
    poses, surfel counts, points -- at 8 / 4 / 2 lanes per source point (one, three / two, eight sequences per GPU).
    And the lists must carry the solve: in its second half no launch may re-search more than 2 % of the points."""
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for on in ("1", "0"):
        out = str(tmp_path / ("lists%s.npz" % on))
        # (lists from the first iteration on: by default they start at iteration 6, when the solve has calmed down; the
        # early start makes thousands of lists fail and be rebuilt, which is the code this test is after)
        subprocess.run([sys.executable, "-c", _LIST_SCRIPT % repo, out, str(B), str(H), str(W)], check=True, timeout=900,
                       env=dict(os.environ, GRADSLAM_HIP_ICP_LISTS=on, GRADSLAM_HIP_ICP_LISTS_FROM="0"))
        outs.append(np.load(out))
    a, b = outs
    assert np.array_equal(a["poses"].view(np.int32), b["poses"].view(np.int32))
    assert np.array_equal(a["n"], b["n"])
    assert np.array_equal(a["pts"].view(np.int32), b["pts"].view(np.int32))
    assert not b["stats"].any()                                  # no lists, no counters
    st = a["stats"]                                              # (frames - 1, B, 3, 64)
    n_lat = ((H + 3) // 4) * ((W + 3) // 4)
    assert st[:, :, :, :2].sum() == 0                            # launches 0 and 1 try no lists
    late = st[:, :, :, 20:40].sum(axis=2)                        # failed + empty + without a list, launches 20 .. 39
    assert late.max() <= 0.02 * n_lat, late.max()
    if n_lat >= 19200:
        assert st[:, :, 0, 2:40].sum() > 0                       # some list failed somewhere (the counters are alive)


@pytest.mark.parametrize("B", [8, 1, 4])
def test_wide_lists_of_hard_queries_leave_results_identical(tmp_path, B):
    """Round 5: a source point several cells from every target (a frame border that looks at a surface under a grazing
    angle: frames 85 .. 88 of the benchmark's camera path, where neighbouring lattice pixels of the right image border
    are 15 - 30 cm apart) is served by cube scans or a block-wide pass over all targets; whatever serves it leaves a
    64-slot list that the 16-lane group checks first in every later launch (gs_knn.h: far_write_from_top,
    block_brute_min_list_multi, wide_list_search).  A proof on the list is exact, so GRADSLAM_HIP_ICP_WIDE=0 must give
    the same bits at 2 / 8 / 4 lanes per source point -- and the scene must really have such points (launches in which
    points have no ordinary list because the 2x2x2 stage cannot prove them)."""
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for on in ("1", "0"):
        out = str(tmp_path / ("wide%s.npz" % on))
        subprocess.run([sys.executable, "-c", _LIST_SCRIPT % repo, out, str(B), "480", "640", "85"], check=True, timeout=900,
                       env=dict(os.environ, GRADSLAM_HIP_ICP_WIDE=on))
        outs.append(np.load(out))
    a, b = outs
    assert np.array_equal(a["poses"].view(np.int32), b["poses"].view(np.int32))
    assert np.array_equal(a["n"], b["n"])
    assert np.array_equal(a["pts"].view(np.int32), b["pts"].view(np.int32))
    # (the counters of the ordinary lists differ: a point its wide list serves keeps no ordinary one)
    assert a["stats"][:, :, 2, 4:40].sum() > 0               # points without an ordinary list in list-checking launches


def test_lanes_per_point_and_block_share_leave_results_identical(tmp_path):
    """The launch geometry of the ICP half-iterations -- lanes per source point (GRADSLAM_HIP_ICP_LANES) and blocks per CU the
    planner aims at (GRADSLAM_HIP_ICP_BLOCKS_PER_CU; round 6: one whenever a single group of row units per block fits it) --
    is not an input of the result: the row unit of 96 points is the granule of every sum (csrc/gs_icp_loop.hip:
    icp_half_plan).  Two sequences of 480x640 at 8 / 4 / 2 lanes and the old / new share: the same poses, counts, points."""
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for lanes, share in (("", ""), ("8", ""), ("4", ""), ("2", ""), ("", "2")):
        out = str(tmp_path / ("lanes%s_%s.npz" % (lanes, share)))
        env = dict(os.environ)
        env.pop("GRADSLAM_HIP_ICP_LANES", None)
        env.pop("GRADSLAM_HIP_ICP_BLOCKS_PER_CU", None)
        if lanes:
            env["GRADSLAM_HIP_ICP_LANES"] = lanes
        if share:
            env["GRADSLAM_HIP_ICP_BLOCKS_PER_CU"] = share
        subprocess.run([sys.executable, "-c", _LIST_SCRIPT % repo, out, "2", "480", "640"], check=True, timeout=900, env=env)
        outs.append(np.load(out))
    a = outs[0]
    for b in outs[1:]:
        assert np.array_equal(a["poses"].view(np.int32), b["poses"].view(np.int32))
        assert np.array_equal(a["n"], b["n"])
        assert np.array_equal(a["pts"].view(np.int32), b["pts"].view(np.int32))


@pytest.mark.parametrize("B,first", [(8, 0), (1, 0), (3, 85)])
def test_persistent_xcd_solve_leaves_results_identical(tmp_path, B, first):
    """Round 6 (opt-in, GRADSLAM_HIP_ICP_PERSIST=1): the list-checking half-iterations of a solve as ONE persistent launch per
    sequence, resident on one XCD (csrc/gs_icp_persist.h: blocks find their XCD at run time, partial rows by plain stores +
    one L2 atomic per half-iteration, readers bypass their L1; source points, lists and listed targets stay in registers /
    LDS across the half-iterations).  Every sum keeps its order and every search its arithmetic, so the bits must be those
    of the launch-per-half-iteration path: poses, surfel counts, points -- at 8 / 1 / 3 sequences per GPU, the last on
    frames 85 .. 88 of the camera path, where points without a provable list (cube scans, wide lists, block passes) exist."""
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for on in ("1", "0"):
        out = str(tmp_path / ("persist%s.npz" % on))
        subprocess.run([sys.executable, "-c", _LIST_SCRIPT % repo, out, str(B), "480", "640", str(first)], check=True, timeout=900,
                       env=dict(os.environ, GRADSLAM_HIP_ICP_PERSIST=on))
        outs.append(np.load(out))
    a, b = outs
    assert np.isfinite(a["poses"]).all()                         # (a block that gives up waiting leaves a NaN pose)
    assert np.array_equal(a["poses"].view(np.int32), b["poses"].view(np.int32))
    assert np.array_equal(a["n"], b["n"])
    assert np.array_equal(a["pts"].view(np.int32), b["pts"].view(np.int32))


def test_pointfusion_1296x968_vs_reference_golden(gs, golden):
    """BASELINE configs[4] resolution against the REAL reference: 3 frames of PointFusion(gradicp, 20 iterations) at
    1296x968 (tests/golden/pf1296_s3.npz, oracle/make_golden_640.py --height 968 --width 1296 --seed 3; minutes of CPU
    per frame there): pose ATE <= 1e-4 m, identical first frame, surfel counts within 0.05 %, point sums as at
    640x480.  78k ICP source points per frame (ordinary + wide candidate lists; 4 lanes per point)."""
    g = golden("pf1296_s3")
    L, H, W = int(g["poses"].shape[0]), int(g["H"]), int(g["W"])
    assert (H, W) == (968, 1296)
    s = make_sequence(L, H, W, seed=int(g["seed"]))
    assert abs(float(s["depths"].astype(np.float64).sum()) - float(g["depth_sum"])) < 1e-6 * float(g["depth_sum"])
    frames = frames_of(gs, [s])
    slam = gs.slam.PointFusion(odom="gradicp", device="cuda")
    pc, prev, counts, sums, rec = gs.Pointclouds(device="cuda"), None, [], [], []
    for f in range(L):
        live = frames[:, f]
        pc, pose = slam.step(pc, live, prev, inplace=True)
        prev = live
        rec.append(host(pose[0, 0]))
        counts.append(pc.points_list[0].shape[0])
        sums.append(host(pc.points_list[0].double().sum(0)))
    rec = np.stack(rec)
    assert ate(rec, g["poses"]) <= 1e-4, ate(rec, g["poses"])
    diff = np.abs(np.asarray(counts) - g["counts"])
    assert counts[0] == g["counts"][0]
    assert diff.max() <= 5e-4 * g["counts"][-1], (counts, g["counts"].tolist())
    for f in range(L):
        np.testing.assert_allclose(sums[f], g["sum_points"][f], rtol=0, atol=1e-5 * counts[f] + 4.0 * diff[f] + 1e-3)


def test_pointfusion_1296x968_vs_oracle(gs):
    """BASELINE configs[4] shape (ScanNet resolution): 3 frames of PointFusion(gradicp, numiters=6) against the oracle's
    frame loop: 78k ICP queries against ~100k+ binned targets per solve, a map beyond 1.5M surfels: poses within 2e-6,
    identical surfel counts, points within 1e-5."""
    from oracle import slam as oslam
    L, H, W = 3, 968, 1296
    s = make_sequence(L, H, W, seed=3)
    frames = frames_of(gs, [s])
    pc, rp = gs.slam.PointFusion(odom="gradicp", numiters=6, device="cuda")(frames)
    poses = s["poses"].copy()
    poses[1:] = poses[:1]
    m, op = oslam.run_sequence(s["colors"], s["depths"], s["intrinsics"][0], poses, numiters=6)
    np.testing.assert_allclose(host(rp[0]), op, rtol=0, atol=2e-6)
    assert pc.points_list[0].shape[0] == len(m) > 1_500_000
    np.testing.assert_allclose(host(pc.points_list[0]), m.points, rtol=1e-5, atol=1e-5)


# Long horizons against the REAL reference.  `calm`: frames for which the reference reproduces ITSELF (its run with an intra-op
# pool of 3 threads against its run with 8: another summation order of its float32 matrix products, nothing else) to well
# below BASELINE's 1e-4 m; there the build is held to BASELINE's bound.  Beyond, the bound is DERIVED from the committed
# record of that comparison (`sens`): every pose within K_SENS x the running maximum of the reference's own
# deviation from itself.  On every horizon the HIP path must reproduce the ORACLE's trajectory bit for bit
# (tests/golden/<name>_oracle.npz, oracle/make_golden_oracle_long.py): the statement that stays exact where the reference
# is chaotic.
#   pf640_l60        benchmark scene ("wave"): the reference's solves stop settling within their 20 iterations at frame 28
#   facets640_l60    inclined planes + ridge WITH 5 % zeroed depth pixels: every solve converges, and the reference still
#                    drifts from itself from frame 5 on (garbage normals next to the zeroed pixels, amplified by the fused map)
#   facets640_nh_l60 the same scene WITHOUT zeroed pixels: the reference agrees with itself to 4e-6 m over all 60 frames --
#                    the scene on which BASELINE's ATE <= 1e-4 m is asserted over the whole horizon (VERDICT r05 #1b)
#   pf1296_s3_l20    1296x968, 20 frames, 3.5 M surfels
K_SENS = 4.0
_LONG_HORIZON = {"pf640_l60": dict(calm=28, ate_calm=1e-5, drift_calm=300, sens="reference_sensitivity_640.json"),
                 "facets640_l60": dict(calm=5, ate_calm=1e-5, drift_calm=100, sens="reference_sensitivity_facets640_l60.json"),
                 "facets640_nh_l60": dict(calm=60, ate_calm=2e-5, drift_calm=300, sens=None),
                 "pf1296_s3_l20": dict(calm=20, ate_calm=1e-5, drift_calm=1500, sens=None)}


@pytest.mark.parametrize("name", ["pf640_l60", "facets640_nh_l60", "facets640_l60", "pf1296_s3_l20"])
def test_pointfusion_long_horizon_vs_reference_golden(gs, golden, name):
    """The long horizon against the REAL reference (goldens recorded by oracle/make_golden_640.py from the imported
    reference) and, bit for bit, against the oracle (oracle/make_golden_oracle_long.py).  While the reference reproduces
    itself (`calm` frames; all 60 on the scene without zeroed pixels): pose ATE <= 1e-4 m (BASELINE.json), every pose within
    1e-4, count drift within the measured bound, mean surfel position within 1e-4.  Beyond: every pose within K_SENS x the
    reference's own deviation from itself (its 3-thread against its 8-thread run, committed), and as close to the ground
    truth as the reference is."""
    import json
    import os
    from gradslam_amd import metrics as M
    g = golden(name)
    L, H, W = int(g["poses"].shape[0]), int(g["H"]), int(g["W"])
    scene = str(g["scene"]) if "scene" in g.files else "wave"
    hole = float(g["hole_frac"]) if "hole_frac" in g.files else 0.05
    s = make_sequence(L, H, W, seed=int(g["seed"]), scene=scene, hole_frac=hole)
    assert abs(float(s["depths"].astype(np.float64).sum()) - float(g["depth_sum"])) < 1e-6 * float(g["depth_sum"])
    frames = frames_of(gs, [s])
    slam = gs.slam.PointFusion(odom="gradicp", device="cuda")
    pc, prev, counts, sums, rec = gs.Pointclouds(device="cuda"), None, [], [], []
    for f in range(L):
        live = frames[:, f]
        pc, pose = slam.step(pc, live, prev, inplace=True)
        prev = live
        rec.append(host(pose[0, 0]))
        counts.append(pc.points_list[0].shape[0])
        sums.append(host(pc.points_list[0].double().sum(0)))
    rec = np.stack(rec)
    b = _LONG_HORIZON[name]
    c = b["calm"]
    a_calm, a_all = M.ate_rmse(rec[:c], g["poses"][:c]), M.ate_rmse(rec, g["poses"])
    d = M.count_drift(counts, g["counts"])
    dev = np.abs(rec - g["poses"]).reshape(L, -1).max(1)
    err_hip = np.linalg.norm(rec[:, :3, 3].astype(np.float64) - g["gt_poses"][:, :3, 3], axis=1)
    err_ref = np.linalg.norm(g["poses"][:, :3, 3].astype(np.float64) - g["gt_poses"][:, :3, 3], axis=1)
    gdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    opath = os.path.join(gdir, name + "_oracle.npz")
    rec_dir = os.environ.get("GRADSLAM_TEST_RECORD")
    if rec_dir:
        with open(os.path.join(rec_dir, "long_horizon_%s.json" % name), "w") as fh:
            json.dump({"ate_calm_m": a_calm, "calm_frames": c, "ate_all_m": a_all, "rpe_all": M.rpe(rec, g["poses"]),
                       "pose_abs_diff_per_frame": dev.tolist(), "count_drift": d,
                       "error_vs_ground_truth_m_hip": err_hip.tolist(), "error_vs_ground_truth_m_reference": err_ref.tolist(),
                       "oracle_golden": os.path.exists(opath)}, fh)
    # the oracle's trajectory, bit for bit, over the whole horizon
    if os.path.exists(opath):
        og = np.load(opath)
        assert np.array_equal(rec.view(np.int32), og["poses"].view(np.int32)), np.abs(rec - og["poses"]).reshape(L, -1).max(1)
        assert np.array_equal(np.asarray(counts, np.int64), og["counts"])
    else:
        assert name == "pf1296_s3_l20"   # (the 1296x968 oracle run is tests/test_hip_batch.py::test_pointfusion_1296x968_vs_oracle)
    # while the reference reproduces itself: BASELINE's bound
    assert a_calm <= 1e-4 and a_calm <= b["ate_calm"], a_calm
    np.testing.assert_allclose(rec[:c], g["poses"][:c], rtol=0, atol=1e-4)
    assert counts[0] == int(g["counts"][0])
    assert max(d["per_frame"][:c]) <= b["drift_calm"] and max(d["per_frame"][:c]) <= 5e-4 * int(g["counts"][c - 1]), d
    for f in range(c):   # mean surfel position: the maps are the same cloud up to the few rows that differ (3 m: the scene's depth)
        np.testing.assert_allclose(sums[f] / counts[f], g["sum_points"][f] / float(g["counts"][f]), rtol=0,
                                   atol=1e-4 + 3.0 * abs(counts[f] - int(g["counts"][f])) / float(g["counts"][f]))
    # beyond: within K_SENS x the reference's deviation from itself, frame by frame (running maximum; the record of the
    # benchmark scene ends at frame 35: its maximum stands for the frames behind it)
    if c < L:
        with open(os.path.join(gdir, b["sens"])) as fh:
            sens = np.asarray(json.load(fh)["pose_abs_diff_per_frame"], np.float64)
        env = np.maximum.accumulate(sens)
        env = np.concatenate([env, np.full(max(L - len(env), 0), env[-1])])[:L]
        assert (dev[c:] <= K_SENS * env[c:] + 1e-4).all(), (dev[c:] / (env[c:] + 1e-12)).max()
        assert err_hip.max() <= 1.25 * err_ref.max() + 1e-4, (err_hip.max(), err_ref.max())   # as close to the truth as the reference
