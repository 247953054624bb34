"""Goldens of the REAL reference (gradslam v0.1.0 imported from /root/reference, oracle/refimport.py) for the API-level
helpers that are not on the SLAM hot path but belong to its drop-in surface:

  geometry.projutils.project_points / unproject_points   (geometry/projutils.py:92-402), several broadcasting cases
  geometry.se3utils.so3_hat / se3_hat / so3_exp           (geometry/se3utils.py:11-74)
  Pointclouds.offset_ / scale_ / rotate_ / transform_ / pinhole_projection_ and + - * / @
                                                          (structures/pointclouds.py:300-384, 399-614)

    python -m oracle.make_golden_api        ->  tests/golden/api_helpers.npz

Build-container only; the inputs are stored next to the outputs (they are tiny)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, REPO)
from oracle import refimport  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden", "api_helpers.npz")


def main():
    refimport.import_reference()
    import torch
    from gradslam.geometry import projutils as P
    from gradslam.geometry import se3utils as S
    from gradslam.structures.pointclouds import Pointclouds

    g = torch.Generator().manual_seed(7)
    r = lambda *s: torch.rand(*s, generator=g)   # noqa: E731
    out = {}

    # ---- project_points: (N,3)x(4,4), (B,P,3)x(4,4), (B,P,3)x(B,4,4), (B,P,4)x(B,4,4); a zero depth in each
    cases = [("a", r(10, 3) + 0.5, r(4, 4)), ("b", r(2, 10, 3) + 0.5, r(4, 4)), ("c", r(2, 10, 3) + 0.5, r(2, 4, 4)),
             ("d", r(2, 10, 4) + 0.5, r(2, 4, 4))]
    for name, cam, proj in cases:
        proj = proj.clone()
        if name == "a":
            proj[2] = 0.0   # z' == 0 for every point: division by 1
        out["pp_%s_cam" % name], out["pp_%s_proj" % name] = cam.numpy(), proj.numpy()
        out["pp_%s_out" % name] = P.project_points(cam, proj).numpy()
    # ---- unproject_points
    cases = [("a", r(10, 2) * 100, r(3, 3), r(10) + 0.5), ("b", r(2, 10, 2) * 100, r(2, 3, 3), r(2, 10) + 0.5),
             ("c", r(2, 10, 3) * 100, r(3, 3), r(2, 10) + 0.5)]
    for name, pix, kinv, d in cases:
        out["up_%s_pix" % name], out["up_%s_kinv" % name], out["up_%s_depth" % name] = pix.numpy(), kinv.numpy(), d.numpy()
        out["up_%s_out" % name] = P.unproject_points(pix, kinv, d).numpy()
    # ---- so3 / se3
    omegas = torch.stack([r(3) - 0.5, (r(3) - 0.5) * 3.0, (r(3) - 0.5) * 1e-8, torch.tensor([0.0, 0.0, 3.1])])
    xis = r(3, 6) - 0.5
    out["omega"], out["xi"] = omegas.numpy(), xis.numpy()
    out["so3_hat"] = torch.stack([S.so3_hat(o) for o in omegas]).numpy()
    out["so3_exp"] = torch.stack([S.so3_exp(o) for o in omegas]).numpy()
    out["se3_hat"] = torch.stack([S.se3_hat(x) for x in xis]).numpy()

    # ---- Pointclouds algebra on a ragged batch of two clouds (5 and 7 points)
    pts = [r(5, 3) + 1.0, r(7, 3) + 1.0]
    nrm = [torch.nn.functional.normalize(r(5, 3) - 0.5, dim=-1), torch.nn.functional.normalize(r(7, 3) - 0.5, dim=-1)]
    for b in range(2):
        out["pc_points%d" % b], out["pc_normals%d" % b] = pts[b].numpy(), nrm[b].numpy()

    def fresh():
        return Pointclouds(points=[p.clone() for p in pts], normals=[n.clone() for n in nrm])

    def rec(tag, pc):
        for b in range(2):
            out["%s_p%d" % (tag, b)] = pc.points_list[b].numpy()
            out["%s_n%d" % (tag, b)] = pc.normals_list[b].numpy()

    def rot(seed):
        q, _ = torch.linalg.qr(torch.rand(3, 3, generator=torch.Generator().manual_seed(seed)))
        return q * torch.sign(torch.linalg.det(q))
    R1, RB = rot(1), torch.stack([rot(2), rot(3)])
    T1 = torch.eye(4); T1[:3, :3] = R1; T1[:3, 3] = torch.tensor([0.1, -0.2, 0.3])
    TB = torch.eye(4).repeat(2, 1, 1); TB[:, :3, :3] = RB; TB[:, :3, 3] = torch.tensor([[0.1, 0.2, 0.3], [-0.3, 0.2, 0.1]])
    K = torch.tensor([[525.0, 0, 319.5, 0], [0, 525.0, 239.5, 0], [0, 0, 1, 0], [0, 0, 0, 1]])
    off3, offB = torch.tensor([0.5, -1.0, 2.0]), torch.tensor([[[0.5, -1.0, 2.0]], [[1.5, 1.0, -2.0]]])
    for k, v in (("R1", R1), ("RB", RB), ("T1", T1), ("TB", TB), ("K", K), ("off3", off3), ("offB", offB)):
        out["arg_" + k] = v.numpy()
    rec("offset_scalar", fresh().offset_(0.25))
    rec("offset_vec", fresh().offset_(off3))
    rec("offset_batch", fresh().offset_(offB))
    rec("scale_scalar", fresh().scale_(1.5))
    rec("scale_vec", fresh().scale_(off3))
    rec("rotate_1_pre", fresh().rotate_(R1))
    rec("rotate_1_post", fresh().rotate_(R1, pre_multiplication=False))
    rec("rotate_B_pre", fresh().rotate_(RB))
    rec("transform_1_pre", fresh().transform_(T1))
    rec("transform_B_pre", fresh().transform_(TB))
    rec("transform_B_post", fresh().transform_(TB, pre_multiplication=False))
    rec("pinhole", fresh().pinhole_projection_(K))
    rec("op_add", fresh() + 0.25)
    rec("op_sub", fresh() - off3)
    rec("op_mul", fresh() * 1.5)
    rec("op_div", fresh() / 2.0)
    rec("op_matmul_R", fresh() @ R1)
    rec("op_matmul_T", fresh() @ TB)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, len(out), "arrays")


if __name__ == "__main__":
    main()
