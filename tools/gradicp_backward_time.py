"""BASELINE configs[2]: gradICP at 640x480 (ds=4 lattice, ~18k x 18k points) with the backward pass through
all 20 iterations: time of forward (taped) + backward, and of the depth -> pose chain through the drivers'
differentiable path."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gradslam_amd as gs
from gradslam_amd import ops
from gradslam_amd.datasets.synthetic import make_sequence

s = make_sequence(3, 480, 640, seed=0)
dv = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
K, pose = dv(s["intrinsics"][0]), dv(s["poses"][0])
sets = []
for f in (0, 2):
    d = dv(s["depths"][f, ..., 0])
    v, n, _, _ = ops.frame_maps(d, K)
    gv, gn = ops.global_maps(v, n, d, pose)
    sets.append(ops.downsample_frame(gv, gn, None, d, 4)[:2])
(tgt, tn), (src, _) = sets
src = src.clone().requires_grad_(True)

def fwd_bwd():
    T, _ = ops.grad_icp(src, tgt, tn, None, 20, 1e-8, None, 2.0, 1.0, 1.0, 200.0)
    T.sum().backward()
    return T

def timeit(f, n=10):
    for _ in range(3):   # the first calls pay one-time costs (code objects, workspace growth, autograd thread)
        f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

t_fb = timeit(fwd_bwd)
with torch.no_grad():
    t_f = timeit(lambda: ops.icp(src.detach(), tgt, tn, numiters=20, return_idx=False))
print("gradICP %d x %d points, 20 iterations: forward only %.3f ms, forward (taped) + backward %.3f ms"
      % (src.shape[0], tgt.shape[0], t_f, t_fb))

# depth -> vertex -> global -> lattice -> gradICP -> pose -> loss, through PointFusion.step
depth = dv(s["depths"][None, :2]).requires_grad_(True)
poses = s["poses"][:2].copy(); poses[1] = poses[0]
def chain():
    frames = gs.RGBDImages(dv(s["colors"][None, :2]), depth, dv(s["intrinsics"][None]), dv(poses[None]))
    slam = gs.slam.PointFusion(odom="gradicp", device="cuda")
    pc, prev, out = gs.Pointclouds(device="cuda"), None, []
    for t in range(2):
        live = frames[:, t]
        pc, p = slam.step(pc, live, prev, inplace=True)
        prev = live; out.append(p)
    loss = out[1].sum()
    g, = torch.autograd.grad(loss, depth)
    return g
print("PointFusion 2 frames 640x480 with d(pose)/d(depth): %.3f ms per call, grad finite %s, nonzero %d"
      % (timeit(chain, 5), bool(torch.isfinite(chain()).all()), int((chain() != 0).sum())))
