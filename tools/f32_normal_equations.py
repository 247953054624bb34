"""EXPERIMENT: how much of the distance between our poses and the reference's is due to accumulating
the ICP normal equations in float64 (the product and the oracle) instead of float32 (the reference: sgemm + LAPACK)?

Runs the CPU oracle's frame loop on the first frames of the benchmarked 640x480 sequence twice -- default (float64 sums,
float64 solve) and GS_ORACLE_NEQ_F32=1 (float32 sums in 16 lanes, float32 LU solve) -- and compares both with the
REAL reference's run (tests/golden/pf640.npz): pose differences, ATE, surfel counts per frame.

    python tools/f32_normal_equations.py [frames]      (CPU only, ~10 s per frame and variant)"""
import json, os, subprocess, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
L = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 5
if "--child" in sys.argv:
    from gradslam_amd.datasets.synthetic import make_sequence
    from oracle import slam as oslam
    s = make_sequence(L, 480, 640, seed=0)
    poses = s["poses"][:L].copy(); poses[1:] = poses[:1]
    counts = []
    m, op = oslam.run_sequence(s["colors"][:L], s["depths"][:L], s["intrinsics"][0], poses, odom="gradicp",
                               per_frame=lambda f, mm, p: counts.append(len(mm)))
    print(json.dumps({"poses": op.tolist(), "counts": counts}))
    sys.exit(0)
g = np.load(os.path.join(REPO, "tests", "golden", "pf640.npz"))
res = {}
for name, env in (("float64 sums (product / oracle)", {}), ("float32 sums + float32 solve", {"GS_ORACLE_NEQ_F32": "1"})):
    out = subprocess.run([sys.executable, os.path.abspath(__file__), str(L), "--child"], env=dict(os.environ, **env),
                         capture_output=True, text=True, check=True).stdout.strip().splitlines()[-1]
    r = json.loads(out)
    P = np.asarray(r["poses"], np.float32)
    ate = float(np.sqrt(((P[:, :3, 3].astype(np.float64) - g["poses"][:L, :3, 3]) ** 2).sum(-1).mean()))
    res[name] = (P, np.asarray(r["counts"]))
    print("%-34s max |dT| vs reference %.2e   ATE %.2e m   surfel count difference per frame %s" % (
        name, float(np.abs(P - g["poses"][:L]).max()), ate, (np.asarray(r["counts"]) - g["counts"][:L]).tolist()))
a, b = res["float64 sums (product / oracle)"], res["float32 sums + float32 solve"]
print("between the two variants: max |dT| %.2e, count difference %s" % (float(np.abs(a[0] - b[0]).max()), (a[1] - b[1]).tolist()))
