"""Round 5, GPU call 2: which sequence makes a late-frame ICP launch slow, and in which phase?
(profiles/r05_frames_b8.txt: at frames 55..104 every launch of SOME solves takes ~47 us instead of ~12, at 8 sequences per
launch; sequence 0 alone shows nothing of the kind.)

    python tools/late_frames_probe.py [frames, default 70]          # part 1: ms per step of every seed alone, by frame
    GRADSLAM_HIP_LIB=.../libgradslam_hip_tl.so GRADSLAM_HIP_ICP_TIMELINE=/tmp/tl.txt python tools/late_frames_probe.py 62 tl
        # part 2 (library built with -DGS_ICP_TIMELINE): the 8 sequences as one batch up to the given frame; per
        # sequence of the last solve's last iteration: block life, phases, left-over / brute-force queries
"""
import multiprocessing as mp
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
L = int(sys.argv[1]) if len(sys.argv) > 1 else 70
MODE = sys.argv[2] if len(sys.argv) > 2 else "seeds"
H, W = 480, 640


def _make(seed):
    import importlib.util
    p = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gradslam_amd", "datasets", "synthetic.py")
    spec = importlib.util.spec_from_file_location("_gs_synthetic", p)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.make_sequence(L, H, W, seed=seed)


if __name__ == "__main__":
    with mp.get_context("fork").Pool(8) as pool:
        seqs = pool.map(_make, range(8))
    import torch
    import gradslam_amd as gs
    from gradslam_amd import ops

    def frames_of(ss):
        st = lambda k: torch.from_numpy(np.stack([s[k] for s in ss])).cuda()  # noqa: E731
        poses = st("poses")
        poses[:, 1:] = poses[:, :1]
        return gs.RGBDImages(st("colors"), st("depths"), st("intrinsics"), poses)

    if MODE == "seeds":
        print("# ms per step (HIP events) of every benchmark sequence ALONE (B = 1), mean over 10-frame bins; map rows at the end")
        for sd in range(8):
            fr = frames_of(seqs[sd:sd + 1])
            slam = gs.slam.PointFusion(odom="gradicp", device="cuda")
            pc, prev, evs = gs.Pointclouds(device="cuda"), None, []
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for f in range(L):
                live = fr[:, f]
                pc, _ = slam.step(pc, live, prev, inplace=True)
                prev = live
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                evs.append(e)
            torch.cuda.synchronize()
            ms = np.array([a.elapsed_time(b) for a, b in zip([e0] + evs[:-1], evs)])
            fa, em, op = ops.localize_list_stats(torch.device("cuda", 0), 0, H, W, 4, pc._buf["points"][0].shape[0])
            print("seed %d: " % sd + " ".join("%d-%d: %.3f" % (a, a + 9, ms[a:a + 10].mean()) for a in range(10, L - 9, 10)) +
                  "   rows %d; last solve: lists failed %d, points without a list (sum over launches) %d" % (
                      pc.points_list[0].shape[0], int(np.sum(fa)), int(np.sum(op))))
            del fr, pc, prev
    else:
        fr = frames_of(seqs)
        slam = gs.slam.PointFusion(odom="gradicp", device="cuda")
        pc, prev = gs.Pointclouds(device="cuda"), None
        for f in range(L):
            live = fr[:, f]
            pc, _ = slam.step(pc, live, prev, inplace=True)
            prev = live
        torch.cuda.synchronize()
        base = os.environ["GRADSLAM_HIP_ICP_TIMELINE"]
        for part, label in (("", "first half"), (".next", "look-ahead")):
            rows = np.loadtxt(base + part, dtype=np.uint64)
            print("== frame %d, iteration %s, %s: %s" % (L - 1, os.environ.get("GRADSLAM_HIP_ICP_TIMELINE_IT", "19"), label,
                                                        open(base + part).readline().strip()))
            rows = rows[rows[:, 5] > 0]
            t0 = rows[:, 1].min()
            blk = rows[:, 0].astype(np.int64)
            for b in range(8):
                r = rows[blk % 8 == b]
                us = lambda a: a.astype(np.float64) / 100.0   # noqa: E731
                life = us(r[:, 2] - r[:, 1])
                pro, sea, rest = us(r[:, 5] - r[:, 1]), us(r[:, 6] - r[:, 5]), us(r[:, 2] - r[:, 6])
                left = r[:, 3].astype(np.int64)
                nun = (r[:, 8] & np.uint64(0xffffffff)).astype(np.int64)
                if r.shape[1] > 13 and (r[:, 11] > 0).any():   # round-5 stamps: check done, failed lists re-searched, how many
                    chk, rs, hp = us(r[:, 11] - r[:, 5]), us(r[:, 12] - r[:, 11]), us(r[:, 6] - r[:, 12])
                    nf = r[:, 13].astype(np.int64)
                    print("         check %5.2f (max %5.2f)  re-search of failed lists %5.2f (max %5.2f; failed per block mean %5.1f max %3d)  "
                          "left-over pass %5.2f (max %5.2f)" % (chk.mean(), chk.max(), rs.mean(), rs.max(), nf.mean(), nf.max(), hp.mean(), hp.max()))
                print("  seq %d: blocks %3d  end of last block %6.2f us  life mean %5.2f max %5.2f | prologue %5.2f  "
                      "check/search+left-overs %5.2f (max %5.2f)  rows %5.2f | left-over queries %5d (max per block %3d)  "
                      "brute-force queries %4d (blocks with any %2d)" % (
                          b, len(r), us(r[:, 2] - t0).max(), life.mean(), life.max(), pro.mean(), sea.mean(), sea.max(),
                          rest.mean(), left.sum(), left.max(), nun.sum(), (nun > 0).sum()))
