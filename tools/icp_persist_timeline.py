"""Phase breakdown of the persistent per-XCD ICP solve (gs_icp_persist.h) from the per-block stamps a library built with
-DGS_ICP_TIMELINE writes (GRADSLAM_HIP_ICP_PERSIST_TIMELINE=<path>): the two half-iterations of one iteration.

    python tools/icp_persist_timeline.py <path>
"""
import sys
import numpy as np
rows = [l for l in open(sys.argv[1]) if not l.startswith("#")]
print(open(sys.argv[1]).readline().strip())
R = []
for l in rows:
    a, b, c, d = l.split("|")
    R.append([int(x) for x in a.split()] + [int(x) for x in b.split()] + [int(x) for x in c.split()] + [int(x) for x in d.split()])
R = np.array(R, dtype=np.float64)
names = ["wait (barrier -> released)", "row loads + sums", "scalar stage + barrier", "list check + rows", "rare passes", "reduction + arrive"]
for seq in sorted(set(R[:, 0])):
    S = R[R[:, 0] == seq]
    print("sequence / XCD %d: %d blocks" % (seq, len(S)))
    for half, off in (("first half", 3), ("look-ahead", 15)):
        st = S[:, off:off + 12] / 100.0   # us
        at_bar, t = st[:, 6], st[:, 0:6]
        d = np.column_stack([t[:, 0] - at_bar, t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2], t[:, 4] - t[:, 3], t[:, 5] - t[:, 4]])
        print("  %s: block life (released -> arrived) mean %.2f max %.2f us; last arrival - first release %.2f us" %
              (half, (t[:, 5] - t[:, 0]).mean(), (t[:, 5] - t[:, 0]).max(), t[:, 5].max() - t[:, 0].min()))
        for k, n in enumerate(names):
            print("    %-28s mean %6.2f  max %6.2f" % (n, d[:, k].mean(), d[:, k].max()))
        rare = st[:, 7] > 0   # blocks that ran the rare passes: check -> research -> hard -> brute -> (rows, list reload) -> rare done
        if rare.any():
            r = st[rare]
            hard = r[:, 9] > 0   # (the out-of-line passes for points the 2x2x2 stage cannot prove: only some blocks)
            last = np.where(hard, r[:, 9], r[:, 7])
            print("    rare passes of %d blocks: research %.2f / %.2f us (mean / max); %d of them with cubes + block pass %.2f / %.2f; new lists + rows %.2f / %.2f" %
                  (rare.sum(), (r[:, 7] - r[:, 3]).mean(), (r[:, 7] - r[:, 3]).max(), hard.sum(),
                   (r[hard, 9] - r[hard, 7]).mean() if hard.any() else 0.0, (r[hard, 9] - r[hard, 7]).max() if hard.any() else 0.0,
                   (r[:, 4] - last).mean(), (r[:, 4] - last).max()))
    fh, la = S[:, 3:15] / 100.0, S[:, 15:27] / 100.0
    print("  half-iteration period (release of the look-ahead - release of the first half): %.2f us" % (la[:, 0].min() - fh[:, 0].min()))
    rel = S[0, 27:] / 100.0   # block start, releases of every half-iteration, end (block 0 of the sequence)
    rel = rel[rel > 0]
    print("  block %d: start -> first half-iteration %.2f us; durations of the half-iterations (us): %s" %
          (S[0, 1], rel[1] - rel[0], " ".join("%.1f" % x for x in np.diff(rel[1:]))))
    print("  whole kernel, start of the first block -> end of the last: %.1f us" % ((S[:, 2].max() - S[:, 27].min()) / 100.0))
