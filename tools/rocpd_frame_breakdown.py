"""Per-frame kernel breakdown from a rocprofv3 rocpd database (kernel-trace): averages over the last
N frames, frame boundaries = gs_frame_maps_kernel launches."""
import sqlite3, collections, sys
db = sqlite3.connect(sys.argv[1])
nf = int(sys.argv[2]) if len(sys.argv) > 2 else 10
rows = db.execute("select name,start,end from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if r[0].startswith("gs_frame_maps_kernel")]
seg = rows[idx[-nf - 1]:idx[-1]]
T = (seg[-1][2] - seg[0][1]) / nf
busy = sum(r[2] - r[1] for r in seg) / nf
print("per frame: wall %.0f us, kernel time %.0f us, %.1f launches" % (T / 1e3, busy / 1e3, len(seg) / nf))
agg = collections.defaultdict(lambda: [0, 0])
for n, s, e in seg:
    k = n.split("(")[0][:80]; agg[k][0] += e - s; agg[k][1] += 1
for n, (t, c) in sorted(agg.items(), key=lambda x: -x[1][0]):
    print("%8.1f us/frame %5.1f launches/frame avg %7.2f us  %s" % (t / nf / 1e3, c / nf, t / c / 1e3, n))
