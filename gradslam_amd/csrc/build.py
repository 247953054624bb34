"""Builds libgradslam_hip.so for gfx950 with hipcc, in-tree (the .so travels to the GPU box).

    python -m gradslam_amd.csrc.build [--force]

-ffp-contract=off is part of the arithmetic contract: the kernels spell
out every FMA explicitly; the compiler must not add or remove any."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ["gs_frame.hip", "gs_assoc.hip", "gs_fuse.hip", "gs_knn.hip", "gs_icp.hip", "gs_icp_loop.hip", "gs_icp_bwd.hip", "gs_ingest.hip"]
HEADERS = ["gs_common.h", "gs_compact.h", "gs_assoc_dev.h", "gs_knn.h", "gs_knn_bbox.h", "gs_icp_math.h", "gs_icp_persist.h", os.path.join("..", "..", "include", "gradslam_hip.h")]
LIB = os.path.join(HERE, "libgradslam_hip.so")
# -fno-slp-vectorize: hipcc otherwise packs adjacent f32 ops into v_pk_* instructions, which issue at a
# quarter of the scalar VALU rate on gfx950 (measured: the brute-force 1-NN kernel ran 1.6x slower packed)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize",
         "-fvisibility=hidden", "-Wall", "-Wno-unused-function"]


# extra compiler flags for debugging builds, e.g. GRADSLAM_HIP_BUILD_FLAGS="-DGS_ICP_TIMELINE" (tools/icp_timeline.py)
FLAGS += os.environ.get("GRADSLAM_HIP_BUILD_FLAGS", "").split()


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.sep not in c or os.path.exists(c)):
            return c
    raise RuntimeError("hipcc not found")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    hipcc = _hipcc()
    # objects compiled with other flags (a debugging build) are stale whatever their age
    stamp = os.path.join(HERE, ".build_flags")
    flags_now = " ".join(FLAGS)
    if not os.path.exists(stamp) or open(stamp).read() != flags_now:
        force = True
    hdrs = [os.path.join(HERE, h) for h in HEADERS]
    jobs = []
    for s in SOURCES:
        src = os.path.join(HERE, s)
        obj = os.path.join(HERE, s.replace(".hip", ".o"))
        if force or _stale(obj, [src] + hdrs):
            jobs.append([hipcc] + FLAGS + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd))
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n%s\n%s" % (" ".join(cmd), r.stderr))
        return r.stderr

    with ThreadPoolExecutor(max_workers=4) as ex:
        for err in ex.map(run, jobs):
            if err.strip() and verbose:
                print(err)
    objs = [os.path.join(HERE, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or _stale(LIB, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    with open(stamp, "w") as f:
        f.write(flags_now)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
