"""Stages the REAL reference next to the oracle so that it can be TIMED on the GPU box's host cores.

    python -m oracle.stage_reference            # build container only (needs /root/reference)

Packs the reference's Python package (/root/reference/gradslam, pure Python, 0.4 MB) into ONE archive under the
git-ignored oracle/_ref/ (gradslam_ref.zip; Python imports it in place -- zipimport -- so no reference source file ever
exists in this tree, tracked or not).  Like the built .so files, oracle/_ref/ is kept out of the history (.gitignore) but travels with
the gpurun snapshot, so bench.py's `cpu_baseline` leg can run gradslam's own CPU path (slam/icpslam.py:140-178 through
PointFusion.step) on the node's host in the same run as the GPU measurement (north_star) -- through
oracle/run_reference.py, in a subprocess, with the shim modules of oracle/shims standing in for the uninstalled
third-party imports (chamferdist's knn_points = the oracle's OpenMP brute force).  TEST INFRASTRUCTURE: the product
(gradslam_amd/) never imports it; nothing under oracle/_ref is committed.
"""
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF_SRC = "/root/reference/gradslam"
REF_ZIP = os.path.join(HERE, "_ref", "gradslam_ref.zip")


def staged():
    return os.path.isfile(REF_ZIP)


def stage(force=False):
    """Returns the archive path, or None when the reference checkout is not present and nothing is staged (the GPU box
    only ever uses what travelled with the snapshot)."""
    import zipfile
    if not os.path.isdir(REF_SRC):
        return REF_ZIP if staged() else None
    if staged() and not force:
        return REF_ZIP
    os.makedirs(os.path.dirname(REF_ZIP), exist_ok=True)
    old_dir = os.path.join(HERE, "_ref", "gradslam")   # (an unpacked copy of an earlier version of this script)
    if os.path.isdir(old_dir):
        shutil.rmtree(old_dir)
    tmp = REF_ZIP + ".tmp"
    with zipfile.ZipFile(tmp, "w", zipfile.ZIP_DEFLATED) as z:
        for root, dirs, files in os.walk(REF_SRC):
            dirs[:] = [d for d in dirs if d != "__pycache__"]
            for f in files:
                if f.endswith(".py"):
                    full = os.path.join(root, f)
                    z.write(full, os.path.join("gradslam", os.path.relpath(full, REF_SRC)))
    os.replace(tmp, REF_ZIP)
    with open(os.path.join(HERE, "_ref", "README"), "w") as f:
        f.write("gradslam_ref.zip: the reference's Python package packed by `python -m oracle.stage_reference`. Git-ignored; never commit.\n")
    return REF_ZIP


if __name__ == "__main__":
    d = stage(force="--force" in sys.argv)
    print(d if d else "reference checkout not present and nothing staged")
