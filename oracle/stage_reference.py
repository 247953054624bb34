"""Stages the REAL reference next to the oracle so that it can be TIMED on the GPU box's host cores.

    python -m oracle.stage_reference            # build container only (needs /root/reference)

Copies the reference's Python package (/root/reference/gradslam, pure Python, 0.4 MB) into the git-ignored
oracle/_ref/gradslam.  Like the built .so files, oracle/_ref/ is kept out of the history (.gitignore) but travels with
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
REF_DST = os.path.join(HERE, "_ref", "gradslam")


def staged():
    return os.path.isfile(os.path.join(REF_DST, "slam", "pointfusion.py"))


def stage(force=False):
    """Returns the staged package directory, or None when the reference checkout is not present (the GPU box)."""
    if not os.path.isdir(REF_SRC):
        return REF_DST if staged() else None
    if staged() and not force:
        return REF_DST
    if os.path.isdir(REF_DST):
        shutil.rmtree(REF_DST)
    os.makedirs(os.path.dirname(REF_DST), exist_ok=True)
    shutil.copytree(REF_SRC, REF_DST, ignore=shutil.ignore_patterns("__pycache__", "*.pyc"))
    with open(os.path.join(HERE, "_ref", "README"), "w") as f:
        f.write("Staged copy of /root/reference/gradslam (python -m oracle.stage_reference). Git-ignored; never commit.\n")
    return REF_DST


if __name__ == "__main__":
    d = stage(force="--force" in sys.argv)
    print(d if d else "reference checkout not present and nothing staged")
