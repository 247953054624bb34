"""Imports the REAL reference (gradslam v0.1.0) from /root/reference with the shim modules in
oracle/shims standing in for its uninstalled third-party imports.  Build-container only:
/root/reference does not exist on the GPU box, so nothing under tests -m gpu, smoke() or
bench.py may call this.  Used by the oracle/make_golden*.py generators."""
import os
import sys
import warnings

REFERENCE_ROOT = "/root/reference"
_SHIMS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "shims")


def available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "gradslam"))


def import_reference():
    if not available():
        raise RuntimeError("reference checkout not present at " + REFERENCE_ROOT)
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (repo, REFERENCE_ROOT, _SHIMS):
        if p not in sys.path:
            sys.path.insert(0, p)
    warnings.filterwarnings("ignore", category=UserWarning)
    warnings.filterwarnings("ignore", category=FutureWarning)
    import gradslam  # noqa: E402
    return gradslam
