"""Reading and time-stamp association of TUM RGB-D text files (the semantics of the reference's
datasets/tumutils.py, which adapts TUM's associate.py): colour, depth and pose streams are not
synchronised, so frames are paired by nearest time stamp within a search radius."""
import bisect
import warnings
from typing import Optional

__all__ = ["read_trajectory", "read_file_list", "associate"]


def _rows(filename):
    with open(filename) as f:
        text = f.read()
    rows = []
    for line in text.replace(",", " ").replace("\t", " ").split("\n"):
        if line and line[0] != "#":
            rows.append([v.strip() for v in line.split(" ") if v.strip() != ""])
    return rows, text.count("\n") + 1


def read_file_list(filename: str, start: Optional[int] = None, end: Optional[int] = None):
    r"""{stamp (str): [fields...]} for lines `stamp f1 f2 ...`; start / end slice the data lines
    (tumutils.py:132-159; comment lines do not count, `end` is checked against the raw line count)."""
    rows, n_lines = _rows(filename)
    start = 0 if start is None else start
    if end is None:
        end = n_lines
    if end > n_lines:
        warnings.warn('"end" was larger than number of frames in "{0}": {1} > {2}'.format(filename, end, n_lines))
    return {r[0]: r[1:] for r in rows[start:end] if len(r) > 1}


def read_trajectory(filename: str, matrix: bool = True):
    r"""{stamp: (tx, ty, tz, qx, qy, qz, qw)} (matrix=False, what the TUM loader uses) or 4x4 matrices;
    lines with an all-zero quaternion or NaNs are dropped (tumutils.py:82-129)."""
    import math
    rows, _ = _rows(filename)
    out = {}
    for r in rows:
        vals = [float(v) for v in r[1:]]
        if vals[3:7] == [0, 0, 0, 0] or any(math.isnan(v) for v in vals):
            continue
        out[r[0]] = vals[:7]
    if matrix:
        import numpy as np
        from .datautils import pointquaternion_to_homogeneous
        return {k: pointquaternion_to_homogeneous(np.asarray(v, np.float64)).astype(np.float64) for k, v in out.items()}
    return out


def associate(first_dict: dict, second_dict: dict, offset: float, max_difference: float):
    r"""Greedy nearest-stamp matching (tumutils.py:162-197): all pairs closer than `max_difference` are
    visited in order of (difference, stamp1, stamp2) and a pair is taken when both stamps are still
    free; the result is sorted by stamp.  Candidates come from a sorted window instead of the
    reference's all-pairs scan; the visiting order, hence the result, is the same."""
    second = sorted((float(b) + offset, b) for b in second_dict.keys())
    times = [t for t, _ in second]
    cand = []
    for a in first_dict.keys():
        ta = float(a)
        lo = bisect.bisect_left(times, ta - max_difference)
        hi = bisect.bisect_right(times, ta + max_difference)
        for t, b in second[lo:hi]:
            d = abs(ta - t)
            if d < max_difference:
                cand.append((d, a, b))
    cand.sort()
    free_a, free_b, matches = set(first_dict.keys()), set(second_dict.keys()), []
    for _, a, b in cand:
        if a in free_a and b in free_b:
            free_a.remove(a)
            free_b.remove(b)
            matches.append((a, b))
    matches.sort()
    return matches
