"""Batched variable-length pointcloud / surfel map with the interface of the reference's
`gradslam.Pointclouds` (structures/pointclouds.py:13-1467), re-designed around growth:

  * every sequence b owns CAPACITY-BACKED buffers points/normals/colors (cap, 3) and features
    (cap, F); the HIP fuse/append kernels write new surfels in place and only the count
    changes.  Capacity grows geometrically, so a 500-frame sequence reallocates O(log N) times
    instead of re-concatenating and re-padding every attribute every frame
    (reference: pointclouds.py:1203-1228, :948-995);
  * `*_list` are zero-copy views buf[:n]; `*_padded` is a zero-copy view when the batch has
    one sequence (the one-sequence-per-GPU layout) and is materialised lazily otherwise.

Only the container logic lives here (torch as memory plumbing); the arithmetic of the SLAM path
is in libgradslam_hip.so.
"""
import time
from typing import List, Optional, Union

import torch

__all__ = ["Pointclouds"]

_ATTRS = ("points", "normals", "colors", "features")


def _canon_device(device):
    return torch.Tensor().to(device).device


class _CountGroup(object):
    """Surfel counts of the sequences of a batch that live ON THE DEVICE (written by the fuse/append kernels).

    The host only keeps upper bounds (launch geometry, capacity).  Every EVERY-th update queues ONE asynchronous
    copy of all the counts into pinned memory; `poll()` tightens the bounds from the copies that have already landed
    and never waits, so the frame loop has no host<->device sync."""
    RING = 8
    wait_s = 0.0    # seconds the host has spent waiting in `advance` (class-wide; bench.py reports host time without it)
    EVERY = 2       # one asynchronous read-back per EVERY updates (a copy + an event on the stream)
    MAX_AHEAD = 2   # read-backs in flight before the host waits for the oldest: the host never runs more than
                    # ~EVERY * (MAX_AHEAD + 1) frames ahead of the device, which keeps the bounds (launch sizes,
                    # capacity) within a few frames of the true counts while the device always has work queued

    def __init__(self, dev, bounds):
        self.dev, self.bounds = dev, [int(x) for x in bounds]   # dev: (B,) int64 on the device
        self._updates = 0
        self._pin = torch.empty((self.RING, len(self.bounds)), dtype=torch.int64).pin_memory()
        self._events = [torch.cuda.Event() for _ in range(self.RING)]
        self._pending = []   # [slot, rows that may have been added to every sequence since that copy]
        self._slot = 0
        self._queue_copy()

    def _queue_copy(self):
        if len(self._pending) == self.RING:
            return
        s = self._slot
        self._slot = (s + 1) % self.RING
        self._pin[s].copy_(self.dev, non_blocking=True)
        self._events[s].record()
        self._pending.append([s, 0])

    def advance(self, dev, max_growth):
        g = int(max_growth)
        self.dev = dev
        self.bounds = [x + g for x in self.bounds]
        for p in self._pending:
            p[1] += g
        self._updates += 1
        if self._updates % self.EVERY == 0:
            self._queue_copy()
            if len(self._pending) > self.MAX_AHEAD:
                t0 = time.perf_counter()
                self._events[self._pending[0][0]].synchronize()   # frames old: the device is still busy behind it
                _CountGroup.wait_s += time.perf_counter() - t0
        self.poll()

    def poll(self):
        while self._pending and self._events[self._pending[0][0]].query():
            s, grown = self._pending.pop(0)
            self.bounds = [min(x, v + grown) for x, v in zip(self.bounds, self._pin[s].tolist())]

    def resolve(self):
        return [int(v) for v in self.dev.cpu().tolist()]

    def tighten(self):
        """One sync: the bounds become the exact counts (used before a buffer would be reallocated: the bounds run a
        few frames x H*W rows ahead of the counts, which for a 1296x968 frame is millions of rows)."""
        vals = self.resolve()
        self.poll()          # every queued copy has landed by now
        self.bounds = vals
        return vals


class _DeviceCount(object):
    """The count of ONE sequence inside a _CountGroup (the group does the read-backs for all its sequences)."""

    def __init__(self, group, index):
        self.group, self.index = group, index

    @property
    def bound(self):
        return self.group.bounds[self.index]

    @property
    def dev(self):
        return self.group.dev[self.index:self.index + 1]

    def poll(self):
        self.group.poll()

    def resolve(self):
        return int(self.dev.item())


class Pointclouds(object):
    r"""Batch of pointclouds (with varying numbers of points).

    Args:
        points (list of (N_b, 3) tensors, or (B, N, 3) tensor, or None)
        normals, colors: same container type and shapes as `points`, or None
        features: list of (N_b, F) tensors or (B, N, F) tensor, or None
        device: device of the internal tensors (default: that of `points`, or cpu when empty)
    """

    def __init__(
        self,
        points: Union[List[torch.Tensor], torch.Tensor, None] = None,
        normals: Union[List[torch.Tensor], torch.Tensor, None] = None,
        colors: Union[List[torch.Tensor], torch.Tensor, None] = None,
        features: Union[List[torch.Tensor], torch.Tensor, None] = None,
        device: Union[torch.device, str, None] = None,
    ):
        if not (points is None or isinstance(points, list) or torch.is_tensor(points)):
            raise TypeError("Expected points to be of type list or tensor or None; got %r" % type(points))
        for name, val in (("normals", normals), ("colors", colors), ("features", features)):
            if not (val is None or isinstance(val, type(points))):
                raise TypeError("Expected %s to be of same type as points (%r); got %r"
                                % (name, type(points), type(val)))
        if points is not None and len(points) == 0:
            raise ValueError("len(points) (= 0) should be > 0")

        self._buf = {k: None for k in _ATTRS}   # per attribute: list of (cap_b, C) tensors or None
        self._dcount = {}                       # b -> _DeviceCount while the count of b is device-side
        self._n_host: List[int] = []            # points per sequence (see the `_n` property)
        self._padded_cache = {}
        self.equisized = None

        if isinstance(points, list):
            shapes = [p.shape for p in points]
            if any(p.ndim != 2 for p in points):
                raise ValueError("ndim of all tensors in points list should be 2")
            if any(s[-1] != 3 for s in shapes):
                raise ValueError("last dim of all tensors in points should have shape 3 (X, Y, Z)")
            self.device = _canon_device(device) if device is not None else points[0].device
            if not (normals is None or [n.shape for n in normals] == shapes):
                raise ValueError("normals tensors should have same shape as points tensors, but didn't")
            if not (colors is None or [c.shape for c in colors] == shapes):
                raise ValueError("colors tensors should have same shape as points tensors, but didn't")
            if not (features is None or all(f.ndim == 2 for f in features)):
                raise ValueError("ndim of all tensors in features list should be 2")
            if not (features is None or [len(f) for f in features] == [s[0] for s in shapes]):
                raise ValueError("number of features per pointcloud has to be equal to number of points")
            if not (features is None or len(set(f.shape[-1] for f in features)) == 1):
                raise ValueError("number of features per pointcloud has to be the same")
            self._n = [int(s[0]) for s in shapes]
            for k, val in zip(_ATTRS, (points, normals, colors, features)):
                self._buf[k] = None if val is None else [v.to(self.device) for v in val]
            self.equisized = len(set(self._n)) == 1
        elif torch.is_tensor(points):
            self.device = _canon_device(device) if device is not None else points.device
            if points.ndim != 3:
                raise ValueError("points should have ndim=3, but had ndim={}".format(points.ndim))
            if points.shape[-1] != 3:
                raise ValueError("last dim of points should have shape 3 (X, Y, Z) but had shape %r"
                                 % (points.shape[-1]))
            if points.shape[0] == 0:
                raise ValueError("Batch size of 0 not supported yet. Got input points shape {}.".format(points.shape))
            if not (normals is None or normals.shape == points.shape):
                raise ValueError("normals tensor should have same shape as points tensor, but didn't: %r != %r"
                                 % (normals.shape, points.shape))
            if not (colors is None or colors.shape == points.shape):
                raise ValueError("colors tensor should have same shape as points tensor, but didn't: %r != %r"
                                 % (colors.shape, points.shape))
            if not (features is None or features.ndim == 3):
                raise ValueError("features should have ndim=3, but had ndim={}".format(features.ndim))
            if not (features is None or features.shape[:-1] == points.shape[:-1]):
                raise ValueError("first 2 dims of features tensor and points tensor should have same shape, "
                                 "but didn't: %r != %r" % (features.shape[:-1], points.shape[:-1]))
            B, N = points.shape[:2]
            self._n = [int(N)] * B
            for k, val in zip(_ATTRS, (points, normals, colors, features)):
                self._buf[k] = None if val is None else [val[b].to(self.device) for b in range(B)]
            self.equisized = True
        else:
            self.device = _canon_device(device) if device is not None else torch.device("cpu")

    # ------------------------------------------------------------------ basic protocol
    @property
    def _n(self):
        """Points per sequence.  Reading it resolves device-side counts (one read-back each)."""
        if self._dcount:
            exact = {}   # one read-back per group
            for b, dc in self._dcount.items():
                if id(dc.group) not in exact:
                    exact[id(dc.group)] = dc.group.resolve()
                self._n_host[b] = exact[id(dc.group)][dc.index]
            self._dcount = {}
            self._invalidate()
        return self._n_host

    @_n.setter
    def _n(self, value):
        self._n_host = value
        self._dcount = {}

    def __len__(self):
        return len(self._n_host)

    @property
    def _B(self):
        return len(self._n_host)

    @property
    def _N(self):
        return max(self._n) if self._n else 0

    def __getitem__(self, index):
        if not self.has_points:
            raise IndexError("cannot index an empty Pointclouds")
        if isinstance(index, int):
            ids = [index]
        elif isinstance(index, slice):
            ids = list(range(len(self)))[index]
        elif isinstance(index, list):
            ids = index
        elif torch.is_tensor(index):
            if index.dim() != 1 or index.dtype.is_floating_point:
                raise IndexError(index)
            ids = index.nonzero().flatten().tolist() if index.dtype == torch.bool else index.tolist()
        else:
            raise IndexError(index)
        if len(ids) == 0:
            raise IndexError("Incorrect indexing at dimension 0, make sure range is within 0 and %d" % len(self))
        lists = {k: (None if self._buf[k] is None else [self._buf[k][i][: self._n[i]] for i in ids]) for k in _ATTRS}
        return Pointclouds(lists["points"], lists["normals"], lists["colors"], lists["features"])

    # ------------------------------------------------------------------ has_*
    @property
    def has_points(self):
        return self._buf["points"] is not None and any(n > 0 for n in self._n)

    @property
    def has_normals(self):
        return self._buf["normals"] is not None

    @property
    def has_colors(self):
        return self._buf["colors"] is not None

    @property
    def has_features(self):
        return self._buf["features"] is not None

    @property
    def num_features(self):
        return None if not self.has_features else self._buf["features"][0].shape[-1]

    @property
    def num_points_per_pointcloud(self):
        return torch.tensor(self._n if self._n else [0], device=self.device)

    # ------------------------------------------------------------------ list / padded views
    def _list(self, k):
        if self._buf[k] is None:
            return None
        return [t[:n] for t, n in zip(self._buf[k], self._n)]

    def _padded(self, k):
        if self._buf[k] is None:
            return None
        if len(self._n) == 1:  # zero-copy: the one-sequence-per-GPU layout
            return self._buf[k][0][: self._n[0]].unsqueeze(0)
        key = (k, tuple(self._n))
        hit = self._padded_cache.get(k)
        if hit is not None and hit[0] == key:
            return hit[1]
        N, C = self._N, self._buf[k][0].shape[-1]
        out = torch.zeros((len(self._n), N, C), dtype=self._buf[k][0].dtype, device=self.device)
        for b, (t, n) in enumerate(zip(self._buf[k], self._n)):
            out[b, :n] = t[:n]
        self._padded_cache[k] = (key, out)
        return out

    # list representation: zero-copy views; the setters follow the reference (structures/pointclouds.py:824-878,
    # :1431-1467: same container type, length and per-sequence shapes, values are cloned)
    def _set_list(self, k, value, first_dim_only=False):
        if not isinstance(value, list):
            raise TypeError("value must be list of torch.Tensors. Got {}".format(type(value)))
        if not self.has_points:
            raise ValueError("cannot set list representation for an empty pointclouds object")
        if len(self) != len(value):
            raise ValueError("value must have same length as pointclouds.points_list. Got {} != {}.".format(
                len(value), len(self)))
        if any(v.ndim != 2 for v in value):
            raise ValueError("ndim of all tensors in value list should be 2")
        pts = self.points_list
        if first_dim_only and any(pts[b].shape[:1] != value[b].shape[:1] for b in range(len(self))):
            raise ValueError("shape of first 2 dims of tensors in value and pointclouds.points_list must match")
        if (not first_dim_only) and any(pts[b].shape != value[b].shape for b in range(len(self))):
            raise ValueError("shape of tensors in value and pointclouds.points_list must match")
        self._store_rows(k, [v.to(self.device) for v in value])

    def _store_rows(self, k, rows):
        """New values of attribute k (one (n_b, C) tensor per sequence) on buffers that keep the CAPACITY of the
        sequence's points buffer: the in-place kernels size every attribute by that capacity (an exact-size clone here
        would be overrun by the next fuse / append)."""
        out = []
        for b, v in enumerate(rows):
            n = int(v.shape[0])
            ref = self._buf["points"][b] if self._buf["points"] is not None else None
            cap = max(n, int(ref.shape[0]) if ref is not None else n)
            buf = torch.empty((cap, v.shape[-1]), dtype=v.dtype, device=self.device)
            buf[:n] = v
            out.append(buf)
        self._buf[k] = out
        self._padded_cache.pop(k, None)

    points_list = property(lambda self: self._list("points"), lambda self, v: self._set_list("points", v))
    normals_list = property(lambda self: self._list("normals"), lambda self, v: self._set_list("normals", v))
    colors_list = property(lambda self: self._list("colors"), lambda self, v: self._set_list("colors", v))
    features_list = property(lambda self: self._list("features"), lambda self, v: self._set_list("features", v, True))

    @property
    def points_padded(self):
        return self._padded("points")

    @property
    def normals_padded(self):
        return self._padded("normals")

    @property
    def colors_padded(self):
        return self._padded("colors")

    @property
    def features_padded(self):
        return self._padded("features")

    @property
    def nonpad_mask(self):
        if not self._n:
            return None
        ar = torch.arange(self._N, device=self.device).unsqueeze(0)
        return ar < torch.tensor(self._n, device=self.device).unsqueeze(1)

    def _set_padded(self, k, value, channels=None):
        if value is None:
            self._buf[k] = None
            return
        if not torch.is_tensor(value):
            raise TypeError("value must be torch.Tensor. Got {}".format(type(value)))
        if not self._n:
            raise ValueError("cannot set padded representation for an empty pointclouds object")
        C = channels if channels is not None else value.shape[-1]
        if value.ndim != 3 or tuple(value.shape) != (len(self._n), self._N, C):
            raise ValueError("value must have shape {}, but had shape {}".format((len(self._n), self._N, C),
                                                                                 tuple(value.shape)))
        if value.device != self.device:
            raise ValueError("value must have the same device as pointclouds object: {} != {}".format(
                value.device, self.device))
        self._store_rows(k, [value[b, :n] for b, n in enumerate(self._n)])

    @points_padded.setter
    def points_padded(self, value):
        self._set_padded("points", value, 3)

    @normals_padded.setter
    def normals_padded(self, value):
        self._set_padded("normals", value, 3)

    @colors_padded.setter
    def colors_padded(self, value):
        self._set_padded("colors", value, 3)

    @features_padded.setter
    def features_padded(self, value):
        self._set_padded("features", value)

    # ------------------------------------------------------------------ surfel-store interface
    # (used by gradslam_amd.slam.fusionutils; not part of the reference API)
    def _invalidate(self):
        self._padded_cache.clear()
        self.equisized = (len(set(self._n)) == 1) if self._n else None

    RESERVE_FRAMES = 16   # capacity a SLAM driver asks for up front, in frames (see _reserve)

    def _init_empty_batch(self, B, num_features, with_normals=True, with_colors=True):
        """Turns an empty map into B empty sequences with the given attribute set."""
        assert not self._n
        self._n = [0] * B
        mk = lambda c: [torch.empty((0, c), dtype=torch.float32, device=self.device) for _ in range(B)]  # noqa: E731
        self._buf["points"] = mk(3)
        self._buf["normals"] = mk(3) if with_normals else None
        self._buf["colors"] = mk(3) if with_colors else None
        self._buf["features"] = mk(num_features) if num_features else None

    def _reserve(self, b, extra, frames_ahead=1):
        """Guarantees room for `extra` more rows in sequence b (geometric growth) and returns the
        capacity-backed buffers (points, normals, colors, features).  `frames_ahead`: the SLAM drivers, which
        append up to `extra` rows per frame for many frames, ask for several frames of room at once."""
        n_b = self._count_of(b)[0]
        need = n_b + int(extra)
        # (the smallest buffer counts: every attribute is written up to the same row)
        cap = min(self._buf[k][b].shape[0] for k in _ATTRS if self._buf[k] is not None)
        if need > cap and b in self._dcount:
            # the host only knows an upper bound of the count: one sync for the exact value is cheaper than copying the
            # map into buffers twice the size (and re-sizing every scratch that follows the capacity) frames early --
            # unless the bound would be back at the capacity within the few frames the host runs ahead (then every one
            # of those frames would sync here): grow now
            grp = self._dcount[b].group
            grp.tighten()
            n_b = self._count_of(b)[0]
            ahead = grp.EVERY * (grp.MAX_AHEAD + 1) + 2
            need = n_b + int(extra) * (ahead if frames_ahead > 1 else 1)
        if need > cap:
            # geometric growth, starting at RESERVE_FRAMES x the request: a surfel map of a few hundred MB is
            # nothing in 288 GB of HBM, and every reallocation (and every size class the bound-sized per-frame
            # temporaries move through) is a hipMalloc in the middle of a sequence
            new_cap = max(need, int(cap * 2), int(frames_ahead) * int(extra), 1024)
            for k in _ATTRS:
                if self._buf[k] is None:
                    continue
                old = self._buf[k][b]
                if old.shape[0] >= new_cap:
                    continue
                new = torch.empty((new_cap, old.shape[-1]), dtype=old.dtype, device=self.device)
                new[:n_b] = old[:n_b]
                self._buf[k][b] = new
            self._padded_cache.clear()
        return tuple(None if self._buf[k] is None else self._buf[k][b] for k in _ATTRS)

    def _set_count(self, b, n):
        self._n[b] = int(n)
        self._invalidate()

    def _count_of(self, b):
        """(upper bound on the host, device int64[1] tensor or None when the host count is exact)."""
        dc = self._dcount.get(b)
        if dc is None:
            return self._n_host[b], None
        dc.poll()
        return dc.bound, dc.dev

    def _tighten_counts(self):
        """Reads the device-side counts back (one sync) and makes the host-side BOUNDS exact, keeping the device counts in
        charge: the next step takes the same path as without this call (used by profiling passes, whose byte counts come
        from the bounds)."""
        if not self._dcount:
            return list(self._n_host)
        for grp in {id(dc.group): dc.group for dc in self._dcount.values()}.values():
            grp.tighten()
        return [self._dcount[b].bound if b in self._dcount else self._n_host[b] for b in range(len(self._n_host))]

    def _set_count_dev(self, b, dev_count, max_growth):
        """The kernels wrote the new count of sequence b to `dev_count`; at most `max_growth` rows
        were added.  Nothing is read back (see _DeviceCount)."""
        dc = self._dcount.get(b)
        if dc is None or len(dc.group.bounds) != 1:
            bound = (self._n_host[b] if dc is None else dc.bound) + int(max_growth)
            self._dcount[b] = _DeviceCount(_CountGroup(dev_count.reshape(1), [bound]), 0)
        else:
            dc.group.advance(dev_count.reshape(1), max_growth)
        self._padded_cache.clear()
        self.equisized = True if len(self._n_host) == 1 else None

    def _set_counts_dev(self, dev_counts, max_growth):
        """Same for every sequence of the batch at once: dev_counts is the (B,) int64 tensor the batched kernels
        wrote; ONE asynchronous read-back serves all sequences."""
        B = len(self._n_host)
        dcs = [self._dcount.get(b) for b in range(B)]
        grp = dcs[0].group if dcs[0] is not None else None
        if grp is not None and len(grp.bounds) == B and all(d is not None and d.group is grp and d.index == b
                                                            for b, d in enumerate(dcs)):
            grp.advance(dev_counts, max_growth)
        else:
            bounds = [(self._n_host[b] if d is None else d.bound) + int(max_growth) for b, d in enumerate(dcs)]
            grp = _CountGroup(dev_counts, bounds)
            for b in range(B):
                self._dcount[b] = _DeviceCount(grp, b)
        self._padded_cache.clear()
        self.equisized = True if B == 1 else None

    # ------------------------------------------------------------------ copies / moves
    def clone(self):
        other = Pointclouds(device=self.device)
        other._n = list(self._n)
        for k in _ATTRS:
            other._buf[k] = None if self._buf[k] is None else [t[:n].clone() for t, n in zip(self._buf[k], self._n)]
        other.equisized = self.equisized
        return other

    def detach(self):
        other = Pointclouds(device=self.device)
        other._n = list(self._n)
        for k in _ATTRS:
            other._buf[k] = None if self._buf[k] is None else [t.detach() for t in self._buf[k]]
        other.equisized = self.equisized
        return other

    def to(self, device, copy: bool = False):
        device = _canon_device(device)
        if not copy and self.device == device:
            return self
        other = self.clone()
        if self.device != device:
            other.device = device
            for k in _ATTRS:
                if other._buf[k] is not None:
                    other._buf[k] = [t.to(device) for t in other._buf[k]]
        return other

    def cpu(self):
        return self.to(torch.device("cpu"))

    def cuda(self):
        return self.to(torch.device("cuda"))

    # ------------------------------------------------------------------ append
    def append_points(self, pointclouds: "Pointclouds"):
        r"""Appends the points of `pointclouds` sequence-wise, in place
        (reference: structures/pointclouds.py:1117-1237)."""
        if not isinstance(pointclouds, type(self)):
            raise TypeError("Append object must be of type gradslam.Pointclouds, but was of type {}.".format(
                type(pointclouds)))
        if not (pointclouds.device == self.device):
            raise ValueError("Device of pointclouds to append and to be appended must match: ({0} != {1})".format(
                pointclouds.device, self.device))
        if not pointclouds.has_points:
            return self
        if self.has_points:
            if len(pointclouds) != len(self):
                raise ValueError("Batch size of pointclouds to append and to be appended must match: ({0} != {1})"
                                 .format(len(pointclouds), len(self)))
            for name in ("normals", "colors", "features"):
                if (self._buf[name] is not None) != (pointclouds._buf[name] is not None):
                    raise ValueError("pointclouds to append and to be appended must either both have or not have "
                                     "{0}: ({1} != {2})".format(name, pointclouds._buf[name] is not None,
                                                                self._buf[name] is not None))
            if self.has_features and self.num_features != pointclouds.num_features:
                raise ValueError("pointclouds to append and to be appended must have the same number of features: "
                                 "({0} != {1})".format(pointclouds.num_features, self.num_features))
            for b in range(len(self)):
                m = pointclouds._n[b]
                if m == 0:
                    continue
                self._reserve(b, m)
                n0 = self._n[b]
                for k in _ATTRS:
                    if self._buf[k] is not None:
                        self._buf[k][b][n0:n0 + m] = pointclouds._buf[k][b][:m]
                self._n[b] = n0 + m
        else:
            self._n = list(pointclouds._n)
            for k in _ATTRS:
                src = pointclouds._buf[k]
                self._buf[k] = None if src is None else [t[:n].clone() for t, n in zip(src, pointclouds._n)]
        self._invalidate()
        return self

    # ------------------------------------------------------------------ rigid-body helpers
    # Container algebra of the reference API (pointclouds.py:399-614).  The SLAM hot path does
    # NOT go through these (projection/association is fused in the HIP kernels).
    def _apply(self, k, fn):
        if self._buf[k] is not None:   # (results land on capacity-backed buffers again: _store_rows)
            self._store_rows(k, [fn(b, t[:n]) for b, (t, n) in enumerate(zip(self._buf[k], self._n))])

    @staticmethod
    def _rigid_rows(t, M, tvec=None):
        """t @ M (+ tvec) for the rows of one cloud.  On the GPU: gs_transform_points_f32 (the FMA chain of the
        reference's batched matmul, then the translation); on the CPU (host-side logic, tests): the same as tensor ops."""
        if t.is_cuda and t.dtype == torch.float32 and not t.requires_grad:
            from .. import ops
            T = torch.zeros((4, 4), dtype=torch.float32, device=t.device)
            T[:3, :3] = M.transpose(0, 1)
            if tvec is not None:
                T[:3, 3] = tvec
            T[3, 3] = 1.0
            return ops.transform_points(t, T)
        out = t @ M
        return out if tvec is None else out + tvec

    def offset_(self, offset):
        if not (torch.is_tensor(offset) or isinstance(offset, (float, int))):
            raise TypeError("Operand should be tensor, float or int but was %r instead" % type(offset))
        if not self.has_points:
            return self
        if torch.is_tensor(offset) and offset.ndim == 3:
            self._apply("points", lambda b, t: t + offset[b if offset.shape[0] > 1 else 0, : t.shape[0] if offset.shape[1] > 1 else 1])
        else:
            self._apply("points", lambda b, t: t + offset)
        return self

    def scale_(self, scale):
        if not (torch.is_tensor(scale) or isinstance(scale, (float, int))):
            raise TypeError("Operand should be tensor, float or int but was %r instead" % type(scale))
        if self.has_points:
            self._apply("points", lambda b, t: t * scale)
        return self

    def rotate_(self, rmat: torch.Tensor, *, pre_multiplication=True):
        if not torch.is_tensor(rmat):
            raise TypeError("Rotation matrix should be tensor, but was %r instead" % type(rmat))
        if not ((rmat.ndim == 2 or rmat.ndim == 3) and rmat.shape[-2:] == (3, 3)):
            raise ValueError("Rotation matrix should be of shape (3, 3) or (B, 3, 3), but was {} instead.".format(
                rmat.shape))
        if rmat.ndim == 3 and rmat.shape[0] != len(self):
            raise ValueError("Rotation matrix batch size ({}) != Pointclouds batch size ({})".format(
                rmat.shape[0], len(self)))
        if not self.has_points:
            return self
        if pre_multiplication:
            rmat = rmat.transpose(-1, -2)
        pick = (lambda b: rmat[b]) if rmat.ndim == 3 else (lambda b: rmat)
        self._apply("points", lambda b, t: self._rigid_rows(t, pick(b)))
        self._apply("normals", lambda b, t: self._rigid_rows(t, pick(b)))
        return self

    def transform_(self, transform: torch.Tensor, *, pre_multiplication=True):
        if not torch.is_tensor(transform):
            raise TypeError("transform should be tensor, but was %r instead" % type(transform))
        if not ((transform.ndim == 2 or transform.ndim == 3) and transform.shape[-2:] == (4, 4)):
            raise ValueError("transform should be of shape (4, 4) or (B, 4, 4), but was {} instead.".format(
                transform.shape))
        if transform.ndim == 3 and transform.shape[0] != len(self):
            raise ValueError("transform batch size ({}) != Pointclouds batch size ({})".format(
                transform.shape[0], len(self)))
        if not self.has_points:
            return self
        rmat, tvec = transform[..., :3, :3], transform[..., :3, 3]
        self.rotate_(rmat, pre_multiplication=pre_multiplication)
        pick = (lambda b: tvec[b]) if tvec.ndim == 2 else (lambda b: tvec)
        self._apply("points", lambda b, t: t + pick(b))
        return self

    def pinhole_projection_(self, intrinsics: torch.Tensor):
        if not torch.is_tensor(intrinsics):
            raise TypeError("intrinsics should be tensor, but was {} instead".format(type(intrinsics)))
        if not ((intrinsics.ndim == 2 or intrinsics.ndim == 3) and intrinsics.shape[-2:] == (4, 4)):
            raise ValueError("intrinsics should be of shape (4, 4) or (B, 4, 4), but was {} instead.".format(
                intrinsics.shape))
        if not self.has_points:
            return self
        pick = (lambda b: intrinsics[b]) if intrinsics.ndim == 3 else (lambda b: intrinsics)

        def proj(b, t):
            K = pick(b)
            if t.is_cuda and t.dtype == torch.float32 and not t.requires_grad:   # gs_project_points_f32
                from .. import ops
                uv = ops.project_points(t, K.reshape(1, 4, 4), max(int(t.shape[0]), 1))
                return torch.cat([uv, torch.ones_like(uv[:, :1])], -1)
            h = torch.cat([t, torch.ones_like(t[:, :1])], -1) @ K.transpose(0, 1)
            z = torch.where(h[:, 2:3] != 0, h[:, 2:3], torch.ones_like(h[:, 2:3]))
            return torch.cat([h[:, :2] / z, torch.ones_like(z)], -1)

        self._apply("points", proj)
        return self

    # arithmetic operators of the reference (structures/pointclouds.py:300-384): out-of-place offset / scale /
    # rotation / rigid transform
    def __add__(self, other):
        try:
            return self.clone().offset_(other)
        except TypeError:
            raise NotImplementedError("Pointclouds + {} currently not implemented.".format(type(other)))

    def __sub__(self, other):
        try:
            return self.clone().offset_(other * -1)
        except TypeError:
            raise NotImplementedError("Pointclouds - {} currently not implemented.".format(type(other)))

    def __mul__(self, other):
        try:
            return self.clone().scale_(other)
        except TypeError:
            raise NotImplementedError("Pointclouds * {} currently not implemented.".format(type(other)))

    def __truediv__(self, other):
        try:
            return self.__mul__(1.0 / other)
        except TypeError:
            raise NotImplementedError("Pointclouds / {} currently not implemented.".format(type(other)))

    def __matmul__(self, other):
        if not torch.is_tensor(other):
            raise NotImplementedError("Pointclouds @ {} currently not implemented.".format(type(other)))
        if not ((other.ndim == 2 or other.ndim == 3) and (other.shape[-2:] == (3, 3) or other.shape[-2:] == (4, 4))):
            msg = "Unsupported shape for Pointclouds @ operand: {}\n".format(other.shape)
            msg += "Use tensor of shape (3, 3) or (B, 3, 3) for rotations, or (4, 4) or (B, 4, 4) for transformations"
            raise ValueError(msg)
        if other.shape[-2:] == (3, 3):
            return self.clone().rotate_(other, pre_multiplication=False)
        return self.clone().transform_(other, pre_multiplication=False)

    def offset(self, offset):
        return self.clone().offset_(offset)

    def scale(self, scale):
        return self.clone().scale_(scale)

    def rotate(self, rmat, *, pre_multiplication=True):
        return self.clone().rotate_(rmat, pre_multiplication=pre_multiplication)

    def transform(self, transform, *, pre_multiplication=True):
        return self.clone().transform_(transform, pre_multiplication=pre_multiplication)

    def pinhole_projection(self, intrinsics):
        return self.clone().pinhole_projection_(intrinsics)
