"""One process per GPU: independent RGB-D sequences shard across ranks.

Sequences of a batch never exchange data (the reference loops `for b in range(B)`,
odometry/icp.py:84), so the data path needs NO collective: rank r runs the sequences
`shard_sequences(B, world, r)` on its own MI355X.  The only exchange is the final gather of the
recovered poses (64 B per frame) and, optionally, the maps (~40 B per surfel), done once with
torch.distributed (backend "nccl" == RCCL over xGMI on ROCm; "gloo" in the CPU tests):
one small object gather (per-sequence counts, attribute widths) followed by ONE padded all_gather of the packed
surfels (poses: one padded all_gather).
"""
import os
from typing import List, Optional

import torch
import torch.distributed as dist

from .structures.pointclouds import Pointclouds

__all__ = ["init_from_env", "shard_sequences", "gather_poses", "gather_maps"]


def init_from_env(backend: Optional[str] = None):
    """Initialises torch.distributed from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun).
    Returns (rank, world, local_rank); a no-op single-process setup when WORLD_SIZE is unset."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:   # GRADSLAM_DIST_BACKEND=gloo: rehearsals of the N-rank path on a box with fewer GPUs
            backend = os.environ.get("GRADSLAM_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
            # device_id: the communicator is bound to this rank's GPU at once (eager init, no guessing from the first
            # collective's tensors)
            dist.init_process_group(backend=backend, rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_sequences(num_sequences: int, world: int, rank: int) -> List[int]:
    """Indices of the sequences rank `rank` owns: contiguous blocks, sizes differ by at most one."""
    if num_sequences < 0 or world < 1 or not (0 <= rank < world):
        raise ValueError("bad shard request: B={} world={} rank={}".format(num_sequences, world, rank))
    base, rem = divmod(num_sequences, world)
    start = rank * base + min(rank, rem)
    return list(range(start, start + base + (1 if rank < rem else 0)))


def _world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def _all_gather(out: List[torch.Tensor], t: torch.Tensor):
    """dist.all_gather; gloo has no device-tensor all_gather, so under gloo (CPU tests, single-GPU rehearsals of the
    N-rank path) device tensors are exchanged through the host."""
    if t.is_cuda and dist.get_backend() == "gloo":
        host = [torch.empty(o.shape, dtype=o.dtype) for o in out]
        dist.all_gather(host, t.cpu())
        for o, h in zip(out, host):
            o.copy_(h)
    else:
        dist.all_gather(out, t)


def _gather_rows(t: torch.Tensor) -> List[torch.Tensor]:
    """all_gather of a (n_r, ...) tensor whose first dimension differs per rank."""
    world = _world()
    if world == 1:
        return [t]
    n = torch.tensor([t.shape[0]], dtype=torch.int64, device=t.device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    _all_gather(counts, n)
    counts = [int(c.item()) for c in counts]
    cap = max(max(counts), 1)
    padded = torch.zeros((cap,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    padded[: t.shape[0]] = t
    out = [torch.empty_like(padded) for _ in range(world)]
    _all_gather(out, padded)
    return [o[:c] for o, c in zip(out, counts)]


def gather_poses(local_poses: torch.Tensor) -> torch.Tensor:
    """(B_local, L, 4, 4) on every rank -> (B_total, L, 4, 4) in rank order on every rank."""
    return torch.cat(_gather_rows(local_poses.contiguous()), 0)


def _gather_to(recv, send, dst):
    """dist.gather to rank dst (gloo: device tensors through the host, see _all_gather)"""
    me = dist.get_rank()
    if send.is_cuda and dist.get_backend() == "gloo":
        host = [torch.empty(send.shape, dtype=send.dtype) for _ in range(dist.get_world_size())] if me == dst else None
        dist.gather(send.cpu(), host, dst=dst)
        if me == dst:
            for o, h in zip(recv, host):
                o.copy_(h)
    else:
        dist.gather(send, recv if me == dst else None, dst=dst)


GATHER_PEAK_BYTES = 0            # bytes the last gather_maps call held at its peak on this rank (buffers + result so far)
ALL_GATHER_MAX_BYTES = 2 << 30   # per rank: beyond this an all-to-all map gather is refused (gather to one rank instead)


def gather_maps(pointclouds: Pointclouds, dst: Optional[int] = None,
                max_bytes: Optional[int] = None) -> Optional[Pointclouds]:
    """Every rank's local maps -> one Pointclouds holding all sequences (rank order).

    dst=None: on every rank (all_gather: every rank receives world x the largest rank's map, so this is for maps of
    benchmark size; above `max_bytes` (default ALL_GATHER_MAX_BYTES) received bytes per rank it falls back to dst=0 with
    a warning).  dst=r: on rank r only, the other ranks get None (gather: 1/world of the traffic and memory).
    The per-sequence counts and attribute widths of every rank travel first (a small object gather), then one padded
    (all_)gather per attribute (points, normals, colors, features)."""
    world = _world()
    if world == 1:
        return pointclouds.clone()   # (a new object, as with world > 1: callers may mutate the result)
    dev = pointclouds.device
    counts = list(pointclouds._n)
    lists = {k: getattr(pointclouds, k + "_list") for k in ("points", "normals", "colors", "features")}
    widths = {k: (0 if v is None or not len(v) else int(v[0].shape[-1])) for k, v in lists.items()}
    meta = [None] * world
    dist.all_gather_object(meta, (counts, widths))
    # attribute set of the merged map: whatever any rank has (ranks without sequences contribute nothing)
    w = {k: max(m[1][k] for m in meta) for k in widths}
    width = sum(w.values())
    rows = sum(counts)
    cap = max(max(sum(m[0]) for m in meta), 1)
    # what an all_gather makes EVERY rank hold at its peak: the gathered result (all rows of all ranks, every attribute)
    # plus one padded collective buffer set of the widest attribute (the collectives go attribute by attribute, below)
    total_rows = sum(sum(m[0]) for m in meta)
    peak_all = (total_rows * width + (world + 1) * cap * max(list(w.values()) + [0])) * 4
    if dst is None and peak_all > (ALL_GATHER_MAX_BYTES if max_bytes is None else max_bytes):
        import warnings
        warnings.warn("gather_maps: an all_gather of the maps would hold %.1f GB on every rank (result + one attribute's "
                      "collective buffers); gathering to rank 0 only" % (peak_all / 1e9))
        dst = 0
    # One padded (all_)gather PER ATTRIBUTE: the receiving rank then holds world x the widest attribute (3 floats per
    # surfel) at a time next to the result, instead of world x all 10 floats of every rank's padded map plus a packed and
    # a padded copy of its own (the peak of the one-collective form; GATHER_PEAK_BYTES reports what this call held).
    global GATHER_PEAK_BYTES
    me = dist.get_rank()
    out = {k: [] for k in w}
    peak = 0
    for k in ("points", "normals", "colors", "features"):
        if not w[k]:
            continue
        send = torch.zeros((cap, w[k]), dtype=torch.float32, device=dev)
        if widths[k] and rows:
            torch.cat(lists[k], 0, out=send[:rows])
        if dst is None:
            recv = [torch.empty_like(send) for _ in range(world)]
            _all_gather(recv, send)
        else:
            recv = [torch.empty_like(send) for _ in range(world)] if me == dst else None
            _gather_to(recv, send, dst)
        held = send.numel() * 4 * (1 + (world if recv is not None else 0)) + sum(t.numel() * 4 for v in out.values() for t in v)
        peak = max(peak, held)
        if recv is not None:
            for r, (cnts, _) in enumerate(meta):
                if cnts:
                    out[k].extend(t.clone() for t in torch.split(recv[r][: sum(cnts)], cnts, 0))
        del send, recv
    GATHER_PEAK_BYTES = peak
    if dst is not None and me != dst:
        return None
    if not out["points"]:
        return Pointclouds(device=dev)
    return Pointclouds(out["points"], out["normals"] or None, out["colors"] or None, out["features"] or None)
