"""N>1 path on CPU: world_size-2 gloo processes shard a batch of sequences, each builds its own
(ragged) maps and poses, and the final gather must reproduce the unsharded result on every rank.
No kernels are involved: the sharded data path has no collective, only this final exchange."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gradslam_amd import multigpu
from gradslam_amd.structures.pointclouds import Pointclouds


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _fake_sequence_result(seq_id, L=3):
    """Deterministic stand-in for one sequence's SLAM output: ragged map + poses."""
    rng = np.random.default_rng(100 + seq_id)
    n = 50 + 37 * seq_id
    pts = torch.from_numpy(rng.standard_normal((n, 3)).astype(np.float32))
    feats = torch.from_numpy(rng.random((n, 1)).astype(np.float32))
    poses = torch.eye(4).repeat(L, 1, 1)
    poses[:, 0, 3] = seq_id + torch.arange(L) * 0.01
    return pts, pts * 2, pts + 1, feats, poses


def _worker(rank, world, port, B, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    r, w, _ = multigpu.init_from_env("gloo")
    assert (r, w) == (rank, world)
    mine = multigpu.shard_sequences(B, world, rank)
    res = [_fake_sequence_result(i) for i in mine]
    if res:
        pc = Pointclouds(points=[x[0] for x in res], normals=[x[1] for x in res], colors=[x[2] for x in res],
                         features=[x[3] for x in res])
        poses = torch.stack([x[4] for x in res])
    else:
        pc, poses = Pointclouds(), torch.zeros((0, 3, 4, 4))
    all_poses = multigpu.gather_poses(poses)
    all_maps = multigpu.gather_maps(pc)
    # gather to one rank only: the others get None, rank 1 the same map as the all_gather
    only1 = multigpu.gather_maps(pc, dst=1)
    assert (only1 is None) == (rank != 1)
    if rank == 1:
        assert only1._n == all_maps._n and all(torch.equal(a, b) for a, b in zip(only1.points_list, all_maps.points_list))
    # an all_gather beyond the size limit falls back to rank 0 (with a warning)
    import warnings
    with warnings.catch_warnings(record=True) as wrn:
        warnings.simplefilter("always")
        capped = multigpu.gather_maps(pc, max_bytes=1)
    assert (capped is None) == (rank != 0) and any("rank 0 only" in str(x.message) for x in wrn)
    torch.save({"poses": all_poses, "points": all_maps.points_list, "features": all_maps.features_list,
                "colors": all_maps.colors_list, "n": all_maps._n}, os.path.join(out_dir, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("B", [2, 5])
def test_shard_and_gather_world2(tmp_path, B):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, B, str(tmp_path)), nprocs=world, join=True)
    expect = [_fake_sequence_result(i) for i in range(B)]
    for rank in range(world):
        got = torch.load(os.path.join(str(tmp_path), "rank%d.pt" % rank))
        assert got["n"] == [e[0].shape[0] for e in expect]
        assert torch.equal(got["poses"], torch.stack([e[4] for e in expect]))
        for b in range(B):
            assert torch.equal(got["points"][b], expect[b][0])
            assert torch.equal(got["colors"][b], expect[b][2])
            assert torch.equal(got["features"][b], expect[b][3])


def test_shard_sequences_partition():
    for B in range(0, 20):
        for world in (1, 2, 3, 4, 8):
            parts = [multigpu.shard_sequences(B, world, r) for r in range(world)]
            assert sorted(sum(parts, [])) == list(range(B))
            sizes = [len(p) for p in parts]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        multigpu.shard_sequences(4, 2, 2)


def test_single_process_gather_is_identity():
    pts, nr, col, ft, poses = _fake_sequence_result(0)
    pc = Pointclouds(points=[pts], normals=[nr], colors=[col], features=[ft])
    out = multigpu.gather_maps(pc)
    assert out is not pc   # a new object, as with world > 1
    assert torch.equal(out.points_list[0], pts) and torch.equal(multigpu.gather_poses(poses[None]), poses[None])
