"""CPU suite: the N>1 path (batch sharding + one all_gather of per-rank metrics) with gloo, world_size 2.

The hot path has no data-path collective: rank r owns a contiguous slice of the batch and
only a small metric vector is exchanged at the end.  Here each rank pushes its slice through the CPU
oracle (no GPU in this suite) and the gathered metrics must reproduce the single-process result.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import pdepth_amd  # noqa: F401
from pdepth_amd import dist as pdist
from pdepth_amd import synth
from util import oracle_batch


def test_shard_range_partitions_exactly():
    for n in (0, 1, 4, 7, 16, 32, 33):
        for world in (1, 2, 3, 8):
            spans = [pdist.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, B, out_path):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    torch.set_num_threads(2)
    r, w, _ = pdist.init_from_env(backend="gloo")
    assert (r, w) == (rank, world) and dist.get_backend() == "gloo"
    lo, hi = pdist.shard_range(B, rank, world)
    batch = synth.make_batch(5, hi - lo, first_item=lo, C=5, D=8, H=12, W=16, V=1, pose="mono")
    _, _, depth = oracle_batch(batch)
    vec = torch.tensor([hi - lo, float(depth.sum()), float(depth.max()), float(rank)], dtype=torch.float32)
    pdist.barrier()
    allm = pdist.gather_metrics(vec)
    slowest = pdist.max_over_ranks(1.0 + rank, torch.device("cpu"))
    if rank == 0:
        np.save(out_path, np.concatenate([allm.numpy().ravel(), [slowest]]))
    pdist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_sharding_matches_single_process(tmp_path):
    B, world = 5, 2
    out = str(tmp_path / "gathered.npy")
    mp.spawn(_worker, args=(world, _free_port(), B, out), nprocs=world, join=True)
    got = np.load(out)
    allm, slowest = got[:-1].reshape(world, 4), got[-1]
    assert slowest == 2.0  # max over ranks of (1 + rank)
    assert allm[:, 0].tolist() == [3.0, 2.0] and allm[:, 3].tolist() == [0.0, 1.0]
    full = synth.make_batch(5, B, C=5, D=8, H=12, W=16, V=1, pose="mono")
    _, _, depth = oracle_batch(full)
    np.testing.assert_allclose(allm[:, 1].sum(), float(depth.sum()), rtol=1e-6)
    np.testing.assert_allclose(allm[:, 2].max(), float(depth.max()), rtol=1e-6)


def test_single_process_paths_need_no_process_group():
    v = torch.tensor([1.0, 2.0])
    assert torch.equal(pdist.gather_metrics(v), v[None])
    assert pdist.max_over_ranks(3.5, torch.device("cpu")) == 3.5
    pdist.barrier()
