"""The N > 1 path on CPU: shard ranges and the score gather with the gloo backend, world_size 2.
(The GPU runs use the same code with backend nccl == RCCL.)"""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

from mgl_amd import dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_ranges_cover_and_balance():
    for n in (0, 1, 7, 8, 9, 1000, 10_000_000):
        for world in (1, 2, 3, 8):
            spans = [dist.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[r][1] == spans[r + 1][0] for r in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1 and sizes == dist.shard_counts(n, world)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_total, out_path):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    r, lr, w = dist.init(backend="gloo")
    assert (r, w) == (rank, world)
    lo, hi = dist.shard_range(n_total, rank, world)
    # stand-in scores: the global pair index times 3 (the real ones come from the GPU kernels)
    local = (torch.arange(lo, hi, dtype=torch.int32) * 3)
    dist.barrier()
    got = dist.gather_scores(local, n_total, dst=0)
    t = dist.max_over_ranks(float(rank + 1), torch.device("cpu"))
    assert t == float(world)
    if rank == 0:
        assert got is not None and got.dtype == torch.int32
        torch.save(got, out_path)
    else:
        assert got is None
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("n_total", [10, 11])
def test_gather_scores_gloo_world2(tmp_path, n_total):
    out = str(tmp_path / "gathered.pt")
    mp.spawn(_worker, args=(2, _free_port(), n_total, out), nprocs=2, join=True)
    got = torch.load(out)
    assert got.tolist() == [3 * k for k in range(n_total)]


def test_single_process_is_identity():
    x = torch.arange(5, dtype=torch.int32)
    assert dist.gather_scores(x) is x
    assert dist.max_over_ranks(1.5, torch.device("cpu")) == 1.5


def _solo_worker(rank, port, out_path):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r, lr, w = dist.init(backend="gloo")
    got = dist.gather_scores(torch.arange(7, dtype=torch.int32), 7, dst=0)
    dist.barrier()
    torch.save(got, out_path)
    torch.distributed.destroy_process_group()


def test_world_size_one_under_torchrun_env(tmp_path):
    """torchrun with one rank still builds the process group and runs the gather (what a 1-GPU
    `python -m torch.distributed.run --nproc-per-node 1 bench.py` does)."""
    out = str(tmp_path / "solo.pt")
    mp.spawn(_solo_worker, args=(_free_port(), out), nprocs=1, join=True)
    assert torch.load(out).tolist() == list(range(7))


def test_workload_shards_concatenate_to_the_global_batch(monkeypatch):
    """bench.py --gpus N (strong scaling, BASELINE configs[2]): rank r generates only pairs shard_range(n, r, N) of the
    ONE seeded workload; the shards of any world size concatenate to what a single rank generates."""
    from mgl_amd import device_batch as db

    monkeypatch.setattr(db, "WORKLOAD_BLOCK", 1 << 10)   # several blocks at test size
    cpu = torch.device("cpu")
    n = 5000
    full = db.window_batch(5, n, cpu, genome_len=1 << 16)
    assert full.n == n and full.uniform
    for world in (2, 3, 8):
        parts = [db.window_batch(5, hi - lo, cpu, genome_len=1 << 16, first=lo)
                 for lo, hi in (dist.shard_range(n, r, world) for r in range(world))]
        assert torch.equal(torch.cat([b.targets for b in parts]), full.targets)
        assert torch.equal(torch.cat([b.queries for b in parts]), full.queries)
        assert torch.equal(torch.cat([b.win for b in parts]), full.win)
    other = db.window_batch(6, 64, cpu, genome_len=1 << 16)
    assert not torch.equal(other.queries, full.queries[:64 * 150])
