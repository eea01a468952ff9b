"""The N > 1 path on CPU: shard ranges and the score gather with the gloo backend, world sizes 2 and 8 (the node BASELINE configs[2]
names: uneven shards, ranks with nothing to do, the padding of the equal-size gather).  The GPU runs use the same code with backend
nccl == RCCL."""
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


def _worker_bench_shape(rank, world, port, n_total, out_path):
    """What one rank of `bench.py --gpus 8 --pairs n_total` does around its kernels: its contiguous shard of the ONE seeded workload
    (blocks of the generator cut at shard borders), a score per pair, the barrier, the gather onto rank 0, the max-over-ranks timing."""
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from mgl_amd import device_batch as db

    db.WORKLOAD_BLOCK = 1 << 10
    r, lr, w = dist.init(backend="gloo")
    lo, hi = dist.shard_range(n_total, rank, world)
    cpu = torch.device("cpu")
    if hi > lo:
        b = db.window_batch(5, hi - lo, cpu, genome_len=1 << 16, first=lo)
        local = b.queries.view(hi - lo, -1).to(torch.int32).sum(dim=1).to(torch.int32)   # stand-in for ScoreMax.max: a function of the pair alone
    else:
        local = torch.zeros(0, dtype=torch.int32)
    dist.barrier()
    got = dist.gather_scores(local, n_total, dst=0)
    assert dist.max_over_ranks(float(rank), cpu) == float(world - 1)
    assert (got is not None) == (rank == 0)
    if rank == 0:
        torch.save(got, out_path)
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("n_total", [100003, 5])
def test_world_size_8_uneven_shards_and_idle_ranks(tmp_path, n_total):
    """BASELINE configs[2]'s node on CPU: eight ranks over gloo, n_total not divisible by eight (100 003: shards of 12 501 and 12 500;
    5: three ranks hold nothing and still take part in every collective); the gathered vector equals what ONE rank computes for the
    whole seeded workload."""
    from mgl_amd import device_batch as db

    out = str(tmp_path / "gathered8.pt")
    mp.spawn(_worker_bench_shape, args=(8, _free_port(), n_total, out), nprocs=8, join=True)
    got = torch.load(out)
    block = db.WORKLOAD_BLOCK
    try:
        db.WORKLOAD_BLOCK = 1 << 10
        full = db.window_batch(5, n_total, torch.device("cpu"), genome_len=1 << 16)
    finally:
        db.WORKLOAD_BLOCK = block
    want = full.queries.view(n_total, -1).to(torch.int32).sum(dim=1).to(torch.int32)
    assert got.dtype == torch.int32 and got.shape == (n_total,) and torch.equal(got, want)


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
