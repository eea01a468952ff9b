"""Multi-GPU plumbing: read pairs shard embarrassingly (each pair is a pure function of its two
sequences, ..._MicrosoftSmithWaterman.cpp:44-71), so there is no data-path collective; the only
exchange is one gather of the int32 alignment scores onto rank 0 at the end (RCCL over xGMI on
GPUs, gloo in the CPU tests).  One process per GPU, launched by torch.distributed.run."""
import os

import torch
import torch.distributed as dist


def env_world():
    """(rank, local_rank, world_size) from the torchrun environment (1 process: 0, 0, 1)."""
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def init(backend=None):
    """Initialise the default process group when launched by torch.distributed.run (RANK in the
    environment) -- also for a single rank, so that a 1-GPU torchrun exercises the RCCL path."""
    rank, local_rank, world = env_world()
    if "RANK" in os.environ and not dist.is_initialized():
        if backend is None:
            # MGL_DIST_BACKEND=gloo: rehearsal of the multi-rank path where the ranks share one GPU (RCCL refuses that)
            backend = os.environ.get("MGL_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            kw["device_id"] = torch.device("cuda", local_rank)  # binds the communicator to this rank's GPU
        elif torch.cuda.is_available():
            torch.cuda.set_device(local_rank % torch.cuda.device_count())
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, local_rank, world


def shard_range(n_total, rank, world):
    """Contiguous shard [lo, hi) of n_total pairs for ``rank``; sizes differ by at most one."""
    base, extra = divmod(int(n_total), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_counts(n_total, world):
    return [shard_range(n_total, r, world)[1] - shard_range(n_total, r, world)[0] for r in range(world)]


def gather_scores(local_scores, n_total=None, dst=0):
    """Gather each rank's int32 score vector onto ``dst`` in shard order.

    Returns the concatenated [n_total] tensor on dst and None elsewhere.  Shards may differ in
    length by one (shard_range): every rank pads to the longest shard so that the collective is
    a plain equal-size gather.
    """
    if not dist.is_initialized():
        return local_scores
    world, rank = dist.get_world_size(), dist.get_rank()
    if n_total is None:
        n = torch.tensor([local_scores.numel()], device=local_scores.device, dtype=torch.int64)
        dist.all_reduce(n)
        n_total = int(n.item())
    counts = shard_counts(n_total, world)
    assert local_scores.numel() == counts[rank], (local_scores.numel(), counts[rank])
    width = max(counts)
    send = local_scores
    if send.numel() != width:
        send = torch.zeros(width, dtype=local_scores.dtype, device=local_scores.device)
        send[: local_scores.numel()] = local_scores
    send = send.contiguous()
    dev = send.device
    if dist.get_backend() == "gloo" and send.is_cuda:
        send = send.cpu()  # gloo moves host memory
    # one collective, the same on every rank (both backends have gather); an error propagates: after a failed
    # collective the communicator is unusable and the ranks must not fall into different calls
    recv = [torch.empty_like(send) for _ in range(world)] if rank == dst else None
    dist.gather(send, recv, dst=dst)
    if rank != dst:
        return None
    return torch.cat([recv[r][: counts[r]] for r in range(world)]).to(dev)


def max_over_ranks(value, device):
    """MAX of a python float over all ranks (for timing)."""
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([float(value)], device="cpu" if dist.get_backend() == "gloo" else device, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def all_ranks(value, device):
    """The python float of every rank, in rank order, on every rank (per-rank step times for the bench line)."""
    if not dist.is_initialized():
        return [float(value)]
    t = torch.tensor([float(value)], device="cpu" if dist.get_backend() == "gloo" else device, dtype=torch.float64)
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [float(x.item()) for x in out]


def min_over_ranks(value, device):
    """MIN of a python number over all ranks (a decision every rank must take the same way, e.g. steps in flight)."""
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([float(value)], device="cpu" if dist.get_backend() == "gloo" else device, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return float(t.item())


def describe():
    """What the communicator itself reports (not the environment): {"world_size", "backend"}; one process: world_size 1, no backend."""
    if not dist.is_initialized():
        return {"world_size": 1, "backend": None}
    return {"world_size": int(dist.get_world_size()), "backend": str(dist.get_backend())}


def barrier():
    if dist.is_initialized():
        if dist.get_backend() == "nccl":
            dist.barrier(device_ids=[torch.cuda.current_device()])
        else:
            dist.barrier()
