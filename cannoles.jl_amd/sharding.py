"""Multi-GPU placement of a batch of independent Newton systems.

The path shards only across independent problems: contiguous, balanced shards, one process
and one device per shard, no collective on the data path.  torch.distributed (RCCL on the GPU box, gloo
in the CPU tests) is used for the barrier that brackets the timed region, the max-over-ranks time and
the gather of per-shard status vectors.
"""
import os


def shard_range(total, world, rank):
    """[start, stop) of the problems owned by `rank`: contiguous, balanced (the first total % world ranks own one
    problem more, so sizes differ by at most one).  With total < world the trailing ranks own nothing (start == stop):
    such a rank creates no handle and only joins the barriers."""
    if world < 1 or not 0 <= rank < world or total < 0:
        raise ValueError("shard_range: need 0 <= rank < world and total >= 0")
    base, rem = divmod(total, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def env_rank():
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def init(backend):
    """process group for the barrier / reductions; MASTER_ADDR defaults to 127.0.0.1"""
    import torch.distributed as dist
    rank, _, world = env_rank()
    if world <= 1:
        return None
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if not dist.is_initialized():
        dist.init_process_group(backend, rank=rank, world_size=world)
    return dist


def max_over_ranks(value, dist, device="cpu"):
    import torch
    if dist is None:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_counts(local_counts, dist, device="cpu"):
    """sum over ranks of a small integer vector (e.g. [problems, successes, factorisations])"""
    import torch
    t = torch.tensor(list(local_counts), dtype=torch.int64, device=device)
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [int(v) for v in t.tolist()]


def timed_region(step, steps, warmup, dist, sync=None, device="cpu"):
    """The timed region of a sharded job, as every rank runs it (bench.py with the HIP executor, the gloo test with a CPU
    stub): `warmup` untimed steps, then exactly `steps` steps bracketed by sync + barrier + sync on both sides; returns the
    MAX over ranks of the wall-clock time.  `step` may be None on a rank whose shard is empty (it still joins the barriers).
    `sync` = device synchronisation (torch.cuda.synchronize on the GPU box, nothing on the CPU)."""
    import time

    def barrier():
        if sync:
            sync()
        if dist is not None:
            dist.barrier()
        if sync:
            sync()

    for _ in range(warmup):
        if step:
            step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        if step:
            step()
    barrier()
    return max_over_ranks(time.perf_counter() - t0, dist, device=device)


def run_shard(total, make_executor, steps, warmup, dist, sync=None, device="cpu"):
    """Per-rank driver: this rank's shard [g0, g1) of `total` problems, an executor for it (`make_executor(g0, g1)` returns an
    object with step() and counts() -> [problems, successes, ...], or None for an empty shard), the timed region, and the
    job-wide sums of the counts.  Returns (elapsed_max_over_ranks, summed_counts, executor, (g0, g1))."""
    rank, _, world = env_rank()
    g0, g1 = shard_range(total, world, rank)
    ex = make_executor(g0, g1) if g1 > g0 else None
    elapsed = timed_region(ex.step if ex else None, steps, warmup, dist, sync=sync, device=device)
    local = ex.counts() if ex else None
    width = len(local) if local is not None else 0
    if dist is not None:  # every rank must contribute a vector of the same length
        import torch
        w = torch.tensor([width], dtype=torch.int64, device=device)
        dist.all_reduce(w, op=dist.ReduceOp.MAX)
        width = int(w.item())
    counts = gather_counts(local if local is not None else [0] * width, dist, device=device)
    return elapsed, counts, ex, (g0, g1)
