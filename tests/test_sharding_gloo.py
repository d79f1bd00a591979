"""The N > 1 path: contiguous shards of independent problems, no data-path collective.  CPU processes over gloo run
the SAME per-rank driver bench.py runs on the GPU box (cannoles.jl_amd/sharding.py::run_shard: shard of the job, executor,
warm-up, barrier-bracketed timed region, max-over-ranks time, job-wide counts) with a stub executor in place of the HIP
handle (the stub solves its shard with the CPU oracle, the only CPU executor there is)."""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

import cannoles_jl_amd  # noqa: F401
from cannoles_jl_amd import sharding, synthetic as syn


def test_shard_range_covers_batch():
    for total in (0, 1, 5, 7, 10, 256, 257):
        for world in (1, 2, 3, 4, 8):
            seen, sizes = [], []
            for r in range(world):
                a, b = sharding.shard_range(total, world, r)
                assert 0 <= a <= b <= total
                seen += list(range(a, b))
                sizes.append(b - a)
            assert seen == list(range(total))
            assert max(sizes) - min(sizes) <= 1   # balanced: 10 over 4 is 3, 3, 2, 2 (not 3, 3, 3, 1)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _StubExecutor:
    """what bench.py's DeviceProblem is on the GPU: owns the shard's inputs, step() = one newton_system! over the shard"""

    def __init__(self, g0, g1, total):
        from oracle import oracle as O
        self.O = O
        self.s = syn.band_structure(60, 3)
        rows, cols = self.s.kkt_pattern()
        vals, rhs = syn.batch_values(self.s, total, cfg=4)   # every rank can regenerate the whole synthetic batch
        self.vals, self.rhs, self.n = vals[g0:g1], rhs[g0:g1], g1 - g0
        self.orc = O.Oracle(self.s.N, rows, cols, O.canonical_perm(self.s.nvar, self.s.nequ, self.s.ncon))
        self.nsteps = 0

    def step(self):
        s = self.s
        self.d, self.ok, _, _, self.nf = self.O.newton_system_batch(self.orc, self.n, s.nvar, s.nequ, s.ncon, self.rhs, self.vals.copy(), None,
                                                                    self.O.default_params())
        self.nsteps += 1

    def counts(self):
        return [self.n, int(self.ok.sum()), int(self.nf.sum())]


def _worker(rank, world, port, total, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    dist = sharding.init("gloo")
    elapsed, counts, ex, (g0, g1) = sharding.run_shard(total, lambda a, b: _StubExecutor(a, b, total), steps=2, warmup=1, dist=dist)
    tmax = sharding.max_over_ranks(1.0 + rank, dist)
    q.put((rank, counts, tmax, float(np.abs(ex.d).sum()) if ex else 0.0, ex.nsteps if ex else 0, elapsed, (g0, g1)))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,total", [(2, 9), (3, 2)])
def test_gloo_sharding_runs_the_per_rank_driver(world, total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ranges = []
    for rank, counts, tmax, _, nsteps, elapsed, rng in res:
        assert counts == [total, total, total]   # every problem solved once, one factorisation each
        assert tmax == float(world)              # max over ranks
        assert nsteps == (3 if rng[1] > rng[0] else 0)  # warm-up + timed steps; an empty shard only joins the barriers
        assert elapsed > 0
        ranges.append(rng)
    assert [r[0] for r in ranges] == [0] + [r[1] for r in ranges[:-1]] and ranges[-1][1] == total  # contiguous cover
    # the shards are disjoint pieces of the same batch: checksums differ and add up to the unsharded run
    from oracle import oracle as O
    s = syn.band_structure(60, 3)
    rows, cols = s.kkt_pattern()
    vals, rhs = syn.batch_values(s, total, cfg=4)
    orc = O.Oracle(s.N, rows, cols, O.canonical_perm(s.nvar, s.nequ, s.ncon))
    d, *_ = O.newton_system_batch(orc, total, s.nvar, s.nequ, s.ncon, rhs, vals.copy(), None, O.default_params())
    assert abs(sum(r[3] for r in res) - float(np.abs(d).sum())) <= 1e-9 * float(np.abs(d).sum())
