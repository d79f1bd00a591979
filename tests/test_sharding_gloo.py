"""The N > 1 path: contiguous shards of independent problems, no data-path collective.  Two CPU
processes over gloo: every rank runs its shard (here through the oracle, the only CPU executor) and the
aggregated counts and the max-over-ranks timing helper behave as bench.py expects."""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

import cannoles_jl_amd  # noqa: F401
from cannoles_jl_amd import sharding, synthetic as syn


def test_shard_range_covers_batch():
    for total in (1, 7, 256, 257):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                a, b = sharding.shard_range(total, world, r)
                assert 0 <= a <= b <= total
                seen += list(range(a, b))
            assert seen == list(range(total))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import oracle as O
    dist = sharding.init("gloo")
    s = syn.band_structure(60, 3)
    rows, cols = s.kkt_pattern()
    a, b = sharding.shard_range(total, world, rank)
    vals, rhs = syn.batch_values(s, total, cfg=4)   # every rank can regenerate the whole synthetic batch
    orc = O.Oracle(s.N, rows, cols, O.canonical_perm(s.nvar, s.nequ, s.ncon))
    d, ok, rho, ro, nf = O.newton_system_batch(orc, b - a, s.nvar, s.nequ, s.ncon, rhs[a:b], vals[a:b].copy(), None, O.default_params())
    dist.barrier()
    counts = sharding.gather_counts([b - a, int(ok.sum()), int(nf.sum())], dist)
    tmax = sharding.max_over_ranks(1.0 + rank, dist)
    q.put((rank, counts, tmax, float(np.abs(d).sum())))
    dist.destroy_process_group()


def test_two_rank_gloo_sharding():
    world, total = 2, 9
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
    for rank, counts, tmax, _ in res:
        assert counts == [total, total, total]   # every problem solved once, one factorisation each
        assert tmax == 2.0                       # max over ranks
    # the shards are disjoint pieces of the same batch: checksums differ and add up to the unsharded run
    from oracle import oracle as O
    s = syn.band_structure(60, 3)
    rows, cols = s.kkt_pattern()
    vals, rhs = syn.batch_values(s, total, cfg=4)
    orc = O.Oracle(s.N, rows, cols, O.canonical_perm(s.nvar, s.nequ, s.ncon))
    d, *_ = O.newton_system_batch(orc, total, s.nvar, s.nequ, s.ncon, rhs, vals.copy(), None, O.default_params())
    assert abs(sum(r[3] for r in res) - float(np.abs(d).sum())) <= 1e-9 * float(np.abs(d).sum())
