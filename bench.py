#!/usr/bin/env python3
"""Newton systems/s (fp64) on a batch of constrained band NLS problems — BASELINE.json's metric.

One "step" = one `newton_system!` (KKT assembly + LDL^T + inertia + rho ladder + solve,
/root/reference/src/CaNNOLeS.jl:1008-1052) over a batch of B independent problems that share the
sparsity pattern of BASELINE config 3 (n = nequ = 1e4, ncon = 50, band Jacobians), inputs already
resident in HBM.  N > 1 GPUs: every rank owns its own shard of B problems (weak scaling, no
collective on the data path; torch.distributed is used only for the barrier and the max-over-ranks
time).  Prints ONE JSON line on rank 0.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus 8 --steps 20 --warmup 3
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); 6290 GB/s is the measured copy rate
CANONICAL_NNZL_CFG3 = 346209  # nnz(L) of the order "r, x natural, lambda" at n=1e4, p=50


def band_batch(s, B, seed):
    """Vectorised generator of B problems of the cfg3 family (same distributions as
    synthetic.band_values; one RNG stream for the whole batch)."""
    rng = np.random.default_rng(seed)
    off = s.offsets()
    vals = np.zeros((B, s.nnzNS))
    r, c = np.asarray(s.hF[0]), np.asarray(s.hF[1])
    dg = r == c
    h = rng.uniform(-0.02, 0.02, (B, len(r)))
    h[:, dg] = rng.uniform(0.1, 1.0, (B, int(dg.sum())))
    vals[:, off[0]:off[1]] = h
    r, c = np.asarray(s.jF[0]), np.asarray(s.jF[1])
    dg = r == c
    j = rng.uniform(-0.5, 0.5, (B, len(r)))
    j[:, dg] = 2.0 + rng.uniform(0, 1, (B, int(dg.sum())))
    vals[:, off[2]:off[3]] = j
    if s.ncon > 0:
        lam = rng.normal(size=(B, s.ncon))
        b = rng.uniform(-0.1, 0.1, (B, s.nvar))
        blk = (np.arange(s.nvar) // s.meta["block"]).astype(int)
        vals[:, off[1]:off[2]] = -(lam[:, blk] * b)
        vals[:, off[3]:off[4]] = rng.uniform(-1, 1, (B, s.nnzjc))
        vals[:, off[5]:off[6]] = -0.1
    vals[:, off[4]:off[5]] = -1.0
    rhs = rng.normal(size=(B, s.N))
    return vals, rhs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=int(os.environ.get("CNL_BENCH_BATCH", 8192)), help="problems per GPU")
    ap.add_argument("--n", type=int, default=10000)
    ap.add_argument("--ncon", type=int, default=50)
    ap.add_argument("--cpu-sample", type=int, default=-1, help="problems timed on the CPU oracle (-1 auto, 0 off)")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != args.gpus:
        if rank == 0:
            print(f"warning: WORLD_SIZE={world} but --gpus {args.gpus}; using WORLD_SIZE", file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist_.init_process_group("nccl", rank=rank, world_size=world)
        dist = dist_

    import cannoles_jl_amd  # noqa: F401
    from cannoles_jl_amd import hipldl, synthetic as syn

    s = syn.band_structure(args.n, args.ncon, name="cfg3")
    rows, cols = s.kkt_pattern()
    B = args.batch
    dev = torch.device("cuda", local_rank)
    # synthetic inputs are generated on the host in chunks (bounded host memory) and uploaded; only the first
    # chunk stays on the host, for the parity guard and the CPU-baseline sample
    vals = torch.empty((B, s.nnzNS), dtype=torch.float64, device=dev)
    rhs = torch.empty((B, s.N), dtype=torch.float64, device=dev)
    CH = 512
    vals_h = rhs_h = None
    for ci, b0 in enumerate(range(0, B, CH)):
        nb = min(CH, B - b0)
        vh, rh = band_batch(s, nb, seed=3000 + 1000 * rank + ci)
        vals[b0:b0 + nb].copy_(torch.from_numpy(vh))
        rhs[b0:b0 + nb].copy_(torch.from_numpy(rh))
        if ci == 0:
            vals_h, rhs_h = vh, rh
    d = torch.zeros((B, s.N), dtype=torch.float64, device=dev)
    rho_old = torch.zeros(B, dtype=torch.float64, device=dev)
    rho = torch.zeros(B, dtype=torch.float64, device=dev)
    nfact = torch.zeros(B, dtype=torch.int32, device=dev)
    succ = torch.zeros(B, dtype=torch.int32, device=dev)
    params = hipldl.default_params()

    LDLT = hipldl.HIPLDLStruct(s.N, rows, cols, None, s.nvar, s.nequ, s.ncon, batch=B, device=local_rank)
    stream = torch.cuda.Stream(device=dev)
    sh = stream.cuda_stream

    def step():
        rho_old.zero_()
        hipldl.newton_system_dev(LDLT, vals.data_ptr(), rhs.data_ptr(), d.data_ptr(), rho_old.data_ptr(), rho.data_ptr(),
                                 nfact.data_ptr(), succ.data_ptr(), params, sh)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.cuda.stream(stream):
        for _ in range(args.warmup):
            step()
        barrier()
        ev0 = torch.cuda.Event(enable_timing=True)
        ev1 = torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record(stream)
        for _ in range(args.steps):
            step()
        ev1.record(stream)
        barrier()
        t1 = time.perf_counter()
    elapsed = t1 - t0
    step_ms = ev0.elapsed_time(ev1) / max(args.steps, 1)  # HIP events on the launch stream: whole step (3 kernels)
    # duration of the dominant kernel alone (the multifrontal kernel), HIP events inside the library on the same
    # stream, measured over a few extra launches AFTER the timed region (timing mode synchronises every call)
    LDLT.set_timing(True)
    kms = []
    with torch.cuda.stream(stream):
        for _ in range(min(5, max(2, args.steps))):
            step()
            kms.append(LDLT.last_kernel_ms())
    LDLT.set_timing(False)
    kern_ms = float(np.mean(kms))
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # SURVEY 8 row f1 (vectors either side of the system, device-resident): timed after the headline region, on the
    # same shapes; algorithmic bytes = J_F and J_c values + r, lambda, F, c read + rhs written (resp. x, r, lambda, d
    # read + xt, rt, lambdat, dlambda written)
    f1 = None
    if rank == 0:
        rv = torch.randn((B, s.nequ), dtype=torch.float64, device=dev)
        lam = torch.randn((B, max(s.ncon, 1)), dtype=torch.float64, device=dev)
        Fx = torch.randn((B, s.nequ), dtype=torch.float64, device=dev)
        cx = torch.randn((B, max(s.ncon, 1)), dtype=torch.float64, device=dev)
        xv = torch.randn((B, s.nvar), dtype=torch.float64, device=dev)
        rhs2 = torch.empty_like(rhs)
        nrm = torch.zeros((B, 2), dtype=torch.float64, device=dev)
        xt, rt, lt, dl = torch.empty_like(xv), torch.empty_like(rv), torch.empty_like(lam), torch.empty_like(lam)

        def timed(fn, reps=5):
            with torch.cuda.stream(stream):
                fn()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                for _ in range(reps):
                    fn()
                e1.record(stream)
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / reps

        ms_rv = timed(lambda: hipldl.residual_vectors_dev(LDLT, vals.data_ptr(), rv.data_ptr(), lam.data_ptr(), Fx.data_ptr(),
                                                          cx.data_ptr(), rhs2.data_ptr(), nrm.data_ptr(), sh))
        ms_tp = timed(lambda: hipldl.trial_point_dev(LDLT, xv.data_ptr(), rv.data_ptr(), lam.data_ptr(), d.data_ptr(), 1e4,
                                                     xt.data_ptr(), rt.data_ptr(), lt.data_ptr(), dl.data_ptr(), sh))
        nnzj = len(s.jF[0]) + len(s.jc[0])
        by_rv = 8 * (nnzj + 2 * s.nequ + 2 * s.ncon + s.N)
        by_tp = 8 * (s.nvar + s.nequ + s.ncon + s.N + s.nvar + s.nequ + 2 * s.ncon)
        f1 = {"residual_vectors": {"ms": ms_rv, "bytes_per_system": by_rv, "GBps": by_rv * B / (ms_rv * 1e-3) / 1e9,
                                   "frac": by_rv * B / (ms_rv * 1e-3) / 1e9 / HBM_PEAK_GBPS},
              "trial_point": {"ms": ms_tp, "bytes_per_system": by_tp, "GBps": by_tp * B / (ms_tp * 1e-3) / 1e9,
                              "frac": by_tp * B / (ms_tp * 1e-3) / 1e9 / HBM_PEAK_GBPS}}
        del rv, lam, Fx, cx, xv, rhs2, nrm, xt, rt, lt, dl

    ok = bool((succ == 1).all().item())
    # parity guard inside the bench: residual of the first problems (size-independent property)
    nchk = min(B, 4)
    import scipy.sparse as sp
    d_h = d[:nchk].cpu().numpy()
    berr = 0.0
    for b in range(nchk):
        Kl = sp.coo_matrix((vals_h[b], (rows - 1, cols - 1)), shape=(s.N, s.N)).tocsr()
        Ks = Kl + sp.tril(Kl, -1).T
        res = Ks @ d_h[b] + rhs_h[b]
        berr = max(berr, np.abs(res).max() / (abs(Ks).sum(axis=1).max() * np.abs(d_h[b]).max() + np.abs(rhs_h[b]).max()))

    if rank == 0:
        systems = B * args.steps * world
        value = systems / elapsed
        nnzL_own = LDLT.info["nnzL"]
        nnzL_star = min(nnzL_own, CANONICAL_NNZL_CFG3) if (args.n, args.ncon) == (10000, 50) else nnzL_own
        b_alg = 12 * s.nnzNS + 24 * s.N + 32 * nnzL_star
        achieved = b_alg * B / (kern_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "r01_traffic.json")
        if os.path.exists(tpath):  # HBM bytes per launch of the dominant kernel from rocprofv3 PMC passes (see profiles/)
            tj = json.load(open(tpath))
            if tj.get("batch") == B and tj.get("workload") == [args.n, args.ncon]:
                traffic = tj.get("hbm_bytes_per_launch")
        out = {
            "metric": "Newton systems/sec (fp64), batched n=1e4 NLS; achieved HBM GB/s vs peak",
            "value": value, "unit": "systems/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"cfg3 band constrained NLS n={args.n} nequ={args.n} ncon={args.ncon}, "
                                   f"{B} independent problems per GPU, one newton_system! per problem per step",
                       "batch_per_gpu": B, "sharding": f"independent problems, {world} shard(s), no collective",
                       "ordering": LDLT.info["order"], "nnzL": nnzL_own, "fronts": LDLT.info["nsuper"],
                       "kernel": LDLT.config, "all_success": ok, "backward_error": berr},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "bytes_per_system": b_alg, "kernel_ms": kern_ms, "kernel": "newton2_kernel",
                         "step_ms": step_ms, "step_frac": b_alg * B / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS},
        }
        out["aux_f1"] = f1  # not part of `value`: rhs assembly + norms, trial point, roofline = HBM
        # CPU baseline: the oracle (restated LDLFactorizations path) on a bounded sample, 1 thread
        ncpu = args.cpu_sample
        if ncpu != 0 and world == 1:  # reported at N=1 only (the multi-GPU runs stay short)
            from oracle import oracle as O
            # the oracle gets the product's own fill-reducing order (the fairest CPU baseline; the canonical
            # r/x/lambda order of SURVEY.md has 2.4x the fill)
            orc = O.Oracle(s.N, rows, cols, LDLT.plan_array("perm").astype(np.int64))
            if ncpu < 0:
                tt = time.perf_counter()
                O.newton_system_batch(orc, 1, s.nvar, s.nequ, s.ncon, rhs_h[:1], vals_h[:1].copy(), None, params)
                one = time.perf_counter() - tt
                ncpu = int(max(4, min(len(vals_h), 12.0 / max(one, 1e-4))))
            tt = time.perf_counter()
            d0, ok0, _, _, nf0 = O.newton_system_batch(orc, ncpu, s.nvar, s.nequ, s.ncon, rhs_h[:ncpu], vals_h[:ncpu].copy(), None, params)
            tc = time.perf_counter() - tt
            dg = d[:min(ncpu, 8)].cpu().numpy()
            perr = float(np.abs(dg - d0[:len(dg)]).max() / np.abs(d0[:len(dg)]).max())
            out["cpu_baseline"] = {"value": ncpu / tc, "unit": "systems/s", "cores": 1, "kind": "port",
                                   "sample": f"first {ncpu} problems of rank 0's batch, oracle/cnl_oracle.c (restated "
                                             f"LDLFactorizations up-looking LDL^T, the product's own ordering, nnzL={orc.nnzL}), "
                                             f"host has {os.cpu_count()} logical cores",
                                   "max_rel_diff_vs_gpu": perr}
            # the same port on all host cores, one problem per thread ("(ii) all host cores"); each thread
            # owns an oracle object, the ctypes calls release the GIL.  Bounded to a few seconds; reported beside the
            # single-core figure, `cores` above stays the contract's figure.
            try:
                import concurrent.futures as cf
                nthr = max(1, min(len(os.sched_getaffinity(0)), 64, len(vals_h)))
                per = max(1, len(vals_h) // nthr)                   # problems per thread (its slice of the first chunk)
                perm64 = LDLT.plan_array("perm").astype(np.int64)
                orcs = [O.Oracle(s.N, rows, cols, perm64) for _ in range(nthr)]
                deadline = time.perf_counter() + 4.0  # bounded: every thread repeats its slice for about 4 s
                done = [0] * nthr

                def work(k):
                    lo = k * per
                    while time.perf_counter() < deadline:
                        O.newton_system_batch(orcs[k], per, s.nvar, s.nequ, s.ncon, rhs_h[lo:lo + per], vals_h[lo:lo + per].copy(), None, params)
                        done[k] += per

                tt = time.perf_counter()
                with cf.ThreadPoolExecutor(nthr) as ex:
                    list(ex.map(work, range(nthr)))
                ta = time.perf_counter() - tt
                out["cpu_baseline"]["all_cores"] = {"value": sum(done) / ta, "threads": nthr, "systems": sum(done)}
            except Exception as e:  # the baseline is a reported figure, never a reason to fail the bench
                out["cpu_baseline"]["all_cores"] = {"error": str(e)}
        print(json.dumps(out))
    LDLT.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
