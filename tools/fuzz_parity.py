"""Randomised parity run (GPU box; not part of the test-suite): many small irregular and band structures, batch sizes that are
not multiples of four, both plan kinds, positive-definite and indefinite top-left blocks (the latter climb the rho ladder) — the
host-pointer newton_system!, then try_to_factorize + solve_ldl! with the rho the ladder left, against the oracle on the order
SURVEY 8d prices (oracle.canonical_perm), i.e. an order the product had no part in.  Decisions bit for bit, d to 1e-8.
usage: fuzz_parity.py [cases] [first seed]      prints one line per failure and a summary; exit code 1 on any failure"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cannoles_jl_amd  # noqa: F401,E402
from cannoles_jl_amd import hipldl, synthetic as syn  # noqa: E402
from oracle import oracle as O  # noqa: E402

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
p = hipldl.default_params()
po = O.default_params()
# FUZZ_OPTS="device_ladder=0,dataflow=0": cnl_options fields applied to every case (bisecting a failure)
extra = {k: int(v) for k, v in (kv.split("=") for kv in os.environ.get("FUZZ_OPTS", "").split(",") if kv)}
fails = 0
subnoise = 0
kinds = {}
for case in range(ncases):
    seed = seed0 + case
    rng = np.random.default_rng(100000 + seed)
    fam = rng.integers(3) if not os.environ.get("FUZZ_WIDE") else rng.integers(7)
    if fam == 3:     # dense Jacobian (BASELINE config 2's shape, small): the dense backend with its device-side ladder
        n = int(rng.integers(20, 150)); m = int(rng.integers(max(2, n // 2), 2 * n + 1))   # also fewer residuals than variables
        s = syn.dense_structure(n, m)
    elif fam == 6:   # degenerate patterns: residual rows without entries, variables no residual touches, constraints on one variable, duplicates
        n = int(rng.integers(4, 70)); m = int(rng.integers(1, 2 * n)); pc = int(rng.integers(0, 4))
        mask = rng.uniform(size=(m, n)) < float(rng.uniform(0.02, 0.2))
        mask[:, rng.integers(n)] = False                      # a variable without residual entries
        if m > 2:
            mask[rng.integers(m), :] = False                  # a residual row without entries
        jr_, jc_ = np.nonzero(mask)
        if len(jr_) == 0:
            jr_, jc_ = np.array([0]), np.array([0])
        dup = rng.integers(len(jr_), size=max(1, len(jr_) // 8))   # duplicate Jacobian entries (summed, src/solver_types.jl:53-59)
        jr_, jc_ = np.concatenate([jr_, jr_[dup]]), np.concatenate([jc_, jc_[dup]])
        hd = np.arange(1, n + 1)
        hF_ = (np.concatenate([hd, hd[: n // 3]]), np.concatenate([hd, hd[: n // 3]]))   # diagonal, some entries twice
        if pc:
            cr_ = np.repeat(np.arange(1, pc + 1), 2); cc_ = rng.integers(1, n + 1, size=2 * pc)
            hc_s = (hd[: n // 2], hd[: n // 2])
        else:
            cr_ = cc_ = np.zeros(0, np.int64); hc_s = (np.zeros(0, np.int64), np.zeros(0, np.int64))
        s = syn.Structure(n, m, pc, hF_, hc_s, (jr_ + 1, jc_ + 1), (cr_, cc_), name="degenerate")
    elif fam == 5:   # a batch just above a (small) staged_max_batch: the chain + a remainder handle, or two halves (capi.cpp, run_split)
        pc = int(rng.integers(1, 4)); blocks = int(rng.integers(30, 80)); n = pc * blocks
        s = syn.band_structure(n, pc, hw=2)
    elif fam == 4:   # band family at larger batches: staged plans with many groups of problems, the in-kernel ladder over several groups
        pc = int(rng.integers(1, 6)); blocks = int(rng.integers(20, 80)); n = pc * blocks
        s = syn.band_structure(n, pc, hw=int(rng.integers(1, 3)))
    elif fam == 0:
        n = int(rng.integers(6, 120)); m = int(rng.integers(max(2, n // 2), 2 * n)); pc = int(rng.integers(0, min(6, n // 2) + 1))
        s = syn.random_structure(n, m, pc, float(rng.uniform(0.03, 0.3)), seed, hess=bool(rng.integers(4)))
    elif fam == 1:
        pc = int(rng.integers(0, 5)); blocks = int(rng.integers(8, 60)); n = (pc if pc else 1) * blocks
        s = syn.band_structure(n, pc, hw=int(rng.integers(1, 4)))
    else:
        n = int(rng.integers(130, 400)); m = int(rng.integers(n, n + 60)); pc = int(rng.integers(0, 4))
        s = syn.random_structure(n, m, pc, float(rng.uniform(0.01, 0.04)), seed)
    B = int(rng.choice([1, 2, 3, 5, 7, 17, 33, 64]))
    if fam == 4:
        B = int(rng.choice([66, 130, 257, 514, 1023]))
    smb = 0
    if fam == 5:
        smb = int(rng.choice([32, 64, 128]))
        B = smb + int(rng.choice([1, 2, 3, 4, 5, 9, smb // 4, smb // 4 + 1, smb // 2, smb - 8]))
    posdef = bool(rng.integers(3))
    if fam in (1, 4, 5):
        vals, rhs = syn.batch_values(s, B, cfg=seed % 7, stress=None if posdef else "ladder")
        if fam in (4, 5) and not posdef:   # only some problems climb
            vg, rg = syn.batch_values(s, B, cfg=3)
            keep = rng.uniform(size=B) < 0.7
            vals[keep], rhs[keep] = vg[keep], rg[keep]
    elif fam == 3:
        vals = np.empty((B, s.nnzNS)); rhs = np.empty((B, s.N))
        for b in range(B):
            vals[b], rhs[b] = syn.dense_values(s, 9000 * seed + b)
        if not posdef:
            off = s.offsets()
            hr, hc_ = np.asarray(s.hF[0]), np.asarray(s.hF[1])
            dgi = off[0] + np.nonzero(hr == hc_)[0]
            for b in range(B):
                if rng.integers(2):
                    vals[b, dgi[: max(1, len(dgi) // 3)]] = -5.0
    else:
        vals = np.empty((B, s.nnzNS)); rhs = np.empty((B, s.N))
        for b in range(B):
            vals[b], rhs[b] = syn.random_values(s, 7000 * seed + b, posdef=posdef or bool(rng.integers(2)))
    if os.environ.get("FUZZ_WIDE") and rng.integers(3) == 0:   # badly scaled problems: every problem of the batch by its own power of ten
        sc = 10.0 ** rng.integers(-5, 6, B)
        off_ = s.offsets()
        vals = vals * sc[:, None]
        vals[:, off_[4]:off_[5]] = -1.0    # the -I block is not the caller's to scale (src/CaNNOLeS.jl:970)
        rhs = rhs * sc[:, None]
    ro_in = np.where(rng.uniform(size=B) < 0.3, 10.0 ** rng.uniform(-6, -1, B), 0.0)
    rows, cols = s.kkt_pattern()
    kind = [hipldl.PLAN_AUTO, hipldl.PLAN_THROUGHPUT, hipldl.PLAN_LATENCY][int(rng.integers(3))]
    ropt = {}
    if fam == 5:
        kind = hipldl.PLAN_AUTO
        ropt["staged_max_batch"] = smb
        if rng.integers(3) == 0:
            ropt["split_tail"] = 0
        if rng.integers(4) == 0:
            ropt["split_batch"] = 2
    if os.environ.get("FUZZ_WIDE"):   # random execution switches on top
        for k, vs in (("dataflow", (1, 0)), ("staged", (1, 0)), ("dense_backend", (1, 0)), ("split_tail", (1, 0)), ("band_form", (1, 0)), ("host_ladder", (1, 0)), ("device_ladder", (1, 0)), ("device_ladder_fused", (0, 1)),
                      ("lean_kernel", (1, 0)), ("rows_in_backward", (1, 0)), ("row_products", (1, 0)), ("dense_graph", (1, 0)),
                      ("band_kernel", (1, 0)), ("band_kernel", (1, 2)), ("band_problems_per_group", (0, 8)), ("band_problems_per_group", (0, 32)), ("f1_tiles", (1, 0)), ("general_dense", (1, 2)),
                      ("dense_panel_blocks", (1, 0)), ("dense_panel_blocks", (1, 2)), ("lds_pad", (1, 0)), ("waves_per_block", (0, 2)), ("multipliers_early", (1, 0))):
            if rng.integers(4) == 0:
                ropt[k] = vs[1]
    tag = f"case {seed} fam {fam} n {s.nvar} m {s.nequ} p {s.ncon} B {B} kind {kind} posdef {posdef} {ropt}"
    if os.environ.get("FUZZ_VERBOSE"):
        print("RUN", tag, flush=True)
    try:
        L = hipldl.HIPLDLStruct(s.N, rows, cols, None, s.nvar, s.nequ, s.ncon, batch=B, options=hipldl.Options(plan_kind=kind, **{**ropt, **extra}))
        kk = L.config["kernel"] + ("+band" if L.config.get("band") else "") + ("+tail" if L.config["tail"] else "") + ("+split" if fam == 5 and L.info["order"].startswith("ndc2") and not L.config["tail"] else "")
        kinds[kk] = kinds.get(kk, 0) + 1
        if os.environ.get("FUZZ_VERBOSE"):
            print("   ", L.config["kernel"], L.info["order"], "tail" if L.config["tail"] else "", flush=True)
        v = vals.copy()
        d = np.full((B, s.N), 7.0)
        d, ok, rho, ro, nf = hipldl.newton_system_(d, s.nvar, s.nequ, s.ncon, rhs, v, L, ro_in if B > 1 else float(ro_in[0]), p)
        ok, rho, ro, nf = (np.atleast_1d(np.asarray(x)) for x in (ok, rho, ro, nf))
        d = np.asarray(d).reshape(B, s.N)
        orc = O.Oracle(s.N, rows, cols, O.canonical_perm(s.nvar, s.nequ, s.ncon))
        v0 = vals.copy()
        d0, ok0, rho0, ro0, nf0 = O.newton_system_batch(orc, B, s.nvar, s.nequ, s.ncon, rhs, v0, ro_in, po)
        bad = []
        skip = set()   # problems whose first pivot is below the rounding noise and was decided differently: no further comparison
        if not (np.array_equal(ok.astype(bool), np.atleast_1d(ok0).astype(bool)) and np.array_equal(nf, np.atleast_1d(nf0))
                and np.array_equal(rho, np.atleast_1d(rho0)) and np.array_equal(ro, np.atleast_1d(ro0))):
            # a first factorisation with a pivot below the rounding noise of the matrix has no determined inertia (DESIGN section 5,
            # test_near_singular_sweep_decisions): such a problem may legitimately enter the ladder on one side and not on the other
            ok0a, nf0a = np.atleast_1d(ok0), np.atleast_1d(nf0)
            differ = [b for b in range(B) if (bool(ok[b]), int(nf[b]), float(rho[b]), float(ro[b])) !=
                      (bool(ok0a[b]), int(nf0a[b]), float(np.atleast_1d(rho0)[b]), float(np.atleast_1d(ro0)[b]))]
            noise = []
            for b in differ:
                # the rung at which one side succeeded and the other went on: the smaller of the two final rho (0 = the first attempt)
                rho_t = min(float(rho[b]), float(np.atleast_1d(rho0)[b]))
                vv = vals[b].copy()
                if rho_t > 0.0:
                    vv[-s.nvar:] = rho_t
                orc.try_to_factorize(vv, s.nvar, s.nequ, s.ncon, po[0])
                Dab = np.abs(orc.D)
                #  the threshold follows the rounding noise a non-pivoting LDL^T of these matrices accumulates — a few
                # thousand eps of the largest pivot (entries are O(1); condensation, division and summation order differ from the
                # oracle's) — instead of a fixed 1e-11; every reclassified problem is printed, and the remaining checks still run for
                # the problems whose decisions agree
                if Dab.min() <= 4096.0 * np.finfo(float).eps * Dab.max():
                    noise.append(b)
                    print(f"   sub-noise: {tag} problem {b}: min|D| / max|D| = {Dab.min() / Dab.max():.2e} at rho = {rho_t:.3e}; "
                          f"hip (ok, nf, rho) = ({bool(ok[b])}, {int(nf[b])}, {float(rho[b]):.3e}), oracle ({bool(ok0a[b])}, {int(nf0a[b])}, {float(np.atleast_1d(rho0)[b]):.3e})", flush=True)
            if len(noise) == len(differ):
                subnoise += 1
                skip = set(noise)
            else:
                bad.append("decisions")
            if os.environ.get("FUZZ_VERBOSE"):
                print("    hip   ok", ok.astype(int), "nf", nf, "rho", rho, "ro", ro, flush=True)
                print("    orcl  ok", np.atleast_1d(ok0).astype(int), "nf", np.atleast_1d(nf0), "rho", np.atleast_1d(rho0), "ro", np.atleast_1d(ro0), flush=True)
                print("    rho_old in", ro_in, flush=True)
        keep = [b for b in range(B) if b not in skip]
        if not np.array_equal(v.reshape(B, -1)[keep, -s.nvar:], v0[keep, -s.nvar:], equal_nan=True):
            bad.append("rho slots")
        def berr(vv, rr, dd):   # normwise backward error of K d = -rhs with the values (rho slots included) the call left
            import scipy.sparse as sp
            Kl = sp.coo_matrix((vv, (rows - 1, cols - 1)), shape=(s.N, s.N)).tocsr()
            K = Kl + sp.tril(Kl, -1).T
            return np.abs(K @ dd + rr).max() / (abs(K).sum(axis=1).max() * np.abs(dd).max() + np.abs(rr).max())
        for b in keep:
            if np.atleast_1d(ok0)[b]:
                if not np.abs(d[b] - d0[b]).max() <= 1e-8 * max(1e-300, np.abs(d0[b]).max()):
                    # ill-conditioned systems (rank-deficient blocks held up by a small rho): the solution is only as good as the
                    # condition number allows on either side — what must hold is the backward error
                    be, be0 = berr(v0[b], rhs[b], d[b]), berr(v0[b], rhs[b], d0[b])
                    if not be <= max(1e-13, 10.0 * be0):   # (no pivoting on either side: element growth shows in both)
                        bad.append(f"d[{b}] rel {np.abs(d[b] - d0[b]).max() / np.abs(d0[b]).max():.2e} backward error {be:.1e} (oracle {be0:.1e})")
            elif not (d[b] == 7.0).all():
                bad.append(f"d[{b}] touched")
        # band handles: the same batch through the DEVICE-pointer entry with `vals` interleaved (cnl_options.batch_layout = 1) must give
        # the host-pointer results bit for bit (the same kernels on another address function; 32 problems per workgroup)
        if L.config.get("band") and B > 1:
            import torch
            dv = torch.device("cuda", 0)
            Li = hipldl.HIPLDLStruct(s.N, rows, cols, None, s.nvar, s.nequ, s.ncon, batch=B,
                                     options=hipldl.Options(plan_kind=hipldl.PLAN_THROUGHPUT, batch_layout=1, band_kernel=ropt.get("band_kernel", 1)))
            tv, trh = torch.from_numpy(vals).to(dv), torch.from_numpy(rhs).to(dv)
            tvi = torch.empty(hipldl.layout_len(Li, 0), dtype=torch.float64, device=dv)
            hipldl.interleave_dev(Li, 0, tv.data_ptr(), tvi.data_ptr(), 0)
            tdi = torch.full((B, s.N), 7.0, dtype=torch.float64, device=dv)
            tro, trho = torch.from_numpy(ro_in.copy()).to(dv), torch.zeros(B, dtype=torch.float64, device=dv)
            tnf, tok = torch.zeros(B, dtype=torch.int32, device=dv), torch.zeros(B, dtype=torch.int32, device=dv)
            hipldl.newton_system_dev(Li, tvi.data_ptr(), trh.data_ptr(), tdi.data_ptr(), tro.data_ptr(), trho.data_ptr(), tnf.data_ptr(), tok.data_ptr(), p, 0)
            hipldl.deinterleave_dev(Li, 0, tvi.data_ptr(), tv.data_ptr(), 0)
            torch.cuda.synchronize()
            same = (np.array_equal(tdi.cpu().numpy(), d) and np.array_equal(tok.cpu().numpy().astype(bool), ok.astype(bool)) and np.array_equal(tnf.cpu().numpy(), nf)
                    and np.array_equal(trho.cpu().numpy(), rho) and np.array_equal(tro.cpu().numpy(), ro) and np.array_equal(tv.cpu().numpy(), v.reshape(B, -1)))
            if not same:
                bad.append("interleaved layout differs from the problem-major call")
            kinds["+interleaved twin"] = kinds.get("+interleaved twin", 0) + 1
            Li.close()
        # the two-call sequence with the rho the ladder left
        okf = np.atleast_1d(hipldl.try_to_factorize(L, v, s.nvar, s.nequ, s.ncon, p[0]))
        if not np.array_equal(okf.astype(bool)[keep], np.atleast_1d(ok0).astype(bool)[keep]):
            bad.append("try_to_factorize with the final rho")
        elif okf.all() and not skip:
            d2 = np.zeros((B, s.N))
            hipldl.solve_ldl_(2.0 * rhs, L.factor, d2)
            for b in range(B):
                if not np.abs(d2[b] - 2.0 * d0[b]).max() <= 2e-8 * max(1e-300, np.abs(d0[b]).max()):
                    be, be0 = berr(v0[b], 2.0 * rhs[b], d2[b]), berr(v0[b], 2.0 * rhs[b], 2.0 * d0[b])
                    if not be <= max(1e-13, 10.0 * be0):
                        bad.append(f"solve d[{b}] backward error {be:.1e} (oracle {be0:.1e})")
        # the same handle again, other values (state left by the first round must not matter): problems swapped round, another rho_old
        if not bad and not skip and B > 1 and os.environ.get("FUZZ_WIDE"):
            permb = rng.permutation(B)
            vals2, rhs2, ro2 = vals[permb].copy(), rhs[permb] * 0.5, ro_in[permb[::-1]].copy()
            vb = vals2.copy()
            db = np.full((B, s.N), 7.0)
            db, okb, rhob, rob, nfb = hipldl.newton_system_(db, s.nvar, s.nequ, s.ncon, rhs2, vb, L, ro2, p)
            vb0 = vals2.copy()
            d0b, ok0b, rho0b, ro0b, nf0b = O.newton_system_batch(orc, B, s.nvar, s.nequ, s.ncon, rhs2, vb0, ro2, po)
            same = (np.array_equal(np.asarray(okb).astype(bool), ok0b.astype(bool)) and np.array_equal(np.asarray(nfb), nf0b)
                    and np.array_equal(np.asarray(rhob), rho0b) and np.array_equal(np.asarray(rob), ro0b))
            if not same:
                # (sub-noise first pivots: the same problems were seen in round one; a different rho_old changes the ladder, not the first attempt)
                first_differs = [b for b in range(B) if (int(np.asarray(nfb)[b]) == 1) != (int(nf0b[b]) == 1)]
                if not first_differs or any(np.abs(orc.D).min() > 1e-11 * np.abs(orc.D).max() for b in first_differs
                                            if orc.try_to_factorize(vals2[b].copy(), s.nvar, s.nequ, s.ncon, po[0]) in (True, False)):
                    bad.append("second round decisions")
            else:
                db = np.asarray(db).reshape(B, s.N)
                for b in range(B):
                    if ok0b[b] and not np.abs(db[b] - d0b[b]).max() <= 1e-8 * max(1e-300, np.abs(d0b[b]).max()):
                        be, be0 = berr(vb0[b], rhs2[b], db[b]), berr(vb0[b], rhs2[b], d0b[b])
                        if not be <= max(1e-13, 10.0 * be0):
                            bad.append(f"second round d[{b}]")
                    elif not ok0b[b] and not (db[b] == 7.0).all():
                        bad.append(f"second round d[{b}] touched")
        L.close()
        if bad:
            fails += 1
            print("FAIL", tag, bad[:4], flush=True)
    except Exception as e:  # noqa: BLE001
        fails += 1
        print("ERROR", tag, repr(e)[:300], flush=True)
print(f"{ncases} cases, {fails} failures, kernels {kinds}" + (f", {subnoise} case(s) with sub-noise pivots decided differently (not failures: a pivot below 4096 eps of the largest at the rung where the sides part; each is printed above)" if subnoise else ""), flush=True)
sys.exit(1 if fails else 0)
