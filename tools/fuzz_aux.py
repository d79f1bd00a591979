"""Randomised run of the device-resident helpers either side of the Newton system (rows f1 / f2 / f4; GPU box; not part of the
test-suite): random and band structures (with and without constraints, with and without a Hessian segment, batch sizes that are
not multiples of anything), against the oracle's restatements —
  residual_vectors  bit for bit (rhs and both infinity norms, NaN included)
  prepare_newton_system  bit for bit (sign bits included)
  trial_point  xt / rt bit for bit, the capped multiplier step to 4e-15
  cgls_multipliers  Jx'r bit for bit, the iteration count, lambda to 1e-8 of its largest component
usage: fuzz_aux.py [cases] [first seed]      one line per failure and a summary; exit code 1 on any failure"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cannoles_jl_amd  # noqa: F401,E402
from cannoles_jl_amd import hipldl, synthetic as syn  # noqa: E402
from oracle import oracle as O  # noqa: E402

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
dev = torch.device("cuda", 0)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
fails = 0
tiled = 0
for case in range(ncases):
    seed = seed0 + case
    rng = np.random.default_rng(200000 + seed)
    fam = int(rng.integers(4))
    if fam == 0:
        n = int(rng.integers(3, 90)); m = int(rng.integers(2, 2 * n + 2)); pc = int(rng.integers(0, min(6, n) + 1))
        s = syn.random_structure(n, m, pc, float(rng.uniform(0.05, 0.4)), seed, hess=bool(rng.integers(4)))
    elif fam == 1:
        pc = int(rng.integers(0, 5)); blocks = int(rng.integers(4, 70)); n = (pc if pc else 1) * blocks
        s = syn.band_structure(n, pc, hw=int(rng.integers(1, 5)))
    elif fam == 3:   # (round 5) several column tiles of row f1 (256 columns each), odd sizes
        pc = int(rng.integers(0, 6)); blocks = int(rng.integers(60, 700)); n = (pc if pc else 1) * blocks
        s = syn.band_structure(n, pc, hw=int(rng.integers(1, 5)))
    else:
        n = int(rng.integers(100, 600)); m = int(rng.integers(n // 2, n + 80)); pc = int(rng.integers(0, 9))
        s = syn.random_structure(n, m, pc, float(rng.uniform(0.005, 0.03)), seed)
    B = int(rng.choice([1, 2, 3, 5, 9, 31, 70]))
    rows, cols = s.kkt_pattern()
    tag = f"case {seed} fam {fam} n {s.nvar} m {s.nequ} p {s.ncon} B {B}"
    bad = []
    try:
        L = hipldl.HIPLDLStruct(s.N, rows, cols, None, s.nvar, s.nequ, s.ncon, batch=B)
        tiled += bool(L.config["f1_tiles"])
        vals = np.stack([syn.random_values(s, seed * 100 + b)[0] if s.name != "band" else syn.band_values(s, seed * 100 + b)[0] for b in range(B)])
        r = rng.standard_normal((B, s.nequ)); lam = rng.standard_normal((B, max(s.ncon, 1)))[:, :s.ncon]
        Fx = rng.standard_normal((B, s.nequ)); cx = rng.standard_normal((B, max(s.ncon, 1)))[:, :s.ncon]
        if rng.integers(3) == 0:
            r[int(rng.integers(B)), int(rng.integers(s.nequ))] = np.nan
        # ---- f1: residual vectors
        tv, tr, tl, tF, tc = t(vals), t(r), t(lam), t(Fx), t(cx)
        trhs = torch.zeros((B, s.N), dtype=torch.float64, device=dev)
        tn = torch.full((B, 2), -1.0, dtype=torch.float64, device=dev)
        hipldl.residual_vectors_dev(L, tv.data_ptr(), tr.data_ptr(), tl.data_ptr() if s.ncon else 0, tF.data_ptr(), tc.data_ptr() if s.ncon else 0,
                                    trhs.data_ptr(), tn.data_ptr(), 0)
        torch.cuda.synchronize()
        rhs, nrm = trhs.cpu().numpy(), tn.cpu().numpy()
        for b in range(min(B, 6)):
            rhs0, (nd0, np0) = O.residual_vectors(rows, cols, vals[b], s.nvar, s.nequ, s.ncon, r[b], lam[b], Fx[b], cx[b])
            if not (np.array_equal(rhs[b], rhs0, equal_nan=True) and np.array_equal(nrm[b], np.array([nd0, np0]), equal_nan=True)):
                bad.append(f"residual_vectors[{b}]")
        # ---- f1 with the Jacobian values read from the model's arrays (cnl_residual_vectors_jac_dev): bit-equal to the vals variant
        off_ = s.offsets()
        tJ_, tJc_ = t(vals[:, off_[2]:off_[3]]), t(vals[:, off_[3]:off_[4]])
        trhs_j = torch.zeros_like(trhs)
        tn_j = torch.full((B, 2), -1.0, dtype=torch.float64, device=dev)
        hipldl.residual_vectors_jac_dev(L, s.nnzjF, s.nnzjc, tJ_.data_ptr(), tJc_.data_ptr() if s.nnzjc else 0, tr.data_ptr(), tl.data_ptr() if s.ncon else 0,
                                        tF.data_ptr(), tc.data_ptr() if s.ncon else 0, trhs_j.data_ptr(), tn_j.data_ptr(), 0)
        torch.cuda.synchronize()
        if not (np.array_equal(trhs_j.cpu().numpy(), rhs, equal_nan=True) and np.array_equal(tn_j.cpu().numpy(), nrm, equal_nan=True)):
            bad.append("residual_vectors from the model's arrays")
        # ---- layout conversions (any handle): round trip and the index function of the header
        ti_ = torch.full((hipldl.layout_len(L, 0),), np.nan, dtype=torch.float64, device=dev)
        hipldl.interleave_dev(L, 0, tv.data_ptr(), ti_.data_ptr(), 0)
        tb_ = torch.zeros_like(tv)
        hipldl.deinterleave_dev(L, 0, ti_.data_ptr(), tb_.data_ptr(), 0)
        torch.cuda.synchronize()
        pp_, ee_ = np.meshgrid(np.arange(B), np.arange(s.nnzNS), indexing="ij")
        want_ = np.zeros(ti_.numel())
        want_[hipldl.il_index(pp_, ee_, s.nnzNS)] = vals
        if not (np.array_equal(ti_.cpu().numpy(), want_) and np.array_equal(tb_.cpu().numpy(), vals)):
            bad.append("interleave / deinterleave")
        # ---- f2: prepare_newton_system
        nhF, nhc, njF, njc = len(s.hF[0]), len(s.hc[0]), s.nnzjF, s.nnzjc
        gn = nhF == 0 or bool(rng.integers(4) == 0)
        hF, hc = rng.standard_normal((B, max(nhF, 1)))[:, :nhF], rng.standard_normal((B, max(nhc, 1)))[:, :nhc]
        if nhc:
            hc[:, 0] = 0.0
        Jx, Jcx = rng.standard_normal((B, njF)), rng.standard_normal((B, max(njc, 1)))[:, :njc]
        delta = rng.uniform(0.01, 1.0, B)
        vals0 = rng.standard_normal((B, s.nnzNS))
        off = s.offsets()
        vals0[:, off[4]:off[5]] = -1.0
        tv0 = t(vals0)
        thF, thc, tJx, tJc, tde = t(hF), t(hc), t(Jx), t(Jcx), t(delta)
        hipldl.prepare_newton_system_dev(L, 0 if (gn and nhF == 0) else nhF, nhc, njF, njc, 0 if gn else thF.data_ptr(), thc.data_ptr() if nhc else 0,
                                         tJx.data_ptr(), tJc.data_ptr() if njc else 0, tde.data_ptr(), tv0.data_ptr(), 0)
        torch.cuda.synchronize()
        got = tv0.cpu().numpy()
        for b in range(min(B, 6)):
            ref = vals0[b].copy()
            O.prepare(ref, s.nvar, s.nequ, s.ncon, nhF, nhc, njF, njc, None if gn else hF[b], hc[b], Jx[b], Jcx[b], delta[b])
            if not (np.array_equal(got[b], ref) and np.array_equal(np.signbit(got[b]), np.signbit(ref))):
                bad.append(f"prepare[{b}]")
        # ---- f2 writing `vals` interleaved (band handles with cnl_options.batch_layout = 1): bit-equal to the pass above after cnl_deinterleave_dev
        if s.name == "band" and B > 1:
            try:
                Li = hipldl.HIPLDLStruct(s.N, rows, cols, None, s.nvar, s.nequ, s.ncon, batch=B, options=hipldl.Options(plan_kind=hipldl.PLAN_THROUGHPUT, batch_layout=1))
            except hipldl.CnlError:
                Li = None   # (outside the band kernels' envelope: half-widths 3, 4)
            if Li is not None:
                tvi = torch.full((hipldl.layout_len(Li, 0),), 3.0, dtype=torch.float64, device=dev)
                hipldl.interleave_dev(Li, 0, t(vals0).data_ptr(), tvi.data_ptr(), 0)
                hipldl.prepare_newton_system_dev(Li, 0 if (gn and nhF == 0) else nhF, nhc, njF, njc, 0 if gn else thF.data_ptr(), thc.data_ptr() if nhc else 0,
                                                 tJx.data_ptr(), tJc.data_ptr() if njc else 0, tde.data_ptr(), tvi.data_ptr(), 0)
                tvb = torch.zeros_like(tv0)
                hipldl.deinterleave_dev(Li, 0, tvi.data_ptr(), tvb.data_ptr(), 0)
                torch.cuda.synchronize()
                gb = tvb.cpu().numpy()
                if not (np.array_equal(gb, got) and np.array_equal(np.signbit(gb), np.signbit(got))):
                    bad.append("prepare (interleaved)")
                ilcases = globals().get("ilcases", 0) + 1
                globals()["ilcases"] = ilcases
                Li.close()
        # ---- f1: trial point
        x = rng.standard_normal((B, s.nvar)); d = rng.standard_normal((B, s.N))
        if s.ncon:
            d[0, s.nvar + s.nequ:] *= 1e6
        tx, td = t(x), t(d)
        txt, trt, tlt, tdl = torch.zeros_like(tx), torch.zeros_like(tr), torch.zeros_like(tl), torch.zeros_like(tl)
        hipldl.trial_point_dev(L, tx.data_ptr(), tr.data_ptr(), tl.data_ptr() if s.ncon else 0, td.data_ptr(), 1e4, txt.data_ptr(), trt.data_ptr(),
                               tlt.data_ptr() if s.ncon else 0, tdl.data_ptr() if s.ncon else 0, 0)
        torch.cuda.synchronize()
        for b in range(min(B, 6)):
            xt0, rt0, lt0, dl0 = O.trial_point(s.nvar, s.nequ, s.ncon, x[b], r[b], lam[b], d[b], 1e4)
            if not (np.array_equal(txt[b].cpu().numpy(), xt0) and np.array_equal(trt[b].cpu().numpy(), rt0, equal_nan=True)):
                bad.append(f"trial_point xt/rt[{b}]")
            if s.ncon and not (np.allclose(tdl[b].cpu().numpy(), dl0, rtol=4e-15, atol=0) and np.allclose(tlt[b].cpu().numpy(), lt0, rtol=4e-15, atol=1e-300)):
                bad.append(f"trial_point lambda[{b}]")
        # ---- f4: CGLS multipliers
        if s.ncon:
            r2 = np.nan_to_num(r, nan=0.3)
            r2[B - 1] = 0.0
            tr2 = t(r2)
            tl2 = torch.zeros((B, s.ncon), dtype=torch.float64, device=dev)
            tj = torch.zeros((B, s.nvar), dtype=torch.float64, device=dev)
            ti = torch.zeros(B, dtype=torch.int32, device=dev)
            hipldl.cgls_multipliers_dev(L, tv.data_ptr(), tr2.data_ptr(), tl2.data_ptr(), tj.data_ptr(), iters_ptr=ti.data_ptr())
            torch.cuda.synchronize()
            lam2, jxtr, its = tl2.cpu().numpy(), tj.cpu().numpy(), ti.cpu().numpy()
            tl3, tj3, ti3 = torch.zeros_like(tl2), torch.zeros_like(tj), torch.zeros_like(ti)
            hipldl.cgls_multipliers_jac_dev(L, s.nnzjF, s.nnzjc, tJ_.data_ptr(), tJc_.data_ptr(), tr2.data_ptr(), tl3.data_ptr(), tj3.data_ptr(), iters_ptr=ti3.data_ptr())
            torch.cuda.synchronize()
            if not (torch.equal(tl3, tl2) and torch.equal(tj3, tj) and torch.equal(ti3, ti)):
                bad.append("cgls from the model's arrays")
            for b in list(range(min(B, 4))) + [B - 1]:
                lam0, jx0, it0 = O.cgls_multipliers(rows, cols, vals[b], s.nvar, s.nequ, s.ncon, r2[b])
                if not np.array_equal(jxtr[b], jx0):
                    bad.append(f"cgls Jxtr[{b}]")
                if its[b] != it0:
                    bad.append(f"cgls iterations[{b}] {its[b]} != {it0}")
                elif not np.abs(lam2[b] - lam0).max() <= 1e-8 * max(1e-300, np.abs(lam0).max()):   # (CGLS on the normal equations: cond^2 amplifies the rounding of a different summation order)
                    bad.append(f"cgls lambda[{b}]")
        L.close()
    except Exception as e:  # noqa: BLE001
        bad.append("ERROR " + repr(e)[:300])
    if bad:
        fails += 1
        print("FAIL", tag, bad[:5], flush=True)
print(f"{ncases} cases ({tiled} with row f1 on column tiles, {globals().get('ilcases', 0)} with row f2 writing the interleaved layout), {fails} failures", flush=True)
sys.exit(1 if fails else 0)
