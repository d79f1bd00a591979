"""Parity of the HIP path (through the C ABI) against the CPU oracle.  -m gpu.

Tolerances (fp64, stated per SURVEY.md §7 "Hard parts"):
  * forward:  max|d - d_oracle| / max|d_oracle| <= 1e-9 on the well-conditioned configs
  * backward: max|K d + rhs| / (||K||_inf max|d| + max|rhs|) <= 1e-13
  * (success, nfact, rho, rho_old) identical to the oracle's.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FWD_TOL = 1e-9
BWD_TOL = 1e-13


def _mods():
    import cannoles_jl_amd  # noqa: F401
    from cannoles_jl_amd import hipldl, synthetic as syn
    from oracle import oracle as O
    return hipldl, syn, O


def backward_error(s, vals, rhs, d):
    import scipy.sparse as sp
    rows, cols = s.kkt_pattern()
    Kl = sp.coo_matrix((vals, (rows - 1, cols - 1)), shape=(s.N, s.N)).tocsr()
    K = Kl + sp.tril(Kl, -1).T
    res = K @ d + rhs
    return np.abs(res).max() / (abs(K).sum(axis=1).max() * np.abs(d).max() + np.abs(rhs).max())


def run_case(s, vals, rhs, rho_old=None, fwd_tol=FWD_TOL, check_fwd=True, options=None, own_order=True):
    hipldl, syn, O = _mods()
    B = vals.shape[0]
    rows, cols = s.kkt_pattern()
    # each side runs with ITS OWN defaults (cnl_default_params / cnlo_default_params, both restating src/CaNNOLeS.jl:48-62):
    # a wrong default on either side shows up as a different ladder
    p = hipldl.default_params()
    p_oracle = O.default_params()
    ro = np.zeros(B) if rho_old is None else np.asarray(rho_old, float)
    LDLT = hipldl.HIPLDLStruct(s.N, rows, cols, None, s.nvar, s.nequ, s.ncon, batch=B, options=options)
    v = vals.copy()
    d = np.zeros((B, s.N))
    d, ok, rho, ro_out, nfact = hipldl.newton_system_(d, s.nvar, s.nequ, s.ncon, rhs, v, LDLT, ro, p)
    if B == 1:
        ok, rho, ro_out, nfact = np.array([ok]), np.array([rho]), np.array([ro_out]), np.array([nfact])
    perm = LDLT.plan_array("perm").astype(np.int64)
    orc = O.Oracle(s.N, rows, cols, perm)
    v0 = vals.copy()
    d0, ok0, rho0, ro0, nf0 = O.newton_system_batch(orc, B, s.nvar, s.nequ, s.ncon, rhs, v0, ro, p_oracle)
    assert np.array_equal(ok, ok0)
    assert np.array_equal(nfact, nf0)
    assert np.array_equal(rho, rho0)
    assert np.array_equal(ro_out, ro0)
    # the rho slots of vals are left as the reference leaves them
    assert np.array_equal(v.reshape(B, -1)[:, -s.nvar:], v0[:, -s.nvar:])
    d = d.reshape(B, s.N)
    for b in range(B):
        if not ok0[b]:
            continue
        vv = v0[b]
        assert backward_error(s, vv, rhs[b], d[b]) <= BWD_TOL
        if check_fwd:
            assert np.abs(d[b] - d0[b]).max() / np.abs(d0[b]).max() <= fwd_tol
    # ... and against the oracle on an elimination order the product had no part in (r-nodes, x in natural order, multipliers:
    # oracle.canonical_perm, the order SURVEY 8d prices): the reference's own order (AMD) is a third one, and decisions that
    # hold on two unrelated orders are not an artefact of handing the oracle the product's permutation
    if own_order:
        orc2 = O.Oracle(s.N, rows, cols, O.canonical_perm(s.nvar, s.nequ, s.ncon))
        v2 = vals.copy()
        d2, ok2, rho2, ro2, nf2 = O.newton_system_batch(orc2, B, s.nvar, s.nequ, s.ncon, rhs, v2, ro, p_oracle)
        assert np.array_equal(ok, ok2) and np.array_equal(nfact, nf2) and np.array_equal(rho, rho2) and np.array_equal(ro_out, ro2)
        if check_fwd:
            for b in range(B):
                if ok2[b]:
                    assert np.abs(d[b] - d2[b]).max() / np.abs(d2[b]).max() <= fwd_tol
    info, cfg = LDLT.info, LDLT.config
    LDLT.close()
    return info, cfg


def test_fixture_mgh01con(built):
    """F1: first Newton system of the reference's MGH01CON model (test/mgh01con.jl), exact answer known."""
    hipldl, syn, O = _mods()
    rows, cols, _ = O.kkt_pattern(2, 2, 1, hF=([1], [1]), hc=([1, 2, 2], [1, 1, 2]), jF=([1, 2, 2], [1, 1, 2]), jc=([1], [1]))
    vals = np.array([88., -0., -0., -0., -1., 24., 10., 1., -1., -1., -0.1, 0., 0.])
    rhs = np.array([0, -44, 0, 0, -1.2])
    LDLT = hipldl.HIPLDLStruct(5, rows, cols, vals, 2, 2, 1)
    assert hipldl.get_vals(LDLT) is vals
    ok, npos, nzer = hipldl.try_to_factorize(LDLT, vals, 2, 2, 1, 2.220446049250313e-16, return_inertia=True)
    assert ok and npos == 2 and nzer == 0
    d = np.zeros(5)
    assert hipldl.solve_ldl_(rhs, LDLT.factor, d) is True
    exact = np.array([-52 / 55, 149 / 55, 52 / 55, 4.4, -236 / 11])
    assert np.abs(d - exact).max() <= 1e-13 * np.abs(exact).max()
    d2 = np.zeros(5)
    d2, ok, rho, rho_old, nfact = hipldl.newton_system_(d2, 2, 2, 1, rhs, vals, LDLT, 0.0, hipldl.default_params())
    assert ok and rho == 0.0 and rho_old == 0.0 and nfact == 1
    assert np.abs(d2 - exact).max() <= 1e-13 * np.abs(exact).max()
    LDLT.close()


@pytest.mark.parametrize("seed", range(6))
def test_random_structures(built, seed):
    hipldl, syn, O = _mods()
    s = syn.random_structure(30 + 5 * seed, 40, 5 if seed % 2 else 0, 0.12, seed)
    vals, rhs = syn.batch_values(s, 3, cfg=seed, gen=syn.random_values)
    run_case(s, vals, rhs)


def test_gauss_newton_no_hessian(built):
    """method=:Newton_noFHess: empty H_F segment (hessian_approx.jl:5-9); underdetermined => rho > 0"""
    hipldl, syn, O = _mods()
    s = syn.random_structure(30, 20, 0, 0.15, 7, hess=False)
    vals, rhs = syn.batch_values(s, 2, cfg=7, gen=syn.random_values)
    run_case(s, vals, rhs, fwd_tol=1e-6)


def test_cfg4_batch(built):
    hipldl, syn, O = _mods()
    s = syn.band_structure(1000, 10)
    vals, rhs = syn.batch_values(s, 64, cfg=4)
    run_case(s, vals, rhs)


def test_cfg5_rho_ladder(built):
    """F3: failures at rho in {0, 6.06e-6, 6.06e-4, 6.06e-2, 6.06}, success at 605.5 => nfact = 6"""
    hipldl, syn, O = _mods()
    s = syn.band_structure(1000, 10)
    vals, rhs = syn.batch_values(s, 8, cfg=5, stress="ladder")
    # mix in healthy problems: the ladder is per problem
    v2, r2 = syn.batch_values(s, 8, cfg=4)
    vals[::2], rhs[::2] = v2[::2], r2[::2]
    run_case(s, vals, rhs)
    run_case(s, vals, rhs, rho_old=np.full(8, 10.0))


def test_cfg5_illconditioned(built):
    hipldl, syn, O = _mods()
    s = syn.band_structure(1000, 10)
    vals, rhs = syn.batch_values(s, 4, cfg=5, stress="illcond")
    run_case(s, vals, rhs, check_fwd=False)


def test_cfg3_single(built):
    hipldl, syn, O = _mods()
    s = syn.band_structure(10000, 50)
    vals, rhs = syn.batch_values(s, 1, cfg=3)
    run_case(s, vals, rhs)


def test_cfg2_dense_small(built):
    hipldl, syn, O = _mods()
    s = syn.dense_structure(60, 120)
    vals, rhs = syn.batch_values(s, 2, cfg=2, gen=syn.dense_values)
    run_case(s, vals, rhs)


def test_factorize_then_solve_batch(built):
    """the two-call plugin path (try_to_factorize; solve_ldl!) on a batch, several right-hand sides"""
    hipldl, syn, O = _mods()
    s = syn.band_structure(400, 8)
    B = 5
    vals, rhs = syn.batch_values(s, B, cfg=4)
    rows, cols = s.kkt_pattern()
    LDLT = hipldl.HIPLDLStruct(s.N, rows, cols, vals, s.nvar, s.nequ, s.ncon, batch=B)
    ok = hipldl.try_to_factorize(LDLT, vals, s.nvar, s.nequ, s.ncon, 2.220446049250313e-16)
    assert ok.all()
    for k in range(2):
        r = rhs * (k + 1)
        d = np.zeros((B, s.N))
        hipldl.solve_ldl_(r, LDLT.factor, d)
        for b in range(B):
            assert backward_error(s, vals[b], r[b], d[b]) <= BWD_TOL
    LDLT.close()


def test_errors(built):
    hipldl, syn, O = _mods()
    s = syn.band_structure(50, 2)
    rows, cols = s.kkt_pattern()
    LDLT = hipldl.HIPLDLStruct(s.N, rows, cols, None, s.nvar, s.nequ, s.ncon)
    with pytest.raises(hipldl.CnlError):
        hipldl.solve_ldl_(np.zeros(s.N), LDLT.factor, np.zeros(s.N))  # solve before factorize
    LDLT.close()
    with pytest.raises(hipldl.CnlError):
        hipldl.HIPLDLStruct(s.N, cols, rows, None, s.nvar, s.nequ, s.ncon)  # upper triangle


def test_reference_problems_through_outer_loop(built):
    """the reference's end-to-end known answers (test/runtests.jl:56-100) with the HIP path as linsolve"""
    hipldl, syn, O = _mods()
    from tests.support.outer_loop import SymNLS, solve
    from tests.test_oracle_pinning import CONSTRAINED, UNCONSTRAINED

    def hip_solver(N, rows, cols, vals, nvar, nequ, ncon):
        return hipldl.HIPLDLStruct(N, rows, cols, vals, nvar, nequ, ncon)

    def hip_newton(LDLT, nvar, nequ, ncon, rhs, vals, rho_old, params):
        d = np.zeros(LDLT.N)
        return hipldl.newton_system_(d, nvar, nequ, ncon, np.ascontiguousarray(rhs), vals, LDLT, rho_old, params)

    p = hipldl.default_params()
    for F, x0, xf in UNCONSTRAINED[:4]:
        out = solve(SymNLS(F, x0), hip_solver, hip_newton, p)
        assert np.allclose(out["solution"], xf, atol=1e-4)
    for F, c, x0, xf in CONSTRAINED:
        out = solve(SymNLS(F, x0, c), hip_solver, hip_newton, p)
        assert np.allclose(out["solution"], xf, atol=1e-4)


def test_cfg3_batch_properties_full_size(built):
    """BASELINE config 3 at full size, batch of 32: size-independent properties —
    backward error of every system and linearity of the solve in the right-hand side."""
    hipldl, syn, O = _mods()
    s = syn.band_structure(10000, 50)
    B = 32
    vals, rhs = syn.batch_values(s, B, cfg=3)
    rows, cols = s.kkt_pattern()
    p = hipldl.default_params()
    LDLT = hipldl.HIPLDLStruct(s.N, rows, cols, None, s.nvar, s.nequ, s.ncon, batch=B)
    d1 = np.zeros((B, s.N))
    d1, ok, rho, ro, nf = hipldl.newton_system_(d1, s.nvar, s.nequ, s.ncon, rhs, vals.copy(), LDLT, np.zeros(B), p)
    assert ok.all() and (nf == 1).all() and (rho == 0).all()
    for b in range(B):
        assert backward_error(s, vals[b], rhs[b], d1[b]) <= BWD_TOL
    d2 = np.zeros((B, s.N))
    d2, ok2, *_ = hipldl.newton_system_(d2, s.nvar, s.nequ, s.ncon, -2.0 * rhs, vals.copy(), LDLT, np.zeros(B), p)
    assert ok2.all()
    assert np.abs(d2 + 2.0 * d1).max() <= 1e-12 * np.abs(d1).max()
    # the two-call path gives the same solution as the fused one
    okf = hipldl.try_to_factorize(LDLT, vals, s.nvar, s.nequ, s.ncon, p[0])
    assert okf.all()
    d3 = np.zeros((B, s.N))
    hipldl.solve_ldl_(rhs, LDLT.factor, d3)
    assert np.abs(d3 - d1).max() <= 1e-10 * np.abs(d1).max()
    LDLT.close()


# ---- residual / optimality vectors and trial point, device-resident -------------------------
def _f1_inputs(s, B, seed):
    hipldl, syn, O = _mods()
    rng = np.random.default_rng(seed)
    if s.name == "band":
        vals, _ = syn.batch_values(s, B, cfg=4)
    else:
        vals = np.stack([syn.random_values(s, seed * 100 + b)[0] for b in range(B)])
    r = rng.standard_normal((B, s.nequ))
    lam = rng.standard_normal((B, max(s.ncon, 1)))[:, :s.ncon]
    Fx = rng.standard_normal((B, s.nequ))
    cx = rng.standard_normal((B, max(s.ncon, 1)))[:, :s.ncon]
    return np.ascontiguousarray(vals), r, np.ascontiguousarray(lam), Fx, np.ascontiguousarray(cx)


@pytest.mark.parametrize("tiles", [1, 0])
@pytest.mark.parametrize("kind", ["band", "band_odd", "cfg3_slice", "random", "nocon", "scattered"])
def test_f1_residual_vectors_bit_exact(built, kind, tiles):
    """rhs = [Jx'r - Jc'λ; F - r; c] and the two infinity norms: per-column COO-order sums, bit-identical to the
    restated COO mul! (src/CaNNOLeS.jl:507-508,519-524,528-529,631-632).  Both kernels of the row: the column tiles streamed
    through LDS (round 5; band and block patterns) and the gather kernel (`f1_tiles = 0`, and every pattern whose tiles' slot
    ranges exceed the windows)."""
    import torch
    hipldl, syn, O = _mods()
    B = 6
    if kind == "band":
        s = syn.band_structure(1000, 10)
    elif kind == "band_odd":
        s, B = syn.band_structure(777, 7), 67       # odd sizes: every alignment of the 16-byte chunks; two problems per workgroup + a rest
    elif kind == "cfg3_slice":
        s, B = syn.band_structure(3000, 15), 3
    elif kind == "random":
        s = syn.random_structure(40, 55, 7, 0.15, seed=3)
    elif kind == "scattered":
        s = syn.random_structure(700, 900, 3, 0.01, seed=9)   # a tile's entries lie anywhere in vals: gather kernel whatever the option says
    else:
        s = syn.random_structure(30, 45, 0, 0.15, seed=5)
    vals, r, lam, Fx, cx = _f1_inputs(s, B, 11)
    if kind in ("random", "band_odd"):
        r[2, 3] = np.nan  # NaN must reach dual and its norm, as in norm(., Inf)
    rows, cols = s.kkt_pattern()
    LDLT = hipldl.HIPLDLStruct(s.N, rows, cols, None, s.nvar, s.nequ, s.ncon, batch=B, options=hipldl.Options(f1_tiles=tiles))
    assert LDLT.config["f1_tiles"] == (bool(tiles) and kind != "scattered")
    dev = torch.device("cuda", 0)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    tv, tr, tl, tF, tc = t(vals), t(r), t(lam), t(Fx), t(cx)
    trhs = torch.zeros((B, s.N), dtype=torch.float64, device=dev)
    tn = torch.full((B, 2), -1.0, dtype=torch.float64, device=dev)
    hipldl.residual_vectors_dev(LDLT, tv.data_ptr(), tr.data_ptr(), tl.data_ptr() if s.ncon else 0, tF.data_ptr(),
                                tc.data_ptr() if s.ncon else 0, trhs.data_ptr(), tn.data_ptr(), 0)
    torch.cuda.synchronize()
    rhs, nrm = trhs.cpu().numpy(), tn.cpu().numpy()
    for b in range(B):
        rhs0, (nd0, np0) = O.residual_vectors(rows, cols, vals[b], s.nvar, s.nequ, s.ncon, r[b], lam[b], Fx[b], cx[b])
        assert np.array_equal(rhs[b], rhs0, equal_nan=True)
        assert np.array_equal(nrm[b], np.array([nd0, np0]), equal_nan=True)
    # the same row reading the Jacobian values from the model's arrays (cnl_residual_vectors_jac_dev): bit-equal
    off = s.offsets()
    tJ, tJc = t(vals[:, off[2]:off[3]]), t(vals[:, off[3]:off[4]])
    trhs2 = torch.zeros((B, s.N), dtype=torch.float64, device=dev)
    tn2 = torch.full((B, 2), -1.0, dtype=torch.float64, device=dev)
    hipldl.residual_vectors_jac_dev(LDLT, s.nnzjF, s.nnzjc, tJ.data_ptr(), tJc.data_ptr() if s.ncon else 0, tr.data_ptr(), tl.data_ptr() if s.ncon else 0,
                                    tF.data_ptr(), tc.data_ptr() if s.ncon else 0, trhs2.data_ptr(), tn2.data_ptr(), 0)
    torch.cuda.synchronize()
    assert np.array_equal(trhs2.cpu().numpy(), rhs, equal_nan=True) and np.array_equal(tn2.cpu().numpy(), nrm, equal_nan=True)
    with pytest.raises(hipldl.CnlError):
        hipldl.residual_vectors_jac_dev(LDLT, s.nnzjF + 1, s.nnzjc, tJ.data_ptr(), tJc.data_ptr() if s.ncon else 0, tr.data_ptr(), tl.data_ptr() if s.ncon else 0,
                                        tF.data_ptr(), tc.data_ptr() if s.ncon else 0, trhs2.data_ptr(), tn2.data_ptr(), 0)
    LDLT.close()


def test_f1_trial_point(built):
    """xt, rt, λt, dλ of the extrapolation step (src/CaNNOLeS.jl:654,661-668), with and without the 1e4 cap on ‖dλ‖."""
    import torch
    hipldl, syn, O = _mods()
    s = syn.band_structure(1000, 10)
    B = 4
    rng = np.random.default_rng(5)
    rows, cols = s.kkt_pattern()
    LDLT = hipldl.HIPLDLStruct(s.N, rows, cols, None, s.nvar, s.nequ, s.ncon, batch=B)
    x, r, lam = rng.standard_normal((B, s.nvar)), rng.standard_normal((B, s.nequ)), rng.standard_normal((B, s.ncon))
    d = rng.standard_normal((B, s.N))
    d[1, s.nvar + s.nequ:] *= 1e5  # cap active for problem 1
    d[3, s.nvar + s.nequ:] = 0.0
    dev = torch.device("cuda", 0)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    tx, tr, tl, td = t(x), t(r), t(lam), t(d)
    txt, trt, tlt, tdl = torch.zeros_like(tx), torch.zeros_like(tr), torch.zeros_like(tl), torch.zeros_like(tl)
    hipldl.trial_point_dev(LDLT, tx.data_ptr(), tr.data_ptr(), tl.data_ptr(), td.data_ptr(), 1e4, txt.data_ptr(), trt.data_ptr(),
                           tlt.data_ptr(), tdl.data_ptr(), 0)
    torch.cuda.synchronize()
    for b in range(B):
        xt0, rt0, lt0, dl0 = O.trial_point(s.nvar, s.nequ, s.ncon, x[b], r[b], lam[b], d[b], 1e4)
        assert np.array_equal(txt[b].cpu().numpy(), xt0)
        assert np.array_equal(trt[b].cpu().numpy(), rt0)
        # the 2-norm is reduced in a different order: a few ulp on the scaled entries
        np.testing.assert_allclose(tdl[b].cpu().numpy(), dl0, rtol=4e-15, atol=0)
        np.testing.assert_allclose(tlt[b].cpu().numpy(), lt0, rtol=4e-15, atol=1e-300)
    assert np.linalg.norm(tdl[1].cpu().numpy()) <= 1e4 * (1 + 1e-14)


def test_aux_kernels_beyond_the_grid_limit(built):
    """prepare_newton_system! and the trial point for MORE problems than a grid dimension holds (65 535): ONE linear grid dimension
    (problem, chunk) — refused with an error, not truncated, when the product leaves 31 bits —; row f1 for more problems than its
    second grid dimension holds (65 535 groups of four: 262 140 problems): launched in slices.  First, last and the problems either
    side of each limit against the oracle."""
    import torch
    hipldl, syn, O = _mods()
    s = syn.band_structure(24, 2)
    B = 263000
    rows, cols = s.kkt_pattern()
    nnz = len(rows)
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev); g.manual_seed(3)
    mk = lambda n: torch.randn((B, n), dtype=torch.float64, device=dev, generator=g)
    nhF, nhc, njF, njc = len(s.hF[0]), len(s.hc[0]), len(s.jF[0]), len(s.jc[0])
    hF, hc, Jx, Jc, de = mk(nhF), mk(nhc), mk(njF), mk(njc), mk(1).abs().reshape(B).contiguous()
    vals = torch.full((B, nnz), 7.0, dtype=torch.float64, device=dev)
    x, r, lam, d = mk(s.nvar), mk(s.nequ), mk(s.ncon), mk(s.N)
    Fx, cx = mk(s.nequ), mk(s.ncon)
    c = lambda a, b: a[b].cpu().numpy()
    probe = (0, 65534, 65535, 65536, 262139, 262140, 262141, B - 1)
    for tiles in (1, 0):
        L = hipldl.HIPLDLStruct(s.N, rows, cols, None, s.nvar, s.nequ, s.ncon, batch=B, options=hipldl.Options(band_kernel=0, f1_tiles=tiles))
        assert tiles or not L.config["f1_tiles"]
        if tiles:
            hipldl.prepare_newton_system_dev(L, nhF, nhc, njF, njc, hF.data_ptr(), hc.data_ptr(), Jx.data_ptr(), Jc.data_ptr(), de.data_ptr(), vals.data_ptr(), 0)
            xt, rt, lt, dl = torch.zeros_like(x), torch.zeros_like(r), torch.zeros_like(lam), torch.zeros_like(lam)
            hipldl.trial_point_dev(L, x.data_ptr(), r.data_ptr(), lam.data_ptr(), d.data_ptr(), 1e4, xt.data_ptr(), rt.data_ptr(), lt.data_ptr(), dl.data_ptr(), 0)
            torch.cuda.synchronize()
            for b in probe:
                v0 = np.full(nnz, 7.0)
                O.prepare(v0, s.nvar, s.nequ, s.ncon, nhF, nhc, njF, njc, c(hF, b), c(hc, b), c(Jx, b), c(Jc, b), float(de[b]))
                assert np.array_equal(c(vals, b), v0), b
                xt0, rt0, lt0, dl0 = O.trial_point(s.nvar, s.nequ, s.ncon, c(x, b), c(r, b), c(lam, b), c(d, b), 1e4)
                assert np.array_equal(c(xt, b), xt0) and np.array_equal(c(rt, b), rt0), b
                np.testing.assert_allclose(c(dl, b), dl0, rtol=4e-15, atol=0)
                np.testing.assert_allclose(c(lt, b), lt0, rtol=4e-15, atol=1e-300)
        rhs = torch.zeros((B, s.N), dtype=torch.float64, device=dev)
        nrm = torch.full((B, 2), -1.0, dtype=torch.float64, device=dev)
        hipldl.residual_vectors_dev(L, vals.data_ptr(), r.data_ptr(), lam.data_ptr(), Fx.data_ptr(), cx.data_ptr(), rhs.data_ptr(), nrm.data_ptr(), 0)
        torch.cuda.synchronize()
        for b in probe:
            rhs0, (nd0, np0) = O.residual_vectors(rows, cols, c(vals, b), s.nvar, s.nequ, s.ncon, c(r, b), c(lam, b), c(Fx, b), c(cx, b))
            assert np.array_equal(c(rhs, b), rhs0), (tiles, b)
            assert np.array_equal(c(nrm, b), np.array([nd0, np0])), (tiles, b)
        L.close()


@pytest.mark.parametrize("hw", [1, 3, 4])
def test_band_halfwidths_general_residual_pivots(built, hw):
    """Band families with other Jacobian half-widths: longer raw-value / product lists (several product rounds, raw
    lists up to the 128-entry limit, records longer than the three prefetch registers) and residual-block pivots
    d_r that are not -1 (the reference always passes -1, src/CaNNOLeS.jl:306, but the ABI takes `vals` as given)."""
    hipldl, syn, O = _mods()
    s = syn.band_structure(600, 6, hw=hw)
    vals, rhs = syn.batch_values(s, 5, cfg=7 + hw)
    off = s.offsets()
    rng = np.random.default_rng(hw)
    vals[:, off[4]:off[5]] = -rng.uniform(0.5, 2.0, (5, s.nequ))
    run_case(s, vals, rhs)


@pytest.mark.parametrize("gn", [False, True])
def test_prepare_newton_system_dev_bit_exact(built, gn):
    """Device twin of prepare_newton_system! (src/CaNNOLeS.jl:947-981) against the oracle's restatement: pure copies and
    sign flips, so bit-exact (including -0.0 in the H_c segment)."""
    import torch
    hipldl, syn, O = _mods()
    s = syn.band_structure(400, 4)
    B = 5
    rows, cols = s.kkt_pattern()
    LDLT = hipldl.HIPLDLStruct(s.N, rows, cols, None, s.nvar, s.nequ, s.ncon, batch=B)
    rng = np.random.default_rng(3)
    nhF, nhc, njF, njc = len(s.hF[0]), len(s.hc[0]), s.nnzjF, s.nnzjc
    hF, hc = rng.standard_normal((B, nhF)), rng.standard_normal((B, nhc))
    hc[:, 0] = 0.0  # -> -0.0
    Jx, Jcx, delta = rng.standard_normal((B, njF)), rng.standard_normal((B, njc)), rng.uniform(0.01, 1.0, B)
    vals0 = np.ones((B, s.nnzNS))
    off = s.offsets()
    vals0[:, off[4]:off[5]] = -1.0
    dev = torch.device("cuda", 0)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    thF, thc, tJx, tJc, tde, tv = t(hF), t(hc), t(Jx), t(Jcx), t(delta), t(vals0)
    hipldl.prepare_newton_system_dev(LDLT, nhF, nhc, njF, njc, 0 if gn else thF.data_ptr(), thc.data_ptr(), tJx.data_ptr(),
                                     tJc.data_ptr(), tde.data_ptr(), tv.data_ptr(), 0)
    torch.cuda.synchronize()
    got = tv.cpu().numpy()
    for b in range(B):
        ref = vals0[b].copy()
        O.prepare(ref, s.nvar, s.nequ, s.ncon, nhF, nhc, njF, njc, None if gn else hF[b], hc[b], Jx[b], Jcx[b], delta[b])
        assert np.array_equal(got[b], ref)
        assert np.array_equal(np.signbit(got[b]), np.signbit(ref))
    with pytest.raises(hipldl.CnlError):
        hipldl.prepare_newton_system_dev(LDLT, nhF + 1, nhc, njF, njc, 0, thc.data_ptr(), tJx.data_ptr(), tJc.data_ptr(), tde.data_ptr(),
                                         tv.data_ptr(), 0)


def test_f3_batched_outer_loop_on_the_hip_path(built):
    """B restated outer loops (src/CaNNOLeS.jl:418-864) over ONE batched handle — every round of
    Newton systems is one cnl_newton_system call with batch = B — reach the reference's known answers
    (test/runtests.jl:82-99) and agree with single-problem runs on the CPU oracle."""
    hipldl, syn, O = _mods()
    from cannoles_jl_amd import batch_solve
    from tests.support.outer_loop import SymNLS, solve
    from tests.test_oracle_pinning import CONSTRAINED, oracle_newton, oracle_solver
    (F0, c0, x00, xf0), (F1, c1, x01, xf1) = CONSTRAINED[0], CONSTRAINED[1]
    cases = [(F0, c0, x00, xf0), (F1, c1, x01, xf1), (F1, c1, [-1.0, 0.5], xf1), (F0, c0, [3.0, -2.0], xf0), (F1, c1, [0.0, 0.0], xf1),
             (F1, c1, [2.0, 2.0], xf1)]
    p = hipldl.default_params()
    res, ncalls = batch_solve.solve_batch([SymNLS(F, x0, c) for F, c, x0, xf in cases], p)
    for k, (F, c, x0, xf) in enumerate(cases):
        one = solve(SymNLS(F, x0, c), oracle_solver, oracle_newton, p)
        assert res[k]["status"] == one["status"] and res[k]["status"] in ("first_order", "small_residual")
        assert np.allclose(res[k]["solution"], xf, atol=1e-4)
        assert np.allclose(res[k]["solution"], one["solution"], atol=1e-6)
    assert ncalls < sum(r["nlinsolve"] for r in res)


def test_multipliers_last_large_fronts(built):
    """With the multipliers ordered last (cnl_options.multipliers_early = 0: the order the cost model used to pick) the top of the tree holds
    fronts of order 17..64, which take the out-of-line classes of the register-front kernel (global staging, two-word
    products, 32- and 64-lane elimination and backward substitution)."""
    hipldl, syn, O = _mods()
    late = hipldl.Options(multipliers_early=0, plan_kind=hipldl.PLAN_THROUGHPUT)
    s = syn.band_structure(2000, 40)
    rows, cols = s.kkt_pattern()
    plan = hipldl.Plan(s.N, rows, cols, s.nvar, s.nequ, s.ncon, options=late)
    assert plan.info["v2"] is not None and plan.info["v2"]["fronts32"] + plan.info["v2"]["fronts64"] > 0
    vals, rhs = syn.batch_values(s, 6, cfg=9)
    info, _ = run_case(s, vals, rhs, options=late)
    assert info["v2"]["fronts32"] + info["v2"]["fronts64"] > 0
    v2, r2 = syn.batch_values(s, 5, cfg=5, stress="ladder")
    run_case(s, v2, r2, check_fwd=False, options=late)


@pytest.mark.parametrize("plan_kind", ["throughput", "latency"])
@pytest.mark.parametrize("shape", [(400, 8, 2), (600, 6, 1), (600, 6, 4), (500, 0, 2), (1000, 10, 2)])
def test_solve_ldl_on_the_register_front_kernel(built, shape, plan_kind):
    """try_to_factorize then solve_ldl! (src/solver_types.jl:69-98), the reference's literal call sequence, with several
    right-hand sides per factorisation.  "throughput": the large-batch analysis, whose fronts are all of the fast class:
    cnl_solve then runs the forward substitution with the stored factor on the register-front kernel.  "latency": the
    small-batch analysis (bushy order, tasks), whatever kernels serve it.  Checked against the oracle's solve_ldl!."""
    hipldl, syn, O = _mods()
    opts = hipldl.Options(plan_kind=hipldl.PLAN_THROUGHPUT) if plan_kind == "throughput" else None
    n, p, hw = shape
    s = syn.band_structure(n, p, hw=hw)
    B = 6
    vals, rhs = syn.batch_values(s, B, cfg=3)
    off = s.offsets()
    vals[:, off[4]:off[5]] = -np.random.default_rng(hw).uniform(0.5, 2.0, (B, s.nequ))
    vals[:, off[6]:off[7]] = 0.125
    rows, cols = s.kkt_pattern()
    LDLT = hipldl.HIPLDLStruct(s.N, rows, cols, vals, s.nvar, s.nequ, s.ncon, batch=B, options=opts)
    if plan_kind == "throughput":
        assert LDLT.info["v2"] is not None and LDLT.info["v2"]["fronts32"] + LDLT.info["v2"]["fronts64"] == 0
        assert LDLT.config["kernel"] == "v2"
    ok = hipldl.try_to_factorize(LDLT, vals, s.nvar, s.nequ, s.ncon, 2.220446049250313e-16)
    assert ok.all()
    perm = LDLT.plan_array("perm").astype(np.int64)
    orc = O.Oracle(s.N, rows, cols, perm)
    for k in range(3):
        r = np.ascontiguousarray(rhs * (k + 1) + k)
        d = np.zeros((B, s.N))
        hipldl.solve_ldl_(r, LDLT.factor, d)
        for b in range(B):
            assert orc.try_to_factorize(vals[b], s.nvar, s.nequ, s.ncon, 2.220446049250313e-16)
            d0 = orc.solve_ldl(r[b])
            assert np.abs(d[b] - d0).max() <= FWD_TOL * np.abs(d0).max()
            assert backward_error(s, vals[b], r[b], d[b]) <= BWD_TOL
    LDLT.close()


def test_cfg2_dense_backend_against_oracle(built):
    """BASELINE config 2 pattern (dense Jacobian, unconstrained) at a size the oracle handles in seconds: served by the dense
    backend (J'WJ by GEMM, blocked dense LDL^T with hand-written fp64 MFMA trailing updates), including the rho ladder and the literal
    try_to_factorize / solve_ldl! sequence."""
    hipldl, syn, O = _mods()
    s = syn.dense_structure(96, 200)
    rows, cols = s.kkt_pattern()
    B = 3
    vals = np.stack([syn.dense_values(s, 2000 + b)[0] for b in range(B)])
    rhs = np.stack([syn.dense_values(s, 2000 + b)[1] for b in range(B)])
    L0 = hipldl.HIPLDLStruct(s.N, rows, cols, None, s.nvar, s.nequ, s.ncon, batch=1)
    assert L0.config["kernel"] == "dense"
    L0.close()
    run_case(s, vals, rhs)
    off = s.offsets()
    v2 = vals.copy()
    v2[:, off[0]:off[1]] = -3.0          # indefinite top-left block: the ladder has to climb
    v2[:, off[4]:off[5]] = -np.random.default_rng(0).uniform(0.5, 2.0, (B, s.nequ))
    run_case(s, v2, rhs, rho_old=np.array([0.0, 2.0, 0.0]), check_fwd=False)
    # two-call path
    LDLT = hipldl.HIPLDLStruct(s.N, rows, cols, vals, s.nvar, s.nequ, s.ncon, batch=B)
    ok = hipldl.try_to_factorize(LDLT, vals, s.nvar, s.nequ, s.ncon, 2.220446049250313e-16)
    assert ok.all()
    for k in range(2):
        r = np.ascontiguousarray(rhs * (k + 1))
        d = np.zeros((B, s.N))
        hipldl.solve_ldl_(r, LDLT.factor, d)
        for b in range(B):
            assert backward_error(s, vals[b], r[b], d[b]) <= BWD_TOL
    LDLT.close()


def test_cfg2_dense_full_size_properties(built):
    """BASELINE config 2 at full size (n = 1000, nequ = 2000, dense Jacobian: 2 004 000 COO entries): backward error of the
    Newton step and linearity of the solve."""
    hipldl, syn, O = _mods()
    s = syn.dense_structure(1000, 2000)
    rows, cols = s.kkt_pattern()
    vals, rhs = syn.dense_values(s, 2002)
    L = hipldl.HIPLDLStruct(s.N, rows, cols, None, s.nvar, s.nequ, s.ncon, batch=1)
    assert L.config["kernel"] == "dense"
    p = hipldl.default_params()
    d1, ok, rho, ro, nf = hipldl.newton_system_(np.zeros(s.N), s.nvar, s.nequ, s.ncon, rhs, vals.copy(), L, 0.0, p)
    assert ok and nf == 1 and rho == 0.0
    assert backward_error(s, vals, rhs, d1) <= BWD_TOL
    d2, ok2, *_ = hipldl.newton_system_(np.zeros(s.N), s.nvar, s.nequ, s.ncon, 3.0 * rhs, vals.copy(), L, 0.0, p)
    assert ok2 and np.abs(d2 - 3.0 * d1).max() <= 1e-12 * np.abs(d1).max()
    L.close()


def test_f4_cgls_multipliers(built):
    """batched CGLS estimate of the multipliers (src/CaNNOLeS.jl:507-518) against the restated recurrence
    (same iteration counts, values to 1e-10) and against the least-squares solution."""
    import torch
    hipldl, syn, O = _mods()
    s = syn.band_structure(1000, 10)
    B = 5
    vals, _ = syn.batch_values(s, B, cfg=4)
    rng = np.random.default_rng(8)
    r = rng.standard_normal((B, s.nequ))
    r[3] = 0.0  # zero residual -> lambda = ones
    rows, cols = s.kkt_pattern()
    LDLT = hipldl.HIPLDLStruct(s.N, rows, cols, None, s.nvar, s.nequ, s.ncon, batch=B)
    dev = torch.device("cuda", 0)
    tv, tr = torch.from_numpy(vals).to(dev), torch.from_numpy(r).to(dev)
    tl = torch.zeros((B, s.ncon), dtype=torch.float64, device=dev)
    tj = torch.zeros((B, s.nvar), dtype=torch.float64, device=dev)
    ti = torch.zeros(B, dtype=torch.int32, device=dev)
    hipldl.cgls_multipliers_dev(LDLT, tv.data_ptr(), tr.data_ptr(), tl.data_ptr(), tj.data_ptr(), iters_ptr=ti.data_ptr())
    torch.cuda.synchronize()
    lam, jxtr, its = tl.cpu().numpy(), tj.cpu().numpy(), ti.cpu().numpy()
    for b in range(B):
        lam0, jx0, it0 = O.cgls_multipliers(rows, cols, vals[b], s.nvar, s.nequ, s.ncon, r[b])
        assert np.array_equal(jxtr[b], jx0)  # mul!(Jxtr, Jx', r): same COO-order sums, separately rounded
        assert its[b] == it0
        np.testing.assert_allclose(lam[b], lam0, rtol=1e-10, atol=1e-12)
    assert np.array_equal(lam[3], np.ones(s.ncon))
    # the same row reading the Jacobian values from the model's arrays (cnl_cgls_multipliers_jac_dev): bit-equal
    off = s.offsets()
    tJ, tJc = tv[:, off[2]:off[3]].contiguous(), tv[:, off[3]:off[4]].contiguous()
    tl2, tj2, ti2 = torch.zeros_like(tl), torch.zeros_like(tj), torch.zeros_like(ti)
    hipldl.cgls_multipliers_jac_dev(LDLT, s.nnzjF, s.nnzjc, tJ.data_ptr(), tJc.data_ptr(), tr.data_ptr(), tl2.data_ptr(), tj2.data_ptr(), iters_ptr=ti2.data_ptr())
    torch.cuda.synchronize()
    assert torch.equal(tl2, tl) and torch.equal(tj2, tj) and torch.equal(ti2, ti)
    LDLT.close()


def test_dense_backend_with_constraints(built):
    """Dense residual Jacobian WITH constraints: the constraint rows border the dense system (order nvar + ncon,
    quasi-definite, factorised without pivoting); includes duplicate H_F / H_c entries and the rho ladder."""
    hipldl, syn, O = _mods()
    n, m, p = 64, 150, 6
    rng = np.random.default_rng(12)
    jr, jc = np.tile(np.arange(1, m + 1), n), np.repeat(np.arange(1, n + 1), m)          # dense J_F, column-major
    hd = np.arange(1, n + 1)
    hF = (np.concatenate([hd, hd[1:]]), np.concatenate([hd, hd[:-1]]))                   # diagonal + first subdiagonal
    hc = (hd, hd)                                                                         # duplicates of the diagonal
    cr = np.repeat(np.arange(1, p + 1), 10)
    cc = np.concatenate([rng.choice(n, 10, replace=False) + 1 for _ in range(p)])
    s = syn.Structure(n, m, p, hF, hc, (jr, jc), (cr, cc), name="dense+con")
    rows, cols = s.kkt_pattern()
    off = s.offsets()
    B = 3
    vals = np.zeros((B, s.nnzNS))
    rhs = rng.standard_normal((B, s.N))
    for b in range(B):
        vals[b, off[0]:off[1]] = np.concatenate([rng.uniform(0.2, 1.0, n), rng.uniform(-0.05, 0.05, n - 1)])
        vals[b, off[1]:off[2]] = rng.uniform(-0.05, 0.05, n)
        vals[b, off[2]:off[3]] = rng.standard_normal(m * n) / np.sqrt(n)
        vals[b, off[3]:off[4]] = rng.uniform(-1, 1, len(cr))
        vals[b, off[4]:off[5]] = -rng.uniform(0.5, 2.0, m)
        vals[b, off[5]:off[6]] = -0.1
    L0 = hipldl.HIPLDLStruct(s.N, rows, cols, None, s.nvar, s.nequ, s.ncon, batch=1)
    assert L0.config["kernel"] == "dense"
    L0.close()
    run_case(s, vals, rhs)
    v2 = vals.copy()
    v2[:, off[0]:off[0] + n] = -4.0  # indefinite: the ladder climbs
    run_case(s, v2, rhs, check_fwd=False)


@pytest.mark.parametrize("shape", [(300, 400, 10, 0.02, 1), (200, 260, 0, 0.04, 2)])
def test_irregular_sparsity_dense_treatment(built, shape):
    """Irregular sparse Jacobians whose fill gives fronts of order > 64: for a small batch the condensed system (order nvar +
    kept rows + ncon) is factorised as ONE dense matrix (condense pass -> blocked dense LDL^T -> post-pass).  Newton step with
    the rho ladder, and the two-call sequence, against the oracle."""
    hipldl, syn, O = _mods()
    n, m, p, dens, seed = shape
    s = syn.random_structure(n, m, p, dens, seed=seed)
    rows, cols = s.kkt_pattern()
    B = 3
    vals = np.stack([syn.random_values(s, 40 + b)[0] for b in range(B)])
    rhs = np.stack([syn.random_values(s, 40 + b)[1] for b in range(B)])
    L0 = hipldl.HIPLDLStruct(s.N, rows, cols, None, s.nvar, s.nequ, s.ncon, batch=B)
    assert L0.info["fmax"] > 64 and L0.config["kernel"] == "dense"
    L0.close()
    run_case(s, vals, rhs)
    v2 = vals.copy()
    off = s.offsets()
    v2[:, off[0]:off[1]] *= -20.0  # indefinite top-left block: ladder
    run_case(s, v2, rhs, check_fwd=False)
    LDLT = hipldl.HIPLDLStruct(s.N, rows, cols, vals, s.nvar, s.nequ, s.ncon, batch=B)
    ok = hipldl.try_to_factorize(LDLT, vals, s.nvar, s.nequ, s.ncon, 2.220446049250313e-16)
    assert ok.all()
    d = np.zeros((B, s.N))
    hipldl.solve_ldl_(rhs, LDLT.factor, d)
    for b in range(B):
        assert backward_error(s, vals[b], rhs[b], d[b]) <= BWD_TOL
    LDLT.close()


# ---- parity hardening (round 2) -----------------------------------------------------------------------------------------
def _fixtures():
    import json
    import os
    return json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fixtures.json")))


def test_fixtures_f2_f3_through_the_hip_path(built):
    """F2 (slot (1,1) of MGH01CON summed from THREE COO entries — H_F, H_c and rho0 — in COO order, src/solver_types.jl:53-59)
    and F3 (rho ladder: failures at 0, rho0, 100 rho0, ... success at 605.5, nfact = 6, src/CaNNOLeS.jl:1029-1047) from
    tests/golden/fixtures.json, through the C ABI."""
    hipldl, syn, O = _mods()
    fx = _fixtures()
    F1, F2, F3 = fx["F1"], fx["F2"], fx["F3"]
    rows, cols = np.array(F1["rows"], np.int64), np.array(F1["cols"], np.int64)
    vals = np.array(F2["vals"])
    rhs = np.array(F1["rhs"])
    L = hipldl.HIPLDLStruct(5, rows, cols, vals, F1["nvar"], F1["nequ"], F1["ncon"])
    ok, npos, nzer = hipldl.try_to_factorize(L, vals, 2, 2, 1, 2.220446049250313e-16, return_inertia=True)
    assert ok and (npos, nzer) == (2, 0)
    d = np.zeros(5)
    hipldl.solve_ldl_(rhs, L.factor, d)
    assert np.abs(d - np.array(F2["d"])).max() <= 1e-12 * np.abs(np.array(F2["d"])).max()
    L.close()
    rows, cols = np.array(F3["rows"], np.int64), np.array(F3["cols"], np.int64)
    vals = np.array(F3["vals"])
    rhs = np.array(F3["rhs"])
    L = hipldl.HIPLDLStruct(6, rows, cols, vals, F3["nvar"], F3["nequ"], F3["ncon"])
    d, ok, rho, rho_old, nfact = hipldl.newton_system_(np.zeros(6), 3, 3, 0, rhs, vals, L, 0.0, hipldl.default_params())
    tried = F3["rho_tried"]
    assert ok and nfact == len(tried) == 6 and rho == tried[-1] and rho_old == tried[-1]
    assert np.array_equal(vals[-3:], np.full(3, tried[-1]))  # the rho slots hold the last rho tried
    K = syn.dense_kkt(type("S", (), {"kkt_pattern": lambda self: (rows, cols), "N": 6})(), vals)
    assert np.abs(K @ d + rhs).max() <= 1e-12 * (np.abs(K).sum(axis=1).max() * np.abs(d).max() + np.abs(rhs).max())
    L.close()


@pytest.mark.parametrize("B,kernel,order,forced,band_nl", [
    (9216, "v2", "canonical", False, 32),   # what bench.py TIMES: band_newton_kernel<32, false> with interleaved factor records (any batch above 8192)
    (8192, "v2", "canonical", False, 16),   # rounds 1-4's headline batch: band_newton_kernel<16, false>, problem-major factor records
    (4608, "v2", "canonical", True, 16), (4608, "v2-staged", "ndc2", False, 0),
    (3584, "v2-staged", "ndc2", False, 0), (1536, "v2-staged", "ndc", False, 0)])
def test_headline_pattern_large_batch_sample_vs_oracle(built, B, kernel, order, forced, band_nl):
    """The headline shape (cfg3 pattern, n = nequ = 1e4, ncon = 50).  B = 9216: the INSTANTIATION bench.py times (every batch above
    8192 problems runs band_newton_kernel<32, false>: 32 problems per workgroup, factor records interleaved over the workgroup —
    csrc/band.hip LINT; bench.py's 16 384 problems differ only in the grid size), with one problem that climbs the rho ladder inside a
    workgroup of convex ones; B = 8192: the band kernels with 16 problems per workgroup; B = 4608 forced onto the throughput plan;
    then a batch between one and two wavefronts per SIMD (B = 4608: the first 4096 problems on the bidirectional chain, the remaining
    512 behind them on a handle of their own with a many-part plan — csrc/capi.cpp, run_split, cnl_options.split_tail), one the
    bidirectional chain serves alone (B = 3584) and one with many large parts (B = 1536): the whole batch through
    cnl_newton_system_dev, a random sample of 32 problems against the oracle; in the split batch one problem of each part also
    climbs the rho ladder.  config["band"] / ["band_nl"] are asserted: which kernel ran is part of what the test pins."""
    import torch
    hipldl, syn, O = _mods()
    import bench as BM
    s = syn.band_structure(10000, 50)
    rows, cols = s.kkt_pattern()
    dev = torch.device("cuda", 0)
    vals = torch.empty((B, s.nnzNS), dtype=torch.float64, device=dev)
    rhs = torch.empty((B, s.N), dtype=torch.float64, device=dev)
    host = {}
    pick = np.sort(np.random.default_rng(5).choice(B, 32, replace=False))
    for b0 in range(0, B, 512):
        vh, rh = BM.band_batch(s, 512, seed=9000 + b0)
        vals[b0:b0 + 512].copy_(torch.from_numpy(vh))
        rhs[b0:b0 + 512].copy_(torch.from_numpy(rh))
        for b in pick[(pick >= b0) & (pick < b0 + 512)]:
            host[int(b)] = (vh[b - b0].copy(), rh[b - b0].copy())
    p = hipldl.default_params()
    split = B == 4608 and not forced
    climbers = (int(pick[0]), int(pick[-1])) if split else ((int(pick[7]),) if band_nl == 32 else ())
    if climbers:   # split: one problem in the chain part and one in the remainder need rho > 0; band<32>: one inside a workgroup
        off = s.offsets()
        hF_r, hF_c = np.asarray(s.hF[0]), np.asarray(s.hF[1])
        dg = torch.from_numpy(off[0] + np.nonzero(hF_r == hF_c)[0][:400]).to(dev)
        for b in climbers:
            vals[b, dg] = -30.0
            host[b][0][dg.cpu().numpy()] = -30.0
        assert not split or pick[0] < 4096 <= pick[-1]
    L = hipldl.HIPLDLStruct(s.N, rows, cols, None, s.nvar, s.nequ, s.ncon, batch=B,
                            options=hipldl.Options(plan_kind=hipldl.PLAN_THROUGHPUT) if forced else None)
    assert L.config["kernel"] == kernel and L.info["order"].startswith(order) and L.config["tail"] == split
    assert L.config["band"] == (band_nl > 0) and L.config["band_nl"] == band_nl
    d = torch.zeros((B, s.N), dtype=torch.float64, device=dev)
    ro = torch.zeros(B, dtype=torch.float64, device=dev)
    rho = torch.ones(B, dtype=torch.float64, device=dev)
    nf = torch.zeros(B, dtype=torch.int32, device=dev)
    ok = torch.zeros(B, dtype=torch.int32, device=dev)
    vals_in = vals.clone() if band_nl == 32 else None
    hipldl.newton_system_dev(L, vals.data_ptr(), rhs.data_ptr(), d.data_ptr(), ro.data_ptr(), rho.data_ptr(), nf.data_ptr(), ok.data_ptr(), p, 0)
    torch.cuda.synchronize()
    if band_nl == 32:
        # what bench.py times since round 6: the same instantiation on `vals` interleaved over groups of 32 problems (cnl_options.batch_layout
        # = 1) — every output bit-equal to the problem-major call that is checked against the oracle below
        Li = hipldl.HIPLDLStruct(s.N, rows, cols, None, s.nvar, s.nequ, s.ncon, batch=B, options=hipldl.Options(batch_layout=hipldl.LAYOUT_INTERLEAVED))
        assert Li.config["band"] and Li.config["band_nl"] == 32 and Li.config["batch_layout"] == 1
        vi = torch.empty(hipldl.layout_len(Li, 0), dtype=torch.float64, device=dev)
        hipldl.interleave_dev(Li, 0, vals_in.data_ptr(), vi.data_ptr(), 0)
        d_i, ro_i, rho_i = torch.zeros_like(d), torch.zeros_like(ro), torch.ones_like(rho)
        nf_i, ok_i = torch.zeros_like(nf), torch.zeros_like(ok)
        hipldl.newton_system_dev(Li, vi.data_ptr(), rhs.data_ptr(), d_i.data_ptr(), ro_i.data_ptr(), rho_i.data_ptr(), nf_i.data_ptr(), ok_i.data_ptr(), p, 0)
        hipldl.deinterleave_dev(Li, 0, vi.data_ptr(), vals_in.data_ptr(), 0)
        torch.cuda.synchronize()
        assert torch.equal(d_i, d) and torch.equal(ro_i, ro) and torch.equal(rho_i, rho) and torch.equal(nf_i, nf) and torch.equal(ok_i, ok)
        assert torch.equal(vals_in, vals)
        Li.close()
        del vi, d_i, vals_in
    nfh, rhoh = nf.cpu().numpy(), rho.cpu().numpy()
    assert bool((ok == 1).all())
    if not climbers:
        assert (nfh == 1).all() and (rhoh == 0).all()
    else:
        assert int((nfh > 1).sum()) == len(climbers)
    orc = O.Oracle(s.N, rows, cols, L.plan_array("perm").astype(np.int64))
    dh = d[torch.from_numpy(pick).to(dev)].cpu().numpy()
    vend = vals[torch.from_numpy(pick).to(dev)].cpu().numpy()
    for k, b in enumerate(pick):
        v0, r0 = host[int(b)]
        vv = v0.copy()
        d0, ok0, rho0, ro0, nf0 = O.newton_system(orc, s.nvar, s.nequ, s.ncon, r0, vv, 0.0, O.default_params())
        assert ok0 and (nf0, rho0) == (int(nfh[b]), float(rhoh[b]))
        assert (nf0 > 1) == (int(b) in climbers)
        assert np.array_equal(vend[k, -s.nvar:], vv[-s.nvar:])
        assert np.abs(dh[k] - d0).max() <= FWD_TOL * np.abs(d0).max()
        assert backward_error(s, vv, r0, dh[k]) <= BWD_TOL
    L.close()


def test_cfg4_batch_256(built):
    """BASELINE config 4 at its stated batch: 256 problems n = nequ = 1e3, ncon = 10, every one against the oracle."""
    hipldl, syn, O = _mods()
    s = syn.band_structure(1000, 10)
    vals, rhs = syn.batch_values(s, 256, cfg=4)
    info, cfg = run_case(s, vals, rhs)
    assert cfg["kernel"] == "v2-staged"


@pytest.mark.parametrize("plan_kind", ["latency", "throughput"])
def test_near_singular_sweep_decisions(built, plan_kind):
    """96 systems whose first factorisation sits near the inertia threshold: the curvature of the last variable eliminated
    is shifted so that its pivot (as the oracle computes it) becomes +-10^e.
      * e in [-11, -8] (64 systems): far above the rounding noise of any LDL^T of these matrices (~1e-15: entries are O(1),
        the summation order of the condensed block and the division differ from the reference's): (success, nfact, rho,
        rho_old) must be the oracle's, bit for bit.
      * e in [-17, -13] (32 systems): the pivot is BELOW that noise, so its sign is not determined by the data — the oracle
        with another ordering would flip it too.  Required there: a decision that follows the rules (nfact = 1 with rho = 0, or
        a rung of the ladder, src/CaNNOLeS.jl:1029-1047) and a solution of the system it reports (backward error with the rho
        left in the rho slots) — and, since round 5, an exact rule instead of a bound on how many may differ: the oracle decides every
        system fourteen times (two unrelated orders x the data and six perturbations of it by 8 ulps); where all agree the product
        must decide the same, where they do not (the sign is not a property of the data) it must take one of their decisions."""
    hipldl, syn, O = _mods()
    opts = hipldl.Options(plan_kind=hipldl.PLAN_THROUGHPUT) if plan_kind == "throughput" else None
    s = syn.band_structure(120, 2)
    rows, cols = s.kkt_pattern()
    B = 96
    vals, rhs = syn.batch_values(s, B, cfg=5)
    off = s.offsets()
    p = hipldl.default_params()
    L = hipldl.HIPLDLStruct(s.N, rows, cols, None, s.nvar, s.nequ, s.ncon, batch=B, options=opts)
    perm = L.plan_array("perm").astype(np.int64)
    if L.config.get("band"):
        # the band kernels (round 5) eliminate the two halves of the chain towards the junction: THEIR last pivot is the junction's
        # last variable, and the oracle is given an order that ends with it (any order with that variable last has the same last
        # pivot: the Schur complement of everything else)
        bi = L.plan_array("band_info")
        jb = int(bi[2]) + 3 if bi[1] == 2 else s.nvar - 1
        perm = np.array([v for v in perm if v != jb] + [jb], dtype=np.int64)
    orc = O.Oracle(s.N, rows, cols, perm)
    hF_r, hF_c = np.asarray(s.hF[0]), np.asarray(s.hF[1])
    rng = np.random.default_rng(11)
    j = [int(v) for v in perm if v < s.nvar][-1]
    ipos = int(np.nonzero(perm == j)[0][0])
    slot = off[0] + int(np.nonzero((hF_r == j + 1) & (hF_c == j + 1))[0][0])
    expo = np.concatenate([rng.uniform(-11, -8, 64), rng.uniform(-17, -13, 32)])
    for b in range(B):
        orc.try_to_factorize(vals[b], s.nvar, s.nequ, s.ncon, p[0])
        vals[b, slot] += float(rng.choice([-1.0, 1.0]) * 10.0 ** expo[b]) - orc.D[ipos]
    v = vals.copy()
    d, ok, rho, ro, nf = hipldl.newton_system_(np.zeros((B, s.N)), s.nvar, s.nequ, s.ncon, rhs, v, L, np.zeros(B), p)
    L.close()
    v0 = vals.copy()
    d0, ok0, rho0, ro0, nf0 = O.newton_system_batch(orc, B, s.nvar, s.nequ, s.ncon, rhs, v0, np.zeros(B), p)
    clear = np.arange(B) < 64
    assert np.array_equal(ok[clear], ok0[clear]) and np.array_equal(nf[clear], nf0[clear])
    assert np.array_equal(rho[clear], rho0[clear]) and np.array_equal(ro[clear], ro0[clear])
    ladder = [0.0, p[5]]
    while ladder[-1] * p[4] <= p[6]:
        ladder.append(ladder[-1] * p[4])
    # EXACT rule (round 5; before: "at most 10 of 32 may differ").  Any LDL' is backward stable only up to a few ulps of the
    # entries, so the sign of a pivot is a property of the DATA exactly when it survives (a) another elimination order
    # (oracle.canonical_perm: r-nodes, x in natural order, multipliers — an order the product had no part in) and (b) relative
    # perturbations of every entry by 8 ulps (six random sign patterns, on both orders).  Where all fourteen oracle runs agree, the
    # product must decide the same; where they do not, it must take one of their decisions.
    orc2 = O.Oracle(s.N, rows, cols, O.canonical_perm(s.nvar, s.nequ, s.ncon))
    decisions = [nf0.copy()]
    prng = np.random.default_rng(5)
    for o_ in (orc, orc2):
        for trial in range(7):
            vp = vals.copy() if (trial == 0) else vals * (1.0 + 8.0 * np.finfo(np.float64).eps * prng.choice([-1.0, 1.0], size=vals.shape))
            if o_ is orc and trial == 0:
                continue
            decisions.append(O.newton_system_batch(o_, B, s.nvar, s.nequ, s.ncon, rhs, vp, np.zeros(B), p)[4].copy())
    decisions = np.stack(decisions)                       # [14, B]
    unanimous = (decisions == decisions[0]).all(axis=0)
    assert unanimous[clear].all()
    differ = undetermined = 0
    for b in range(B):
        assert ok[b] and 1 <= nf[b] <= len(ladder)
        assert rho[b] == ladder[nf[b] - 1] and ro[b] == rho[b]
        assert np.array_equal(v[b, -s.nvar:], np.full(s.nvar, rho[b]))       # what the rho slots hold
        # a pivot of 1e-8 .. 1e-17 without pivoting is not backward stable (growth ~ 1/pivot), in the reference either: the
        # solution is only required to be finite and to solve the reported system to working accuracy times that growth
        assert np.isfinite(d[b]).all()
        if b < 64:
            assert backward_error(s, v[b], rhs[b], d[b]) <= 1e-7
        differ += int(nf[b] != nf0[b])
        if not unanimous[b]:
            undetermined += 1
            assert nf[b] in decisions[:, b], (b, expo[b], nf[b], decisions[:, b])
        else:
            assert nf[b] == nf0[b], (b, expo[b], nf[b], decisions[:, b])
    print(f"near-singular sweep ({plan_kind}): {differ} of 32 sub-noise decisions differ from the oracle's on the product's order; "
          f"{undetermined} are undetermined (the oracle's decision changes with the order or with 8 ulps on the entries), so the exact "
          f"rule pins {32 - undetermined} of the 32 sub-noise systems here (plus all 64 clear ones, bit for bit)")


def test_cfg2_dense_full_size_against_oracle(built):
    """BASELINE config 2 at full size (n = 1000, nequ = 2000, 2 004 000 COO entries) against the oracle once (seconds on
    one core): same decision, forward error at the stated tolerance."""
    hipldl, syn, O = _mods()
    s = syn.dense_structure(1000, 2000)
    vals, rhs = syn.dense_values(s, 2003)
    run_case(s, vals[None, :], rhs[None, :])


@pytest.mark.parametrize("kind", ["dense", "sparse"])
def test_factorize_newton_solve_sequence(built, kind):
    """cnl_factorize(A), cnl_newton_system(B), cnl_solve(rhs): the solve uses the LAST factorisation, B's (with B's rho),
    never a mixture of B's factor and A's values."""
    hipldl, syn, O = _mods()
    if kind == "dense":
        s = syn.dense_structure(96, 200)
        gen = lambda k: syn.dense_values(s, 3000 + k)
    else:
        s = syn.band_structure(300, 4)
        gen = lambda k: syn.band_values(s, 3000 + k)
    rows, cols = s.kkt_pattern()
    (vA, rA), (vB, rB) = gen(0), gen(1)
    off = s.offsets()
    vB[off[0]:off[1]] *= 1.5
    p = hipldl.default_params()
    L = hipldl.HIPLDLStruct(s.N, rows, cols, vA, s.nvar, s.nequ, s.ncon)
    assert hipldl.try_to_factorize(L, vA, s.nvar, s.nequ, s.ncon, p[0])
    vB2 = vB.copy()
    d, ok, rho, ro, nf = hipldl.newton_system_(np.zeros(s.N), s.nvar, s.nequ, s.ncon, rB, vB2, L, 0.0, p)
    assert ok
    d2 = np.zeros(s.N)
    hipldl.solve_ldl_(rA, L.factor, d2)   # K_B d2 = -rA
    assert backward_error(s, vB2, rA, d2) <= BWD_TOL
    L.close()


def test_default_params_of_product_and_oracle_agree(built):
    """cnl_default_params (the product) and cnlo_default_params (the oracle) restate ParamCaNNOLeS(Float64)
    (src/CaNNOLeS.jl:48-62) independently; run_case gives each side its own, this pins them to each other and to the
    reference's closed forms on the GPU box as well."""
    hipldl, syn, O = _mods()
    p, q = hipldl.default_params(), O.default_params()
    assert np.array_equal(p, q)
    eps = np.finfo(float).eps
    assert p[0] == eps and p[1] == np.sqrt(eps) and p[2] == 1 / 3 and p[3] == 8.0 and p[4] == 100.0
    # rho0 = eps^T(1/3) is a pow with the exponent 0.3333333333333333 (src/CaNNOLeS.jl:56), 6.055454452393343e-6 as SURVEY a12
    # quotes it — not cbrt(eps), which is 2.5 ulp away (rounds 1 and 2 had cbrt on BOTH sides: invisible until each side got its
    # own defaults and this literal)
    assert p[5] == 6.055454452393343e-06 == eps ** (1 / 3)
    assert p[6] == eps ** -2.0 and p[7] == np.sqrt(eps) and p[8] == eps ** 0.25


@pytest.mark.parametrize("B", [1, 4])
def test_dataflow_timeout_is_counted_and_recovered(built, B):
    """The dataflow execution of the smallest batches waits on device counters (kernels2.hip, spin_until).  With
    cnl_options.dataflow_spin_limit = 1 practically every wait gives up: the attempt's results are then not to be trusted, the
    waits are COUNTED (cnl_dataflow_timeouts) and the sequential launch behind the attempt redoes the batch — newton_system!,
    try_to_factorize and solve_ldl! must still give the oracle's answers.  (Round 2: a wait that gave up ran on silently.)"""
    hipldl, syn, O = _mods()
    s = syn.band_structure(2000, 10)
    rows, cols = s.kkt_pattern()
    vals, rhs = syn.batch_values(s, B, cfg=5, stress="ladder" if B > 1 else None)
    if B > 1:
        v4, r4 = syn.batch_values(s, B, cfg=4)
        vals[1:], rhs[1:] = v4[1:], r4[1:]      # problem 0 climbs the ladder, the others succeed at rho = 0
    p = hipldl.default_params()
    L = hipldl.HIPLDLStruct(s.N, rows, cols, None, s.nvar, s.nequ, s.ncon, batch=B, options=hipldl.Options(dataflow_spin_limit=1))
    assert L.config["kernel"] == "v2-staged"
    orc = O.Oracle(s.N, rows, cols, L.plan_array("perm").astype(np.int64))
    v = vals.copy()
    d, ok, rho, ro, nf = hipldl.newton_system_(np.zeros((B, s.N)), s.nvar, s.nequ, s.ncon, rhs, v, L, np.zeros(B), p)
    t1 = L.dataflow_timeouts()
    assert t1 > 0, "spin limit 1 did not provoke a single timeout: the test does not test anything"
    d0, ok0, rho0, ro0, nf0 = O.newton_system_batch(orc, B, s.nvar, s.nequ, s.ncon, rhs, vals.copy(), np.zeros(B), O.default_params())
    ok, rho, ro, nf = (np.atleast_1d(a) for a in (ok, rho, ro, nf))
    assert np.array_equal(ok, ok0) and np.array_equal(nf, nf0) and np.array_equal(rho, rho0) and np.array_equal(ro, ro0)
    d = d.reshape(B, s.N)
    for b in range(B):
        assert np.abs(d[b] - d0[b]).max() <= FWD_TOL * np.abs(d0[b]).max()
    # the two-call sequence under the same conditions
    okf, npos, nzer = hipldl.try_to_factorize(L, v, s.nvar, s.nequ, s.ncon, p[0], return_inertia=True)
    okf = np.atleast_1d(okf)
    for b in range(B):
        ob, np0, nz0 = orc.try_to_factorize(v.reshape(B, -1)[b], s.nvar, s.nequ, s.ncon, p[0], return_inertia=True)
        assert bool(okf[b]) == bool(ob) and (int(np.atleast_1d(npos)[b]), int(np.atleast_1d(nzer)[b])) == (np0, nz0)
    assert okf.all()
    d2 = np.zeros((B, s.N))
    hipldl.solve_ldl_(rhs, L.factor, d2)
    for b in range(B):
        assert backward_error(s, v.reshape(B, -1)[b], rhs[b], d2.reshape(B, -1)[b]) <= BWD_TOL
    assert L.dataflow_timeouts() > t1
    L.close()
    # and an undisturbed handle counts nothing
    L2 = hipldl.HIPLDLStruct(s.N, rows, cols, None, s.nvar, s.nequ, s.ncon, batch=B)
    hipldl.newton_system_(np.zeros((B, s.N)), s.nvar, s.nequ, s.ncon, rhs, vals.copy(), L2, np.zeros(B), p)
    assert L2.dataflow_timeouts() == 0
    L2.close()


@pytest.mark.parametrize("lean", [1, 0])
@pytest.mark.parametrize("B", [5, 44])
@pytest.mark.parametrize("shape", [(600, 6, 2), (1000, 10, 2), (400, 0, 2)])
def test_lean_and_full_instantiations_agree_with_the_oracle(built, shape, B, lean):
    """Plans whose fronts are all fast-class row-form fronts run the kernels' LEAN instantiation (no MODE_SOLVE section, no
    out-of-line front classes, no product lists: csrc/kernels2.hip); cnl_options.lean_kernel = 0 keeps the full one.  Both, on
    throughput and latency plans, with a ladder problem and a hopeless one, against the oracle; solve_ldl! (which always runs the
    full instantiation) must find the factor the lean one stored."""
    hipldl, syn, O = _mods()
    n, p, hw = shape
    s = syn.band_structure(n, p, hw=hw)
    rows, cols = s.kkt_pattern()
    vals, rhs = syn.batch_values(s, B, cfg=3)
    off = s.offsets()
    vl, rl = syn.batch_values(s, B, cfg=5, stress="ladder")
    vals[B - 2], rhs[B - 2] = vl[B - 2], rl[B - 2]
    vals[1, off[0]:off[1]] = np.nan
    for kind in (hipldl.PLAN_THROUGHPUT, hipldl.PLAN_LATENCY):
        opts = hipldl.Options(plan_kind=kind, lean_kernel=lean)
        info, cfg = run_case(s, vals, rhs, options=opts, check_fwd=False)
        assert lean or not cfg["lean"]
        if kind == hipldl.PLAN_THROUGHPUT and p > 0:   # chain-like order: every front in row form
            assert cfg["lean"] == bool(lean)
    L = hipldl.HIPLDLStruct(s.N, rows, cols, None, s.nvar, s.nequ, s.ncon, batch=B, options=hipldl.Options(plan_kind=hipldl.PLAN_THROUGHPUT, lean_kernel=lean))
    prm = hipldl.default_params()
    good = [b for b in range(B) if b not in (1, B - 2)]
    ok = hipldl.try_to_factorize(L, vals, s.nvar, s.nequ, s.ncon, prm[0])
    assert all(ok[b] for b in good) and not ok[1]
    d = np.zeros((B, s.N))
    hipldl.solve_ldl_(rhs, L.factor, d)
    for b in good[:6]:
        assert backward_error(s, vals[b], rhs[b], d[b]) <= BWD_TOL
    L.close()


def test_split_batch_factorize_then_solve(built):
    """try_to_factorize + solve_ldl! (src/solver_types.jl:69-98) on a batch between one and two wavefronts per SIMD: the handle
    runs its first 4096 problems on the bidirectional chain and the remaining 904 on a handle of their own (run_split); both
    parts must give the oracle's inertia decisions and solutions, and solve_ldl! must find each part's own factor."""
    hipldl, syn, O = _mods()
    s = syn.band_structure(600, 6)
    rows, cols = s.kkt_pattern()
    B = 5000
    v8, r8 = syn.batch_values(s, 8, cfg=3)
    rng = np.random.default_rng(3)
    vals = np.tile(v8, (B // 8, 1)) * (1.0 + 1e-3 * rng.standard_normal((B, 1)))
    rhs = np.tile(r8, (B // 8, 1)) + 1e-3 * np.arange(B)[:, None]
    off = s.offsets()
    vals[:, off[4]:off[5]] = -1.0
    bad = [7, 4999]
    hF_r, hF_c = np.asarray(s.hF[0]), np.asarray(s.hF[1])
    dg = off[0] + np.nonzero(hF_r == hF_c)[0]
    for b in bad:
        vals[b, dg[:50]] = -40.0        # wrong inertia at rho = 0
    L = hipldl.HIPLDLStruct(s.N, rows, cols, None, s.nvar, s.nequ, s.ncon, batch=B)
    assert L.config["kernel"] == "v2-staged" and L.info["order"].startswith("ndc2")
    p = hipldl.default_params()
    ok, npos, nzer = hipldl.try_to_factorize(L, vals, s.nvar, s.nequ, s.ncon, p[0], return_inertia=True)
    assert ok.sum() == B - 2 and not ok[7] and not ok[4999]
    orc = O.Oracle(s.N, rows, cols, L.plan_array("perm").astype(np.int64))
    d = np.zeros((B, s.N))
    hipldl.solve_ldl_(rhs, L.factor, d)
    for b in (0, 7, 2500, 4095, 4096, 4500, 4999):
        ob, np0, nz0 = orc.try_to_factorize(vals[b], s.nvar, s.nequ, s.ncon, p[0], return_inertia=True)
        assert bool(ok[b]) == ob and (int(npos[b]), int(nzer[b])) == (np0, nz0)
        if ob:
            d0 = orc.solve_ldl(rhs[b])
            assert np.abs(d[b] - d0).max() <= FWD_TOL * np.abs(d0).max()
    L.close()


@pytest.mark.parametrize("nA,kernel,order", [(4096, "v2-staged", "ndc2"), (8192, "v2", "canonical")])
def test_split_tail_remainder_on_its_own_plan(built, nA, kernel, order):
    """A batch a little above the one that fills the machine on the bidirectional chain (4096 + r problems, r <= 1024): the first
    4096 problems run on the handle's chain plan, the remainder on a handle of its own with the many-part plan of a batch of r, one
    behind the other (csrc/capi.cpp, run_split; cnl_options.split_tail).  Against the two-halves form (split_tail = 0) on the same
    data: every decision (success, nfact, rho, rho_old, the rho slots) bit for bit, d to the forward tolerance (the remainder's plan
    has another elimination order); a sample of both parts against the oracle — ladder climbers in both parts, a hopeless problem
    in the remainder.  Then the call sequences that must find each part's own factor: try_to_factorize_dev + solve_dev, and the
    chunked host-pointer newton_system (which factorises EVERY problem in the first handle's storage) followed by solve_ldl!.
    Second case: the single stream (one wavefront per four problems, 8192 problems fill the machine) with 200 problems more — without
    the remainder handle they cost a second round of all fronts."""
    import torch
    hipldl, syn, O = _mods()
    s = syn.band_structure(600, 6)
    rows, cols = s.kkt_pattern()
    B = nA + 200
    v8, r8 = syn.batch_values(s, 8, cfg=3)
    rng = np.random.default_rng(11)
    vals = np.tile(v8, (B // 8, 1)) * (1.0 + 1e-3 * rng.standard_normal((B, 1)))
    rhs = np.tile(r8, (B // 8, 1)) + 1e-3 * np.arange(B)[:, None]
    off = s.offsets()
    vals[:, off[4]:off[5]] = -1.0
    hF_r, hF_c = np.asarray(s.hF[0]), np.asarray(s.hF[1])
    dg = off[0] + np.nonzero(hF_r == hF_c)[0]
    climbers = [5, nA - 6, nA, nA + 104, B - 1]
    for b in climbers:
        vals[b, dg[:50]] = -40.0        # wrong inertia at rho = 0: climbs the ladder
    hopeless = nA + 5
    vals[hopeless, off[0]:off[1]] = np.nan
    ro_in = np.zeros(B)
    ro_in[nA + 104] = 2.5e-3
    p = hipldl.default_params()
    dev = torch.device("cuda", 0)
    res = {}
    for tail in (1, 0):
        # (band_kernel = 0: the machine loads of the register-front single stream; the band kernels' own are in test_band_remainder_handle)
        L = hipldl.HIPLDLStruct(s.N, rows, cols, None, s.nvar, s.nequ, s.ncon, batch=B, options=hipldl.Options(split_tail=tail, band_kernel=0))
        assert L.config["kernel"] == kernel and L.info["order"].startswith(order) and L.config["tail"] == bool(tail)
        v = torch.from_numpy(vals).to(dev)
        r = torch.from_numpy(rhs).to(dev)
        d = torch.full((B, s.N), 7.0, dtype=torch.float64, device=dev)
        ro = torch.from_numpy(ro_in).to(dev)
        rho = torch.ones(B, dtype=torch.float64, device=dev)
        nf = torch.zeros(B, dtype=torch.int32, device=dev)
        ok = torch.zeros(B, dtype=torch.int32, device=dev)
        hipldl.newton_system_dev(L, v.data_ptr(), r.data_ptr(), d.data_ptr(), ro.data_ptr(), rho.data_ptr(), nf.data_ptr(), ok.data_ptr(), p, 0)
        torch.cuda.synchronize()
        assert L.dataflow_timeouts() == 0
        out = [ok.cpu().numpy(), nf.cpu().numpy(), rho.cpu().numpy(), ro.cpu().numpy(), v[:, -s.nvar:].cpu().numpy(), d.cpu().numpy()]
        # solve_dev right behind it: another right-hand side with the factors the call left (each part's own)
        d2 = torch.zeros((B, s.N), dtype=torch.float64, device=dev)
        r2 = r * 0.5
        hipldl._check(hipldl.lib().cnl_solve_dev(L._h, r2.data_ptr(), d2.data_ptr(), 0))
        torch.cuda.synchronize()
        out.append(d2.cpu().numpy())
        # try_to_factorize_dev (rho slots as the ladder left them) + solve_dev
        ok2 = torch.zeros(B, dtype=torch.int32, device=dev)
        hipldl._check(hipldl.lib().cnl_factorize_dev(L._h, v.data_ptr(), float(p[0]), ok2.data_ptr(), 0))
        d3 = torch.zeros((B, s.N), dtype=torch.float64, device=dev)
        hipldl._check(hipldl.lib().cnl_solve_dev(L._h, r.data_ptr(), d3.data_ptr(), 0))
        torch.cuda.synchronize()
        out += [ok2.cpu().numpy(), d3.cpu().numpy()]
        # chunked host-pointer call, then solve_ldl! with host pointers
        vh = vals.copy()
        dh = np.full((B, s.N), 7.0)
        dh, okh, rhoh, roh, nfh = hipldl.newton_system_(dh, s.nvar, s.nequ, s.ncon, rhs, vh, L, ro_in, p)
        d4 = np.zeros((B, s.N))
        hipldl.solve_ldl_(0.5 * rhs, L.factor, d4)
        out += [np.asarray(okh), np.asarray(nfh), np.asarray(rhoh), np.asarray(roh), vh[:, -s.nvar:].copy(), np.array(dh, copy=True).reshape(B, s.N), d4]
        res[tail] = out
        if tail:
            perm = L.plan_array("perm").astype(np.int64)
        L.close()
    a, b_ = res[1], res[0]
    good = a[0] == 1
    assert good.sum() == B - 1 and not good[hopeless]
    for k in (0, 1, 2, 3):
        assert np.array_equal(a[k], b_[k]), k
        assert np.array_equal(a[k], a[9 + k].astype(a[k].dtype)), k     # the host-pointer call decides the same
    assert np.array_equal(a[4], b_[4], equal_nan=True) and np.array_equal(a[4], a[13], equal_nan=True)
    assert (a[1][climbers] > 1).all() and (a[1][[0, 100, nA + 1, nA + 54]] == 1).all()
    assert np.array_equal(a[7][good], np.ones(B - 1, dtype=a[7].dtype))     # the refactorisation with the final rho succeeds
    for k in (5, 6, 8, 14, 15):     # d of newton_system_dev, solve_dev, factorize_dev + solve_dev, host newton_system, host solve
        x, y = a[k], b_[k]
        if k in (5, 14):
            assert (x[hopeless] == 7.0).all() and (y[hopeless] == 7.0).all()
        scale = np.abs(y[good]).max(axis=1, keepdims=True)
        assert (np.abs(x[good] - y[good]) <= FWD_TOL * scale).all(), k
    assert (np.abs(a[6][good] - 0.5 * a[5][good]) <= FWD_TOL * np.abs(a[5][good]).max(axis=1, keepdims=True)).all()
    assert (np.abs(a[8][good] - a[5][good]) <= FWD_TOL * np.abs(a[5][good]).max(axis=1, keepdims=True)).all()
    assert (np.abs(a[15][good] - 0.5 * a[14][good]) <= FWD_TOL * np.abs(a[14][good]).max(axis=1, keepdims=True)).all()
    orc = O.Oracle(s.N, rows, cols, perm)
    po = O.default_params()
    for b in climbers + [0, nA - 1, nA + 1, hopeless]:
        vv = vals[b].copy()
        d0, ok0, rho0, ro0, nf0 = O.newton_system(orc, s.nvar, s.nequ, s.ncon, rhs[b], vv, float(ro_in[b]), po)
        assert bool(ok0) == bool(a[0][b]) and (nf0, rho0, ro0) == (int(a[1][b]), float(a[2][b]), float(a[3][b])), b
        assert np.array_equal(a[4][b], vv[-s.nvar:], equal_nan=True)
        if ok0:
            assert np.abs(a[5][b] - d0).max() <= FWD_TOL * np.abs(d0).max()
            assert backward_error(s, vv, rhs[b], a[5][b]) <= BWD_TOL


@pytest.mark.parametrize("B", [4097, 8193, 8192 + 4096 + 3])
def test_split_tail_odd_remainders(built, B):
    """Remainders that are not multiples of four problems (a wavefront serves four): one problem behind a full load of the single
    stream (its handle runs the single-system dataflow plan), one behind the chain's 4096, and a remainder that is itself split
    (4099 problems behind 8192: the remainder handle owns a remainder handle).  Host-pointer newton_system (chunked), every decision
    and a sample of solutions against the oracle; a ladder climber sits in the last problem."""
    hipldl, syn, O = _mods()
    s = syn.band_structure(240, 4)
    rows, cols = s.kkt_pattern()
    v8, r8 = syn.batch_values(s, 8, cfg=3)
    rng = np.random.default_rng(B)
    vals = np.tile(v8, (B // 8 + 1, 1))[:B] * (1.0 + 1e-3 * rng.standard_normal((B, 1)))
    rhs = np.tile(r8, (B // 8 + 1, 1))[:B] + 1e-3 * np.arange(B)[:, None]
    off = s.offsets()
    vals[:, off[4]:off[5]] = -1.0
    hF_r, hF_c = np.asarray(s.hF[0]), np.asarray(s.hF[1])
    dg = off[0] + np.nonzero(hF_r == hF_c)[0]
    vals[B - 1, dg[:40]] = -40.0
    p = hipldl.default_params()
    L = hipldl.HIPLDLStruct(s.N, rows, cols, None, s.nvar, s.nequ, s.ncon, batch=B, options=hipldl.Options(band_kernel=0))
    if B > 8192:
        assert L.config["tail"] and L.config["kernel"] == "v2"
    v = vals.copy()
    d = np.full((B, s.N), 7.0)
    d, ok, rho, ro, nf = hipldl.newton_system_(d, s.nvar, s.nequ, s.ncon, rhs, v, L, np.zeros(B), p)
    assert ok.all() and nf[B - 1] > 1 and (nf[:B - 1] == 1).all()
    d2 = np.zeros((B, s.N))
    hipldl.solve_ldl_(2.0 * rhs, L.factor, d2)
    orc = O.Oracle(s.N, rows, cols, L.plan_array("perm").astype(np.int64))
    po = O.default_params()
    for b in sorted({0, 4095, 4096, 4097 % B, 8191 % B, 8192 % B, B - 2, B - 1}):
        vv = vals[b].copy()
        d0, ok0, rho0, ro0, nf0 = O.newton_system(orc, s.nvar, s.nequ, s.ncon, rhs[b], vv, 0.0, po)
        assert ok0 and (nf0, rho0, ro0) == (int(nf[b]), float(rho[b]), float(ro[b])), b
        assert np.array_equal(v[b, -s.nvar:], vv[-s.nvar:])
        assert np.abs(d[b] - d0).max() <= FWD_TOL * np.abs(d0).max()
        assert np.abs(d2[b] - 2.0 * d0).max() <= FWD_TOL * 2.0 * np.abs(d0).max()
    L.close()


def test_host_pointer_call_pipelined_in_chunks(built):
    """cnl_newton_system with host pointers on a batch above 96 MB runs chunk by chunk (upload of chunk c + 1, compute of chunk c
    and download of chunk c - 1 overlap; csrc/capi.cpp, newton_system_pipelined).  Every output must equal the device-resident
    call's on the same handle, bit for bit — including a problem that climbs the rho ladder (its rho slots are written back)
    and a hopeless one (its d stays as the caller left it) — and a sample is checked against the oracle."""
    import torch
    hipldl, syn, O = _mods()
    s = syn.band_structure(10000, 50)
    rows, cols = s.kkt_pattern()
    B = 100
    assert B * (s.nnzNS + s.N) * 8 >= 96 << 20
    v8, r8 = syn.batch_values(s, 8, cfg=3)
    vals, rhs = np.tile(v8, (13, 1))[:B].copy(), np.tile(r8, (13, 1))[:B].copy()
    rhs += np.arange(B)[:, None] * 1e-3
    off = s.offsets()
    vals[37, off[0]:off[1]] = np.nan              # hopeless
    hF_r, hF_c = np.asarray(s.hF[0]), np.asarray(s.hF[1])
    dg = off[0] + np.nonzero(hF_r == hF_c)[0]
    vals[70, dg[:500]] = -30.0                    # indefinite top-left block: needs rho
    p = hipldl.default_params()
    L = hipldl.HIPLDLStruct(s.N, rows, cols, None, s.nvar, s.nequ, s.ncon, batch=B)
    v = vals.copy()
    d = np.full((B, s.N), 7.0)
    d, ok, rho, ro, nf = hipldl.newton_system_(d, s.nvar, s.nequ, s.ncon, rhs, v, L, np.zeros(B), p)
    assert not ok[37] and np.all(d[37] == 7.0) and ok[70] and nf[70] > 1 and ok.sum() == B - 1
    dev = torch.device("cuda", 0)
    tv, tr = torch.from_numpy(vals.copy()).to(dev), torch.from_numpy(rhs).to(dev)
    td = torch.full((B, s.N), 7.0, dtype=torch.float64, device=dev)
    tro, trho = torch.zeros(B, dtype=torch.float64, device=dev), torch.zeros(B, dtype=torch.float64, device=dev)
    tnf, tsu = torch.zeros(B, dtype=torch.int32, device=dev), torch.zeros(B, dtype=torch.int32, device=dev)
    hipldl.newton_system_dev(L, tv.data_ptr(), tr.data_ptr(), td.data_ptr(), tro.data_ptr(), trho.data_ptr(), tnf.data_ptr(), tsu.data_ptr(), p, 0)
    torch.cuda.synchronize()
    assert np.array_equal(tsu.cpu().numpy().astype(bool), ok) and np.array_equal(tnf.cpu().numpy(), nf)
    assert np.array_equal(trho.cpu().numpy(), rho) and np.array_equal(tro.cpu().numpy(), ro)
    good = np.nonzero(ok & (nf == 1))[0]
    assert np.array_equal(td.cpu().numpy()[good], d[good])
    # the problem that climbed the ladder: on the host-pointer call the host drives the ladder (staged rungs, lean solve), the
    # device-pointer call runs the device ladder — same decisions (above), solutions equal to rounding
    for b in np.nonzero(ok & (nf > 1))[0]:
        tdb = td[b].cpu().numpy()
        assert np.abs(tdb - d[b]).max() <= 1e-11 * np.abs(d[b]).max()
    assert np.array_equal(tv.cpu().numpy()[:, -s.nvar:], v[:, -s.nvar:])
    orc = O.Oracle(s.N, rows, cols, L.plan_array("perm").astype(np.int64))
    for b in (0, 36, 38, 70, 99):
        d0, ok0, rho0, ro0, nf0 = O.newton_system(orc, s.nvar, s.nequ, s.ncon, rhs[b], vals[b].copy(), 0.0, O.default_params())
        assert (bool(ok[b]), int(nf[b]), float(rho[b]), float(ro[b])) == (ok0, nf0, rho0, ro0)
        assert np.abs(d[b] - d0).max() <= FWD_TOL * np.abs(d0).max()
    L.close()


def test_multi_device_resident_twins_share_one_analysis(built):
    """cnl_multi_*_dev: every shard's arrays already live on its device, the calls only enqueue and cnl_multi_synchronize waits —
    nothing crosses PCIe per call (the host-pointer cnl_multi_* calls are link-bound by construction).  One GPU here: the device
    is named twice.  Shards of equal size share ONE symbolic analysis; here the sizes are 5 and 4, so each gets the plan cnl_create
    would pick for its own batch (round 4: one analysis per distinct shard size)."""
    import torch
    hipldl, syn, O = _mods()
    s = syn.band_structure(300, 4)
    rows, cols = s.kkt_pattern()
    B = 9
    vals, rhs = syn.batch_values(s, B, cfg=4)
    vl, rl = syn.batch_values(s, B, cfg=5, stress="ladder")
    vals[6], rhs[6] = vl[6], rl[6]
    p = hipldl.default_params()
    M = hipldl.MultiHIPLDLStruct(s.N, rows, cols, s.nvar, s.nequ, s.ncon, B, [0, 0])
    assert [(a, c) for a, c, _ in M.shards] == [(0, 5), (5, 4)]
    dev = torch.device("cuda", 0)
    sh = []
    for a, c, _ in M.shards:
        sh.append(dict(vals=torch.from_numpy(vals[a:a + c].copy()).to(dev), rhs=torch.from_numpy(rhs[a:a + c].copy()).to(dev),
                       d=torch.zeros((c, s.N), dtype=torch.float64, device=dev), ro=torch.zeros(c, dtype=torch.float64, device=dev),
                       rho=torch.zeros(c, dtype=torch.float64, device=dev), nf=torch.zeros(c, dtype=torch.int32, device=dev),
                       su=torch.zeros(c, dtype=torch.int32, device=dev)))
    torch.cuda.synchronize()
    ptr = lambda k: [x[k].data_ptr() for x in sh]
    M.newton_system_dev(ptr("vals"), ptr("rhs"), ptr("d"), ptr("ro"), ptr("rho"), ptr("nf"), ptr("su"), p)
    M.synchronize()
    L = hipldl.HIPLDLStruct(s.N, rows, cols, None, s.nvar, s.nequ, s.ncon, batch=B)
    v1 = vals.copy()
    d1, ok1, rho1, ro1, nf1 = hipldl.newton_system_(np.zeros((B, s.N)), s.nvar, s.nequ, s.ncon, rhs, v1, L, np.zeros(B), p)
    L.close()
    dm = np.concatenate([x["d"].cpu().numpy() for x in sh])
    assert np.array_equal(np.concatenate([x["su"].cpu().numpy() for x in sh]).astype(bool), ok1)
    assert np.array_equal(np.concatenate([x["nf"].cpu().numpy() for x in sh]), nf1)
    assert np.array_equal(np.concatenate([x["rho"].cpu().numpy() for x in sh]), rho1)
    assert np.array_equal(np.concatenate([x["ro"].cpu().numpy() for x in sh]), ro1)
    assert np.array_equal(np.concatenate([x["vals"].cpu().numpy() for x in sh]), v1) and nf1[6] > 1
    for b in range(B):
        assert np.abs(dm[b] - d1.reshape(B, -1)[b]).max() <= 1e-12 * np.abs(d1).max()
    # the two-call sequence, device-resident
    M.factorize_dev(ptr("vals"), p[0], ptr("su"))
    M.solve_dev(ptr("rhs"), ptr("d"))
    M.synchronize()
    assert all(bool((x["su"] == 1).all().item()) for x in sh)   # the rho slots now hold the rho the ladder ended on
    d2 = np.concatenate([x["d"].cpu().numpy() for x in sh])
    for b in (0, 4, 6, 8):
        assert backward_error(s, v1[b], rhs[b], d2[b]) <= BWD_TOL
    M.close()


def test_multi_device_handle_shards_one_caller(built):
    """cnl_multi_*: one caller, several devices.  This box has one GPU: the device list names it three times, which
    exercises the real code path — three handles, three host threads, contiguous balanced shards of 10 problems (4, 3, 3) — and
    must reproduce the single-handle run problem by problem, including a problem that needs the rho ladder."""
    hipldl, syn, O = _mods()
    s = syn.band_structure(300, 4)
    rows, cols = s.kkt_pattern()
    B = 10
    vals, rhs = syn.batch_values(s, B, cfg=4)
    vl, rl = syn.batch_values(s, B, cfg=5, stress="ladder")
    vals[5], rhs[5] = vl[5], rl[5]
    p = hipldl.default_params()
    M = hipldl.MultiHIPLDLStruct(s.N, rows, cols, s.nvar, s.nequ, s.ncon, B, [0, 0, 0])
    assert [(a, c) for a, c, _ in M.shards] == [(0, 4), (4, 3), (7, 3)]
    vm = vals.copy()
    dm, okm, rhom, rom, nfm = M.newton_system_(np.zeros((B, s.N)), rhs, vm, np.zeros(B), p)
    L = hipldl.HIPLDLStruct(s.N, rows, cols, None, s.nvar, s.nequ, s.ncon, batch=B)
    v1 = vals.copy()
    d1, ok1, rho1, ro1, nf1 = hipldl.newton_system_(np.zeros((B, s.N)), s.nvar, s.nequ, s.ncon, rhs, v1, L, np.zeros(B), p)
    L.close()
    assert np.array_equal(okm, ok1) and np.array_equal(nfm, nf1) and np.array_equal(rhom, rho1) and np.array_equal(rom, ro1)
    assert nfm[5] > 1 and np.array_equal(vm, v1)
    for b in range(B):
        assert np.abs(dm[b] - d1.reshape(B, -1)[b]).max() <= 1e-12 * np.abs(d1).max()
    # the two-call sequence over the shards
    ok = M.try_to_factorize(vals, p[0])
    assert ok.sum() == B - 1 and not ok[5]
    d2 = np.zeros((B, s.N))
    M.solve_ldl_(rhs, d2)
    for b in (0, 4, 9):
        assert backward_error(s, vals[b], rhs[b], d2[b]) <= BWD_TOL
    M.close()
    # more devices than problems: the surplus stays idle
    M2 = hipldl.MultiHIPLDLStruct(s.N, rows, cols, s.nvar, s.nequ, s.ncon, 2, [0, 0, 0, 0])
    assert [(a, c) for a, c, _ in M2.shards] == [(0, 1), (1, 1)]
    M2.close()


def test_multi_device_shards_with_remainder_handles(built):
    """Two shards of 8200 problems each (device list [0, 0]): every shard is a full load of the single stream plus 8 problems on a
    remainder handle of its own (cnl_options.split_tail), its host thread drives both.  Results of the two-call sequence and of
    newton_system against a sample from the oracle, a ladder climber in each shard's remainder."""
    hipldl, syn, O = _mods()
    s = syn.band_structure(120, 3)
    rows, cols = s.kkt_pattern()
    B = 16400
    v8, r8 = syn.batch_values(s, 8, cfg=3)
    rng = np.random.default_rng(21)
    vals = np.tile(v8, (B // 8, 1)) * (1.0 + 1e-3 * rng.standard_normal((B, 1)))
    rhs = np.tile(r8, (B // 8, 1)) + 1e-3 * np.arange(B)[:, None]
    off = s.offsets()
    vals[:, off[4]:off[5]] = -1.0
    hF_r, hF_c = np.asarray(s.hF[0]), np.asarray(s.hF[1])
    dg = off[0] + np.nonzero(hF_r == hF_c)[0]
    climbers = [8195, 16399]
    for b in climbers:
        vals[b, dg[:30]] = -40.0
    p = hipldl.default_params()
    M = hipldl.MultiHIPLDLStruct(s.N, rows, cols, s.nvar, s.nequ, s.ncon, B, [0, 0])
    assert [(a, c) for a, c, _ in M.shards] == [(0, 8200), (8200, 8200)]
    v = vals.copy()
    d, ok, rho, ro, nf = M.newton_system_(np.full((B, s.N), 7.0), rhs, v, np.zeros(B), p)
    assert ok.all() and (nf[climbers] > 1).all() and (np.delete(nf, climbers) == 1).all()
    okf = M.try_to_factorize(v, p[0])        # the rho slots as the ladder left them: every factorisation succeeds
    assert okf.all()
    d2 = np.zeros((B, s.N))
    M.solve_ldl_(rhs, d2)
    M.close()
    orc = O.Oracle(s.N, rows, cols, O.canonical_perm(s.nvar, s.nequ, s.ncon))
    po = O.default_params()
    for b in [0, 8191, 8192, 8199, 8200, 16391, 16392] + climbers:
        vv = vals[b].copy()
        d0, ok0, rho0, ro0, nf0 = O.newton_system(orc, s.nvar, s.nequ, s.ncon, rhs[b], vv, 0.0, po)
        assert ok0 and (nf0, rho0, ro0) == (int(nf[b]), float(rho[b]), float(ro[b])), b
        assert np.array_equal(v[b, -s.nvar:], vv[-s.nvar:])
        assert np.abs(d[b] - d0).max() <= FWD_TOL * np.abs(d0).max()
        assert np.abs(d2[b] - d0).max() <= FWD_TOL * np.abs(d0).max()


@pytest.mark.parametrize("fill,wide,seed0,ncases", [(None, False, 0, 120), ("0x3f800001", False, 0, 120), ("0xffffffff", False, 300, 120), ("0xffffffff", False, 9200, 300),
                                                    ("0x3f800001", True, 9000, 200), ("0xffffffff", True, 9200, 200), ("0x7fc00000", True, 9400, 200)])
def test_randomised_parity_is_independent_of_what_earlier_kernels_left(built, fill, wide, seed0, ncases):
    """tools/fuzz_parity.py (small irregular and band structures, batches that are not multiples of four, every plan kind, ladder
    climbers; every decision bit for bit and d to 1e-8 against the oracle on an order the product had no part in) — in a process
    of its own, once as it is and six times with every launch preceded by kernels that leave a byte pattern in the LDS, the scratch
    memory and the vector registers of the device (CNL_DBG_SCRATCHFILL / CNL_DBG_LDSFILL).  Round 4: the staged instantiations with
    out-of-line front classes took decisions that depended on the scratch contents of earlier kernels — invisible to a test-suite
    whose processes start with zeroed scratch.  Round 5: the cause is found (register spills stored with EXEC = 0, profiles/HISTORY.md 4c) and
    every plan runs staged with the in-kernel ladder again; 1 260 cases, 1 140 of them with garbage fills: the range 9 200 .. 9 499
    holds the cases that faulted deterministically before the fix (9 236, 9 472), the three wide runs (600 cases) draw random settings
    of every remaining execution option (the band kernels', row f1's and the dense route's included)."""
    import subprocess, sys
    env = dict(os.environ)
    if fill:
        env.update(CNL_DBG_SCRATCHFILL="1", CNL_DBG_LDSFILL=fill)
    if wide:   # + the dense-Jacobian family, band batches up to 1023, batches just above a small staged_max_batch (split / remainder
        env.update(FUZZ_WIDE="1")   # handle), random execution switches
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_parity.py"), str(ncases), str(seed0)],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    assert f"{ncases} cases, 0 failures" in r.stdout


def test_two_host_threads_with_handles_of_their_own(built):
    """Two host threads, each with a handle of its own (different patterns: different kernels, LDS sizes and plans), calling
    newton_system! concurrently 40 times: every call's results equal the same call made alone.  (The per-function limit of dynamic
    LDS is process-wide state of the runtime; the launchers always set it to the device's limit, so that one thread cannot lower
    it under the other's launch.)"""
    import threading
    hipldl, syn, O = _mods()
    specs = [(syn.band_structure(2000, 10), 16, 4), (syn.random_structure(60, 80, 4, 0.1, seed=9), 5, 3), (syn.band_structure(300, 3), 1, 4)]
    p = hipldl.default_params()
    jobs = []
    for s, B, cfg in specs:
        rows, cols = s.kkt_pattern()
        if s.name == "band":
            vals, rhs = syn.batch_values(s, B, cfg=cfg)
        else:
            vals = np.stack([syn.random_values(s, 50 + b)[0] for b in range(B)]); rhs = np.stack([syn.random_values(s, 50 + b)[1] for b in range(B)])
        L = hipldl.HIPLDLStruct(s.N, rows, cols, None, s.nvar, s.nequ, s.ncon, batch=B)
        d = np.zeros((B, s.N))
        ref = hipldl.newton_system_(d, s.nvar, s.nequ, s.ncon, rhs, vals.copy(), L, np.zeros(B) if B > 1 else 0.0, p)
        jobs.append((s, B, L, vals, rhs, [np.array(x, copy=True) for x in ref]))
    errors = []

    def work(k):
        s, B, L, vals, rhs, ref = jobs[k]
        try:
            for it in range(40):
                d = np.zeros((B, s.N))
                out = hipldl.newton_system_(d, s.nvar, s.nequ, s.ncon, rhs, vals.copy(), L, np.zeros(B) if B > 1 else 0.0, p)
                for a, b in zip(out, ref):
                    if not np.array_equal(np.asarray(a), b):
                        errors.append((k, it))
                        return
        except Exception as e:  # noqa: BLE001
            errors.append((k, repr(e)))

    threads = [threading.Thread(target=work, args=(k,)) for k in range(len(jobs))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for job in jobs:
        job[2].close()
    assert not errors, errors


def test_failed_problems_leave_d_untouched(built):
    """newton_system! solves only after a successful factorisation (src/CaNNOLeS.jl:1049): the caller's d of a problem whose
    ladder runs out (rho > rho_max) is left as it was, through the host-pointer call."""
    hipldl, syn, O = _mods()
    s = syn.band_structure(300, 4)
    rows, cols = s.kkt_pattern()
    B = 6
    vals, rhs = syn.batch_values(s, B, cfg=4)
    off = s.offsets()
    vals[2, off[0]:off[1]] = np.nan   # no rho repairs a NaN Hessian: the ladder runs out
    p = hipldl.default_params()
    L = hipldl.HIPLDLStruct(s.N, rows, cols, None, s.nvar, s.nequ, s.ncon, batch=B)
    d = np.full((B, s.N), 7.0)
    d, ok, rho, ro, nf = hipldl.newton_system_(d, s.nvar, s.nequ, s.ncon, rhs, vals.copy(), L, np.zeros(B), p)
    L.close()
    assert list(ok) == [True, True, False, True, True, True] and rho[2] > p[6]
    assert np.array_equal(d.reshape(B, -1)[2], np.full(s.N, 7.0))
    for b in (0, 1, 3, 5):
        assert backward_error(s, vals[b], rhs[b], d.reshape(B, -1)[b]) <= BWD_TOL


def test_two_call_sequence_single_system_latency_plan(built):
    """try_to_factorize + solve_ldl! (the reference's literal sequence, src/solver_types.jl:69-98) on ONE system of the cfg3
    shape: the latency plan runs both calls stage by stage as well.  Oracle comparison + the inertia counts."""
    hipldl, syn, O = _mods()
    s = syn.band_structure(10000, 50)
    rows, cols = s.kkt_pattern()
    vals, rhs = syn.band_values(s, 3001)
    L = hipldl.HIPLDLStruct(s.N, rows, cols, vals, s.nvar, s.nequ, s.ncon)
    assert L.config["kernel"] == "v2-staged"
    ok, npos, nzer = hipldl.try_to_factorize(L, vals, s.nvar, s.nequ, s.ncon, 2.220446049250313e-16, return_inertia=True)
    assert ok and npos == s.nvar and nzer == 0
    orc = O.Oracle(s.N, rows, cols, L.plan_array("perm").astype(np.int64))
    assert orc.try_to_factorize(vals, s.nvar, s.nequ, s.ncon, 2.220446049250313e-16)
    for k in range(2):
        r = rhs * (k + 1) - k
        d = np.zeros(s.N)
        hipldl.solve_ldl_(r, L.factor, d)
        d0 = orc.solve_ldl(r)
        assert np.abs(d - d0).max() <= FWD_TOL * np.abs(d0).max()
        assert backward_error(s, vals, r, d) <= BWD_TOL
    # an indefinite Hessian: the Bool and the counts of a failed factorisation
    v2 = vals.copy()
    off = s.offsets()
    v2[off[0]:off[1]] = -5.0
    ok2, npos2, nzer2 = hipldl.try_to_factorize(L, v2, s.nvar, s.nequ, s.ncon, 2.220446049250313e-16, return_inertia=True)
    ok0, npos0, nzer0 = orc.try_to_factorize(v2, s.nvar, s.nequ, s.ncon, 2.220446049250313e-16, return_inertia=True)
    assert (ok2, npos2, nzer2) == (ok0, npos0, nzer0) and not ok2
    L.close()


@pytest.mark.parametrize("shape", [(300, 4), (300, 0), (600, 6)])
def test_f3_device_resident_lockstep_outer_loop(built, shape):
    """SURVEY 8 row f3, device-resident: B problems of a closed-form band family run the reference's outer/inner loop
    (src/CaNNOLeS.jl:418-864, line search :1054-1112) in LOCKSTEP with vals, rhs and d in HBM throughout (device_loop.py:
    cnl_prepare_newton_system_dev, cnl_residual_vectors_dev, cnl_newton_system_dev, cnl_trial_point_dev,
    cnl_cgls_multipliers_dev).  Each problem must take the decisions it takes alone: status, outer iterations, Newton
    systems, factorisations and backtracking steps equal those of the single-problem loop (outer_loop.solve, pinned to the
    reference's known answers) run with the CPU oracle as linear solver, and the solutions agree."""
    import torch
    hipldl, syn, O = _mods()
    from cannoles_jl_amd import device_loop as DL, outer_loop
    from tests.test_oracle_pinning import oracle_newton, oracle_solver
    n, p = shape
    s = syn.band_structure(n, p)
    B = 12
    fam = DL.BandQuadFamily(s, B, seed=n + p, torch=torch, device="cuda:0", curvature=1.5, start=1.0, noise=0.5)
    prm = hipldl.default_params()
    got = DL.solve_batch_device(fam, prm)
    work = 0
    for b in range(B):
        one = outer_loop.solve(fam.host_model(b), oracle_solver, oracle_newton, prm)
        assert one["status"] == "first_order" and got["status"][b] == one["status"]
        assert (got["iter"][b], got["nlinsolve"][b], got["nfact"][b], got["nbk"][b]) == (one["iter"], one["nlinsolve"], one["nfact"], one["nbk"]), b
        assert np.allclose(got["solution"][b], one["solution"], atol=1e-7, rtol=1e-7)
        if p:
            assert np.allclose(got["multipliers"][b], one["multipliers"], atol=1e-6, rtol=1e-6)
        assert abs(got["objective"][b] - one["objective"]) <= 1e-9 * max(1.0, one["objective"])
        work += one["nlinsolve"]
    assert got["nfact"].sum() > got["nlinsolve"].sum()        # the rho ladder was climbed somewhere
    assert got["steps"] < work                                 # lockstep: far fewer batched rounds than Newton systems
    # the loop's bookkeeping as four device kernels (round 4, csrc/outer_step.hip) against the framework form of rounds 2-3
    old = DL.solve_batch_device_framework(fam, prm)
    assert old["steps"] == got["steps"] and old["status"] == got["status"]
    for k in ("iter", "nlinsolve", "nfact", "nbk"):
        assert np.array_equal(old[k], got[k]), k
    assert np.allclose(old["solution"], got["solution"], atol=1e-9, rtol=1e-9)


@pytest.mark.parametrize("dataflow", [True, False])
@pytest.mark.parametrize("B", [1, 3, 7])
def test_tiny_batches_dataflow_and_per_stage_execution(built, B, dataflow):
    """Batches of at most eight problems run their latency plan with ONE launch per phase, tasks waiting on device counters for
    their children / parent (`dataflow`), larger ones with a launch per stage (forced here with cnl_options.dataflow = 0): both must
    give the oracle's (success, nfact, rho, rho_old) and solutions, with a problem that climbs the rho ladder and — for B > 1 —
    one whose ladder runs out (its d stays untouched, its rho slots are written back), through the host-pointer call (whose
    results come back through the pinned block of the handle) and through newton_system!'s factorize/solve siblings."""
    hipldl, syn, O = _mods()
    opts = None if dataflow else hipldl.Options(dataflow=0)
    s = syn.band_structure(2000, 10)
    rows, cols = s.kkt_pattern()
    vals, rhs = syn.batch_values(s, B, cfg=5, stress="ladder")
    off = s.offsets()
    if B > 1:
        vals[1, off[0]:off[1]] = np.nan   # no rho repairs it
    p = hipldl.default_params()
    L = hipldl.HIPLDLStruct(s.N, rows, cols, None, s.nvar, s.nequ, s.ncon, batch=B, options=opts)
    assert L.config["kernel"] == "v2-staged"
    v = vals.copy()
    d = np.full((B, s.N), 7.0)
    d, ok, rho, ro, nf = hipldl.newton_system_(d, s.nvar, s.nequ, s.ncon, rhs, v, L, np.zeros(B), p)
    ok, rho, ro, nf = (np.atleast_1d(a) for a in (ok, rho, ro, nf))
    orc = O.Oracle(s.N, rows, cols, L.plan_array("perm").astype(np.int64))
    v0 = vals.copy()
    d0, ok0, rho0, ro0, nf0 = O.newton_system_batch(orc, B, s.nvar, s.nequ, s.ncon, rhs, v0, np.zeros(B), p)
    assert np.array_equal(ok, ok0) and np.array_equal(nf, nf0) and np.array_equal(rho, rho0) and np.array_equal(ro, ro0)
    assert nf0.max() > 1
    assert np.array_equal(v.reshape(B, -1)[:, -s.nvar:], v0.reshape(B, -1)[:, -s.nvar:])
    d = d.reshape(B, s.N)
    for b in range(B):
        if ok0[b]:
            assert backward_error(s, v0.reshape(B, -1)[b], rhs.reshape(B, -1)[b], d[b]) <= BWD_TOL
        else:
            assert np.array_equal(d[b], np.full(s.N, 7.0))
    # factorize + solve on the same handle (staged as well): the solve uses this factorisation
    good = vals.copy()
    good[:, off[0]:off[1]] = np.nan_to_num(good[:, off[0]:off[1]], nan=1.0)
    good[:, off[6]:off[7]] = 10.0          # a rho at which every problem factorises
    okf = np.atleast_1d(hipldl.try_to_factorize(L, good if B > 1 else good[0], s.nvar, s.nequ, s.ncon, p[0]))
    assert okf.all()
    x = np.zeros((B, s.N))
    hipldl.solve_ldl_(rhs if B > 1 else rhs[0], L.factor, x if B > 1 else x[0])
    for b in range(B):
        assert backward_error(s, good[b], rhs.reshape(B, -1)[b], x[b]) <= BWD_TOL
    L.close()


@pytest.mark.parametrize("B,order", [(256, "ndc"), (3200, "ndc2")])
def test_two_call_sequence_on_large_part_plans(built, B, order):
    """try_to_factorize + solve_ldl! (src/solver_types.jl:69-98) on the plans mid-size batches get — a few large parts as
    tasks, the bidirectional chain with two — through the device entry points; a sample of problems against the oracle."""
    import torch
    hipldl, syn, O = _mods()
    import bench as BM
    s = syn.band_structure(10000, 50)
    rows, cols = s.kkt_pattern()
    vh, rh = BM.band_batch(s, 64, seed=4100)
    reps = (B + 63) // 64
    dev = torch.device("cuda", 0)
    vals = torch.from_numpy(np.tile(vh, (reps, 1))[:B]).to(dev)
    rhs = torch.from_numpy(np.tile(rh, (reps, 1))[:B]).to(dev)
    L = hipldl.HIPLDLStruct(s.N, rows, cols, None, s.nvar, s.nequ, s.ncon, batch=B)
    assert L.config["kernel"] == "v2-staged" and L.info["order"].startswith(order)
    lib = hipldl.lib()
    ok = torch.zeros(B, dtype=torch.int32, device=dev)
    d = torch.zeros((B, s.N), dtype=torch.float64, device=dev)
    hipldl._check(lib.cnl_factorize_dev(L._h, vals.data_ptr(), 2.220446049250313e-16, ok.data_ptr(), 0))
    hipldl._check(lib.cnl_solve_dev(L._h, rhs.data_ptr(), d.data_ptr(), 0))
    torch.cuda.synchronize()
    assert bool((ok == 1).all())
    orc = O.Oracle(s.N, rows, cols, L.plan_array("perm").astype(np.int64))
    for b in (0, 17, 63, B - 1):
        k = b % 64
        assert orc.try_to_factorize(vh[k], s.nvar, s.nequ, s.ncon, 2.220446049250313e-16)
        d0 = orc.solve_ldl(rh[k])
        got = d[b].cpu().numpy()
        assert np.abs(got - d0).max() <= FWD_TOL * np.abs(d0).max()
        assert backward_error(s, vh[k], rh[k], got) <= BWD_TOL
    L.close()


@pytest.mark.parametrize("B", [96, 640])
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_structures_mid_size_batches(built, seed, B):
    """Irregular sparsity at batch sizes where the latency analysis also offers its large-part orders: whatever plan and
    kernel serve it, every problem must reproduce the oracle (decisions and solutions)."""
    hipldl, syn, O = _mods()
    s = syn.random_structure(90 + 10 * seed, 130, 6 if seed % 2 else 0, 0.03, seed)
    v8, r8 = syn.batch_values(s, 8, cfg=seed, gen=syn.random_values)
    reps = (B + 7) // 8
    vals, rhs = np.tile(v8, (reps, 1))[:B].copy(), np.tile(r8, (reps, 1))[:B].copy()
    info, cfg = run_case(s, vals, rhs)
    # fronts of order 73 .. 95: too large for the register-front kernel.  Round 2 sent such batches to the general kernel
    # ("v1", correct but slow); they now run as 64 x 64 tiles on the dense machinery (MFMA trailing updates) at any batch size
    # that fits (csrc/capi.cpp; 2.1 M against 0.3 .. 0.5 M systems/s at 640 problems, tools/time_irregular.py)
    assert info["fmax"] > 64 and cfg["kernel"] == "dense"
    # the general kernel itself stays covered
    if B == 96:
        _, cfg1 = run_case(s, vals[:24], rhs[:24], options=hipldl.Options(general_dense=0))
        assert cfg1["kernel"] == "v1"


@pytest.mark.gpu
@pytest.mark.parametrize("B", [5, 300])
def test_residual_components_in_the_backward_sweep(built, B):
    """Lean plans recover the residual components d_r inside the backward sweep (csrc/plan.h, B_ROWS_FLAG: the front that owns a
    condensed row holds all its columns) instead of in a post-pass; cnl_options.rows_in_backward = 0 keeps the post-pass.
    Both against the oracle, throughput and latency plans, general residual pivots d_r (not the reference's -1), a ladder problem
    and a hopeless one whose d must stay untouched; the two executions must agree to rounding."""
    hipldl, syn, O = _mods()
    s = syn.band_structure(600, 6)
    rows, cols = s.kkt_pattern()
    vals, rhs = syn.batch_values(s, B, cfg=3)
    off = s.offsets()
    rng = np.random.default_rng(11)
    vals[:, off[4]:off[5]] = -rng.uniform(0.5, 2.0, (B, s.nequ))
    vl, rl = syn.batch_values(s, B, cfg=5, stress="ladder")
    vals[B - 2], rhs[B - 2] = vl[B - 2], rl[B - 2]
    vals[1, off[0]:off[1]] = np.nan
    for kind in (hipldl.PLAN_THROUGHPUT, hipldl.PLAN_LATENCY):
        out = {}
        for rib in (1, 0):
            opts = hipldl.Options(plan_kind=kind, rows_in_backward=rib)
            info, cfg = run_case(s, vals, rhs, options=opts, check_fwd=False)
            L = hipldl.HIPLDLStruct(s.N, rows, cols, None, s.nvar, s.nequ, s.ncon, batch=B, options=opts)
            assert bool(L.plan_array("brec")[7] & 256) == (bool(rib) and cfg["lean"])
            d = np.full((B, s.N), 7.0)
            d, ok, rho, rho_old, nfact = hipldl.newton_system_(d, s.nvar, s.nequ, s.ncon, rhs, vals.copy(), L, np.zeros(B), hipldl.default_params())
            assert not ok[1] and (d[1] == 7.0).all()
            out[rib] = d
            L.close()
        if kind == hipldl.PLAN_THROUGHPUT:
            assert cfg["lean"]
        good = [b for b in range(B) if b != 1]
        scale = np.abs(out[0][good]).max(axis=1, keepdims=True)
        assert (np.abs(out[1][good] - out[0][good]) / scale).max() <= 1e-12





@pytest.mark.parametrize("B", [1, 4, 120])   # 120: past the single pinned block of the smallest batches (separate copies)
def test_host_driven_ladder_equals_the_device_ladder(built, B):
    """The small-batch host-pointer newton_system! drives the rho ladder from the host — every rung a staged try_to_factorize
    (csrc/capi.cpp; src/CaNNOLeS.jl:1023-1047) — instead of handing failed problems to the sequential device launch
    (cnl_options.host_ladder = 0): both must return the oracle's (success, nfact, rho, rho_old) bit for bit, the rho slots of
    vals as the reference leaves them, a solution for the problems that succeed and an untouched d for the hopeless one —
    for a batch that mixes a convex problem (no ladder), ladder climbers with rho_old = 0 and rho_old > 0, and a hopeless one."""
    hipldl, syn, O = _mods()
    s = syn.band_structure(600, 6)
    rows, cols = s.kkt_pattern()
    vals, rhs = syn.batch_values(s, B, cfg=5, stress="ladder")
    off = s.offsets()
    ro_in = np.zeros(B)
    if B > 1:
        v3, r3 = syn.batch_values(s, B, cfg=3)
        vals[0], rhs[0] = v3[0], r3[0]                # convex: first factorisation succeeds
        vals[1, off[0]:off[1]] = np.nan               # no rho repairs it
        ro_in[3] = 2.5e-3                             # the ladder starts from max(rho_min, kappa_dec rho_old)
    p = hipldl.default_params()
    po = O.default_params()
    orc = O.Oracle(s.N, rows, cols, O.canonical_perm(s.nvar, s.nequ, s.ncon))
    res = {}
    # 1: host-driven ladder, 0: the sequential device launch, 2: the library's default (the in-kernel device ladder, round 4)
    for hl in (1, 0, 2):
        L = hipldl.HIPLDLStruct(s.N, rows, cols, None, s.nvar, s.nequ, s.ncon, batch=B,
                                options=None if hl == 2 else hipldl.Options(host_ladder=hl, device_ladder=0))
        v = vals.copy()
        d = np.full((B, s.N), 7.0)
        out = hipldl.newton_system_(d, s.nvar, s.nequ, s.ncon, rhs, v, L, ro_in if B > 1 else 0.0, p)
        res[hl] = tuple(np.atleast_1d(np.asarray(x)).copy() for x in out[1:]) + (np.array(d, copy=True).reshape(B, s.N), v.reshape(B, -1).copy())
        L.close()
    for other in (0, 2):
        for k in range(4):
            assert np.array_equal(res[1][k], res[other][k]), (other, k)     # success, rho, rho_old, nfact: bit for bit
        assert np.array_equal(res[1][5], res[other][5], equal_nan=True)   # vals: the rho slots as the reference leaves them
        for b in range(B):
            if res[1][0][b]:
                assert np.abs(res[1][4][b] - res[other][4][b]).max() <= 1e-9 * np.abs(res[1][4][b]).max()
            else:
                assert (res[other][4][b] == 7.0).all()
    d0, ok0, rho0, ro0, nf0 = O.newton_system_batch(orc, B, s.nvar, s.nequ, s.ncon, rhs, vals.copy(), ro_in, po)
    assert np.array_equal(res[1][0].astype(bool), np.atleast_1d(ok0)) and np.array_equal(res[1][3], np.atleast_1d(nf0))
    assert np.array_equal(res[1][1], np.atleast_1d(rho0)) and np.array_equal(res[1][2], np.atleast_1d(ro0))
    assert res[1][3].max() >= 3                              # the ladder was climbed
    for b in range(B):
        if res[1][0][b]:
            assert backward_error(s, res[1][5][b], rhs[b], res[1][4][b]) <= BWD_TOL
            assert np.abs(res[1][4][b] - res[0][4][b]).max() <= 1e-9 * np.abs(res[0][4][b]).max()
        else:
            assert (res[1][4][b] == 7.0).all() and (res[0][4][b] == 7.0).all()


def _ladder_mix(syn, s, B):
    """A batch that mixes a convex problem (no ladder), ladder climbers with rho_old = 0 and rho_old > 0 and a hopeless one."""
    vals, rhs = syn.batch_values(s, B, cfg=5, stress="ladder")
    off = s.offsets()
    ro_in = np.zeros(B)
    if B > 1:
        v3, r3 = syn.batch_values(s, B, cfg=3)
        vals[0], rhs[0] = v3[0], r3[0]                # convex: first factorisation succeeds
        vals[1, off[0]:off[1]] = np.nan               # no rho repairs it
        ro_in[3] = 2.5e-3                             # the ladder starts from max(rho_min, kappa_dec rho_old)
        for b in range(5, B, 3):                      # a third of the larger batches is convex
            vals[b], rhs[b] = v3[b], r3[b]
    return vals, rhs, ro_in


@pytest.mark.parametrize("mode", ["fused", "behind", "sequential"])
@pytest.mark.parametrize("B", [1, 4, 120, 256])
def test_device_ladder_of_the_dev_entry(built, B, mode):
    """cnl_newton_system_dev on staged handles climbs the rho ladder of src/CaNNOLeS.jl:1029-1047 INSIDE one launch in which every
    task of the elimination tree has a wavefront of its own (kernels2.hip, phase 2): `fused` — the launch makes the first attempt
    too where the batch is small enough for the dataflow execution —, `behind` — only behind the staged first attempt —, and
    `sequential` — the old one-wavefront-per-four-problems launch (cnl_options.device_ladder = 0).  All three must return the
    oracle's (success, nfact, rho, rho_old) bit for bit, leave the rho slots of vals and rho_old as the reference leaves them, a
    solution for the problems that succeed and an untouched d for the hopeless one; nothing is synchronised or read back by the
    call itself (device pointers in, device pointers out), and a second call on the same handle gives the same answers."""
    import torch
    hipldl, syn, O = _mods()
    s = syn.band_structure(600, 6)
    rows, cols = s.kkt_pattern()
    vals, rhs, ro_in = _ladder_mix(syn, s, B)
    p = hipldl.default_params()
    opts = {"fused": hipldl.Options(device_ladder_fused=1), "behind": hipldl.Options(), "sequential": hipldl.Options(device_ladder=0)}[mode]
    L = hipldl.HIPLDLStruct(s.N, rows, cols, None, s.nvar, s.nequ, s.ncon, batch=B, options=opts)
    assert L.config["kernel"] == "v2-staged"
    orc = O.Oracle(s.N, rows, cols, L.plan_array("perm").astype(np.int64))
    d0, ok0, rho0, ro0, nf0 = O.newton_system_batch(orc, B, s.nvar, s.nequ, s.ncon, rhs, vals.copy(), ro_in, O.default_params())
    ok0, rho0, ro0, nf0 = (np.atleast_1d(a) for a in (ok0, rho0, ro0, nf0))
    v_ref = vals.copy()
    O.newton_system_batch(orc, B, s.nvar, s.nequ, s.ncon, rhs, v_ref, ro_in, O.default_params())   # leaves the slots as the reference does
    assert nf0.max() >= 3
    dev = torch.device("cuda:0")
    for rep in range(2):
        t_vals = torch.tensor(vals, device=dev)
        t_rhs = torch.tensor(rhs, device=dev)
        t_d = torch.full((B, s.N), 7.0, dtype=torch.float64, device=dev)
        t_ro = torch.tensor(ro_in, device=dev)
        t_rho = torch.full((B,), -1.0, dtype=torch.float64, device=dev)
        t_nf = torch.full((B,), -1, dtype=torch.int32, device=dev)
        t_ok = torch.full((B,), -1, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        hipldl.newton_system_dev(L, t_vals.data_ptr(), t_rhs.data_ptr(), t_d.data_ptr(), t_ro.data_ptr(), t_rho.data_ptr(), t_nf.data_ptr(),
                                 t_ok.data_ptr(), p, stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        ok, rho, ro, nf = t_ok.cpu().numpy(), t_rho.cpu().numpy(), t_ro.cpu().numpy(), t_nf.cpu().numpy()
        assert np.array_equal(ok.astype(bool), ok0.astype(bool)), (mode, rep)
        assert np.array_equal(nf, nf0), (mode, rep, nf, nf0)
        assert np.array_equal(rho, rho0) and np.array_equal(ro, ro0), (mode, rep)
        v = t_vals.cpu().numpy()
        assert np.array_equal(v, v_ref, equal_nan=True)      # only the rho slots differ from the input, and as the reference leaves them
        d = t_d.cpu().numpy()
        for b in range(B):
            if ok0[b]:
                assert backward_error(s, v[b], rhs[b], d[b]) <= BWD_TOL
                assert np.abs(d[b] - d0[b]).max() <= FWD_TOL * np.abs(d0[b]).max()
            else:
                assert (d[b] == 7.0).all()
    assert L.dataflow_timeouts() == 0
    # solve_ldl! uses the factorisation the ladder ended on
    t_vals = torch.tensor(vals, device=dev)
    t_ro = torch.tensor(ro_in, device=dev)
    hipldl.newton_system_dev(L, t_vals.data_ptr(), t_rhs.data_ptr(), t_d.data_ptr(), t_ro.data_ptr(), t_rho.data_ptr(), t_nf.data_ptr(),
                             t_ok.data_ptr(), p, stream=torch.cuda.current_stream().cuda_stream)
    rhs2 = np.random.default_rng(5).standard_normal((B, s.N))
    t_rhs2 = torch.tensor(rhs2, device=dev)
    t_x = torch.zeros((B, s.N), dtype=torch.float64, device=dev)
    hipldl.lib().cnl_solve_dev(L._h, t_rhs2.data_ptr(), t_x.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    x, v = t_x.cpu().numpy(), t_vals.cpu().numpy()
    for b in range(B):
        if ok0[b]:
            assert backward_error(s, v[b], rhs2[b], x[b]) <= BWD_TOL
    L.close()


def test_device_ladder_single_cfg3_system_and_timeouts(built):
    """One system of BASELINE config 3's size that climbs to nfact = 6 through the device-pointer call (the entry the device-resident
    loops use): results as the oracle's; then the same with cnl_options.dataflow_spin_limit = 1 — practically every wait of the
    fused launch gives up, the waits are counted and the sequential launch behind it redoes the call from the caller's untouched
    inputs (rho_old and the rho slots are only COMMITTED when no wait gave up)."""
    import torch
    hipldl, syn, O = _mods()
    s = syn.band_structure(10000, 50)
    rows, cols = s.kkt_pattern()
    vals, rhs = syn.batch_values(s, 1, cfg=5, stress="ladder")
    p = hipldl.default_params()
    dev = torch.device("cuda:0")
    for spin in (0, 1):
        L = hipldl.HIPLDLStruct(s.N, rows, cols, None, s.nvar, s.nequ, s.ncon, batch=1,
                                options=hipldl.Options(dataflow_spin_limit=spin) if spin else None)
        orc = O.Oracle(s.N, rows, cols, L.plan_array("perm").astype(np.int64))
        v0 = vals.copy()
        d0, ok0, rho0, ro0, nf0 = O.newton_system_batch(orc, 1, s.nvar, s.nequ, s.ncon, rhs, v0, np.zeros(1), O.default_params())
        assert int(np.atleast_1d(nf0)[0]) == 6
        t_vals, t_rhs = torch.tensor(vals, device=dev), torch.tensor(rhs, device=dev)
        t_d = torch.zeros((1, s.N), dtype=torch.float64, device=dev)
        t_ro = torch.zeros(1, dtype=torch.float64, device=dev)
        t_rho = torch.zeros(1, dtype=torch.float64, device=dev)
        t_nf = torch.zeros(1, dtype=torch.int32, device=dev)
        t_ok = torch.zeros(1, dtype=torch.int32, device=dev)
        hipldl.newton_system_dev(L, t_vals.data_ptr(), t_rhs.data_ptr(), t_d.data_ptr(), t_ro.data_ptr(), t_rho.data_ptr(), t_nf.data_ptr(),
                                 t_ok.data_ptr(), p, stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert int(t_ok[0]) == 1 and int(t_nf[0]) == 6
        assert float(t_rho[0]) == float(np.atleast_1d(rho0)[0]) and float(t_ro[0]) == float(np.atleast_1d(ro0)[0])
        assert np.array_equal(t_vals.cpu().numpy(), v0)
        d = t_d.cpu().numpy()[0]
        assert backward_error(s, v0[0], rhs[0], d) <= BWD_TOL
        assert np.abs(d - d0[0]).max() <= FWD_TOL * np.abs(d0[0]).max()
        assert (L.dataflow_timeouts() > 0) == bool(spin)
        L.close()


@pytest.mark.parametrize("order", ["torch_first", "lib_first", "no_torch"])
def test_import_order_does_not_matter(built, order):
    """A drop-in library must not depend on import order.  A PyTorch wheel bundles its own HIP runtime; loaded beside the system
    ROCm's, the runtime that initialises second finds no device (measured on the pool: `lib_first` failed with "no ROCm-capable
    device" until round 4).  cannoles.jl_amd/hipldl.py::_one_hip_runtime makes both orders end with ONE runtime in the process;
    without torch the library runs on the runtime it is linked against.  Each order in a fresh process (no GPU call in this one
    decides anything)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "rt_probe.py"), order], capture_output=True, text=True, timeout=600)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-2000:]
    assert "\nOK" in "\n" + r.stdout, out[-2000:]
    assert "RUNTIMES 1 " in r.stdout, out[-2000:]


def test_solve_after_a_failed_factorisation(built):
    """The reference never calls solve_ldl! after a failed try_to_factorize (src/CaNNOLeS.jl:1049).  Through the C ABI a caller can:
    with one problem that is a call-sequence error (CNL_ERR_STATE, d untouched); in a batch the rows of the failed problems stay
    as the caller passed them and the others are solved (round 3 returned `NaN-free garbage` with CNL_OK)."""
    hipldl, syn, O = _mods()
    s = syn.band_structure(400, 4)
    rows, cols = s.kkt_pattern()
    p = hipldl.default_params()
    off = s.offsets()
    vals, rhs = syn.batch_values(s, 4, cfg=4)
    bad = vals.copy()
    bad[2, off[0]:off[1]] = -50.0          # H_F far from positive definite: wrong inertia at the rho slots as given
    L = hipldl.HIPLDLStruct(s.N, rows, cols, None, s.nvar, s.nequ, s.ncon, batch=4)
    ok = hipldl.try_to_factorize(L, bad, s.nvar, s.nequ, s.ncon, p[0])
    assert list(ok) == [True, True, False, True]
    d = np.full((4, s.N), 7.0)
    hipldl.solve_ldl_(rhs, L.factor, d)
    assert (d[2] == 7.0).all()
    for b in (0, 1, 3):
        assert backward_error(s, bad[b], rhs[b], d[b]) <= BWD_TOL
    L.close()
    L1 = hipldl.HIPLDLStruct(s.N, rows, cols, None, s.nvar, s.nequ, s.ncon, batch=1)
    assert not hipldl.try_to_factorize(L1, bad[2], s.nvar, s.nequ, s.ncon, p[0])
    d1 = np.full(s.N, 7.0)
    with pytest.raises(hipldl.CnlError) as ei:
        hipldl.solve_ldl_(rhs[2], L1.factor, d1)
    assert ei.value.code == 5 and (d1 == 7.0).all()
    assert hipldl.try_to_factorize(L1, vals[2], s.nvar, s.nequ, s.ncon, p[0])
    hipldl.solve_ldl_(rhs[2], L1.factor, d1)
    assert backward_error(s, vals[2], rhs[2], d1) <= BWD_TOL
    L1.close()


@pytest.mark.parametrize("B", [1, 4, 4608])
def test_duplicate_slots_are_summed_in_coo_order_on_the_device(built, B):
    """set_vals! sums the COO entries of one matrix slot in COO order (src/solver_types.jl:53-59: zero, then `+=` entry by entry).
    Here three variables without any Jacobian entry are appended to a band problem; their diagonal slot holds THREE COO entries —
    H_F (1e16), H_c (-1e16) and the rho slot (1.0) — and is its own pivot (nothing else touches the node).  Only the COO order
    gives (1e16 - 1e16) + 1 = 1; any other order gives 0 (a zero pivot: the factorisation would fail).  So success and
    d_k == -rhs_k / 1 BIT FOR BIT show the order of the device's sums (round 3 checked fixture F2's slot only through d at 1e-12).
    Staged plans (B = 1, 4) and the single-stream throughput kernel (B = 4608)."""
    hipldl, syn, O = _mods()
    base = syn.band_structure(40, 4)
    n0, extra = base.nvar, 3
    n = n0 + extra
    iso = np.arange(n0 + 1, n + 1)
    hF = (np.concatenate([base.hF[0], iso]), np.concatenate([base.hF[1], iso]))
    hc = (np.arange(1, n + 1), np.arange(1, n + 1))
    s = syn.Structure(n, base.nequ, base.ncon, hF, hc, base.jF, base.jc, name="band+isolated", meta=dict(base.meta))
    rows, cols = s.kkt_pattern()
    off, ob = s.offsets(), base.offsets()
    vals = np.zeros((B, s.nnzNS))
    rhs = np.zeros((B, s.N))
    v0, r0 = syn.batch_values(base, min(B, 8), cfg=4)
    for b in range(B):
        vb, rb = v0[b % len(v0)], r0[b % len(v0)]
        vals[b, off[0]:off[0] + base.nnzhF] = vb[ob[0]:ob[1]]
        vals[b, off[0] + base.nnzhF:off[1]] = 1e16                       # H_F on the isolated diagonal
        vals[b, off[1]:off[1] + n0] = vb[ob[1]:ob[2]]
        vals[b, off[1] + n0:off[2]] = -1e16                              # H_c (stored negated by prepare_newton_system!)
        vals[b, off[2]:off[3]] = vb[ob[2]:ob[3]]
        vals[b, off[3]:off[4]] = vb[ob[3]:ob[4]]
        vals[b, off[4]:off[5]] = -1.0
        vals[b, off[5]:off[6]] = vb[ob[5]:ob[6]]
        vals[b, off[6] + n0:off[7]] = 1.0                                # the rho slots of the isolated variables
        rhs[b, :n0] = rb[:n0]
        rhs[b, n0:n] = [3.25, -0.1 * (b + 1), 7.0]
        rhs[b, n:] = rb[n0:]
    p = hipldl.default_params()
    L = hipldl.HIPLDLStruct(s.N, rows, cols, None, s.nvar, s.nequ, s.ncon, batch=B)
    v = vals.copy()
    d, ok, rho, ro, nf = hipldl.newton_system_(np.zeros((B, s.N)), s.nvar, s.nequ, s.ncon, rhs, v, L, np.zeros(B), p)
    ok, nf = np.atleast_1d(ok), np.atleast_1d(nf)
    assert ok.all() and (nf == 1).all()
    d = d.reshape(B, s.N)
    assert np.array_equal(d[:, n0:n], -rhs[:, n0:n])          # pivot exactly 1.0: bit for bit
    orc = O.Oracle(s.N, rows, cols, L.plan_array("perm").astype(np.int64))
    for b in range(min(B, 4)):
        d0, ok0, _, _, nf0 = O.newton_system_batch(orc, 1, s.nvar, s.nequ, s.ncon, rhs[b:b + 1], vals[b:b + 1].copy(), np.zeros(1), O.default_params())
        assert bool(np.atleast_1d(ok0)[0]) and int(np.atleast_1d(nf0)[0]) == 1
        assert np.array_equal(d0.reshape(-1)[n0:n], d[b, n0:n])
        assert np.abs(d[b] - d0.reshape(-1)).max() <= FWD_TOL * np.abs(d0).max()
    # the two-call sequence sums the same way
    okf, npos, nzer = hipldl.try_to_factorize(L, vals if B > 1 else vals[0], s.nvar, s.nequ, s.ncon, p[0], return_inertia=True)
    assert np.atleast_1d(okf).all() and (np.atleast_1d(npos) == s.nvar).all() and (np.atleast_1d(nzer) == 0).all()
    L.close()


@pytest.mark.parametrize("shape,B", [((2000, 10), 4608), ((2000, 10), 5), ((600, 0), 4608)])
def test_band_form_elimination_equals_the_general_one(built, shape, B):
    """Fronts whose pivot rows are structurally zero outside two or three fixed columns and the four columns below the pivot — the
    chain fronts of band problems — run an elimination WITHOUT the other row updates (csrc/kernels2.hip, eliminate16_dpp<LATE, BNF>;
    the analysis proves the zeros symbolically, csrc/analysis.cpp).  The omitted updates multiply exact zeros, so the factor, the
    inertia and the solution must be the general elimination's (cnl_options.band_form = 0): same decisions, d equal to the last bit
    (up to the sign of an exact zero), both within the oracle's tolerances."""
    hipldl, syn, O = _mods()
    n, p = shape
    s = syn.band_structure(n, p)
    rows, cols = s.kkt_pattern()
    v8, r8 = syn.batch_values(s, 8, cfg=4)
    vals, rhs = np.tile(v8, ((B + 7) // 8, 1))[:B].copy(), np.tile(r8, ((B + 7) // 8, 1))[:B].copy()
    vals[min(3, B - 1)] = syn.band_values(s, 5003, stress="ladder")[0]     # one problem climbs the ladder
    prm = hipldl.default_params()
    out = {}
    for bf in (1, 0):
        L = hipldl.HIPLDLStruct(s.N, rows, cols, None, s.nvar, s.nequ, s.ncon, batch=B, options=hipldl.Options(band_form=bf))
        rec = L.plan_array("rec")
        from tests.support.rec_sim import R_HDR, R_RECLEN, R_FSOFF, R_FLAGS
        off = nband = nfr = 0
        while off < len(rec) and rec[off + R_RECLEN] > 0:
            H = rec[off:off + R_HDR]
            nband += int((int(H[R_FLAGS]) >> 8) == 16 and not (int(H[R_FLAGS]) & 2) and int(H[R_FSOFF]) != 0)
            nfr += 1
            off += int(H[R_RECLEN])
        assert (nband > 0.5 * nfr) if (bf and B >= 4608) else (nband == 0 or bf)
        d = np.zeros((B, s.N))
        v = vals.copy()
        d, ok, rho, ro, nf = hipldl.newton_system_(d, s.nvar, s.nequ, s.ncon, rhs, v, L, np.zeros(B), prm)
        out[bf] = (np.array(d).reshape(B, s.N).copy(), np.atleast_1d(ok).copy(), np.atleast_1d(rho).copy(), np.atleast_1d(nf).copy(), L.plan_array("perm").astype(np.int64))
        L.close()
    assert np.array_equal(out[1][1], out[0][1]) and np.array_equal(out[1][2], out[0][2]) and np.array_equal(out[1][3], out[0][3])
    assert out[1][1].all() and out[1][3].max() > 1
    assert np.array_equal(out[1][0], out[0][0])      # (-0.0 == 0.0 under array_equal)
    orc = O.Oracle(s.N, rows, cols, out[1][4])
    for b in range(min(B, 6)):
        d0, ok0, rho0, ro0, nf0 = O.newton_system(orc, s.nvar, s.nequ, s.ncon, rhs[b], vals[b].copy(), 0.0, O.default_params())
        assert ok0 and nf0 == int(out[1][3][b]) and rho0 == float(out[1][2][b])
        assert np.abs(out[1][0][b] - d0).max() <= FWD_TOL * np.abs(d0).max()


def test_bench_multi_front_end(built):
    """bench.py --multi: the whole job from ONE process through cnl_multi_newton_system_dev + cnl_multi_synchronize (the
    second front end).  A small instance of the same driver: two shards on the one device, every problem must succeed and the rate
    must be a number."""
    import torch
    hipldl, syn, O = _mods()
    import bench as BM
    s = syn.band_structure(300, 4, name="cfg3")
    rows, cols = s.kkt_pattern()
    dt, nprob, ok, shards = BM.multi_front_end(torch, hipldl, s, rows, cols, [0, 0], 20, steps=3, warmup=1)
    assert ok and nprob == 40 and dt > 0
    assert [(a, c) for a, c, _ in shards] == [(0, 20), (20, 20)]


# ---- round 5: band kernels (csrc/band.h, band.hip): newton_system! of throughput handles whose pattern is a band ----------------------

def _band_opts(hipldl, **kw):
    return hipldl.Options(plan_kind=hipldl.PLAN_THROUGHPUT, **kw)


@pytest.mark.parametrize("n,p,B,nl,hw", [(200, 4, 5, 16, 2), (96, 2, 3, 16, 2), (1000, 10, 37, 8, 2), (1000, 10, 37, 32, 2), (1000, 10, 70, 16, 2), (360, 6, 19, 16, 1),
                                         (400, 0, 9, 16, 2), (10000, 50, 33, 16, 2), (24, 2, 7, 16, 2), (40, 2, 7, 16, 2)])
def test_band_kernels_against_the_oracle(built, n, p, B, nl, hw):
    """one lane per (problem, half of the chain): decisions identical to the oracle's on the product's order and on the canonical one,
    d within the forward / backward bar; batches that are no multiple of the workgroup's problems, one part (n < 80) and two"""
    hipldl, syn, O = _mods()
    s = syn.band_structure(n, p, hw=hw)
    vals, rhs = syn.batch_values(s, B, cfg=4)
    info, cfg = run_case(s, vals, rhs, options=_band_opts(hipldl, band_problems_per_group=nl))
    assert cfg["band"]


def test_band_kernels_ladder_hopeless_and_rho_old(built):
    """problems that climb the rho ladder (nfact = 6, fixture F3's rule), one that no rho rescues (d untouched, rho_old kept), a
    start from rho_old > 0 (src/CaNNOLeS.jl:1030,1036) — mixed with convex problems in one workgroup"""
    hipldl, syn, O = _mods()
    s = syn.band_structure(600, 6)
    B = 21
    vals, rhs = syn.batch_values(s, B, cfg=4)
    for b in (1, 5, 17, 20):
        vals[b], rhs[b] = syn.band_values(s, 5000 + b, stress="ladder")
    vals[9, s.offsets()[0]] = -1e300
    ro = np.zeros(B)
    ro[5] = 0.3
    ro[2] = 1e-3
    info, cfg = run_case(s, vals, rhs, rho_old=ro, options=_band_opts(hipldl))
    assert cfg["band"]


def test_band_and_register_front_kernels_agree(built):
    """the same batch through cnl_options.band_kernel = 0 and 1: identical decisions, d equal to rounding (the two kernels sum in
    different orders: DESIGN section 5)"""
    hipldl, syn, O = _mods()
    s = syn.band_structure(2000, 10)
    B = 48
    vals, rhs = syn.batch_values(s, B, cfg=4)
    vals[7], rhs[7] = syn.band_values(s, 77, stress="ladder")
    rows, cols = s.kkt_pattern()
    p = hipldl.default_params()
    out = {}
    for bk in (0, 1):
        L = hipldl.HIPLDLStruct(s.N, rows, cols, None, s.nvar, s.nequ, s.ncon, batch=B, options=_band_opts(hipldl, band_kernel=bk))
        assert L.config["band"] == bool(bk)
        v = vals.copy()
        d = np.zeros((B, s.N))
        out[bk] = [np.array(x, copy=True) for x in hipldl.newton_system_(d, s.nvar, s.nequ, s.ncon, rhs, v, L, np.zeros(B), p)] + [v]
        L.close()
    for a, b in zip(out[0][1:], out[1][1:]):
        assert np.array_equal(a, b)
    assert np.abs(out[0][0] - out[1][0]).max() <= 1e-12 * np.abs(out[0][0]).max()


def test_band_newton_system_then_solve_ldl(built):
    """The reference's own two-call sequence on a band handle (try_to_factorize then solve_ldl!, src/CaNNOLeS.jl:1023,1049,
    src/solver_types.jl:69-98) and solve_ldl! behind newton_system!: since round 6 every call runs on the band kernels —
    try_to_factorize is the forward sweep (inertia counts equal to the oracle's), solve_ldl! factorises the values of the last
    factorisation again and sweeps the new right-hand side in one launch.  No register-front launch happens (cnl_launch_counts), and a
    solve with the right-hand side of the newton_system! call reproduces its d bit for bit (the same arithmetic on the same values,
    the rho slots holding what the ladder left)."""
    hipldl, syn, O = _mods()
    s = syn.band_structure(800, 8)
    B = 20
    vals, rhs = syn.batch_values(s, B, cfg=4)
    vals[3], rhs[3] = syn.band_values(s, 33, stress="ladder")
    rows, cols = s.kkt_pattern()
    p = hipldl.default_params()
    L = hipldl.HIPLDLStruct(s.N, rows, cols, None, s.nvar, s.nequ, s.ncon, batch=B, options=_band_opts(hipldl))
    assert L.config["band"]
    c0 = hipldl.launch_counts()
    ok, npos, nzer = hipldl.try_to_factorize(L, vals, s.nvar, s.nequ, s.ncon, p[0], return_inertia=True)
    assert ok.sum() == B - 1 and not ok[3]
    orc = O.Oracle(s.N, rows, cols, O.canonical_perm(s.nvar, s.nequ, s.ncon))
    for b in range(B):
        ok0, np0, nz0 = orc.try_to_factorize(vals[b], s.nvar, s.nequ, s.ncon, O.default_params()[0], return_inertia=True)
        assert (bool(ok[b]), int(npos[b]), int(nzer[b])) == (ok0, np0, nz0), b
    # the two-call sequence of the problems that hold a factor: d = -K^-1 rhs
    d1 = np.full((B, s.N), 7.0)
    assert hipldl.solve_ldl_(rhs, L.factor, d1) is True
    assert np.all(d1[3] == 7.0)   # the failed problem's rows stay as the caller passed them
    for b in (0, 4, 19):
        assert backward_error(s, vals[b], rhs[b], d1[b]) <= BWD_TOL
    v = vals.copy()
    d = np.zeros((B, s.N))
    d, ok2, rho, ro, nf = hipldl.newton_system_(d, s.nvar, s.nequ, s.ncon, rhs, v, L, np.zeros(B), p)
    assert ok2.all() and nf[3] == 6
    d = np.array(d, copy=True).reshape(B, s.N)
    assert np.array_equal(d1[[0, 4, 19]], d[[0, 4, 19]])   # two calls == the fused call where nothing climbed
    d2 = np.zeros((B, s.N))
    assert hipldl.solve_ldl_(rhs, L.factor, d2) is True
    assert np.array_equal(d2, d)
    rhs3 = np.random.default_rng(5).standard_normal(rhs.shape)
    d3 = np.zeros((B, s.N))
    hipldl.solve_ldl_(rhs3, L.factor, d3)
    for b in (0, 3, 19):
        assert backward_error(s, v[b], rhs3[b], d3[b]) <= BWD_TOL
    c1 = hipldl.launch_counts()
    assert c1["band"] - c0["band"] == 5 and c1["register_front"] == c0["register_front"] and c1["general"] == c0["general"]
    L.close()


def test_band_two_call_sequence_device_pointers(built):
    """cnl_factorize_dev -> cnl_solve_dev on a band handle above 8192 problems (band_newton_kernel<32>, interleaved factor records):
    success flags as the fused call reports them, d bit-equal to cnl_newton_system_dev's where nothing climbed; the `_dev` twins keep
    the caller's d_vals pointer for the solve (include/cannoles_hip.h)."""
    import torch
    hipldl, syn, O = _mods()
    s = syn.band_structure(240, 4)
    B = 8192 + 40
    rows, cols = s.kkt_pattern()
    v8, r8 = syn.batch_values(s, 8, cfg=3)
    rng = np.random.default_rng(2)
    vals = np.tile(v8, (B // 8 + 1, 1))[:B] * (1.0 + 1e-3 * rng.standard_normal((B, 1)))
    rhs = np.tile(r8, (B // 8 + 1, 1))[:B] + 1e-3 * np.arange(B)[:, None]
    off = s.offsets()
    vals[:, off[4]:off[5]] = -1.0
    hF_r, hF_c = np.asarray(s.hF[0]), np.asarray(s.hF[1])
    dg = off[0] + np.nonzero(hF_r == hF_c)[0]
    vals[B - 3, dg[:40]] = -40.0   # fails at rho = 0
    dev = torch.device("cuda", 0)
    tv, tr = torch.from_numpy(vals).to(dev), torch.from_numpy(rhs).to(dev)
    td = torch.full((B, s.N), 7.0, dtype=torch.float64, device=dev)
    su = torch.zeros(B, dtype=torch.int32, device=dev)
    L = hipldl.HIPLDLStruct(s.N, rows, cols, None, s.nvar, s.nequ, s.ncon, batch=B)
    assert L.config["band"] and L.config["band_nl"] == 32
    p = hipldl.default_params()
    c0 = hipldl.launch_counts()
    hipldl._check(hipldl.lib().cnl_factorize_dev(L._h, tv.data_ptr(), float(p[0]), su.data_ptr(), 0))
    hipldl._check(hipldl.lib().cnl_solve_dev(L._h, tr.data_ptr(), td.data_ptr(), 0))
    torch.cuda.synchronize()
    suh = su.cpu().numpy()
    assert suh.sum() == B - 1 and suh[B - 3] == 0
    d_two = td.cpu().numpy()
    assert np.all(d_two[B - 3] == 7.0)
    td2 = torch.zeros((B, s.N), dtype=torch.float64, device=dev)
    ro, rho = torch.zeros(B, dtype=torch.float64, device=dev), torch.zeros(B, dtype=torch.float64, device=dev)
    nf, ok = torch.zeros(B, dtype=torch.int32, device=dev), torch.zeros(B, dtype=torch.int32, device=dev)
    hipldl.newton_system_dev(L, tv.data_ptr(), tr.data_ptr(), td2.data_ptr(), ro.data_ptr(), rho.data_ptr(), nf.data_ptr(), ok.data_ptr(), p, 0)
    torch.cuda.synchronize()
    d_fused = td2.cpu().numpy()
    keep = np.arange(B) != B - 3
    assert np.array_equal(d_two[keep], d_fused[keep])
    assert bool((ok == 1).all()) and int(nf[B - 3]) > 1
    c1 = hipldl.launch_counts()
    assert c1["band"] - c0["band"] == 3 and c1["register_front"] == c0["register_front"]
    orc = O.Oracle(s.N, rows, cols, O.canonical_perm(s.nvar, s.nequ, s.ncon))
    for b in (0, 31, 32, 8191, 8192, B - 1):
        d0, ok0, _, _, _ = O.newton_system(orc, s.nvar, s.nequ, s.ncon, rhs[b], vals[b].copy(), 0.0, O.default_params())
        assert ok0 and np.abs(d_two[b] - d0).max() <= FWD_TOL * np.abs(d0).max()
    L.close()


@pytest.mark.parametrize("B", [16384 + 37, 8192 + 5])
def test_band_remainder_handle(built, B):
    """band kernels: 512 workgroups of 32 problems are resident at once (16 384 problems); a batch of one such load + r problems gives
    the remainder a handle of its own (csrc/capi.cpp, run_split), a batch below it runs in one launch.  Decisions and a sample of
    solutions against the oracle; a ladder climber sits in the last problem (in the remainder), solve_ldl! behind it finds each
    part's factor."""
    hipldl, syn, O = _mods()
    s = syn.band_structure(240, 4)
    rows, cols = s.kkt_pattern()
    v8, r8 = syn.batch_values(s, 8, cfg=3)
    rng = np.random.default_rng(B)
    vals = np.tile(v8, (B // 8 + 1, 1))[:B] * (1.0 + 1e-3 * rng.standard_normal((B, 1)))
    rhs = np.tile(r8, (B // 8 + 1, 1))[:B] + 1e-3 * np.arange(B)[:, None]
    off = s.offsets()
    vals[:, off[4]:off[5]] = -1.0
    hF_r, hF_c = np.asarray(s.hF[0]), np.asarray(s.hF[1])
    dg = off[0] + np.nonzero(hF_r == hF_c)[0]
    vals[B - 1, dg[:40]] = -40.0
    p = hipldl.default_params()
    L = hipldl.HIPLDLStruct(s.N, rows, cols, None, s.nvar, s.nequ, s.ncon, batch=B)
    assert L.config["band"] and L.config["tail"] == (B > 16384)
    v = vals.copy()
    d = np.full((B, s.N), 7.0)
    d, ok, rho, ro, nf = hipldl.newton_system_(d, s.nvar, s.nequ, s.ncon, rhs, v, L, np.zeros(B), p)
    assert ok.all() and nf[B - 1] > 1 and (nf[:B - 1] == 1).all()
    d2 = np.zeros((B, s.N))
    hipldl.solve_ldl_(2.0 * rhs, L.factor, d2)
    orc = O.Oracle(s.N, rows, cols, O.canonical_perm(s.nvar, s.nequ, s.ncon))
    po = O.default_params()
    for b in sorted({0, 31, 32, 8191, 8192, 16383 % B, 16384 % B, B - 2, B - 1}):
        vv = vals[b].copy()
        d0, ok0, rho0, ro0, nf0 = O.newton_system(orc, s.nvar, s.nequ, s.ncon, rhs[b], vv, 0.0, po)
        assert ok0 and (nf0, rho0, ro0) == (int(nf[b]), float(rho[b]), float(ro[b])), b
        assert np.array_equal(v[b, -s.nvar:], vv[-s.nvar:])
        assert np.abs(d[b] - d0).max() <= FWD_TOL * np.abs(d0).max()
        assert np.abs(d2[b] - 2.0 * d0).max() <= FWD_TOL * 2.0 * np.abs(d0).max()
    L.close()


# ---- cnl_options.batch_layout = CNL_LAYOUT_INTERLEAVED (round 6: `vals` interleaved over groups of 32 problems, include/cannoles_hip.h) ----
def _il_handles(hipldl, s, B, nl=0, **kw):
    rows, cols = s.kkt_pattern()
    mk = lambda **o: hipldl.HIPLDLStruct(s.N, rows, cols, None, s.nvar, s.nequ, s.ncon, batch=B, options=_band_opts(hipldl, **o, **kw))
    LA, LB = mk(band_problems_per_group=32), mk(batch_layout=hipldl.LAYOUT_INTERLEAVED, band_problems_per_group=nl)
    assert LA.config["band"] and LA.config["band_nl"] == 32 and LA.config["batch_layout"] == 0
    # (the layout's groups are 32 problems whatever the workgroups hold: 16 per workgroup up to 8192 problems, 32 above)
    assert LB.config["band"] and LB.config["band_nl"] == (nl or (32 if B > 8192 else 16)) and LB.config["batch_layout"] == 1
    return LA, LB


@pytest.mark.parametrize("n,p,B,hw,nl", [(200, 4, 5, 2, 0), (1000, 10, 37, 2, 0), (360, 6, 70, 1, 8), (400, 0, 33, 2, 32), (10000, 50, 64, 2, 0), (240, 4, 8192 + 40, 2, 0),
                                          (600, 6, 77, 2, 16)])
def test_interleaved_vals_layout(built, n, p, B, hw, nl):
    """The band kernels on `vals` interleaved over groups of 32 problems (CNL_LAYOUT_INTERLEAVED) against the oracle, and bit-equal in
    every output — d, flags, rho, rho_old, nfact, the rho slots written back into vals, try_to_factorize -> solve_ldl! — to the same
    kernels on the reference's problem-major arrays; batches that are no multiple of 32, ladder climbers, a hopeless problem, rho_old > 0."""
    import torch
    hipldl, syn, O = _mods()
    s = syn.band_structure(n, p, hw=hw)
    if B > 1000:
        v8, r8 = syn.batch_values(s, 8, cfg=3)
        rng = np.random.default_rng(5)
        vals = np.tile(v8, (B // 8 + 1, 1))[:B] * (1.0 + 1e-3 * rng.standard_normal((B, 1)))
        rhs = np.tile(r8, (B // 8 + 1, 1))[:B] + 1e-3 * np.arange(B)[:, None]
        off = s.offsets()
        vals[:, off[4]:off[5]] = -1.0
    else:
        vals, rhs = syn.batch_values(s, B, cfg=4)
    climbers = [b for b in (1, 4, 17, 36, 69, 8200) if b < B]
    for b in climbers:
        vals[b], rhs[b] = syn.band_values(s, 5000 + b, stress="ladder")
    hopeless = 3 if B > 3 else None
    if hopeless is not None:
        vals[hopeless, s.offsets()[0]] = -1e300
    ro_h = np.zeros(B)
    ro_h[0] = 1e-3
    if B > 4:
        ro_h[4] = 0.3
    LA, LB = _il_handles(hipldl, s, B, nl)
    dev = torch.device("cuda", 0)
    p_ = hipldl.default_params()
    tr = torch.from_numpy(rhs).to(dev)
    out = {}
    for name, L in (("pm", LA), ("il", LB)):
        tv = torch.from_numpy(vals).to(dev)
        if name == "il":
            assert hipldl.layout_len(L, 0) == hipldl.il_len(B, s.nnzNS) and hipldl.layout_len(L, 1) == hipldl.il_len(B, s.N)
            tvi = torch.full((hipldl.layout_len(L, 0),), 9.0, dtype=torch.float64, device=dev)
            hipldl.interleave_dev(L, 0, tv.data_ptr(), tvi.data_ptr(), 0)
            tin = tvi
        else:
            tin = tv
        su = torch.zeros(B, dtype=torch.int32, device=dev)
        td2 = torch.full((B, s.N), 7.0, dtype=torch.float64, device=dev)
        hipldl._check(hipldl.lib().cnl_factorize_dev(L._h, tin.data_ptr(), float(p_[0]), su.data_ptr(), 0))
        hipldl._check(hipldl.lib().cnl_solve_dev(L._h, tr.data_ptr(), td2.data_ptr(), 0))
        td = torch.full((B, s.N), 3.0, dtype=torch.float64, device=dev)
        ro, rho = torch.from_numpy(ro_h).to(dev), torch.zeros(B, dtype=torch.float64, device=dev)
        nf, ok = torch.zeros(B, dtype=torch.int32, device=dev), torch.zeros(B, dtype=torch.int32, device=dev)
        hipldl.newton_system_dev(L, tin.data_ptr(), tr.data_ptr(), td.data_ptr(), ro.data_ptr(), rho.data_ptr(), nf.data_ptr(), ok.data_ptr(), p_, 0)
        if name == "il":
            hipldl.deinterleave_dev(L, 0, tvi.data_ptr(), tv.data_ptr(), 0)
        torch.cuda.synchronize()
        out[name] = [x.cpu().numpy() for x in (su, td2, td, ro, rho, nf, ok, tv)]
        L.close()
    for a, b in zip(out["pm"], out["il"]):
        assert np.array_equal(a, b)
    su, d_two, d, ro_o, rho, nf, ok, v_after = out["il"]
    if hopeless is not None:
        assert ok[hopeless] == 0 and np.all(d[hopeless] == 3.0)
    assert all(nf[b] > 1 for b in climbers)
    # ... and against the oracle (decisions, the rho slots, d) on a sample that holds every special problem
    rows, cols = s.kkt_pattern()
    orc = O.Oracle(s.N, rows, cols, O.canonical_perm(s.nvar, s.nequ, s.ncon))
    sample = sorted(set([0, B - 1, B // 2] + climbers + ([hopeless] if hopeless is not None else []) + [b for b in (4, 31, 32, 63) if b < B]))
    for b in sample:
        v0 = vals[b].copy()
        d0, ok0, rho0, ro0, nf0 = O.newton_system(orc, s.nvar, s.nequ, s.ncon, rhs[b], v0, float(ro_h[b]), O.default_params())
        assert bool(ok[b]) == bool(ok0) and nf[b] == nf0 and rho[b] == rho0 and ro_o[b] == ro0
        assert np.array_equal(v_after[b, -s.nvar:], v0[-s.nvar:])
        if ok0:
            assert np.abs(d[b] - d0).max() <= FWD_TOL * np.abs(d0).max()


@pytest.mark.parametrize("B", [1, 33, 96])
def test_interleave_conversions_and_index(built, B):
    """cnl_interleave_dev / cnl_deinterleave_dev against the index function of the header (hipldl.il_index): every element where the
    formula puts it, pads zero, round trip exact; vals (which = 0) and N-vectors (which = 1)"""
    import torch
    hipldl, syn, O = _mods()
    s = syn.band_structure(150, 3)
    rows, cols = s.kkt_pattern()
    L = hipldl.HIPLDLStruct(s.N, rows, cols, None, s.nvar, s.nequ, s.ncon, batch=B, options=_band_opts(hipldl, batch_layout=1))
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(B)
    for which, length in ((0, s.nnzNS), (1, s.N)):
        a = rng.standard_normal((B, length))
        ta = torch.from_numpy(a).to(dev)
        n = hipldl.layout_len(L, which)
        ti = torch.full((n,), np.nan, dtype=torch.float64, device=dev)
        hipldl.interleave_dev(L, which, ta.data_ptr(), ti.data_ptr(), 0)
        tb = torch.zeros_like(ta)
        hipldl.deinterleave_dev(L, which, ti.data_ptr(), tb.data_ptr(), 0)
        torch.cuda.synchronize()
        got = ti.cpu().numpy()
        want = np.zeros(n)
        pp, ee = np.meshgrid(np.arange(B), np.arange(length), indexing="ij")
        want[hipldl.il_index(pp, ee, length)] = a
        assert np.array_equal(got, want)
        assert np.array_equal(tb.cpu().numpy(), a)
    with pytest.raises(hipldl.CnlError):
        hipldl.layout_len(L, 2)
    L.close()


@pytest.mark.parametrize("gn", [False, True])
def test_prepare_newton_system_dev_interleaved(built, gn):
    """row f2 writing `vals` interleaved: bit-equal to the problem-major prepare (which the oracle checks above) after cnl_deinterleave_dev,
    including what prepare_newton_system! leaves alone (the -I segment, H_F of the Gauss-Newton variants) and the pads"""
    import torch
    hipldl, syn, O = _mods()
    s = syn.band_structure(400, 4)
    B = 37
    LA, LB = _il_handles(hipldl, s, B)
    rng = np.random.default_rng(3)
    nhF, nhc, njF, njc = len(s.hF[0]), len(s.hc[0]), s.nnzjF, s.nnzjc
    hF, hc = rng.standard_normal((B, nhF)), rng.standard_normal((B, nhc))
    hc[:, 0] = 0.0  # -> -0.0
    Jx, Jcx, delta = rng.standard_normal((B, njF)), rng.standard_normal((B, njc)), rng.uniform(0.01, 1.0, B)
    vals0 = rng.standard_normal((B, s.nnzNS))
    dev = torch.device("cuda", 0)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    thF, thc, tJx, tJc, tde = t(hF), t(hc), t(Jx), t(Jcx), t(delta)
    tva = t(vals0)
    hipldl.prepare_newton_system_dev(LA, nhF, nhc, njF, njc, 0 if gn else thF.data_ptr(), thc.data_ptr(), tJx.data_ptr(), tJc.data_ptr(),
                                     tde.data_ptr(), tva.data_ptr(), 0)
    tvi = torch.full((hipldl.layout_len(LB, 0),), 5.0, dtype=torch.float64, device=dev)
    hipldl.interleave_dev(LB, 0, t(vals0).data_ptr(), tvi.data_ptr(), 0)
    before = tvi.clone()
    hipldl.prepare_newton_system_dev(LB, nhF, nhc, njF, njc, 0 if gn else thF.data_ptr(), thc.data_ptr(), tJx.data_ptr(), tJc.data_ptr(),
                                     tde.data_ptr(), tvi.data_ptr(), 0)
    tvb = torch.zeros_like(tva)
    hipldl.deinterleave_dev(LB, 0, tvi.data_ptr(), tvb.data_ptr(), 0)
    torch.cuda.synchronize()
    a, b = tva.cpu().numpy(), tvb.cpu().numpy()
    assert np.array_equal(a, b) and np.array_equal(np.signbit(a), np.signbit(b))
    # nothing outside the problems' elements was touched (pads, the spare blocks, the rows of the last group beyond the batch)
    mask = np.ones(tvi.numel(), bool)
    pp, ee = np.meshgrid(np.arange(B), np.arange(s.nnzNS), indexing="ij")
    mask[hipldl.il_index(pp, ee, s.nnzNS)] = False
    assert np.array_equal(tvi.cpu().numpy()[mask], before.cpu().numpy()[mask])
    LA.close()
    LB.close()


def test_interleaved_layout_is_refused_where_it_is_not_served(built):
    """CNL_LAYOUT_INTERLEAVED is the band kernels': cnl_create refuses other handles; on such a handle the host-pointer entry points and the
    rows that read problem-major vals (f1, f4) fail loudly instead of misreading the array"""
    import torch
    hipldl, syn, O = _mods()
    s = syn.random_structure(30, 40, 3, 0.2, 1)
    rows, cols = s.kkt_pattern()
    with pytest.raises(hipldl.CnlError):
        hipldl.HIPLDLStruct(s.N, rows, cols, None, s.nvar, s.nequ, s.ncon, batch=64, options=hipldl.Options(batch_layout=1))
    s = syn.band_structure(200, 4)
    rows, cols = s.kkt_pattern()
    with pytest.raises(hipldl.CnlError):
        hipldl.HIPLDLStruct(s.N, rows, cols, None, s.nvar, s.nequ, s.ncon, batch=64, options=_band_opts(hipldl, batch_layout=1, band_kernel=0))
    with pytest.raises(hipldl.CnlError):
        hipldl.HIPLDLStruct(s.N, rows, cols, None, s.nvar, s.nequ, s.ncon, batch=64, options=_band_opts(hipldl, batch_layout=2))
    B = 6
    L = hipldl.HIPLDLStruct(s.N, rows, cols, None, s.nvar, s.nequ, s.ncon, batch=B, options=_band_opts(hipldl, batch_layout=1))
    vals, rhs = syn.batch_values(s, B, cfg=4)
    with pytest.raises(hipldl.CnlError):
        hipldl.newton_system_(np.zeros((B, s.N)), s.nvar, s.nequ, s.ncon, rhs, vals.copy(), L, np.zeros(B), hipldl.default_params())
    with pytest.raises(hipldl.CnlError):
        hipldl.try_to_factorize(L, vals, s.nvar, s.nequ, s.ncon, 1e-10)
    dev = torch.device("cuda", 0)
    z = torch.zeros(B * max(s.nnzNS, s.N), dtype=torch.float64, device=dev)
    with pytest.raises(hipldl.CnlError):
        hipldl._check(hipldl.lib().cnl_residual_vectors_dev(L._h, z.data_ptr(), z.data_ptr(), z.data_ptr(), z.data_ptr(), z.data_ptr(), z.data_ptr(), z.data_ptr(), 0))
    # ... while the twins that read the model's arrays serve it
    hipldl.residual_vectors_jac_dev(L, s.nnzjF, s.nnzjc, z.data_ptr(), z.data_ptr(), z.data_ptr(), z.data_ptr(), z.data_ptr(), z.data_ptr(), z.data_ptr(), z.data_ptr(), 0)
    torch.cuda.synchronize()
    L.close()


def test_multi_device_shards_with_interleaved_vals(built):
    """cnl_multi_*_dev with cnl_options.batch_layout = 1: every shard's `vals` is an interleaved array of ITS problems (groups of 32 start
    at the shard's first problem); decisions and d as the single problem-major handle gives them, the rho slots written back in place"""
    import torch
    hipldl, syn, O = _mods()
    s = syn.band_structure(300, 4)
    rows, cols = s.kkt_pattern()
    B = 75
    vals, rhs = syn.batch_values(s, B, cfg=4)
    vl, rl = syn.batch_values(s, B, cfg=5, stress="ladder")
    for b in (6, 40, 74):
        vals[b], rhs[b] = vl[b], rl[b]
    p = hipldl.default_params()
    M = hipldl.MultiHIPLDLStruct(s.N, rows, cols, s.nvar, s.nequ, s.ncon, B, [0, 0], options=_band_opts(hipldl, batch_layout=1))
    dev = torch.device("cuda", 0)
    sh = []
    for a, c, _ in M.shards:
        vi = np.zeros(hipldl.il_len(c, s.nnzNS))
        pp, ee = np.meshgrid(np.arange(c), np.arange(s.nnzNS), indexing="ij")
        idx = hipldl.il_index(pp, ee, s.nnzNS)
        vi[idx] = vals[a:a + c]
        sh.append(dict(vals=torch.from_numpy(vi).to(dev), rhs=torch.from_numpy(rhs[a:a + c].copy()).to(dev), idx=idx,
                       d=torch.zeros((c, s.N), dtype=torch.float64, device=dev), ro=torch.zeros(c, dtype=torch.float64, device=dev),
                       rho=torch.zeros(c, dtype=torch.float64, device=dev), nf=torch.zeros(c, dtype=torch.int32, device=dev),
                       su=torch.zeros(c, dtype=torch.int32, device=dev)))
    ptr = lambda k: [x[k].data_ptr() for x in sh]
    M.newton_system_dev(ptr("vals"), ptr("rhs"), ptr("d"), ptr("ro"), ptr("rho"), ptr("nf"), ptr("su"), p)
    M.synchronize()
    L = hipldl.HIPLDLStruct(s.N, rows, cols, None, s.nvar, s.nequ, s.ncon, batch=B, options=_band_opts(hipldl, band_problems_per_group=32))
    v1 = vals.copy()
    d1, ok1, rho1, ro1, nf1 = hipldl.newton_system_(np.zeros((B, s.N)), s.nvar, s.nequ, s.ncon, rhs, v1, L, np.zeros(B), p)
    L.close()
    assert np.array_equal(np.concatenate([x["su"].cpu().numpy() for x in sh]).astype(bool), ok1) and nf1[6] > 1 and nf1[74] > 1
    assert np.array_equal(np.concatenate([x["nf"].cpu().numpy() for x in sh]), nf1)
    assert np.array_equal(np.concatenate([x["rho"].cpu().numpy() for x in sh]), rho1)
    assert np.array_equal(np.concatenate([x["d"].cpu().numpy() for x in sh]), d1.reshape(B, -1))
    assert np.array_equal(np.concatenate([x["vals"].cpu().numpy()[x["idx"]] for x in sh]), v1)
    with pytest.raises(hipldl.CnlError):   # the host-pointer front end takes the reference's arrays only
        M.newton_system_(np.zeros((B, s.N)), rhs, vals.copy(), np.zeros(B), p)
    M.close()
