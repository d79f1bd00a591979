"""Oracle vs the hand-derived golden fixtures and independent numpy checks (CPU)."""
import json
import os

import numpy as np
import pytest

from oracle import oracle as O

import cannoles_jl_amd  # noqa: F401
from cannoles_jl_amd import synthetic as syn

FIX = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "fixtures.json")))


def inertia_dense(K, tol=0.0):
    w = np.linalg.eigvalsh(K)
    return int((w > tol).sum()), int((w < -tol).sum())


def test_params_defaults(params):
    """ParamCaNNOLeS(Float64), src/CaNNOLeS.jl:48-62 (values listed in SURVEY.md §8 a12)"""
    exp = [2.220446049250313e-16, 1.4901161193847656e-8, 1 / 3, 8.0, 100.0, 6.055454452393343e-6, 2.028240960365167e31,
           1.4901161193847656e-8, 1.220703125e-4]
    # bit for bit: rounds 1-2 compared with rtol 1e-15 and let cbrt(eps) (2.5 ulp from the reference's eps^(1/3)) through
    assert [float(v) for v in params] == exp


def test_pattern_builder_matches_reference_layout():
    """7 segments [H_F | H_c | J_F | J_c | -I | -dI | rI], src/CaNNOLeS.jl:276-315, on the MGH01CON structures"""
    f = FIX["F1"]
    rows, cols, vals0 = O.kkt_pattern(2, 2, 1, hF=([1], [1]), hc=([1, 2, 2], [1, 1, 2]), jF=([1, 2, 2], [1, 1, 2]), jc=([1], [1]))
    assert rows.tolist() == f["rows"] and cols.tolist() == f["cols"]
    assert vals0.tolist() == [1, 1, 1, 1, 1, 1, 1, 1, -1, -1, 1, 1, 1]  # vals .= 1, -I segment = -1 (:279, :306)
    s = syn.band_structure(40, 4)
    r2, c2 = s.kkt_pattern()
    r3, c3, _ = O.kkt_pattern(s.nvar, s.nequ, s.ncon, s.hF, s.hc, s.jF, s.jc)
    assert np.array_equal(r2, r3) and np.array_equal(c2, c3)


def test_prepare_newton_system_layout():
    f = FIX["F1"]
    vals = np.ones(13)
    vals[8:10] = -1
    O.prepare(vals, 2, 2, 1, 1, 3, 3, 1, [88.0], [0.0, 0.0, 0.0], [-1.0, 24.0, 10.0], [1.0], 0.1)
    assert np.array_equal(vals, np.array(f["vals"]))


@pytest.mark.parametrize("perm", ["natural", "canonical"])
def test_fixture_F1(params, perm):
    """natural order hits K22 = 0 (zero pivot => failure => rho ladder); r-first order succeeds at rho = 0"""
    f = FIX["F1"]
    rows, cols = np.array(f["rows"]), np.array(f["cols"])
    L = O.Oracle(5, rows, cols, None if perm == "natural" else O.canonical_perm(2, 2, 1))
    ok, npos, nzer = L.try_to_factorize(np.array(f["vals"]), 2, 2, 1, params[0], return_inertia=True)
    if perm == "natural":
        assert not ok and nzer >= 1
        d, ok2, rho, rho_old, nfact = O.newton_system(L, 2, 2, 1, np.array(f["rhs"]), np.array(f["vals"]), 0.0, params)
        assert ok2 and nfact == 2 and rho == params[5] and rho_old == rho
        assert np.allclose(d, FIX["F2"]["d"], rtol=1e-9)
    else:
        assert ok and (npos, nzer) == (2, 0)
        assert np.allclose(L.solve_ldl(np.array(f["rhs"])), f["d"], rtol=1e-14)
        d, ok2, rho, rho_old, nfact = O.newton_system(L, 2, 2, 1, np.array(f["rhs"]), np.array(f["vals"]), 0.0, params)
        assert ok2 and nfact == 1 and rho == 0.0 and rho_old == 0.0
        assert np.allclose(d, f["d"], rtol=1e-14)


def test_fixture_F2_duplicates_sum_in_coo_order():
    f = FIX["F1"]
    L = O.Oracle(5, np.array(f["rows"]), np.array(f["cols"]), O.canonical_perm(2, 2, 1))
    for mode in (0, 1):  # scatter map / per-entry binary search (what SparseMatrixCSC setindex! pays)
        L.set_mode = mode
        L.set_vals(np.array(FIX["F2"]["vals"]))
        Ap, Ai, Ax = L.csc_upper()
        assert Ax[0] == FIX["F2"]["slot11_float_sum"]  # (88 + -0) + rho0, the COO order
        assert len(Ax) == 10  # 13 COO entries, 3 duplicates merged


def test_fixture_F3_rho_ladder(params):
    f = FIX["F3"]
    L = O.Oracle(6, np.array(f["rows"]), np.array(f["cols"]), O.canonical_perm(3, 3, 0))
    vals = np.array(f["vals"])
    for rho in f["rho_tried"][:-1]:
        v = vals.copy()
        v[-3:] = rho
        assert not L.try_to_factorize(v, 3, 3, 0, params[0])
    d, ok, rho, rho_old, nfact = O.newton_system(L, 3, 3, 0, np.array(f["rhs"]), vals, 0.0, params)
    assert ok and nfact == f["nfact"] and rho == f["rho"] and rho_old == rho
    assert np.array_equal(vals[-3:], np.full(3, f["rho"]))  # the driver leaves rho in the rho slots
    # rho_old != 0 seeds the first retry with max(rhomin, rho_old/3) and grows by kappa_inc = 8
    vals = np.array(f["vals"])
    d, ok, rho, rho_old, nfact = O.newton_system(L, 3, 3, 0, np.array(f["rhs"]), vals, 3.0, params)
    assert ok and rho == 64.0 and nfact == 4 and rho_old == rho  # 0, 1 (= 3/3), 8, 64


@pytest.mark.parametrize("seed", range(4))
def test_oracle_vs_dense_numpy(params, seed):
    """independent check of the mathematical contract: K d = -rhs and inertia (n, m+p, 0)"""
    s = syn.random_structure(12 + seed, 15, 3 if seed % 2 else 0, 0.3, seed)
    vals, rhs = syn.random_values(s, seed)
    K = syn.dense_kkt(s, vals)
    rows, cols = s.kkt_pattern()
    for perm in (None, O.canonical_perm(s.nvar, s.nequ, s.ncon)):
        L = O.Oracle(s.N, rows, cols, perm)
        ok, npos, nzer = L.try_to_factorize(vals, s.nvar, s.nequ, s.ncon, params[0], return_inertia=True)
        pos, neg = inertia_dense(K)
        assert ok == (pos == s.nvar and neg == s.nequ + s.ncon)
        if ok:
            d = L.solve_ldl(rhs)
            assert np.allclose(K @ d, -rhs, atol=1e-10 * max(1, np.abs(rhs).max()))


def test_oracle_rejects_malformed():
    with pytest.raises(ValueError):
        O.Oracle(3, [1, 2, 5], [1, 1, 1])
    with pytest.raises(ValueError):
        O.Oracle(3, [1, 2, 3], [1, 2, 3], perm=[0, 0, 1])


# ---- numpy restatements of the vectors either side of the Newton system ------------------------
def test_f1_residual_vectors_oracle_matches_dense():
    """O.residual_vectors (COO-order transposed products, src/CaNNOLeS.jl:507-524) against dense J' products."""
    import cannoles_jl_amd  # noqa: F401
    from cannoles_jl_amd import synthetic as syn
    from oracle import oracle as O
    s = syn.random_structure(25, 40, 6, 0.2, seed=9)
    vals, _ = syn.random_values(s, 4)
    rows, cols = s.kkt_pattern()
    rng = np.random.default_rng(2)
    r, lam, Fx, cx = rng.standard_normal(s.nequ), rng.standard_normal(s.ncon), rng.standard_normal(s.nequ), rng.standard_normal(s.ncon)
    rhs, (nd, npr) = O.residual_vectors(rows, cols, vals, s.nvar, s.nequ, s.ncon, r, lam, Fx, cx)
    K = np.zeros((s.N, s.N))
    np.add.at(K, (rows - 1, cols - 1), vals)
    JF = K[s.nvar:s.nvar + s.nequ, :s.nvar]
    Jc = K[s.nvar + s.nequ:, :s.nvar]
    dual = JF.T @ r - Jc.T @ lam
    np.testing.assert_allclose(rhs[:s.nvar], dual, rtol=1e-13, atol=1e-13)
    assert np.array_equal(rhs[s.nvar:s.nvar + s.nequ], Fx - r)
    assert np.array_equal(rhs[s.nvar + s.nequ:], cx)
    assert nd == np.abs(rhs[:s.nvar]).max() and npr == np.abs(rhs[s.nvar:]).max()


def test_f1_trial_point_oracle():
    from oracle import oracle as O
    n, m, p = 5, 7, 3
    rng = np.random.default_rng(0)
    x, r, lam, d = rng.standard_normal(n), rng.standard_normal(m), rng.standard_normal(p), rng.standard_normal(n + m + p)
    xt, rt, lt, dl = O.trial_point(n, m, p, x, r, lam, d)
    assert np.array_equal(xt, x + d[:n]) and np.array_equal(rt, r + d[n:n + m])
    assert np.array_equal(dl, -d[n + m:]) and np.array_equal(lt, lam - d[n + m:])
    d[n + m:] *= 1e6
    _, _, lt2, dl2 = O.trial_point(n, m, p, x, r, lam, d)
    assert abs(np.linalg.norm(dl2) - 1e4) <= 1e-8
    np.testing.assert_allclose(lt2, lam + dl2)


def test_f4_cgls_oracle_matches_lstsq():
    """O.cgls_multipliers (Krylov.jl's CGLS recurrence on A = Jc', b = Jx' r; src/CaNNOLeS.jl:507-518) against lstsq."""
    import cannoles_jl_amd  # noqa: F401
    from cannoles_jl_amd import synthetic as syn
    from oracle import oracle as O
    s = syn.band_structure(300, 6)
    vals, _ = syn.band_values(s, 5)
    rows, cols = s.kkt_pattern()
    r = np.random.default_rng(1).standard_normal(s.nequ)
    lam, Jxtr, it = O.cgls_multipliers(rows, cols, vals, s.nvar, s.nequ, s.ncon, r)
    K = np.zeros((s.N, s.N))
    np.add.at(K, (rows - 1, cols - 1), vals)
    JF, Jc = K[s.nvar:s.nvar + s.nequ, :s.nvar], K[s.nvar + s.nequ:, :s.nvar]
    np.testing.assert_allclose(Jxtr, JF.T @ r, rtol=1e-12, atol=1e-12)
    ref = np.linalg.lstsq(Jc.T, JF.T @ r, rcond=None)[0]
    assert 0 < it <= s.ncon + 2
    np.testing.assert_allclose(lam, ref, rtol=1e-6, atol=1e-8)
    lam0, _, it0 = O.cgls_multipliers(rows, cols, vals, s.nvar, s.nequ, s.ncon, np.zeros(s.nequ))
    assert it0 == 0 and np.array_equal(lam0, np.ones(s.ncon))  # norm(lambda) == 0 -> ones
