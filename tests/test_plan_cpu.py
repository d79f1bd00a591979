"""Host-side analysis and C-ABI surface, no GPU: the plan is executed by the numpy interpreter
(tests/support/plan_sim.py) and compared with the oracle."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import cannoles_jl_amd  # noqa: F401
from cannoles_jl_amd import hipldl, synthetic as syn
from oracle import oracle as O
from tests.support.plan_sim import PlanSim

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def check_structure(s, vals, rhs, params, rho_old=0.0, fwd_tol=1e-9, expect=None, options=None):
    rows, cols = s.kkt_pattern()
    pl = hipldl.Plan(s.N, rows, cols, s.nvar, s.nequ, s.ncon, options=options)
    perm = pl.array("perm").astype(np.int64)
    assert sorted(perm.tolist()) == list(range(s.N))
    sim = PlanSim(pl)
    sim.check_layout()
    d, ok, rho, ro, nf = sim.newton_system(vals.copy(), rhs, s.nvar, s.nequ, s.ncon, rho_old, params)
    orc = O.Oracle(s.N, rows, cols, perm)
    d0, ok0, rho0, ro0, nf0 = O.newton_system(orc, s.nvar, s.nequ, s.ncon, rhs, vals.copy(), rho_old, params)
    assert (ok, rho, ro, nf) == (ok0, rho0, ro0, nf0)
    if expect is not None:
        assert (ok, nf) == expect
    if ok:
        assert np.abs(d - d0).max() <= fwd_tol * np.abs(d0).max()
    return pl


def test_abi_exports_every_declared_symbol(built):
    """every function declared in include/cannoles_hip.h is exported by the shared library"""
    hdr = open(os.path.join(ROOT, "include", "cannoles_hip.h")).read()
    declared = set(re.findall(r"\b(cnl_[a-z0-9_]+)\s*\(", hdr))
    lib = C.CDLL(hipldl.LIB_PATH)
    for sym in sorted(declared):
        assert hasattr(lib, sym), f"{sym} declared in cannoles_hip.h but not exported"
    assert declared >= set(hipldl.ABI_SYMBOLS)
    assert hipldl.lib().cnl_version() >= 200


def test_default_params_match_reference(built, params):
    assert np.array_equal(hipldl.default_params(), params)


def test_options_are_arguments_not_environment(built, monkeypatch):
    """cnl_options (include/cannoles_hip.h) replaces the environment switches of earlier rounds: a struct of another ABI
    revision is refused, the defaults are the choices cnl_create makes by itself, and the old environment variables no longer
    change the plan (the numerical path of a handle depends on its arguments only)."""
    o = hipldl.Options()
    assert (o.plan_kind, o.staged_max_batch, o.band_kernel, o.staged, o.dataflow, o.device_ladder, o.tuning) == (0, 4096, 1, 1, 1, 1, b"")
    assert len(hipldl.cnl_options._fields_) <= 15   # the public structure stays small; the rest: csrc/options.h via `tuning`
    s = syn.band_structure(400, 8)
    rows, cols = s.kkt_pattern()
    bad = hipldl.Options()
    bad.struct_size = 8
    with pytest.raises(hipldl.CnlError) as e:
        hipldl.Plan(s.N, rows, cols, s.nvar, s.nequ, s.ncon, batch=4, options=bad)
    assert e.value.code == 1
    base = hipldl.Plan(s.N, rows, cols, s.nvar, s.nequ, s.ncon, batch=64)
    for k, v in (("CNL_NO_EARLY", "1"), ("CNL_STAGED_MAX", "0"), ("CNL_NO_CONDENSE", "1"), ("CNL_NO_V2", "1"), ("CNL_FORCE_ORDER", "canonical"),
                 ("CNL_ORDER", "0"), ("CNL_TASK_CAP", "1")):
        monkeypatch.setenv(k, v)
    again = hipldl.Plan(s.N, rows, cols, s.nvar, s.nequ, s.ncon, batch=64)
    assert again.info == base.info and np.array_equal(again.array("rec"), base.array("rec")) and np.array_equal(again.array("tasks"), base.array("tasks"))
    forced = hipldl.Plan(s.N, rows, cols, s.nvar, s.nequ, s.ncon, batch=64, options=hipldl.Options(plan_kind=hipldl.PLAN_THROUGHPUT))
    assert len(forced.array("tasks")) == 0 and len(base.array("tasks")) > 0
    # the large-batch analysis and an explicit throughput plan are the same thing
    big = hipldl.Plan(s.N, rows, cols, s.nvar, s.nequ, s.ncon)
    assert forced.info == big.info


def test_no_device_is_a_loud_error(built):
    """there is no CPU fallback: creating a solver without a HIP device must fail"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    s = syn.band_structure(20, 2)
    rows, cols = s.kkt_pattern()
    with pytest.raises(hipldl.CnlError) as e:
        hipldl.HIPLDLStruct(s.N, rows, cols, None, s.nvar, s.nequ, s.ncon)
    assert e.value.code == 4 and "no CPU fallback" in str(e.value)


def test_pattern_errors(built):
    s = syn.band_structure(20, 2)
    rows, cols = s.kkt_pattern()
    with pytest.raises(hipldl.CnlError) as e:
        hipldl.Plan(s.N, cols, rows, s.nvar, s.nequ, s.ncon)  # upper triangle
    assert e.value.code == 3
    with pytest.raises(hipldl.CnlError) as e:
        hipldl.Plan(s.N + 1, rows, cols, s.nvar, s.nequ, s.ncon)  # N != nvar + nequ + ncon
    assert e.value.code == 2
    bad = rows.copy()
    bad[0] = s.N + 5
    with pytest.raises(hipldl.CnlError) as e:
        hipldl.Plan(s.N, bad, cols, s.nvar, s.nequ, s.ncon)
    assert e.value.code == 3


def test_fixture_mgh01con(built, params):
    import json
    f = json.load(open(os.path.join(ROOT, "tests", "golden", "fixtures.json")))["F1"]
    pl = hipldl.Plan(5, f["rows"], f["cols"], 2, 2, 1)
    d, ok, rho, ro, nf = PlanSim(pl).newton_system(np.array(f["vals"]), np.array(f["rhs"]), 2, 2, 1, 0.0, params)
    assert ok and nf == 1 and rho == 0.0
    assert np.allclose(d, f["d"], rtol=1e-13)


@pytest.mark.parametrize("seed", range(6))
def test_random_structures(built, params, seed):
    s = syn.random_structure(25 + 6 * seed, 35, 4 if seed % 2 else 0, 0.12, seed)
    vals, rhs = syn.random_values(s, seed)
    check_structure(s, vals, rhs, params)


def test_band_cfg4_and_ladder(built, params):
    s = syn.band_structure(1000, 10)
    vals, rhs = syn.band_values(s, 4000)
    pl = check_structure(s, vals, rhs, params, expect=(True, 1))
    assert pl.info["ncond"] == 1000 and pl.info["v2"] is not None
    vals, rhs = syn.band_values(s, 5000, stress="ladder")
    check_structure(s, vals, rhs, params, expect=(True, 6))
    check_structure(s, vals, rhs, params, rho_old=10.0)


def test_gauss_newton_and_dense(built, params):
    s = syn.random_structure(30, 20, 0, 0.15, 7, hess=False)   # underdetermined: needs rho > 0
    vals, rhs = syn.random_values(s, 7)
    check_structure(s, vals, rhs, params, fwd_tol=1e-6)
    s = syn.dense_structure(40, 80)
    vals, rhs = syn.dense_values(s, 2000)
    check_structure(s, vals, rhs, params)


def test_without_condensation_and_v1_only(built, params):
    """the residual-block condensation and the register-front streams are optional layers: the plain plan agrees"""
    s = syn.band_structure(300, 6)
    vals, rhs = syn.band_values(s, 4001)
    pl = check_structure(s, vals, rhs, params, options=hipldl.Options(condense=0))
    assert pl.info["ncond"] == 0
    pl = check_structure(s, vals, rhs, params, options=hipldl.Options(condense=0, register_front=0))
    assert pl.info["v2"] is None


def test_cfg3_plan_quality(built):
    """the headline pattern: fill below the canonical r/x/lambda order of SURVEY.md §8 and small fronts"""
    s = syn.band_structure(10000, 50)
    rows, cols = s.kkt_pattern()
    pl = hipldl.Plan(s.N, rows, cols, s.nvar, s.nequ, s.ncon)
    assert pl.info["nnzL_exact"] < 346209
    assert pl.info["v2"]["fronts16"] > 0.95 * pl.info["nsuper"]
    # band form (round 4): the chain fronts of the headline plan carry it — pivot rows structurally zero outside the two lowest
    # columns (right-hand side, multiplier) and the four columns below the pivot; 41 % of the row updates are not compiled in
    from tests.support.rec_sim import R_HDR, R_RECLEN, R_FSOFF, R_FLAGS, R_NPIV, R_NUPD, band_rows
    for band_form, expect in ((1, True), (0, False)):
        rec = hipldl.Plan(s.N, rows, cols, s.nvar, s.nequ, s.ncon, options=hipldl.Options(band_form=band_form)).array("rec")
        off = nband = upd_all = upd_band = 0
        while off < len(rec) and rec[off + R_RECLEN] > 0:
            H = rec[off:off + R_HDR]
            bw = int(H[R_FSOFF]) if (int(H[R_FLAGS]) >> 8) == 16 and not (int(H[R_FLAGS]) & 2) else 0
            nband += bw != 0
            npiv, nupd = int(H[R_NPIV]), int(H[R_NUPD])
            for i in range(nupd + npiv, nupd, -1):
                upd_all += i
                upd_band += len(band_rows(bw, i))
            off += int(H[R_RECLEN])
        if expect:
            assert nband >= 0.9 * pl.info["nsuper"] and upd_band <= 0.65 * upd_all, (nband, upd_band, upd_all)
        else:
            assert nband == 0 and upd_band == upd_all


# ---- record streams of the register-front kernel, interpreted on the CPU -----------------------------------------
@pytest.mark.parametrize("shape", [(200, 4, 2), (600, 6, 1), (600, 6, 3), (400, 0, 2), (1000, 50, 2)])
def test_record_streams_reproduce_the_oracle(shape):
    """tests/support/rec_sim.py executes the forward / backward record streams (direct records: plain entries, raw
    values, products, extend-add tables, L panels, solution indices in the caller's numbering) for one problem and
    must reproduce the oracle's inertia and solution."""
    import cannoles_jl_amd  # noqa: F401
    from cannoles_jl_amd import hipldl, synthetic as syn
    from oracle import oracle as O
    from tests.support.rec_sim import RecSim
    n, p, hw = shape
    # multipliers last: keeps the large (packed, two-word-product) fronts in play
    opts = hipldl.Options(multipliers_early=0) if p == 50 else None
    s = syn.band_structure(n, p, hw=hw)
    rows, cols = s.kkt_pattern()
    plan = hipldl.Plan(s.N, rows, cols, s.nvar, s.nequ, s.ncon, options=opts)
    assert plan.info["v2"] is not None and plan.info["ncond"] == s.nequ
    vals, rhs = syn.band_values(s, 77)
    off = s.offsets()
    vals[off[4]:off[5]] = -np.random.default_rng(1).uniform(0.5, 2.0, s.nequ)  # general residual pivots
    vals[off[6]:off[7]] = 0.25                                                  # some rho
    sim = RecSim(plan, s.nnzNS, s.N)
    eig_tol = 2.220446049250313e-16
    L, npos, nzer = sim.forward(vals, rhs, eig_tol)
    assert sim.stats["products"] > 0 and sim.stats["raw"] > 0
    if p == 50:
        assert sim.stats["packed"] > 0  # large fronts: packed staging, two-word products
    d = sim.backward(L, s.N, vals, rhs)
    perm = plan.array("perm").astype(np.int64)
    orc = O.Oracle(s.N, rows, cols, perm)
    ok, pos0, zer0 = orc.try_to_factorize(vals, s.nvar, s.nequ, s.ncon, eig_tol, return_inertia=True)
    d0 = orc.solve_ldl(rhs)
    dr = vals[off[4]:off[5]]
    assert sim.owned == s.nequ  # every condensed pivot is owned by exactly one front ...
    assert sim.own_pos == int((dr > eig_tol).sum()) and sim.own_zer == int((np.abs(dr) <= eig_tol).sum())  # ... which counts it
    assert npos + sim.own_pos == pos0 and nzer + sim.own_zer == zer0
    kept = ~np.isnan(d)
    # plans the lean kernel takes recover the residual components in the backward records (all of them or none)
    back_rows = bool(plan.array("brec")[7] & 256)
    if shape in ((200, 4, 2), (600, 6, 1)):
        assert back_rows   # (rows of more than five entries or fronts of more than 16 rows keep the product lists and the post-pass)
    if back_rows:
        assert kept.all()
    else:
        assert kept.sum() == s.N - s.nequ and not kept[s.nvar:s.nvar + s.nequ].any()
    assert np.abs(d[kept] - d0[kept]).max() <= 1e-10 * np.abs(d0).max()


@pytest.mark.parametrize("shape", [(200, 4, 2, 1), (1000, 10, 2, 64), (1000, 10, 2, 1), (600, 6, 3, 8), (400, 0, 2, 4), (2400, 12, 2, 1024), (3000, 6, 2, 2048), (10000, 50, 2, 1)])
def test_staged_plans_reproduce_the_oracle(shape):
    """Latency plans (what cnl_create builds for small batches: bushy order, elimination tree cut into tasks) executed by
    tests/support/rec_sim.py::StagedSim the way the STAGED kernel runs them — every task on its own LDS stack, task roots
    through the global scratch, tasks of a stage in either order, or all tasks in a depth-first order of the parent links (the
    dataflow execution) — must reproduce the oracle's inertia and solution."""
    from tests.support.rec_sim import StagedSim
    n, p, hw, batch = shape
    s = syn.band_structure(n, p, hw=hw)
    rows, cols = s.kkt_pattern()
    plan = hipldl.Plan(s.N, rows, cols, s.nvar, s.nequ, s.ncon, batch=batch)
    tasks = plan.array("tasks").reshape(-1, 8)
    assert len(tasks) > 1 and plan.array("stage_ptr")[-1] == len(tasks)
    if shape == (10000, 50, 2, 1):
        # the headline pattern's nested-dissection order has ONE front with 17 condensed residual rows: the analysis cuts it in two
        # (Options::split_positions), so that every front takes the row form and the plan runs the lean kernel with the residual
        # components recovered by the backward records
        assert plan.info["order"] == "nd32+early" and plan.info["nsuper"] == 1926 and plan.array("brec")[7] & 256
    if batch >= 1024:   # mid-size batches: a few large parts in canonical order, each one task
        assert plan.info["order"].startswith("ndc") and (tasks[:, 2] - tasks[:, 1]).max() > 20
    vals, rhs = syn.band_values(s, 78)
    off = s.offsets()
    vals[off[4]:off[5]] = -np.random.default_rng(2).uniform(0.5, 2.0, s.nequ)
    vals[off[6]:off[7]] = 0.125
    eig_tol = 2.220446049250313e-16
    perm = plan.array("perm").astype(np.int64)
    orc = O.Oracle(s.N, rows, cols, perm)
    ok, pos0, zer0 = orc.try_to_factorize(vals, s.nvar, s.nequ, s.ncon, eig_tol, return_inertia=True)
    d0 = orc.solve_ldl(rhs)
    for reverse, dataflow in ((False, False), (True, False), (False, True)):
        sim = StagedSim(plan, s.nnzNS, s.N)
        d, npos, nzer = sim.run(vals, rhs, eig_tol, s.N, reverse=reverse, dataflow=dataflow)
        assert (npos, nzer) == (pos0, zer0)
        kept = ~np.isnan(d)
        assert kept.sum() == (s.N if plan.array("brec")[7] & 256 else s.N - s.nequ)
        assert np.abs(d[kept] - d0[kept]).max() <= 1e-10 * np.abs(d0).max()


def test_analysis_does_not_depend_on_its_host_threads(built):
    """round 6: the candidate orders of the symbolic analysis are built and evaluated on host threads; the results land in a fixed order,
    so the plan (order, supernodes, record streams, tasks) is the same on one thread and on many — for a latency plan, a mid-size plan and
    a throughput plan"""
    s = syn.band_structure(1200, 12)
    rows, cols = s.kkt_pattern()
    for batch in (1, 200, 20000):
        a = hipldl.Plan(s.N, rows, cols, s.nvar, s.nequ, s.ncon, batch=batch, options=hipldl.Options(analysis_threads=1))
        b = hipldl.Plan(s.N, rows, cols, s.nvar, s.nequ, s.ncon, batch=batch, options=hipldl.Options(analysis_threads=7))
        c = hipldl.Plan(s.N, rows, cols, s.nvar, s.nequ, s.ncon, batch=batch)
        assert a.info == b.info == c.info
        for name in ("perm", "rec", "brec", "tasks"):
            assert np.array_equal(a.array(name), b.array(name)) and np.array_equal(a.array(name), c.array(name)), (batch, name)


def test_interleaved_layout_index_function():
    """CNL_LAYOUT_INTERLEAVED (include/cannoles_hip.h): the host mirror of the index function is a bijection of (problem, element) into
    the array cnl_layout_len sizes, keeps the eight doubles of a block together and the 32 problems of a group 64 bytes apart"""
    from cannoles_jl_amd import hipldl
    for B, n in ((1, 5), (33, 64), (70, 1001)):
        pp, ee = np.meshgrid(np.arange(B), np.arange(n), indexing="ij")
        idx = hipldl.il_index(pp, ee, n)
        assert idx.min() == 0 and idx.max() < hipldl.il_len(B, n) and np.unique(idx).size == B * n
        assert hipldl.il_len(B, n) == (B + 31) // 32 * ((n + 7) // 8 + 1) * 256
        m = min(n, 8)
        assert np.all(idx[:, 1:m] - idx[:, 0:m - 1] == 1)
        if B > 1:
            assert np.all(idx[1:min(B, 32), 0] - idx[0:min(B, 32) - 1, 0] == 8)
        if n > 8:
            assert idx[0, 8] - idx[0, 0] == 256
