"""Band program (csrc/band.h, band.cpp) without a GPU: the generator's step / row / epoch blocks are executed by the numpy
interpreter tests/support/band_sim.py — operand pieces, eight-slot window, factor records through the out ring, junction of
the two parts, exactly as csrc/band.hip does — and compared with the oracle (restated /root/reference/src/CaNNOLeS.jl:1008-1052
+ src/solver_types.jl:53-98) on an elimination order the band path has no part in."""
import numpy as np
import pytest

import cannoles_jl_amd  # noqa: F401
from cannoles_jl_amd import hipldl, synthetic as syn
from oracle import oracle as O
from tests.support.band_sim import BandSim


def _plan(s, **opt):
    rows, cols = s.kkt_pattern()
    return hipldl.Plan(s.N, rows, cols, s.nvar, s.nequ, s.ncon, options=hipldl.Options(plan_kind=hipldl.PLAN_THROUGHPUT, **opt)), rows, cols


def _check(s, vals, rhs, params, rho_old=0.0, **opt):
    pl, rows, cols = _plan(s, **opt)
    sim = BandSim(pl)
    assert sim.ok, "the pattern is a band: the generator must accept it"
    B = vals.shape[0]
    v = vals.copy()
    d, ok, rho, ro, nf = sim.newton_system(v, rhs, s.nvar, rho_old, params)
    orc = O.Oracle(s.N, rows, cols, O.canonical_perm(s.nvar, s.nequ, s.ncon))
    v0 = vals.copy()
    d0, ok0, rho0, ro0, nf0 = O.newton_system_batch(orc, B, s.nvar, s.nequ, s.ncon, rhs, v0, np.full(B, float(rho_old)), params)
    assert np.array_equal(ok, ok0) and np.array_equal(nf, nf0) and np.array_equal(rho, rho0) and np.array_equal(ro, ro0)
    assert np.array_equal(v[:, -s.nvar:], v0[:, -s.nvar:])   # rho slots left as the reference leaves them
    for b in range(B):
        if ok0[b]:
            assert np.abs(d[b] - d0[b]).max() <= 1e-11 * np.abs(d0[b]).max()
    return sim


@pytest.mark.parametrize("n,p,hw,kernel", [(200, 4, 2, 1), (200, 0, 2, 1), (200, 0, 2, 2), (360, 6, 1, 1), (1000, 10, 2, 1), (96, 2, 2, 1)])
def test_band_program_reproduces_the_oracle(built, params, n, p, hw, kernel):
    """the generator's program (fifteen 64-byte operand pieces per epoch), interpreted on the CPU, against the oracle: one and two parts,
    half-widths 1 and 2, with and without constraints"""
    s = syn.band_structure(n, p, hw=hw)
    vals, rhs = syn.batch_values(s, 3, cfg=4)
    sim = _check(s, vals, rhs, params, band_kernel=kernel)
    assert sim.nparts == (1 if kernel == 2 or n < 80 else 2)


def test_band_program_ladder_and_hopeless(built, params):
    """cfg5's ladder problems climb to rho = 605.5 with six factorisations (fixture F3's rule); a problem no rho rescues gives up
    with rho > rho_max and leaves rho_old alone (src/CaNNOLeS.jl:1036-1047)"""
    s = syn.band_structure(400, 4)
    vals = np.stack([syn.band_values(s, 5000 + b, stress="ladder")[0] for b in range(3)])
    rhs = np.stack([syn.band_values(s, 5000 + b, stress="ladder")[1] for b in range(3)])
    vals[2, s.offsets()[0]] = -1e300   # hopeless
    _check(s, vals, rhs, params)
    _check(s, vals, rhs, params, rho_old=0.3)


def test_band_program_full_size_headline_pattern(built, params):
    """BASELINE config 3 (n = nequ = 1e4, ncon = 50): two parts of 5 002 steps, fifteen operand pieces per epoch"""
    s = syn.band_structure(10000, 50)
    vals, rhs = syn.batch_values(s, 2, cfg=3)
    sim = _check(s, vals, rhs, params)
    assert sim.nparts == 2 and sim.parts[0]["nsteps"] == sim.parts[1]["nsteps"] == 5002
    assert sim.lsize * 8 < 0.5e6   # factor records: six doubles per pivot


def test_patterns_that_are_no_band_are_refused(built):
    """irregular sparsity, a dense Jacobian, a residual row wider than the band: band_info[0] == 0 and the handle keeps the
    register-front kernel"""
    for s in (syn.random_structure(60, 80, 4, 0.1, seed=3), syn.dense_structure(40, 60), syn.band_structure(200, 4, hw=3)):
        pl, _, _ = _plan(s)
        assert pl.array("band_info")[0] == 0
    pl, _, _ = _plan(syn.band_structure(200, 4), band_kernel=0)
    assert pl.array("band_info")[0] == 0


def test_options_outside_the_public_structure_travel_as_tuning_pairs(built):
    """round 6: cnl_options holds a dozen switches; every other switch of csrc/options.h is a key=value pair of its `tuning` string.
    A tuning key changes the plan exactly like the old field did, public fields may be named there too, and an unknown key or a
    malformed pair is CNL_ERR_ARG (not silently ignored)."""
    s = syn.band_structure(200, 4)
    rows, cols = s.kkt_pattern()
    o = hipldl.Options(plan_kind=hipldl.PLAN_THROUGHPUT, band_form=0, lean_kernel=0)
    assert o.tuning == b"band_form=0,lean_kernel=0" and o.plan_kind == hipldl.PLAN_THROUGHPUT
    a = hipldl.Plan(s.N, rows, cols, s.nvar, s.nequ, s.ncon, batch=64, options=hipldl.Options(tuning="plan_kind=1,force_order=canonical"))
    b = hipldl.Plan(s.N, rows, cols, s.nvar, s.nequ, s.ncon, batch=64, options=hipldl.Options(plan_kind=hipldl.PLAN_THROUGHPUT, force_order="canonical"))
    assert a.info == b.info and np.array_equal(a.array("perm"), b.array("perm")) and a.info["order"].startswith("canonical")
    for bad in ("no_such_switch=1", "lean_kernel", "lean_kernel=x", "staged_large_fronts=1", "band_wide_pieces=1"):
        with pytest.raises(hipldl.CnlError) as e:
            hipldl.Plan(s.N, rows, cols, s.nvar, s.nequ, s.ncon, batch=4, options=hipldl.Options(tuning=bad))
        assert e.value.code == 1
