"""Parity checks of the EXPERIMENT kernel band_newton_mw_kernel (loader wavefronts, csrc/band.hip; +8 % at 8192 problems only, not in the
product build).  Build the library with  make CXXFLAGS="-O3 -std=c++17 -fPIC -DCNL_EXPERIMENT=1 -DBAND_MW"  (cnl_version() turns
negative: the bindings need CANNOLES_HIP_ALLOW_EXPERIMENT=1), then run this script on a GPU box: every case through run_case of
tests/test_gpu_parity.py against the oracle, and bit-equality with band_newton_kernel."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_parity as T  # noqa: E402


def main():
    hipldl, syn, O = T._mods()
    mv = int(os.environ.get("MW_VARIANT", "1"))   # tuning key band_movers: 1 .. 3 (csrc/band.hip)
    for (n, p, B, hw) in [(200, 4, 5, 2), (1000, 10, 37, 2), (1000, 10, 70, 2), (360, 6, 19, 1), (400, 0, 33, 2), (10000, 50, 33, 2)]:
        s = syn.band_structure(n, p, hw=hw)
        vals, rhs = syn.batch_values(s, B, cfg=4)
        info, cfg = T.run_case(s, vals, rhs, options=T._band_opts(hipldl, band_movers=mv))
        assert cfg["band"] and cfg["band_movers"] and cfg["band_nl"] == (32 if mv == 2 else 16)
        print("ok", n, p, B, hw, flush=True)
    s = syn.band_structure(600, 6)
    B = 45
    vals, rhs = syn.batch_values(s, B, cfg=4)
    for b in (1, 5, 17, 20, 44):
        vals[b], rhs[b] = syn.band_values(s, 5000 + b, stress="ladder")
    vals[9, s.offsets()[0]] = -1e300
    ro = np.zeros(B)
    ro[5] = 0.3
    ro[2] = 1e-3
    T.run_case(s, vals, rhs, rho_old=ro, options=T._band_opts(hipldl, band_movers=mv))
    rows, cols = s.kkt_pattern()
    p = hipldl.default_params()
    out = {}
    for mw in (0, mv):
        L = hipldl.HIPLDLStruct(s.N, rows, cols, None, s.nvar, s.nequ, s.ncon, batch=B, options=T._band_opts(hipldl, band_movers=mw, band_problems_per_group=32 if mv == 2 else 16))
        assert L.config["band"] and L.config["band_movers"] == bool(mw)
        okf, npos, nzer = hipldl.try_to_factorize(L, vals, s.nvar, s.nequ, s.ncon, p[0], return_inertia=True)
        v = vals.copy()
        d = np.full((B, s.N), 3.0)
        d, ok, rho, ro_out, nf = hipldl.newton_system_(d, s.nvar, s.nequ, s.ncon, rhs, v, L, ro, p)
        d2 = np.full((B, s.N), 3.0)
        hipldl.solve_ldl_(rhs, L.factor, d2)
        out[mw] = [np.array(x, copy=True) for x in (okf, npos, nzer, d, ok, rho, ro_out, nf, v, d2)]
        L.close()
    for a, b in zip(out[0], out[mv]):
        assert np.array_equal(a, b)
    print("bit-equal to band_newton_kernel: ok")


if __name__ == "__main__":
    main()
