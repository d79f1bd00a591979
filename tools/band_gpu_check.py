import sys, time
sys.path.insert(0,'/root/repo')
import numpy as np
import cannoles_jl_amd
from cannoles_jl_amd import hipldl, synthetic as syn
from oracle import oracle as O
def run(n, p, B, bk=1, nl=16, stress=None, ncheck=4):
    s = syn.band_structure(n, p)
    rows, cols = s.kkt_pattern()
    if stress:
        vals = np.stack([syn.band_values(s, 5000+b, stress=stress)[0] for b in range(B)]); rhs = np.stack([syn.band_values(s, 5000+b, stress=stress)[1] for b in range(B)])
    else:
        vals, rhs = syn.batch_values(s, B, cfg=4)
    par = hipldl.default_params()
    L = hipldl.HIPLDLStruct(s.N, rows, cols, None, s.nvar, s.nequ, s.ncon, batch=B, options=hipldl.Options(plan_kind=hipldl.PLAN_THROUGHPUT, band_kernel=bk, band_problems_per_group=nl))
    print("config", L.config)
    d = np.zeros((B, s.N)); v = vals.copy()
    d, ok, rho, ro, nf = hipldl.newton_system_(d, s.nvar, s.nequ, s.ncon, rhs, v, L, np.zeros(B), par)
    orc = O.Oracle(s.N, rows, cols, O.canonical_perm(s.nvar, s.nequ, s.ncon))
    idx = list(range(min(B, ncheck))) + ([B-1] if B > ncheck else [])
    worst = 0
    for b in idx:
        vv = vals[b].copy()
        d0, ok0, rho0, ro0, nf0 = O.newton_system(orc, s.nvar, s.nequ, s.ncon, rhs[b], vv, 0.0, par)
        assert (bool(ok[b]), float(rho[b]), float(ro[b]), int(nf[b])) == (ok0, rho0, ro0, nf0), (b, ok[b], rho[b], ro[b], nf[b], ok0, rho0, ro0, nf0)
        assert np.array_equal(v[b, -s.nvar:], vv[-s.nvar:])
        err = np.abs(d[b]-d0).max()/np.abs(d0).max()
        worst = max(worst, err)
    print(f"n={n} p={p} B={B} bk={bk} nl={nl} stress={stress}: OK, worst rel err {worst:.2e}, nf {nf[:4]}")
    L.close()
if __name__ == "__main__":
    run(200, 0, 3, bk=2)
    run(200, 0, 3)
    run(200, 4, 5)
    run(1000, 10, 37)
    run(1000, 10, 37, nl=8)
    run(1000, 10, 37, nl=32)
    run(1000, 10, 20, stress="ladder")
    run(10000, 50, 64)
