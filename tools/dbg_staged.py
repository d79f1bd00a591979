import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import cannoles_jl_amd
from cannoles_jl_amd import hipldl, synthetic as syn
n, p, B = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
s = syn.band_structure(n, p)
rows, cols = s.kkt_pattern()
vals, rhs = syn.batch_values(s, B, cfg=4)
prm = hipldl.default_params()
def run(options=None):
    L = hipldl.HIPLDLStruct(s.N, rows, cols, None, s.nvar, s.nequ, s.ncon, batch=B, options=options)
    d = np.zeros((B, s.N))
    d, ok, rho, ro, nf = hipldl.newton_system_(d, s.nvar, s.nequ, s.ncon, rhs, vals.copy(), L, np.zeros(B), prm)
    cfg = L.config; info = L.info
    perm = L.plan_array("perm"); tk = L.plan_array("tasks").reshape(-1, 6)
    L.close()
    return d.reshape(B, s.N), np.atleast_1d(ok), np.atleast_1d(nf), cfg, info, perm, tk
d1, ok1, nf1, cfg1, info1, perm1, tk1 = run()
d0, ok0, nf0, cfg0, info0, perm0, tk0 = run(hipldl.Options(staged=0))
print(cfg1["kernel"], cfg0["kernel"], info1["order"], info1["nsuper"], "tasks", len(tk1))
print("ok", ok1.all(), ok0.all(), "nfact", nf1.max(), nf0.max())
for b in range(B):
    e = np.abs(d1[b] - d0[b])
    if e.max() > 1e-9 * np.abs(d0[b]).max():
        bad = np.nonzero(e > 1e-9 * np.abs(d0[b]).max())[0]
        print("problem", b, "max err", e.max(), "nbad", len(bad), "first bad comps", bad[:12], "last", bad[-5:])
        if b > 2: break
print("done")
