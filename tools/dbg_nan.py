import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import cannoles_jl_amd
from cannoles_jl_amd import hipldl, synthetic as syn
s = syn.band_structure(300, 4); rows, cols = s.kkt_pattern()
B = 6
vals, rhs = syn.batch_values(s, B, cfg=4)
off = s.offsets()
vals[2, off[0]:off[1]] = np.nan
p = hipldl.default_params()
for env in ({}, {"CNL_NO_STAGED": "1"}, {"CNL_STAGED_MAX": "0"}):
    os.environ.pop("CNL_NO_STAGED", None); os.environ.pop("CNL_STAGED_MAX", None); os.environ.update(env)
    L = hipldl.HIPLDLStruct(s.N, rows, cols, None, s.nvar, s.nequ, s.ncon, batch=B)
    d = np.full((B, s.N), 7.0)
    d, ok, rho, ro, nf = hipldl.newton_system_(d, s.nvar, s.nequ, s.ncon, rhs, vals.copy(), L, np.zeros(B), p)
    d = d.reshape(B, -1)
    print(env, L.config["kernel"], L.info["order"], "ok", list(ok), "nf", list(nf))
    for b in range(B):
        bad = np.nonzero(~np.isfinite(d[b]))[0]
        print("   b", b, "nonfinite", len(bad), bad[:8], "untouched", int((d[b] == 7.0).sum()))
    L.close()
