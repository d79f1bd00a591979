"""Details of one case of tools/fuzz_parity.py: the product against the oracle on three elimination orders."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cannoles_jl_amd  # noqa
from cannoles_jl_amd import hipldl, synthetic as syn
from oracle import oracle as O
seed = int(sys.argv[1])
p = hipldl.default_params(); po = O.default_params()
rng = np.random.default_rng(100000 + seed)
fam = rng.integers(3)
if fam == 0:
    n = int(rng.integers(6, 120)); m = int(rng.integers(max(2, n // 2), 2 * n)); pc = int(rng.integers(0, min(6, n // 2) + 1))
    s = syn.random_structure(n, m, pc, float(rng.uniform(0.03, 0.3)), seed, hess=bool(rng.integers(4)))
elif fam == 1:
    pc = int(rng.integers(0, 5)); blocks = int(rng.integers(8, 60)); n = (pc if pc else 1) * blocks
    s = syn.band_structure(n, pc, hw=int(rng.integers(1, 4)))
else:
    n = int(rng.integers(130, 400)); m = int(rng.integers(n, n + 60)); pc = int(rng.integers(0, 4))
    s = syn.random_structure(n, m, pc, float(rng.uniform(0.01, 0.04)), seed)
B = int(rng.choice([1, 2, 3, 5, 7, 17, 33, 64]))
posdef = bool(rng.integers(3))
if fam == 1:
    vals, rhs = syn.batch_values(s, B, cfg=seed % 7, stress=None if posdef else "ladder")
else:
    vals = np.empty((B, s.nnzNS)); rhs = np.empty((B, s.N))
    for b in range(B):
        vals[b], rhs[b] = syn.random_values(s, 7000 * seed + b, posdef=posdef or bool(rng.integers(2)))
ro_in = np.where(rng.uniform(size=B) < 0.3, 10.0 ** rng.uniform(-6, -1, B), 0.0)
rows, cols = s.kkt_pattern()
kind = [hipldl.PLAN_AUTO, hipldl.PLAN_THROUGHPUT, hipldl.PLAN_LATENCY][int(rng.integers(3))]
for kd in (kind, hipldl.PLAN_THROUGHPUT, hipldl.PLAN_LATENCY):
    L = hipldl.HIPLDLStruct(s.N, rows, cols, None, s.nvar, s.nequ, s.ncon, batch=B, options=hipldl.Options(plan_kind=kd))
    v = vals.copy(); d = np.full((B, s.N), 7.0)
    d, ok, rho, ro, nf = hipldl.newton_system_(d, s.nvar, s.nequ, s.ncon, rhs, v, L, ro_in if B > 1 else float(ro_in[0]), p)
    print("product kind", kd, L.config["kernel"], L.info["order"], "ok", np.atleast_1d(ok).astype(int), "nf", np.atleast_1d(nf), "rho", np.atleast_1d(rho))
    perm = L.plan_array("perm").astype(np.int64)
    L.close()
for name, pm in (("product perm", perm), ("canonical", O.canonical_perm(s.nvar, s.nequ, s.ncon)), ("natural", None)):
    orc = O.Oracle(s.N, rows, cols, pm)
    d0, ok0, rho0, ro0, nf0 = O.newton_system_batch(orc, B, s.nvar, s.nequ, s.ncon, rhs, vals.copy(), ro_in, po)
    print("oracle", name, "ok", ok0.astype(int), "nf", nf0, "rho", rho0)
    # pivots of the last factorisation of problem 0 closest to the threshold
    for b in range(min(B, 4)):
        vv = vals[b].copy()
        for k in range(int(nf0[b])):
            pass
orc = O.Oracle(s.N, rows, cols, O.canonical_perm(s.nvar, s.nequ, s.ncon))
for b in range(B):
    vv = vals[b].copy()
    ok_b, npos, nzer = orc.try_to_factorize(vv, s.nvar, s.nequ, s.ncon, po[0], return_inertia=True)
    D = orc.D
    print("problem", b, "first factorisation (canonical): ok", ok_b, "npos", npos, "nzero", nzer, "smallest |D|", np.sort(np.abs(D))[:3])
