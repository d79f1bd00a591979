"""Synthetic NLS Newton-system workloads, numpy only.

A workload is what the reference's solver would hand to the linear-solver
plugin: the NLPModels structures (1-based COO, lower triangle for Hessians),
from which the 7-segment KKT pattern of /root/reference/src/CaNNOLeS.jl:256-315
is built, and per-problem values laid out as `prepare_newton_system!`
(src/CaNNOLeS.jl:947-981) leaves them, plus the right-hand side
[dual; primal] of src/CaNNOLeS.jl:631-632.

Seeds follow SURVEY §8d: seed = 1000*cfg + problem index.
"""
from dataclasses import dataclass, field

import numpy as np


@dataclass
class Structure:
    """NLPModels structure queries of one model family (all 1-based)."""
    nvar: int
    nequ: int
    ncon: int
    hF: tuple  # hess_structure_residual  (rows, cols), lower triangle
    hc: tuple  # hess_structure           (rows, cols), lower triangle
    jF: tuple  # jac_structure_residual   (rows, cols)
    jc: tuple  # jac_structure            (rows, cols)
    name: str = ""
    meta: dict = field(default_factory=dict)

    @property
    def N(self):
        return self.nvar + self.nequ + self.ncon

    @property
    def nnzhF(self):
        return len(self.hF[0])

    @property
    def nnzhc(self):
        return len(self.hc[0]) if self.ncon > 0 else 0

    @property
    def nnzjF(self):
        return len(self.jF[0])

    @property
    def nnzjc(self):
        return len(self.jc[0]) if self.ncon > 0 else 0

    @property
    def nnzNS(self):
        return self.nnzhF + self.nnzhc + self.nnzjF + self.nnzjc + self.N

    def kkt_pattern(self):
        """rows, cols (1-based int64) of the Newton system, in the reference's
        segment order [H_F | H_c | J_F | J_c | -I | -dI | rI]
        (src/CaNNOLeS.jl:276-315)."""
        n, m, p = self.nvar, self.nequ, self.ncon
        i64 = np.int64
        seg_r = [np.asarray(self.hF[0], i64)]
        seg_c = [np.asarray(self.hF[1], i64)]
        if p > 0:
            seg_r.append(np.asarray(self.hc[0], i64))
            seg_c.append(np.asarray(self.hc[1], i64))
        seg_r.append(np.asarray(self.jF[0], i64) + n)
        seg_c.append(np.asarray(self.jF[1], i64))
        if p > 0:
            seg_r.append(np.asarray(self.jc[0], i64) + n + m)
            seg_c.append(np.asarray(self.jc[1], i64))
        seg_r.append(np.arange(n + 1, n + m + 1, dtype=i64))
        seg_c.append(np.arange(n + 1, n + m + 1, dtype=i64))
        if p > 0:
            seg_r.append(np.arange(n + m + 1, n + m + p + 1, dtype=i64))
            seg_c.append(np.arange(n + m + 1, n + m + p + 1, dtype=i64))
        seg_r.append(np.arange(1, n + 1, dtype=i64))
        seg_c.append(np.arange(1, n + 1, dtype=i64))
        return np.concatenate(seg_r), np.concatenate(seg_c)

    def offsets(self):
        """start offsets (0-based) of the 7 segments + total."""
        o = [0]
        for c in (self.nnzhF, self.nnzhc, self.nnzjF, self.nnzjc, self.nequ, self.ncon, self.nvar):
            o.append(o[-1] + c)
        return o


def band_structure(n, p, name="band", hw=2):
    """cfg3/cfg4 family: nequ = n; J_F band |i-j|<=hw (hw=2 in the BASELINE configs); H_F lower band of
    half-width hw; H_c diagonal; J_c row k nonzero on columns (n/p)(k-1)+1..(n/p)k."""
    assert p == 0 or n % p == 0
    jr, jc = [], []
    for k in range(-hw, hw + 1):
        i = np.arange(max(0, -k), min(n, n - k))  # residual row i (0-based), column i+k
        jr.append(i)
        jc.append(i + k)
    jr = np.concatenate(jr)
    jc = np.concatenate(jc)
    o = np.lexsort((jc, jr))  # row-major, as a row-wise AD Jacobian would be
    jF = (jr[o] + 1, jc[o] + 1)
    hr, hcl = [], []
    for k in range(0, hw + 1):
        j = np.arange(0, n - k)
        hr.append(j + k)
        hcl.append(j)
    hr = np.concatenate(hr)
    hcl = np.concatenate(hcl)
    o = np.lexsort((hr, hcl))  # column-major lower triangle (test/mgh01con.jl:148-162 convention)
    hF = (hr[o] + 1, hcl[o] + 1)
    if p > 0:
        w = n // p
        hc = (np.arange(1, n + 1), np.arange(1, n + 1))
        jcs = (np.repeat(np.arange(1, p + 1), w), np.arange(1, n + 1))
    else:
        hc = (np.zeros(0, np.int64), np.zeros(0, np.int64))
        jcs = (np.zeros(0, np.int64), np.zeros(0, np.int64))
    return Structure(n, n, p, hF, hc, jF, jcs, name=name, meta={"block": n // p if p else 0})


def band_values(s, seed, delta=0.1, stress=None):
    """Values for one problem of the band family.
    Returns vals (nnzNS, in prepare_newton_system! layout with rho=0) and rhs (N).
    stress=None | "ladder" (H_F diag=-10 on 10% of the variables, J scaled so
    that ||J||_2<=1: forces the rho ladder up to 605.5, nfact=6) |
    "illcond" (diag(J_F)*1e-8)."""
    rng = np.random.default_rng(seed)
    n, m, p = s.nvar, s.nequ, s.ncon
    off = s.offsets()
    vals = np.zeros(s.nnzNS)
    # H_F
    r, c = np.asarray(s.hF[0]), np.asarray(s.hF[1])
    h = np.where(r == c, rng.uniform(0.1, 1.0, len(r)), rng.uniform(-0.02, 0.02, len(r)))
    if stress == "ladder":
        neg = rng.uniform(size=n) < 0.10
        neg[0] = True
        h = np.where(r == c, np.where(neg[c - 1], -10.0, h), 0.0)
    vals[off[0]:off[1]] = h
    # J_F
    r, c = np.asarray(s.jF[0]), np.asarray(s.jF[1])
    j = np.where(r == c, 2.0 + rng.uniform(0, 1, len(r)), rng.uniform(-0.5, 0.5, len(r)))
    if stress == "ladder":
        j = j / 6.0  # row sums of |J| <= 3+4*0.5 = 5 -> ||J||_2 <= 5/6 < 1
    if stress == "illcond":
        j = np.where(r == c, j * 1e-8, j)
    vals[off[2]:off[3]] = j
    if p > 0:
        lam = rng.normal(size=p)
        b = rng.uniform(-0.1, 0.1, n)
        blk = (np.arange(n) // s.meta["block"]).astype(int)
        hcv = lam[blk] * b  # hess_coord!(x, lambda; obj_weight=0)
        if stress == "ladder":
            hcv = hcv * 0.0
        vals[off[1]:off[2]] = -hcv  # stored negated (src/CaNNOLeS.jl:971-972)
        jc = rng.uniform(-1, 1, s.nnzjc)
        if stress == "ladder":
            jc = jc / np.sqrt(s.meta["block"]) / 2.0
        vals[off[3]:off[4]] = jc
        vals[off[5]:off[6]] = -delta
    vals[off[4]:off[5]] = -1.0
    vals[off[6]:off[7]] = 0.0
    rhs = rng.normal(size=s.N)
    return vals, rhs


def dense_structure(n, m, name="dense"):
    """cfg2: unconstrained, J_F dense m x n in column-major COO, H_F diagonal."""
    jr = np.tile(np.arange(1, m + 1), n)
    jc = np.repeat(np.arange(1, n + 1), m)
    hF = (np.arange(1, n + 1), np.arange(1, n + 1))
    z = np.zeros(0, np.int64)
    return Structure(n, m, 0, hF, (z, z), (jr, jc), (z, z), name=name)


def dense_values(s, seed):
    rng = np.random.default_rng(seed)
    off = s.offsets()
    vals = np.zeros(s.nnzNS)
    vals[off[0]:off[1]] = rng.uniform(0.1, 1.0, s.nvar)
    vals[off[2]:off[3]] = rng.normal(size=s.nnzjF) / np.sqrt(s.nvar)
    vals[off[4]:off[5]] = -1.0
    rhs = rng.normal(size=s.N)
    return vals, rhs


def random_structure(n, m, p, density, seed, hess=True, name="random"):
    """Irregular sparse family used by the parity tests: random J_F / J_c
    patterns with duplicates allowed in the Hessian structures (the reference
    sums duplicate COO entries, src/solver_types.jl:53-59)."""
    rng = np.random.default_rng(seed)

    def rnd(rows, cols, dens, ensure_rows=True):
        mask = rng.uniform(size=(rows, cols)) < dens
        if ensure_rows:
            for i in range(rows):
                mask[i, rng.integers(cols)] = True
        r, c = np.nonzero(mask)
        return r + 1, c + 1

    jF = rnd(m, n, density)
    jc = rnd(p, n, density * 2) if p > 0 else (np.zeros(0, np.int64), np.zeros(0, np.int64))
    if hess:
        mask = np.tril(rng.uniform(size=(n, n)) < density)
        np.fill_diagonal(mask, True)
        r, c = np.nonzero(mask)
        hF = (r + 1, c + 1)
        # constraint Hessian structure: full-Lagrangian structure incl. duplicates of H_F slots
        mask2 = np.tril(rng.uniform(size=(n, n)) < density / 2)
        np.fill_diagonal(mask2, True)
        r2, c2 = np.nonzero(mask2)
        hc = (r2 + 1, c2 + 1) if p > 0 else (np.zeros(0, np.int64), np.zeros(0, np.int64))
    else:
        z = np.zeros(0, np.int64)
        hF, hc = (z, z), ((np.arange(1, n + 1), np.arange(1, n + 1)) if p > 0 else (z, z))
    return Structure(n, m, p, hF, hc, jF, jc, name=name)


def random_values(s, seed, delta=0.1, posdef=True):
    rng = np.random.default_rng(seed)
    off = s.offsets()
    vals = np.zeros(s.nnzNS)
    r, c = np.asarray(s.hF[0]), np.asarray(s.hF[1])
    if len(r):
        h = rng.uniform(-0.05, 0.05, len(r))
        h = np.where(r == c, rng.uniform(0.5, 1.5, len(r)) if posdef else rng.uniform(-1.5, 0.5, len(r)), h)
        vals[off[0]:off[1]] = h
    if s.ncon > 0:
        r, c = np.asarray(s.hc[0]), np.asarray(s.hc[1])
        vals[off[1]:off[2]] = -rng.uniform(-0.05, 0.05, len(r))
        vals[off[3]:off[4]] = rng.uniform(-1, 1, s.nnzjc)
        vals[off[5]:off[6]] = -delta
    vals[off[2]:off[3]] = rng.uniform(-1, 1, s.nnzjF)
    vals[off[4]:off[5]] = -1.0
    rhs = rng.normal(size=s.N)
    return vals, rhs


def batch_values(s, B, cfg, gen=band_values, **kw):
    """vals (B, nnzNS), rhs (B, N) for problems 0..B-1 of config `cfg`."""
    vals = np.empty((B, s.nnzNS))
    rhs = np.empty((B, s.N))
    for b in range(B):
        vals[b], rhs[b] = gen(s, 1000 * cfg + b, **kw)
    return vals, rhs


def dense_kkt(s, vals):
    """Dense symmetric K = sparse(rows, cols, vals) + strict-lower transpose,
    duplicates summed (small problems only; independent numpy check)."""
    rows, cols = s.kkt_pattern()
    K = np.zeros((s.N, s.N))
    np.add.at(K, (rows - 1, cols - 1), vals)
    return K + np.tril(K, -1).T
