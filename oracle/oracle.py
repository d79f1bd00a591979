"""ctypes front-end of the CPU oracle (oracle/cnl_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package never imports this module.

The class mirrors the reference's `LDLFactStruct <: LinearSolverStruct`
(/root/reference/src/solver_types.jl:45-98) and the free function mirrors
`newton_system!` (/root/reference/src/CaNNOLeS.jl:1008-1052).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libcnl_oracle.so")

_i64p = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")


def build(force=False):
    """gcc-compile the oracle (a few hundred ms)."""
    src = os.path.join(_HERE, "cnl_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.cnlo_default_params.argtypes = [_f64p]
        L.cnlo_kkt_pattern.restype = C.c_int64
        L.cnlo_kkt_pattern.argtypes = [C.c_int64] * 3 + [C.c_int64, _i64p, _i64p] * 4 + [_i64p, _i64p, _f64p]
        L.cnlo_prepare.argtypes = [C.c_int64] * 7 + [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, _f64p]
        L.cnlo_create.restype = C.c_void_p
        L.cnlo_create.argtypes = [C.c_int64, C.c_int64, _i64p, _i64p, C.c_void_p]
        L.cnlo_destroy.argtypes = [C.c_void_p]
        L.cnlo_nnzL.restype = C.c_int64
        L.cnlo_nnzL.argtypes = [C.c_void_p]
        L.cnlo_nnzA.restype = C.c_int64
        L.cnlo_nnzA.argtypes = [C.c_void_p]
        L.cnlo_flops.restype = C.c_double
        L.cnlo_flops.argtypes = [C.c_void_p]
        for f in ("cnlo_D", "cnlo_Ax", "cnlo_Ap", "cnlo_Ai"):
            getattr(L, f).restype = C.c_void_p
            getattr(L, f).argtypes = [C.c_void_p]
        L.cnlo_set_vals.argtypes = [C.c_void_p, _f64p, C.c_int]
        L.cnlo_try_to_factorize.restype = C.c_int
        L.cnlo_try_to_factorize.argtypes = [C.c_void_p, _f64p, C.c_int64, C.c_int64, C.c_int64, C.c_double,
                                            C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.cnlo_solve_ldl.restype = C.c_int
        L.cnlo_solve_ldl.argtypes = [C.c_void_p, _f64p, _f64p]
        L.cnlo_newton_system.argtypes = [C.c_void_p, _f64p, C.c_int64, C.c_int64, C.c_int64, _f64p, _f64p,
                                         C.c_double, _f64p, C.c_int, _f64p]
        L.cnlo_newton_system_batch.argtypes = [C.c_void_p, C.c_int64, _f64p, C.c_int64, C.c_int64, C.c_int64,
                                               _f64p, _f64p, C.c_void_p, _f64p, C.c_int, _f64p]
        _lib = L
    return _lib


def default_params():
    """ParamCaNNOLeS(Float64) — src/CaNNOLeS.jl:48-62, as a length-9 vector
    [eig_tol, dmin, kdec, kinc, klargeinc, rho0, rhomax, rhomin, gammaA]."""
    p = np.zeros(9)
    lib().cnlo_default_params(p)
    return p


def _i64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.int64).ravel())


def _f64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64).ravel())


def kkt_pattern(nvar, nequ, ncon, hF=((), ()), hc=((), ()), jF=((), ()), jc=((), ())):
    """1-based COO pattern of the Newton system (src/CaNNOLeS.jl:256-315).
    Each argument is a (rows, cols) pair of 1-based structures as NLPModels
    returns them.  Returns rows, cols (int64) and the ctor's initial vals."""
    hFr, hFc = _i64(hF[0]), _i64(hF[1])
    hcr, hcc = _i64(hc[0]), _i64(hc[1])
    jFr, jFc = _i64(jF[0]), _i64(jF[1])
    jcr, jcc = _i64(jc[0]), _i64(jc[1])
    nnzhc = len(hcr) if ncon > 0 else 0
    nnzjc = len(jcr) if ncon > 0 else 0
    nnz = len(hFr) + nnzhc + len(jFr) + nnzjc + nvar + nequ + ncon
    rows = np.zeros(nnz, np.int64)
    cols = np.zeros(nnz, np.int64)
    vals = np.zeros(nnz)
    got = lib().cnlo_kkt_pattern(nvar, nequ, ncon, len(hFr), hFr, hFc, len(hcr), hcr, hcc,
                                 len(jFr), jFr, jFc, len(jcr), jcr, jcc, rows, cols, vals)
    assert got == nnz
    return rows, cols, vals


def prepare(vals, nvar, nequ, ncon, nnzhF, nnzhc, nnzjF, nnzjc, hF_vals, hc_vals, jF_vals, jc_vals, delta):
    """prepare_newton_system! (src/CaNNOLeS.jl:947-981), in place on vals."""
    def ptr(a):
        if a is None:
            return None
        a = _f64(a)
        keep.append(a)
        return a.ctypes.data
    keep = []
    assert vals.dtype == np.float64 and vals.flags.c_contiguous
    lib().cnlo_prepare(nvar, nequ, ncon, nnzhF, nnzhc, nnzjF, nnzjc, ptr(hF_vals), ptr(hc_vals),
                       ptr(jF_vals), ptr(jc_vals), float(delta), vals)
    return vals


def canonical_perm(nvar, nequ, ncon):
    """0-based elimination order "r-nodes, then x in natural order, then the
    multipliers" — the order SURVEY §8d defines nnz(L*) on."""
    return np.concatenate([np.arange(nvar, nvar + nequ), np.arange(nvar),
                           np.arange(nvar + nequ, nvar + nequ + ncon)]).astype(np.int64)


class Oracle:
    """Mirror of LDLFactStruct (src/solver_types.jl:45-65): built from the
    1-based lower-triangular COO pattern; `perm` replaces the AMD ordering."""

    def __init__(self, N, rows, cols, perm=None):
        self.N = int(N)
        self.rows = _i64(rows)
        self.cols = _i64(cols)
        self.nnz = len(self.rows)
        self._perm = None if perm is None else _i64(perm)
        h = lib().cnlo_create(self.N, self.nnz, self.rows, self.cols,
                              None if self._perm is None else self._perm.ctypes.data)
        if not h:
            raise ValueError("malformed pattern or permutation")
        self._h = C.c_void_p(h)
        self.set_mode = 0

    def __del__(self):
        if getattr(self, "_h", None):
            lib().cnlo_destroy(self._h)
            self._h = None

    @property
    def nnzL(self):
        return lib().cnlo_nnzL(self._h)

    @property
    def nnzA(self):
        return lib().cnlo_nnzA(self._h)

    @property
    def flops(self):
        return lib().cnlo_flops(self._h)

    @property
    def D(self):
        p = lib().cnlo_D(self._h)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_double)), shape=(self.N,)).copy()

    def csc_upper(self):
        """(colptr, rowval, nzval) of A = Symmetric(triu(sparse(cols,rows,vals)))"""
        n = self.nnzA
        Ap = np.ctypeslib.as_array(C.cast(lib().cnlo_Ap(self._h), C.POINTER(C.c_int64)), shape=(self.N + 1,)).copy()
        Ai = np.ctypeslib.as_array(C.cast(lib().cnlo_Ai(self._h), C.POINTER(C.c_int64)), shape=(n,)).copy()
        Ax = np.ctypeslib.as_array(C.cast(lib().cnlo_Ax(self._h), C.POINTER(C.c_double)), shape=(n,)).copy()
        return Ap, Ai, Ax

    def set_vals(self, vals):
        lib().cnlo_set_vals(self._h, _f64(vals), self.set_mode)

    def try_to_factorize(self, vals, nvar, nequ, ncon, eig_tol, return_inertia=False):
        npos, nzer = C.c_int64(0), C.c_int64(0)
        ok = lib().cnlo_try_to_factorize(self._h, _f64(vals), nvar, nequ, ncon, float(eig_tol),
                                         self.set_mode, C.byref(npos), C.byref(nzer))
        if return_inertia:
            return bool(ok), int(npos.value), int(nzer.value)
        return bool(ok)

    def solve_ldl(self, rhs):
        d = np.zeros(self.N)
        lib().cnlo_solve_ldl(self._h, _f64(rhs), d)
        return d


def newton_system(LDLT, nvar, nequ, ncon, rhs, vals, rho_old, params):
    """newton_system! (src/CaNNOLeS.jl:1008-1052).  `vals` (float64 ndarray) is
    mutated like the reference mutates get_vals(LDLT).
    Returns d, solve_success, rho, rho_old, nfact."""
    assert vals.dtype == np.float64 and vals.flags.c_contiguous
    d = np.zeros(LDLT.N)
    out = np.zeros(4)
    lib().cnlo_newton_system(LDLT._h, d, nvar, nequ, ncon, _f64(rhs), vals, float(rho_old), _f64(params),
                             LDLT.set_mode, out)
    return d, bool(out[0]), float(out[1]), float(out[2]), int(out[3])


def newton_system_batch(LDLT, B, nvar, nequ, ncon, rhs, vals, rho_old, params):
    """B independent systems with one pattern, problem-major arrays."""
    N = LDLT.N
    assert vals.shape == (B, LDLT.nnz) and vals.dtype == np.float64 and vals.flags.c_contiguous
    rhs = np.ascontiguousarray(rhs, dtype=np.float64).reshape(B, N)
    d = np.zeros((B, N))
    out = np.zeros((B, 4))
    ro = None if rho_old is None else _f64(rho_old)
    lib().cnlo_newton_system_batch(LDLT._h, B, d.reshape(-1), nvar, nequ, ncon, rhs.reshape(-1), vals.reshape(-1),
                                   None if ro is None else ro.ctypes.data, _f64(params), LDLT.set_mode,
                                   out.reshape(-1))
    return d, out[:, 0].astype(bool), out[:, 1].copy(), out[:, 2].copy(), out[:, 3].astype(np.int64)


# ---- vectors either side of the Newton system --------------------------------------------
def residual_vectors(rows, cols, vals, nvar, nequ, ncon, r, lam, Fx, cx):
    """Restates /root/reference/src/CaNNOLeS.jl:507-508,519-524,528-529 (and :722-726,730-731 at the trial
    point) and the rhs assembly :631-632 for ONE problem:
        Jxtr = Jx' r; Jcxtλ = Jc' λ  (SparseMatricesCOO mul!: y[col[k]] += val[k] * x[row[k]] for k = 1, 2, ...)
        dual = Jxtr - Jcxtλ; primal = [Fx - r; cx]; rhs = [dual; primal]; norms = (‖dual‖∞, ‖primal‖∞).
    The Jacobian values are the J_F / J_c segments of `vals` (prepare_newton_system!, :953-967), identified in the
    pattern (1-based `rows`, `cols`) as the entries with column <= nvar < row.  Pure-Python loops: small cases only.
    Test infrastructure, not part of the product."""
    rows = np.asarray(rows); cols = np.asarray(cols); vals = np.asarray(vals, dtype=np.float64)
    Jxtr = np.zeros(nvar); Jcxtl = np.zeros(nvar)
    for k in range(len(rows)):
        i, j = int(rows[k]) - 1, int(cols[k]) - 1
        if j < nvar <= i:
            if i < nvar + nequ:
                Jxtr[j] += vals[k] * r[i - nvar]
            else:
                Jcxtl[j] += vals[k] * lam[i - nvar - nequ]
    dual = Jxtr - Jcxtl
    primal = np.concatenate([np.asarray(Fx, dtype=np.float64) - np.asarray(r, dtype=np.float64),
                             np.asarray(cx, dtype=np.float64) if ncon else np.zeros(0)])
    rhs = np.concatenate([dual, primal])

    def ninf(v):  # norm(v, Inf): NaN propagates, empty -> 0
        return float(np.max(np.abs(v))) if len(v) and not np.isnan(v).any() else (float("nan") if len(v) else 0.0)

    return rhs, (ninf(dual), ninf(primal))


def trial_point(nvar, nequ, ncon, x, r, lam, d, max_dlambda=1e4):
    """Restates /root/reference/src/CaNNOLeS.jl:654,661-668: dλ = -d[n+m+1:N]; xt = x + dx; rt = r + dr;
    if norm(dλ) > Mdλ: dλ = dλ * Mdλ / norm(dλ); λt = λ + dλ.  Returns (xt, rt, λt, dλ)."""
    d = np.asarray(d, dtype=np.float64)
    dl = -d[nvar + nequ:nvar + nequ + ncon]
    xt = np.asarray(x, dtype=np.float64) + d[:nvar]
    rt = np.asarray(r, dtype=np.float64) + d[nvar:nvar + nequ]
    nrm = float(np.sqrt(np.sum(dl * dl)))
    if nrm > max_dlambda:
        dl = dl * max_dlambda / nrm
    return xt, rt, np.asarray(lam, dtype=np.float64) + dl, dl


def cgls_multipliers(rows, cols, vals, nvar, nequ, ncon, r, atol=None, rtol=None, itmax=0, ones_if_zero=True):
    """Restates /root/reference/src/CaNNOLeS.jl:507-518: Jxtr = Jx' r; lambda = cgls(Jc', Jxtr); lambda = 1 if it is
    zero.  The CGLS recurrence is Krylov.jl's (un-vendored dependency, Project.toml compat 0.10; textbook CGLS started at 0,
    stop when ||A' res|| <= atol + rtol ||A' b||, defaults sqrt(eps), itmax = m + n).  Returns (lambda, Jxtr, iterations).
    Test infrastructure, not part of the product."""
    eps = np.finfo(np.float64).eps
    atol = np.sqrt(eps) if atol is None else atol
    rtol = np.sqrt(eps) if rtol is None else rtol
    rows = np.asarray(rows); cols = np.asarray(cols); vals = np.asarray(vals, dtype=np.float64)
    i0, j0 = rows - 1, cols - 1
    selF = (j0 < nvar) & (i0 >= nvar) & (i0 < nvar + nequ)
    selC = (j0 < nvar) & (i0 >= nvar + nequ)
    Jxtr = np.zeros(nvar)
    np.add.at(Jxtr, j0[selF], vals[selF] * np.asarray(r, dtype=np.float64)[i0[selF] - nvar])
    A = np.zeros((nvar, ncon))  # A = Jc'
    np.add.at(A, (j0[selC], i0[selC] - nvar - nequ), vals[selC])
    x = np.zeros(ncon)
    res = Jxtr.copy()
    s = A.T @ res
    p = s.copy()
    gamma = float(s @ s)
    tol = atol + rtol * np.sqrt(gamma)
    itmax = nvar + ncon if itmax <= 0 else itmax
    it = 0
    while it < itmax and np.sqrt(gamma) > tol:
        q = A @ p
        delta = float(q @ q)
        if delta == 0.0:
            break
        alpha = gamma / delta
        x += alpha * p
        res -= alpha * q
        s = A.T @ res
        gnext = float(s @ s)
        p = s + (gnext / gamma) * p
        gamma = gnext
        it += 1
    if ones_if_zero and ncon and np.linalg.norm(x) == 0:
        x[:] = 1.0
    return x, Jxtr, it
