"""Test models for the outer loop.

The reference's own tests never call the linear-solver boundary directly; they assert final solutions
of `cannoles(nls)` (/root/reference/test/runtests.jl:56-113).  To check the oracle (and the HIP path) against
those known answers, the tests drive the restated outer loop (`cannoles.jl_amd/outer_loop.py`, mirror of
`solve!`, /root/reference/src/CaNNOLeS.jl:418-864) with the models below.

Model callbacks (residual, Jacobians, Hessians) come from sympy, standing in for ADNLPModels; dense
structures are used (the reference accepts any NLPModels structure).
"""
import numpy as np
import sympy as sp


class SymNLS:
    """Equality-constrained NLS model  min 1/2 ||F(x)||^2  s.t. c(x) = 0  with exact derivatives."""

    def __init__(self, F, x0, c=None):
        n = len(x0)
        xs = sp.symbols(f"x0:{n}")
        Fx = sp.Matrix(F(xs))
        self.nvar, self.nequ = n, Fx.shape[0]
        self.x0 = np.asarray(x0, float)
        cx = sp.Matrix(c(xs)) if c is not None else sp.Matrix(0, 1, [])
        self.ncon = cx.shape[0]
        rs = sp.symbols(f"r0:{self.nequ}")
        ls = sp.symbols(f"l0:{max(self.ncon, 1)}")[:self.ncon]
        self._F = sp.lambdify([xs], Fx, "numpy")
        self._J = sp.lambdify([xs], Fx.jacobian(xs), "numpy")
        HF = sum((rs[i] * sp.hessian(Fx[i], xs) for i in range(self.nequ)), sp.zeros(n, n))
        self._HF = sp.lambdify([xs, rs], HF, "numpy")
        self._c = sp.lambdify([xs], cx, "numpy")
        self._Jc = sp.lambdify([xs], cx.jacobian(xs) if self.ncon else sp.zeros(0, n), "numpy")
        Hc = sum((ls[i] * sp.hessian(cx[i], xs) for i in range(self.ncon)), sp.zeros(n, n))
        self._Hc = sp.lambdify([xs, ls], Hc, "numpy")
        il, jl = np.tril_indices(n)
        # dense structures, 1-based, lower triangle for Hessians (column-major like test/mgh01con.jl:148-162)
        o = np.lexsort((il, jl))
        self.h_rows, self.h_cols = il[o] + 1, jl[o] + 1
        jr, jc = np.nonzero(np.ones((self.nequ, n)))
        self.jF_rows, self.jF_cols = jr + 1, jc + 1
        cr, cc = np.nonzero(np.ones((self.ncon, n)))
        self.jc_rows, self.jc_cols = cr + 1, cc + 1
        self.neval = 0

    def residual(self, x):
        self.neval += 1
        return np.asarray(self._F(x), float).ravel()

    def jac_residual(self, x):
        return np.asarray(self._J(x), float).reshape(self.nequ, self.nvar)

    def hess_coord_residual(self, x, r):
        H = np.asarray(self._HF(x, r), float).reshape(self.nvar, self.nvar)
        return H[self.h_rows - 1, self.h_cols - 1]

    def cons(self, x):
        self.neval += 1
        return np.asarray(self._c(x), float).ravel()

    def jac(self, x):
        return np.asarray(self._Jc(x), float).reshape(self.ncon, self.nvar)

    def hess_coord_cons(self, x, lam):
        """hess_coord!(nls, x, lambda; obj_weight = 0): lower triangle of sum_i lambda_i Hess c_i
        (NLPModels convention: the Lagrangian is f + lambda'c)."""
        if self.ncon == 0:
            return np.zeros(0)
        H = np.asarray(self._Hc(x, lam), float).reshape(self.nvar, self.nvar)
        return H[self.h_rows - 1, self.h_cols - 1]


from cannoles_jl_amd.outer_loop import solve  # noqa: E402,F401  (the loop itself lives in the package: row f3)
