#!/usr/bin/env python3
"""Writes tests/golden/fixtures.json: known-answer vectors at the linear-solver boundary.

The reference (Julia) cannot run in this environment and its tests hold no vector at this boundary,
so the fixtures are DERIVED BY HAND from the reference's own test model and formulas, with exact
rational arithmetic where an exact answer exists:

F1  first Newton system of MGH01CON (/root/reference/test/mgh01con.jl; x0 = [-1.2, 1], method :Newton):
    pattern per src/CaNNOLeS.jl:256-315, values per prepare_newton_system! (:947-981), rhs per :631-632,
    lambda0 = -107.8 from the least-squares multiplier estimate (:512-518), delta = 0.1 (:615).
    Expected d = -K^-1 rhs solved exactly with fractions.
F2  same system with rho = rho0 on the two rho slots: slot (1,1) is the COO-order sum of three entries.
F3  rho ladder: H = -10 I, ||J||_2 <= 1  =>  failures at rho in {0, rho0, 1e2 rho0, 1e4 rho0, 1e6 rho0},
    success at 1e8 rho0 = 605.5454452393343: nfact = 6, rho_old_out = rho (src/CaNNOLeS.jl:1029-1046).
"""
import json
import os
from fractions import Fraction as Fr

rho0 = 6.055454452393343e-06  # eps^T(1/3) = pow(eps, 0.3333333333333333), src/CaNNOLeS.jl:56 (not cbrt: 6.0554544523933395e-06)


def solve_exact(K, b):
    n = len(b)
    A = [[Fr(K[i][j]) for j in range(n)] + [Fr(b[i])] for i in range(n)]
    for c in range(n):
        p = next(r for r in range(c, n) if A[r][c] != 0)
        A[c], A[p] = A[p], A[c]
        for r in range(n):
            if r != c and A[r][c] != 0:
                f = A[r][c] / A[c][c]
                A[r] = [x - f * y for x, y in zip(A[r], A[c])]
    return [A[i][n] / A[i][i] for i in range(n)]


def main():
    # ---- F1 ----
    x0 = [Fr(-12, 10), Fr(1)]
    Fx = [1 - x0[0], 10 * (x0[1] - x0[0] ** 2)]           # residual!
    r = list(Fx)
    J = [[Fr(-1), Fr(0)], [-20 * x0[0], Fr(10)]]          # jac_coord_residual!
    Jtr = [J[0][0] * r[0] + J[1][0] * r[1], J[0][1] * r[0] + J[1][1] * r[1]]
    # Jc = [1 0]: least-squares multiplier min |Jc' lam - Jtr| -> lam = Jtr[0]
    lam = Jtr[0]
    dual = [Jtr[0] - lam, Jtr[1]]
    cx = [x0[0]]
    delta = Fr(1, 10)                                      # max(dmin, min(0.1*1, |dual|inf + |primal|inf))
    hF = [-20 * r[1]]                                      # hess_coord_residual!
    rows = [1, 1, 2, 2, 3, 4, 4, 5, 3, 4, 5, 1, 2]
    cols = [1, 1, 1, 2, 1, 1, 2, 1, 3, 4, 5, 1, 2]
    vals = [hF[0], Fr(0), Fr(0), Fr(0), J[0][0], J[1][0], J[1][1], Fr(1), Fr(-1), Fr(-1), -delta, Fr(0), Fr(0)]
    rhs = dual + [Fx[0] - r[0], Fx[1] - r[1]] + cx
    K = [[Fr(0)] * 5 for _ in range(5)]
    for i, j, v in zip(rows, cols, vals):
        K[i - 1][j - 1] += v
        if i != j:
            K[j - 1][i - 1] += v
    sol = solve_exact(K, rhs)
    d = [-s for s in sol]
    f1 = {"nvar": 2, "nequ": 2, "ncon": 1, "rows": rows, "cols": cols, "vals": [float(v) for v in vals],
          "rhs": [float(v) for v in rhs], "d": [float(v) for v in d], "d_exact": [str(v) for v in d],
          "inertia": [2, 3, 0], "lambda0": float(lam), "delta": float(delta)}
    # ---- F2 ----
    vals2 = list(vals)
    vals2[11] = vals2[12] = Fr(rho0)
    K2 = [[Fr(0)] * 5 for _ in range(5)]
    for i, j, v in zip(rows, cols, vals2):
        K2[i - 1][j - 1] += v
        if i != j:
            K2[j - 1][i - 1] += v
    d2 = [-s for s in solve_exact(K2, rhs)]
    f2 = {"vals": [float(v) for v in vals2], "slot11_float_sum": (float(vals[0]) + float(vals[1])) + rho0,
          "d": [float(v) for v in d2]}
    # ---- F3 ---- (n = 3, m = 3, p = 0): H = -10 I, J = 0.5 I
    n = m = 3
    rows3 = [1, 2, 3] + [4, 5, 6] + [4, 5, 6] + [1, 2, 3]
    cols3 = [1, 2, 3] + [1, 2, 3] + [4, 5, 6] + [1, 2, 3]
    vals3 = [-10.0] * 3 + [0.5] * 3 + [-1.0] * 3 + [0.0] * 3
    ladder = [0.0] + [rho0 * 100.0 ** k for k in range(5)]
    f3 = {"nvar": n, "nequ": m, "ncon": 0, "rows": rows3, "cols": cols3, "vals": vals3, "rhs": [1.0, 2.0, 3.0, 4.0, 5.0, 6.0],
          "rho_tried": ladder, "nfact": 6, "rho": ladder[-1]}
    out = {"F1": f1, "F2": f2, "F3": f3, "rho0": rho0}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fixtures.json")
    with open(path, "w") as fh:
        json.dump(out, fh, indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
