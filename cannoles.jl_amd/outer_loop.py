"""Host-side restatement of the reference's outer/inner regularisation loop.

`solve` mirrors `SolverCore.solve!` (/root/reference/src/CaNNOLeS.jl:418-864), the Armijo `line_search`
(:1054-1112) and the helpers (:872-920) in numpy for ONE problem, with the Newton step delegated to a pluggable
`newton_system` callable — the product's HIP path (see batch_solve.py, which runs many of these loops over one
batched handle) or, in the tests, the CPU oracle.  The model object provides residual / jac_residual /
hess_coord_residual / cons / jac / hess_coord_cons and the structure arrays (h_rows, h_cols, jF_rows, jF_cols,
jc_rows, jc_cols, 1-based), like an NLPModels model does.  The Krylov CGLS call of the reference (min-norm
least-squares multipliers, :512-518) is the same CGLS recurrence in numpy (`cgls` below).  Logging,
timing and callback plumbing are not restated.
"""
import numpy as np



def cgls(A, b, atol=None, rtol=None, itmax=0):
    """CGLS for min ||A x - b|| started at 0 — the recurrence of Krylov.jl's `cgls` (the reference's multiplier estimate,
    src/CaNNOLeS.jl:512-518, 880-882): stop when ||A' res|| <= atol + rtol ||A' b|| (defaults sqrt(eps)) or after m + n
    steps.  Same recurrence as csrc/kernels_aux.hip (cnl_cgls_multipliers_dev), which the device-resident loop uses."""
    eps = np.finfo(np.float64).eps
    atol = np.sqrt(eps) if atol is None else atol
    rtol = np.sqrt(eps) if rtol is None else rtol
    m, n = A.shape
    x = np.zeros(n)
    res = np.asarray(b, float).copy()
    s = A.T @ res
    pdir = s.copy()
    gamma = float(s @ s)
    tol = atol + rtol * np.sqrt(gamma)
    itmax = m + n if itmax <= 0 else itmax
    it = 0
    while it < itmax and np.sqrt(gamma) > tol:
        q = A @ pdir
        delta = float(q @ q)
        if delta == 0.0:
            break
        alpha = gamma / delta
        x += alpha * pdir
        res -= alpha * q
        s = A.T @ res
        gnext = float(s @ s)
        pdir = s + (gnext / gamma) * pdir
        gamma = gnext
        it += 1
    return x

def solve(nls, make_solver, newton_system, params, method="Newton", x=None, lam=None, max_iter=-1, max_eval=100000,
          max_inner=10000, atol=None, rtol=None, Fatol=None, Frtol=None, always_accept_extrapolation=False,
          delta_dec=0.1):
    """src/CaNNOLeS.jl:418-864.  `make_solver(N, rows, cols, vals, nvar, nequ, ncon)` builds the linear-solver
    object (LinearSolverStruct); `newton_system(LDLT, nvar, nequ, ncon, rhs, vals, rho_old, params)` returns
    (d, solve_success, rho, rho_old, nfact).  Returns a dict with solution, status, iter, nfact, nlinsolve."""
    eps = np.finfo(float).eps
    atol = np.sqrt(eps) if atol is None else atol
    rtol = np.sqrt(eps) if rtol is None else rtol
    Fatol = np.sqrt(eps) if Fatol is None else Fatol
    Frtol = eps if Frtol is None else Frtol
    n, m, p = nls.nvar, nls.nequ, nls.ncon
    N = n + m + p
    use_hF = method in ("Newton", "Newton_vanishing")
    nnzhF = len(nls.h_rows) if use_hF else 0
    nnzhc = len(nls.h_rows) if p > 0 else 0
    nnzjF, nnzjc = len(nls.jF_rows), len(nls.jc_rows) if p > 0 else 0
    # pattern, src/CaNNOLeS.jl:276-315
    rows = np.concatenate([nls.h_rows[:nnzhF], nls.h_rows[:nnzhc], nls.jF_rows + n, (nls.jc_rows + n + m)[:nnzjc],
                           np.arange(n + 1, n + m + 1), np.arange(n + m + 1, N + 1), np.arange(1, n + 1)]).astype(np.int64)
    cols = np.concatenate([nls.h_cols[:nnzhF], nls.h_cols[:nnzhc], nls.jF_cols, nls.jc_cols[:nnzjc],
                           np.arange(n + 1, n + m + 1), np.arange(n + m + 1, N + 1), np.arange(1, n + 1)]).astype(np.int64)
    vals = np.ones(len(rows))
    o_jF = nnzhF + nnzhc
    o_jc = o_jF + nnzjF
    o_I = o_jc + nnzjc
    o_d = o_I + m
    o_rho = o_d + p
    vals[o_I:o_d] = -1.0
    LDLT = make_solver(N, rows, cols, vals, n, m, p)

    x = nls.x0.copy() if x is None else np.asarray(x, float).copy()
    lam = np.zeros(p) if lam is None else np.asarray(lam, float).copy()
    rho = rho_old = 0.0
    delta = 1.0
    Fx = nls.residual(x)
    if not np.all(np.isfinite(Fx)):
        raise ValueError("Initial point gives Inf or Nan")
    fx = Fx @ Fx / 2
    Jx = nls.jac_residual(x)
    cx = nls.cons(x) if p else np.zeros(0)
    Jcx = nls.jac(x) if p else np.zeros((0, n))
    r = Fx.copy()
    Jxtr = Jx.T @ r

    def ls_multipliers(rhs_vec):  # krylov_solve!(cgls_workspace, Jcx', Jxtr): min || Jcx' lam - rhs || (src/CaNNOLeS.jl:512-518)
        return cgls(Jcx.T, rhs_vec) if p else np.zeros(0)

    lam = ls_multipliers(Jxtr)
    if p and np.linalg.norm(lam) == 0:
        lam[:] = 1.0
    dual = Jxtr - Jcx.T @ lam
    primal = np.concatenate([Fx - r, cx])
    normdual = np.linalg.norm(dual, np.inf)
    normprimal = np.linalg.norm(primal, np.inf) if len(primal) else 0.0
    smax = 100.0
    epsF = Fatol + Frtol * 2 * np.sqrt(fx)
    epstol = atol + rtol * normdual
    epsc = np.sqrt(epstol)

    def dual_scaling(l_):
        return max(smax, np.abs(l_).sum() / p) / smax if p > 0 else 1.0

    def small_res_check():
        nonlocal r, Jxtr, lam, dual, primal
        r = Fx.copy()
        Jxtr = Jx.T @ r
        lam = ls_multipliers(Jxtr)
        dual = Jxtr - Jcx.T @ lam
        primal = np.concatenate([np.zeros(m), cx])
        return (np.linalg.norm(cx, np.inf) if p else 0.0), np.linalg.norm(dual, np.inf)

    small_residual = 2 * np.sqrt(fx) <= epsF and np.linalg.norm(cx) <= epsc
    first_order = max(normdual / dual_scaling(lam), normprimal) <= epstol
    if small_residual and not first_order:
        normprimal, normdual = small_res_check()
        first_order = max(normdual / dual_scaling(lam), normprimal) <= epstol
    eta = 1.0 if p else 0.0
    it = 0
    nfact = nlinsolve = nbk = 0
    epsk = 1e3
    broken = False
    tired = nls.neval > max_eval

    def status():
        if first_order:
            return "first_order"
        if small_residual:
            return "small_residual"
        if broken:
            return "exception"
        if tired:
            return "max_eval"
        if max_iter >= 0 and it > max_iter:
            return "max_iter"
        return "unknown"

    phi = lambda Fv, cv, lv, et: Fv @ Fv / 2 - lv @ cv + et * (cv @ cv) / 2
    d = np.zeros(N)
    dlam = np.zeros(p)
    st = status()
    while st == "unknown":
        combined = normdual + normprimal
        delta = max(params[1], min(delta_dec * delta, combined))
        inner = 0
        combined_hat = np.inf
        first_iteration = True
        xt, rt, lamt, Ft, ct = x.copy(), r.copy(), lam.copy(), Fx.copy(), cx.copy()
        Jt, Jct = Jx, Jcx
        normdualhat, normprimalhat = normdual, normprimal
        while first_iteration or not (combined_hat <= 0.99 * combined + epsk or tired):
            first_iteration = False
            if inner != 1 or always_accept_extrapolation:
                # prepare_newton_system!, src/CaNNOLeS.jl:947-981
                if use_hF:
                    vals[:nnzhF] = nls.hess_coord_residual(x, r)
                vals[o_jF:o_jc] = Jx[nls.jF_rows - 1, nls.jF_cols - 1]
                if p > 0:
                    vals[nnzhF:o_jF] = -nls.hess_coord_cons(x, lam)
                    vals[o_jc:o_I] = Jcx[nls.jc_rows - 1, nls.jc_cols - 1]
                    vals[o_d:o_rho] = -delta
                vals[o_rho:] = 0.0
                rhs = np.concatenate([dual, primal])
                d, ok, rho, rho_old, nf = newton_system(LDLT, n, m, p, rhs, vals, rho_old, params)
                nfact += nf
                nlinsolve += 1
                if rho > params[6] or not ok or not np.all(np.isfinite(d)) or fx >= 1e60:
                    broken = True
                    break
                dlam = -d[n + m:]
            dx, dr = d[:n], d[n:n + m]
            if inner == 0:
                epsk = max(min(1e3 * delta, 99 * epsk / 100), 9 * epsk / 10)
                xt = x + dx
                rt = r + dr
                if np.linalg.norm(dlam) > 1e4:
                    dlam = dlam * 1e4 / np.linalg.norm(dlam)
                lamt = lam + dlam
                Ft = nls.residual(xt)
                ct = nls.cons(xt) if p else np.zeros(0)
            else:
                # line_search, src/CaNNOLeS.jl:1054-1112
                Dphi = (Jx.T @ Fx) @ dx - dx @ (Jcx.T @ (lam - cx / delta) if p else np.zeros(n))
                if p > 0:
                    eta = 1 / delta
                assert Dphi < 0
                xt = x + dx
                Ft = nls.residual(xt)
                ct = nls.cons(xt) if p else np.zeros(0)
                phix = phi(Fx, cx, lam, eta)
                phit = phi(Ft, ct, lam, eta)
                alpha = 1.0
                while not (phit <= phix + params[8] * alpha * Dphi):
                    nbk += 1
                    alpha /= 4
                    xt = x + alpha * dx
                    Ft = nls.residual(xt)
                    ct = nls.cons(xt) if p else np.zeros(0)
                    phit = phi(Ft, ct, lam, eta)
                    if alpha < eps ** 2:
                        raise RuntimeError("alpha too small")
                rt = Ft.copy()
                lamt = lam - cx / delta if p else lam.copy()
            Jt = nls.jac_residual(xt)
            Jct = nls.jac(xt) if p else np.zeros((0, n))
            Jxtr = Jt.T @ rt
            dual = Jxtr - Jct.T @ lamt
            primal = np.concatenate([Ft - rt, ct])
            normdualhat = np.linalg.norm(dual, np.inf)
            normprimalhat = np.linalg.norm(primal, np.inf)
            combined_hat = normdualhat + normprimalhat
            if inner > 0 or always_accept_extrapolation or combined_hat <= 0.99 * combined + epsk:
                x, r, Fx, cx, Jx, Jcx = xt.copy(), rt.copy(), Ft.copy(), ct.copy(), Jt, Jct
                fx = Fx @ Fx / 2
            if combined_hat <= 0.99 * combined + epsk:
                lam = lamt.copy()
            else:
                Jxtr = Jx.T @ r
                dual = Jxtr - Jcx.T @ lam
            if p > 0 and inner > 0 and normdualhat <= 0.99 * normdual + epsk / 2 and normprimalhat > 0.99 * normprimal + epsk / 2:
                delta = max(delta / 10, params[1])
            inner += 1
            tired = nls.neval > max_eval or inner > max_inner
        normdual, normprimal = normdualhat, normprimalhat
        first_order = max(normdual / dual_scaling(lam), normprimal) <= epstol
        small_residual = 2 * np.sqrt(fx) <= epsF and np.linalg.norm(cx) <= epsc
        if small_residual and not first_order:
            normprimal, normdual = small_res_check()
            first_order = max(normdual / dual_scaling(lam), normprimal) <= epstol
        it += 1
        st = status()
        if st == "unknown" and inner > max_inner >= 0:
            st = "stalled"
    return {"solution": x, "multipliers": lam, "status": st, "iter": it, "nfact": nfact, "nlinsolve": nlinsolve,
            "nbk": nbk, "objective": fx}
