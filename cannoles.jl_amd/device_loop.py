"""Device-resident batched outer loop: B problems of one closed-form family are solved in
LOCKSTEP with everything in HBM between iterations.

`solve_batch_device` restates `SolverCore.solve!` (/root/reference/src/CaNNOLeS.jl:418-864), the Armijo `line_search`
(:1054-1112) and `small_residual` handling for a BATCH: every quantity of the single-problem loop (outer_loop.solve, the
numpy mirror that is pinned to the reference's known answers) becomes a [B, ...] tensor and every branch a mask, so each
problem follows exactly the decisions it would take alone.  One "global step" is one pass of the inner `while` for every
active problem.  The linear algebra of the path goes through the C ABI on device pointers:

    prepare_newton_system!  -> cnl_prepare_newton_system_dev      (row a4 / f2)
    rhs = [dual; primal], norms -> cnl_residual_vectors_dev       (row f1)
    newton_system!          -> cnl_newton_system_dev               (the hot path)
    xt, rt, dlambda cap, lambdat -> cnl_trial_point_dev            (row f1)
    least-squares multipliers -> cnl_cgls_multipliers_dev          (row f4)

Only the model callbacks (residuals, Jacobian / Hessian values) are torch expressions of a closed-form family
(`BandQuadFamily`); `vals`, `rhs` and `d` never cross PCIe.  There is no CPU fallback.
"""
import numpy as np


class BandQuadFamily:
    """Closed-form constrained NLS family on the band structure of synthetic.band_structure (BASELINE configs 3 / 4):
        F_i(x) = sum_{j in band(i)} A_ij x_j + q_i x_i^2 / 2 - y_i,        c_k(x) = sum_{j in block k} C_kj x_j - e_k.
    Jacobian values in the structure's COO order: A + [i == j] q_i x_i and C; sum_i r_i Hess F_i = diag(q r) on the
    lower-band Hessian structure (off-diagonal slots are structural zeros); the constraints are linear (zero Hessian).
    `host_model(b)` is the numpy twin of problem b with the callbacks outer_loop.solve expects (an NLPModels-like object)."""

    def __init__(self, s, B, seed, torch, device, curvature=0.3, start=0.3, noise=0.01):
        self.s, self.B, self.torch, self.device = s, int(B), torch, device
        n, m, p = s.nvar, s.nequ, s.ncon
        rng = np.random.default_rng(seed)
        jr, jc = np.asarray(s.jF[0]) - 1, np.asarray(s.jF[1]) - 1
        self.jr, self.jc = jr, jc
        A = np.where(jr == jc, 2.0 + rng.uniform(0, 1, (B, len(jr))), rng.uniform(-0.5, 0.5, (B, len(jr))))
        q = rng.uniform(-curvature, curvature, (B, n))
        xs = rng.normal(size=(B, n))                      # a point near which the residual is small
        cr, cc = (np.asarray(s.jc[0]) - 1, np.asarray(s.jc[1]) - 1) if p else (np.zeros(0, int), np.zeros(0, int))
        self.cr, self.cc = cr, cc
        Cv = rng.uniform(-1, 1, (B, len(cr)))
        # padded per-row entry lists: sums over a short fixed dimension instead of atomics (deterministic)
        self.row_ent = self._row_lists(jr, m, len(jr))
        self.crow_ent = self._row_lists(cr, p, len(cr)) if p else None
        self.h = dict(A=A, q=q, Cv=Cv, x0=xs + start * rng.normal(size=(B, n)))
        self.h["y"] = self._F_np(A, q, xs, np.zeros((B, m))) + noise * rng.normal(size=(B, m))
        self.h["e"] = self._c_np(Cv, xs, np.zeros((B, p)))
        self.hr, self.hcl = np.asarray(s.hF[0]) - 1, np.asarray(s.hF[1]) - 1
        t = lambda a, dt=None: torch.as_tensor(a, dtype=dt or torch.float64, device=device)
        self.d = {k: t(v) for k, v in self.h.items()}
        self.jr_t, self.jc_t = t(jr, torch.long), t(jc, torch.long)
        self.cc_t = t(cc, torch.long)
        self.row_ent_t = t(self.row_ent, torch.long)
        self.crow_ent_t = t(self.crow_ent, torch.long) if p else None
        self.jdiag_t = t((jr == jc).astype(np.float64))
        self.hdiag_t = t((self.hr == self.hcl).astype(np.float64))
        self.hr_t = t(self.hr, torch.long)

    @staticmethod
    def _row_lists(rows, nrows, pad):
        cnt = np.bincount(rows, minlength=nrows) if len(rows) else np.zeros(nrows, int)
        K = int(cnt.max()) if nrows else 0
        out = np.full((nrows, max(K, 1)), pad, np.int64)
        fill = np.zeros(nrows, int)
        for e, r in enumerate(rows):
            out[r, fill[r]] = e
            fill[r] += 1
        return out

    # ---- numpy forms (data generation and the host twins)
    def _F_np(self, A, q, x, y):
        prod = np.concatenate([A * x[:, self.jc], np.zeros((len(x), 1))], axis=1)
        return prod[:, self.row_ent].sum(axis=2) + 0.5 * q * x * x - y

    def _c_np(self, Cv, x, e):
        if self.s.ncon == 0:
            return np.zeros((len(x), 0))
        prod = np.concatenate([Cv * x[:, self.cc], np.zeros((len(x), 1))], axis=1)
        return prod[:, self.crow_ent].sum(axis=2) - e

    # ---- device callbacks, batched: X [B, n] -> ...
    def residual(self, X):
        t = self.torch
        prod = t.cat([self.d["A"] * X[:, self.jc_t], t.zeros((self.B, 1), dtype=t.float64, device=self.device)], dim=1)
        return prod[:, self.row_ent_t].sum(dim=2) + 0.5 * self.d["q"] * X * X - self.d["y"]

    def jac_vals(self, X):
        return self.d["A"] + self.jdiag_t * (self.d["q"] * X)[:, self.jr_t]

    def hess_vals(self, X, R):
        return self.hdiag_t * (self.d["q"] * R)[:, self.hr_t]

    def cons(self, X):
        t = self.torch
        if self.s.ncon == 0:
            return t.zeros((self.B, 1), dtype=t.float64, device=self.device)
        prod = t.cat([self.d["Cv"] * X[:, self.cc_t], t.zeros((self.B, 1), dtype=t.float64, device=self.device)], dim=1)
        return prod[:, self.crow_ent_t].sum(dim=2) - self.d["e"]

    def jacc_vals(self, X):
        return self.d["Cv"]

    def host_model(self, b):
        fam, s = self, self.s
        n, m, p = s.nvar, s.nequ, s.ncon
        A, q, y, Cv, e = (fam.h[k][b] for k in ("A", "q", "y", "Cv", "e"))

        class M:
            nvar, nequ, ncon = n, m, p
            x0 = fam.h["x0"][b].copy()
            h_rows, h_cols = np.asarray(s.hF[0]), np.asarray(s.hF[1])
            jF_rows, jF_cols = np.asarray(s.jF[0]), np.asarray(s.jF[1])
            jc_rows, jc_cols = (np.asarray(s.jc[0]), np.asarray(s.jc[1])) if p else (np.zeros(0, np.int64), np.zeros(0, np.int64))
            neval = 0

            def residual(self, x):
                self.neval += 1
                return fam._F_np(A[None], q[None], x[None], y[None])[0]

            def jac_residual(self, x):
                J = np.zeros((m, n))
                J[fam.jr, fam.jc] = A + (fam.jr == fam.jc) * (q * x)[fam.jr]
                return J

            def hess_coord_residual(self, x, r):
                return (fam.hr == fam.hcl) * (q * r)[fam.hr]

            def cons(self, x):
                return fam._c_np(Cv[None], x[None], e[None])[0]

            def jac(self, x):
                J = np.zeros((p, n))
                J[fam.cr, fam.cc] = Cv
                return J

            def hess_coord_cons(self, x, lam):
                return np.zeros(len(self.h_rows))

        return M()


def kkt_pattern_of(fam):
    """rows, cols (1-based int64) and the segment offsets, exactly as outer_loop.solve builds them from a model
    (/root/reference/src/CaNNOLeS.jl:276-315): with constraints the H_c segment has the model's Hessian structure."""
    s = fam.s
    n, m, p = s.nvar, s.nequ, s.ncon
    hr, hc = np.asarray(s.hF[0]), np.asarray(s.hF[1])
    nnzhF, nnzhc = len(hr), (len(hr) if p > 0 else 0)
    jFr, jFc = np.asarray(s.jF[0]), np.asarray(s.jF[1])
    jcr, jcc = (np.asarray(s.jc[0]), np.asarray(s.jc[1])) if p else (np.zeros(0, np.int64), np.zeros(0, np.int64))
    N = n + m + p
    rows = np.concatenate([hr, hr[:nnzhc], jFr + n, jcr + n + m, np.arange(n + 1, n + m + 1), np.arange(n + m + 1, N + 1), np.arange(1, n + 1)]).astype(np.int64)
    cols = np.concatenate([hc, hc[:nnzhc], jFc, jcc, np.arange(n + 1, n + m + 1), np.arange(n + m + 1, N + 1), np.arange(1, n + 1)]).astype(np.int64)
    return rows, cols, (nnzhF, nnzhc, len(jFr), len(jcr))


PROFILE = False   # tools/time_device_loop.py --profile: wall time per section of a global step (synchronises at every section)


def solve_batch_device(fam, params=None, max_steps=400, max_inner=10000, atol=None, rtol=None, Fatol=None, Frtol=None, delta_dec=0.1,
                       device_index=0, layout="auto"):
    """All B problems of `fam` in lockstep on the device.  Returns a dict of numpy arrays: solution [B, n], multipliers,
    status (list of strings), iter, nfact, nlinsolve, nbk, objective, and `steps` (global steps = batched Newton rounds).

    Round 4: the masks and the masked state updates of a global step are FOUR device kernels working in place on the state
    (cnl_outer_begin_dev / _newton_done_dev / _trial_done_dev / _end_dev, csrc/outer_step.hip) instead of ~150 framework launches;
    the step costs two host reads of a few flag words.  The model callbacks, the Armijo line search and the rare small-residual
    branch are still framework expressions."""
    import ctypes as C
    from . import hipldl
    t = fam.torch
    dev = fam.device
    eps = float(np.finfo(float).eps)
    atol = np.sqrt(eps) if atol is None else atol
    rtol = np.sqrt(eps) if rtol is None else rtol
    Fatol = np.sqrt(eps) if Fatol is None else Fatol
    Frtol = eps if Frtol is None else Frtol
    params = hipldl.default_params() if params is None else np.ascontiguousarray(params, dtype=np.float64)
    dmin, rhomax, gammaA = float(params[1]), float(params[6]), float(params[8])
    s, B = fam.s, fam.B
    n, m, p = s.nvar, s.nequ, s.ncon
    N = n + m + p
    P = max(p, 1)
    rows, cols, (nnzhF, nnzhc, nnzjF, nnzjc) = kkt_pattern_of(fam)
    nnz = len(rows)
    # `vals` interleaved over groups of 32 problems where the band kernels serve the batch (cnl_options.batch_layout, round 6): row f2
    # writes that layout, newton_system reads it; rows f1 / f4 read the Jacobian values from the model's arrays, whatever the layout
    L = None
    if layout in ("auto", "interleaved"):
        try:
            L = hipldl.HIPLDLStruct(N, rows, cols, None, n, m, p, batch=B, device=device_index, options=hipldl.Options(batch_layout=hipldl.LAYOUT_INTERLEAVED))
        except hipldl.CnlError:
            if layout == "interleaved":
                raise
    if L is None:
        L = hipldl.HIPLDLStruct(N, rows, cols, None, n, m, p, batch=B, device=device_index)
    lib = hipldl.lib()
    f64 = dict(dtype=t.float64, device=dev)
    Z = lambda *sh: t.zeros(sh, **f64)
    ZI = lambda dt, *sh: t.zeros(sh, dtype=dt, device=dev)
    o_I = nnzhF + nnzhc + nnzjF + nnzjc

    st = t.cuda.current_stream(dev).cuda_stream
    ptr = lambda a: a.data_ptr()

    def new_vals():
        v = t.ones((B, nnz), **f64)
        v[:, o_I:o_I + m] = -1.0     # the -I block is set once (src/CaNNOLeS.jl:306); prepare never writes it
        if L.config.get("batch_layout"):
            vi = t.empty(hipldl.layout_len(L, 0), **f64)
            hipldl.interleave_dev(L, 0, ptr(v), ptr(vi), st)
            return vi
        return v

    vals_cur = new_vals()   # (ONE vals array: only newton_system reads it; the trial point's products read Jt)
    hc0 = Z(B, max(nnzhc, 1))

    def prepare(vals, hF, Jv_, Jcv_, delta_):
        hipldl.prepare_newton_system_dev(L, nnzhF, nnzhc, nnzjF, nnzjc, ptr(hF) if hF is not None else 0, ptr(hc0) if p else 0, ptr(Jv_),
                                         ptr(Jcv_) if p else 0, ptr(delta_) if p else 0, ptr(vals), st)

    def resid_vectors(Jv_, Jcv_, r_, lam_, F_, c_, rhs_out, nrm_out):
        """[dual; primal] = [Jx'r - Jc'lam; F - r; c] and the two infinity norms; the Jacobian values come from the model's arrays
        (cnl_residual_vectors_jac_dev: no prepare pass in front — round 6; rounds 2-5 copied them into `vals` first)"""
        hipldl.residual_vectors_jac_dev(L, nnzjF, nnzjc, ptr(Jv_), ptr(Jcv_) if p else 0, ptr(r_), ptr(lam_) if p else 0, ptr(F_), ptr(c_) if p else 0,
                                        ptr(rhs_out), ptr(nrm_out), st)

    def multipliers(Jv_, Jcv_, r_, ones_if_zero):
        lam_ = Z(B, P)
        if p:
            hipldl.cgls_multipliers_jac_dev(L, nnzjF, nnzjc, ptr(Jv_), ptr(Jcv_), ptr(r_), ptr(lam_), 0, None, None, 0, ones_if_zero, 0, st)
        return lam_

    W = lambda mask, a, b: t.where(mask if a.dim() == 1 else mask[:, None], a, b)
    smax = 100.0
    dual_scaling = lambda l_: (t.clamp(l_.abs().sum(dim=1) / p, min=smax) / smax) if p > 0 else t.ones(B, **f64)
    ninf = lambda a: a.abs().max(dim=1).values if a.shape[1] else Z(B)
    cnorm2 = lambda c_: t.sqrt((c_ * c_).sum(dim=1)) if p else Z(B)

    # ---- state (every array is updated IN PLACE from here on: the kernels hold its address) -------------------------
    x = fam.d["x0"].clone()
    Fx = fam.residual(x).contiguous()
    fx = (0.5 * (Fx * Fx).sum(dim=1)).contiguous()
    Jv = fam.jac_vals(x).contiguous()
    Jcv = fam.jacc_vals(x)          # the family's constraints are linear: one array serves the current and the trial point
    Jcv = Jcv.contiguous() if p else Z(B, 1)
    cx = fam.cons(x).contiguous()
    r = Fx.clone()
    delta = t.ones(B, **f64)
    lam = multipliers(Jv, Jcv, r, True)
    rhs_cur, nrm0 = Z(B, N), Z(B, 2)
    resid_vectors(Jv, Jcv, r, lam, Fx, cx, rhs_cur, nrm0)
    normdual, normprimal = nrm0[:, 0].clone(), nrm0[:, 1].clone()
    epsF = (Fatol + Frtol * 2 * t.sqrt(fx)).contiguous()
    epstol = (atol + rtol * normdual).contiguous()
    epsc = t.sqrt(epstol).contiguous()

    rv_rhs, rv_nrm = Z(B, N), Z(B, 2)

    def small_res_check(mask):
        """src/CaNNOLeS.jl:873-897 for the problems of `mask`: r = F, least-squares multipliers, dual, primal = [0; c] (in place)"""
        r2 = W(mask, Fx, r)
        lam2 = multipliers(Jv, Jcv, r2, False)
        lam.copy_(W(mask, lam2, lam))
        resid_vectors(Jv, Jcv, r2, lam, r2, cx, rv_rhs, rv_nrm)   # F - r = 0 for the masked problems
        rhs_cur.copy_(W(mask, rv_rhs, rhs_cur))
        normdual.copy_(W(mask, rv_nrm[:, 0], normdual))
        normprimal.copy_(W(mask, ninf(cx[:, :p]) if p else Z(B), normprimal))
        r.copy_(r2)

    small_residual = (2 * t.sqrt(fx) <= epsF) & (cnorm2(cx) <= epsc)
    first_order = t.maximum(normdual / dual_scaling(lam), normprimal) <= epstol
    chk0 = small_residual & ~first_order
    if bool(chk0.any()):
        small_res_check(chk0)
        first_order = t.maximum(normdual / dual_scaling(lam), normprimal) <= epstol
    UNKNOWN, FIRST, SMALL, EXC, TIRED, STALL = 0, 1, 2, 3, 4, 5
    status = t.where(first_order, FIRST, t.where(small_residual, SMALL, UNKNOWN)).to(t.int32).contiguous()
    eta = t.full((B,), 1.0 if p else 0.0, **f64)
    epsk = t.full((B,), 1e3, **f64)
    rho_old = Z(B)
    it, inner = ZI(t.int32, B), ZI(t.int64, B)
    nfact, nlin, nbk = ZI(t.int64, B), ZI(t.int64, B), ZI(t.int64, B)
    phase0 = t.ones(B, dtype=t.bool, device=dev)
    combined, combined_hat = Z(B), Z(B)
    ndh, nph = normdual.clone(), normprimal.clone()
    d = Z(B, N)
    xt, rt, lamt, Ft, ct = x.clone(), r.clone(), lam.clone(), Fx.clone(), cx.clone()
    Jt = Jv.clone()
    rhs_t, nrm_t = Z(B, N), Z(B, 2)
    d_new, rho_new, ro_tmp = Z(B, N), Z(B), Z(B)
    nf_new, ok_new = ZI(t.int32, B), ZI(t.int32, B)
    xt_e, rt_e, lamt_e, dlam_e = Z(B, n), Z(B, m), Z(B, P), Z(B, P)
    xl, Fl, cl, lam_ls = Z(B, n), Z(B, m), Z(B, P), Z(B, P)
    alpha, Dphi, phix = Z(B), Z(B), Z(B)
    masks = {k: t.zeros(B, dtype=t.bool, device=dev) for k in ("act", "need", "brk", "ext", "lsm", "rej", "chk", "done_in", "tired", "small_res", "bt")}
    flags = ZI(t.int32, 8)
    flags_h = t.zeros(8, dtype=t.int32).pin_memory()
    S = hipldl.cnl_outer_state()
    for k, v in dict(B=B, n=n, m=m, p=p, P=P, N=N, nnzjF=nnzjF, nnzjc=nnzjc, max_inner=max_inner, dmin=dmin, rhomax=rhomax,
                     delta_dec=delta_dec, smax=smax, gammaA=gammaA, eps2=eps ** 2).items():
        setattr(S, k, v)
    arrays = dict(status=status, it=it, flags=flags, nf_new=nf_new, ok_new=ok_new, inner=inner, nfact=nfact, nlin=nlin, phase0=phase0,
                  normdual=normdual, normprimal=normprimal, combined=combined, combined_hat=combined_hat, delta=delta, ndh=ndh, nph=nph, fx=fx,
                  epsk=epsk, epstol=epstol, epsF=epsF, epsc=epsc, rho_old=rho_old, d=d, d_new=d_new, ro_tmp=ro_tmp, rho_new=rho_new,
                  x=x, r=r, Fx=Fx, cx=cx, Jv=Jv, Jcv=Jcv, lam=lam, rhs_cur=rhs_cur, xt=xt, rt=rt, Ft=Ft, ct=ct, Jt=Jt, Jct=Jcv, lamt=lamt,
                  rhs_t=rhs_t, nrm_t=nrm_t, xt_e=xt_e, rt_e=rt_e, lamt_e=lamt_e, ls_g=rv_rhs, xl=xl, Fl=Fl, cl=cl, lam_ls=lam_ls, alpha=alpha,
                  Dphi=Dphi, phix=phix, eta=eta, nbk=nbk, **masks)
    for k, v in arrays.items():
        assert v.is_contiguous(), k
        setattr(S, k, v.data_ptr())
    Sref = C.byref(S)
    chk_ = hipldl._check

    def read_flags():
        flags_h.copy_(flags, non_blocking=True)
        t.cuda.current_stream(dev).synchronize()
        return flags_h.tolist()

    import time as _time
    prof = {} if PROFILE else None

    def tick(name, t0):
        if prof is not None:
            t.cuda.synchronize(dev)
            prof[name] = prof.get(name, 0.0) + (_time.perf_counter() - t0)
        return _time.perf_counter()

    t.cuda.synchronize(dev)
    t_loop0 = _time.perf_counter()
    steps = 0
    # host synchronisations per global step: one for the branch flags of cnl_outer_begin_dev, one for (rejected, small-residual) behind
    # cnl_outer_trial_done_dev, and one per round of backtracking when a line search runs
    while steps < max_steps:
        tk = _time.perf_counter()
        chk_(lib.cnl_outer_begin_dev(Sref, st))
        any_act, any_need, any_ext, any_ls = read_flags()[:4]
        if not any_act:
            break
        steps += 1
        tk = tick("begin", tk)
        # ---- Newton step (skipped on the iteration right after a rejected extrapolation), :627-652
        if any_need:
            prepare(vals_cur, fam.hess_vals(x, r), Jv, Jcv, delta)
            ro_tmp.copy_(rho_old)
            hipldl.newton_system_dev(L, ptr(vals_cur), ptr(rhs_cur), ptr(d_new), ptr(ro_tmp), ptr(rho_new), ptr(nf_new), ptr(ok_new), params, st)
        chk_(lib.cnl_outer_newton_done_dev(Sref, 1 if any_need else 0, st))
        dx = d[:, :n]
        tk = tick("newton", tk)
        # ---- extrapolation step, :654-668
        if any_ext:   # (a superset test: problems that broke above are masked out by `ext`)
            hipldl.trial_point_dev(L, ptr(x), ptr(r), ptr(lam) if p else 0, ptr(d), 1e4, ptr(xt_e), ptr(rt_e), ptr(lamt_e) if p else 0,
                                   ptr(dlam_e) if p else 0, st)
            chk_(lib.cnl_outer_extrapolated_dev(Sref, st))
        tk = tick("extrapolation", tk)
        # ---- Armijo line search on the merit function, :1054-1112
        if any_ls:
            resid_vectors(Jv, Jcv, Fx, lam_ls, Fx, cx, rv_rhs, rv_nrm)      # dual part: Jx'Fx - Jc'(lam - c/delta), lam_ls by cnl_outer_newton_done_dev
            chk_(lib.cnl_outer_ls_begin_dev(Sref, st))
            Fl.copy_(fam.residual(xl))
            cl.copy_(fam.cons(xl))
            chk_(lib.cnl_outer_ls_test_dev(Sref, 1, st))
            while read_flags()[6]:
                chk_(lib.cnl_outer_ls_step_dev(Sref, st))
                Fl.copy_(fam.residual(xl))       # (rows of problems that do not backtrack are recomputed from an unchanged xl: same values)
                cl.copy_(fam.cons(xl))
                chk_(lib.cnl_outer_ls_test_dev(Sref, 0, st))
            chk_(lib.cnl_outer_ls_take_dev(Sref, st))
        tk = tick("line_search", tk)
        Ft.copy_(fam.residual(xt))
        ct.copy_(fam.cons(xt))
        # ---- optimality measures at the trial point, :722-732; acceptance and the state update, :733-800
        Jt.copy_(fam.jac_vals(xt))
        resid_vectors(Jt, Jcv, rt, lamt, Ft, ct, rhs_t, nrm_t)
        tk = tick("trial_eval", tk)
        chk_(lib.cnl_outer_trial_done_dev(Sref, st))
        any_rej, any_chk = read_flags()[4:6]
        tk = tick("trial_done", tk)
        if any_rej:   # dual at (x, r, lam) again; primal keeps the trial's value, as in the reference (:742-747)
            resid_vectors(Jv, Jcv, r, lam, Fx, cx, rv_rhs, rv_nrm)
            rhs_cur[:, :n] = t.where(masks["rej"][:, None], rv_rhs[:, :n], rhs_cur[:, :n])
        if any_chk:
            small_res_check(masks["chk"])
        chk_(lib.cnl_outer_end_dev(Sref, st))
        tk = tick("rej_chk_end", tk)
    t.cuda.synchronize(dev)
    loop_seconds = _time.perf_counter() - t_loop0
    names = {UNKNOWN: "unknown", FIRST: "first_order", SMALL: "small_residual", EXC: "exception", TIRED: "max_eval", STALL: "stalled"}
    out = {"solution": x.cpu().numpy(), "multipliers": lam[:, :p].cpu().numpy(), "status": [names[int(v)] for v in status.cpu().numpy()],
           "iter": it.cpu().numpy(), "nfact": nfact.cpu().numpy(), "nlinsolve": nlin.cpu().numpy(), "nbk": nbk.cpu().numpy(),
           "objective": fx.cpu().numpy(), "steps": steps, "kernel": L.config["kernel"], "vals_layout": "interleaved" if L.config.get("batch_layout") else "problem-major",
           "loop_seconds": loop_seconds}   # the global steps alone (the symbolic analysis of the pattern and the start-up evaluations are not in it)
    if prof is not None:
        out["profile_ms_per_step"] = {k: 1e3 * v / max(steps, 1) for k, v in prof.items()}
    L.close()
    return out


def solve_batch_device_framework(fam, params=None, max_steps=400, max_inner=10000, atol=None, rtol=None, Fatol=None, Frtol=None, delta_dec=0.1,
                       device_index=0):
    """The loop as rounds 2-3 ran it: masks and masked state updates as ~150 framework launches per global step.  Kept as the
    executable restatement solve_batch_device is compared with (tests/test_gpu_parity.py).
    All B problems of `fam` in lockstep on the device.  Returns a dict of numpy arrays: solution [B, n], multipliers,
    status (list of strings), iter, nfact, nlinsolve, nbk, objective, and `steps` (global steps = batched Newton rounds)."""
    from . import hipldl
    t = fam.torch
    dev = fam.device
    eps = float(np.finfo(float).eps)
    atol = np.sqrt(eps) if atol is None else atol
    rtol = np.sqrt(eps) if rtol is None else rtol
    Fatol = np.sqrt(eps) if Fatol is None else Fatol
    Frtol = eps if Frtol is None else Frtol
    params = hipldl.default_params() if params is None else np.ascontiguousarray(params, dtype=np.float64)
    dmin, rhomax, gammaA = float(params[1]), float(params[6]), float(params[8])
    s, B = fam.s, fam.B
    n, m, p = s.nvar, s.nequ, s.ncon
    N = n + m + p
    P = max(p, 1)
    rows, cols, (nnzhF, nnzhc, nnzjF, nnzjc) = kkt_pattern_of(fam)
    nnz = len(rows)
    L = hipldl.HIPLDLStruct(N, rows, cols, None, n, m, p, batch=B, device=device_index)
    f64 = dict(dtype=t.float64, device=dev)
    Z = lambda *sh: t.zeros(sh, **f64)
    o_I = nnzhF + nnzhc + nnzjF + nnzjc

    def new_vals():
        v = t.ones((B, nnz), **f64)
        v[:, o_I:o_I + m] = -1.0     # the -I block is set once (src/CaNNOLeS.jl:306); prepare never writes it
        return v

    vals_cur, vals_t = new_vals(), new_vals()
    hc0 = Z(B, max(nnzhc, 1))
    st = t.cuda.current_stream(dev).cuda_stream
    ptr = lambda a: a.data_ptr()

    def prepare(vals, hF, Jv, Jcv, delta):
        hipldl.prepare_newton_system_dev(L, nnzhF, nnzhc, nnzjF, nnzjc, ptr(hF) if hF is not None else 0, ptr(hc0) if p else 0, ptr(Jv),
                                         ptr(Jcv) if p else 0, ptr(delta) if p else 0, ptr(vals), st)

    rv_rhs, rv_nrm = Z(B, N), Z(B, 2)

    def resid_vectors(vals, r_, lam_, F_, c_):
        """[dual; primal] = [Jx'r - Jc'lam; F - r; c] and the two infinity norms, from the J segments of `vals`"""
        rv_nrm.zero_()
        hipldl.residual_vectors_dev(L, ptr(vals), ptr(r_), ptr(lam_) if p else 0, ptr(F_), ptr(c_) if p else 0, ptr(rv_rhs), ptr(rv_nrm), st)
        return rv_rhs.clone(), rv_nrm[:, 0].clone(), rv_nrm[:, 1].clone()

    def multipliers(vals, r_, ones_if_zero):
        lam_ = Z(B, P)
        if p:
            hipldl.cgls_multipliers_dev(L, ptr(vals), ptr(r_), ptr(lam_), 0, None, None, 0, ones_if_zero, 0, st)
        return lam_

    W = lambda mask, a, b: t.where(mask if a.dim() == 1 else mask[:, None], a, b)
    smax = 100.0
    dual_scaling = lambda l_: (t.clamp(l_.abs().sum(dim=1) / p, min=smax) / smax) if p > 0 else t.ones(B, **f64)
    ninf = lambda a: a.abs().max(dim=1).values if a.shape[1] else Z(B)

    # ---- start, src/CaNNOLeS.jl:470-560
    x = fam.d["x0"].clone()
    Fx = fam.residual(x)
    fx = 0.5 * (Fx * Fx).sum(dim=1)
    Jv, Jcv = fam.jac_vals(x), fam.jacc_vals(x)
    cx = fam.cons(x)
    r = Fx.clone()
    delta = t.ones(B, **f64)
    prepare(vals_cur, None, Jv, Jcv, delta)
    lam = multipliers(vals_cur, r, True)
    rhs_cur, normdual, normprimal = resid_vectors(vals_cur, r, lam, Fx, cx)
    epsF = Fatol + Frtol * 2 * t.sqrt(fx)
    epstol = atol + rtol * normdual
    epsc = t.sqrt(epstol)
    cnorm2 = lambda c_: t.sqrt((c_ * c_).sum(dim=1)) if p else Z(B)

    def small_res_check(mask, lam, rhs_cur, normdual, normprimal, r):
        """src/CaNNOLeS.jl:873-897 for the problems of `mask`: r = F, least-squares multipliers, dual, primal = [0; c]"""
        r2 = W(mask, Fx, r)
        prepare(vals_cur, None, Jv, Jcv, delta)
        lam2 = multipliers(vals_cur, r2, False)
        lam_n = W(mask, lam2, lam)
        rhs2, nd2, _ = resid_vectors(vals_cur, r2, lam_n, r2, cx)   # F - r = 0 for the masked problems
        rhs_n = W(mask, rhs2, rhs_cur)
        return lam_n, rhs_n, W(mask, nd2, normdual), W(mask, ninf(cx[:, :p]) if p else Z(B), normprimal), r2

    small_residual = (2 * t.sqrt(fx) <= epsF) & (cnorm2(cx) <= epsc)
    first_order = t.maximum(normdual / dual_scaling(lam), normprimal) <= epstol
    chk = small_residual & ~first_order
    if bool(chk.any()):
        lam, rhs_cur, normdual, normprimal, r = small_res_check(chk, lam, rhs_cur, normdual, normprimal, r)
        first_order = t.maximum(normdual / dual_scaling(lam), normprimal) <= epstol
    UNKNOWN, FIRST, SMALL, EXC, TIRED, STALL = 0, 1, 2, 3, 4, 5
    status = t.where(first_order, FIRST, t.where(small_residual, SMALL, UNKNOWN)).to(t.int32)
    eta = t.full((B,), 1.0 if p else 0.0, **f64)
    epsk = t.full((B,), 1e3, **f64)
    rho_old = Z(B)
    it = t.zeros(B, dtype=t.int32, device=dev)
    inner = t.zeros(B, dtype=t.int64, device=dev)
    nfact = t.zeros(B, dtype=t.int64, device=dev)
    nlin = t.zeros(B, dtype=t.int64, device=dev)
    nbk = t.zeros(B, dtype=t.int64, device=dev)
    phase0 = t.ones(B, dtype=t.bool, device=dev)
    combined, combined_hat = Z(B), Z(B)
    ndh, nph = normdual.clone(), normprimal.clone()
    d = Z(B, N)
    xt, rt, lamt, Ft, ct = x.clone(), r.clone(), lam.clone(), Fx.clone(), cx.clone()
    d_new, rho_new = Z(B, N), Z(B)
    nf_new = t.zeros(B, dtype=t.int32, device=dev)
    ok_new = t.zeros(B, dtype=t.int32, device=dev)
    xt_e, rt_e, lamt_e, dlam_e = Z(B, n), Z(B, m), Z(B, P), Z(B, P)
    phi = lambda F_, c_, l_, et: 0.5 * (F_ * F_).sum(dim=1) - ((l_ * c_).sum(dim=1) if p else 0.0) + (et * (c_ * c_).sum(dim=1) / 2 if p else 0.0)
    steps = 0
    # host synchronisations per global step: one for the branch flags below, one for (rejected, small-residual) further down,
    # and one per round of backtracking when a line search runs
    while steps < max_steps:
        act = status == UNKNOWN
        if not bool(act.any()):
            break
        steps += 1
        # ---- start of an outer iteration, src/CaNNOLeS.jl:612-626
        so = act & phase0
        combined = W(so, normdual + normprimal, combined)
        delta = W(so, t.clamp(t.minimum(delta_dec * delta, combined), min=dmin), delta)
        inner = t.where(so, 0, inner)
        combined_hat = W(so, t.full_like(combined, float("inf")), combined_hat)
        ndh, nph = W(so, normdual, ndh), W(so, normprimal, nph)
        phase0 = phase0 & ~so
        # ---- Newton step (skipped on the iteration right after a rejected extrapolation), :627-652
        need = act & (inner != 1)
        brk = t.zeros(B, dtype=t.bool, device=dev)
        any_need, any_ext, any_ls = t.stack([need.any(), (act & (inner == 0)).any(), (act & (inner > 0)).any()]).tolist()
        if any_need:
            prepare(vals_cur, fam.hess_vals(x, r), Jv, Jcv, delta)
            ro_tmp = rho_old.clone()
            hipldl.newton_system_dev(L, ptr(vals_cur), ptr(rhs_cur), ptr(d_new), ptr(ro_tmp), ptr(rho_new), ptr(nf_new), ptr(ok_new), params, st)
            d = W(need, d_new, d)
            rho_old = W(need, ro_tmp, rho_old)
            nfact = nfact + t.where(need, nf_new.to(t.int64), 0)
            nlin = nlin + need.to(t.int64)
            # `broken`: the inner loop is left at once; the end-of-iteration tests below still run for it (:638-652)
            brk = need & ((rho_new > rhomax) | (ok_new == 0) | ~t.isfinite(d_new).all(dim=1) | (fx >= 1e60))
            act = act & ~brk
        dx = d[:, :n]
        ext, lsm = act & (inner == 0), act & (inner > 0)
        # ---- extrapolation step, :654-668
        if any_ext:   # (a superset test: problems that broke above are masked out by `ext`)
            epsk = W(ext, t.maximum(t.minimum(1e3 * delta, 99 * epsk / 100), 9 * epsk / 10), epsk)
            hipldl.trial_point_dev(L, ptr(x), ptr(r), ptr(lam) if p else 0, ptr(d), 1e4, ptr(xt_e), ptr(rt_e), ptr(lamt_e) if p else 0,
                                   ptr(dlam_e) if p else 0, st)
            xt, rt, lamt = W(ext, xt_e, xt), W(ext, rt_e, rt), W(ext, lamt_e, lamt)
        # ---- Armijo line search on the merit function, :1054-1112
        if any_ls:
            lam_ls = lam - cx / delta[:, None] if p else lam
            prepare(vals_cur, None, Jv, Jcv, delta)
            g, _, _ = resid_vectors(vals_cur, Fx, lam_ls, Fx, cx)      # dual part: Jx'Fx - Jc'(lam - c/delta)
            Dphi = (g[:, :n] * dx).sum(dim=1)
            if p:
                eta = W(lsm, 1.0 / delta, eta)
            phix = phi(Fx, cx, lam, eta)
            alpha = t.ones(B, **f64)
            xl = x + dx
            Fl, cl = fam.residual(xl), fam.cons(xl)
            bt = lsm & ~(phi(Fl, cl, lam, eta) <= phix + gammaA * alpha * Dphi)
            while bool(bt.any()):
                nbk = nbk + bt.to(t.int64)
                alpha = W(bt, alpha / 4, alpha)
                xl = W(bt, x + alpha[:, None] * dx, xl)
                F2, c2 = fam.residual(xl), fam.cons(xl)
                Fl, cl = W(bt, F2, Fl), W(bt, c2, cl)
                bt = bt & ~(phi(Fl, cl, lam, eta) <= phix + gammaA * alpha * Dphi) & (alpha >= eps ** 2)
            xt, rt = W(lsm, xl, xt), W(lsm, Fl, rt)
            lamt = W(lsm, lam_ls, lamt)
        Ft, ct = fam.residual(xt), fam.cons(xt)
        # ---- optimality measures at the trial point, :722-732
        Jt, Jct = fam.jac_vals(xt), fam.jacc_vals(xt)
        prepare(vals_t, None, Jt, Jct, delta)
        rhs_t, nd_t, np_t = resid_vectors(vals_t, rt, lamt, Ft, ct)
        ndh, nph = W(act, nd_t, ndh), W(act, np_t, nph)
        combined_hat = W(act, ndh + nph, combined_hat)
        good = combined_hat <= 0.99 * combined + epsk
        acc_state = act & ((inner > 0) | good)
        x, r, Fx, cx = W(acc_state, xt, x), W(acc_state, rt, r), W(acc_state, Ft, Fx), W(acc_state, ct, cx)
        Jv, Jcv = W(acc_state, Jt, Jv), W(acc_state, Jct, Jcv)
        fx = W(acc_state, 0.5 * (Ft * Ft).sum(dim=1), fx)
        acc_lam = act & good
        lam = W(acc_lam, lamt, lam)
        rhs_cur = W(act, rhs_t, rhs_cur)
        rej = act & ~good
        delta_next = delta
        if p:
            dr_ = act & (inner > 0) & (ndh <= 0.99 * normdual + epsk / 2) & (nph > 0.99 * normprimal + epsk / 2)
            delta_next = W(dr_, t.clamp(delta / 10, min=dmin), delta)
        inner = inner + act.to(t.int64)
        tired = inner > max_inner
        # ---- end of the inner loop -> end of the outer iteration, :765-800 (tests first: one synchronisation for both branches)
        done_in = (act & (good | tired)) | brk
        normdual, normprimal = W(done_in, ndh, normdual), W(done_in, nph, normprimal)
        first_order = t.maximum(normdual / dual_scaling(lam), normprimal) <= epstol
        small_residual = (2 * t.sqrt(fx) <= epsF) & (cnorm2(cx) <= epsc)
        chk = done_in & small_residual & ~first_order
        any_rej, any_chk = t.stack([rej.any(), chk.any()]).tolist()
        if any_rej:   # dual at (x, r, lam) again; primal keeps the trial's value, as in the reference (:742-747)
            prepare(vals_cur, None, Jv, Jcv, delta)
            rhs_r, _, _ = resid_vectors(vals_cur, r, lam, Fx, cx)
            rhs_cur = t.cat([W(rej, rhs_r[:, :n], rhs_cur[:, :n]), rhs_cur[:, n:]], dim=1)
        delta = delta_next
        if any_chk:
            lam, rhs_cur, normdual, normprimal, r = small_res_check(chk, lam, rhs_cur, normdual, normprimal, r)
            first_order = t.maximum(normdual / dual_scaling(lam), normprimal) <= epstol
        it = it + done_in.to(t.int32)
        new_status = t.where(first_order, FIRST, t.where(small_residual, SMALL, t.where(brk, EXC, t.where(tired, STALL, UNKNOWN))))   # inner > max_inner: `stalled` (src/CaNNOLeS.jl:846)
        status = t.where(done_in, new_status.to(t.int32), status)
        phase0 = phase0 | done_in
    t.cuda.synchronize(dev)
    names = {UNKNOWN: "unknown", FIRST: "first_order", SMALL: "small_residual", EXC: "exception", TIRED: "max_eval", STALL: "stalled"}
    out = {"solution": x.cpu().numpy(), "multipliers": lam[:, :p].cpu().numpy(), "status": [names[int(v)] for v in status.cpu().numpy()],
           "iter": it.cpu().numpy(), "nfact": nfact.cpu().numpy(), "nlinsolve": nlin.cpu().numpy(), "nbk": nbk.cpu().numpy(),
           "objective": fx.cpu().numpy(), "steps": steps, "kernel": L.config["kernel"]}
    L.close()
    return out
