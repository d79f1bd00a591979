/*
 * cnl_oracle.c — CPU restatement of the CaNNOLeS inner Newton step.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under the product package may import,
 * link or call this file; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, as the checker / reported baseline.
 *
 * What it restates (citations are into /root/reference):
 *   - KKT COO pattern, 7 segments [H_F | H_c | J_F | J_c | -I | -dI | rI]
 *       src/CaNNOLeS.jl:256-315
 *   - per-iteration value fill (prepare_newton_system!)   src/CaNNOLeS.jl:947-981
 *   - LDLFactStruct ctor: Symmetric(triu(sparse(cols,rows,vals,N,N)),:U)
 *       + ldl_analyze                                      src/solver_types.jl:61-65
 *   - set_vals!: zero, then per-entry += in COO order      src/solver_types.jl:53-59
 *   - try_to_factorize: set_vals!, ldl_factorize!, inertia src/solver_types.jl:79-98
 *   - solve_ldl!: ldiv! then negate                        src/solver_types.jl:69-77
 *   - newton_system!: rho ladder                           src/CaNNOLeS.jl:1008-1052
 *   - ParamCaNNOLeS defaults                               src/CaNNOLeS.jl:48-62
 *
 * The factorisation arithmetic itself is NOT in the reference tree: it lives
 * in the un-vendored dependency LDLFactorizations.jl (compat "0.10",
 * Project.toml:20; no Manifest, so no exact pin), a Julia translation of
 * Tim Davis' LDL package.  This file restates that published algorithm
 * (ldl_symbolic / ldl_numeric / ldl_lsolve / ldl_dsolve / ldl_ltsolve /
 * ldl_perm / ldl_permt: up-looking, no pivoting, no supernodes).  The
 * fill-reducing ordering of the reference (SuiteSparse AMD through AMD.jl)
 * is not reproducible here; the permutation is an INPUT of cnlo_create.
 * LDL^T results (d, inertia, success) are permutation independent up to
 * round-off as long as no exact zero pivot is hit.
 *
 * Parity pinning: the reference cannot run here (Julia; no julia binary) and
 * its tests hold no golden vector at the try_to_factorize/solve_ldl!/
 * newton_system! boundary.  The oracle is pinned (tests/test_oracle_*.py)
 * against (a) fixtures hand-derived from the reference's own test model
 * test/mgh01con.jl, (b) the end-to-end solutions asserted in
 * test/runtests.jl:65-100,116-214 through a restated outer loop, and
 * (c) independent dense numpy/scipy solves.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef int64_t i64;

/* ------------------------------------------------------------------ */
/* ParamCaNNOLeS(Float64) defaults — src/CaNNOLeS.jl:48-62             */
/* order: eig_tol, dmin, kdec, kinc, klargeinc, rho0, rhomax, rhomin, gammaA */
void cnlo_default_params(double *p) {
  const double eps = 2.220446049250313e-16;
  p[0] = eps;                 /* eig_tol   = eps            */
  p[1] = sqrt(eps);           /* delta_min = sqrt(eps)      */
  p[2] = 1.0 / 3.0;           /* kappa_dec = 1//3           */
  p[3] = 8.0;                 /* kappa_inc = 8              */
  p[4] = fmin(100.0, 8 * 16); /* kappa_largeinc = min(100, sizeof(T)*16) */
  p[5] = pow(eps, 1.0 / 3.0); /* rho0      = eps^T(1/3): pow with exponent 0.333..., not cbrt (src/CaNNOLeS.jl:56) */
  p[6] = pow(eps, -2.0);      /* rho_max   = eps^-2         */
  p[7] = sqrt(eps);           /* rho_min   = sqrt(eps)      */
  p[8] = pow(eps, 0.25);      /* gamma_A   = eps^(1/4)      */
}

/* ------------------------------------------------------------------ */
/* KKT pattern — src/CaNNOLeS.jl:256-315.  All indices 1-based, as the
 * reference holds them.  Returns nnzNS; rows/cols/vals must have room for
 * nnzhF+nnzhc+nnzjF+nnzjc+nvar+nequ+ncon entries.  vals is initialised as
 * the ctor does (1 everywhere, -1 on the -I segment).                  */
i64 cnlo_kkt_pattern(i64 nvar, i64 nequ, i64 ncon,
                     i64 nnzhF, const i64 *hF_rows, const i64 *hF_cols,
                     i64 nnzhc, const i64 *hc_rows, const i64 *hc_cols,
                     i64 nnzjF, const i64 *jF_rows, const i64 *jF_cols,
                     i64 nnzjc, const i64 *jc_rows, const i64 *jc_cols,
                     i64 *rows, i64 *cols, double *vals) {
  if (ncon == 0) { nnzhc = 0; nnzjc = 0; } /* :256, :289, :298 */
  i64 nnzNS = nnzhF + nnzhc + nnzjF + nnzjc + nvar + nequ + ncon; /* :273 */
  i64 k = 0, i;
  for (i = 0; i < nnzNS; i++) vals[i] = 1.0;                      /* :279 */
  for (i = 0; i < nnzhF; i++, k++) { rows[k] = hF_rows[i]; cols[k] = hF_cols[i]; }
  for (i = 0; i < nnzhc; i++, k++) { rows[k] = hc_rows[i]; cols[k] = hc_cols[i]; }
  for (i = 0; i < nnzjF; i++, k++) { rows[k] = jF_rows[i] + nvar; cols[k] = jF_cols[i]; }
  for (i = 0; i < nnzjc; i++, k++) { rows[k] = jc_rows[i] + nvar + nequ; cols[k] = jc_cols[i]; }
  for (i = 0; i < nequ; i++, k++) { rows[k] = cols[k] = nvar + 1 + i; vals[k] = -1.0; } /* :304-306 */
  for (i = 0; i < ncon; i++, k++) { rows[k] = cols[k] = nvar + nequ + 1 + i; }
  for (i = 0; i < nvar; i++, k++) { rows[k] = cols[k] = 1 + i; }
  return nnzNS;
}

/* prepare_newton_system! — src/CaNNOLeS.jl:947-981.
 * hF_vals: hess_coord_residual!(x, r) (may be NULL => segment untouched, the
 *          Newton_noFHess / skipped Newton_vanishing case, hessian_approx.jl:48-60)
 * hc_vals: hess_coord!(x, lambda; obj_weight=0); stored NEGATED (:971-972).   */
void cnlo_prepare(i64 nvar, i64 nequ, i64 ncon, i64 nnzhF, i64 nnzhc, i64 nnzjF, i64 nnzjc,
                  const double *hF_vals, const double *hc_vals, const double *jF_vals,
                  const double *jc_vals, double delta, double *vals) {
  i64 i, o;
  if (ncon == 0) { nnzhc = 0; nnzjc = 0; }
  if (hF_vals) for (i = 0; i < nnzhF; i++) vals[i] = hF_vals[i];
  o = nnzhF + nnzhc;
  for (i = 0; i < nnzjF; i++) vals[o + i] = jF_vals[i];           /* :967-968 */
  if (ncon > 0) {
    o = nnzhF;
    for (i = 0; i < nnzhc; i++) vals[o + i] = -hc_vals[i];        /* :970-972 */
    o = nnzhF + nnzhc + nnzjF;
    for (i = 0; i < nnzjc; i++) vals[o + i] = jc_vals[i];         /* :973-974 */
    o = nnzhF + nnzhc + nnzjF + nnzjc + nequ;
    for (i = 0; i < ncon; i++) vals[o + i] = -delta;              /* :975-976 */
  }
  o = nnzhF + nnzhc + nnzjF + nnzjc + nequ + ncon;
  for (i = 0; i < nvar; i++) vals[o + i] = 0.0;                   /* :978-979 */
}

/* ------------------------------------------------------------------ */
typedef struct {
  i64 N, nnz;      /* order, number of COO entries                         */
  i64 *rows, *cols;/* 0-based copies of the COO structure (lower triangle) */
  /* A = Symmetric(triu(sparse(cols, rows, vals)), :U): upper CSC, merged  */
  i64 *Ap, *Ai;    /* colptr (N+1), rowval (nnzA), sorted within a column  */
  double *Ax;
  i64 nnzA;
  i64 *map;        /* COO entry k -> slot in Ax (scatter map)              */
  /* full symmetric pattern (both triangles) used by the permuted LDL      */
  i64 *Fp, *Fi, *Fsrc; /* Fsrc: slot of Ax that holds the value            */
  /* LDL (Davis) workspace                                                 */
  i64 *P, *Pinv, *Parent, *Lnz, *Lp, *Li, *Flag, *Pattern;
  double *Lx, *D, *Y;
  i64 nnzL;
  i64 factor_rank; /* return value of the last ldl_numeric                 */
} cnlo;

static int cmp_i64(const void *a, const void *b) {
  i64 x = *(const i64 *)a, y = *(const i64 *)b;
  return (x > y) - (x < y);
}

void cnlo_destroy(cnlo *h) {
  if (!h) return;
  free(h->rows); free(h->cols); free(h->Ap); free(h->Ai); free(h->Ax); free(h->map);
  free(h->Fp); free(h->Fi); free(h->Fsrc);
  free(h->P); free(h->Pinv); free(h->Parent); free(h->Lnz); free(h->Lp); free(h->Li);
  free(h->Flag); free(h->Pattern); free(h->Lx); free(h->D); free(h->Y);
  free(h);
}

/* ldl_symbolic of Davis' LDL package applied to the permuted full pattern
 * (what LDLFactorizations.ldl_analyze does after the ordering).          */
static void ldl_symbolic(cnlo *h) {
  i64 n = h->N, k, p, i, kk;
  for (k = 0; k < n; k++) {
    h->Parent[k] = -1; h->Flag[k] = k; h->Lnz[k] = 0;
    kk = h->P[k];
    for (p = h->Fp[kk]; p < h->Fp[kk + 1]; p++) {
      i = h->Pinv[h->Fi[p]];
      if (i < k) {
        for (; h->Flag[i] != k; i = h->Parent[i]) {
          if (h->Parent[i] == -1) h->Parent[i] = k;
          h->Lnz[i]++;
          h->Flag[i] = k;
        }
      }
    }
  }
  h->Lp[0] = 0;
  for (k = 0; k < n; k++) h->Lp[k + 1] = h->Lp[k] + h->Lnz[k];
}

/* LDLFactStruct(N, rows, cols, vals) — src/solver_types.jl:61-65.
 * rows1/cols1: 1-based COO, lower triangle (rows >= cols) as built by the
 * pattern above.  perm: 0-based elimination order (perm[k] = original index
 * eliminated k-th), or NULL for the natural order.  Returns NULL on a
 * malformed pattern.                                                     */
cnlo *cnlo_create(i64 N, i64 nnz, const i64 *rows1, const i64 *cols1, const i64 *perm) {
  cnlo *h = (cnlo *)calloc(1, sizeof(cnlo));
  i64 k, j, p;
  h->N = N; h->nnz = nnz;
  h->rows = (i64 *)malloc(sizeof(i64) * (nnz + 1));
  h->cols = (i64 *)malloc(sizeof(i64) * (nnz + 1));
  for (k = 0; k < nnz; k++) {
    h->rows[k] = rows1[k] - 1; h->cols[k] = cols1[k] - 1;
    if (h->rows[k] < 0 || h->rows[k] >= N || h->cols[k] < 0 || h->cols[k] >= N) { cnlo_destroy(h); return NULL; }
  }
  /* sparse(cols, rows, vals): entry k sits at (row=cols[k], col=rows[k]);
   * triu keeps row <= col, i.e. COO entries with cols[k] <= rows[k] (all of
   * them for a lower-triangular COO; the others are dropped as triu does). */
  i64 *key = (i64 *)malloc(sizeof(i64) * (nnz + 1));
  i64 nk = 0;
  for (k = 0; k < nnz; k++)
    if (h->cols[k] <= h->rows[k]) key[nk++] = h->rows[k] * N + h->cols[k]; /* col-major key */
  qsort(key, nk, sizeof(i64), cmp_i64);
  i64 nu = 0;
  for (k = 0; k < nk; k++) if (k == 0 || key[k] != key[k - 1]) key[nu++] = key[k];
  h->nnzA = nu;
  h->Ap = (i64 *)calloc(N + 1, sizeof(i64));
  h->Ai = (i64 *)malloc(sizeof(i64) * (nu + 1));
  h->Ax = (double *)calloc(nu + 1, sizeof(double));
  for (k = 0; k < nu; k++) { h->Ap[key[k] / N + 1]++; h->Ai[k] = key[k] % N; }
  for (j = 0; j < N; j++) h->Ap[j + 1] += h->Ap[j];
  /* scatter map by binary search (what getindex/setindex! do per entry)   */
  h->map = (i64 *)malloc(sizeof(i64) * (nnz + 1));
  for (k = 0; k < nnz; k++) {
    if (h->cols[k] > h->rows[k]) { h->map[k] = -1; continue; }
    i64 kk = h->rows[k] * N + h->cols[k];
    i64 lo = 0, hi = nu - 1;
    while (lo < hi) { i64 mid = (lo + hi) / 2; if (key[mid] < kk) lo = mid + 1; else hi = mid; }
    h->map[k] = lo;
  }
  free(key);
  /* full symmetric pattern */
  h->Fp = (i64 *)calloc(N + 2, sizeof(i64));
  for (j = 0; j < N; j++)
    for (p = h->Ap[j]; p < h->Ap[j + 1]; p++) {
      h->Fp[j + 1]++;
      if (h->Ai[p] != j) h->Fp[h->Ai[p] + 1]++;
    }
  for (j = 0; j < N; j++) h->Fp[j + 1] += h->Fp[j];
  i64 nf = h->Fp[N];
  h->Fi = (i64 *)malloc(sizeof(i64) * (nf + 1));
  h->Fsrc = (i64 *)malloc(sizeof(i64) * (nf + 1));
  i64 *nxt = (i64 *)malloc(sizeof(i64) * (N + 1));
  for (j = 0; j < N; j++) nxt[j] = h->Fp[j];
  for (j = 0; j < N; j++)
    for (p = h->Ap[j]; p < h->Ap[j + 1]; p++) {
      i64 i = h->Ai[p];
      h->Fi[nxt[j]] = i; h->Fsrc[nxt[j]++] = p;
      if (i != j) { h->Fi[nxt[i]] = j; h->Fsrc[nxt[i]++] = p; }
    }
  free(nxt);
  /* ordering + symbolic */
  h->P = (i64 *)malloc(sizeof(i64) * (N + 1));
  h->Pinv = (i64 *)malloc(sizeof(i64) * (N + 1));
  for (k = 0; k < N; k++) h->Pinv[k] = -1;
  for (k = 0; k < N; k++) {
    h->P[k] = perm ? perm[k] : k;
    if (h->P[k] < 0 || h->P[k] >= N || h->Pinv[h->P[k]] != -1) { cnlo_destroy(h); return NULL; }
    h->Pinv[h->P[k]] = k;
  }
  h->Parent = (i64 *)malloc(sizeof(i64) * (N + 1));
  h->Lnz = (i64 *)malloc(sizeof(i64) * (N + 1));
  h->Lp = (i64 *)malloc(sizeof(i64) * (N + 2));
  h->Flag = (i64 *)malloc(sizeof(i64) * (N + 1));
  h->Pattern = (i64 *)malloc(sizeof(i64) * (N + 1));
  h->D = (double *)calloc(N + 1, sizeof(double));
  h->Y = (double *)calloc(N + 1, sizeof(double));
  ldl_symbolic(h);
  h->nnzL = h->Lp[N];
  h->Li = (i64 *)malloc(sizeof(i64) * (h->nnzL + 1));
  h->Lx = (double *)malloc(sizeof(double) * (h->nnzL + 1));
  h->factor_rank = 0;
  return h;
}

i64 cnlo_nnzL(const cnlo *h) { return h->nnzL; }
i64 cnlo_nnzA(const cnlo *h) { return h->nnzA; }
/* sum over columns of Lnz^2: the FMA count of the numeric phase */
double cnlo_flops(const cnlo *h) {
  double s = 0; i64 k;
  for (k = 0; k < h->N; k++) { i64 c = h->Lp[k + 1] - h->Lp[k]; s += (double)c * (double)c; }
  return s;
}
const double *cnlo_D(const cnlo *h) { return h->D; }
const double *cnlo_Ax(const cnlo *h) { return h->Ax; }
const i64 *cnlo_Ap(const cnlo *h) { return h->Ap; }
const i64 *cnlo_Ai(const cnlo *h) { return h->Ai; }

/* set_vals! — src/solver_types.jl:53-59: nzval .= 0; A.data[cols[i],rows[i]] += vals[i]
 * in COO order (so the floating-point sum of duplicates is in COO order).
 * mode 0: precomputed scatter map; mode 1: per-entry binary search in the
 * column, which is what SparseMatrixCSC getindex+setindex! pay.           */
void cnlo_set_vals(cnlo *h, const double *vals, int mode) {
  i64 k;
  memset(h->Ax, 0, sizeof(double) * h->nnzA);
  if (mode == 0) {
    for (k = 0; k < h->nnz; k++) if (h->map[k] >= 0) h->Ax[h->map[k]] += vals[k];
  } else {
    for (k = 0; k < h->nnz; k++) {
      i64 i = h->cols[k], j = h->rows[k]; /* A.data[cols, rows] */
      if (i > j) continue;
      i64 t;
      for (t = 0; t < 2; t++) { /* getindex then setindex!: two searches */
        i64 lo = h->Ap[j], hi = h->Ap[j + 1] - 1;
        while (lo < hi) { i64 mid = (lo + hi) / 2; if (h->Ai[mid] < i) lo = mid + 1; else hi = mid; }
        if (t == 1) h->Ax[lo] += vals[k];
      }
    }
  }
}

/* ldl_numeric (Davis) on P A P^T, as LDLFactorizations.ldl_factorize! does.
 * Returns N on completion or the index k of the first exactly-zero pivot. */
static i64 ldl_numeric(cnlo *h) {
  i64 n = h->N, k, p, i, len, top, kk, p2;
  double yi, lki;
  for (k = 0; k < n; k++) {
    h->Y[k] = 0.0; top = n; h->Flag[k] = k; h->Lnz[k] = 0;
    kk = h->P[k];
    for (p = h->Fp[kk]; p < h->Fp[kk + 1]; p++) {
      i = h->Pinv[h->Fi[p]];
      if (i <= k) {
        h->Y[i] += h->Ax[h->Fsrc[p]];
        for (len = 0; h->Flag[i] != k; i = h->Parent[i]) { h->Pattern[len++] = i; h->Flag[i] = k; }
        while (len > 0) h->Pattern[--top] = h->Pattern[--len];
      }
    }
    h->D[k] = h->Y[k]; h->Y[k] = 0.0;
    for (; top < n; top++) {
      i = h->Pattern[top]; yi = h->Y[i]; h->Y[i] = 0.0;
      p2 = h->Lp[i] + h->Lnz[i];
      for (p = h->Lp[i]; p < p2; p++) h->Y[h->Li[p]] -= h->Lx[p] * yi;
      lki = yi / h->D[i];
      h->D[k] -= lki * yi;
      h->Li[p2] = k; h->Lx[p2] = lki; h->Lnz[i]++;
    }
    if (h->D[k] == 0.0) return k;
  }
  return n;
}

/* try_to_factorize(::LDLFactStruct, ...) — src/solver_types.jl:79-98 */
int cnlo_try_to_factorize(cnlo *h, const double *vals, i64 nvar, i64 nequ, i64 ncon,
                          double eig_tol, int set_mode, i64 *npos, i64 *nzer) {
  i64 N = nvar + nequ + ncon, i, pos = 0, zer = 0;
  cnlo_set_vals(h, vals, set_mode);
  h->factor_rank = ldl_numeric(h);
  for (i = 0; i < N; i++) {
    double di = h->D[i];
    pos += di > eig_tol;
    zer += fabs(di) <= eig_tol;
  }
  if (npos) *npos = pos;
  if (nzer) *nzer = zer;
  return pos == nvar && zer == 0;
}

/* solve_ldl! — src/solver_types.jl:69-77: ldiv!(d, factor, rhs); d .= -d.
 * ldiv! = permute, L solve, D solve, L^T solve, un-permute (Davis ldl_*).  */
int cnlo_solve_ldl(cnlo *h, const double *rhs, double *d) {
  i64 n = h->N, j, p;
  double *y = h->Y;
  for (j = 0; j < n; j++) y[j] = rhs[h->P[j]];                          /* ldl_perm   */
  for (j = 0; j < n; j++)                                               /* ldl_lsolve */
    for (p = h->Lp[j]; p < h->Lp[j] + h->Lnz[j]; p++) y[h->Li[p]] -= h->Lx[p] * y[j];
  for (j = 0; j < n; j++) y[j] /= h->D[j];                              /* ldl_dsolve */
  for (j = n - 1; j >= 0; j--)                                          /* ldl_ltsolve*/
    for (p = h->Lp[j]; p < h->Lp[j] + h->Lnz[j]; p++) y[j] -= h->Lx[p] * y[h->Li[p]];
  for (j = 0; j < n; j++) d[h->P[j]] = y[j];                            /* ldl_permt  */
  for (j = 0; j < n; j++) { d[j] = -d[j]; y[j] = 0.0; }
  return 1;
}

/* newton_system! — src/CaNNOLeS.jl:1008-1052.  vals is mutated (rho tail),
 * exactly as the reference mutates get_vals(LDLT).
 * out[0]=solve_success out[1]=rho out[2]=rho_old out[3]=nfact            */
void cnlo_newton_system(cnlo *h, double *d, i64 nvar, i64 nequ, i64 ncon, const double *rhs,
                        double *vals, double rho_old, const double *params, int set_mode,
                        double *out) {
  const double eig_tol = params[0], kdec = params[2], kinc = params[3], klarge = params[4],
               rho0 = params[5], rhomax = params[6], rhomin = params[7];
  i64 nfact = 0, i, o = h->nnz - nvar;
  double rho = 0.0;
  int success = cnlo_try_to_factorize(h, vals, nvar, nequ, ncon, eig_tol, set_mode, 0, 0); /* :1023 */
  nfact++;
  if (!success) {
    rho = rho_old == 0.0 ? rho0 : fmax(rhomin, kdec * rho_old);                           /* :1030 */
    for (i = 0; i < nvar; i++) vals[o + i] = rho;
    success = cnlo_try_to_factorize(h, vals, nvar, nequ, ncon, eig_tol, set_mode, 0, 0);
    nfact++;
    while (!success && rho <= rhomax) {                                                    /* :1035 */
      rho = rho_old == 0.0 ? klarge * rho : kinc * rho;
      if (rho <= rhomax) {
        for (i = 0; i < nvar; i++) vals[o + i] = rho;
        success = cnlo_try_to_factorize(h, vals, nvar, nequ, ncon, eig_tol, set_mode, 0, 0);
        nfact++;
      }
    }
    if (rho <= rhomax) rho_old = rho;                                                      /* :1044 */
  }
  int solve_success = success ? cnlo_solve_ldl(h, rhs, d) : 0;                             /* :1049 */
  out[0] = solve_success; out[1] = rho; out[2] = rho_old; out[3] = (double)nfact;
}

/* Batched driver used only as the timed CPU baseline: B problems that share
 * one pattern, values problem-major (vals[b*nnz+k]); single thread.        */
void cnlo_newton_system_batch(cnlo *h, i64 B, double *d, i64 nvar, i64 nequ, i64 ncon,
                              const double *rhs, double *vals, const double *rho_old,
                              const double *params, int set_mode, double *out) {
  i64 b, N = nvar + nequ + ncon;
  for (b = 0; b < B; b++)
    cnlo_newton_system(h, d + b * N, nvar, nequ, ncon, rhs + b * N, vals + b * h->nnz,
                       rho_old ? rho_old[b] : 0.0, params, set_mode, out + 4 * b);
}
