/* abi_smoke.c — a plain C99 caller of the C ABI (include/cannoles_hip.h), nothing else: no Python, no torch, no C++.
 * It replays the call sequence the reference's plugin surface makes (/root/reference/src/solver_types.jl:1-15,
 * src/CaNNOLeS.jl:1008-1052) — the sequence a Julia `ccall` binding issues — on two known-answer fixtures:
 *   F1  first Newton system of the reference's MGH01CON model (test/mgh01con.jl): inertia (2, 3, 0), d known as exact rationals
 *       (tests/golden/make_fixtures.py);  create -> try_to_factorize -> solve_ldl! -> newton_system! -> destroy
 *   F3  H = -10 I, J = 0.5 I: the rho ladder fails at rho = 0, rho0, 100 rho0, ... and succeeds at rho0 * 100^4 = 605.5 (nfact = 6)
 * Build: gcc -std=c99 -Wall -Wextra -pedantic -I include tests/c_abi/abi_smoke.c -L cannoles.jl_amd -lcannoles_hip -lm
 * Prints "ABI_SMOKE_OK" and exits 0, or says what differed and exits 1.                                                    */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "cannoles_hip.h"

#define CHECK(call)                                                                          \
  do {                                                                                       \
    int rc_ = (call);                                                                        \
    if (rc_ != CNL_OK) { printf("%s -> %d: %s\n", #call, rc_, cnl_last_error()); return 1; } \
  } while (0)

static double maxrel(const double* a, const double* b, int n) {
  double e = 0.0, m = 0.0;
  for (int i = 0; i < n; i++) { e = fmax(e, fabs(a[i] - b[i])); m = fmax(m, fabs(b[i])); }
  return e / m;
}

int main(void) {
  double params[9];
  if (cnl_version() < 200) { printf("cnl_version() = %d\n", (int)cnl_version()); return 1; }
  cnl_default_params(params);
  if (params[0] != 2.220446049250313e-16 || params[5] != 6.055454452393343e-6) { printf("default parameters differ from ParamCaNNOLeS\n"); return 1; }

  /* ---- F1 ------------------------------------------------------------------------------------------------------------ */
  {
    const int64_t rows[13] = {1, 1, 2, 2, 3, 4, 4, 5, 3, 4, 5, 1, 2}, cols[13] = {1, 1, 1, 2, 1, 1, 2, 1, 3, 4, 5, 1, 2};
    double vals[13] = {88.0, -0.0, -0.0, -0.0, -1.0, 24.0, 10.0, 1.0, -1.0, -1.0, -0.1, 0.0, 0.0};
    const double rhs[5] = {0.0, -44.0, 0.0, 0.0, -1.2};
    const double exact[5] = {-52.0 / 55.0, 149.0 / 55.0, 52.0 / 55.0, 4.4, -236.0 / 11.0};
    cnl_handle* h = NULL;
    CHECK(cnl_create(&h, 5, 13, rows, cols, 2, 2, 1, 1, 0));
    int32_t ok = -1;
    int64_t npos = -1, nzero = -1;
    CHECK(cnl_factorize(h, vals, params[0], &ok, &npos, &nzero));
    if (ok != 1 || npos != 2 || nzero != 0) { printf("F1 try_to_factorize: success %d inertia (%lld, ., %lld)\n", (int)ok, (long long)npos, (long long)nzero); return 1; }
    double d[5] = {0, 0, 0, 0, 0};
    CHECK(cnl_solve(h, rhs, d));
    if (maxrel(d, exact, 5) > 1e-13) { printf("F1 solve_ldl!: relative error %.3e\n", maxrel(d, exact, 5)); return 1; }
    double d2[5] = {0, 0, 0, 0, 0}, rho_old = 0.0, rho = -1.0, rho_old_out = -1.0;
    int32_t nfact = -1, success = -1;
    CHECK(cnl_newton_system(h, vals, rhs, d2, &rho_old, params, &rho, &rho_old_out, &nfact, &success));
    if (success != 1 || nfact != 1 || rho != 0.0 || rho_old_out != 0.0 || maxrel(d2, exact, 5) > 1e-13) {
      printf("F1 newton_system!: success %d nfact %d rho %g rho_old %g error %.3e\n", (int)success, (int)nfact, rho, rho_old_out, maxrel(d2, exact, 5));
      return 1;
    }
    CHECK(cnl_destroy(h));
  }
  /* ---- F3 ------------------------------------------------------------------------------------------------------------ */
  {
    const int64_t rows[12] = {1, 2, 3, 4, 5, 6, 4, 5, 6, 1, 2, 3}, cols[12] = {1, 2, 3, 1, 2, 3, 4, 5, 6, 1, 2, 3};
    double vals[12] = {-10.0, -10.0, -10.0, 0.5, 0.5, 0.5, -1.0, -1.0, -1.0, 0.0, 0.0, 0.0};
    const double rhs[6] = {1.0, 2.0, 3.0, 4.0, 5.0, 6.0};
    cnl_handle* h = NULL;
    CHECK(cnl_create(&h, 6, 12, rows, cols, 3, 3, 0, 1, 0));
    double d[6] = {0, 0, 0, 0, 0, 0}, rho_old = 0.0, rho = -1.0, rho_old_out = -1.0;
    int32_t nfact = -1, success = -1;
    CHECK(cnl_newton_system(h, vals, rhs, d, &rho_old, params, &rho, &rho_old_out, &nfact, &success));
    const double want = params[5] * 1e8;   /* rho0 * kappa_largeinc^4 */
    if (success != 1 || nfact != 6 || fabs(rho - want) > 1e-12 * want || rho_old_out != rho) {
      printf("F3 ladder: success %d nfact %d rho %.17g (expected %.17g) rho_old %.17g\n", (int)success, (int)nfact, rho, want, rho_old_out);
      return 1;
    }
    for (int i = 9; i < 12; i++)
      if (vals[i] != rho) { printf("F3: rho slot %d holds %.17g, not the last rho tried\n", i, vals[i]); return 1; }
    /* K d = -rhs with K = [(rho - 10) I, 0.5 I; 0.5 I, -I] */
    for (int i = 0; i < 3; i++) {
      const double r1 = (rho - 10.0) * d[i] + 0.5 * d[3 + i] + rhs[i], r2 = 0.5 * d[i] - d[3 + i] + rhs[3 + i];
      if (fabs(r1) > 1e-10 || fabs(r2) > 1e-10) { printf("F3 residual %g %g\n", r1, r2); return 1; }
    }
    /* a second call after a failed factorisation must not solve: batch = 1 -> CNL_ERR_STATE */
    double v2[12] = {-10.0, -10.0, -10.0, 0.5, 0.5, 0.5, -1.0, -1.0, -1.0, 0.0, 0.0, 0.0};
    int32_t ok = -1;
    CHECK(cnl_factorize(h, v2, params[0], &ok, NULL, NULL));
    if (ok != 0) { printf("F3 at rho = 0 must fail the inertia test\n"); return 1; }
    if (cnl_solve(h, rhs, d) != CNL_ERR_STATE) { printf("solve_ldl! after a failed factorisation must be a call-sequence error\n"); return 1; }
    CHECK(cnl_destroy(h));
  }
  printf("ABI_SMOKE_OK\n");
  return 0;
}
