/*
 * cannoles_hip.h — C ABI of the MI355X-native Newton-system backend for CaNNOLeS.
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++/torch types.
 * Every entry point cites the reference interface it replaces
 * (paths under /root/reference).  The Julia binding a maintainer would add
 * is shown in INTEGRATION.md and shipped in cannoles.jl_amd/julia/.
 *
 * Conventions
 *   - Index arrays are int64, 1-based, exactly as Julia holds `rows`/`cols`
 *     (src/CaNNOLeS.jl:276-315).  Values are double (Float64 only; other
 *     element types must stay on the reference's LDLFactStruct).
 *   - The COO pattern is the lower triangle of the KKT matrix in the
 *     reference's 7-segment order [H_F | H_c | J_F | J_c | -I | -dI | rI];
 *     duplicates are summed in COO order (src/solver_types.jl:53-59); the
 *     LAST nvar entries are the rho slots (src/CaNNOLeS.jl:1027).
 *   - A handle serves a batch of `batch` independent problems that share the
 *     pattern (batch = 1 is the reference's one-solver-one-problem case).
 *     Batched arrays are problem-major: vals[b*nnz + k], rhs[b*N + i].
 *   - Host-pointer entry points copy in/out and keep no pointer afterwards.
 *   - cnl_solve / cnl_solve_dev use the LAST factorisation of the handle, whichever call made it (cnl_factorize or
 *     cnl_newton_system, with the rho it ended on).  The `_dev` twins keep the caller's d_vals pointer for that
 *     (sparse backends re-read the Jacobian entries during the solve): d_vals must stay alive and unmodified between a
 *     `_dev` factorisation and the `_dev` solves that use it.  The dense backend snapshots what it needs.
 *     `_dev` entry points take device pointers (HBM-resident data) and a
 *     hipStream_t passed as void*; they are asynchronous on that stream.
 *   - Return value: 0 = OK, otherwise one of CNL_ERR_*; cnl_last_error()
 *     describes the last failure on the calling thread.  Numerical failure
 *     (wrong inertia, zero pivot) is NOT an error: it is reported through
 *     `success`, like the reference's Bool (src/solver_types.jl:96-97).
 *   - There is no CPU fallback: without a usable HIP device every call that
 *     needs one fails with CNL_ERR_HIP.
 */
#ifndef CANNOLES_HIP_H
#define CANNOLES_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CNL_OK 0
#define CNL_ERR_ARG 1      /* null pointer / bad handle / bad size              */
#define CNL_ERR_DIM 2      /* inconsistent or unsupported dimensions            */
#define CNL_ERR_PATTERN 3  /* malformed COO pattern (range, upper triangle)     */
#define CNL_ERR_HIP 4      /* HIP runtime error or no device                    */
#define CNL_ERR_STATE 5    /* call sequence error (solve before factorize)      */
#define CNL_ERR_INTERNAL 9

typedef struct cnl_plan cnl_plan;     /* host-only symbolic analysis            */
typedef struct cnl_handle cnl_handle; /* plan + device state for one batch      */

const char* cnl_last_error(void);
/* library / ABI version: major*10000 + minor*100 + patch */
int32_t cnl_version(void);

/* ParamCaNNOLeS(Float64) defaults, src/CaNNOLeS.jl:48-62, in the order
 * [eig_tol, delta_min, kappa_dec, kappa_inc, kappa_largeinc, rho0, rho_max, rho_min, gamma_A] */
void cnl_default_params(double params[9]);

/* ---- options ----------------------------------------------------------------------------------------------------------
 * Everything that selects a plan or an execution is an ARGUMENT: the library reads no environment variable that changes what it
 * computes (CNL_VERBOSE only adds log lines on stderr; the debugging aids CNL_DBG_LDSFILL / CNL_DBG_SCRATCHFILL — read once per process —
 * launch kernels that leave a byte pattern in LDS, scratch and registers in front of every launch, CNL_DBG_GUARD surrounds every device
 * allocation with filled guard zones — the library's results must not, and do not, depend on either).  cnl_options_init fills the
 * defaults — the choices cnl_create makes by itself; the `_ex` entry points take a modified copy.  A drop-in caller never needs it (the
 * reference has no counterpart: `ldl_analyze` takes no options, src/solver_types.jl:63).
 * The structure holds the switches that select a plan kind or allow / forbid a kernel family or an execution; everything finer
 * (orderings, thresholds, ablation switches of single kernels, measured-slower experiments) is reachable through `tuning`, a string of
 * "key=value" pairs whose keys are listed in cannoles.jl_amd/csrc/options.h — used by the measurement tools and the randomised
 * option fuzz of the test-suite; an unknown key is CNL_ERR_ARG.                                                                   */
#define CNL_PLAN_AUTO 0        /* by batch size: latency plan up to staged_max_batch problems, throughput plan above           */
#define CNL_PLAN_THROUGHPUT 1  /* least total work, one sequential record stream per four problems                             */
#define CNL_PLAN_LATENCY 2     /* bushy elimination tree cut into tasks (staged execution)                                      */
#define CNL_LAYOUT_PROBLEM_MAJOR 0   /* vals[b * nnz + k]: problem after problem, the reference's vector per problem               */
#define CNL_LAYOUT_INTERLEAVED 1     /* groups of 32 problems interleaved in blocks of eight doubles:
                                        vals[((b / 32 * (nnz_blocks) + k / 8) * 32 + b % 32) * 8 + k % 8], nnz_blocks = (nnz + 7) / 8 + 1;
                                        cnl_layout_len gives the array's length, cnl_interleave_dev / cnl_deinterleave_dev convert,
                                        cnl_prepare_newton_system_dev (row f2) writes it directly.  For device-resident callers of
                                        band-structured batches: every load of the band kernels then moves 512 contiguous bytes
                                        (16 384 problems of the headline pattern: 12.5 -> 11.0 ms; nothing to gain below 8 193
                                        problems).  rhs and d are problem-major in both layouts.  cnl_create fails (CNL_ERR_ARG) when
                                        the band kernels do not serve the handle.                                                   */
typedef struct cnl_options {
  int32_t struct_size;         /* sizeof(cnl_options), set by cnl_options_init (ABI evolution)                                  */
  int32_t plan_kind;           /* CNL_PLAN_*                                                                                    */
  int64_t staged_max_batch;    /* CNL_PLAN_AUTO: largest batch planned for latency (default 4096)                              */
  int32_t verbose;             /* 1: log plan decisions on stderr                                                               */
  int32_t band_kernel;         /* 1 (default): a throughput handle whose pattern is a band in the natural order of the variables
                                  (every residual row and Hessian entry within five consecutive variables, every constraint row a
                                  run of its own) runs on the band kernels (csrc/band.h: one lane per problem and half of the chain,
                                  operands streamed through LDS) — newton_system, try_to_factorize (the forward sweep alone) and
                                  solve_ldl! (which factorises the values of the last factorisation again and sweeps the new
                                  right-hand side in the same launch); 2: the same with the chain in one part; 0: the register-front
                                  kernel                                                                                        */
  int32_t dense_backend;       /* 1: dense residual blocks (BASELINE config 2) go to the dense backend (fp64 MFMA)              */
  int32_t staged;              /* 1: latency plans run stage by stage (tasks of the elimination tree on different wavefronts)   */
  int32_t dataflow;            /* 1: smallest batches run all tasks in one launch per phase, waiting on device counters         */
  int32_t device_ladder;       /* 1: cnl_newton_system(_dev) on staged handles climbs the rho ladder inside ONE launch in which every
                                  task of the elimination tree has a wavefront of its own; 0: the sequential launch              */
  int32_t host_ladder;         /* 1: where the in-kernel ladder is not available (device_ladder = 0, split handles, the dense backend)
                                  the host-pointer cnl_newton_system drives the rho ladder from the host, every rung a staged
                                  try_to_factorize; 0: the sequential launch                                                    */
  int32_t split_tail;          /* 1: the remainder of a batch above a machine-filling one runs on a handle of its own, with the plan
                                  cnl_create picks for a batch of that size, behind the rest on the caller's stream             */
  int32_t multi_share_plan;    /* 1: cnl_multi_create analyses the pattern once for all shards of equal plan kind               */
  int32_t batch_layout;        /* layout of the `vals` arrays of the DEVICE-pointer entry points (CNL_LAYOUT_*; host-pointer calls take
                                  the reference's arrays, one problem after the other, always).  CNL_LAYOUT_INTERLEAVED: see below  */
  char force_order[32];        /* name of an ordering candidate to force ("" = none)                                            */
  char tuning[192];            /* "key=value[,key=value...]": any switch of csrc/options.h (measurement / test use)             */
} cnl_options;
void cnl_options_init(cnl_options* opt);

/* ---- symbolic analysis (host only; replaces `ldl_analyze`, ------------------
 *      src/solver_types.jl:61-65: sparse()+triu()+ldl_analyze) ---------------- */
int cnl_plan_create(cnl_plan** plan, int64_t N, int64_t nnz, const int64_t* rows1, const int64_t* cols1,
                    int64_t nvar, int64_t nequ, int64_t ncon);
/* The analysis cnl_create(batch) runs: batches too small to give every CU a wavefront get an order with a bushy elimination
 * tree, cut into tasks that run on different wavefronts (plan arrays "tasks", "stage_ptr"); large batches get the order with
 * the least total work (one sequential record stream per four problems).  cnl_plan_create is the large-batch analysis.      */
int cnl_plan_create_for_batch(cnl_plan** plan, int64_t N, int64_t nnz, const int64_t* rows1, const int64_t* cols1,
                              int64_t nvar, int64_t nequ, int64_t ncon, int64_t batch);
/* the same with explicit options (opt == NULL: defaults).  batch <= 0 with CNL_PLAN_AUTO means the large-batch analysis. */
int cnl_plan_create_ex(cnl_plan** plan, int64_t N, int64_t nnz, const int64_t* rows1, const int64_t* cols1,
                       int64_t nvar, int64_t nequ, int64_t ncon, int64_t batch, const cnl_options* opt);
void cnl_plan_destroy(cnl_plan* plan);
/* info[0]=N [1]=nnz [2]=unique nnz(K) [3]=nsuper [4]=nnz(L) stored (strictly lower, with relaxed zeros)
 * [5]=nnz(L) of the ordering without relaxation [6]=factor storage doubles/problem [7]=largest front order
 * [8]=forward work-stack doubles [9]=backward work-stack doubles [10]=largest panel doubles
 * [11]=FMAs per factorisation [12]=assembly entries
 * [13]=fronts per class of the register-front kernel, packed (order<=16 | <=32 << 20 | <=64 << 40), -1 if unused
 * [14]=its LDS need, packed (update stack doubles | staging doubles << 20 | record words << 40), -1 if unused
 * [15]=residual nodes eliminated by static condensation (csrc/condense.h).  [4],[5] include their L rows. */
int cnl_plan_info(const cnl_plan* plan, int64_t info[16]);
/* Copy a named int32 index array of the plan.  "perm": elimination order in the reference's numbering
 * (0-based; condensed residual nodes first).  Multifrontal plan of the (condensed) system: "inner_perm",
 * "fronts" (16 int32 per front, struct FrontHdr in csrc/plan.h), "seg_ptr", "asm_pos", "asm_src",
 * "child_idx", "rel_idx".  Condensation lists (csrc/condense.h): "c_ptr", "c_a", "c_b", "c_d", "orig_of",
 * "r_orig", "r_dsrc", "r_ptr", "r_jsrc", "r_jx".  Record streams of the register-front kernel (csrc/plan.h; empty when that
 * kernel does not serve the plan): "rec", "brec".  Staged plans: "tasks" (8 int32 per task, struct Task in csrc/plan.h: stage,
 * first front, end front, record offset, backward record offset, 1 if a root, parent task or -1, number of child tasks),
 * "stage_ptr".  With out == NULL only *count is set.
 * Used by the tests' plan simulator and to hand the ordering to the oracle.                          */
int cnl_plan_get(const cnl_plan* plan, const char* name, int32_t* out, int64_t* count);
const char* cnl_plan_order_name(const cnl_plan* plan);

/* ---- solver object (replaces LDLFactStruct(N, rows, cols, vals), ---------------
 *      src/solver_types.jl:45-51,61-65; called at src/CaNNOLeS.jl:327) --------- */
int cnl_create(cnl_handle** h, int64_t N, int64_t nnz, const int64_t* rows1, const int64_t* cols1,
               int64_t nvar, int64_t nequ, int64_t ncon, int64_t batch, int device);
/* cnl_create with explicit options (opt == NULL: defaults = cnl_create) */
int cnl_create_ex(cnl_handle** h, int64_t N, int64_t nnz, const int64_t* rows1, const int64_t* cols1,
                  int64_t nvar, int64_t nequ, int64_t ncon, int64_t batch, int device, const cnl_options* opt);
int cnl_destroy(cnl_handle* h);
/* Number of dataflow waits that gave up since the handle was created (see cnl_options.dataflow_spin_limit).  Every such call
 * was redone by the sequential execution, so results are right; a non-zero count says the device did not run the workgroups
 * of a launch concurrently / in index order (time-slicing, a debugger).  Synchronises the handle's last call.                 */
int cnl_dataflow_timeouts(cnl_handle* h, int64_t* count);
const cnl_plan* cnl_get_plan(const cnl_handle* h);

/* try_to_factorize(LDLT, vals, nvar, nequ, ncon, eig_tol) — src/solver_types.jl:79-98.
 * vals: batch*nnz values (rho slots as given).  success[b] = 1 iff the factorisation completed
 * with #{d > eig_tol} == nvar and #{|d| <= eig_tol} == 0.  npos/nzero (optional, batch each)
 * return the two counts.                                                                      */
int cnl_factorize(cnl_handle* h, const double* vals, double eig_tol, int32_t* success, int64_t* npos, int64_t* nzero);

/* solve_ldl!(rhs, factor, d) — src/solver_types.jl:69-77: d = -(K^-1 rhs); rhs untouched.
 * Requires a preceding cnl_factorize / cnl_newton_system.  The reference never solves after a failed factorisation
 * (src/CaNNOLeS.jl:1049); here: with batch = 1 that call sequence returns CNL_ERR_STATE (nothing is written); in a batch the rows of
 * d that belong to problems whose last host-pointer factorisation failed are left untouched.  After a device-pointer
 * factorisation the flags are not known to the host: cnl_solve / cnl_solve_dev then write unspecified (finite or not) values for
 * the problems that failed — the caller holds d_success and must not use those rows.                                          */
int cnl_solve(cnl_handle* h, const double* rhs, double* d);

/* newton_system!(d, nvar, nequ, ncon, rhs, vals, LDLT, rho_old, params) — src/CaNNOLeS.jl:1008-1052,
 * one call: factorise at rho=0, climb the rho ladder per problem on failure, solve.  Both entries decide the ladder on the device:
 * the single stream of large batches inside its one launch, latency plans (small and mid-size batches) inside one fused launch in
 * which every task of the elimination tree has a wavefront of its own (cnl_options.device_ladder; kernels2.hip, phase 2).  Only where
 * that launch cannot hold a batch's tasks at once — and on the dense backend — the ladder of THIS (host-pointer) entry is driven from
 * the host, every rung a parallel try_to_factorize of the batch (cnl_options.host_ladder), and the device-pointer twin falls back
 * to the sequential launch.
 * vals (batch*nnz) is mutated: the rho slots receive the last rho tried, as the reference leaves them.
 * rho_old: batch inputs.  Outputs (batch each): rho, rho_old_out, nfact, success (= solve_success).  */
int cnl_newton_system(cnl_handle* h, double* vals, const double* rhs, double* d, const double* rho_old,
                      const double params[9], double* rho, double* rho_old_out, int32_t* nfact, int32_t* success);

/* Device-resident twins.  d_vals/d_rhs/d_d are device pointers with the layouts above.
 * d_rho_old is read and updated in place (rho_old_out); d_rho, d_nfact, d_success are outputs. */
int cnl_factorize_dev(cnl_handle* h, const double* d_vals, double eig_tol, int32_t* d_success, void* stream);
int cnl_solve_dev(cnl_handle* h, const double* d_rhs, double* d_d, void* stream);
int cnl_newton_system_dev(cnl_handle* h, double* d_vals, const double* d_rhs, double* d_d, double* d_rho_old,
                          double* d_rho, int32_t* d_nfact, int32_t* d_success, const double params[9], void* stream);

/* ---- vectors either side of the Newton system, device-resident ------------------
 * They let `rhs` and `d` stay in HBM between inner iterations instead of crossing PCIe every call.
 *
 * cnl_residual_vectors_dev: rhs = [dual ; primal] and their infinity norms, as src/CaNNOLeS.jl computes them at
 * :507-508, :519-524, :528-529 (start), :722-726, :730-731 (trial point) and assembles at :631-632:
 *     dual = Jx' r - Jc' lambda,   primal = [F - r ; c],   norms[2b] = ||dual||_inf, norms[2b+1] = ||primal||_inf.
 * The Jacobian values are read from the J_F / J_c segments of `vals` (prepare_newton_system!, :953-967).  Both
 * products are accumulated per column in COO order (the order of the reference's COO mul!) and then subtracted.
 * Batched layouts: r, Fx [batch][nequ]; lambda, cx [batch][ncon] (may be NULL when ncon == 0); rhs [batch][N].   */
int cnl_residual_vectors_dev(cnl_handle* h, const double* d_vals, const double* d_r, const double* d_lambda, const double* d_Fx,
                             const double* d_cx, double* d_rhs, double* d_norms, void* stream);

/* The same two rows with the Jacobian values read from the MODEL's arrays Jx [batch][nnzjF], Jcx [batch][nnzjc] — what
 * prepare_newton_system! copies into the J_F / J_c segments of vals, and what the reference's own products use (`mul!(Jxtr, Jx', r)`,
 * src/CaNNOLeS.jl:507: the Jx operator is built on Jx_vals) — instead of from `vals`: no prepare pass is needed in front of them, and they
 * serve handles of either batch_layout.  Results bit-identical to the `vals` variants (same kernels, another base pointer).  The pattern's
 * J_F and J_c entries must be one run of COO slots each (the reference's 7-segment layout; CNL_ERR_STATE otherwise).               */
int cnl_residual_vectors_jac_dev(cnl_handle* h, int64_t nnzjF, int64_t nnzjc, const double* d_Jx, const double* d_Jcx, const double* d_r,
                                 const double* d_lambda, const double* d_Fx, const double* d_cx, double* d_rhs, double* d_norms, void* stream);
int cnl_cgls_multipliers_jac_dev(cnl_handle* h, int64_t nnzjF, int64_t nnzjc, const double* d_Jx, const double* d_Jcx, const double* d_r,
                                 double* d_lambda, double* d_Jxtr, double atol, double rtol, int64_t itmax, int ones_if_zero,
                                 int32_t* d_iters, void* stream);

/* cnl_prepare_newton_system_dev: prepare_newton_system!(meth, vals, nls, x, lambda, r, Jx_vals, Jcx_vals, delta, Fx) —
 * src/CaNNOLeS.jl:947-981 — for a batch whose model values already live on the device: the value
 * arrays are copied into the segments of `vals` ([H_F | H_c | J_F | J_c | -I | -delta I | rho I], sizes nnzhF, nnzhc,
 * nnzjF, nnzjc, nequ, ncon, nvar):  H_F <- hF (skipped when d_hF == NULL: the Gauss-Newton variants do not touch it),
 * H_c <- -hc, J_F <- Jx, J_c <- Jcx, -delta I <- -delta[b], rho I <- 0; the -I segment is never written.
 * Layouts: hF [batch][nnzhF], hc [batch][nnzhc], Jx [batch][nnzjF], Jcx [batch][nnzjc], delta [batch].          */
int cnl_prepare_newton_system_dev(cnl_handle* h, int64_t nnzhF, int64_t nnzhc, int64_t nnzjF, int64_t nnzjc, const double* d_hF,
                                  const double* d_hc, const double* d_Jx, const double* d_Jcx, const double* d_delta, double* d_vals,
                                  void* stream);
/* On a handle created with cnl_options.batch_layout = CNL_LAYOUT_INTERLEAVED, cnl_prepare_newton_system_dev WRITES d_vals interleaved
 * (the model's arrays stay problem-major), and cnl_factorize_dev / cnl_newton_system_dev read — the rho slots: write — d_vals in that
 * layout; d_rhs and d_d are problem-major.  cnl_residual_vectors_dev and cnl_cgls_multipliers_dev read problem-major vals only
 * (CNL_ERR_STATE on such a handle: use their `_jac` twins above), the host-pointer entry points likewise.
 *   cnl_layout_len: doubles of the interleaved array for the handle's batch, which = 0: vals, 1: an N-vector per problem;
 *   cnl_interleave_dev / cnl_deinterleave_dev: problem-major -> interleaved / back (out of place; pads are written as zeros).     */
int cnl_layout_len(const cnl_handle* h, int which, int64_t* doubles);
int cnl_interleave_dev(cnl_handle* h, int which, const double* d_src, double* d_dst, void* stream);
int cnl_deinterleave_dev(cnl_handle* h, int which, const double* d_src, double* d_dst, void* stream);

/* cnl_cgls_multipliers_dev: least-squares multiplier estimate  min || Jc' lambda - Jx' r ||  (SURVEY 8 row f4), what the
 * reference obtains with `mul!(Jxtr, Jx', r); krylov_solve!(cgls_workspace, Jcx', Jxtr)` at src/CaNNOLeS.jl:507-518 and
 * :880-882: CGLS started at 0, stopped when ||Jc res|| <= atol + rtol ||Jc b|| or after itmax steps (itmax <= 0: nvar + ncon,
 * Krylov.jl's default; its default tolerances are sqrt(eps)).  ones_if_zero != 0 applies `if norm(lambda) == 0: lambda .= 1`
 * (:515-517).  d_Jxtr (optional) receives Jx' r, d_iters (optional) the iteration counts.  ncon <= 1024.                */
int cnl_cgls_multipliers_dev(cnl_handle* h, const double* d_vals, const double* d_r, double* d_lambda, double* d_Jxtr, double atol,
                             double rtol, int64_t itmax, int ones_if_zero, int32_t* d_iters, void* stream);

/* cnl_trial_point_dev: the extrapolation step's trial point, src/CaNNOLeS.jl:654,661-668:
 *     xt = x + d[1:n],  rt = r + d[n+1:n+m],  dlambda = -d[n+m+1:N], scaled by max_dlambda/||dlambda||_2 when that
 *     norm exceeds max_dlambda (the reference uses 1e4),  lambdat = lambda + dlambda.                               */
int cnl_trial_point_dev(cnl_handle* h, const double* d_x, const double* d_r, const double* d_lambda, const double* d_d,
                        double max_dlambda, double* d_xt, double* d_rt, double* d_lambdat, double* d_dlambda, void* stream);

/* ---- one caller, several devices -------------------------------------------------------------------------
 * The path shards only across independent problems: `batch` problems are cut into contiguous balanced shards, one per entry
 * of `devices` (the first batch % ndev shards hold one problem more; with batch < ndev the surplus devices stay idle), each
 * shard with a handle of its own on its device.  The calls below take the arrays of the WHOLE batch (problem-major, as the
 * single-handle calls), run every shard from a host thread of its own and return when all are done; there is no collective
 * and no traffic between the devices.  This is what a single `cannoles()`-style caller (one backend object selected at
 * src/CaNNOLeS.jl:322-332) uses to drive all GPUs of a node without becoming one process per GPU.                          */
typedef struct cnl_multi cnl_multi;
int cnl_multi_create(cnl_multi** m, int64_t N, int64_t nnz, const int64_t* rows1, const int64_t* cols1, int64_t nvar, int64_t nequ,
                     int64_t ncon, int64_t batch, const int* devices, int ndev);
/* the same with explicit options (opt == NULL: defaults).  The pattern is analysed ONCE for all shards (cnl_options.multi_share_plan). */
int cnl_multi_create_ex(cnl_multi** m, int64_t N, int64_t nnz, const int64_t* rows1, const int64_t* cols1, int64_t nvar, int64_t nequ,
                        int64_t ncon, int64_t batch, const int* devices, int ndev, const cnl_options* opt);
int cnl_multi_destroy(cnl_multi* m);
/* shards actually created: *nshards, and (optional arrays of that length) first problem, problems, device index */
int cnl_multi_shards(const cnl_multi* m, int64_t* nshards, int64_t* start, int64_t* count, int32_t* device);
int cnl_multi_factorize(cnl_multi* m, const double* vals, double eig_tol, int32_t* success, int64_t* npos, int64_t* nzero);
int cnl_multi_solve(cnl_multi* m, const double* rhs, double* d);
int cnl_multi_newton_system(cnl_multi* m, double* vals, const double* rhs, double* d, const double* rho_old, const double params[9],
                            double* rho, double* rho_old_out, int32_t* nfact, int32_t* success);

/* Device-resident twins: entry i of every pointer array is shard i's device array on shard i's device (problem-major,
 * count[i] problems; cnl_multi_shards).  The calls enqueue every shard's work on streams[i] (NULL: the shard handle's own
 * stream) from the calling thread and return; nothing crosses PCIe or xGMI.  cnl_multi_synchronize waits for all shards.   */
int cnl_multi_factorize_dev(cnl_multi* m, const double* const* d_vals, double eig_tol, int32_t* const* d_success, void* const* streams);
int cnl_multi_solve_dev(cnl_multi* m, const double* const* d_rhs, double* const* d_d, void* const* streams);
int cnl_multi_newton_system_dev(cnl_multi* m, double* const* d_vals, const double* const* d_rhs, double* const* d_d,
                                double* const* d_rho_old, double* const* d_rho, int32_t* const* d_nfact, int32_t* const* d_success,
                                const double params[9], void* const* streams);
int cnl_multi_synchronize(cnl_multi* m, void* const* streams);

/* ---- bookkeeping of the batched outer / inner loop, device-resident -----------------------------------------
 * A caller that runs `solve!` (src/CaNNOLeS.jl:612-864) for a batch of problems in lockstep keeps the loop's state as [batch, ...]
 * arrays in HBM and every branch as a per-problem mask (cannoles.jl_amd/device_loop.py).  The four entry points below do the
 * masks and the masked state updates of one global step IN PLACE on that state, one launch each (the reference's tests in the
 * reference's operation order; minimum / maximum propagate NaN).  `cnl_outer_state` is plain device pointers and sizes; every
 * array is problem-major.  Status codes: 0 unknown (active), 1 first_order, 2 small_residual, 3 exception, 4 max_eval (never produced: this loop has no evaluation counter), 5 stalled (inner > max_inner, src/CaNNOLeS.jl:846).  The entry points
 * take no handle: they launch on the calling thread's CURRENT device, which must be the one the arrays live on, and are asynchronous on
 * `stream`; they return CNL_ERR_ARG for a null state / missing array and CNL_ERR_HIP when the launch fails.  Every array is required
 * — the multiplier / constraint arrays lam, cx, ct, lamt, lamt_e (cl, lam_ls for the line search) have rows of P = max(p, 1) entries and
 * are required also when p == 0 —; only Jcv / Jct may be NULL when nnzjc == 0.
 *   cnl_outer_begin_dev        :612-626  start of an outer iteration for the problems in phase0; act, need (a Newton system is due);
 *                                        flags[0..3] = any active / any need / any extrapolation step / any line-search step
 *   cnl_outer_newton_done_dev  :633-659  did_newton != 0: d, rho_old, nfact, nlin from the Newton call's outputs (d_new, ro_tmp,
 *                                        nf_new, rho_new, ok_new) where `need`; brk (`broken`); then ext / lsm and eps_k where ext
 *   cnl_outer_extrapolated_dev :661-668  xt, rt, lamt <- the trial point cnl_trial_point_dev left in xt_e, rt_e, lamt_e, where ext
 *   cnl_outer_trial_done_dev   :722-800  optimality measures at the trial point (nrm_t = the norms of cnl_residual_vectors_dev),
 *                                        acceptance, x / r / Fx / cx / Jv / Jcv / lam / rhs_cur / fx / delta / inner updates, end-of-
 *                                        inner-loop tests; rej, chk, done_in; flags[4] = any rejected, flags[5] = any small-residual check
 *   cnl_outer_end_dev          :800-857  statuses, outer-iteration counters, phase0                                                */
typedef struct cnl_outer_state {
  int64_t B, n, m, p, P /* max(p, 1): row length of lam, cx, ct, lamt */, N, nnzjF, nnzjc, max_inner;
  double dmin, rhomax, delta_dec, smax;
  int32_t *status, *it, *flags /* [8] */, *nf_new, *ok_new;
  int64_t *inner, *nfact, *nlin;
  uint8_t *phase0, *act, *need, *brk, *ext, *lsm, *rej, *chk, *done_in, *tired, *small_res;
  double *normdual, *normprimal, *combined, *combined_hat, *delta, *ndh, *nph, *fx, *epsk, *epstol, *epsF, *epsc, *rho_old;
  double *d, *d_new, *ro_tmp, *rho_new;
  double *x, *r, *Fx, *cx, *Jv, *Jcv, *lam, *rhs_cur;
  double *xt, *rt, *Ft, *ct, *Jt, *Jct, *lamt, *rhs_t, *nrm_t /* [B][2] */;
  double *xt_e, *rt_e, *lamt_e;
  /* Armijo line search (src/CaNNOLeS.jl:1054-1112): ls_g = the [dual; primal] vector of cnl_residual_vectors_dev at (Fx, lam_ls) —
   * its first n entries are the gradient of the merit function; xl / Fl / cl = trial point, F and c there (the caller evaluates
   * the model); lam_ls = lam - c / delta; per problem: alpha, Dphi, phix, eta, nbk; bt = still backtracking; flags[6] = any bt  */
  double gammaA, eps2;
  double *ls_g, *xl, *Fl, *cl, *lam_ls, *alpha, *Dphi, *phix, *eta;
  int64_t* nbk;
  uint8_t* bt;
} cnl_outer_state;
int cnl_outer_begin_dev(const cnl_outer_state* st, void* stream);
int cnl_outer_newton_done_dev(const cnl_outer_state* st, int did_newton, void* stream);
int cnl_outer_extrapolated_dev(const cnl_outer_state* st, void* stream);
int cnl_outer_trial_done_dev(const cnl_outer_state* st, void* stream);
/* line search of the problems in lsm: ls_begin (Dphi, eta, phi(x), alpha = 1, xl = x + dx) -> [caller: Fl, cl = F(xl), c(xl)] ->
 * ls_test(first = 1) -> while flags[6]: ls_step (alpha / 4, xl, nbk) -> [caller: Fl, cl] -> ls_test(0); then ls_take (xt, rt, lamt)   */
int cnl_outer_ls_begin_dev(const cnl_outer_state* st, void* stream);
int cnl_outer_ls_test_dev(const cnl_outer_state* st, int first, void* stream);
int cnl_outer_ls_step_dev(const cnl_outer_state* st, void* stream);
int cnl_outer_ls_take_dev(const cnl_outer_state* st, void* stream);
int cnl_outer_end_dev(const cnl_outer_state* st, void* stream);

/* Device time, in milliseconds, of the multifrontal kernel (the dominant kernel) of the last call,
 * measured with HIP events on the call's stream.  Enabling timing makes every call synchronise on
 * its stream, so it is meant for measurement loops, not for production (0 if not enabled).      */
int cnl_set_timing(cnl_handle* h, int enable);
int cnl_last_kernel_ms(cnl_handle* h, float* ms);

/* Kernel configuration actually chosen: cfg[0]=threads per problem, [1]=problems per workgroup,
 * [2]=LDS bytes per workgroup, [3]=1 if the work stack lives in LDS else 0 (global scratch),
 * [4]=grid size (those five describe the general kernel, kernels.hip), [5]=2 if the register-front
 * kernel (kernels2.hip) serves newton_system/factorize (4: with staged execution of the first attempt), 3 dense backend, else 1;
 * + 16 when newton_system / factorize run the LEAN instantiation (fast-class fronts with row-form products only), [6]=its wavefronts per workgroup,
 * [7]=its LDS bytes per workgroup; [5] + 32 when the remainder of the batch runs on a handle of its own (cnl_options.split_tail),
 * + 64 when cnl_newton_system runs on the band kernels (then bits 8-15 = problems per workgroup, bits 16-23 = parts of the chain),
 * + 128 when cnl_residual_vectors_dev runs on column tiles. */
int cnl_get_config(const cnl_handle* h, int64_t cfg[8]);
/* Launches of the Newton-system kernels since the library was loaded, per kernel family: counts[0] band kernels (csrc/band.hip),
 * [1] register-front kernel (csrc/kernels2.hip, staged launches not included), [2] general kernel (csrc/kernels.hip).  Lets a test
 * pin WHICH kernel served a call sequence (e.g. that solve_ldl! behind a band factorisation launches no second kernel family). */
int cnl_launch_counts(int64_t counts[3]);

#ifdef __cplusplus
}
#endif
#endif /* CANNOLES_HIP_H */
