// kernels.h — launch interface between the C ABI (capi.cpp) and the HIP kernels.
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstdint>

#include "plan.h"

namespace cnl {

// device-resident copy of the plan (index data shared by every problem)
struct DevPlan {
  const FrontHdr* fronts;
  const int32_t* seg_ptr;
  const int32_t* asm_pos;
  const int32_t* asm_src;
  const int32_t* child_idx;
  const int32_t* rel_idx;
  const int32_t* perm;
  int32_t nsuper, N, nnz, rho_begin, nvar, nequ, ncon;
  int32_t fmax;
  int32_t pb_off;        // offset (doubles) of the panel buffer inside a problem's work area
  int32_t wv_off;        // offset of the two pivot-row staging vectors (2*fmax doubles)
  int32_t work_doubles;  // work area per problem
  int64_t lsize;         // factor storage per problem (doubles)
  int64_t vstride, rstride, dstride;  // per-problem strides (doubles) of vals / rhs / d
};

struct KernelConfig {
  int tpp = 64;          // threads per problem (16, 32, 64, 256, 1024)
  int ppb = 1;           // problems per workgroup
  int lds_work = 1;      // work area in LDS (else global scratch)
  size_t lds_bytes = 0;  // dynamic LDS per workgroup
};

// device-resident v2 plan (record streams, see plan.h / kernels2.hip)
struct DevPlan2 {
  const int32_t* rec;
  const int32_t* brec;
  int32_t nsuper, N, nnz, rho_begin, nvar;
  int32_t N0;            // length of the right-hand side the records address (outer N with direct records)
  int32_t count_d;       // 1: the fronts count the condensed residual pivots they own (no separate inertia pass)
  int32_t reccap;        // words of the forward record buffer (one per wave)
  int32_t breccap;       // words per backward record buffer (two per wave, same LDS area)
  int32_t recwords;      // words of that area = max(reccap, 2 * breccap)
  int32_t u2_peak;       // doubles: per-problem LDS update stack; the staging triangle follows it
  int32_t prob_doubles;  // doubles of LDS per problem
  int32_t jraw_off;      // doubles: offset of the raw-value area of the on-the-fly condensation inside the per-problem LDS
  int32_t bpanel_off;    // doubles: offset of the backward sweep's two panel buffers inside the per-problem LDS (behind the x stack)
  int64_t gs_doubles;    // doubles of global scratch per problem
  int64_t lsize;
  int64_t vstride, rstride, dstride;  // per-problem strides (doubles) of vals / rhs / d
};

// device-resident condensation lists (condense.h)
struct DevCond {
  const int32_t *c_ptr, *c_a, *c_b, *c_d, *c_order;
  const int32_t *ch_slot, *ch_rng, *ch_tile, *rng_start, *rng_len, *c_la, *c_lb, *c_ld, *ch_tptr, *tile_src;  // LDS-tiled condense (condense.h)
  const uint64_t* c_pack;
  int32_t tile_max, chunk_ncon_max, chunk_nslot_max, tiled_ok;
  const int32_t *r_dsrc, *r_ptr, *r_jsrc, *r_jx;
  const int32_t *red_of, *cidx_of, *orig_of, *r_orig;
  int32_t N, nnz, nvar, N2, ncs, ncond;
  int64_t cstride;
};

// Transposed-Jacobian lists of the KKT pattern: for every variable column j the entries of
// J_F and J_c in COO order, as (slot in vals, index into r resp. lambda).
struct DevJt {
  const int32_t *ptrF, *slotF, *idxF;  // [nvar + 1], [nnz(J_F)] x 2
  const int32_t *ptrC, *slotC, *idxC;  // [nvar + 1], [nnz(J_c)] x 2
  const int32_t *rptrC, *rslotC, *rcolC;  // J_c by rows: [ncon + 1], [nnz(J_c)] x 2 (slot in vals, variable) — CGLS (row f4)
  int32_t nvar, nequ, ncon, N, nnz;
  // (round 5) column tiles of row f1: when the entries of RVT_COLS consecutive columns lie in short slot / index ranges (band
  // and block patterns in COO order do), a workgroup streams those ranges of `vals`, r and lambda coalesced into LDS and forms the
  // column sums from there (residual_vectors_tiled_kernel).  rv_ntiles == 0: the gather kernel serves the pattern.
  const int32_t* rv_tiles;    // [rv_ntiles][RVT_TW]
  const uint32_t* rv_table;   // [rv_ntiles][RVT_KF + RVT_KC + 1][RVT_COLS]: window offsets of a column's first entries, counts
  int32_t rv_ntiles, rv_lds_doubles, rv_primal_tiles;
};
// where rows f1 / f4 read the Jacobian values: value of slot k (a position in `vals`) of problem b = vF[b * sF + k] for the J_F
// entries, vC[b * sC + k] for the J_c entries.  From `vals` itself: {vals, nnz, vals, nnz}; from the model's arrays Jx / Jcx (what
// prepare_newton_system! copies into those segments, src/CaNNOLeS.jl:968-974): {Jx - first J_F slot, nnz(J_F), Jcx - first J_c slot, nnz(J_c)}
struct JacSrc {
  const double* vF; long long sF;
  const double* vC; long long sC;
  int safeF, safeC;   // a slot of each kind that exists in every problem (the gather kernel's loads of absent entries; vC is never null)
};
constexpr int RVT_COLS = 256;   // columns per tile: one per thread of a 256-thread workgroup (two per thread: 190 registers)
constexpr int RVT_KF = 6, RVT_KC = 2;   // entries of a column held in the table (the rest of a longer column: index lists)
constexpr int RVT_MAXF = 2046, RVT_MAXR = 510, RVT_MAXC = 510, RVT_MAXL = 510;   // window limits (doubles): 4 + 1 + 1 + 1 chunks of 16 bytes per thread
constexpr int RVT_PROWS = 2048;  // rows per primal tile when the dual tiles cannot own the primal rows
enum { RVT_FSLO = 0, RVT_WF, RVT_RLO, RVT_WR, RVT_CSLO, RVT_WC, RVT_LLO, RVT_WL, RVT_OWNLO, RVT_OWNHI, RVT_TW = 12 };

enum { MODE_NEWTON = 0, MODE_FACTOR = 1, MODE_SOLVE = 2 };

struct LaunchArgs {
  int mode;
  int batch;
  double* vals;          // [batch][nnz]   (NEWTON: rho tail written back; FACTOR: read only)
  const double* rhs;     // [batch][N]     (NEWTON, SOLVE)
  int layout;            // band kernels, 32 problems per workgroup: bit 0 = vals, bit 1 = rhs interleaved over the workgroup's problems (band.h: band_il_index)
  double* d;             // [batch][N]     (NEWTON, SOLVE)
  double* L;             // [batch][lsize] factor storage
  double* scratch;       // [batch][work_doubles] when !lds_work
  double* rho_old;       // [batch] in/out (NEWTON)
  double* rho;           // [batch] out    (NEWTON)
  int32_t* nfact;        // [batch] out    (NEWTON)
  int32_t* success;      // [batch] out    (NEWTON: solve_success, FACTOR: success)
  int64_t* npos;         // [batch] optional (FACTOR)
  int64_t* nzero;        // [batch] optional (FACTOR)
  double params[9];      // ParamCaNNOLeS; params[0] = eig_tol also for FACTOR
  const int* extra_pos;  // [batch] optional: inertia counts of pivots eliminated outside the kernel (condensed r nodes)
  const int* extra_zer;
  // staged execution (latency plans, kernels2.hip STAGED): one launch per phase when `dep` is given (a task waits for its
  // children resp. its parent on device counters), else one launch per stage and phase
  const int32_t* tasks;  // device: 6 words per task {record offset, fronts, backward record offset, 1 if a root of the forest,
                         //                            parent task or -1, number of child tasks}
  int task0, ntasks;     // tasks of this launch
  int phase;             // 0: forward (assembly + elimination) of the tasks, 1: backward sweep of the tasks
  int nquads;            // groups of four problems
  int* gcnt;             // [batch][2] pivot counts summed over the tasks
  int skip_done;         // classic launch behind a staged attempt: problems with success[b] == 1 are left alone
  int ntasks_all;        // tasks of the plan (the counters of the backward phase follow those of the forward phase)
  int df_live;           // 1: this launch spans several stages (tasks wait on the counters; signals are released); 0: one stage,
                         //    its dependencies are complete by launch order (no wait, plain counter bump)
  int df_waves;          // the top stages of the tree whose wavefronts together number at most this run as ONE launch per phase
  int* dep;              // [2][tasks][nquads] dataflow counters (zeroed per call): children done (forward), task done (backward)
  // A dataflow wait that gives up after spin_limit polls bumps both counters: status_total is sticky (cnl_dataflow_timeouts),
  // status_call is zeroed with `dep` at the start of every staged call and makes the classic launch behind the staged attempt
  // redo the whole batch sequentially (only_if_status: that launch exits at once when the attempt had no timeout)
  int lean;              // 1: the plan qualifies for the kernels' LEAN instantiation (fast-class fronts, row-form products only)
  int back_rows;         // 1: (lean) the backward records carry the rows sections (plan.h, B_ROWS_FLAG): no post-pass behind the launch
  int* status_total;
  int* status_call;
  int spin_limit;
  int only_if_status;
  // In-kernel rho ladder of staged plans (phase == 2, kernels2.hip): ONE launch in which every task of the elimination tree of a
  // group of four problems has a wavefront of its own; per rung the tasks factorise in dataflow fashion, the last one to finish
  // applies the ladder rule of src/CaNNOLeS.jl:1029-1047 to the group's four problems and publishes the decision, the others
  // wait for it; the backward sweeps of the tasks follow in the same launch.  All wavefronts of a launch must be resident at
  // once (the host sizes lad_slots for that).
  int* lad;              // [nquads][LAD_WORDS] control block per group of four problems (zeroed per call), see kernels2.hip
  int* lgcnt;            // [batch][2] pivot counts of the current rung (zeroed per call, reset by the deciding wavefront)
  int* ldep;             // [2][tasks][nquads] counters of the fused launch: children done (monotone over the rungs), task done (backward)
  int lad_first;         // 1: the launch makes the first attempt too (rho as given); 0: only groups with a problem that failed the
                         //    staged first attempt are processed (the others exit at once)
  int lad_slots;         // groups of problems this launch covers, starting at quad0; wavefront index = task * lad_slots + slot
  int quad0;
  // host side of it (launch_newton2_staged): lad_mode 0 = no in-kernel ladder (the caller's sequential launch takes the failed
  // problems), 1 = the fused launch(es) behind the staged first attempt, 2 = the fused launch makes the first attempt too;
  // lad_capacity = wavefronts of this kernel the device holds at once; lad_zero_ints = ints to zero from gcnt on (gcnt, lgcnt,
  // lad and ldep are one allocation, zeroed with one memset per call)
  int lad_mode, lad_capacity;
  long long lad_zero_ints;
};
constexpr int LAD_WORDS = 32;  // ints per group: [0] tasks that finished the rung, [1] epoch (2 * rung + final), [4..7] state of the
                               // four problems (0 active, 1 factorised, 2 gave up), doubles at [8..15] rho last written to the
                               // slots ("wrote"; 0: never laddered), at [16..23] the rho_old to commit

// device-resident band program (band.h; kernels: band.hip)
struct BandDev {
  const int32_t* fops[2];
  const int32_t* bops[2];
  const int32_t* epochs[2];
  const int32_t* borders[2];
  int32_t nsteps[2], nepochs[2];
  long long loff[2];
  int32_t nparts, m0, n, N, nnz, nvar;
  long long lsize;
};
// newton_system! / try_to_factorize of a.batch problems on the band kernels, nl problems per workgroup (8, 16 or 32); a.L = the
// band factor storage [batch][P.lsize]
hipError_t launch_band(const BandDev& P, int nl, const LaunchArgs& a, hipStream_t stream);
size_t band_lds_bytes(int nparts, int nl);
// EXPERIMENT builds (-DCNL_EXPERIMENT=1 -DBAND_MW): the same program with loader wavefronts (band.hip, band_newton_mw_kernel)
hipError_t launch_band_mw(const BandDev& P, int variant, const LaunchArgs& a, hipStream_t stream);
size_t band_mw_lds_bytes(int variant);
int band_mw_group(int variant);   // problems per group of the variant (0: no such variant)

// returns hipSuccess or the launch error
hipError_t launch_newton(const DevPlan& P, const KernelConfig& cfg, const LaunchArgs& a, hipStream_t stream);
hipError_t launch_newton2(const DevPlan2& P, int wpb, size_t lds_bytes, const LaunchArgs& a, hipStream_t stream);
// one attempt at the rho given in vals, stage by stage (stage_ptr: host array of nstages + 1 task offsets)
hipError_t launch_newton2_staged(const DevPlan2& P, int wpb, size_t lds_bytes, LaunchArgs a, const int32_t* stage_ptr, int nstages, hipStream_t stream);
hipError_t launch_condense(const DevCond& C, const double* vals, const double* rhs, double* cbuf, int slot_begin, int slot_end,
                           int batch, hipStream_t stream);
// LDS-tiled variant over all chunks; mask: 1 matrix slots, 2 rho slots, 4 right-hand-side slots
hipError_t launch_condense_tiled(const DevCond& C, const double* vals, const double* rhs, double* cbuf, int mask, int nchunks,
                                 int batch, hipStream_t stream);
hipError_t launch_cond_inertia(const DevCond& C, const double* vals, int* extra_pos, int* extra_zer, double eig_tol, int batch,
                               hipStream_t stream);
hipError_t launch_prepare(int nnzhF, int nnzhc, int nnzjF, int nnzjc, int nvar, int nequ, int ncon, const double* hF, const double* hc,
                          const double* Jx, const double* Jcx, const double* delta, double* vals, int batch, int interleaved, hipStream_t stream);
// problem-major <-> interleaved over groups of 32 problems (band.h: band_il_index); rows of `len` doubles
hipError_t launch_interleave(const double* src, double* dst, int batch, long long len, int to_interleaved, hipStream_t stream);
hipError_t launch_cgls(const DevJt& J, const JacSrc& S, const double* r, double* lambda, double* Jxtr, double* ws, int32_t* iters,
                       double atol, double rtol, int itmax, int ones_if_zero, int batch, hipStream_t stream);
hipError_t launch_residual_vectors(const DevJt& J, const JacSrc& S, const double* r, const double* lambda, const double* Fx,
                                   const double* cx, double* rhs, double* norms, int batch, hipStream_t stream);
hipError_t launch_trial_point(const DevJt& J, const double* x, const double* r, const double* lambda, const double* d,
                              double max_dlambda, double* xt, double* rt, double* lambdat, double* dlambda, int batch,
                              hipStream_t stream);
hipError_t launch_fill_rho(double* vals, long long nnz, int nvar, const double* rho, const int* active, int batch, hipStream_t stream);
hipError_t launch_lds_fill(int pattern, hipStream_t stream);   // debugging aid, see kernels_aux.hip
hipError_t launch_expand(const DevCond& C, double* vals, const double* rhs, const double* d2, const double* cbuf, double* dout,
                         const int* success, int copy_rho_tail, int batch, hipStream_t stream);
// largest dynamic LDS a workgroup may use on the current device
size_t max_lds_bytes();
int lds_attr_cap(int need);   // value for hipFuncAttributeMaxDynamicSharedMemorySize: the device's limit (kernels2.hip)

}  // namespace cnl
