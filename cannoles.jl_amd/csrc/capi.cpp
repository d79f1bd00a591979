// capi.cpp — the C ABI declared in include/cannoles_hip.h.
//
// Host-side driver: owns the symbolic plan, uploads it once, chooses the kernel
// configuration for the front sizes at hand, stages host buffers for the
// host-pointer entry points and launches the fused kernel (kernels.hip).
// There is no CPU execution path: every numeric entry point needs the device.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/cannoles_hip.h"
#include "band.h"
#include "condense.h"
#include "dense.h"
#include "kernels.h"
#include "options.h"
#include "plan.h"

struct cnl_plan {
  cnl::Cond C;   // static condensation of the residual block (outer -> condensed system)
  cnl::Plan P;   // multifrontal plan of the (condensed) system
  int64_t N = 0, nnz = 0, nvar = 0, nequ = 0, ncon = 0;  // outer dimensions, as the reference sees them
  std::vector<int32_t> perm_outer;
  bool latency = false;  // ordered and cut into tasks for small batches (staged execution, csrc/plan.h)
  bool prefer_dense = false;  // latency plan with fronts of the 64 class on a small condensed system: small batches go the dense route
  cnl::DensePlan D;  // dense residual block (BASELINE config 2): served by the dense backend, csrc/dense.h
  std::vector<int32_t> gpos;  // non-empty: the condensed system may be treated as ONE dense matrix (position of every K2 slot)
  cnl::Tuning opt{};          // the switches the plan was built with (options.h; the handle reads its execution switches from here)
  std::atomic<int> refs{1};   // handles of a cnl_multi share one analysis (read-only after creation)
  bool split_mode = false;    // bidirectional-chain plan for a batch between one and two wavefronts per SIMD (capi.cpp, run_split)
  cnl::BandPlan band;         // (round 5) band program of a throughput plan (csrc/band.h); band.ok == false: the pattern is no band
  std::vector<int32_t> band_info, band_pinfo[2];
};

#ifndef CNL_PIPE_UPLOADERS
#define CNL_PIPE_UPLOADERS 2
#endif
struct cnl_handle {
  cnl_plan* plan = nullptr;
  int device = 0;
  int64_t batch = 1;
  std::vector<void*> dev_allocs;
  cnl::DevPlan dp{};
  cnl::KernelConfig cfg{};
  // v2 (register-front kernel): used for newton_system / factorize when every front has order <= 64
  bool use_v2 = false;
  bool staged = false;    // newton_system: first attempt stage by stage (tasks of the elimination tree on different wavefronts)
  const int32_t* d_tasks = nullptr;
  int* d_gcnt = nullptr;
  void* pin = nullptr;    // pinned host block for the results of small host-pointer calls
  size_t pin_bytes = 0;
  int* d_dep = nullptr;   // dataflow counters of the staged execution (nullptr: one launch per stage)
  int* d_status = nullptr;  // [1] dataflow waits that gave up (sticky; kernels2.hip spin_until)
  // counters of a staged call, ONE allocation behind d_gcnt, zeroed with one memset per call:
  // [gcnt 2B | lgcnt 2B | lad LAD_WORDS * nquads | ldep 2 * tasks * nquads | stat 2 | dep 2 * tasks * nquads (dataflow only)]
  int *d_lgcnt = nullptr, *d_lad = nullptr, *d_ldep = nullptr, *d_stat = nullptr;
  long long zero_ints = 0;
  int resident_waves = 0;   // wavefronts of the register-front kernel the device holds at once
  int lad_mode = 0;         // in-kernel rho ladder of staged newton_system calls (kernels.h): 0 none, 1 behind the staged attempt, 2 fused
  bool ladder_ran = false;  // the last launch_staged enqueued fused ladder launches (their commit / redo launch must follow)
  int ntasks = 0;
  int df_waves = 1024;
  std::vector<int32_t> stage_ptr;
  bool v2_solve = false;  // cnl_solve runs on the register-front kernel too (direct records, every front of the fast class)
  bool first_attempt_only = false;  // newton_system on a staged handle: no sequential launch behind the staged attempt (the host ladder follows)
  int* d_act = nullptr;   // [batch] problems whose rho slots the host ladder rewrites
  bool lean = false;      // every front of the fast class with row-form (or no) products: the kernels' LEAN instantiation serves it
  cnl::DevPlan2 dp2{};
  int wpb2 = 1;
  size_t lds2 = 0;
  double* d_gs = nullptr;
  // condensation state
  cnl::DevCond dc{};
  double* d_cbuf = nullptr;   // [batch][cstride]
  double* d_d2 = nullptr;     // [batch][N2]
  int *d_xpos = nullptr, *d_xzer = nullptr;
  const double* last_vals = nullptr;  // device vals of the last factorisation (needed to condense later right-hand sides)
  double* d_L = nullptr;
  double* d_scratch = nullptr;
  // staging for the host-pointer API
  double *d_vals = nullptr, *d_rhs = nullptr, *d_d = nullptr, *d_rho_old = nullptr, *d_rho = nullptr;
  int32_t *d_nfact = nullptr, *d_success = nullptr;
  int64_t *d_npos = nullptr, *d_nzero = nullptr;
  hipStream_t stream = nullptr;
  int64_t split_staged = 0;   // > 0: problems [0, split_staged) run staged, the rest single-stream, concurrently (run_split)
  bool split_halves = false;  // ... or (round 4): the rest runs staged as well, BEHIND the first part on the same stream (two halves)
  bool in_split = false;
  // (round 4) a batch a little above what fills the machine on the bidirectional chain (staged_max_batch < batch <= 5/4 of it):
  // problems [0, split_staged) run on this handle's chain plan, the REMAINDER on a handle of its own with the many-part latency
  // plan cnl_create picks for that small batch, one behind the other on the caller's stream (run_split)
  cnl_handle* tail = nullptr;
  bool tail_redone = false;   // (per call) a dataflow wait of the remainder handle gave up: its redo launch has been through the whole device
                              // ladder for those problems — the host ladder must leave them alone
  bool tail_fresh = false;    // the factors of the remainder live in the tail handle (false: in this handle's storage — chunked host calls)
  hipStream_t aux_stream = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  static constexpr int kPipeUp = CNL_PIPE_UPLOADERS;  // host threads that upload chunks of a host-pointer call (each on its own stream)
  hipStream_t pipe_stream[kPipeUp + 1] = {};  // chunked host-pointer calls: the uploaders' compute streams, one more for the results
  std::vector<hipEvent_t> pipe_ev;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  bool timing = false;
  float last_ms = 0.f;
  bool factorized = false;
  // success flags of the last HOST-pointer factorisation (cnl_factorize / cnl_newton_system), for cnl_solve: the reference never
  // solves after a failed factorisation (src/CaNNOLeS.jl:1049) — a one-problem cnl_solve then is a call-sequence error, and a
  // batched one leaves the rows of the failed problems untouched.  Unknown (empty) after a device-pointer factorisation.
  std::vector<char> last_ok;
  cnl::DevJt djt{};  // transposed-Jacobian lists (row f1: residual / optimality vectors on the device)
  cnl::DenseState* dense = nullptr;
  cnl::DenseState* gdense = nullptr;  // dense treatment of an arbitrary condensed system (irregular sparsity, small batch)
  cnl::GeneralOps gops{};
  double* d_cgls_ws = nullptr;  // [batch][2 * nvar] workspace of cnl_cgls_multipliers_dev, allocated on first use
  // (round 5) band kernels (csrc/band.h): newton_system of a throughput handle whose pattern is a band
  bool band = false;
  cnl::BandDev bd{};
  int band_nl = 16;            // problems per workgroup
  bool jac_segments = false;   // the J_F and the J_c entries are one run of slots each: [jf_lo, jf_lo + jf_n), [jc_lo, jc_lo + jc_n)
  int64_t jf_lo = 0, jf_n = 0, jc_lo = 0, jc_n = 0;
  int layout = 0;              // band handles: bit 0 = vals (cnl_options.batch_layout), bit 1 = rhs interleaved over groups of 32 problems (band.h)
  int band_mw = 0;             // EXPERIMENT builds: the kernel with loader wavefronts serves the handle (band.hip, band_newton_mw_kernel)
  double* d_Lband = nullptr;   // [batch][bd.lsize] factor records of the band kernels
};

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& m) {
  g_err = m;
  return code;
}

#define HIPCHK(expr)                                                                                   \
  do {                                                                                                 \
    hipError_t e_ = (expr);                                                                            \
    if (e_ != hipSuccess) return fail(CNL_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)

template <class T>
int upload(cnl_handle* h, const std::vector<T>& v, const T** out) {
  void* p = nullptr;
  size_t bytes = std::max<size_t>(v.size(), 1) * sizeof(T);
  HIPCHK(hipMalloc(&p, bytes));
  h->dev_allocs.push_back(p);
  if (!v.empty()) HIPCHK(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  *out = (const T*)p;
  return CNL_OK;
}

template <class T>
int dalloc(cnl_handle* h, T** out, size_t count) {
  void* p = nullptr;
  static const size_t G = getenv("CNL_DBG_GUARD") ? (size_t)atol(getenv("CNL_DBG_GUARD")) : 0;   // debugging aid: NaN-filled guard zones
  const size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
  HIPCHK(hipMalloc(&p, bytes + 2 * G));
  if (G) HIPCHK(hipMemset(p, getenv("CNL_DBG_GUARD_PAT") ? atoi(getenv("CNL_DBG_GUARD_PAT")) : 0xFF, bytes + 2 * G));
  h->dev_allocs.push_back(p);
  *out = (T*)(static_cast<char*>(p) + G);
  if (G && getenv("CNL_DBG_GUARD_LOG")) fprintf(stderr, "[dalloc] #%zu %p + %zu bytes\n", h->dev_allocs.size(), (void*)*out, bytes);
  return CNL_OK;
}

int choose_config(cnl_handle* h) {
  const cnl::Plan& P = h->plan->P;
  cnl::DevPlan& dp = h->dp;
  int64_t pb_off = std::max<int64_t>(P.fwd_peak, P.bwd_peak);
  pb_off = (pb_off + 1) & ~(int64_t)1;
  int64_t wv_off = pb_off + ((P.panel_max + 1) & ~1);
  int64_t work = wv_off + 2 * (int64_t)((P.fmax + 1) & ~1);
  if (work >= ((int64_t)1 << 30)) return fail(CNL_ERR_DIM, "work area too large");
  dp.pb_off = (int32_t)pb_off;
  dp.wv_off = (int32_t)wv_off;
  dp.work_doubles = (int32_t)work;
  size_t maxlds = cnl::max_lds_bytes();
  if (maxlds == 0) return fail(CNL_ERR_HIP, "cannot query LDS size (no HIP device?)");
  maxlds = std::min<size_t>(maxlds, 160 * 1024);
  cnl::KernelConfig& c = h->cfg;
  const size_t hdr = 16 * sizeof(double);
  const size_t per = (size_t)work * sizeof(double);
  int tpp, ppb, ldsw;
  if (hdr + per <= maxlds) {
    ldsw = 1;
    if (P.fmax <= 96) {
      tpp = 64;
      // problems per workgroup: leave room for two workgroups per CU when the work area
      // allows it, and spread small batches over the 256 CUs
      size_t fit = (maxlds - hdr) / per;
      size_t fit2 = maxlds / 2 > hdr ? (maxlds / 2 - hdr) / per : 0;
      size_t cap = fit2 >= 1 ? fit2 : fit;
      size_t want = (size_t)std::min<int64_t>(16, std::max<int64_t>(1, h->batch / 256));
      ppb = 1;
      for (int cand : {16, 8, 4, 2, 1})
        if ((size_t)cand <= cap && (size_t)cand <= want) { ppb = cand; break; }
    } else {
      tpp = P.fmax <= 400 ? 256 : 1024;
      ppb = 1;
    }
  } else {
    ldsw = 0;
    tpp = P.fmax <= 96 ? 64 : (P.fmax <= 400 ? 256 : 1024);
    ppb = tpp == 64 ? 4 : 1;
  }
  const cnl::Tuning& o = h->plan->opt;
  if (o.v1_tpp > 0) tpp = o.v1_tpp;
  if (o.v1_ppb > 0) ppb = o.v1_ppb;
  if (o.v1_lds >= 0) ldsw = o.v1_lds;
  c.tpp = tpp; c.ppb = ppb; c.lds_work = ldsw;
  c.lds_bytes = hdr + (ldsw ? (size_t)ppb * per : 0);
  if (c.lds_bytes > maxlds) return fail(CNL_ERR_DIM, "kernel configuration exceeds LDS");
  return CNL_OK;
}

int setup_v2(cnl_handle* h) {
  const cnl::Plan& P = h->plan->P;
  h->use_v2 = false;
  const cnl::Tuning& o = h->plan->opt;
  if (!P.v2_ok || !o.register_front) return CNL_OK;
  if (!h->plan->gpos.empty() && (o.general_dense == 2 || (h->plan->prefer_dense && h->batch <= 16))) return CNL_OK;  // the dense route (plan_create_impl); 2: wherever it is possible
  cnl::DevPlan2& d = h->dp2;
  // streams are over-read by the prefetcher: pad with zeros
  std::vector<int32_t> rec(P.rec), brec(P.brec);
  rec.resize(rec.size() + 2048, 0);
  brec.resize(brec.size() + 2048, 0);
  int rc;
  if ((rc = upload(h, rec, &d.rec))) return rc;
  if ((rc = upload(h, brec, &d.brec))) return rc;
  d.nsuper = P.nsuper; d.N = (int32_t)P.N; d.nnz = (int32_t)P.nnz; d.rho_begin = P.rho_begin; d.nvar = (int32_t)P.nvar;
  d.N0 = (int32_t)P.N;
  d.reccap = (P.rec_maxlen + 64 + 3) & ~3;  // + slack: the product loop reads up to 48 words past a list
  d.breccap = (P.brec_maxlen + 3) & ~3;
  d.recwords = std::max(d.reccap, 2 * d.breccap);
  d.u2_peak = P.u2_peak;
  // (round 5, found by the randomised run with lds_pad = 0: the out-of-line elimination of a class-64 front publishes its pivot row at
  //  lb[0 .. 65] of the staging area — lanes beyond the pivot park their value at index TE + 1 —, two doubles more than the 64 reserved
  //  here; without padding between the problems they landed in the next problem's update stack)
  d.jraw_off = (int32_t)((P.u2_peak + std::max<int64_t>(P.fs2_max, 72) + 1) & ~(int64_t)1);
  d.bpanel_off = (int32_t)((P.bwd_peak + 2 + 1) & ~(int64_t)1);  // end of the backward sweep's x stack
  // the raw-value area (128 doubles per problem) is needed only by fast fronts whose products come as lists (plan.h: RF_ROWS)
  int64_t prob = std::max<int64_t>((int64_t)d.jraw_off + (P.rec_direct && P.listprod_fronts > 0 ? 128 : 0), (int64_t)d.bpanel_off);
  // per-problem areas 32 banks apart modulo 64 (prob_doubles = 16 mod 32): the 16 lanes of two neighbouring problems
  // then touch disjoint LDS banks when they read the same row of their images (env CNL_LDS_PAD=0 disables)
  prob = (prob + 1) & ~(int64_t)1;
  if (o.lds_pad) while (prob % 32 != 16) prob += 2;
  d.prob_doubles = (int32_t)prob;
  d.gs_doubles = P.gs_doubles + 64;
  d.lsize = h->dp.lsize;  // padded stride, see cnl_create
  d.vstride = h->dp.vstride; d.rstride = h->dp.rstride; d.dstride = h->dp.dstride;
  if (P.rec_direct) {  // the assembly lists address the caller's arrays
    d.nnz = P.nnz_outer; d.rho_begin = P.nnz_outer - (int32_t)P.nvar;
    d.vstride = P.nnz_outer; d.rstride = P.n_outer;
    d.N0 = P.n_outer;
    if (P.d_outer) d.dstride = P.n_outer;
    d.count_d = P.d_owned == (int64_t)h->plan->C.r_dsrc.size() ? 1 : 0;  // every condensed pivot is staged by some front
  }
  // the kernel addresses vals / rhs / L of the 4 problems of a wave with 32-bit byte offsets from the first one
  if (4 * 8 * (uint64_t)std::max<int64_t>({d.lsize, d.vstride, d.rstride, d.dstride, (int64_t)d.nnz + d.N0}) >= (1ull << 32)) return CNL_OK;
  const size_t wave_bytes = ((size_t)(d.recwords >> 1) + 4 * (size_t)d.prob_doubles + 16) * sizeof(double);   // + 16: counters, flags, slow_front's scalars (kernels2.hip)
  size_t maxlds = std::min<size_t>(cnl::max_lds_bytes(), 160 * 1024);
  if (wave_bytes + 512 > maxlds) return CNL_OK;  // does not fit: stay on v1
  // waves per workgroup: small workgroups give the dispatcher freedom; 2 keeps the launch grid moderate
  int wpb = 1;
  if (o.waves_per_block > 0) wpb = std::max(1, std::min(4, o.waves_per_block));
  while (wpb > 1 && wpb * wave_bytes + 512 > maxlds) wpb--;
  h->wpb2 = wpb;
  h->lds2 = wpb * wave_bytes + 512;
  if ((rc = dalloc(h, &h->d_gs, (size_t)h->batch * (size_t)d.gs_doubles))) return rc;
  h->use_v2 = true;
  h->staged = false;
  if ((rc = dalloc(h, &h->d_status, 1))) return rc;
  if (hipMemset(h->d_status, 0, sizeof(int)) != hipSuccess) return fail(CNL_ERR_HIP, "hipMemset failed");
  {
    // wavefronts of this kernel the device holds at once: two per SIMD by the register budget of every instantiation
    // (profiles/r04_kernel_resources.txt), and what the LDS of a CU holds
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, h->device) == hipSuccess && prop.multiProcessorCount > 0) {
      const size_t per_wg = (size_t)wpb * wave_bytes + 512;
      const long long by_lds = (long long)(std::min<size_t>(prop.maxSharedMemoryPerMultiProcessor ? prop.maxSharedMemoryPerMultiProcessor : maxlds, 160 * 1024) / per_wg) * wpb;
      const long long resident = (long long)prop.multiProcessorCount * std::max<long long>(1, std::min<long long>(8, by_lds));
      h->resident_waves = (int)std::min<long long>(resident, 1 << 20);
    }
  }
  // Rounds 4 - 5, for the record (profiles/HISTORY.md 4b item 8, 4c): staged handles on plans with out-of-line front classes (order 17 .. 64) gave
  // history-dependent wrong decisions and memory faults.  Three causes, all found with garbage left in LDS / scratch / registers in
  // front of every launch (CNL_DBG_SCRATCHFILL, CNL_DBG_LDSFILL) and tools/fuzz_parity.py: the update-matrix slots of the global scratch
  // were padded for 16-lane rows whatever the class of the front (analysis.cpp); the class-64 elimination publishes its pivot row two
  // doubles past the LDS staging area (setup above); and — the one that survived both — the compiler placed the register spills of the
  // call to the out-of-line front path IN FRONT of the EXEC restore of the join block behind the lane-divergent `if (dep_wait)`, so a
  // task with nothing to wait for stored no spills and reloaded garbage (kernels2.hip: DEP_WAITING; tools/check_spill_exec.py checks the
  // ISA of every build for the pattern).  No restriction is left: every stageable plan runs staged, with the in-kernel ladder.
  if (!P.tasks.empty() && P.rec_direct && P.d_outer && d.count_d && o.staged) {
    std::vector<int32_t> tk;
    for (const cnl::Task& t : P.tasks) { tk.push_back(t.rec_off); tk.push_back(t.f1 - t.f0); tk.push_back(t.brec_off); tk.push_back(t.is_root); tk.push_back(t.parent); tk.push_back(t.nchild); }
    if ((rc = upload(h, tk, &h->d_tasks))) return rc;
    // dataflow execution: per (task, group of four problems) a count of finished children (forward) and a done flag (backward)
    h->ntasks = (int)P.tasks.size();
    // (measured, tools/sweep_dataflow.py: one system 0.132 against 0.165 ms, eight 0.177 against 0.193 ms; cfg4's pattern with
    //  29 tasks: 32 problems 0.110 against 0.131 ms.  With more wavefronts than about half the machine's slots the waiting ones
    //  crowd out the working ones — cfg3, 501 tasks: sixteen problems 0.233 against 0.196 ms, 256: 82 k against 367 k systems/s —
    //  so only the top stages whose tasks x groups of problems number at most 1024 run that way; env CNL_DATAFLOW_WAVES)
    // a wavefront that waits occupies its slot: never more waiting wavefronts than the device holds at once (the scheme
    // relies on the lowest unfinished workgroup being resident; kernels2.hip)
    h->df_waves = o.dataflow_waves > 0 ? o.dataflow_waves : 1024;
    if (h->resident_waves > 0) h->df_waves = std::min(h->df_waves, h->resident_waves);
    {
      const size_t B = (size_t)h->batch, nq = (B + 3) / 4, tq = 2 * (size_t)h->ntasks * nq;
      const size_t total = 4 * B + (size_t)cnl::LAD_WORDS * nq + tq + 2 + (o.dataflow ? tq : 0);
      if ((rc = dalloc(h, &h->d_gcnt, total))) return rc;
      if (hipMemset(h->d_gcnt, 0, total * sizeof(int)) != hipSuccess) return fail(CNL_ERR_HIP, "hipMemset failed");
      h->d_lgcnt = h->d_gcnt + 2 * B;
      h->d_lad = h->d_lgcnt + 2 * B;
      h->d_ldep = h->d_lad + (size_t)cnl::LAD_WORDS * nq;
      h->d_stat = h->d_ldep + tq;   // per-call status words (kernels2.hip, spin_until)
      if (o.dataflow) h->d_dep = h->d_stat + 2;
      h->zero_ints = (long long)total;
    }
    h->stage_ptr = P.stage_ptr;
    h->staged = true;
    // The in-kernel rho ladder needs all tasks of a group of four problems resident at once.  With more groups than the device
    // holds the fused launch is repeated over ranges of groups; beyond four such launches the sequential launch keeps the job
    // (plans of very many tasks on batches that large do not occur: the planner gives large batches few, large tasks).
    h->lad_mode = 0;
    // Plans of a few LARGE tasks (the bidirectional chain of mid-size batches) keep the sequential launch when one fused launch
    // cannot hold the batch: a rung there is the same chain of fronts either way, and two fused launches of two wavefronts per SIMD
    // lose to one sequential launch of one (cfg5 at 4096 problems: 3.9 against 2.9 ms).
    if (o.device_ladder && h->resident_waves >= h->ntasks) {
      const long long slots = h->resident_waves / h->ntasks, nq = (h->batch + 3) / 4;
      const long long launches = (nq + slots - 1) / slots;
      if (launches == 1 || (launches <= 4 && h->ntasks >= 16)) h->lad_mode = o.device_ladder_fused ? 2 : 1;
    }
  }
  h->v2_solve = P.rec_direct && P.d_outer && P.ncls[1] == 0 && P.ncls[2] == 0 && !o.v1_solve;
  h->lean = o.lean_kernel && P.rec_direct && P.d_outer && d.count_d && P.ncls[1] == 0 && P.ncls[2] == 0 && P.listprod_fronts == 0;
  return CNL_OK;
}

// debugging aid (include/cannoles_hip.h): CNL_DBG_LDSFILL=<byte pattern> — kernels that leave the pattern in LDS, scratch and registers
// run in front of every launch.  The variable is read ONCE per process (round 6: it was a getenv per launch in the product build).
const int* dbg_ldsfill() {
  static const int pat = [] { const char* e = getenv("CNL_DBG_LDSFILL"); return e ? (int)strtol(e, nullptr, 0) : -1; }();
  static const bool on = getenv("CNL_DBG_LDSFILL") != nullptr;
  return on ? &pat : nullptr;
}

// launches per kernel family since the library was loaded (cnl_launch_counts): [band kernels, register-front kernel, general kernel]
std::atomic<long long> g_launches[3];

int launch(cnl_handle* h, cnl::LaunchArgs& a, hipStream_t stream) {
  if (const int* pat = dbg_ldsfill()) (void)cnl::launch_lds_fill(*pat, stream);
  a.batch = (int)h->batch;
  a.lean = h->lean ? 1 : 0;
  a.back_rows = (h->lean && h->plan->P.back_rows) ? 1 : 0;
  a.L = h->d_L;
  a.scratch = h->d_scratch;
  hipError_t e;
  if (h->timing) HIPCHK(hipEventRecord(h->ev0, stream));  // events bracket the multifrontal kernel only
  if (h->band && !a.skip_done && !a.only_if_status && (a.mode != cnl::MODE_NEWTON || a.rhs)) {
    // band handles: all three calls of the plugin surface run on the band kernels (round 6).  try_to_factorize is the forward
    // sweep alone; solve_ldl! factorises the values of the last factorisation again (a.vals = the handle's last_vals, rho slots as
    // the ladder left them) and sweeps the new right-hand side in the same launch — the band kernels' six-double records hold
    // z = c / d of the one right-hand side they were computed with, so there is no stored factor a second right-hand side could use,
    // and the register-front refactorisation rounds 5 put in front of such a solve (a second, slower kernel whose flags nobody read)
    // is gone
    cnl::LaunchArgs b = a;
    b.L = h->d_Lband;
    b.layout = h->layout;
    e = h->band_mw ? cnl::launch_band_mw(h->bd, h->band_mw, b, stream) : cnl::launch_band(h->bd, h->band_nl, b, stream);
    g_launches[0]++;
  } else if (h->layout) {
    return fail(CNL_ERR_STATE, "this call is not served by the band kernels: a handle with batch_layout = CNL_LAYOUT_INTERLEAVED has no other");
  } else if (h->use_v2 && (a.mode != cnl::MODE_SOLVE || h->v2_solve)) {
    a.scratch = h->d_gs;
    e = cnl::launch_newton2(h->dp2, h->wpb2, h->lds2, a, stream);
    g_launches[1]++;
  } else {
    e = cnl::launch_newton(h->dp, h->cfg, a, stream);
    g_launches[2]++;
  }
  if (e != hipSuccess)
    return fail(CNL_ERR_HIP, std::string("kernel launch (tpp=") + std::to_string(h->cfg.tpp) + " ppb=" + std::to_string(h->cfg.ppb) +
                                 " lds=" + std::to_string(h->cfg.lds_work) + "): " + hipGetErrorString(e));
  if (h->timing) HIPCHK(hipEventRecord(h->ev1, stream));
  return CNL_OK;
}

// one staged pass over the tasks of a latency plan (first attempt of newton_system, try_to_factorize, or solve_ldl!)
int launch_staged(cnl_handle* h, cnl::LaunchArgs& a, hipStream_t stream) {
  if (const int* pat = dbg_ldsfill()) (void)cnl::launch_lds_fill(*pat, stream);
  a.batch = (int)h->batch; a.L = h->d_L; a.scratch = h->d_gs;
  a.tasks = h->d_tasks; a.gcnt = h->d_gcnt; a.skip_done = 0; a.dep = h->d_dep; a.df_waves = h->df_waves;
  a.lean = h->lean ? 1 : 0;
  a.back_rows = (h->lean && h->plan->P.back_rows) ? 1 : 0;
  a.status_total = h->d_status;
  a.status_call = h->d_stat;   // (nullptr in views of the handle: one launch per stage, nothing waits)
  a.lad = h->d_lad; a.lgcnt = h->d_lgcnt; a.ldep = h->d_ldep; a.lad_zero_ints = h->d_stat ? h->zero_ints : 0;
  a.lad_capacity = h->resident_waves;
  a.lad_mode = (a.mode == cnl::MODE_NEWTON && h->d_lad && !h->first_attempt_only) ? h->lad_mode : 0;
  h->ladder_ran = a.lad_mode != 0;
  a.spin_limit = h->plan->opt.dataflow_spin_limit > 0 ? h->plan->opt.dataflow_spin_limit : (1 << 22);
  if (h->timing) HIPCHK(hipEventRecord(h->ev0, stream));
  hipError_t e = cnl::launch_newton2_staged(h->dp2, h->wpb2, h->lds2, a, h->stage_ptr.data(), (int)h->stage_ptr.size() - 1, stream);
  if (e != hipSuccess) return fail(CNL_ERR_HIP, std::string("staged launch: ") + hipGetErrorString(e));
  return CNL_OK;
}

// run() on problems [b0, b0 + nb) of the handle: the base pointer of every per-problem device array of the handle is moved to
// problem b0 and the batch set to nb for the lifetime of the view (b0 a multiple of 4: a wavefront serves four problems).
// Dataflow counters are per handle, not per view: views run one launch per stage.
struct SubBatch {
  cnl_handle* h;
  int64_t batch;
  double *L, *gs, *scratch, *cbuf, *d2, *Lband;
  int *xpos, *xzer, *gcnt, *dep, *lad, *stat;
  const double* last_vals;
  bool staged;
  SubBatch(cnl_handle* h_, int64_t b0, int64_t nb, bool allow_staged = true) : h(h_) {
    last_vals = h->last_vals; staged = h->staged;
    if (h->last_vals) h->last_vals += b0 * h->plan->nnz;
    if (!allow_staged) h->staged = false;
    batch = h->batch; L = h->d_L; gs = h->d_gs; scratch = h->d_scratch; cbuf = h->d_cbuf; d2 = h->d_d2;
    xpos = h->d_xpos; xzer = h->d_xzer; gcnt = h->d_gcnt; dep = h->d_dep; lad = h->d_lad; stat = h->d_stat;
    Lband = h->d_Lband;
    if (h->d_Lband) h->d_Lband += b0 * h->bd.lsize;
    const cnl::Cond& C = h->plan->C;
    h->batch = nb;
    h->d_L += b0 * h->dp.lsize;
    if (h->d_gs) h->d_gs += b0 * h->dp2.gs_doubles;
    if (h->d_scratch) h->d_scratch += b0 * (int64_t)h->dp.work_doubles;
    if (h->d_cbuf) h->d_cbuf += b0 * C.cstride;
    if (h->d_d2) h->d_d2 += b0 * C.N2;
    if (h->d_xpos) h->d_xpos += b0;
    if (h->d_xzer) h->d_xzer += b0;
    if (h->d_gcnt) h->d_gcnt += 2 * b0;
    h->d_dep = nullptr; h->d_lad = nullptr; h->d_stat = nullptr;   // views run one launch per stage and keep the sequential ladder
  }
  ~SubBatch() {
    h->batch = batch; h->d_L = L; h->d_gs = gs; h->d_scratch = scratch; h->d_cbuf = cbuf; h->d_d2 = d2;
    h->d_xpos = xpos; h->d_xzer = xzer; h->d_gcnt = gcnt; h->d_dep = dep; h->d_lad = lad; h->d_stat = stat;
    h->last_vals = last_vals; h->staged = staged; h->d_Lband = Lband;
  }
};

int run(cnl_handle* h, cnl::LaunchArgs& a, double* d_vals, const double* d_rhs, double* d_d, hipStream_t stream);

// Batches between one and two wavefronts per SIMD (4096 .. 8192 problems of cfg3's size): the single stream gives a group of
// four problems ONE wavefront for 1000 fronts, the bidirectional chain TWO for 500 each, and the machine holds 2048 wavefronts.
// With x groups on the chain and y on the stream, 2 x + y = 2048 fills every slot whatever the batch: the chain part runs staged
// on the caller's stream, the rest single-stream on a second stream of the handle, forked and joined with events (no host
// synchronisation).  Both parts use the SAME plan — the chain order has the throughput order's fronts, and the classic launch
// runs any plan's records from end to end (it already does behind every staged attempt).
int run_split(cnl_handle* h, cnl::LaunchArgs& a, double* d_vals, const double* d_rhs, double* d_d, hipStream_t stream) {
  if (!h->aux_stream && !h->tail && !h->split_halves) {
    HIPCHK(hipStreamCreateWithFlags(&h->aux_stream, hipStreamNonBlocking));
    HIPCHK(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
  }
  const int64_t nA = h->split_staged, nB = h->batch - nA, nnz = h->plan->nnz, N = h->plan->N;
  const bool tm = h->timing;
  if (tm) HIPCHK(hipEventRecord(h->ev0, stream));
  h->timing = false;
  h->in_split = true;
  if (h->tail) {
    // 4096 problems fill every wavefront slot on the bidirectional chain (4.1 ms at cfg3's size); a remainder of r <= 1024 problems
    // takes 0.5 .. 1.7 ms on its own many-part plan, where two halves of the whole batch need 2 x 3 ms (4608 problems: 5.95 -> 5.0 ms)
    cnl_handle* t = h->tail;
    const bool on_tail = a.mode != cnl::MODE_SOLVE || h->tail_fresh;
    int rc;
    {
      SubBatch view(h, 0, nA, true);
      cnl::LaunchArgs b = a;
      rc = run(h, b, d_vals, d_rhs, d_d, stream);
    }
    if (rc == CNL_OK) {
      cnl::LaunchArgs b = a;
      if (b.rho_old) b.rho_old += nA;
      if (b.rho) b.rho += nA;
      if (b.nfact) b.nfact += nA;
      if (b.success) b.success += nA;
      if (b.npos) b.npos += nA;
      if (b.nzero) b.nzero += nA;
      double* tv = d_vals ? d_vals + nA * nnz : nullptr;
      const double* tr = d_rhs ? d_rhs + nA * N : nullptr;
      double* td = d_d ? d_d + nA * N : nullptr;
      if (on_tail) {
        t->first_attempt_only = h->first_attempt_only;
        rc = run(t, b, tv, tr, td, stream);
        t->first_attempt_only = false;
        if (rc == CNL_OK && a.mode != cnl::MODE_SOLVE) { t->last_vals = tv; t->factorized = true; h->tail_fresh = true; }
      } else {
        SubBatch view(h, nA, nB, true);
        rc = run(h, b, tv, tr, td, stream);
      }
    }
    h->in_split = false;
    h->timing = tm;
    if (rc) return rc;
    if (a.mode == cnl::MODE_FACTOR) h->last_vals = d_vals;
    if (tm) {
      HIPCHK(hipEventRecord(h->ev1, stream));
      HIPCHK(hipEventSynchronize(h->ev1));
      HIPCHK(hipEventElapsedTime(&h->last_ms, h->ev0, h->ev1));
    }
    return CNL_OK;
  }
  if (h->split_halves) {
    // Two halves, each on the bidirectional chain (two wavefronts per group of problems), one behind the other on the caller's
    // stream: a half of 2304 .. 3840 problems runs at 0.81 .. 0.96 M systems/s, where the single-stream part of the concurrent
    // split needs its >= 6 ms however few problems it holds (4608 problems: 6.5 ms = 707 k systems/s; two halves: ~5.7 ms).
    int rc = CNL_OK;
    for (int part = 0; part < 2 && rc == CNL_OK; part++) {
      const int64_t b0 = part ? nA : 0, nb = part ? nB : nA;
      SubBatch view(h, b0, nb, true);
      cnl::LaunchArgs b = a;
      if (b.rho_old) b.rho_old += b0;
      if (b.rho) b.rho += b0;
      if (b.nfact) b.nfact += b0;
      if (b.success) b.success += b0;
      if (b.npos) b.npos += b0;
      if (b.nzero) b.nzero += b0;
      rc = run(h, b, d_vals ? d_vals + b0 * nnz : nullptr, d_rhs ? d_rhs + b0 * N : nullptr, d_d ? d_d + b0 * N : nullptr, stream);
    }
    h->in_split = false;
    h->timing = tm;
    if (rc) return rc;
    if (a.mode == cnl::MODE_FACTOR) h->last_vals = d_vals;
    if (tm) {
      HIPCHK(hipEventRecord(h->ev1, stream));
      HIPCHK(hipEventSynchronize(h->ev1));
      HIPCHK(hipEventElapsedTime(&h->last_ms, h->ev0, h->ev1));
    }
    return CNL_OK;
  }
  HIPCHK(hipEventRecord(h->ev_fork, stream));
  HIPCHK(hipStreamWaitEvent(h->aux_stream, h->ev_fork, 0));
  int rc;
  {
    SubBatch view(h, nA, nB, false);
    cnl::LaunchArgs b = a;
    if (b.rho_old) b.rho_old += nA;
    if (b.rho) b.rho += nA;
    if (b.nfact) b.nfact += nA;
    if (b.success) b.success += nA;
    if (b.npos) b.npos += nA;
    if (b.nzero) b.nzero += nA;
    rc = run(h, b, d_vals ? d_vals + nA * nnz : nullptr, d_rhs ? d_rhs + nA * N : nullptr, d_d ? d_d + nA * N : nullptr, h->aux_stream);
  }
  if (rc == CNL_OK) {
    SubBatch view(h, 0, nA, true);
    rc = run(h, a, d_vals, d_rhs, d_d, stream);
  }
  h->in_split = false;
  h->timing = tm;
  // join also when an enqueue failed: work already on the second stream must not overlap a later call's use of the handle's arrays
  const hipError_t je = hipEventRecord(h->ev_join, h->aux_stream);
  const hipError_t we = je == hipSuccess ? hipStreamWaitEvent(stream, h->ev_join, 0) : je;
  if (rc) { if (we != hipSuccess) (void)hipStreamSynchronize(h->aux_stream); return rc; }
  HIPCHK(we);
  if (a.mode == cnl::MODE_FACTOR) h->last_vals = d_vals;
  if (tm) {
    HIPCHK(hipEventRecord(h->ev1, stream));
    HIPCHK(hipEventSynchronize(h->ev1));
    HIPCHK(hipEventElapsedTime(&h->last_ms, h->ev0, h->ev1));
  }
  return CNL_OK;
}

// Behind a staged try_to_factorize / solve_ldl! that ran in dataflow fashion: the sequential execution of the same call, which
// exits at once unless a dataflow wait of the attempt gave up (kernels2.hip, spin_until).  newton_system has its classic launch
// anyway (the rho ladder of the problems that failed the first attempt).
int launch_redo(cnl_handle* h, cnl::LaunchArgs& a, hipStream_t stream) {
  if (!a.status_call || (!h->d_dep && !h->ladder_ran)) return CNL_OK;  // one launch per stage: nothing waits, nothing can time out
  const bool tm = h->timing;
  h->timing = false;
  a.only_if_status = 1;
  const int rc = launch(h, a, stream);
  a.only_if_status = 0;
  h->timing = tm;
  return rc;
}

// One call of the path on device-resident data: [condense ->] multifrontal kernel [-> expand].
int run(cnl_handle* h, cnl::LaunchArgs& a, double* d_vals, const double* d_rhs, double* d_d, hipStream_t stream) {
  const cnl::Cond& C = h->plan->C;
  int rc = CNL_OK;
  if (h->split_staged > 0 && !h->in_split && (h->staged || h->tail) && h->split_staged < h->batch) return run_split(h, a, d_vals, d_rhs, d_d, stream);
  if (h->dense || h->gdense)
    if (const int* pat = dbg_ldsfill()) (void)cnl::launch_lds_fill(*pat, stream);
  if (h->dense) {
    // dense residual block: J'WJ + tiled dense LDL^T on the fp64 matrix cores (csrc/dense.hip); asynchronous, the rho ladder
    // is decided on the device
    std::string err;
    if (h->timing) HIPCHK(hipEventRecord(h->ev0, stream));
    rc = cnl::dense_run(h->dense, h->plan->D, a.mode, d_vals, d_rhs, d_d, a.rho_old, a.rho, a.nfact, a.success, a.npos, a.nzero,
                        a.params, stream, err);
    if (rc) return fail(rc == 5 ? CNL_ERR_STATE : CNL_ERR_HIP, "dense backend: " + err);
    if (h->timing) {
      HIPCHK(hipEventRecord(h->ev1, stream));
      HIPCHK(hipEventSynchronize(h->ev1));
      HIPCHK(hipEventElapsedTime(&h->last_ms, h->ev0, h->ev1));
    }
    return CNL_OK;
  }
  if (h->gdense) {
    // condensed system as one dense matrix: condense pass -> dense LDL^T / solves (csrc/dense.hip) -> post-pass
    const int B = (int)h->batch;
    const int s_mat = (int)(C.ncs + C.nvar), s_all = (int)C.cstride;
    hipError_t e = hipSuccess;
    std::string err;
    if (h->timing) HIPCHK(hipEventRecord(h->ev0, stream));
    if (a.mode == cnl::MODE_SOLVE) {
      if (!h->last_vals) return fail(CNL_ERR_STATE, "cnl_solve before cnl_factorize");
      e = C.tiled_ok ? cnl::launch_condense_tiled(h->dc, h->last_vals, d_rhs, h->d_cbuf, 4, C.ch_region[3], B, stream)
                     : cnl::launch_condense(h->dc, h->last_vals, d_rhs, h->d_cbuf, s_mat, s_all, B, stream);
    } else {
      const bool nw = a.mode == cnl::MODE_NEWTON;
      e = C.tiled_ok ? cnl::launch_condense_tiled(h->dc, d_vals, nw ? d_rhs : nullptr, h->d_cbuf, nw ? 7 : 3, C.ch_region[3], B, stream)
                     : cnl::launch_condense(h->dc, d_vals, nw ? d_rhs : nullptr, h->d_cbuf, 0, nw ? s_all : s_mat, B, stream);
      if (e == hipSuccess) e = cnl::launch_cond_inertia(h->dc, d_vals, h->d_xpos, h->d_xzer, a.params[0], B, stream);
    }
    if (e != hipSuccess) return fail(CNL_ERR_HIP, std::string("condense: ") + hipGetErrorString(e));
    rc = cnl::dense_run_general(h->gdense, h->gops, a.mode, h->d_cbuf, h->d_xpos, h->d_xzer, h->d_d2,
                                d_vals ? d_vals + (C.nnz - C.nvar) : nullptr, C.nnz, a.rho_old, a.rho, a.nfact, a.success, a.npos, a.nzero,
                                a.params, stream, err);
    if (rc) return fail(CNL_ERR_HIP, "dense backend: " + err);
    if (a.mode == cnl::MODE_FACTOR) h->last_vals = d_vals;
    else {
      const double* vsrc = a.mode == cnl::MODE_SOLVE ? h->last_vals : d_vals;
      e = cnl::launch_expand(h->dc, const_cast<double*>(vsrc), d_rhs, h->d_d2, h->d_cbuf, d_d, a.mode == cnl::MODE_NEWTON ? a.success : nullptr, 0,
                             B, stream);
      if (e != hipSuccess) return fail(CNL_ERR_HIP, std::string("expand: ") + hipGetErrorString(e));
    }
    if (h->timing) {
      HIPCHK(hipEventRecord(h->ev1, stream));
      HIPCHK(hipEventSynchronize(h->ev1));
      HIPCHK(hipEventElapsedTime(&h->last_ms, h->ev0, h->ev1));
    }
    return CNL_OK;
  }
  if (!C.active) {
    a.vals = d_vals; a.rhs = d_rhs; a.d = d_d;
    rc = launch(h, a, stream);
  } else {
    const int B = (int)h->batch;
    const int s_mat = (int)(C.ncs + C.nvar), s_all = (int)C.cstride;
    double* crhs = h->d_cbuf ? h->d_cbuf + s_mat : nullptr;   // (band handles own no condensed buffer)
    hipError_t e = hipSuccess;
    const bool direct = h->use_v2 && h->plan->P.rec_direct;  // the register-front kernel condenses on the fly
    const bool count_d = direct && h->dp2.count_d;  // the kernel counts the condensed pivots itself
    if (direct && a.mode == cnl::MODE_NEWTON) {
      if (!count_d) {
        e = cnl::launch_cond_inertia(h->dc, d_vals, h->d_xpos, h->d_xzer, a.params[0], B, stream);
        if (e != hipSuccess) return fail(CNL_ERR_HIP, std::string("condense: ") + hipGetErrorString(e));
      }
      const bool d_outer = h->plan->P.d_outer;  // the kernel writes the kept components of d itself
      a.vals = d_vals; a.rhs = d_rhs; a.d = d_outer ? d_d : h->d_d2;
      a.extra_pos = count_d ? nullptr : h->d_xpos; a.extra_zer = count_d ? nullptr : h->d_xzer;
      if (h->staged) {
        // first attempt (rho as given) stage by stage: the tasks of the elimination tree run on different wavefronts; the
        // problems that fail it (rare) go through the whole ladder in the classic launch behind it
        if ((rc = launch_staged(h, a, stream))) return rc;
        if (h->ladder_ran) {
          // the problems that failed the attempt have climbed the rho ladder inside the fused launch(es) (kernels2.hip, phase 2);
          // the sequential launch behind them commits rho_old and the rho slots — or, if a wait gave up, redoes the whole call
          if ((rc = launch_redo(h, a, stream))) return rc;
        } else if (h->first_attempt_only) {
          // the host ladder follows; the sequential launch only if a dataflow wait of the attempt gave up (it then redoes the whole
          // call on the device, ladder included: the host finds the per-call status word set and leaves the results alone)
          if (h->d_dep && (rc = launch_redo(h, a, stream))) return rc;
        } else {
          a.skip_done = 1;
          const bool tm = h->timing;
          h->timing = false;
          rc = launch(h, a, stream);
          h->timing = tm;
          if (rc) return rc;
        }
        if (h->timing) HIPCHK(hipEventRecord(h->ev1, stream));
      } else if ((rc = launch(h, a, stream))) return rc;
      // (the lean instantiation has recovered the residual components in its backward sweep: plan.h, B_ROWS_FLAG)
      if (!(h->lean && h->plan->P.back_rows)) {
        e = cnl::launch_expand(h->dc, d_vals, d_rhs, d_outer ? nullptr : h->d_d2, h->d_cbuf, d_d, a.success, 0, B, stream);
        if (e != hipSuccess) return fail(CNL_ERR_HIP, std::string("expand: ") + hipGetErrorString(e));
      }
    } else if (direct && a.mode == cnl::MODE_FACTOR) {
      if (!count_d) {
        e = cnl::launch_cond_inertia(h->dc, d_vals, h->d_xpos, h->d_xzer, a.params[0], B, stream);
        if (e != hipSuccess) return fail(CNL_ERR_HIP, std::string("condense: ") + hipGetErrorString(e));
      }
      a.vals = d_vals; a.extra_pos = count_d ? nullptr : h->d_xpos; a.extra_zer = count_d ? nullptr : h->d_xzer;
      if (h->staged) {  // try_to_factorize stage by stage (the elimination tree's tasks on different wavefronts)
        if ((rc = launch_staged(h, a, stream))) return rc;
        if ((rc = launch_redo(h, a, stream))) return rc;
        if (h->timing) HIPCHK(hipEventRecord(h->ev1, stream));
      } else if ((rc = launch(h, a, stream))) return rc;
      h->last_vals = d_vals;
    } else if (a.mode == cnl::MODE_NEWTON) {
      e = C.tiled_ok ? cnl::launch_condense_tiled(h->dc, d_vals, d_rhs, h->d_cbuf, 7, C.ch_region[3], B, stream)
                     : cnl::launch_condense(h->dc, d_vals, d_rhs, h->d_cbuf, 0, s_all, B, stream);
      if (e == hipSuccess) e = cnl::launch_cond_inertia(h->dc, d_vals, h->d_xpos, h->d_xzer, a.params[0], B, stream);
      if (e != hipSuccess) return fail(CNL_ERR_HIP, std::string("condense: ") + hipGetErrorString(e));
      a.vals = h->d_cbuf; a.rhs = crhs; a.d = h->d_d2; a.extra_pos = h->d_xpos; a.extra_zer = h->d_xzer;
      if ((rc = launch(h, a, stream))) return rc;
      e = cnl::launch_expand(h->dc, d_vals, d_rhs, h->d_d2, h->d_cbuf, d_d, a.success, 1, B, stream);
      if (e != hipSuccess) return fail(CNL_ERR_HIP, std::string("expand: ") + hipGetErrorString(e));
    } else if (a.mode == cnl::MODE_FACTOR) {
      e = C.tiled_ok ? cnl::launch_condense_tiled(h->dc, d_vals, nullptr, h->d_cbuf, 3, C.ch_region[3], B, stream)
                     : cnl::launch_condense(h->dc, d_vals, nullptr, h->d_cbuf, 0, s_mat, B, stream);
      if (e == hipSuccess) e = cnl::launch_cond_inertia(h->dc, d_vals, h->d_xpos, h->d_xzer, a.params[0], B, stream);
      if (e != hipSuccess) return fail(CNL_ERR_HIP, std::string("condense: ") + hipGetErrorString(e));
      a.vals = h->d_cbuf; a.extra_pos = h->d_xpos; a.extra_zer = h->d_xzer;
      if ((rc = launch(h, a, stream))) return rc;
      h->last_vals = d_vals;
    } else if (direct && h->v2_solve) {
      // solve_ldl! on the register-front kernel: forward substitution with the stored factor, backward sweep, post-pass
      if (!h->last_vals) return fail(CNL_ERR_STATE, "cnl_solve before cnl_factorize");
      a.vals = const_cast<double*>(h->last_vals); a.rhs = d_rhs; a.d = d_d;
      if (h->staged) {  // solve_ldl! stage by stage: forward substitution of the tasks, then their backward sweeps
        if ((rc = launch_staged(h, a, stream))) return rc;
        if ((rc = launch_redo(h, a, stream))) return rc;
        if (h->timing) HIPCHK(hipEventRecord(h->ev1, stream));
      } else if ((rc = launch(h, a, stream))) return rc;
      // (lean plans: the solve-only instantiation has recovered the residual components in its backward sweep)
      if (!(h->lean && h->plan->P.back_rows)) {
        e = cnl::launch_expand(h->dc, const_cast<double*>(h->last_vals), d_rhs, nullptr, h->d_cbuf, d_d, nullptr, 0, B, stream);
        if (e != hipSuccess) return fail(CNL_ERR_HIP, std::string("expand: ") + hipGetErrorString(e));
      }
    } else {
      if (!h->last_vals) return fail(CNL_ERR_STATE, "cnl_solve before cnl_factorize");
      e = C.tiled_ok ? cnl::launch_condense_tiled(h->dc, h->last_vals, d_rhs, h->d_cbuf, 4, C.ch_region[3], B, stream)
                     : cnl::launch_condense(h->dc, h->last_vals, d_rhs, h->d_cbuf, s_mat, s_all, B, stream);
      if (e != hipSuccess) return fail(CNL_ERR_HIP, std::string("condense: ") + hipGetErrorString(e));
      a.rhs = crhs; a.d = h->d_d2;
      if ((rc = launch(h, a, stream))) return rc;
      e = cnl::launch_expand(h->dc, const_cast<double*>(h->last_vals), d_rhs, h->d_d2, h->d_cbuf, d_d, nullptr, 0, B, stream);
      if (e != hipSuccess) return fail(CNL_ERR_HIP, std::string("expand: ") + hipGetErrorString(e));
    }
  }
  if (rc) return rc;
  if (h->timing) {
    HIPCHK(hipEventSynchronize(h->ev1));
    HIPCHK(hipEventElapsedTime(&h->last_ms, h->ev0, h->ev1));
  }
  return CNL_OK;
}

int ensure_staging(cnl_handle* h) {
  if (h->d_vals) return CNL_OK;
  const cnl_plan& P = *h->plan;
  int rc;
  if ((rc = dalloc(h, &h->d_vals, (size_t)h->batch * P.nnz))) return rc;
  if ((rc = dalloc(h, &h->d_rhs, (size_t)h->batch * P.N))) return rc;
  if ((rc = dalloc(h, &h->d_d, (size_t)h->batch * P.N))) return rc;
  // the per-problem results of a call in ONE block, [rho | rho_old | nfact | success] and [npos | nzero]: a small host-pointer
  // call brings each group back with one copy (every copy of a few bytes is a transfer of its own on the stream: ~8 us)
  const size_t B = (size_t)h->batch;
  if ((rc = dalloc(h, &h->d_rho, 3 * B + 8))) return rc;
  h->d_rho_old = h->d_rho + B;
  h->d_nfact = reinterpret_cast<int32_t*>(h->d_rho_old + B);
  h->d_success = h->d_nfact + B;
  if ((rc = dalloc(h, &h->d_npos, B * 2 + 64))) return rc;   // (+ 64: the phase stamps of diagnostic builds land here)
  h->d_nzero = h->d_npos + B;
  return CNL_OK;
}

}  // namespace

struct cnl_multi {
  std::vector<cnl_handle*> h;
  std::vector<int64_t> start, count;
  std::vector<int> device;
  int64_t N = 0, nnz = 0, batch = 0;
  // one persistent host thread per shard (created with the handle, bound to the shard's device once): the host-pointer
  // calls hand each of them a job and wait; no thread is created or joined per call
  std::vector<std::thread> workers;
  std::mutex mu;
  std::condition_variable cv_job, cv_done;
  std::function<int(size_t)> job;
  uint64_t generation = 0;
  size_t pending = 0;
  bool stop = false;
  std::vector<int> rc;
  std::vector<std::string> msg;
};

namespace {
void multi_worker(cnl_multi* m, size_t i) {
  (void)hipSetDevice(m->device[i]);  // the thread's current device for its whole life
  uint64_t seen = 0;
  for (;;) {
    std::function<int(size_t)> f;
    {
      std::unique_lock<std::mutex> lk(m->mu);
      m->cv_job.wait(lk, [&] { return m->stop || m->generation != seen; });
      if (m->stop) return;
      seen = m->generation;
      f = m->job;
    }
    const int r = f(i);
    {
      std::lock_guard<std::mutex> lk(m->mu);
      m->rc[i] = r;
      if (r) m->msg[i] = g_err;  // thread-local in the worker: carry it over
      if (--m->pending == 0) m->cv_done.notify_all();
    }
  }
}

// runs f(i) for every shard on the shard's worker thread; the first failure (in shard order) becomes the caller's error
template <class F>
int multi_run(cnl_multi* m, F f) {
  const size_t n = m->h.size();
  {
    std::unique_lock<std::mutex> lk(m->mu);
    m->job = f;
    m->rc.assign(n, CNL_OK);
    m->msg.assign(n, std::string());
    m->pending = n;
    m->generation++;
    m->cv_job.notify_all();
    m->cv_done.wait(lk, [&] { return m->pending == 0; });
    m->job = nullptr;
  }
  for (size_t i = 0; i < n; i++)
    if (m->rc[i]) return fail(m->rc[i], "shard " + std::to_string(i) + " (device " + std::to_string(m->device[i]) + "): " + m->msg[i]);
  return CNL_OK;
}
}  // namespace


extern "C" {

static int create_from_plan(cnl_handle** hout, cnl_plan* plan, const int64_t* rows1, const int64_t* cols1, int64_t batch, int device);

const char* cnl_last_error(void) { return g_err.c_str(); }
// 0.2.0 (round 4: in-kernel device ladder, cnl_options grew).  An EXPERIMENT build (timing probes, diagnostic stamps: results may be
// wrong, see kernels2.hip) reports a NEGATIVE version; hipldl.py and the Julia glue refuse to load one unless asked to.
#ifdef CNL_EXPERIMENT
int32_t cnl_version(void) { return -200; }
#else
int32_t cnl_version(void) { return 200; }
#endif

void cnl_default_params(double p[9]) {
  const double eps = 2.220446049250313e-16;  // eps(Float64); src/CaNNOLeS.jl:48-62
  p[0] = eps;
  p[1] = std::sqrt(eps);
  p[2] = 1.0 / 3.0;
  p[3] = 8.0;
  p[4] = std::min(100.0, 8.0 * 16.0);
  p[5] = std::pow(eps, 1.0 / 3.0);  // the reference writes eps^T(1/3): pow with the exponent 0.333..., NOT cbrt (2.5 ulp apart)
  p[6] = std::pow(eps, -2.0);
  p[7] = std::sqrt(eps);
  p[8] = std::pow(eps, 0.25);
}

void cnl_options_init(cnl_options* o) {
  if (!o) return;
  std::memset(o, 0, sizeof(*o));
  const cnl::Tuning t;   // the defaults live in options.h
  o->struct_size = (int32_t)sizeof(cnl_options);
  o->plan_kind = t.plan_kind;
  // measured on MI355X (cfg3 pattern, tools/cmp_staged_threshold.py): latency plans with a few large canonical parts reach
  // 600 k systems/s at 2048 problems and 613 k at 4096 (the single stream: 342 k and 567 k); from 5120 on the single stream wins
  o->staged_max_batch = t.staged_max_batch;
  o->verbose = t.verbose; o->band_kernel = t.band_kernel; o->dense_backend = t.dense_backend; o->staged = t.staged; o->dataflow = t.dataflow;
  o->device_ladder = t.device_ladder; o->host_ladder = t.host_ladder; o->split_tail = t.split_tail; o->multi_share_plan = t.multi_share_plan;
  o->batch_layout = t.batch_layout;
}

static int plan_create_impl(cnl_plan** plan, int64_t N, int64_t nnz, const int64_t* rows1, const int64_t* cols1, int64_t nvar,
                            int64_t nequ, int64_t ncon, int latency, int par, double slots, const cnl::Tuning& o);
static int plan_create_tuned(cnl_plan** plan, int64_t N, int64_t nnz, const int64_t* rows1, const int64_t* cols1, int64_t nvar,
                             int64_t nequ, int64_t ncon, int64_t batch, const cnl::Tuning& o);
static int create_tuned(cnl_handle** hout, int64_t N, int64_t nnz, const int64_t* rows1, const int64_t* cols1, int64_t nvar,
                        int64_t nequ, int64_t ncon, int64_t batch, int device, const cnl::Tuning& o);

// public options -> the internal switch set: the defaults, the public fields, then the `tuning` pairs (which may name any switch of
// options.h, public ones included); rejects a struct of another ABI revision and unknown keys
static int resolve_options(const cnl_options* in, cnl::Tuning& out) {
  out = cnl::Tuning();
  if (!in) return CNL_OK;
  if (in->struct_size != (int32_t)sizeof(cnl_options)) return fail(CNL_ERR_ARG, "cnl_options.struct_size does not match this library (use cnl_options_init)");
  out.plan_kind = in->plan_kind; out.staged_max_batch = in->staged_max_batch; out.verbose = in->verbose; out.band_kernel = in->band_kernel;
  out.dense_backend = in->dense_backend; out.staged = in->staged; out.dataflow = in->dataflow; out.device_ladder = in->device_ladder;
  out.host_ladder = in->host_ladder; out.split_tail = in->split_tail; out.multi_share_plan = in->multi_share_plan; out.batch_layout = in->batch_layout;
  std::memcpy(out.force_order, in->force_order, sizeof(out.force_order));
  out.force_order[sizeof(out.force_order) - 1] = 0;
  char tun[sizeof(in->tuning) + 1];
  std::memcpy(tun, in->tuning, sizeof(in->tuning));
  tun[sizeof(in->tuning)] = 0;
  const std::string err = cnl::tuning_parse(out, tun);
  if (!err.empty()) return fail(CNL_ERR_ARG, err);
  return CNL_OK;
}

int cnl_plan_create_ex(cnl_plan** plan, int64_t N, int64_t nnz, const int64_t* rows1, const int64_t* cols1, int64_t nvar,
                       int64_t nequ, int64_t ncon, int64_t batch, const cnl_options* opt) {
  cnl::Tuning o;
  int rc = resolve_options(opt, o);
  if (rc) return rc;
  return plan_create_tuned(plan, N, nnz, rows1, cols1, nvar, nequ, ncon, batch, o);
}

static int plan_create_tuned(cnl_plan** plan, int64_t N, int64_t nnz, const int64_t* rows1, const int64_t* cols1, int64_t nvar,
                             int64_t nequ, int64_t ncon, int64_t batch, const cnl::Tuning& o) {
  int rc = CNL_OK;
  int latency = 0;
  if (o.plan_kind == CNL_PLAN_LATENCY) latency = 1;
  else if (o.plan_kind == CNL_PLAN_AUTO) latency = batch >= 1 && batch <= (o.staged_max_batch > 0 ? o.staged_max_batch : 4096);
  else if (o.plan_kind != CNL_PLAN_THROUGHPUT) return fail(CNL_ERR_ARG, "unknown plan_kind");
  if (!latency) {
    rc = plan_create_impl(plan, N, nnz, rows1, cols1, nvar, nequ, ncon, 0, 0, 0, o);
    // Between one and two wavefronts per SIMD the single stream leaves wavefront slots idle (686 k systems/s at 5120 problems of
    // cfg3's pattern between 958 k at 4096 and 964 k at 8192): when the bidirectional chain is available at the throughput
    // order's cost, the handle runs part of the batch on it and the rest single-stream, concurrently (run_split).
    const int64_t smb = o.staged_max_batch > 0 ? o.staged_max_batch : 4096;
    if (rc == CNL_OK && o.plan_kind == CNL_PLAN_AUTO && o.split_batch && batch > smb && batch <= 2 * smb - smb / 8 && o.force_order[0] == 0) {
      cnl::Tuning o2 = o;
      std::snprintf(o2.force_order, sizeof(o2.force_order), "ndc2+early");
      cnl_plan* alt = nullptr;
      const int nq = (int)((smb + 3) / 4);
      if (plan_create_impl(&alt, N, nnz, rows1, cols1, nvar, nequ, ncon, 1, std::max(1, 2048 / nq), 2048.0 / nq, o2) == CNL_OK) {
        const bool same_work = alt->latency && alt->P.order_name == "ndc2+early" && alt->P.tasks.size() >= 2 &&
                               alt->P.cost <= 1.05 * (*plan)->P.cost && alt->P.nsuper <= (*plan)->P.nsuper + 8;
        if (same_work) {
          alt->split_mode = true;
          std::memset(alt->opt.force_order, 0, sizeof(alt->opt.force_order));
          cnl_plan_destroy(*plan);
          *plan = alt;
        } else {
          cnl_plan_destroy(alt);
        }
      }
    }
    return rc;
  }
  if (batch < 1) return fail(CNL_ERR_ARG, "a latency plan needs the batch size");
  const int nquads = (int)((batch + 3) / 4);
  return plan_create_impl(plan, N, nnz, rows1, cols1, nvar, nequ, ncon, 1, std::max(1, 2048 / nquads), 2048.0 / nquads, o);
}

int cnl_plan_create(cnl_plan** plan, int64_t N, int64_t nnz, const int64_t* rows1, const int64_t* cols1, int64_t nvar,
                    int64_t nequ, int64_t ncon) {
  return cnl_plan_create_ex(plan, N, nnz, rows1, cols1, nvar, nequ, ncon, 0, nullptr);
}

int cnl_plan_create_for_batch(cnl_plan** plan, int64_t N, int64_t nnz, const int64_t* rows1, const int64_t* cols1, int64_t nvar,
                              int64_t nequ, int64_t ncon, int64_t batch) {
  if (batch < 1) return fail(CNL_ERR_ARG, "batch out of range");
  return cnl_plan_create_ex(plan, N, nnz, rows1, cols1, nvar, nequ, ncon, batch, nullptr);
}

// latency != 0: plan for a small batch — order chosen by the critical path, tree cut into tasks (par = wavefront slots per
// group of four problems)
static int plan_create_impl(cnl_plan** plan, int64_t N, int64_t nnz, const int64_t* rows1, const int64_t* cols1, int64_t nvar,
                            int64_t nequ, int64_t ncon, int latency, int par, double slots, const cnl::Tuning& o) {
  if (!plan || !rows1 || !cols1) return fail(CNL_ERR_ARG, "null argument");
  cnl_plan* p = new cnl_plan();
  p->N = N; p->nnz = nnz; p->nvar = nvar; p->nequ = nequ; p->ncon = ncon;
  p->latency = latency != 0;
  p->opt = o;
  const bool verbose = o.verbose != 0 || getenv("CNL_VERBOSE") != nullptr;  // logging only
  std::string msg;
  cnl::Options opt;
  opt.latency = latency; opt.par = std::max(1, par); opt.slots = slots;
  opt.order_mode = o.order_mode; opt.nd_leaf = o.nd_leaf; opt.relax = o.relax; opt.task_cap = o.task_cap;
  opt.early = o.multipliers_early; opt.register_front = o.register_front; opt.ubig = o.ubig; opt.wait_thr = o.wait_thr;
  opt.verbose = verbose ? 1 : 0; opt.force_order = o.force_order; opt.threads = o.analysis_threads;
  const auto t_start = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {   // (verbose log: seconds since the analysis started)
    if (verbose) fprintf(stderr, "[cnl] analysis %-28s %.3f s\n", what, std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count());
  };
  int rc = cnl::build_condensation(p->C, N, nnz, rows1, cols1, nvar, nequ, ncon, msg, o.condense != 0);
  lap("condensation");
  if (!rc) {
    if (p->C.active)
      rc = cnl::build_plan(p->P, p->C.N2, p->C.ncs + nvar, p->C.rows2.data(), p->C.cols2.data(), nvar, p->C.nequ2, ncon, opt, msg);
    else
      rc = cnl::build_plan(p->P, N, nnz, rows1, cols1, nvar, nequ, ncon, opt, msg);
  }
  lap("ordering + plan");
  if (rc) {
    delete p;
    *plan = nullptr;
    return fail(rc, msg);
  }
  // register-front kernel on a condensed system: rewrite the assembly lists against the ORIGINAL arrays, so
  // that the kernel condenses on the fly and the separate condense pass disappears (CNL_NO_DIRECT=1 keeps it)
  if (p->C.active && p->P.v2_ok && o.direct_records) {
    cnl::DirectLists D{p->C.c_ptr.data(), p->C.c_a.data(), p->C.c_b.data(), p->C.c_d.data(), (int32_t)nnz, (int32_t)N};
    const int32_t old_len = p->P.rec_maxlen;
    const size_t old_words = p->P.rec.size();
    p->P.row_products = o.row_products != 0;
    p->P.band_form = o.band_form != 0;
    // A plan whose fronts are ALL fast-class row-form fronts runs the kernels' lean instantiation and recovers the residual
    // components in its backward sweep (no post-pass): worth more than the few rounds small fronts save with product lists, so
    // every front is given the row form when that makes the whole plan lean.  A front whose residual rows do not fit the row form's
    // sixteen lanes is cut in two (same order, one more front) and the records are written again.
    // (round 6: such plans are written with the row form at once — the records with the lists' threshold of 72 products were
    //  written first and thrown away, one of five passes over the records of a latency plan)
    const bool fast_only = p->P.ncls[1] == 0 && p->P.ncls[2] == 0;
    const bool want_lean = o.row_products && o.lean_kernel && fast_only;
    p->P.row_min_products = want_lean ? 1 : 72;
    int drc = cnl::write_forward_records(p->P, &D);
    if (!drc && want_lean && p->P.listprod_fronts > 0) {
      if (!drc && p->P.listprod_fronts > 0 && !p->P.rows_overflow.empty() && p->P.rows_overflow.size() <= 64) {
        cnl::Options opt2 = opt;
        opt2.split_positions = p->P.rows_overflow;
        opt2.force_order = p->P.order_name;   // the same order, one more front: only that candidate is built again
        cnl::Plan P2;
        std::string msg2;
        if (cnl::build_plan(P2, p->C.N2, p->C.ncs + nvar, p->C.rows2.data(), p->C.cols2.data(), nvar, p->C.nequ2, ncon, opt2, msg2) == 0 &&
            P2.v2_ok && P2.ncls[1] == 0 && P2.ncls[2] == 0) {
          P2.row_products = true; P2.row_min_products = 1; P2.band_form = o.band_form != 0;
          if (cnl::write_forward_records(P2, &D) == 0 && P2.listprod_fronts == 0) {
            if (verbose) fprintf(stderr, "[cnl] %zu front(s) with more than 16 residual rows cut in two: %d -> %d fronts\n", p->P.rows_overflow.size(), p->P.nsuper, P2.nsuper);
            p->P = std::move(P2);
          }
        }
      }
      if (!drc && p->P.listprod_fronts > 0) {  // some front cannot take the row form: the lists' threshold again
        p->P.row_min_products = 72;
        drc = cnl::write_forward_records(p->P, &D);
      }
    }
    if (!drc) {
      // the backward records name the solution component of every pivot: switch them to the caller's numbering, so
      // that the kernel writes the kept components straight into `d` (no reduced solution vector, no copy pass)
      std::vector<int32_t>& br = p->P.brec;
      size_t r0 = 0;
      while (r0 + cnl::B_HDR <= br.size() && br[r0 + cnl::B_RECLEN] > 0) {
        const int32_t npiv = br[r0 + cnl::B_NPIV], nupd = br[r0 + cnl::B_NUPD];
        // a task root of a staged plan names the solution components of its update rows too
        const int32_t i0 = br[r0 + cnl::B_PXOFF] == cnl::B_PX_GLOBAL ? 1 : nupd + 1;
        for (int32_t i = i0; i < 1 + nupd + npiv; i++) br[r0 + cnl::B_HDR + i] = p->C.orig_of[br[r0 + cnl::B_HDR + i]];
        r0 += (size_t)br[r0 + cnl::B_RECLEN];
      }
      p->P.d_outer = true;
      cnl::finalize_tasks(p->P);  // the records moved
      // plans the lean kernel takes: the residual components of the rows a front owns are recovered in its backward step
      if (o.lean_kernel && o.rows_in_backward && p->P.ncls[1] == 0 && p->P.ncls[2] == 0 && p->P.listprod_fronts == 0) {
        const cnl::Cond& Cc = p->C;
        cnl::BackRowsIn in{Cc.r_orig.data(), Cc.r_dsrc.data(), Cc.r_ptr.data(), Cc.r_jsrc.data(), Cc.r_jx.data(), (int32_t)Cc.r_orig.size()};
        const int brc = cnl::write_backward_rows(p->P, in);
        if (verbose) fprintf(stderr, "[cnl] backward rows: %s\n", brc ? "not possible" : "ok");
      }
    } else {
      p->P.tasks.clear();  // staged execution needs the direct records
    }
    if (verbose)
      fprintf(stderr, "[cnl] direct records: %s, rec words %zu -> %zu, longest %d -> %d\n", drc ? "not possible" : "ok", old_words,
              p->P.rec.size(), old_len, p->P.rec_maxlen);
  }
  // A latency order that cannot run staged (setup_v2's conditions: tasks, direct records, solution components in the caller's
  // numbering, every condensed pivot counted by a front) would run on the single sequential stream, where it is only a worse
  // order — more total work, chosen for a critical path nothing exploits: take the throughput analysis instead.
  if (latency && p->opt.plan_kind != CNL_PLAN_LATENCY) {
    const bool stageable = o.staged && !p->P.tasks.empty() && p->P.v2_ok && p->P.rec_direct && p->P.d_outer &&
                           p->P.d_owned == (int64_t)p->C.r_dsrc.size();
    if (!stageable) {
      if (verbose) fprintf(stderr, "[cnl] latency plan cannot be staged: falling back to the throughput analysis\n");
      delete p;
      return plan_create_impl(plan, N, nnz, rows1, cols1, nvar, nequ, ncon, 0, 0, 0, o);
    }
  }
  if (o.dense_backend) cnl::detect_dense(p->D, N, nnz, rows1, cols1, nvar, nequ, ncon);
  lap("records");
  // (round 5) large batches of band-structured problems: the sliding-window elimination with one lane per (problem, part)
  if (!latency && o.band_kernel && p->C.active && !p->D.active) {
    cnl::build_band_plan(p->band, N, nnz, rows1, cols1, nvar, nequ, ncon, o.band_kernel == 2 ? 1 : 2);
    if (verbose) fprintf(stderr, "[cnl] band program: %s%s\n", p->band.ok ? "ok" : "no: ", p->band.ok ? "" : p->band.why.c_str());
  }
  {
    const cnl::BandPlan& Bp = p->band;
    p->band_info = {Bp.ok ? 1 : 0, Bp.nparts, Bp.m0, Bp.n, Bp.N, Bp.nnz, (int32_t)Bp.lsize, 0};
    for (int q = 0; q < 2; q++) p->band_pinfo[q] = {Bp.part[q].nsteps, Bp.part[q].nepochs, Bp.part[q].npiv, Bp.part[q].nevents, (int32_t)Bp.part[q].loff};
  }
  // Irregular sparsity: when the fill makes fronts larger than the register-front kernel takes and the condensed system is of
  // moderate order, one dense LDL^T of the whole condensed matrix beats the general multifrontal kernel by far
  // (csrc/dense.h; chosen at handle creation for small batches; CNL_NO_GDENSE=1 disables)
  // (round 5) ... and so does a latency plan whose fronts reach the 64 class while the whole condensed system is of order <= 512: the
  // dense route costs 0.075 ms + 0.28 us per unit of order for one system (tools/time_dense_route.py), the staged walk over fronts of
  // that size 0.2 ms and more (n = 133, fronts up to 59: 0.215 against 0.106 ms).  The handle takes it for batches up to 16.
  p->prefer_dense = latency && p->P.v2_ok && p->P.ncls[2] > 0 && p->C.N2 <= 512;
  if (p->C.active && !p->D.active && (!p->P.v2_ok || p->prefer_dense || o.general_dense == 2) && p->C.N2 >= 96 && p->C.N2 <= 4096 && o.general_dense) {
    p->gpos.resize(p->C.ncs);
    for (int64_t s2 = 0; s2 < p->C.ncs; s2++) p->gpos[s2] = (int32_t)((p->C.rows2[s2] - 1) + p->C.N2 * (p->C.cols2[s2] - 1));
  }
  // elimination order in the reference's numbering: condensed residual nodes first
  if (p->C.active) {
    p->perm_outer.assign(p->C.r_orig.begin(), p->C.r_orig.end());
    for (int32_t e : p->P.perm) p->perm_outer.push_back(p->C.orig_of[e]);
    std::vector<int64_t>().swap(p->C.rows2);
    std::vector<int64_t>().swap(p->C.cols2);
  } else {
    p->perm_outer = p->P.perm;
  }
  lap("band program + done");
  *plan = p;
  return CNL_OK;
}

void cnl_plan_destroy(cnl_plan* plan) {
  if (plan && plan->refs.fetch_sub(1) == 1) delete plan;
}

int cnl_plan_info(const cnl_plan* plan, int64_t info[16]) {
  if (!plan || !info) return fail(CNL_ERR_ARG, "null argument");
  const cnl::Plan& P = plan->P;
  std::memset(info, 0, 16 * sizeof(int64_t));
  info[0] = plan->N; info[1] = plan->nnz; info[2] = P.nnzK; info[3] = P.nsuper; info[4] = P.nnzL; info[5] = P.nnzL_exact;
  if (plan->C.active) {  // the L rows of the condensed residual pivots (J_r / d_r) belong to the factor too
    info[4] += (int64_t)plan->C.r_jsrc.size();
    info[5] += (int64_t)plan->C.r_jsrc.size();
  }
  info[6] = P.lsize; info[7] = P.fmax; info[8] = P.fwd_peak; info[9] = P.bwd_peak; info[10] = P.panel_max;
  info[11] = (int64_t)P.flops; info[12] = (int64_t)P.asm_src.size();
  info[13] = P.v2_ok ? ((int64_t)P.ncls[0] | ((int64_t)P.ncls[1] << 20) | ((int64_t)P.ncls[2] << 40)) : -1;
  info[14] = P.v2_ok ? ((int64_t)P.u2_peak | ((int64_t)P.fs2_max << 20) | ((int64_t)std::max(P.rec_maxlen, P.brec_maxlen) << 40)) : -1;
  info[15] = plan->C.active ? (int64_t)plan->C.r_orig.size() : 0;  // condensed residual nodes
  return CNL_OK;
}

const char* cnl_plan_order_name(const cnl_plan* plan) { return plan ? plan->P.order_name.c_str() : ""; }

int cnl_plan_get(const cnl_plan* plan, const char* name, int32_t* out, int64_t* count) {
  if (!plan || !name || !count) return fail(CNL_ERR_ARG, "null argument");
  const cnl::Plan& P = plan->P;
  const int32_t* src = nullptr;
  int64_t n = 0;
  std::string s(name);
  const cnl::Cond& C = plan->C;
  if (s == "perm") { src = plan->perm_outer.data(); n = (int64_t)plan->perm_outer.size(); }
  else if (s == "inner_perm") { src = P.perm.data(); n = (int64_t)P.perm.size(); }
  else if (s == "c_ptr") { src = C.c_ptr.data(); n = (int64_t)C.c_ptr.size(); }
  else if (s == "c_a") { src = C.c_a.data(); n = (int64_t)C.c_a.size(); }
  else if (s == "c_b") { src = C.c_b.data(); n = (int64_t)C.c_b.size(); }
  else if (s == "c_d") { src = C.c_d.data(); n = (int64_t)C.c_d.size(); }
  else if (s == "orig_of") { src = C.orig_of.data(); n = (int64_t)C.orig_of.size(); }
  else if (s == "r_orig") { src = C.r_orig.data(); n = (int64_t)C.r_orig.size(); }
  else if (s == "r_dsrc") { src = C.r_dsrc.data(); n = (int64_t)C.r_dsrc.size(); }
  else if (s == "r_ptr") { src = C.r_ptr.data(); n = (int64_t)C.r_ptr.size(); }
  else if (s == "r_jsrc") { src = C.r_jsrc.data(); n = (int64_t)C.r_jsrc.size(); }
  else if (s == "r_jx") { src = C.r_jx.data(); n = (int64_t)C.r_jx.size(); }
  else if (s == "fronts") { src = reinterpret_cast<const int32_t*>(P.fronts.data()); n = (int64_t)P.fronts.size() * 16; }
  else if (s == "seg_ptr") { src = P.seg_ptr.data(); n = (int64_t)P.seg_ptr.size(); }
  else if (s == "asm_pos") { src = P.asm_pos.data(); n = (int64_t)P.asm_pos.size(); }
  else if (s == "asm_src") { src = P.asm_src.data(); n = (int64_t)P.asm_src.size(); }
  else if (s == "child_idx") { src = P.child_idx.data(); n = (int64_t)P.child_idx.size(); }
  else if (s == "rel_idx") { src = P.rel_idx.data(); n = (int64_t)P.rel_idx.size(); }
  else if (s == "tasks") { src = reinterpret_cast<const int32_t*>(P.tasks.data()); n = (int64_t)P.tasks.size() * 8; }  // struct Task, csrc/plan.h
  else if (s == "stage_ptr") { src = P.stage_ptr.data(); n = (int64_t)P.stage_ptr.size(); }
  else if (s == "rec") { src = P.rec.data(); n = P.v2_ok ? (int64_t)P.rec.size() : 0; }     // record streams of the
  else if (s == "brec") { src = P.brec.data(); n = P.v2_ok ? (int64_t)P.brec.size() : 0; }  // register-front kernel
  else if (s == "band_info") { src = plan->band_info.data(); n = (int64_t)plan->band_info.size(); }   // band program (csrc/band.h)
  else if (s.rfind("band_", 0) == 0 && s.size() >= 6 && (s.back() == '0' || s.back() == '1')) {
    const int q = s.back() - '0';
    const cnl::BandPart& Q = plan->band.part[q];
    const std::string k = s.substr(5, s.size() - 6);
    if (k == "part") { src = plan->band_pinfo[q].data(); n = (int64_t)plan->band_pinfo[q].size(); }
    else if (k == "fops") { src = Q.fops.data(); n = (int64_t)Q.fops.size(); }
    else if (k == "bops") { src = Q.bops.data(); n = (int64_t)Q.bops.size(); }
    else if (k == "epochs") { src = Q.epochs.data(); n = (int64_t)Q.epochs.size(); }
    else if (k == "borders") { src = Q.borders.data(); n = (int64_t)Q.borders.size(); }
    else return fail(CNL_ERR_ARG, "unknown plan array: " + s);
  }
  else return fail(CNL_ERR_ARG, "unknown plan array: " + s);
  if (out) {
    if (*count < n) return fail(CNL_ERR_ARG, "buffer too small");
    std::memcpy(out, src, (size_t)n * sizeof(int32_t));
  }
  *count = n;
  return CNL_OK;
}

int cnl_create(cnl_handle** hout, int64_t N, int64_t nnz, const int64_t* rows1, const int64_t* cols1, int64_t nvar,
               int64_t nequ, int64_t ncon, int64_t batch, int device) {
  return cnl_create_ex(hout, N, nnz, rows1, cols1, nvar, nequ, ncon, batch, device, nullptr);
}

int cnl_create_ex(cnl_handle** hout, int64_t N, int64_t nnz, const int64_t* rows1, const int64_t* cols1, int64_t nvar,
                  int64_t nequ, int64_t ncon, int64_t batch, int device, const cnl_options* opt) {
  if (!hout) return fail(CNL_ERR_ARG, "null handle pointer");
  *hout = nullptr;
  cnl::Tuning o;
  const int rc0 = resolve_options(opt, o);
  if (rc0) return rc0;
  return create_tuned(hout, N, nnz, rows1, cols1, nvar, nequ, ncon, batch, device, o);
}

static int create_tuned(cnl_handle** hout, int64_t N, int64_t nnz, const int64_t* rows1, const int64_t* cols1, int64_t nvar,
                        int64_t nequ, int64_t ncon, int64_t batch, int device, const cnl::Tuning& o) {
  *hout = nullptr;
  if (batch < 1 || batch > (1 << 24)) return fail(CNL_ERR_ARG, "batch out of range");
  int ndev = 0;
  const hipError_t ce = hipGetDeviceCount(&ndev);
  if (ce != hipSuccess || ndev == 0)
    return fail(CNL_ERR_HIP, std::string("no HIP device available (this backend has no CPU fallback): hipGetDeviceCount -> ") +
                                 hipGetErrorString(ce) + ", " + std::to_string(ndev) + " device(s)");
  if (device < 0 || device >= ndev) return fail(CNL_ERR_ARG, "device index out of range");
  cnl_plan* plan = nullptr;
  // small batches cannot fill the chip with one wavefront per four problems: plan for latency (bushy order, tasks)
  int rc = plan_create_tuned(&plan, N, nnz, rows1, cols1, nvar, nequ, ncon, batch, o);
  if (rc) return rc;
  return create_from_plan(hout, plan, rows1, cols1, batch, device);
}

// device state for `batch` problems of an analysed pattern; takes ownership of `plan` (freed with the handle, or here on failure)
static int create_from_plan(cnl_handle** hout, cnl_plan* plan, const int64_t* rows1, const int64_t* cols1, int64_t batch, int device) {
  const int64_t N = plan->N, nnz = plan->nnz, nvar = plan->nvar, nequ = plan->nequ, ncon = plan->ncon;
  int rc = CNL_OK;
  cnl_handle* h = new cnl_handle();
  h->plan = plan; h->device = device; h->batch = batch;
  auto bail = [&](int code) { cnl_destroy(h); return code; };
  if (hipSetDevice(device) != hipSuccess) return bail(fail(CNL_ERR_HIP, "hipSetDevice failed"));
  const cnl::Plan& P = plan->P;
  cnl::DevPlan& dp = h->dp;
  if ((rc = upload(h, P.fronts, &dp.fronts))) return bail(rc);
  if ((rc = upload(h, P.seg_ptr, &dp.seg_ptr))) return bail(rc);
  if ((rc = upload(h, P.asm_pos, &dp.asm_pos))) return bail(rc);
  if ((rc = upload(h, P.asm_src, &dp.asm_src))) return bail(rc);
  if ((rc = upload(h, P.child_idx, &dp.child_idx))) return bail(rc);
  if ((rc = upload(h, P.rel_idx, &dp.rel_idx))) return bail(rc);
  if ((rc = upload(h, P.perm, &dp.perm))) return bail(rc);
  dp.nsuper = P.nsuper; dp.N = (int32_t)P.N; dp.nnz = (int32_t)P.nnz; dp.rho_begin = P.rho_begin;
  dp.nvar = (int32_t)P.nvar; dp.nequ = (int32_t)P.nequ; dp.ncon = (int32_t)P.ncon;
  dp.fmax = (P.fmax + 1) & ~1;
  // factor storage stride per problem: + 16 zero doubles that are never written.  The solve sweeps read a panel row as 16
  // lanes, so the last rows of a problem's factor are over-read by up to 15 entries, which are multiplied by zeros; without
  // the pad they would be the first entries of the NEXT problem's factor, and a NaN / Inf there (a neighbour whose
  // factorisation broke down) would turn 0 * x into NaN in this problem's solution
  dp.lsize = P.lsize + 16;
  const cnl::Cond& C = plan->C;
  dp.vstride = C.active ? C.cstride : (int64_t)nnz;
  dp.rstride = C.active ? C.cstride : N;
  dp.dstride = C.active ? C.N2 : N;
  if (C.active) {
    cnl::DevCond& dc = h->dc;
    std::vector<int32_t> cidx(N, -1);
    for (size_t q = 0; q < C.r_orig.size(); q++) cidx[C.r_orig[q]] = (int32_t)q;
    if ((rc = upload(h, C.c_ptr, &dc.c_ptr))) return bail(rc);
    if ((rc = upload(h, C.c_a, &dc.c_a))) return bail(rc);
    if ((rc = upload(h, C.c_b, &dc.c_b))) return bail(rc);
    if ((rc = upload(h, C.c_d, &dc.c_d))) return bail(rc);
    if ((rc = upload(h, C.c_order, &dc.c_order))) return bail(rc);
    if ((rc = upload(h, C.ch_slot, &dc.ch_slot))) return bail(rc);
    if ((rc = upload(h, C.ch_rng, &dc.ch_rng))) return bail(rc);
    if ((rc = upload(h, C.ch_tile, &dc.ch_tile))) return bail(rc);
    if ((rc = upload(h, C.rng_start, &dc.rng_start))) return bail(rc);
    if ((rc = upload(h, C.rng_len, &dc.rng_len))) return bail(rc);
    if ((rc = upload(h, C.c_la, &dc.c_la))) return bail(rc);
    if ((rc = upload(h, C.c_lb, &dc.c_lb))) return bail(rc);
    if ((rc = upload(h, C.c_ld, &dc.c_ld))) return bail(rc);
    if ((rc = upload(h, C.ch_tptr, &dc.ch_tptr))) return bail(rc);
    if ((rc = upload(h, C.tile_src, &dc.tile_src))) return bail(rc);
    if ((rc = upload(h, C.c_pack, &dc.c_pack))) return bail(rc);
    dc.tile_max = C.tile_max; dc.chunk_ncon_max = C.chunk_ncon_max; dc.chunk_nslot_max = C.chunk_nslot_max; dc.tiled_ok = C.tiled_ok ? 1 : 0;
    if ((rc = upload(h, C.r_dsrc, &dc.r_dsrc))) return bail(rc);
    if ((rc = upload(h, C.r_ptr, &dc.r_ptr))) return bail(rc);
    if ((rc = upload(h, C.r_jsrc, &dc.r_jsrc))) return bail(rc);
    if ((rc = upload(h, C.r_jx, &dc.r_jx))) return bail(rc);
    if ((rc = upload(h, C.red_of, &dc.red_of))) return bail(rc);
    if ((rc = upload(h, cidx, &dc.cidx_of))) return bail(rc);
    if ((rc = upload(h, C.orig_of, &dc.orig_of))) return bail(rc);
    if ((rc = upload(h, C.r_orig, &dc.r_orig))) return bail(rc);
    dc.N = (int32_t)N; dc.nnz = (int32_t)nnz; dc.nvar = (int32_t)nvar; dc.N2 = (int32_t)C.N2; dc.ncs = (int32_t)C.ncs;
    dc.ncond = (int32_t)C.r_orig.size(); dc.cstride = C.cstride;
    // (d_cbuf / d_d2 — the condensed buffer and the reduced solution of the stand-alone condensation passes — are allocated below,
    //  once it is known whether the band kernels serve the handle: they never touch them)
    if ((rc = dalloc(h, &h->d_xpos, (size_t)batch))) return bail(rc);
    if ((rc = dalloc(h, &h->d_xzer, (size_t)batch))) return bail(rc);
  }
  if ((rc = choose_config(h))) return bail(rc);
  if ((rc = setup_v2(h))) return bail(rc);
  if (plan->band.ok && plan->opt.band_kernel && h->use_v2 && !h->staged && h->v2_solve && h->lean && plan->P.back_rows && !plan->latency && !plan->split_mode) {
    // band kernels for newton_system (csrc/band.h); the register-front kernel keeps try_to_factorize / solve_ldl!
    const cnl::BandPlan& Bp = plan->band;
    cnl::BandDev& bd = h->bd;
    for (int q = 0; q < Bp.nparts; q++) {
      if ((rc = upload(h, Bp.part[q].fops, &bd.fops[q]))) return bail(rc);
      if ((rc = upload(h, Bp.part[q].bops, &bd.bops[q]))) return bail(rc);
      if ((rc = upload(h, Bp.part[q].epochs, &bd.epochs[q]))) return bail(rc);
      if ((rc = upload(h, Bp.part[q].borders, &bd.borders[q]))) return bail(rc);
      bd.nsteps[q] = Bp.part[q].nsteps; bd.nepochs[q] = Bp.part[q].nepochs; bd.loff[q] = Bp.part[q].loff;
    }
    bd.nparts = Bp.nparts; bd.m0 = Bp.m0; bd.n = Bp.n; bd.N = Bp.N; bd.nnz = Bp.nnz; bd.nvar = (int32_t)nvar; bd.lsize = Bp.lsize;
    // 16 problems per workgroup (two workgroups = four wavefronts per CU: one per SIMD) up to the 8192 problems that fills; above,
    // 32 per workgroup (the LDS of a CU holds two such workgroups: 16384 problems resident) — tools/time_band.py
    h->band_nl = plan->opt.band_problems_per_group > 0 ? plan->opt.band_problems_per_group : (batch > 8192 ? 32 : 16);
    if (h->band_nl != 8 && h->band_nl != 16 && h->band_nl != 32) return bail(fail(CNL_ERR_ARG, "band_problems_per_group must be 8, 16 or 32"));
    if (plan->opt.band_movers > 0 && cnl::band_mw_group(plan->opt.band_movers) == 0)
      return bail(fail(CNL_ERR_ARG, "tuning key band_movers (1 .. 3) needs a library built with -DCNL_EXPERIMENT=1 -DBAND_MW (csrc/band.hip)"));
    if (plan->opt.band_movers > 0 && bd.nparts == 2 && cnl::band_mw_lds_bytes(plan->opt.band_movers) <= std::min<size_t>(cnl::max_lds_bytes(), 160 * 1024)) {
      h->band_mw = plan->opt.band_movers;
      h->band_nl = cnl::band_mw_group(h->band_mw);
    }
    // 32-bit byte offsets inside a workgroup's problems
    const uint64_t span = 8ull * (uint64_t)h->band_nl * (uint64_t)std::max<int64_t>({(int64_t)nnz, N, bd.lsize});
    if (span < (1ull << 32) && cnl::band_lds_bytes(bd.nparts, h->band_nl) <= std::min<size_t>(cnl::max_lds_bytes(), 160 * 1024)) {
      // (+ 32 problems: the band kernels interleave the records of a workgroup's problems, the last workgroup's region is a whole one)
      if ((rc = dalloc(h, &h->d_Lband, ((size_t)batch + 32) * (size_t)bd.lsize + 64))) return bail(rc);
      if (hipMemset(h->d_Lband, 0, (((size_t)batch + 32) * (size_t)bd.lsize + 64) * sizeof(double)) != hipSuccess) return bail(fail(CNL_ERR_HIP, "hipMemset failed"));
      h->band = true;
    }
  }
  if (plan->opt.batch_layout != CNL_LAYOUT_PROBLEM_MAJOR) {
    // the interleaved layout is the band kernels' (groups of 32 problems = one workgroup of the 32-problem instantiation)
    if (plan->opt.batch_layout != CNL_LAYOUT_INTERLEAVED) return bail(fail(CNL_ERR_ARG, "cnl_options.batch_layout: unknown layout"));
    if (!h->band)
      return bail(fail(CNL_ERR_ARG, "batch_layout = CNL_LAYOUT_INTERLEAVED needs a handle the band kernels serve "
                                    "(band-structured pattern, throughput plan, cnl_options.band_kernel != 0; csrc/band.h)"));
    h->layout = 1 | (plan->opt.band_rhs_interleaved ? 2 : 0);
  }
  // Storage only the register-front / general kernels and the stand-alone condensation passes use.  A band handle runs all three
  // calls of the plugin surface on the band kernels (round 6), so it owns the band factor records alone: 0.48 MB per problem of
  // cfg3's size instead of 0.48 + 1.16 (factor panels) + 0.64 (condensed buffer, reduced solution) — 16 384 problems: 29 GB less,
  // and twice the batch fits the 288 GB of a device beside the caller's arrays.
  if (plan->C.active && !h->band) {
    if ((rc = dalloc(h, &h->d_cbuf, (size_t)batch * (size_t)plan->C.cstride))) return bail(rc);
    if ((rc = dalloc(h, &h->d_d2, (size_t)batch * (size_t)plan->C.N2))) return bail(rc);
  }
  if (plan->split_mode && h->staged) {
    // x groups of four problems on the chain (two wavefronts each), the rest on the single stream: 2 x + y = 2048 slots
    const int64_t nquads = (batch + 3) / 4, x = std::max<int64_t>(0, 2048 - nquads);
    h->split_staged = std::min<int64_t>(batch, 4 * x);
    if (h->plan->opt.split_batch == 1 && batch <= 6400) {
      // two halves on the chain (multiples of four problems), sequentially — measured against chain + single stream concurrently
      // (cfg3's size, same box, k systems/s): 4608: 768 / 707, 5120: 836 / 790, 6144: 931 / 898, 6656: 912 / ~935, 7168: 956 / 959,
      // 7424: 904 / ~965 — halves up to 6400 problems (see run_split)
      h->split_halves = true;
      h->split_staged = (((batch + 1) / 2) + 3) & ~(int64_t)3;
    }
    if (h->split_staged == 0) h->staged = false;  // the whole batch on the single stream
    // a remainder of at most a quarter of the machine-filling batch: its own handle with its own (many-part) plan — run_split
    const int64_t smb = ((plan->opt.staged_max_batch > 0 ? plan->opt.staged_max_batch : 4096)) & ~(int64_t)3;
    if (h->staged && plan->opt.split_batch == 1 && plan->opt.split_tail != 0 && batch > smb && batch - smb <= smb / 4) {
      cnl_handle* t = nullptr;
      if (create_tuned(&t, N, nnz, rows1, cols1, nvar, nequ, ncon, batch - smb, device, plan->opt) == CNL_OK) {
        if (t->staged && t->use_v2 && !t->dense && !t->gdense && !t->tail && t->split_staged == 0) {
          h->tail = t; h->split_halves = false; h->split_staged = smb;
        } else {
          cnl_destroy(t);
        }
      }
      if (hipSetDevice(device) != hipSuccess) return bail(fail(CNL_ERR_HIP, "hipSetDevice failed"));
    }
  }
  if (!plan->latency && !plan->split_mode && h->use_v2 && !h->staged && h->resident_waves > 0 && plan->opt.plan_kind == CNL_PLAN_AUTO &&
      plan->opt.split_tail != 0 && plan->opt.force_order[0] == 0 && !plan->D.active && plan->gpos.empty()) {
    // The single stream gives a group of four problems ONE wavefront for all fronts, and the device holds `resident_waves` of
    // them: a batch of k full machine loads + r problems runs k + 1 rounds, the last one for the r problems alone (cfg3's size:
    // 8448 problems 13.0 ms against 7.9 ms for 8192).  The remainder as a batch of its own has a better plan (many parts, the
    // bidirectional chain, ...): it gets a handle of its own, enqueued behind the full loads (run_split).
    // (band kernels: 512 workgroups of 32 problems are resident at once)
    const int64_t cap = h->band ? 16384 : 4 * (int64_t)h->resident_waves, r = batch % cap;
    if (batch > cap && r > 0 && r <= cap - cap / 16 && !h->layout) {   // (an interleaved batch is one array of whole groups)
      cnl_handle* t = nullptr;
      if (create_tuned(&t, N, nnz, rows1, cols1, nvar, nequ, ncon, r, device, plan->opt) == CNL_OK) {
        if (t->staged && t->use_v2 && !t->dense && !t->gdense) { h->tail = t; h->split_staged = batch - r; }
        else cnl_destroy(t);
      }
      if (hipSetDevice(device) != hipSuccess) return bail(fail(CNL_ERR_HIP, "hipSetDevice failed"));
    }
  }
  if (h->plan->D.active) {
    std::string derr;
    int drc = cnl::dense_create(&h->dense, h->plan->D, batch, derr, h->plan->opt.dense_graph != 0, h->plan->opt.dense_syrk_wgs, h->plan->opt.dense_panel_blocks);
    if (drc) return bail(fail(CNL_ERR_HIP, "dense backend: " + derr));
  } else if (!h->plan->gpos.empty() && !h->use_v2 &&
             // S0, S and G in 64 x 64 tiles per problem: small batches always, larger ones while the tiles stay below 8 GB
             // (round 2 stopped at 16 problems; batches of small irregular systems then fell to the general kernel)
             (batch <= 16 || (double)batch * 3.0 * 32768.0 * std::pow(std::ceil((double)h->plan->C.N2 / 64.0), 2) <= 8e9)) {
    std::string derr;
    const cnl::Cond& C2 = h->plan->C;
    h->gops.ns = (int32_t)C2.N2; h->gops.nv = (int32_t)nvar; h->gops.nslots = (int32_t)C2.ncs; h->gops.cstride = C2.cstride;
    if ((rc = upload(h, h->plan->gpos, &h->gops.d_pos))) return bail(rc);
    int drc = cnl::dense_create_general(&h->gdense, (int32_t)C2.N2, (int32_t)nvar, (int32_t)C2.ncs, h->gops.d_pos, batch, derr, h->plan->opt.dense_graph != 0, h->plan->opt.dense_panel_blocks);
    if (drc) return bail(fail(CNL_ERR_HIP, "dense backend: " + derr));
  }
  {
    // factor storage, zero-filled and padded: the row prefetch of the backward pass reads (never uses) a little past a panel
    const size_t ldoubles = (h->band ? 0 : (size_t)batch * (size_t)dp.lsize) + 4096;   // (band handles: see above)
    if ((rc = dalloc(h, &h->d_L, ldoubles))) return bail(rc);
    if (hipMemset(h->d_L, 0, ldoubles * sizeof(double)) != hipSuccess) return bail(fail(CNL_ERR_HIP, "hipMemset failed"));
  }
  if (!h->cfg.lds_work)
    if ((rc = dalloc(h, &h->d_scratch, (size_t)batch * (size_t)dp.work_doubles))) return bail(rc);
  if (hipStreamCreate(&h->stream) != hipSuccess) return bail(fail(CNL_ERR_HIP, "hipStreamCreate failed"));
  if (hipEventCreate(&h->ev0) != hipSuccess || hipEventCreate(&h->ev1) != hipSuccess)
    return bail(fail(CNL_ERR_HIP, "hipEventCreate failed"));
  {
    // transposed-Jacobian lists from the pattern: entries with column <= nvar < row, in COO order per column
    std::vector<int32_t> ptrF(nvar + 1, 0), ptrC(nvar + 1, 0), slotF, idxF, slotC, idxC;
    for (int64_t e = 0; e < nnz; e++) {
      const int64_t r0 = rows1[e] - 1, c0 = cols1[e] - 1;
      if (c0 < nvar && r0 >= nvar) (r0 < nvar + nequ ? ptrF : ptrC)[c0 + 1]++;
    }
    for (int64_t j2 = 0; j2 < nvar; j2++) { ptrF[j2 + 1] += ptrF[j2]; ptrC[j2 + 1] += ptrC[j2]; }
    slotF.resize(ptrF[nvar]); idxF.resize(ptrF[nvar]); slotC.resize(ptrC[nvar]); idxC.resize(ptrC[nvar]);
    std::vector<int32_t> fillF(ptrF.begin(), ptrF.end() - 1), fillC(ptrC.begin(), ptrC.end() - 1);
    for (int64_t e = 0; e < nnz; e++) {
      const int64_t r0 = rows1[e] - 1, c0 = cols1[e] - 1;
      if (!(c0 < nvar && r0 >= nvar)) continue;
      if (r0 < nvar + nequ) { const int32_t q = fillF[c0]++; slotF[q] = (int32_t)e; idxF[q] = (int32_t)(r0 - nvar); }
      else { const int32_t q = fillC[c0]++; slotC[q] = (int32_t)e; idxC[q] = (int32_t)(r0 - nvar - nequ); }
    }
    cnl::DevJt& J = h->djt;
    if ((rc = upload(h, ptrF, &J.ptrF))) return bail(rc);
    // the J_F / J_c entries as segments of `vals` (the reference's 7-segment layout, src/CaNNOLeS.jl:256-315): rows f1 / f4 can
    // read them from the model's arrays instead (cnl_residual_vectors_jac_dev) when each kind occupies one run of slots
    {
      auto run = [](const std::vector<int32_t>& sl, int64_t& lo) {
        if (sl.empty()) { lo = 0; return true; }
        const auto mm = std::minmax_element(sl.begin(), sl.end());
        lo = *mm.first;
        return (int64_t)*mm.second - *mm.first + 1 == (int64_t)sl.size();
      };
      h->jac_segments = run(slotF, h->jf_lo) && run(slotC, h->jc_lo);
      h->jf_n = (int64_t)slotF.size(); h->jc_n = (int64_t)slotC.size();
    }
    if ((rc = upload(h, slotF, &J.slotF))) return bail(rc);
    if ((rc = upload(h, idxF, &J.idxF))) return bail(rc);
    if ((rc = upload(h, ptrC, &J.ptrC))) return bail(rc);
    if ((rc = upload(h, slotC, &J.slotC))) return bail(rc);
    if ((rc = upload(h, idxC, &J.idxC))) return bail(rc);
    {
      // J_c by rows (CGLS, row f4)
      std::vector<int32_t> rptr(ncon + 1, 0), rslot(slotC.size()), rcol(slotC.size());
      for (int64_t j2 = 0; j2 < nvar; j2++) for (int32_t q = ptrC[j2]; q < ptrC[j2 + 1]; q++) rptr[idxC[q] + 1]++;
      for (int64_t k2 = 0; k2 < ncon; k2++) rptr[k2 + 1] += rptr[k2];
      std::vector<int32_t> fillr(rptr.begin(), rptr.end() - 1);
      for (int64_t j2 = 0; j2 < nvar; j2++)
        for (int32_t q = ptrC[j2]; q < ptrC[j2 + 1]; q++) { const int32_t w = fillr[idxC[q]]++; rslot[w] = slotC[q]; rcol[w] = (int32_t)j2; }
      if ((rc = upload(h, rptr, &J.rptrC))) return bail(rc);
      if ((rc = upload(h, rslot, &J.rslotC))) return bail(rc);
      if ((rc = upload(h, rcol, &J.rcolC))) return bail(rc);
    }
    J.nvar = (int32_t)nvar; J.nequ = (int32_t)nequ; J.ncon = (int32_t)ncon; J.N = (int32_t)N; J.nnz = (int32_t)nnz;
    // (round 5) column tiles of row f1 (kernels.h: DevJt::rv_*): the slot / index ranges of every tile of RVT_COLS columns
    if (plan->opt.f1_tiles && nvar > 0) {
      const int32_t nt = (int32_t)((nvar + cnl::RVT_COLS - 1) / cnl::RVT_COLS);
      std::vector<int32_t> tiles((size_t)nt * cnl::RVT_TW, 0);
      std::vector<uint32_t> table((size_t)nt * (cnl::RVT_KF + cnl::RVT_KC + 1) * cnl::RVT_COLS, 0u);
      bool ok = true;
      int32_t lds_max = 0;
      for (int32_t t = 0; t < nt && ok; t++) {
        const int64_t c0 = (int64_t)t * cnl::RVT_COLS, c1 = std::min<int64_t>(nvar, c0 + cnl::RVT_COLS);
        int32_t fslo = INT32_MAX, fshi = -1, rlo = INT32_MAX, rhi = -1, cslo = INT32_MAX, cshi = -1, llo = INT32_MAX, lhi = -1;
        for (int32_t q = ptrF[c0]; q < ptrF[c1]; q++) { fslo = std::min(fslo, slotF[q]); fshi = std::max(fshi, slotF[q]); rlo = std::min(rlo, idxF[q]); rhi = std::max(rhi, idxF[q]); }
        for (int32_t q = ptrC[c0]; q < ptrC[c1]; q++) { cslo = std::min(cslo, slotC[q]); cshi = std::max(cshi, slotC[q]); llo = std::min(llo, idxC[q]); lhi = std::max(lhi, idxC[q]); }
        int32_t* T = &tiles[(size_t)t * cnl::RVT_TW];
        T[cnl::RVT_FSLO] = fshi < 0 ? 0 : fslo; T[cnl::RVT_WF] = fshi < 0 ? 0 : fshi - fslo + 1;
        T[cnl::RVT_RLO] = rhi < 0 ? 0 : rlo;   T[cnl::RVT_WR] = rhi < 0 ? 0 : rhi - rlo + 1;
        T[cnl::RVT_CSLO] = cshi < 0 ? 0 : cslo; T[cnl::RVT_WC] = cshi < 0 ? 0 : cshi - cslo + 1;
        T[cnl::RVT_LLO] = lhi < 0 ? 0 : llo;   T[cnl::RVT_WL] = lhi < 0 ? 0 : lhi - llo + 1;
        if (T[cnl::RVT_WF] > cnl::RVT_MAXF || T[cnl::RVT_WR] > cnl::RVT_MAXR || T[cnl::RVT_WC] > cnl::RVT_MAXC || T[cnl::RVT_WL] > cnl::RVT_MAXL) { ok = false; break; }
        auto even = [](int32_t w) { return (w + 3) & ~1; };   // a window and the double its 16-byte alignment may put in front
        lds_max = std::max(lds_max, even(T[cnl::RVT_WF]) + even(T[cnl::RVT_WR]) + even(T[cnl::RVT_WC]) + even(T[cnl::RVT_WL]));
        uint32_t* tab = &table[(size_t)t * (cnl::RVT_KF + cnl::RVT_KC + 1) * cnl::RVT_COLS];
        for (int64_t c = c0; c < c1; c++) {
          const int32_t nF = ptrF[c + 1] - ptrF[c], nC = ptrC[c + 1] - ptrC[c];
          if (nF > 255 || nC > 255) { ok = false; break; }
          for (int32_t u = 0; u < std::min(nF, cnl::RVT_KF); u++)
            tab[(size_t)u * cnl::RVT_COLS + (c - c0)] = (uint32_t)(slotF[ptrF[c] + u] - T[cnl::RVT_FSLO]) | (uint32_t)(idxF[ptrF[c] + u] - T[cnl::RVT_RLO]) << 16;
          for (int32_t u = 0; u < std::min(nC, cnl::RVT_KC); u++)
            tab[(size_t)(cnl::RVT_KF + u) * cnl::RVT_COLS + (c - c0)] = (uint32_t)(slotC[ptrC[c] + u] - T[cnl::RVT_CSLO]) | (uint32_t)(idxC[ptrC[c] + u] - T[cnl::RVT_LLO]) << 16;
          tab[(size_t)(cnl::RVT_KF + cnl::RVT_KC) * cnl::RVT_COLS + (c - c0)] = (uint32_t)nF | (uint32_t)nC << 8;
        }
      }
      if (ok) {
        // the residual rows a tile has in LDS anyway are the rows whose primal entry F - r it writes: possible when the tiles' row
        // ranges are ordered and cover 0 .. nequ without gaps (a band); otherwise tiles of rows of their own follow the column tiles
        bool own = nequ > 0;
        int32_t prev = 0;
        for (int32_t t = 0; t < nt && own; t++) {
          int32_t* T = &tiles[(size_t)t * cnl::RVT_TW];
          const int32_t lo = T[cnl::RVT_RLO], hi = lo + T[cnl::RVT_WR];
          const int32_t next_lo = t + 1 < nt ? tiles[(size_t)(t + 1) * cnl::RVT_TW + cnl::RVT_RLO] : (int32_t)nequ;
          const int32_t own_hi = t + 1 < nt ? std::min(hi, std::max(next_lo, prev)) : (int32_t)nequ;
          if (prev < lo || own_hi > hi || own_hi < prev) { own = false; break; }
          T[cnl::RVT_OWNLO] = prev; T[cnl::RVT_OWNHI] = own_hi;
          prev = own_hi;
        }
        if (own && prev != nequ) own = false;
        if (!own) for (int32_t t = 0; t < nt; t++) tiles[(size_t)t * cnl::RVT_TW + cnl::RVT_OWNLO] = tiles[(size_t)t * cnl::RVT_TW + cnl::RVT_OWNHI] = 0;
        if ((rc = upload(h, tiles, &J.rv_tiles))) return bail(rc);
        if ((rc = upload(h, table, &J.rv_table))) return bail(rc);
        J.rv_ntiles = nt; J.rv_lds_doubles = lds_max;
        J.rv_primal_tiles = own ? 0 : (int32_t)((nequ + cnl::RVT_PROWS - 1) / cnl::RVT_PROWS);
      }
      if (std::getenv("CNL_VERBOSE")) fprintf(stderr, "[cnl] row f1: %s\n", ok ? "column tiles" : "gather kernel (a tile's windows exceed the limits)");
    }
  }
  *hout = h;
  return CNL_OK;
}

int cnl_destroy(cnl_handle* h) {
  if (!h) return CNL_OK;
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  for (void* p : h->dev_allocs) (void)hipFree(p);
  if (h->pin) (void)hipHostFree(h->pin);
  cnl::dense_destroy(h->dense);
  cnl::dense_destroy(h->gdense);
  if (h->tail) { cnl_destroy(h->tail); h->tail = nullptr; (void)hipSetDevice(h->device); }
  if (h->aux_stream) { (void)hipStreamSynchronize(h->aux_stream); (void)hipStreamDestroy(h->aux_stream); }
  if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
  if (h->ev_join) (void)hipEventDestroy(h->ev_join);
  for (hipEvent_t e : h->pipe_ev) (void)hipEventDestroy(e);
  for (hipStream_t st : h->pipe_stream) if (st) (void)hipStreamDestroy(st);
  if (h->ev0) (void)hipEventDestroy(h->ev0);
  if (h->ev1) (void)hipEventDestroy(h->ev1);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  cnl_plan_destroy(h->plan);
  delete h;
  return CNL_OK;
}

const cnl_plan* cnl_get_plan(const cnl_handle* h) { return h ? h->plan : nullptr; }

int cnl_dataflow_timeouts(cnl_handle* h, int64_t* count) {
  if (!h || !count) return fail(CNL_ERR_ARG, "null argument");
  *count = 0;
  if (h->d_status) {
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipDeviceSynchronize());
    int v = 0;
    HIPCHK(hipMemcpy(&v, h->d_status, sizeof(int), hipMemcpyDeviceToHost));
    *count = v;
  }
  if (h->tail) {
    int64_t tc = 0;
    const int rc = cnl_dataflow_timeouts(h->tail, &tc);
    if (rc) return rc;
    *count += tc;
  }
  return CNL_OK;
}

int cnl_prepare_newton_system_dev(cnl_handle* h, int64_t nnzhF, int64_t nnzhc, int64_t nnzjF, int64_t nnzjc, const double* d_hF,
                                  const double* d_hc, const double* d_Jx, const double* d_Jcx, const double* d_delta, double* d_vals,
                                  void* stream) {
  if (!h || !d_vals || !d_Jx) return fail(CNL_ERR_ARG, "null argument");
  const cnl::DevJt& J = h->djt;
  if (nnzhF < 0 || nnzhc < 0 || nnzjF < 0 || nnzjc < 0 || nnzhF + nnzhc + nnzjF + nnzjc + J.nequ + J.ncon + J.nvar != J.nnz)
    return fail(CNL_ERR_DIM, "segment sizes do not add up to nnz (7-segment layout of src/CaNNOLeS.jl:256-315)");
  if (J.ncon > 0 && (!d_hc || !d_Jcx || !d_delta)) return fail(CNL_ERR_ARG, "hc / Jcx / delta are required when ncon > 0");
  if (J.ncon == 0 && (nnzhc != 0 || nnzjc != 0)) return fail(CNL_ERR_DIM, "constraint segments must be empty when ncon == 0");
  HIPCHK(hipSetDevice(h->device));
  hipError_t e = cnl::launch_prepare((int)nnzhF, (int)nnzhc, (int)nnzjF, (int)nnzjc, J.nvar, J.nequ, J.ncon, d_hF, d_hc, d_Jx, d_Jcx,
                                     d_delta, d_vals, (int)h->batch, h->layout & 1, (hipStream_t)stream);
  if (e != hipSuccess) return fail(CNL_ERR_HIP, std::string("prepare_newton_system: ") + hipGetErrorString(e));
  return CNL_OK;
}

static int cgls_impl(cnl_handle* h, const cnl::JacSrc& S, const double* d_r, double* d_lambda, double* d_Jxtr, double atol,
                     double rtol, int64_t itmax, int ones_if_zero, int32_t* d_iters, void* stream) {
  if (h->djt.ncon == 0) return CNL_OK;  // nothing to estimate
  if (!d_lambda) return fail(CNL_ERR_ARG, "null lambda");
  if (h->djt.ncon > 1024) return fail(CNL_ERR_DIM, "cnl_cgls_multipliers_dev supports at most 1024 constraints");
  HIPCHK(hipSetDevice(h->device));
  if (!h->d_cgls_ws) {
    int rc = dalloc(h, &h->d_cgls_ws, (size_t)h->batch * 2 * (size_t)h->djt.nvar);
    if (rc) return rc;
  }
  if (itmax <= 0) itmax = (int64_t)h->djt.nvar + h->djt.ncon;  // Krylov.jl's default: m + n
  hipError_t e = cnl::launch_cgls(h->djt, S, d_r, d_lambda, d_Jxtr, h->d_cgls_ws, d_iters, atol, rtol, (int)itmax, ones_if_zero,
                                  (int)h->batch, (hipStream_t)stream);
  if (e != hipSuccess) return fail(CNL_ERR_HIP, std::string("cgls: ") + hipGetErrorString(e));
  return CNL_OK;
}

int cnl_cgls_multipliers_dev(cnl_handle* h, const double* d_vals, const double* d_r, double* d_lambda, double* d_Jxtr, double atol,
                             double rtol, int64_t itmax, int ones_if_zero, int32_t* d_iters, void* stream) {
  if (!h || !d_vals || !d_r) return fail(CNL_ERR_ARG, "null argument");
  if (h->layout & 1) return fail(CNL_ERR_STATE, "cnl_cgls_multipliers_dev reads problem-major vals: on a handle with batch_layout = CNL_LAYOUT_INTERLEAVED use cnl_cgls_multipliers_jac_dev");
  return cgls_impl(h, cnl::JacSrc{d_vals, h->djt.nnz, d_vals, h->djt.nnz, 0, 0}, d_r, d_lambda, d_Jxtr, atol, rtol, itmax, ones_if_zero, d_iters, stream);
}

// the Jacobian values from the model's arrays (problem-major [batch][nnz(J_F)], [batch][nnz(J_c)]: what cnl_prepare_newton_system_dev takes)
static int jac_source(cnl_handle* h, int64_t nnzjF, int64_t nnzjc, const double* d_Jx, const double* d_Jcx, cnl::JacSrc* S) {
  if (!h->jac_segments) return fail(CNL_ERR_STATE, "the J_F / J_c entries of this pattern are not one run of slots each (7-segment layout of src/CaNNOLeS.jl:256-315)");
  if (nnzjF != h->jf_n || nnzjc != h->jc_n) return fail(CNL_ERR_DIM, "nnzjF / nnzjc do not match the pattern's Jacobian entries");
  if (h->jf_n == 0) return fail(CNL_ERR_STATE, "the pattern has no J_F entries");
  if (!d_Jx || (h->jc_n > 0 && !d_Jcx)) return fail(CNL_ERR_ARG, "null Jacobian values");
  // (no J_c entries: the J_c source aliases the J_F one, nothing is read through it but the gather kernel's loads of absent entries)
  if (h->jc_n > 0) *S = cnl::JacSrc{d_Jx - h->jf_lo, h->jf_n, d_Jcx - h->jc_lo, h->jc_n, (int)h->jf_lo, (int)h->jc_lo};
  else *S = cnl::JacSrc{d_Jx - h->jf_lo, h->jf_n, d_Jx - h->jf_lo, h->jf_n, (int)h->jf_lo, (int)h->jf_lo};
  return CNL_OK;
}

int cnl_cgls_multipliers_jac_dev(cnl_handle* h, int64_t nnzjF, int64_t nnzjc, const double* d_Jx, const double* d_Jcx, const double* d_r,
                                 double* d_lambda, double* d_Jxtr, double atol, double rtol, int64_t itmax, int ones_if_zero, int32_t* d_iters,
                                 void* stream) {
  if (!h || !d_r) return fail(CNL_ERR_ARG, "null argument");
  cnl::JacSrc S{};
  if (int rc = jac_source(h, nnzjF, nnzjc, d_Jx, d_Jcx, &S)) return rc;
  return cgls_impl(h, S, d_r, d_lambda, d_Jxtr, atol, rtol, itmax, ones_if_zero, d_iters, stream);
}

static int residual_vectors_impl(cnl_handle* h, const cnl::JacSrc& S, const double* d_r, const double* d_lambda, const double* d_Fx,
                                 const double* d_cx, double* d_rhs, double* d_norms, void* stream) {
  if (!d_r || !d_Fx || !d_rhs || !d_norms) return fail(CNL_ERR_ARG, "null argument");
  if (h->djt.ncon > 0 && (!d_lambda || !d_cx)) return fail(CNL_ERR_ARG, "lambda / c are required when ncon > 0");
  HIPCHK(hipSetDevice(h->device));
  hipError_t e = cnl::launch_residual_vectors(h->djt, S, d_r, d_lambda, d_Fx, d_cx, d_rhs, d_norms, (int)h->batch, (hipStream_t)stream);
  if (e != hipSuccess) return fail(CNL_ERR_HIP, std::string("residual_vectors: ") + hipGetErrorString(e));
  return CNL_OK;
}

int cnl_residual_vectors_dev(cnl_handle* h, const double* d_vals, const double* d_r, const double* d_lambda, const double* d_Fx,
                             const double* d_cx, double* d_rhs, double* d_norms, void* stream) {
  if (!h || !d_vals) return fail(CNL_ERR_ARG, "null argument");
  if (h->layout & 1) return fail(CNL_ERR_STATE, "cnl_residual_vectors_dev reads problem-major vals: on a handle with batch_layout = CNL_LAYOUT_INTERLEAVED use cnl_residual_vectors_jac_dev");
  return residual_vectors_impl(h, cnl::JacSrc{d_vals, h->djt.nnz, d_vals, h->djt.nnz, 0, 0}, d_r, d_lambda, d_Fx, d_cx, d_rhs, d_norms, stream);
}

int cnl_residual_vectors_jac_dev(cnl_handle* h, int64_t nnzjF, int64_t nnzjc, const double* d_Jx, const double* d_Jcx, const double* d_r,
                                 const double* d_lambda, const double* d_Fx, const double* d_cx, double* d_rhs, double* d_norms, void* stream) {
  if (!h) return fail(CNL_ERR_ARG, "null argument");
  cnl::JacSrc S{};
  if (int rc = jac_source(h, nnzjF, nnzjc, d_Jx, d_Jcx, &S)) return rc;
  return residual_vectors_impl(h, S, d_r, d_lambda, d_Fx, d_cx, d_rhs, d_norms, stream);
}

int cnl_trial_point_dev(cnl_handle* h, const double* d_x, const double* d_r, const double* d_lambda, const double* d_d,
                        double max_dlambda, double* d_xt, double* d_rt, double* d_lambdat, double* d_dlambda, void* stream) {
  if (!h || !d_x || !d_r || !d_d || !d_xt || !d_rt) return fail(CNL_ERR_ARG, "null argument");
  if (h->djt.ncon > 0 && (!d_lambda || !d_lambdat || !d_dlambda)) return fail(CNL_ERR_ARG, "lambda vectors are required when ncon > 0");
  HIPCHK(hipSetDevice(h->device));
  hipError_t e = cnl::launch_trial_point(h->djt, d_x, d_r, d_lambda, d_d, max_dlambda, d_xt, d_rt, d_lambdat, d_dlambda, (int)h->batch,
                                         (hipStream_t)stream);
  if (e != hipSuccess) return fail(CNL_ERR_HIP, std::string("trial_point: ") + hipGetErrorString(e));
  return CNL_OK;
}

// ---- cnl_options.batch_layout = CNL_LAYOUT_INTERLEAVED: lengths and conversions (csrc/band.h: band_il_index) ----
static int layout_rowlen(const cnl_handle* h, int which, int64_t* len) {
  if (which == 0) *len = h->djt.nnz;
  else if (which == 1) *len = (int64_t)h->djt.nvar + h->djt.nequ + h->djt.ncon;
  else return fail(CNL_ERR_ARG, "which: 0 = vals, 1 = an N-vector per problem (rhs)");
  return CNL_OK;
}
int cnl_layout_len(const cnl_handle* h, int which, int64_t* doubles) {
  if (!h || !doubles) return fail(CNL_ERR_ARG, "null argument");
  int64_t len = 0;
  if (int rc = layout_rowlen(h, which, &len)) return rc;
  *doubles = cnl::band_il_len(h->batch, len);
  return CNL_OK;
}
static int convert_layout(cnl_handle* h, int which, const double* src, double* dst, int to_interleaved, void* stream) {
  if (!h || !src || !dst) return fail(CNL_ERR_ARG, "null argument");
  if (src == dst) return fail(CNL_ERR_ARG, "the conversion is not in place");
  int64_t len = 0;
  if (int rc = layout_rowlen(h, which, &len)) return rc;
  HIPCHK(hipSetDevice(h->device));
  hipError_t e = cnl::launch_interleave(src, dst, (int)h->batch, len, to_interleaved, (hipStream_t)stream);
  if (e != hipSuccess) return fail(CNL_ERR_HIP, std::string("interleave: ") + hipGetErrorString(e));
  return CNL_OK;
}
int cnl_interleave_dev(cnl_handle* h, int which, const double* d_src, double* d_dst, void* stream) { return convert_layout(h, which, d_src, d_dst, 1, stream); }
int cnl_deinterleave_dev(cnl_handle* h, int which, const double* d_src, double* d_dst, void* stream) { return convert_layout(h, which, d_src, d_dst, 0, stream); }

int cnl_set_timing(cnl_handle* h, int enable) {
  if (!h) return fail(CNL_ERR_ARG, "null handle");
  h->timing = enable != 0;
  return CNL_OK;
}

int cnl_last_kernel_ms(cnl_handle* h, float* ms) {
  if (!h || !ms) return fail(CNL_ERR_ARG, "null argument");
  *ms = h->last_ms;
  return CNL_OK;
}

int cnl_launch_counts(int64_t counts[3]) {
  if (!counts) return fail(CNL_ERR_ARG, "null argument");
  for (int k = 0; k < 3; k++) counts[k] = g_launches[k].load();
  return CNL_OK;
}

int cnl_get_config(const cnl_handle* h, int64_t cfg[8]) {
  if (!h || !cfg) return fail(CNL_ERR_ARG, "null argument");
  std::memset(cfg, 0, 8 * sizeof(int64_t));
  cfg[0] = h->cfg.tpp; cfg[1] = h->cfg.ppb; cfg[2] = (int64_t)h->cfg.lds_bytes; cfg[3] = h->cfg.lds_work;
  cfg[4] = (h->batch + h->cfg.ppb - 1) / h->cfg.ppb;
  cfg[5] = (h->dense || h->gdense) ? 3 : (h->use_v2 ? (h->staged ? 4 : 2) : 1);
  if (h->lean && !h->dense && !h->gdense) cfg[5] |= 16;  // newton_system / factorize run the kernels' LEAN instantiation
  if (h->tail) cfg[5] |= 32;                             // the remainder of the batch runs on a handle of its own (split_tail)
  if (h->band) cfg[5] |= 64 | ((int64_t)h->band_nl << 8) | ((int64_t)h->bd.nparts << 16) | (h->band_mw ? (int64_t)1 << 24 : 0) | ((int64_t)h->layout << 25);   // newton_system runs on the band kernels (csrc/band.h): problems per workgroup, parts
  if (h->djt.rv_ntiles > 0) cfg[5] |= 128;               // row f1 runs on column tiles (kernels.h: DevJt::rv_*)
  cfg[6] = h->wpb2;
  cfg[7] = (int64_t)h->lds2;
  return CNL_OK;
}

#ifdef CNL_STAMPS
int cnl_debug_stamps(cnl_handle* h, int64_t* out, int64_t n) {
  if (!h || !h->d_npos) return fail(CNL_ERR_ARG, "no stamps");
  HIPCHK(hipMemcpy(out, h->d_npos, (size_t)n * sizeof(int64_t), hipMemcpyDeviceToHost));
  return CNL_OK;
}
#endif

// ---- device-pointer entry points -------------------------------------------------
int cnl_factorize_dev(cnl_handle* h, const double* d_vals, double eig_tol, int32_t* d_success, void* stream) {
  if (!h || !d_vals || !d_success) return fail(CNL_ERR_ARG, "null argument");
  HIPCHK(hipSetDevice(h->device));
  cnl::LaunchArgs a{};
  a.mode = cnl::MODE_FACTOR;
  a.success = d_success;
  a.params[0] = eig_tol;
  int rc = run(h, a, const_cast<double*>(d_vals), nullptr, nullptr, (hipStream_t)stream);
  if (rc == CNL_OK) { h->factorized = true; h->last_vals = d_vals; h->last_ok.clear(); }
  return rc;
}

int cnl_solve_dev(cnl_handle* h, const double* d_rhs, double* d_d, void* stream) {
  if (!h || !d_rhs || !d_d) return fail(CNL_ERR_ARG, "null argument");
  if (!h->factorized) return fail(CNL_ERR_STATE, "cnl_solve before cnl_factorize");
  HIPCHK(hipSetDevice(h->device));
  cnl::LaunchArgs a{};
  a.mode = cnl::MODE_SOLVE;
  return run(h, a, nullptr, d_rhs, d_d, (hipStream_t)stream);
}

int cnl_newton_system_dev(cnl_handle* h, double* d_vals, const double* d_rhs, double* d_d, double* d_rho_old, double* d_rho,
                          int32_t* d_nfact, int32_t* d_success, const double params[9], void* stream) {
  if (!h || !d_vals || !d_rhs || !d_d || !d_rho_old || !d_rho || !d_nfact || !d_success || !params)
    return fail(CNL_ERR_ARG, "null argument");
  HIPCHK(hipSetDevice(h->device));
  cnl::LaunchArgs a{};
  a.mode = cnl::MODE_NEWTON;
  a.rho_old = d_rho_old; a.rho = d_rho; a.nfact = d_nfact; a.success = d_success;
  std::memcpy(a.params, params, 9 * sizeof(double));
  int rc = run(h, a, d_vals, d_rhs, d_d, (hipStream_t)stream);
  if (rc == CNL_OK) { h->factorized = true; h->last_vals = d_vals; h->last_ok.clear(); }
  return rc;
}

// ---- host-pointer entry points (what the Julia glue ccalls) ------------------------
int cnl_factorize(cnl_handle* h, const double* vals, double eig_tol, int32_t* success, int64_t* npos, int64_t* nzero) {
  if (!h || !vals || !success) return fail(CNL_ERR_ARG, "null argument");
  if (h->layout) return fail(CNL_ERR_STATE, "host-pointer calls take the reference's problem-major arrays: this handle was created with batch_layout = CNL_LAYOUT_INTERLEAVED (device-pointer entry points only)");
  HIPCHK(hipSetDevice(h->device));
  int rc = ensure_staging(h);
  if (rc) return rc;
  const cnl_plan& P = *h->plan;
  const size_t B = (size_t)h->batch;
  HIPCHK(hipMemcpyAsync(h->d_vals, vals, B * P.nnz * sizeof(double), hipMemcpyHostToDevice, h->stream));
  cnl::LaunchArgs a{};
  a.mode = cnl::MODE_FACTOR;
  a.success = h->d_success; a.npos = h->d_npos; a.nzero = h->d_nzero;
  a.params[0] = eig_tol;
  if ((rc = run(h, a, h->d_vals, nullptr, nullptr, h->stream))) return rc;
  h->last_vals = h->d_vals;
  {
    // results through the handle's pinned block: asynchronous copies (a copy into pageable memory blocks, one after the other),
    // the inertia counts with one copy (npos | nzero are one block on the device, ensure_staging), a single synchronisation
    const size_t o_su = 0, o_np = (B * 4 + 7) & ~(size_t)7, total = o_np + B * 16;
    if (!h->pin || h->pin_bytes < total) {
      if (h->pin) (void)hipHostFree(h->pin);
      h->pin = nullptr;
      HIPCHK(hipHostMalloc(&h->pin, total, hipHostMallocDefault));
      h->pin_bytes = total;
    }
    char* pb = static_cast<char*>(h->pin);
    HIPCHK(hipMemcpyAsync(pb + o_su, h->d_success, B * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
    if (npos || nzero) HIPCHK(hipMemcpyAsync(pb + o_np, h->d_npos, B * 16, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    std::memcpy(success, pb + o_su, B * sizeof(int32_t));
    if (npos) std::memcpy(npos, pb + o_np, B * sizeof(int64_t));
    if (nzero) std::memcpy(nzero, pb + o_np + B * 8, B * sizeof(int64_t));
  }
  h->factorized = true;
  h->last_ok.assign(success, success + B);
  return CNL_OK;
}

int cnl_solve(cnl_handle* h, const double* rhs, double* d) {
  if (!h || !rhs || !d) return fail(CNL_ERR_ARG, "null argument");
  if (h->layout) return fail(CNL_ERR_STATE, "host-pointer calls take the reference's problem-major arrays: this handle was created with batch_layout = CNL_LAYOUT_INTERLEAVED (device-pointer entry points only)");
  if (!h->factorized) return fail(CNL_ERR_STATE, "cnl_solve before cnl_factorize");
  HIPCHK(hipSetDevice(h->device));
  int rc = ensure_staging(h);
  if (rc) return rc;
  const cnl_plan& P = *h->plan;
  const size_t B = (size_t)h->batch;
  HIPCHK(hipMemcpyAsync(h->d_rhs, rhs, B * P.N * sizeof(double), hipMemcpyHostToDevice, h->stream));
  cnl::LaunchArgs a{};
  a.mode = cnl::MODE_SOLVE;
  const bool known = h->last_ok.size() == B;
  if (known && B == 1 && !h->last_ok[0])
    return fail(CNL_ERR_STATE, "cnl_solve after a factorisation that failed (success = 0): there is no factor to solve with "
                               "(the reference calls solve_ldl! only after a successful try_to_factorize, src/CaNNOLeS.jl:1049)");
  if ((rc = run(h, a, nullptr, h->d_rhs, h->d_d, h->stream))) return rc;
  if (!known) {
    HIPCHK(hipMemcpyAsync(d, h->d_d, B * P.N * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  } else {
    size_t b0 = 0;
    while (b0 < B) {  // rows of the problems that hold a factor; the others stay as the caller passed them
      while (b0 < B && !h->last_ok[b0]) b0++;
      size_t b1 = b0;
      while (b1 < B && h->last_ok[b1]) b1++;
      if (b1 > b0) HIPCHK(hipMemcpyAsync(d + b0 * P.N, h->d_d + b0 * P.N, (b1 - b0) * P.N * sizeof(double), hipMemcpyDeviceToHost, h->stream));
      b0 = b1;
    }
  }
  HIPCHK(hipStreamSynchronize(h->stream));
  return CNL_OK;
}

// Large host-pointer batches, chunk by chunk: the call is bound by the host link (1.1 MB per system go up, 0.25 MB come back),
// which is full duplex — while chunk c + 1 goes up (calling thread, alternating between two streams) chunk c is computed and the
// results of chunk c - 1 come down (a helper thread of the call, third stream).  Copies between pageable host memory and the
// device block their host thread, hence the second thread; the compute of a chunk hides behind the upload of the next.
static int host_ladder_run(cnl_handle* h, const double params[9], const double* rho_old, double* rho, double* rho_old_out, int32_t* nfact,
                           int32_t* success, char* up, int32_t* su_pin);

static int newton_system_pipelined(cnl_handle* h, double* vals, const double* rhs, double* d, const double* rho_old, const double params[9],
                                   double* rho, double* rho_old_out, int32_t* nfact, int32_t* success, size_t chunk) {
  const cnl_plan& P = *h->plan;
  const size_t B = (size_t)h->batch;
  const size_t nchunks = (B + chunk - 1) / chunk;
  for (hipStream_t& st : h->pipe_stream) if (!st) HIPCHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  while (h->pipe_ev.size() < nchunks) {
    hipEvent_t e;
    HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    h->pipe_ev.push_back(e);
  }
  std::mutex mu;        // posted[], abort_, and the handle view (SubBatch mutates the handle: one enqueue at a time)
  std::condition_variable cv;
  std::vector<char> posted(nchunks, 0);
  bool abort_ = false;
  int wrc = CNL_OK;
  std::string wmsg;
  const int device = h->device;
  std::thread down([&]() {
    (void)hipSetDevice(device);
    hipStream_t st = h->pipe_stream[cnl_handle::kPipeUp];
    auto chk = [&](hipError_t e, const char* what) {
      if (e != hipSuccess && wrc == CNL_OK) { wrc = CNL_ERR_HIP; wmsg = std::string(what) + ": " + hipGetErrorString(e); }
      return e == hipSuccess;
    };
    for (size_t c = 0; c < nchunks; c++) {
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return posted[c] || abort_; });
        if (!posted[c]) return;
      }
      const size_t b0 = c * chunk, nb = std::min(chunk, B - b0);
      if (!chk(hipEventSynchronize(h->pipe_ev[c]), "hipEventSynchronize")) continue;
      if (!chk(hipMemcpyAsync(success + b0, h->d_success + b0, nb * sizeof(int32_t), hipMemcpyDeviceToHost, st), "copy success")) continue;
      if (!chk(hipStreamSynchronize(st), "hipStreamSynchronize")) continue;
      // the reference leaves d untouched when the factorisation fails (src/CaNNOLeS.jl:1049): maximal runs of successes
      size_t q0 = 0;
      while (q0 < nb) {
        while (q0 < nb && !success[b0 + q0]) q0++;
        size_t q1 = q0;
        while (q1 < nb && success[b0 + q1]) q1++;
        if (q1 > q0) chk(hipMemcpyAsync(d + (b0 + q0) * P.N, h->d_d + (b0 + q0) * P.N, (q1 - q0) * P.N * sizeof(double), hipMemcpyDeviceToHost, st), "copy d");
        q0 = q1;
      }
      chk(hipMemcpyAsync(rho + b0, h->d_rho + b0, nb * sizeof(double), hipMemcpyDeviceToHost, st), "copy rho");
      chk(hipMemcpyAsync(rho_old_out + b0, h->d_rho_old + b0, nb * sizeof(double), hipMemcpyDeviceToHost, st), "copy rho_old");
      chk(hipMemcpyAsync(nfact + b0, h->d_nfact + b0, nb * sizeof(int32_t), hipMemcpyDeviceToHost, st), "copy nfact");
      if (P.nvar > 0)
        chk(hipMemcpy2DAsync(vals + b0 * P.nnz + (P.nnz - P.nvar), (size_t)P.nnz * sizeof(double), h->d_vals + b0 * P.nnz + (P.nnz - P.nvar),
                             (size_t)P.nnz * sizeof(double), (size_t)P.nvar * sizeof(double), nb, hipMemcpyDeviceToHost, st), "copy rho slots");
      chk(hipStreamSynchronize(st), "hipStreamSynchronize");
    }
  });
  // two uploaders (the calling thread and a helper) take alternate chunks, each on a stream of its own: a copy out of pageable
  // memory keeps a host thread busy (the runtime pins or stages the pages), and one thread alone does not fill the link
  constexpr int NU = cnl_handle::kPipeUp;
  int urc[NU];
  for (int& r_ : urc) r_ = CNL_OK;
  std::string umsg[NU];
  auto uploader = [&](int u) {
    if (u) (void)hipSetDevice(device);
    hipStream_t st = h->pipe_stream[u];
    for (size_t c = (size_t)u; c < nchunks; c += NU) {
      {
        std::lock_guard<std::mutex> lk(mu);
        if (abort_) return;
      }
      const size_t b0 = c * chunk, nb = std::min(chunk, B - b0);
      hipError_t e = hipMemcpyAsync(h->d_vals + b0 * P.nnz, vals + b0 * P.nnz, nb * P.nnz * sizeof(double), hipMemcpyHostToDevice, st);
      if (e == hipSuccess) e = hipMemcpyAsync(h->d_rhs + b0 * P.N, rhs + b0 * P.N, nb * P.N * sizeof(double), hipMemcpyHostToDevice, st);
      if (e == hipSuccess) e = rho_old ? hipMemcpyAsync(h->d_rho_old + b0, rho_old + b0, nb * sizeof(double), hipMemcpyHostToDevice, st)
                                       : hipMemsetAsync(h->d_rho_old + b0, 0, nb * sizeof(double), st);
      int r = CNL_OK;
      std::string m;
      if (e != hipSuccess) { r = CNL_ERR_HIP; m = std::string("upload: ") + hipGetErrorString(e); }
      std::lock_guard<std::mutex> lk(mu);
      if (r == CNL_OK) {
        SubBatch view(h, (int64_t)b0, (int64_t)nb);
        cnl::LaunchArgs a{};
        a.mode = cnl::MODE_NEWTON;
        a.rho_old = h->d_rho_old + b0; a.rho = h->d_rho + b0; a.nfact = h->d_nfact + b0; a.success = h->d_success + b0;
        std::memcpy(a.params, params, 9 * sizeof(double));
        r = run(h, a, h->d_vals + b0 * P.nnz, h->d_rhs + b0 * P.N, h->d_d + b0 * P.N, st);
        if (r) m = g_err;
        if (r == CNL_OK && hipEventRecord(h->pipe_ev[c], st) != hipSuccess) { r = CNL_ERR_HIP; m = "hipEventRecord failed"; }
      }
      if (r != CNL_OK) { urc[u] = r; umsg[u] = m; abort_ = true; cv.notify_all(); return; }
      posted[c] = 1;
      cv.notify_all();
    }
  };
  const bool tm = h->timing;
  h->timing = false;
  h->tail_fresh = false;   // the chunks factorise every problem in THIS handle's storage (views), the remainder's handle is not used
  // staged handles: the chunks run the first attempt only; the problems that failed it go through the host-driven ladder on the
  // whole (now device-resident) batch behind the last chunk (see cnl_newton_system)
  const bool host_ladder = h->staged && !h->dense && !h->gdense && h->plan->opt.host_ladder != 0 && h->use_v2 && P.P.rec_direct && P.P.d_outer;
  h->first_attempt_only = host_ladder;
  std::vector<std::thread> ups;
  for (int u = 1; u < NU; u++) ups.emplace_back(uploader, u);
  uploader(0);
  for (std::thread& th : ups) th.join();
  h->first_attempt_only = false;
  h->timing = tm;
  down.join();
  for (int u = 0; u < NU; u++) (void)hipStreamSynchronize(h->pipe_stream[u]);
  for (int u = 0; u < NU; u++) if (urc[u] != CNL_OK) return fail(urc[u], umsg[u]);
  if (wrc != CNL_OK) return fail(wrc, "download: " + wmsg);
  h->last_vals = h->d_vals;
  if (host_ladder) {
    std::vector<char> failed(B, 0);
    bool any_failed = false;
    for (size_t b = 0; b < B; b++) { failed[b] = !success[b]; any_failed |= failed[b] != 0; }
    if (any_failed) {
      const size_t need = B * 16;
      if (!h->pin || h->pin_bytes < need) {
        if (h->pin) (void)hipHostFree(h->pin);
        h->pin = nullptr;
        HIPCHK(hipHostMalloc(&h->pin, need, hipHostMallocDefault));
        h->pin_bytes = need;
      }
      char* pb2 = static_cast<char*>(h->pin);
      int rc = host_ladder_run(h, params, rho_old, rho, rho_old_out, nfact, success, pb2, reinterpret_cast<int32_t*>(pb2 + B * 12));
      if (rc) return rc;
      for (size_t b = 0; b < B; b++) {
        if (!failed[b]) continue;
        if (success[b]) HIPCHK(hipMemcpyAsync(d + b * P.N, h->d_d + b * P.N, (size_t)P.N * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        if (P.nvar > 0)
          HIPCHK(hipMemcpyAsync(vals + b * P.nnz + (P.nnz - P.nvar), h->d_vals + b * P.nnz + (P.nnz - P.nvar), (size_t)P.nvar * sizeof(double),
                                hipMemcpyDeviceToHost, h->stream));
      }
      HIPCHK(hipStreamSynchronize(h->stream));
    }
  }
  h->factorized = true;
    h->last_ok.assign(success, success + B);
  return CNL_OK;
}

// The rho ladder of src/CaNNOLeS.jl:1023-1047 driven from the host for the problems whose first (staged) factorisation failed:
// every rung is a staged try_to_factorize of the batch and a read-back of the success flags; the solve follows (cnl_newton_system).
// up: pinned staging of 12 bytes per problem; su_pin: pinned, batch ints.  On return rho / rho_old_out / nfact / success hold the
// reference's results and the device holds the factors and the solution of everything that succeeded.
static int host_ladder_run(cnl_handle* h, const double params[9], const double* rho_old, double* rho, double* rho_old_out, int32_t* nfact,
                           int32_t* success, char* up, int32_t* su_pin) {
  const cnl_plan& P = *h->plan;
  const size_t B = (size_t)h->batch;
  int rc;
  const double rho0 = params[5], rhomax = params[6], rhomin = params[7], kdec = params[2], kinc = params[3], klarge = params[4];
  if (!h->d_act && (rc = dalloc(h, &h->d_act, B))) return rc;
  double* up_rho = reinterpret_cast<double*>(up);
  int32_t* up_act = reinterpret_cast<int32_t*>(up + B * 8);
  std::vector<char> act(B, 0);
  std::vector<double> ro_in(B);
  // split handles: only the chain part [0, split_staged) ran the first attempt alone; the single-stream part has been through the
  // whole device ladder already (a problem that exhausted it there must not climb again: nfact would count twice)
  const size_t first_only = (h->split_staged > 0 && (size_t)h->split_staged < B && !h->split_halves && (!h->tail || h->tail_redone)) ? (size_t)h->split_staged : B;
  bool any_act = false;
  for (size_t b = 0; b < B; b++) {
    ro_in[b] = rho_old ? rho_old[b] : 0.0;
    if (!success[b] && b < first_only) { act[b] = 1; any_act = true; rho[b] = ro_in[b] == 0.0 ? rho0 : std::max(rhomin, kdec * ro_in[b]); }
  }
  while (any_act) {
    for (size_t b = 0; b < B; b++) { up_rho[b] = rho[b]; up_act[b] = act[b]; }
    HIPCHK(hipMemcpyAsync(h->d_rho, up_rho, B * 8, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->d_act, up_act, B * 4, hipMemcpyHostToDevice, h->stream));
    hipError_t e = cnl::launch_fill_rho(h->d_vals, P.nnz, (int)P.nvar, h->d_rho, h->d_act, (int)B, h->stream);
    if (e != hipSuccess) return fail(CNL_ERR_HIP, std::string("fill_rho: ") + hipGetErrorString(e));
    cnl::LaunchArgs f{};
    f.mode = cnl::MODE_FACTOR;
    f.success = h->d_success; f.npos = h->d_npos; f.nzero = h->d_nzero;
    std::memcpy(f.params, params, 9 * sizeof(double));
    if ((rc = run(h, f, h->d_vals, nullptr, nullptr, h->stream))) return rc;
    HIPCHK(hipMemcpyAsync(su_pin, h->d_success, B * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    any_act = false;
    for (size_t b = 0; b < B; b++) {
      if (!act[b]) continue;
      nfact[b]++;
      if (su_pin[b]) { success[b] = 1; act[b] = 0; continue; }
      rho[b] = ro_in[b] == 0.0 ? klarge * rho[b] : kinc * rho[b];
      if (rho[b] > rhomax) act[b] = 0;   // the ladder ran out: rho keeps the value beyond rhomax, the slots the last one tried
      else any_act = true;
    }
  }
  for (size_t b = 0; b < first_only; b++)
    if (rho[b] != 0.0 && rho[b] <= rhomax) rho_old_out[b] = rho[b];   // (rho != 0: the problem entered the ladder)
  // solve_ldl! for everything that holds a valid factor now (the problems of the first attempt are solved again: same factor)
  cnl::LaunchArgs sv{};
  sv.mode = cnl::MODE_SOLVE;
  std::memcpy(sv.params, params, 9 * sizeof(double));
  h->last_vals = h->d_vals;
  return run(h, sv, nullptr, h->d_rhs, h->d_d, h->stream);
}

int cnl_newton_system(cnl_handle* h, double* vals, const double* rhs, double* d, const double* rho_old, const double params[9],
                      double* rho, double* rho_old_out, int32_t* nfact, int32_t* success) {
  if (!h || !vals || !rhs || !d || !params || !rho || !rho_old_out || !nfact || !success) return fail(CNL_ERR_ARG, "null argument");
  if (h->layout) return fail(CNL_ERR_STATE, "host-pointer calls take the reference's problem-major arrays: this handle was created with batch_layout = CNL_LAYOUT_INTERLEAVED (device-pointer entry points only)");
  HIPCHK(hipSetDevice(h->device));
  int rc = ensure_staging(h);
  if (rc) return rc;
  const cnl_plan& P = *h->plan;
  const size_t B = (size_t)h->batch;
  if (!h->dense && !h->gdense && !h->timing && B >= 32 && B * (size_t)(P.nnz + P.N) * sizeof(double) >= ((size_t)96 << 20)) {
    // eight chunks or more, each a multiple of four problems and at least 16 MB of upload
    size_t chunk = std::max<size_t>(16, ((B / 8) + 3) & ~(size_t)3);
    const size_t per = (size_t)(P.nnz + P.N) * sizeof(double);
    while (chunk * per < ((size_t)16 << 20) && chunk < B) chunk += 4;
    // band kernels: the factor records of a workgroup's problems (8, 16 or 32) are interleaved in ONE region of the factor storage;
    // chunks run concurrently on different streams, so their regions must not share a workgroup
    if (h->band) chunk = (chunk + 31) & ~(size_t)31;
    if (chunk < B) return newton_system_pipelined(h, vals, rhs, d, rho_old, params, rho, rho_old_out, nfact, success, chunk);
  }
  HIPCHK(hipMemcpyAsync(h->d_vals, vals, B * P.nnz * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipMemcpyAsync(h->d_rhs, rhs, B * P.N * sizeof(double), hipMemcpyHostToDevice, h->stream));
  if (rho_old) HIPCHK(hipMemcpyAsync(h->d_rho_old, rho_old, B * sizeof(double), hipMemcpyHostToDevice, h->stream));
  else HIPCHK(hipMemsetAsync(h->d_rho_old, 0, B * sizeof(double), h->stream));
  if (h->dense && !h->timing && h->plan->opt.host_ladder != 0) {
    // Dense backend, host-pointer call: the reference's own sequence — try_to_factorize, the rho ladder on the host, solve_ldl! — with
    // the library's multi-kernel factorisation for EVERY rung.  The fused device call decides its ladder on the device, where a
    // problem that failed is refactorised by ONE workgroup walking all tiles (dn_ladder: 7.8 ms per rung at cfg2's size against
    // 0.35 ms for the launch sequence): 48 ms for a cfg2 system that climbs to nfact = 7.  Costs one more synchronisation when
    // nothing fails.
    cnl::LaunchArgs f{};
    f.mode = cnl::MODE_FACTOR;
    f.success = h->d_success; f.npos = h->d_npos; f.nzero = h->d_nzero;
    std::memcpy(f.params, params, 9 * sizeof(double));
    if ((rc = run(h, f, h->d_vals, nullptr, nullptr, h->stream))) return rc;
    HIPCHK(hipMemcpyAsync(success, h->d_success, B * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    bool any_failed = false;
    for (size_t b = 0; b < B; b++) {
      rho[b] = 0.0; nfact[b] = 1; rho_old_out[b] = rho_old ? rho_old[b] : 0.0;
      any_failed |= !success[b];
    }
    if (any_failed) {
      const size_t need = B * 16;
      if (!h->pin || h->pin_bytes < need) {
        if (h->pin) (void)hipHostFree(h->pin);
        h->pin = nullptr;
        HIPCHK(hipHostMalloc(&h->pin, need, hipHostMallocDefault));
        h->pin_bytes = need;
      }
      char* pb2 = static_cast<char*>(h->pin);
      if ((rc = host_ladder_run(h, params, rho_old, rho, rho_old_out, nfact, success, pb2, reinterpret_cast<int32_t*>(pb2 + B * 12)))) return rc;
    } else {
      cnl::LaunchArgs sv{};
      sv.mode = cnl::MODE_SOLVE;
      std::memcpy(sv.params, params, 9 * sizeof(double));
      if ((rc = run(h, sv, nullptr, h->d_rhs, h->d_d, h->stream))) return rc;
    }
    for (size_t b = 0; b < B; b++) {
      if (success[b]) HIPCHK(hipMemcpyAsync(d + b * P.N, h->d_d + b * P.N, (size_t)P.N * sizeof(double), hipMemcpyDeviceToHost, h->stream));
      if (nfact[b] > 1 && P.nvar > 0)
        HIPCHK(hipMemcpyAsync(vals + b * P.nnz + (P.nnz - P.nvar), h->d_vals + b * P.nnz + (P.nnz - P.nvar), (size_t)P.nvar * sizeof(double),
                              hipMemcpyDeviceToHost, h->stream));
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    h->last_vals = h->d_vals;
    h->factorized = true;
    h->last_ok.assign(success, success + B);
    return CNL_OK;
  }
  cnl::LaunchArgs a{};
  a.mode = cnl::MODE_NEWTON;
  a.rho_old = h->d_rho_old; a.rho = h->d_rho;
  a.nfact = h->d_nfact; a.success = h->d_success;
#ifdef CNL_STAMPS
  a.npos = h->d_npos;  // diagnostic build: per-wave phase stamps land here (cnl_debug_stamps)
#endif
  std::memcpy(a.params, params, 9 * sizeof(double));
  // Small batches on a staged handle: the rho ladder is driven from the HOST, like the reference's own newton_system!
  // (src/CaNNOLeS.jl:1023-1047) — every rung is a staged try_to_factorize (all tasks of the elimination tree in parallel, ~0.1 ms)
  // and a read-back of the success flags.  The sequential launch that takes failed problems through the ladder on the device
  // walks the latency plan's fronts one after the other on ONE wavefront per four problems: 7.7 ms per rung for a system of
  // cfg3's size (38 ms for nfact = 6, ten times one CPU core of the oracle); it stays the device-pointer calls' fallback.
  const bool small = B * (size_t)P.N * sizeof(double) <= ((size_t)1 << 20);
  // (handles whose in-kernel device ladder is available use that one: no round trip per rung — 0.67 against 0.9 ms for one system of
  //  cfg3's size that climbs to nfact = 6)
  const bool host_ladder = h->staged && !h->dense && !h->gdense && h->plan->opt.host_ladder != 0 && h->use_v2 && P.P.rec_direct && P.P.d_outer &&
                           (h->lad_mode == 0 || h->split_staged > 0);
  h->first_attempt_only = host_ladder;
  rc = run(h, a, h->d_vals, h->d_rhs, h->d_d, h->stream);
  h->first_attempt_only = false;
  if (rc) return rc;
  h->last_vals = h->d_vals;
  // Small batches (the reference's own call: one system): every result goes to ONE pinned block of the handle with asynchronous
  // copies and a single synchronisation; the caller's arrays are then filled on the host by the reference's rules (d only
  // where the factorisation succeeded, the rho slots of vals only where the ladder wrote them).  Copies into pageable memory
  // block one by one, and the success flags would need a round trip of their own before d may be copied.
  if (small) {
    const size_t o_d = 0, o_tail = o_d + B * P.N * 8, o_rho = o_tail + B * P.nvar * 8, o_ro = o_rho + B * 8, o_nf = o_ro + B * 8,
                 o_su = o_nf + B * 4, o_up = (o_su + B * 4 + 7) & ~(size_t)7, total = o_up + B * 12 + 8;
    if (!h->pin || h->pin_bytes < total) {
      if (h->pin) (void)hipHostFree(h->pin);
      h->pin = nullptr;
      HIPCHK(hipHostMalloc(&h->pin, total, hipHostMallocDefault));
      h->pin_bytes = total;
    }
    char* pb = static_cast<char*>(h->pin);
    HIPCHK(hipMemcpyAsync(pb + o_d, h->d_d, B * P.N * 8, hipMemcpyDeviceToHost, h->stream));
    // (with the host-driven ladder the first attempt never writes the rho slots: they come back only behind a ladder, or behind the
    //  sequential redo of a dataflow time-out)
    // (the rho slots come back only where a ladder wrote them: nfact > 1 — known after the first synchronisation; copying them
    //  on every call cost the common one-system call 40 us of its 0.24 ms)
    bool tail_valid = false;
    // rho, rho_old, nfact, success: one block on both sides (ensure_staging), one copy
    static_assert(sizeof(double) == 8 && sizeof(int32_t) == 4, "layout of the result block");
    HIPCHK(hipMemcpyAsync(pb + o_rho, h->d_rho, B * 24, hipMemcpyDeviceToHost, h->stream));
    int32_t* up_status = reinterpret_cast<int32_t*>(pb + o_up + B * 12);
    *up_status = 0;
    // per-call status of the dataflow execution (kernels2.hip, spin_until): waits that gave up.  Only calls that ran in dataflow
    // fashion on the whole handle can set it (split handles run through views: one launch per stage, nothing waits)
    if (host_ladder && h->d_dep && h->d_stat && h->split_staged == 0)
      HIPCHK(hipMemcpyAsync(up_status, h->d_stat, 4, hipMemcpyDeviceToHost, h->stream));
    //  ... and of the remainder handle's call, which does run in dataflow fashion on a handle of its own
    int32_t tail_status = 0;
    if (host_ladder && h->tail && h->tail->d_dep && h->tail->d_stat)
      HIPCHK(hipMemcpyAsync(&tail_status, h->tail->d_stat, 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    h->tail_redone = tail_status != 0;
    std::memcpy(rho, pb + o_rho, B * 8);
    std::memcpy(rho_old_out, pb + o_ro, B * 8);
    std::memcpy(nfact, pb + o_nf, B * 4);
    std::memcpy(success, pb + o_su, B * 4);
    bool any_failed = false;
    for (size_t b = 0; b < B; b++) any_failed |= !success[b];
    if (host_ladder && any_failed && *up_status == 0) {
      if ((rc = host_ladder_run(h, params, rho_old, rho, rho_old_out, nfact, success, pb + o_up, reinterpret_cast<int32_t*>(pb + o_su)))) return rc;
      HIPCHK(hipMemcpyAsync(pb + o_d, h->d_d, B * P.N * 8, hipMemcpyDeviceToHost, h->stream));
      if (P.nvar > 0) {
        HIPCHK(hipMemcpy2DAsync(pb + o_tail, (size_t)P.nvar * 8, h->d_vals + (P.nnz - P.nvar), (size_t)P.nnz * 8, (size_t)P.nvar * 8, B,
                                hipMemcpyDeviceToHost, h->stream));
        tail_valid = true;
      }
      HIPCHK(hipStreamSynchronize(h->stream));
    } else if (P.nvar > 0) {
      // the device climbed a ladder (in-kernel ladder, sequential launch, or the redo behind a dataflow time-out): rho slots back
      bool climbed = false;
      for (size_t b = 0; b < B; b++) climbed |= nfact[b] > 1;
      if (climbed) {
        HIPCHK(hipMemcpy2DAsync(pb + o_tail, (size_t)P.nvar * 8, h->d_vals + (P.nnz - P.nvar), (size_t)P.nnz * 8, (size_t)P.nvar * 8, B,
                                hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        tail_valid = true;
      }
    }
    for (size_t b = 0; b < B; b++) {
      if (success[b]) std::memcpy(d + b * P.N, pb + o_d + b * P.N * 8, (size_t)P.N * 8);
      // rho tail of vals (the reference mutates get_vals(LDLT)[end-nvar+1:end] on retries only, src/CaNNOLeS.jl:1031,1038); the
      // device copy holds what the caller passed wherever the ladder did not write, so copying it back always is the same
      if (P.nvar > 0 && tail_valid) std::memcpy(vals + b * P.nnz + (P.nnz - P.nvar), pb + o_tail + b * P.nvar * 8, (size_t)P.nvar * 8);
    }
    h->factorized = true;
    h->last_ok.assign(success, success + B);
    return CNL_OK;
  }
  // the reference leaves d untouched when the factorisation fails (src/CaNNOLeS.jl:1049): copy back the rows that succeeded
  HIPCHK(hipMemcpyAsync(success, h->d_success, B * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
  int32_t status_word = 0;
  if (host_ladder && h->d_dep && h->d_stat && h->split_staged == 0)
    HIPCHK(hipMemcpyAsync(&status_word, h->d_stat, 4, hipMemcpyDeviceToHost, h->stream));
  int32_t tail_status2 = 0;
  if (host_ladder && h->tail && h->tail->d_dep && h->tail->d_stat)
    HIPCHK(hipMemcpyAsync(&tail_status2, h->tail->d_stat, 4, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  h->tail_redone = tail_status2 != 0;
  bool laddered = false;
  if (host_ladder && status_word == 0) {
    bool any_failed = false;
    for (size_t b = 0; b < B; b++) any_failed |= !success[b];
    if (any_failed) {
      // first attempt's per-problem results (rho = 0, rho_old as given, nfact = 1), then the host ladder on top of them
      HIPCHK(hipMemcpyAsync(rho, h->d_rho, B * sizeof(double), hipMemcpyDeviceToHost, h->stream));
      HIPCHK(hipMemcpyAsync(rho_old_out, h->d_rho_old, B * sizeof(double), hipMemcpyDeviceToHost, h->stream));
      HIPCHK(hipMemcpyAsync(nfact, h->d_nfact, B * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
      HIPCHK(hipStreamSynchronize(h->stream));
      const size_t need = B * 16;
      if (!h->pin || h->pin_bytes < need) {
        if (h->pin) (void)hipHostFree(h->pin);
        h->pin = nullptr;
        HIPCHK(hipHostMalloc(&h->pin, need, hipHostMallocDefault));
        h->pin_bytes = need;
      }
      char* pb2 = static_cast<char*>(h->pin);
      if ((rc = host_ladder_run(h, params, rho_old, rho, rho_old_out, nfact, success, pb2, reinterpret_cast<int32_t*>(pb2 + B * 12)))) return rc;
      laddered = true;
    }
  }
  {
    size_t b0 = 0;
    while (b0 < B) {  // maximal runs of successful problems: one copy in the common case
      while (b0 < B && !success[b0]) b0++;
      size_t b1 = b0;
      while (b1 < B && success[b1]) b1++;
      if (b1 > b0) HIPCHK(hipMemcpyAsync(d + b0 * P.N, h->d_d + b0 * P.N, (b1 - b0) * P.N * sizeof(double), hipMemcpyDeviceToHost, h->stream));
      b0 = b1;
    }
  }
  if (!laddered) {
    HIPCHK(hipMemcpyAsync(rho, h->d_rho, B * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(rho_old_out, h->d_rho_old, B * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(nfact, h->d_nfact, B * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
  }
  // rho tail of vals (the reference mutates get_vals(LDLT)[end-nvar+1:end], src/CaNNOLeS.jl:1031,1038)
  if (P.nvar > 0)
    HIPCHK(hipMemcpy2DAsync(vals + (P.nnz - P.nvar), (size_t)P.nnz * sizeof(double), h->d_vals + (P.nnz - P.nvar),
                            (size_t)P.nnz * sizeof(double), (size_t)P.nvar * sizeof(double), B, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  h->factorized = true;
    h->last_ok.assign(success, success + B);
  return CNL_OK;
}


// ---- one caller, several devices: contiguous balanced shards of the batch, one handle + one host thread per
//      device, no collective — the devices never exchange data --------------------------------------------------------------
int cnl_multi_create(cnl_multi** mout, int64_t N, int64_t nnz, const int64_t* rows1, const int64_t* cols1, int64_t nvar, int64_t nequ,
                     int64_t ncon, int64_t batch, const int* devices, int ndev) {
  return cnl_multi_create_ex(mout, N, nnz, rows1, cols1, nvar, nequ, ncon, batch, devices, ndev, nullptr);
}

int cnl_multi_create_ex(cnl_multi** mout, int64_t N, int64_t nnz, const int64_t* rows1, const int64_t* cols1, int64_t nvar, int64_t nequ,
                        int64_t ncon, int64_t batch, const int* devices, int ndev, const cnl_options* opt) {
  if (!mout || !devices) return fail(CNL_ERR_ARG, "null argument");
  *mout = nullptr;
  if (ndev < 1 || ndev > 64 || batch < 1) return fail(CNL_ERR_ARG, "need 1 <= ndev <= 64 and batch >= 1");
  cnl::Tuning o;
  int rc = resolve_options(opt, o);
  if (rc) return rc;
  int navail = 0;
  if (hipGetDeviceCount(&navail) != hipSuccess || navail == 0) return fail(CNL_ERR_HIP, "no HIP device available (this backend has no CPU fallback)");
  for (int i = 0; i < ndev; i++) if (devices[i] < 0 || devices[i] >= navail) return fail(CNL_ERR_ARG, "device index out of range");
  cnl_multi* m = new cnl_multi();
  m->N = N; m->nnz = nnz; m->batch = batch;
  const int64_t base = batch / ndev, rem = batch % ndev;
  for (int i = 0; i < ndev; i++) {
    const int64_t cnt = base + (i < rem ? 1 : 0);
    if (cnt == 0) continue;  // more devices than problems: the surplus devices stay idle
    m->start.push_back(i * base + std::min<int64_t>(i, rem));
    m->count.push_back(cnt);
    m->device.push_back(devices[i]);
  }
  // The symbolic analysis runs ONCE per shard size ("done once on host and broadcast"): the plan is reference-counted,
  // every shard's handle uploads its own copy of the index data to its device.  Shard sizes differ by at most one problem: one
  // analysis when the batch divides evenly, two otherwise.
  // (round 4: one analysis per DISTINCT shard size — at most two, the sizes differ by at most one problem — so that every shard
  //  runs exactly the plan cnl_create would pick for its own batch: with the first shard's plan for all, a shard on the other
  //  side of a planning boundary (latency / throughput, split eligibility) got its neighbour's plan)
  cnl_plan* shared[2] = {nullptr, nullptr};
  int64_t shared_count[2] = {-1, -1};
  for (size_t i = 0; i < m->count.size() && !rc; i++) {
    cnl_plan* plan = nullptr;
    if (o.multi_share_plan) {
      const int k = shared_count[0] == m->count[i] ? 0 : (shared_count[1] == m->count[i] ? 1 : (shared_count[0] < 0 ? 0 : 1));
      if (shared_count[k] != m->count[i]) {
        if (shared[k]) { cnl_plan_destroy(shared[k]); shared[k] = nullptr; }   // (a third size: cannot happen with balanced shards)
        shared_count[k] = m->count[i];
      }
      if (!shared[k]) rc = plan_create_tuned(&shared[k], N, nnz, rows1, cols1, nvar, nequ, ncon, m->count[i], o);
      if (!rc) { plan = shared[k]; plan->refs.fetch_add(1); }
    } else {
      rc = plan_create_tuned(&plan, N, nnz, rows1, cols1, nvar, nequ, ncon, m->count[i], o);
    }
    cnl_handle* h = nullptr;
    if (!rc) rc = create_from_plan(&h, plan, rows1, cols1, m->count[i], m->device[i]);  // owns one reference, also when it fails
    if (rc) {
      const std::string keep = g_err;
      for (cnl_plan* p : shared) cnl_plan_destroy(p);
      cnl_multi_destroy(m);
      return fail(rc, "shard " + std::to_string(i) + ": " + keep);
    }
    m->h.push_back(h);
  }
  for (cnl_plan* p : shared) cnl_plan_destroy(p);  // the handles keep theirs
  m->rc.assign(m->h.size(), CNL_OK);
  m->msg.assign(m->h.size(), std::string());
  for (size_t i = 0; i < m->h.size(); i++) m->workers.emplace_back(multi_worker, m, i);
  *mout = m;
  return CNL_OK;
}

int cnl_multi_destroy(cnl_multi* m) {
  if (!m) return CNL_OK;
  {
    std::lock_guard<std::mutex> lk(m->mu);
    m->stop = true;
    m->cv_job.notify_all();
  }
  for (auto& t : m->workers) t.join();
  for (cnl_handle* h : m->h) cnl_destroy(h);
  delete m;
  return CNL_OK;
}

int cnl_multi_shards(const cnl_multi* m, int64_t* nshards, int64_t* start, int64_t* count, int32_t* device) {
  if (!m || !nshards) return fail(CNL_ERR_ARG, "null argument");
  *nshards = (int64_t)m->h.size();
  for (size_t i = 0; i < m->h.size(); i++) {
    if (start) start[i] = m->start[i];
    if (count) count[i] = m->count[i];
    if (device) device[i] = m->device[i];
  }
  return CNL_OK;
}

int cnl_multi_factorize(cnl_multi* m, const double* vals, double eig_tol, int32_t* success, int64_t* npos, int64_t* nzero) {
  if (!m || !vals || !success) return fail(CNL_ERR_ARG, "null argument");
  return multi_run(m, [&](size_t i) {
    const int64_t s = m->start[i];
    return cnl_factorize(m->h[i], vals + s * m->nnz, eig_tol, success + s, npos ? npos + s : nullptr, nzero ? nzero + s : nullptr);
  });
}

int cnl_multi_solve(cnl_multi* m, const double* rhs, double* d) {
  if (!m || !rhs || !d) return fail(CNL_ERR_ARG, "null argument");
  return multi_run(m, [&](size_t i) {
    const int64_t s = m->start[i];
    return cnl_solve(m->h[i], rhs + s * m->N, d + s * m->N);
  });
}

int cnl_multi_newton_system(cnl_multi* m, double* vals, const double* rhs, double* d, const double* rho_old, const double params[9],
                            double* rho, double* rho_old_out, int32_t* nfact, int32_t* success) {
  if (!m || !vals || !rhs || !d || !params || !rho || !rho_old_out || !nfact || !success) return fail(CNL_ERR_ARG, "null argument");
  return multi_run(m, [&](size_t i) {
    const int64_t s = m->start[i];
    return cnl_newton_system(m->h[i], vals + s * m->nnz, rhs + s * m->N, d + s * m->N, rho_old ? rho_old + s : nullptr, params, rho + s,
                             rho_old_out + s, nfact + s, success + s);
  });
}

// ---- device-pointer twins: shard i's arrays live on shard i's device (problem-major, count[i] problems).  The calls only
//      ENQUEUE (from the calling thread, one shard after the other) and return; cnl_multi_synchronize waits for all shards.
//      streams[i] == NULL (or streams == NULL): the shard handle's own stream.
static hipStream_t shard_stream(cnl_multi* m, size_t i, void* const* streams) {
  return streams && streams[i] ? (hipStream_t)streams[i] : m->h[i]->stream;
}

int cnl_multi_factorize_dev(cnl_multi* m, const double* const* d_vals, double eig_tol, int32_t* const* d_success, void* const* streams) {
  if (!m || !d_vals || !d_success) return fail(CNL_ERR_ARG, "null argument");
  for (size_t i = 0; i < m->h.size(); i++) {
    const int rc = cnl_factorize_dev(m->h[i], d_vals[i], eig_tol, d_success[i], shard_stream(m, i, streams));
    if (rc) return fail(rc, "shard " + std::to_string(i) + ": " + g_err);
  }
  return CNL_OK;
}

int cnl_multi_solve_dev(cnl_multi* m, const double* const* d_rhs, double* const* d_d, void* const* streams) {
  if (!m || !d_rhs || !d_d) return fail(CNL_ERR_ARG, "null argument");
  for (size_t i = 0; i < m->h.size(); i++) {
    const int rc = cnl_solve_dev(m->h[i], d_rhs[i], d_d[i], shard_stream(m, i, streams));
    if (rc) return fail(rc, "shard " + std::to_string(i) + ": " + g_err);
  }
  return CNL_OK;
}

int cnl_multi_newton_system_dev(cnl_multi* m, double* const* d_vals, const double* const* d_rhs, double* const* d_d,
                                double* const* d_rho_old, double* const* d_rho, int32_t* const* d_nfact, int32_t* const* d_success,
                                const double params[9], void* const* streams) {
  if (!m || !d_vals || !d_rhs || !d_d || !d_rho_old || !d_rho || !d_nfact || !d_success || !params) return fail(CNL_ERR_ARG, "null argument");
  for (size_t i = 0; i < m->h.size(); i++) {
    const int rc = cnl_newton_system_dev(m->h[i], d_vals[i], d_rhs[i], d_d[i], d_rho_old[i], d_rho[i], d_nfact[i], d_success[i], params,
                                         shard_stream(m, i, streams));
    if (rc) return fail(rc, "shard " + std::to_string(i) + ": " + g_err);
  }
  return CNL_OK;
}

int cnl_multi_synchronize(cnl_multi* m, void* const* streams) {
  if (!m) return fail(CNL_ERR_ARG, "null argument");
  for (size_t i = 0; i < m->h.size(); i++) {
    HIPCHK(hipSetDevice(m->device[i]));
    HIPCHK(hipStreamSynchronize(shard_stream(m, i, streams)));
  }
  return CNL_OK;
}

}  // extern "C"
