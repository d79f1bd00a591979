// plan.h — host-side symbolic analysis of the CaNNOLeS Newton (KKT) system.
//
// Replaces what the reference gets from `ldl_analyze` (AMD ordering + etree +
// column counts; /root/reference/src/solver_types.jl:61-65) with a static
// multifrontal "plan" shared by every problem of a batch: elimination order,
// supernodes (fronts), assembly / extend-add index maps and the LDS layout of
// the update-matrix stack.  The plan is pure index data; the HIP kernels in
// kernels.hip execute it.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace cnl {

// One front (supernode), 16 x int32 = 64 bytes so a wave fetches it with
// scalar loads.
struct FrontHdr {
  int32_t npiv;       // pivots eliminated in this front
  int32_t nupd;       // update rows (excluding the rhs row)
  int32_t foff;       // offset (doubles) of the packed front in the work stack
  int32_t ubase;      // offset where the update matrix is left for the parent
  int32_t seg_begin;  // assembly rounds [seg_begin, seg_end) into seg_ptr
  int32_t seg_end;
  int32_t child_begin;// children [child_begin, child_end) into child_idx
  int32_t child_end;
  int32_t rel_begin;  // rel map of THIS front into its parent: rel_idx[rel_begin + i], i in [0, 1+nupd)
  int32_t first_piv;  // elimination index of the first pivot of the front
  int32_t xoff;       // offset (doubles) of the front's solution vector in the backward stack
  int32_t parent;     // parent front or -1
  int32_t lptr_lo;    // offset (doubles) of the L panel in the per-problem factor storage
  int32_t lptr_hi;    //   (lptr = lptr_lo | lptr_hi << 31)
  int32_t indep;      // leading `indep` pivots (first eliminated) are mutually independent
  int32_t pad;
};

struct Options {
  int order_mode = -1;     // -1 auto, 0 canonical (r, x natural, lambda), 1 ND, 2 MD
  int nd_leaf = 0;         // 0 = sweep
  int relax = -1;          // relaxed-amalgamation budget (extra zeros per merged column); -1 = default
  int lds_budget_doubles = 0; // 0 = default
  int with_rhs_row = 1;
  // staged execution for small batches (kernels2.hip, STAGED): choose the order by the critical path of the elimination
  // tree and cut the tree into tasks that run on different wavefronts.  par = wavefront slots available per group of four
  // problems (the throughput term of the score), task_cap = largest number of fronts of a bottom task.
  int latency = 0;
  int par = 256;
  double slots = 0;  // the same as a fraction (wavefront slots per group of four problems; 0: use par)
  int task_cap = 0;
  // the rest mirrors cnl_options (include/cannoles_hip.h): explicit switches instead of environment variables, so that the
  // numerical path of a handle depends on its arguments only
  int early = 1;           // consider the "+early" candidates (a multiplier right behind the last variable it touches)
  int register_front = 1;  // allow the register-front kernel (record streams) when every front has order <= 64
  int ubig = 17;           // update matrices of order above this live in the global scratch
  int wait_thr = 2;        // update matrices that wait for more than this many fronts go to the global scratch
  int verbose = 0;
  int threads = 0;         // host threads for the candidate orders (0: one per candidate, at most 16 / the hardware's); the plan does not depend on it
  std::string force_order; // pick an ordering candidate by name (experiments)
  // elimination positions (of the chosen order) at which a supernode must be cut in two: the caller found a front whose condensed
  // residual rows do not fit the row form's sixteen lanes (Plan::rows_overflow) and asks for the same plan with that front split
  std::vector<int32_t> split_positions;
};

// One task of the staged execution: fronts [f0, f1) (a complete subtree, or a single front above the cut) processed by
// one wavefront; tasks of a stage are independent, a task's children belong to earlier stages.
struct Task {
  int32_t stage, f0, f1, rec_off, brec_off, is_root;
  int32_t parent, nchild;  // task of the parent front of this task's root (-1: none); number of tasks whose parent this is
};

struct Plan {
  int64_t N = 0, nnz = 0, nvar = 0, nequ = 0, ncon = 0;
  int64_t nnzK = 0;          // unique lower-triangular entries
  // ordering: perm[k] = original 0-based index eliminated k-th
  std::vector<int32_t> perm, iperm;
  // fronts in post-order
  int32_t nsuper = 0;
  std::vector<FrontHdr> fronts;
  // assembly: front s owns rounds seg_begin..seg_end; round r = entries
  // [seg_ptr[r], seg_ptr[r+1]) of (asm_pos, asm_src).  Within a round all
  // positions are distinct; rounds are applied in order, which reproduces the
  // COO-order summation of duplicates of set_vals! (solver_types.jl:53-59).
  // asm_src < nnz: COO entry;  asm_src >= nnz: rhs[asm_src - nnz].
  std::vector<int32_t> seg_ptr, asm_pos, asm_src;
  std::vector<int32_t> child_idx;
  std::vector<int32_t> rel_idx;
  // universal triangular decode: tri_row[t] = i with i(i+1)/2 <= t < (i+1)(i+2)/2
  int32_t fmax = 0;          // largest front order (1 + nupd + npiv)
  int64_t lsize = 0;         // doubles of factor storage per problem
  int64_t nnzL = 0;          // strictly-lower entries of L incl. explicit zeros of relaxed supernodes (no rhs row, no diagonal)
  int64_t nnzL_exact = 0;    // fill of the ordering without relaxation
  double flops = 0;          // FMAs of the numeric phase as executed (incl. rhs row)
  int32_t fwd_peak = 0;      // doubles of work stack needed by the forward (factor) pass
  int32_t bwd_peak = 0;      // doubles needed by the backward pass
  int32_t panel_max = 0;     // largest L panel (doubles)
  int32_t rho_begin = 0;     // COO entries >= rho_begin are the rho slots (nnz - nvar)
  std::string order_name;
  double cost = 0;           // model cost used to choose the ordering

  // ---- v2 ("register front") streams: valid when every front has order <= 64 ----
  // Forward records (post-order) and backward records (reverse post-order) are
  // self-describing word streams read sequentially by a wavefront; layouts in
  // kernels2.hip (R_* / B_* constants).
  bool v2_ok = false;
  std::vector<int32_t> rec, brec;
  int32_t rec_maxlen = 0, brec_maxlen = 0;  // words
  int32_t u2_peak = 0;       // doubles of LDS update-matrix stack per problem
  int32_t fs2_max = 0;       // doubles of LDS front staging per problem (fronts of order <= 32)
  int64_t gs_doubles = 0;    // doubles of global scratch per problem (large fronts / update matrices)
  int32_t ncls[3] = {0, 0, 0};  // fronts per class (order <=16, <=32, <=64)
  std::vector<int32_t> v2_cls, v2_fsglob, v2_uglob, v2_uoff, v2_fsoff;  // per front (kept so that the records can be rewritten)
  // "direct" records: the assembly lists address the ORIGINAL vals / rhs (duplicate rounds) and carry the
  // products of the condensed residual rows, so the multifrontal kernel condenses on the fly
  bool rec_direct = false;
  int64_t d_owned = 0;   // number of condensed residual pivots staged (and counted) by the direct records
  bool row_products = true;   // direct records of fast fronts may use the row form (RF_ROWS)
  int32_t row_min_products = 72;  // ... from this many products on (analysis.cpp); 1 when that makes the whole plan row-form (lean kernel)
  int32_t rows_fronts = 0, listprod_fronts = 0;  // fast fronts in row form / with product lists
  bool band_form = true;     // (cnl_options.band_form) let the records carry band forms
  int32_t band_fronts = 0;   // fast fronts whose records carry a band form (R_FSOFF of a fast front: nfix | hw << 8, 0 = none):
                             // every pivot row is structurally zero outside its nfix lowest columns (right-hand side,
                             // multipliers) and the hw columns right below the pivot; the kernel's elimination then omits the
                             // other row updates (kernels2.hip, eliminate16_dpp<LATE, BNF>; analysis.cpp, write_forward_records)
  std::vector<int32_t> rows_overflow;  // fronts refused by the row form for MORE THAN 16 rows: the elimination position in their middle
  bool back_rows = false;  // the backward records recover the condensed residual components themselves (write_backward_rows)
  bool d_outer = false;  // backward records name solution components in the caller's numbering (set with rec_direct)
  int32_t nnz_outer = 0, n_outer = 0;  // outer (reference) nnz and N when rec_direct
  // staged execution (empty: the plan runs as one sequential stream per group of four problems)
  std::vector<Task> tasks;          // sorted by stage
  std::vector<int32_t> stage_ptr;   // tasks of stage s: [stage_ptr[s], stage_ptr[s + 1])
  double cpath = 0;                 // model cost of the critical path of the chosen order
};
// offsets of the tasks inside the record streams (call again after the forward records are rewritten)
void finalize_tasks(Plan& P);

// contributions of every source of the condensed system in terms of the original arrays (see condense.h)
struct DirectLists {
  const int32_t *c_ptr, *c_a, *c_b, *c_d;
  int32_t nnz_outer, n_outer;
};
// (Re)writes P.rec.  D == nullptr: sources are the condensed buffer's slots.  Returns 0, or 1 when the plan
// cannot be expressed with direct records (then P.rec is left in the indirect form).
int write_forward_records(Plan& P, const DirectLists* D);
// Appends the backward-rows sections to P.brec (see B_ROWS_FLAG) from the row-form forward records and the condensed rows'
// Jacobian lists (condense.h: r_orig, r_dsrc, r_ptr, r_jsrc, r_jx; ncond rows).  Returns 0 and sets P.back_rows, or 1 (P.brec
// untouched) when some front is not in row form or some row is not covered.
struct BackRowsIn { const int32_t *r_orig, *r_dsrc, *r_ptr, *r_jsrc, *r_jx; int32_t ncond; };
int write_backward_rows(Plan& P, const BackRowsIn& in);

// record layouts shared by analysis.cpp (writer) and kernels2.hip (reader)
enum {
  R_NPIV = 0, R_NUPD, R_RECLEN, R_NASM, R_NCHILD, R_UOFF, R_FLAGS, R_FSOFF, R_LPTR_LO, R_LPTR_HI, R_NASMV,
  R_ASM_OFF, R_CHILD_OFF, R_NPROD, R_NRAW, R_NRD, R_HDR = 16
};
enum { RF_U_GLOBAL = 1, RF_FS_GLOBAL = 2, RF_ROWS = 4 };
// RF_ROWS (fast fronts with direct records): the products of the on-the-fly condensation are organised PER RESIDUAL ROW instead
// of as a list of (position, three raw-value indices) words.  Lane l of a problem takes residual row l of the front (at most 16):
// it gathers the row's pivot d_r, up to ROWS_KM Jacobian entries and its right-hand-side entry into registers, forms
// w = -1/d_r once and adds  (J_p w) J_q  to the position of pair (p, q) — no LDS staging of raw values, no index decode.
// The section replaces the raw-value list of the record (R_NRAW = ROWS_WORDS words, R_NPROD = 0 products):
//   [16 pivot sources][ROWS_KM x 16 Jacobian sources][16 right-hand-side sources (>= nnz)]
//   [ROWS_PW x 16 position words: byte k & 3 of word k >> 2 = image position of pair k]
// pair k: (p, q), q <= p < ROWS_KM at p (p + 1) / 2 + q, then (rhs, q) at ROWS_KM (ROWS_KM + 1) / 2 + q.  Absent operands read
// a valid dummy source and their pairs go to the padding slots FAST_IMG_TRI + lane.  Rows whose pivot this front owns come first
// (their number is the high half of R_NPROD, as before); R_NRD holds the number of rows.
enum { ROWS_KM = 5, ROWS_NPAIR = ROWS_KM * (ROWS_KM + 1) / 2 + ROWS_KM, ROWS_PW = (ROWS_NPAIR + 3) / 4,
       ROWS_WORDS = 16 * (1 + ROWS_KM + 1 + ROWS_PW) };
enum { C_UOFF = 0, C_TUC, C_FLAGS, C_PAD, C_HDR = 4 };
// LDS image of a fast front (order <= 16): the packed lower triangle, (a, b) -> a(a+1)/2 + b, 136 doubles, followed by 16
// slots that only the padding entries of the assembly lists add to.  (Round 1 used a 16 x 16 strided image; the triangle
// brings the LDS need of a wavefront from 16.8 to 12.7 KB: twelve wavefronts per CU instead of nine.)
enum { FAST_IMG_TRI = 136, FAST_IMG_DOUBLES = 152 };
enum { B_NPIV = 0, B_NUPD, B_RECLEN, B_XOFF, B_PXOFF, B_LPTR_LO, B_LPTR_HI, B_CLS, B_HDR = 8 };
// Backward rows (write_backward_rows): the residual components of the condensed rows a front OWNS are recovered inside the
// backward sweep, right behind the front's own solution components — every column of such a row belongs to the front (the row
// is a clique whose first column is a pivot here), so  d_r = (sum_p J_p x[l_p] - rhs_r) / d_r  needs nothing but the front's x.
// B_CLS then carries  cls | B_ROWS_FLAG | rows << 16  and the record ends with a section of BROWS_WORDS words at
// (B_HDR + f + 3) & ~3:  [16 pivot sources][ROWS_KM x 16 Jacobian sources][16 right-hand-side sources (>= nnz)][16 index words:
// nibble p = local index of operand p's column, bits 20..22 = number of operands, bit 23 = the lane holds a row].
enum { B_ROWS_FLAG = 256, BROWS_WORDS = 16 * (ROWS_KM + 3) };
enum { B_PX_NONE = -1, B_PX_GLOBAL = -2 };  // B_PXOFF: no parent / parent solved by another task: the update rows name solution components

// Builds the plan.  rows1/cols1: 1-based COO of the lower triangle, duplicates
// allowed (summed).  Returns 0 or an error code (see cannoles_hip.h); msg gets
// a description.
int build_plan(Plan& P, int64_t N, int64_t nnz, const int64_t* rows1, const int64_t* cols1,
               int64_t nvar, int64_t nequ, int64_t ncon, const Options& opt, std::string& msg);

}  // namespace cnl
