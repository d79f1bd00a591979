// band.h — "band program": the Newton system of a band-structured problem as a sliding-window elimination in which ONE LANE
// serves one (problem, part) pair (kernels: band.hip).
//
// The reference factorises K = [H + rho I, J', Jc'; J, -I, 0; Jc, 0, -dI] (/root/reference/src/CaNNOLeS.jl:282) with a general
// sparse LDL' (src/solver_types.jl:79-98).  When the variables in their natural order give a band — every residual row and every
// Hessian entry spans at most BAND_HW + 1 consecutive variables, every constraint row covers a contiguous run — the LDL' of the
// permuted K (residual nodes first, then the variables in order, every multiplier right behind the last variable it touches) is a
// fixed-shape recurrence over a window of BAND_HW + 1 variables, one live multiplier and the right-hand side: 27 doubles of
// state.  That state fits the registers of ONE lane, so the elimination needs no cross-lane traffic at all: a wavefront runs up
// to 64 independent recurrences with plain fp64 FMAs, and what remains of the problem is data movement — the caller's arrays
// are problem-major (vals[b][nnz], include/cannoles_hip.h), so every operand stream is loaded in 64-byte pieces (8 lanes x 8
// bytes per problem) and handed to its lane through LDS.
//
// The chain is cut in TWO parts that run on the two wavefronts of a workgroup: part 0 eliminates variables 0 .. m0-1 upwards,
// part 1 eliminates n-1 .. m0+4 downwards; the four variables in between are the junction, a dense 4 x 4 system the first
// wavefront finishes.  Any symmetric permutation has the inertia of K, so success / inertia / rho-ladder decisions are those of
// the reference (src/solver_types.jl:90-97, src/CaNNOLeS.jl:1023-1047).
//
// This file: the host-side generator (band.cpp) that checks the structure and writes the program, and the program's format.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace cnl {

constexpr int BAND_HW = 4;          // variables a pivot is coupled to inside the band (window = BAND_HW + 1 band slots)
constexpr int BAND_NB = BAND_HW + 1;
constexpr int BAND_EPOCH = 8;       // steps per epoch (one round of operand pieces) = window slots: variable number t of a part lives in
                                    // slot t % 8 (five consecutive slots are live at a time), so step u of an epoch always works on the same slots
constexpr int BAND_NS = BAND_EPOCH;
constexpr int BAND_NPIECE = 15;     // 64-byte operand pieces per epoch and lane (BASELINE config 3 needs 15 in the forward sweep)
constexpr int BAND_REC_MAX = 256;    // ints of step + row blocks per epoch (LDS record buffer of a wavefront)
constexpr int BAND_LREC = 6;        // factor doubles per pivot: band multipliers, border multiplier, z  (see band.hip)

// LDS block of one lane, in doubles: [operand pieces | out ring | zero cell].  The out ring holds the factor records of HALF an
// epoch in the forward sweep (flushed behind steps 3 and 7) and the solution components of an epoch in the backward sweep.
// 153 doubles = 1224 bytes per lane: 128 lanes (two workgroups of 32 problems) fit the 160 KB of a CU.
constexpr int BAND_IN_OFF = 0;
constexpr int BAND_LOUT_OFF = BAND_NPIECE * 8;                       // forward: factor records of half an epoch
constexpr int BAND_LOUT_MAX = 32;
constexpr int BAND_DX_OFF = BAND_LOUT_OFF;                           // backward: solution components of the epoch's pivots
constexpr int BAND_DX_MAX = 8;
constexpr int BAND_DR_OFF = BAND_DX_OFF + BAND_DX_MAX;               // backward: residual components of the epoch's rows
constexpr int BAND_DR_MAX = 24;
constexpr int BAND_ZERO_OFF = BAND_LOUT_OFF + BAND_LOUT_MAX;         // holds 0.0: every absent operand reads it
constexpr int BAND_LANE_DOUBLES = BAND_ZERO_OFF + 1;                 // 153: odd, so that 32 lanes reading one offset hit 32 bank pairs
static_assert(BAND_DR_OFF + BAND_DR_MAX <= BAND_ZERO_OFF && BAND_LANE_DOUBLES % 2 == 1, "lane block layout");

// cnl_options.batch_layout = 1 (include/cannoles_hip.h): `vals` interleaved over groups of BAND_IL_GROUP problems in blocks of eight
// doubles — element e of problem p at ((p / 32 * band_il_blocks(nnz) + e / 8) * 32 + p % 32) * 8 + e % 8.  One spare block per problem:
// an operand piece is eight doubles from ANY element, the last one may reach into the block behind the array's last.
constexpr int BAND_IL_GROUP = 32;
constexpr long long band_il_blocks(long long len) { return (len + 7) / 8 + 1; }
constexpr long long band_il_len(long long batch, long long len) { return (batch + BAND_IL_GROUP - 1) / BAND_IL_GROUP * band_il_blocks(len) * (BAND_IL_GROUP * 8); }
constexpr long long band_il_index(long long p, long long e, long long len) {
  return ((p / BAND_IL_GROUP * band_il_blocks(len) + e / 8) * BAND_IL_GROUP + p % BAND_IL_GROUP) * 8 + e % 8;
}

// step block (BAND_SW ints); LDS offsets are BYTES inside the lane block
enum {
  BS_FLAGS = 0,        // BF_* | rows << 8
  BS_DG0, BS_DG1, BS_DG2,   // plain diagonal entries of the entering variable, in COO order
  BS_RHO,              // its rho slot (the last COO entry of that position), replaced by the ladder's rho when that is active
  BS_OD,               // BS_OD + 2 (k - 1) + dup: entries coupling the entering variable with the one entered k steps earlier
  BS_BC0 = BS_OD + 2 * BAND_HW, BS_BC1,   // entries coupling it with the live border (multiplier) row
  BS_RX,               // its right-hand-side entry
  BS_LB,               // LDS offset of the factor record of this step's border pivot (forward: out ring; backward: operand piece)
  BS_LX,               // ... of this step's band pivot
  BS_DX,               // backward: offset in the dx-out ring of the band pivot's solution component
  BS_BORDER,           // index into the border table of the border that enters / is pivoted in this step
  BS_SPARE,
  BAND_SW = 20
};
enum { BF_ENTER_B = 1, BF_PIVOT_B = 2, BF_PIVOT_X = 4, BF_ENTER_X = 8 };
// row block (BAND_RW ints) behind the step block, one per residual row completed by the step
enum { BR_DI = 0, BR_J0 /* + live position: 0 = the step's pivot .. HW = the entering variable */, BR_RR = BR_J0 + BAND_NB, BR_DR, BAND_RW = 8 };
// epoch block
enum {
  BE_FP = 0,                       // forward operand pieces: element index | array << 28 (0 vals, 1 rhs); -1 unused
  BE_BP = BE_FP + BAND_NPIECE,     // backward operand pieces (array 2: the factor)
  BE_LBASE = BE_BP + BAND_NPIECE,  // first factor double of the epoch's steps 0 .. 3 and their number, then of its steps 4 .. 7
  BE_LCNT, BE_LBASE2, BE_LCNT2,
  BE_DXLO, BE_DXCNT,               // solution components of the epoch's band pivots: d[lo .. lo + cnt)
  BE_DRLO, BE_DRCNT,               // ... of its residual rows
  BE_NSTEP,                        // steps of the epoch
  BE_FOFF,                         // first int of the epoch's step blocks in fops / bops, and their length: the blocks of an
  BE_BOFF,                         //   epoch go through an LDS record buffer of BAND_REC_MAX ints per wavefront
  BE_OPLEN,
  BAND_EW = 44
};
// border table: BAND_BW ints per border
enum { BB_DSRC = 0, BB_RHS, BB_DOUT, BB_SPARE, BAND_BW = 4 };

struct BandPart {
  int32_t nsteps = 0, nepochs = 0, npiv = 0, nevents = 0;
  std::vector<int32_t> fops, bops;   // step (+ row) blocks in forward / backward order
  std::vector<int32_t> epochs;       // BAND_EW ints per epoch
  std::vector<int32_t> borders;      // BAND_BW ints per border
  int64_t loff = 0;                  // first factor double of the part inside a problem's factor storage
};

struct BandPlan {
  bool ok = false;
  std::string why;            // why not, when !ok
  int32_t nparts = 0;
  int32_t m0 = 0;             // part 0 pivots variables [0, m0), part 1 pivots [m0 + BAND_HW, n) downwards (nparts == 2)
  int32_t n = 0, N = 0, nnz = 0;
  int64_t lsize = 0;          // factor doubles per problem
  BandPart part[2];
};

// rows1/cols1: the reference's 1-based COO pattern (src/CaNNOLeS.jl:256-315).  Fills B (B.ok, B.why); nparts_wanted: 1 or 2.
void build_band_plan(BandPlan& B, int64_t N, int64_t nnz, const int64_t* rows1, const int64_t* cols1, int64_t nvar, int64_t nequ,
                     int64_t ncon, int nparts_wanted);

}  // namespace cnl
