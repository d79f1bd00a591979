// dense.h — backend for plans whose residual block is DENSE (BASELINE config 2: every residual row holds all variables).
// The multifrontal machinery has nothing to exploit there: the condensed system
//     S = [ H + rho I - J' diag(1/d_r) J    Jc' ]      (order n + p, dense; top-left = H + rho I + J'J for the
//         [ Jc                             -dI  ]       reference's d_r = -1; quasi-definite, so no pivoting is needed)
// is formed and factorised in 64 x 64 tiles by hand-written kernels (J'WJ and the trailing updates on the fp64 matrix
// cores, v_mfma_f64_16x16x4_f64; panel steps in registers), a blocked dense LDL^T without pivoting.  Same contract as the other kernels: inertia rule of
// /root/reference/src/solver_types.jl:90-97, rho ladder of /root/reference/src/CaNNOLeS.jl:1008-1052, d = -K^-1 rhs.
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstdint>
#include <string>
#include <vector>

namespace cnl {

struct DensePlan {
  bool active = false;
  int32_t n = 0, m = 0, p = 0, nnz = 0;  // variables, residual rows, constraints, COO entries; the dense system has order n + p
  // slot lists built from the pattern (0-based COO entry numbers)
  std::vector<int32_t> jslot;     // [m * n] column-major: entry of J(i, j) at i + m * j
  std::vector<int32_t> dslot;     // [m]  diagonal of the -I block
  std::vector<int32_t> hslot, hpos;  // every other entry (H_F, H_c, J_c, -delta I; duplicates allowed): slot, position i' + (n + p) * j'
};

// Returns true and fills D when the pattern qualifies: nequ > 0, every residual row has exactly one entry per variable and
// one diagonal entry, and the constraint rows (any pattern) couple only to the variables.
bool detect_dense(DensePlan& D, int64_t N, int64_t nnz, const int64_t* rows1, const int64_t* cols1, int64_t nvar, int64_t nequ,
                  int64_t ncon);

struct DenseState;  // device buffers
int dense_create(DenseState** st, const DensePlan& D, int64_t batch, std::string& err, bool use_graph = true, int syrk_wgs = 0, int panel_blocks = 1);
void dense_destroy(DenseState* st);

// mode: 0 newton (ladder + solve), 1 factorize, 2 solve.  All pointers are device pointers, problem-major as in the ABI.
// Asynchronous on `stream`: the rho ladder is decided on the device.
int dense_run(DenseState* st, const DensePlan& D, int mode, double* vals, const double* rhs, double* d, double* rho_old,
              double* rho, int32_t* nfact, int32_t* success, int64_t* npos, int64_t* nzero, const double params[9],
              hipStream_t stream, std::string& err);

// ---- the same dense machinery for an ARBITRARY condensed system (csrc/condense.h) of moderate order: plans whose fill makes
// the fronts larger than the register-front kernel takes (irregular sparsity) and whose batch is small.  The caller (capi.cpp)
// owns the condensed buffer; per problem it hands over the slots of K2, and gets the factor / solution back.
struct GeneralOps {
  int32_t ns = 0, nv = 0;          // order of the condensed system, number of variables (they carry rho)
  int32_t nslots = 0;              // unique lower-triangular slots of K2 (without the rho entries)
  const int32_t* d_pos = nullptr;  // device: position i + ns * j of every slot
  int64_t cstride = 0;             // doubles per problem of the condensed buffer [slots | rho (nv) | rhs (ns)]
};
// d_pos (device, owned by the caller): position i + ns * j of every slot of the condensed system
int dense_create_general(DenseState** st, int32_t ns, int32_t nv, int32_t nslots, const int32_t* d_pos, int64_t batch, std::string& err,
                         bool use_graph = true, int panel_blocks = 1);
// cbuf: condensed buffer of the batch (matrix part filled; rhs part filled for mode 0 / 2); xpos / xzer: inertia counts of the
// condensed residual pivots per problem (device ints); d2: [batch][ns] receives -x (reduced numbering); rho_fill: per problem
// pointer stride info to write the last rho tried back into the caller's vals (vals + b * nnz + rho_begin, nv entries).
int dense_run_general(DenseState* st, const GeneralOps& G, int mode, const double* cbuf, const int* xpos, const int* xzer, double* d2,
                      double* vals_rho0, int64_t vals_stride, double* rho_old, double* rho, int32_t* nfact, int32_t* success,
                      int64_t* npos, int64_t* nzero, const double params[9], hipStream_t stream, std::string& err);

}  // namespace cnl
