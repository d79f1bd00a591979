// condense.cpp — builds the static condensation of the residual (-I) block (see condense.h).
#include "condense.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <numeric>

namespace cnl {

int build_condensation(Cond& C, int64_t N, int64_t nnz, const int64_t* rows1, const int64_t* cols1, int64_t nvar,
                       int64_t nequ, int64_t ncon, std::string& msg, bool enable) {
  C = Cond();
  C.N = N; C.nnz = nnz; C.nvar = nvar; C.nequ = nequ; C.ncon = ncon;
  if (!enable) return 0;
  if (N <= 0 || nvar < 0 || nequ < 0 || ncon < 0 || nvar + nequ + ncon != N || nnz < nvar) return 0;  // build_plan reports it
  if (nequ == 0) return 0;
  if (N + nnz >= ((int64_t)1 << 29)) return 0;
  // the last nvar entries must be the rho slots (i, i), i = 1..nvar (src/CaNNOLeS.jl:313-315)
  for (int64_t k = 0; k < nvar; k++)
    if (rows1[nnz - nvar + k] != k + 1 || cols1[nnz - nvar + k] != k + 1) return 0;
  const int64_t nbody = nnz - nvar;
  // duplicates per slot
  std::vector<int64_t> key(nnz);
  for (int64_t k = 0; k < nnz; k++) {
    int64_t i = rows1[k] - 1, j = cols1[k] - 1;
    if (i < 0 || i >= N || j < 0 || j >= N || i < j) return 0;  // malformed: let build_plan produce the error
    key[k] = j * N + i;
  }
  std::vector<int64_t> order(nnz);
  std::iota(order.begin(), order.end(), 0);
  std::sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return key[a] < key[b] || (key[a] == key[b] && a < b); });
  std::vector<char> dup(nnz, 0);
  for (int64_t t = 1; t < nnz; t++)
    if (key[order[t]] == key[order[t - 1]]) dup[order[t]] = dup[order[t - 1]] = 1;
  // rows of the residual nodes
  const int32_t dense_thr = std::max(16, (int32_t)(10.0 * std::sqrt((double)N)));
  std::vector<std::vector<int32_t>> rowent(nequ);  // COO entries of row r (J entries)
  std::vector<int32_t> diag(nequ, -1);
  std::vector<char> bad(nequ, 0);
  for (int64_t k = 0; k < nbody; k++) {
    int64_t i = rows1[k] - 1, j = cols1[k] - 1;
    if (i >= nvar && i < nvar + nequ) {
      int64_t r = i - nvar;
      if (j == i) { if (diag[r] >= 0 || dup[k]) bad[r] = 1; diag[r] = (int32_t)k; }
      else if (j < nvar) { if (dup[k]) bad[r] = 1; rowent[r].push_back((int32_t)k); }
      else bad[r] = 1;
    }
    if (j >= nvar && j < nvar + nequ && i != j) bad[j - nvar] = 1;  // something below the diagonal in column r
  }
  C.red_of.assign(N, -1);
  int64_t ncond = 0;
  for (int64_t r = 0; r < nequ; r++) {
    if (diag[r] < 0 || (int32_t)rowent[r].size() > dense_thr) bad[r] = 1;
    if (!bad[r]) ncond++;
  }
  if (ncond == 0) return 0;
  // reduced numbering: x, kept residual nodes, multipliers
  int64_t nxt = 0;
  for (int64_t v = 0; v < nvar; v++) C.red_of[v] = (int32_t)nxt++;
  for (int64_t r = 0; r < nequ; r++) if (bad[r]) C.red_of[nvar + r] = (int32_t)nxt++;
  C.nequ2 = nxt - nvar;
  for (int64_t v = nvar + nequ; v < N; v++) C.red_of[v] = (int32_t)nxt++;
  C.N2 = nxt;
  C.orig_of.assign(C.N2, -1);
  for (int64_t v = 0; v < N; v++) if (C.red_of[v] >= 0) C.orig_of[C.red_of[v]] = (int32_t)v;
  // contributions to the slots of K2
  struct Ct { int64_t key; int64_t ord; int32_t a, b, d; };
  std::vector<Ct> cts;
  cts.reserve(nbody + ncond * 16);
  for (int64_t k = 0; k < nbody; k++) {
    int64_t i = rows1[k] - 1, j = cols1[k] - 1;
    int32_t i2 = C.red_of[i], j2 = C.red_of[j];
    if (i2 < 0 || j2 < 0) continue;  // belongs to a condensed row
    cts.push_back({(int64_t)j2 * C.N2 + i2, k, (int32_t)k, -1, -1});
  }
  int64_t pord = nnz;
  for (int64_t r = 0; r < nequ; r++) {
    if (bad[r]) continue;
    C.r_orig.push_back((int32_t)(nvar + r));
    C.r_dsrc.push_back(diag[r]);
    C.r_ptr.push_back((int32_t)C.r_jsrc.size());
    const auto& re = rowent[r];
    for (int32_t k : re) { C.r_jsrc.push_back(k); C.r_jx.push_back((int32_t)(cols1[k] - 1)); }
    for (size_t p = 0; p < re.size(); p++)
      for (size_t q = p; q < re.size(); q++) {
        int64_t xa = cols1[re[p]] - 1, xb = cols1[re[q]] - 1;
        int64_t hi = std::max(xa, xb), lo = std::min(xa, xb);
        cts.push_back({lo * C.N2 + hi, pord++, re[p], re[q], diag[r]});
      }
  }
  C.r_ptr.push_back((int32_t)C.r_jsrc.size());
  std::sort(cts.begin(), cts.end(), [](const Ct& a, const Ct& b) { return a.key < b.key || (a.key == b.key && a.ord < b.ord); });
  C.c_ptr.clear(); C.c_a.clear(); C.c_b.clear(); C.c_d.clear();
  C.rows2.clear(); C.cols2.clear();
  for (size_t t = 0; t < cts.size(); t++) {
    if (t == 0 || cts[t].key != cts[t - 1].key) {
      C.c_ptr.push_back((int32_t)C.c_a.size());
      C.rows2.push_back(cts[t].key % C.N2 + 1);
      C.cols2.push_back(cts[t].key / C.N2 + 1);
    }
    C.c_a.push_back(cts[t].a); C.c_b.push_back(cts[t].b); C.c_d.push_back(cts[t].d);
  }
  C.ncs = (int64_t)C.rows2.size();
  // rho slots
  for (int64_t k = 0; k < nvar; k++) {
    C.c_ptr.push_back((int32_t)C.c_a.size());
    C.c_a.push_back((int32_t)(nnz - nvar + k)); C.c_b.push_back(-1); C.c_d.push_back(-1);
    C.rows2.push_back(k + 1); C.cols2.push_back(k + 1);
  }
  // right-hand side of the condensed system: rhs_v - sum_r J_rv rhs_r / d_r
  std::vector<std::vector<int32_t>> xr(nvar);  // per x: indices into r_jsrc
  for (size_t q = 0; q < C.r_jsrc.size(); q++) xr[C.r_jx[q]].push_back((int32_t)q);
  std::vector<int32_t> rowof(C.r_jsrc.size());
  for (size_t rr = 0; rr + 1 < C.r_ptr.size(); rr++)
    for (int32_t q = C.r_ptr[rr]; q < C.r_ptr[rr + 1]; q++) rowof[q] = (int32_t)rr;
  for (int64_t v = 0; v < C.N2; v++) {
    C.c_ptr.push_back((int32_t)C.c_a.size());
    int32_t o = C.orig_of[v];
    C.c_a.push_back((int32_t)(nnz + o)); C.c_b.push_back(-1); C.c_d.push_back(-1);
    if (o < nvar)
      for (int32_t q : xr[o]) {
        int32_t rr = rowof[q];
        C.c_a.push_back(C.r_jsrc[q]); C.c_b.push_back((int32_t)(nnz + C.r_orig[rr])); C.c_d.push_back(C.r_dsrc[rr]);
      }
  }
  C.c_ptr.push_back((int32_t)C.c_a.size());
  C.cstride = C.ncs + nvar + C.N2;
  {
    const int32_t n_mat = (int32_t)(C.ncs + nvar), n_all = (int32_t)C.cstride;
    C.c_order.resize(n_all);
    std::iota(C.c_order.begin(), C.c_order.end(), 0);
    // identity: measured on MI355X, sorting the slots by contribution count (no divergence inside a
    // wavefront) is 2x SLOWER than the natural column-major order, whose gathers and stores coalesce
    (void)n_mat;
  }
  // ---- chunks for the LDS-staged kernel: a chunk is a COLUMN RANGE [c0, c1) of the condensed system and
  // owns three contiguous slot ranges: the matrix slots of those columns, their rho slots and their
  // right-hand-side slots.  They share most sources (the Jacobian rows of the residuals touching the
  // columns), so one staged tile serves all three.
  {
    const int32_t SLOTS_MAX = 256, TMAX = 896, GAP = 2;  // measured best on MI355X (448/1536: 8.4 ms, 256/896: 5.9 ms, 128/512: 6.9 ms at B = 8192)
    // first matrix slot of every column (slots are column-major; rows2/cols2 still hold the pattern)
    std::vector<int32_t> colstart(C.N2 + 1, 0);
    {
      std::vector<int32_t> cnt(C.N2 + 1, 0);
      for (int64_t s_ = 0; s_ < C.ncs; s_++) cnt[C.cols2[s_] - 1]++;
      colstart[0] = 0;
      for (int64_t j = 0; j < C.N2; j++) colstart[j + 1] = colstart[j] + cnt[j];
    }
    C.c_la.assign(C.c_a.size(), 0); C.c_lb.assign(C.c_a.size(), -1); C.c_ld.assign(C.c_a.size(), -1);
    C.ch_slot.clear(); C.ch_rng.clear(); C.ch_tile.clear(); C.rng_start.clear(); C.rng_len.clear();
    C.ch_tptr.clear(); C.tile_src.clear();
    std::vector<int32_t> src;
    auto ranges_of = [&](int32_t c0, int32_t c1, int32_t rg[6]) {
      rg[0] = colstart[c0]; rg[1] = colstart[c1] - colstart[c0];                       // matrix slots
      int32_t x0 = std::min<int32_t>(c0, (int32_t)nvar), x1 = std::min<int32_t>(c1, (int32_t)nvar);
      rg[2] = (int32_t)C.ncs + x0; rg[3] = x1 - x0;                                     // rho slots
      rg[4] = (int32_t)(C.ncs + nvar) + c0; rg[5] = c1 - c0;                            // rhs slots
    };
    int32_t c0 = 0;
    while (c0 < C.N2) {
      int32_t c1 = c0 + 1, rg[6];
      while (c1 < C.N2) {  // grow while the slot count stays within the workgroup's reach
        ranges_of(c0, c1 + 1, rg);
        if (rg[1] + rg[3] + rg[5] > SLOTS_MAX) break;
        c1++;
      }
      std::vector<int32_t> rs, rl;
      int32_t tile = 0;
      while (true) {  // shrink until the staged tile fits
        ranges_of(c0, c1, rg);
        src.clear();
        for (int q = 0; q < 3; q++)
          for (int32_t s_ = rg[2 * q]; s_ < rg[2 * q] + rg[2 * q + 1]; s_++)
            for (int32_t c = C.c_ptr[s_]; c < C.c_ptr[s_ + 1]; c++) {
              src.push_back(C.c_a[c]);
              if (C.c_b[c] >= 0) { src.push_back(C.c_b[c]); src.push_back(C.c_d[c]); }
            }
        std::sort(src.begin(), src.end());
        src.erase(std::unique(src.begin(), src.end()), src.end());
        rs.clear(); rl.clear(); tile = 0;
        for (size_t i = 0; i < src.size();) {
          size_t j = i;
          while (j + 1 < src.size() && src[j + 1] - src[j] <= GAP && ((src[j + 1] < nnz) == (src[i] < nnz))) j++;
          rs.push_back(src[i]); rl.push_back(src[j] - src[i] + 1);
          tile += src[j] - src[i] + 1;
          i = j + 1;
        }
        if (tile <= TMAX || c1 == c0 + 1) break;
        c1 = c0 + std::max(1, (c1 - c0) / 2);
      }
      // (a column with a huge fan-in makes the tile too large for LDS: the plain slot kernel is used then)
      std::vector<int32_t> roff(rs.size());
      int32_t o = 0;
      for (size_t i = 0; i < rs.size(); i++) { roff[i] = o; o += rl[i]; }
      auto local = [&](int32_t g) {
        size_t i = std::upper_bound(rs.begin(), rs.end(), g) - rs.begin() - 1;
        return roff[i] + (g - rs[i]);
      };
      for (int q = 0; q < 3; q++)
        for (int32_t s_ = rg[2 * q]; s_ < rg[2 * q] + rg[2 * q + 1]; s_++)
          for (int32_t c = C.c_ptr[s_]; c < C.c_ptr[s_ + 1]; c++) {
            C.c_la[c] = local(C.c_a[c]);
            if (C.c_b[c] >= 0) { C.c_lb[c] = local(C.c_b[c]); C.c_ld[c] = local(C.c_d[c]); }
          }
      C.ch_tptr.push_back((int32_t)C.tile_src.size());
      for (size_t i = 0; i < rs.size(); i++)
        for (int32_t k = 0; k < rl[i]; k++) C.tile_src.push_back(rs[i] + k);
      for (int q = 0; q < 6; q++) C.ch_slot.push_back(rg[q]);  // six words per chunk: (start, len) x 3
      C.ch_tile.push_back(tile);
      C.tile_max = std::max(C.tile_max, tile);
      c0 = c1;
    }
    C.ch_tptr.push_back((int32_t)C.tile_src.size());
    C.tiled_ok = C.tile_max <= 4608;
    C.c_pack.assign(C.c_a.size(), 0);
    if (C.tiled_ok)
      for (size_t c = 0; c < C.c_a.size(); c++)
        C.c_pack[c] = (uint64_t)C.c_la[c] | ((uint64_t)(C.c_lb[c] + 1) << 16) | ((uint64_t)(C.c_ld[c] + 1) << 32);
    for (size_t k = 0; k < C.ch_tile.size(); k++) {
      int32_t ncon_ = 0, nsl = 0;
      for (int q = 0; q < 3; q++) {
        int32_t s0_ = C.ch_slot[6 * k + 2 * q], len_ = C.ch_slot[6 * k + 2 * q + 1];
        ncon_ += C.c_ptr[s0_ + len_] - C.c_ptr[s0_];
        nsl += len_;
      }
      C.chunk_ncon_max = std::max(C.chunk_ncon_max, ncon_);
      C.chunk_nslot_max = std::max(C.chunk_nslot_max, nsl);
    }
    C.ch_region[0] = 0; C.ch_region[1] = C.ch_region[2] = C.ch_region[3] = (int32_t)C.ch_tile.size();
    // the tiled kernel's whole LDS need (kernels_aux.hip, launch_condense_tiled: four problems' tiles + a chunk's contribution and
    // slot lists) must fit a workgroup: a dense Jacobian gives every slot of J'J one contribution per residual row, and a chunk of
    // 256 such slots exceeded the limit (found by the option fuzz of round 6 with dense_backend = 0: the launch failed with
    // "invalid argument") — such patterns keep the plain slot kernel
    const size_t lds_need = (size_t)CONDENSE_TILED_PROBLEMS * (size_t)C.tile_max * 8 + (size_t)C.chunk_ncon_max * 8 + ((size_t)C.chunk_nslot_max + 8) * 4;
    if (lds_need > 144 * 1024) C.tiled_ok = false;
  }
  C.active = true;
  msg.clear();
  return 0;
}

}  // namespace cnl
