// band.cpp — generator of the band program (band.h): checks that the KKT pattern of /root/reference/src/CaNNOLeS.jl:256-315 is a
// band in the natural order of the variables and writes, per part of the chain, the step / row / epoch blocks the kernels of
// band.hip execute.  Pure index arithmetic; no values are touched.
#include "band.h"

#include <algorithm>
#include <cstdio>
#include <map>

namespace cnl {

namespace {

struct XEnt { int32_t src; };
struct RowInfo {
  int32_t dsrc = -1, ndiag = 0;
  std::vector<int32_t> xs, srcs;   // columns and their COO entries
};
struct BorderInfo {
  int32_t dsrc = -1, ndiag = 0;
  std::vector<int32_t> xs, srcs;
};

// one step of a part before the sources are turned into LDS offsets (sources: >= 0 index into vals, <= -2: rhs index -(v + 2), -1 none)
struct RowOp { int32_t di, j[BAND_NB], rr, r; };
struct StepOp {
  int32_t flags = 0;
  int32_t dg[3] = {-1, -1, -1}, rho = -1, od[2 * BAND_HW], bc[2] = {-1, -1}, rx = -1;
  int32_t border = -1;       // border table index (enter / pivot)
  int32_t xpiv = -1;         // variable pivoted (BF_PIVOT_X)
  int32_t lev_b = -1, lev_x = -1;   // factor event numbers of the step's pivots
  std::vector<RowOp> rows;
  StepOp() { for (int i = 0; i < 2 * BAND_HW; i++) od[i] = -1; }
};
inline int32_t rhs_src(int64_t i) { return (int32_t)(-(i + 2)); }

// packs the sources of one epoch into 64-byte pieces (eight consecutive elements of one array); returns false when more than
// BAND_NPIECE are needed.  (Round 5 also had a layout with four 128-byte pieces per epoch: measured equal, removed in round 6.)
struct Packer {
  int32_t len[3];
  struct Piece { int32_t arr, base, slot; };
  std::vector<Piece> pieces;
  bool pack(std::vector<int32_t> (&need)[3]) {
    pieces.clear();
    for (int a = 0; a < 3; a++) {
      std::vector<int32_t>& v = need[a];
      std::sort(v.begin(), v.end());
      v.erase(std::unique(v.begin(), v.end()), v.end());
      size_t i = 0;
      while (i < v.size()) {
        int32_t b = v[i];
        if (b > len[a] - 8) b = len[a] - 8;   // the piece must stay inside the array
        if (b < 0) return false;
        pieces.push_back({a, b, (int32_t)pieces.size()});
        while (i < v.size() && v[i] < b + 8) i++;
      }
    }
    return (int)pieces.size() <= BAND_NPIECE;
  }
  // LDS byte offset of element e of array a
  int32_t off(int a, int32_t e) const {
    for (const Piece& pc : pieces)
      if (pc.arr == a && e >= pc.base && e < pc.base + 8) return (int32_t)((BAND_IN_OFF + 8 * pc.slot + (e - pc.base)) * 8);
    return -1;
  }
};

}  // namespace

void build_band_plan(BandPlan& B, int64_t N, int64_t nnz, const int64_t* rows1, const int64_t* cols1, int64_t nvar, int64_t nequ,
                     int64_t ncon, int nparts_wanted) {
  B = BandPlan();
  auto no = [&](const std::string& w) { B.ok = false; B.why = w; };
  const int64_t n = nvar, m = nequ, p = ncon;
  if (N != n + m + p) return no("N != nvar + nequ + ncon");
  if (n < 2 * BAND_NB + 2 || nnz < 16 || N < 16 || nnz >= (1 << 27) || N >= (1 << 27)) return no("size out of range");
  B.n = (int32_t)n; B.N = (int32_t)N; B.nnz = (int32_t)nnz;
  // ---- classify the COO entries ---------------------------------------------------------------------------------------
  // (x, x): plain entries of the top-left block, the last nvar COO entries being the rho slots (src/CaNNOLeS.jl:312-315)
  std::vector<std::vector<int32_t>> xx_of_col(n);   // entries (a, b), a >= b, stored with their source, per column b, COO order
  std::vector<int32_t> xx_row(nnz, -1);
  std::vector<RowInfo> R(m);
  std::vector<BorderInfo> L(p);
  for (int64_t e = 0; e < nnz; e++) {
    const int64_t r0 = rows1[e] - 1, c0 = cols1[e] - 1;
    if (r0 < 0 || r0 >= N || c0 < 0 || c0 > r0) return no("entry outside the lower triangle");
    if (r0 < n) {
      if (r0 - c0 > BAND_HW) return no("Hessian entry outside the band");
      xx_of_col[c0].push_back((int32_t)e);
      xx_row[e] = (int32_t)r0;
    } else if (r0 < n + m) {
      RowInfo& q = R[r0 - n];
      if (c0 < n) { q.xs.push_back((int32_t)c0); q.srcs.push_back((int32_t)e); }
      else if (c0 == r0) { q.dsrc = (int32_t)e; q.ndiag++; }
      else return no("entry inside the residual block off its diagonal");
    } else {
      BorderInfo& q = L[r0 - n - m];
      if (c0 < n) { q.xs.push_back((int32_t)c0); q.srcs.push_back((int32_t)e); }
      else if (c0 == r0) { q.dsrc = (int32_t)e; q.ndiag++; }
      else return no("entry coupling a multiplier with a residual or another multiplier");
    }
  }
  for (int64_t r = 0; r < m; r++) {
    RowInfo& q = R[r];
    if (q.ndiag != 1) return no("residual row without exactly one diagonal entry");
    if (!q.xs.empty()) {
      const auto mm = std::minmax_element(q.xs.begin(), q.xs.end());
      if (*mm.second - *mm.first > BAND_HW) return no("residual row wider than the band");
      std::vector<int32_t> s = q.xs;
      std::sort(s.begin(), s.end());
      if (std::adjacent_find(s.begin(), s.end()) != s.end()) return no("duplicate Jacobian entry");
    }
  }
  for (int64_t k = 0; k < p; k++) {
    BorderInfo& q = L[k];
    if (q.ndiag != 1) return no("multiplier without exactly one diagonal entry");
    if (q.xs.empty()) return no("constraint row without entries");
  }
  const int64_t rho_begin = nnz - n;
  // ---- the split --------------------------------------------------------------------------------------------------------
  int nparts = nparts_wanted >= 2 && n >= 16 * BAND_NB ? 2 : 1;
  int32_t m0 = (int32_t)n;   // nparts == 1: everything in part 0
  if (nparts == 2) {
    // part 0 sees variables [0, m0 + HW), part 1 sees [m0, n): every constraint row must lie inside one of the two ranges
    auto valid = [&](int64_t c) {
      if (c < BAND_NB || c + BAND_HW + BAND_NB > n) return false;
      for (int64_t k = 0; k < p; k++) {
        const auto mm = std::minmax_element(L[k].xs.begin(), L[k].xs.end());
        if (!(*mm.second < c + BAND_HW || *mm.first >= c)) return false;
      }
      return true;
    };
    const int64_t want = (n - BAND_HW) / 2;
    int64_t found = -1;
    for (int64_t dlt = 0; dlt <= n / 4 && found < 0; dlt++) {
      if (valid(want + dlt)) found = want + dlt;
      else if (valid(want - dlt)) found = want - dlt;
    }
    if (found < 0) nparts = 1;
    else m0 = (int32_t)found;
  }
  B.nparts = nparts; B.m0 = m0;
  // ---- per part ---------------------------------------------------------------------------------------------------------
  int64_t loff = 0;
  for (int part = 0; part < nparts; part++) {
    BandPart& Q = B.part[part];
    // sequence of the variables the part sees, in entering order; the first npiv of them are pivoted
    std::vector<int32_t> seq;
    int32_t npiv;
    if (nparts == 1) { for (int32_t x = 0; x < n; x++) seq.push_back(x); npiv = (int32_t)n; }
    else if (part == 0) { for (int32_t x = 0; x < m0 + BAND_HW; x++) seq.push_back(x); npiv = m0; }
    else { for (int32_t x = (int32_t)n - 1; x >= m0; x--) seq.push_back(x); npiv = (int32_t)n - m0 - BAND_HW; }
    std::vector<int32_t> pos(n, -1);
    for (size_t t = 0; t < seq.size(); t++) pos[seq[t]] = (int32_t)t;
    // does the part own position (a, b) / row r?  (two parts: part 0 owns what lies entirely below m0 + HW)
    auto owns = [&](int32_t hi) { return nparts == 1 || (part == 0 ? hi < m0 + BAND_HW : hi >= m0 + BAND_HW); };
    const int32_t nsteps_raw = npiv + BAND_HW;   // step u enters seq[u] and pivots seq[u - HW]
    std::vector<StepOp> S(nsteps_raw);
    for (int32_t u = 0; u < nsteps_raw; u++) {
      if (u < (int32_t)seq.size()) S[u].flags |= BF_ENTER_X;
      if (u >= BAND_HW && u - BAND_HW < npiv) { S[u].flags |= BF_PIVOT_X; S[u].xpiv = seq[u - BAND_HW]; }
    }
    // plain entries of the top-left block
    for (int64_t c = 0; c < n; c++)
      for (int32_t e : xx_of_col[c]) {
        const int32_t a = xx_row[e], b = (int32_t)c;
        if (!owns(std::max(a, b))) continue;
        if (pos[a] < 0 || pos[b] < 0) return no("internal: entry outside its part");
        const int32_t u = std::max(pos[a], pos[b]), k = std::abs(pos[a] - pos[b]);
        StepOp& st = S[u];
        if (k == 0) {
          if (e >= rho_begin) { if (st.rho >= 0) return no("two rho slots on one variable"); st.rho = e; }
          else {
            if (st.rho >= 0) return no("plain entry behind the rho slot");
            int q = 0;
            while (q < 3 && st.dg[q] >= 0) q++;
            if (q == 3) return no("more than three plain entries on one diagonal position");
            st.dg[q] = e;
          }
        } else {
          int32_t* o = st.od + 2 * (k - 1);
          if (o[0] < 0) o[0] = e;
          else if (o[1] < 0) o[1] = e;
          else return no("more than two entries on one off-diagonal position");
        }
      }
    // right-hand side of the variables
    for (size_t t = 0; t < seq.size(); t++)
      if (owns(seq[t])) S[t].rx = rhs_src(seq[t]);
    // residual rows: row r is condensed at the step where its last column enters — or later, while its first column is still
    // unpivoted (the step of that pivot included: rows come before the pivot), so that rows which complete together (the end of
    // the chain: the last column completes three rows) are spread over the following steps instead of piling their operands
    // into one epoch
    struct RowAt { int32_t r, u, deadline; };
    std::vector<RowAt> mine_rows;
    for (int64_t r = 0; r < m; r++) {
      const RowInfo& q = R[r];
      int32_t hi = -1;
      for (int32_t x : q.xs) hi = std::max(hi, x);
      if (q.xs.empty()) { if (part == 0) mine_rows.push_back({(int32_t)r, 0, 0}); continue; }
      if (!owns(hi)) continue;
      int32_t plo = 1 << 30, phi = -1;
      for (int32_t x : q.xs) { if (pos[x] < 0) return no("internal: row outside its part"); plo = std::min(plo, pos[x]); phi = std::max(phi, pos[x]); }
      if (phi - plo > BAND_HW) return no("residual row wider than the band");
      mine_rows.push_back({(int32_t)r, phi, std::min(plo + BAND_HW, nsteps_raw - 1)});
    }
    std::stable_sort(mine_rows.begin(), mine_rows.end(), [&](const RowAt& x, const RowAt& y) { return x.u != y.u ? x.u < y.u : (part == 0 ? x.r < y.r : x.r > y.r); });
    {
      int32_t next_free = 0;
      for (RowAt& ra : mine_rows) {
        const int32_t want = std::max(ra.u, next_free);
        if (want <= ra.deadline) ra.u = want;
        next_free = std::max(next_free, ra.u + 1);
      }
    }
    for (const RowAt& ra : mine_rows) {
      const RowInfo& q = R[ra.r];
      const int32_t u = ra.u;
      RowOp ro;
      ro.di = q.dsrc; ro.rr = rhs_src(n + ra.r); ro.r = ra.r;
      for (int s = 0; s < BAND_NB; s++) ro.j[s] = -1;
      for (size_t i = 0; i < q.xs.size(); i++) {
        const int32_t k = pos[q.xs[i]] - (u - BAND_HW);   // live position: 0 = the step's pivot, HW = the variable entering at step u
        if (k < 0 || k > BAND_HW) return no("internal: row column outside the window");
        ro.j[k] = q.srcs[i];
      }
      S[u].rows.push_back(ro);
    }
    for (StepOp& st : S) {
      if (st.rows.size() > 15) return no("more than 15 residual rows complete at one variable");
      // the backward sweep writes the residual components of an epoch as one contiguous run: rows in index order along the part
      std::sort(st.rows.begin(), st.rows.end(), [&](const RowOp& x, const RowOp& y) { return part == 0 ? x.r < y.r : x.r > y.r; });
    }
    // borders (multipliers)
    std::vector<std::pair<int32_t, int32_t>> live;   // (enter step, pivot step)
    for (int64_t k = 0; k < p; k++) {
      const BorderInfo& q = L[k];
      int32_t lo = 1 << 30, hi = -1;
      for (int32_t x : q.xs) { lo = std::min(lo, x); hi = std::max(hi, x); }
      const bool mine = nparts == 1 || (part == 0 ? hi < m0 + BAND_HW : !(hi < m0 + BAND_HW));
      if (!mine) continue;
      int32_t plo = 1 << 30, phi = -1;
      for (int32_t x : q.xs) { if (pos[x] < 0) return no("constraint row straddles the split"); plo = std::min(plo, pos[x]); phi = std::max(phi, pos[x]); }
      if (phi >= nsteps_raw) return no("internal: border beyond the part");
      // a border must be pivoted before the last variable it touches: with pivot step phi that variable (entered at phi) is
      // pivoted at phi + HW or belongs to the junction
      const int32_t bi = (int32_t)(Q.borders.size() / BAND_BW);
      Q.borders.push_back(q.dsrc); Q.borders.push_back((int32_t)(n + m + k)); Q.borders.push_back((int32_t)(n + m + k)); Q.borders.push_back(0);
      if (S[plo].flags & (BF_ENTER_B | BF_PIVOT_B)) return no("two multipliers live at once");
      if (S[phi].flags & (BF_ENTER_B | BF_PIVOT_B) && plo != phi) return no("two multipliers live at once");
      S[plo].flags |= BF_ENTER_B; S[plo].border = bi;
      S[phi].flags |= BF_PIVOT_B; S[phi].border = bi;
      live.push_back({plo, phi});
      for (size_t i = 0; i < q.xs.size(); i++) {
        StepOp& st = S[pos[q.xs[i]]];
        if (st.bc[0] < 0) st.bc[0] = q.srcs[i];
        else if (st.bc[1] < 0) st.bc[1] = q.srcs[i];
        else return no("more than two entries on one constraint position");
      }
    }
    std::sort(live.begin(), live.end());
    for (size_t i = 1; i < live.size(); i++)
      if (live[i].first <= live[i - 1].second) return no("two multipliers live at once");
    // a step may carry one border index only: enter and pivot of DIFFERENT borders in one step were refused above; a border that
    // enters and is pivoted in the same step (one column) is fine
    const int32_t nsteps = nsteps_raw;
    Q.nsteps = nsteps; Q.npiv = npiv;
    // factor events in forward order: border pivot, then band pivot
    int32_t nev = 0;
    for (StepOp& st : S) {
      if (st.flags & BF_PIVOT_B) st.lev_b = nev++;
      if (st.flags & BF_PIVOT_X) st.lev_x = nev++;
    }
    Q.nevents = nev;
    Q.loff = loff;
    const int64_t lpart = (int64_t)nev * BAND_LREC;
    loff += (lpart + 16 + 7) & ~(int64_t)7;   // + 16: slack behind a part's records
    // ---- epochs: pack the operands into pieces, turn sources into LDS offsets ------------------------------------------
    Q.epochs.clear();
    Q.fops.clear(); Q.bops.clear();
    std::vector<std::vector<int32_t>> fblocks(nsteps), bblocks(nsteps);
    Packer pk;
    pk.len[0] = (int32_t)nnz; pk.len[1] = (int32_t)N; pk.len[2] = (int32_t)(lpart + 16);
    if (lpart + 16 >= (1 << 27)) return no("factor too long");
    const int32_t ZB = BAND_ZERO_OFF * 8;
    // An epoch is a run of BAND_EPOCH steps whose operands must fit the pieces and whose outputs the rings.
    std::string why_not;
    auto try_epoch = [&](const int32_t u0, const int32_t u1, int32_t* E) -> bool {
      auto no = [&](const std::string& w) { why_not = w; return false; };
      // factor events, pivots and rows of the epoch
      int32_t ev_lo = 1 << 30, ev_hi = -1, x_lo = 1 << 30, x_hi = -1, x_cnt = 0, r_lo = 1 << 30, r_hi = -1, r_cnt = 0;
      for (int32_t u = u0; u < u1; u++) {
        const StepOp& st = S[u];
        for (int32_t ev : {st.lev_b, st.lev_x}) if (ev >= 0) { ev_lo = std::min(ev_lo, ev); ev_hi = std::max(ev_hi, ev); }
        if (st.flags & BF_PIVOT_X) { x_lo = std::min(x_lo, st.xpiv); x_hi = std::max(x_hi, st.xpiv); x_cnt++; }
        for (const RowOp& ro : st.rows) { r_lo = std::min(r_lo, ro.r); r_hi = std::max(r_hi, ro.r); r_cnt++; }
      }
      int32_t oplen = 0;
      for (int32_t u = u0; u < u1; u++) oplen += BAND_SW + BAND_RW * (int32_t)S[u].rows.size();
      if (oplen > BAND_REC_MAX) return no("the blocks of an epoch do not fit the record buffer");
      E[BE_OPLEN] = oplen;
      const int32_t lbase = ev_hi >= 0 ? ev_lo * BAND_LREC : 0, lcnt = ev_hi >= 0 ? (ev_hi - ev_lo + 1) * BAND_LREC : 0;
      // the out ring takes the factor records of half an epoch (steps 0 .. 3, then 4 .. 7)
      const int32_t uh = std::min(u1, u0 + BAND_EPOCH / 2);
      int32_t evh_lo = 1 << 30, evh_hi = -1, evg_lo = 1 << 30, evg_hi = -1;
      for (int32_t u = u0; u < u1; u++)
        for (int32_t ev : {S[u].lev_b, S[u].lev_x})
          if (ev >= 0) {
            if (u < uh) { evh_lo = std::min(evh_lo, ev); evh_hi = std::max(evh_hi, ev); }
            else { evg_lo = std::min(evg_lo, ev); evg_hi = std::max(evg_hi, ev); }
          }
      const int32_t lbase1 = evh_hi >= 0 ? evh_lo * BAND_LREC : 0, lcnt1 = evh_hi >= 0 ? (evh_hi - evh_lo + 1) * BAND_LREC : 0;
      const int32_t lbase2 = evg_hi >= 0 ? evg_lo * BAND_LREC : 0, lcnt2 = evg_hi >= 0 ? (evg_hi - evg_lo + 1) * BAND_LREC : 0;
      if (lcnt1 > BAND_LOUT_MAX || lcnt2 > BAND_LOUT_MAX) return no("more factor records in half an epoch than the ring holds");
      if (x_cnt && x_hi - x_lo + 1 != x_cnt) return no("pivots of an epoch are not consecutive variables");
      if (r_cnt && r_hi - r_lo + 1 != r_cnt) return no("rows of an epoch are not consecutive");
      if (x_cnt > BAND_DX_MAX || r_cnt > BAND_DR_MAX) return no("more outputs in an epoch than the rings hold");
      E[BE_LBASE] = lbase1; E[BE_LCNT] = lcnt1; E[BE_LBASE2] = lbase2; E[BE_LCNT2] = lcnt2;
      E[BE_DXLO] = x_cnt ? x_lo : 0; E[BE_DXCNT] = x_cnt;
      E[BE_DRLO] = r_cnt ? (int32_t)(n + r_lo) : 0; E[BE_DRCNT] = r_cnt;
      // forward operands
      for (int dir = 0; dir < 2; dir++) {
        std::vector<int32_t> need[3];
        auto want = [&](int32_t s) {
          if (s == -1) return;
          if (s >= 0) need[0].push_back(s); else need[1].push_back(-(s + 2));
        };
        for (int32_t u = u0; u < u1; u++) {
          const StepOp& st = S[u];
          if (dir == 0) {
            for (int q = 0; q < 3; q++) want(st.dg[q]);
            want(st.rho);
            for (int q = 0; q < 2 * BAND_HW; q++) want(st.od[q]);
            want(st.bc[0]); want(st.bc[1]); want(st.rx);
          }
          for (const RowOp& ro : st.rows) {
            want(ro.di); want(ro.rr);
            for (int s = 0; s < BAND_NB; s++) want(ro.j[s]);
          }
        }
        if (dir == 1) for (int32_t i = 0; i < lcnt; i++) need[2].push_back(lbase + i);
        if (!pk.pack(need)) {
          std::string w = "an epoch needs more operand pieces than a lane holds (part " + std::to_string(part) + ", steps " + std::to_string(u0) + ".." + std::to_string(u1) + (dir ? ", backward" : ", forward") + ":";
          for (auto& pc : pk.pieces) w += " " + std::to_string(pc.arr) + ":" + std::to_string(pc.base);
          return no(w + ")");
        }
        int32_t* PP = E + (dir == 0 ? BE_FP : BE_BP);
        for (int k = 0; k < BAND_NPIECE; k++) PP[k] = -1;
        for (const Packer::Piece& pc : pk.pieces) PP[pc.slot] = pc.base | (pc.arr << 28);
        auto off = [&](int32_t s) -> int32_t {
          if (s == -1) return ZB;
          return s >= 0 ? pk.off(0, s) : pk.off(1, -(s + 2));
        };
        for (int32_t u = u0; u < u1; u++) {
          const StepOp& st = S[u];
          std::vector<int32_t>& blk = dir == 0 ? fblocks[u] : bblocks[u];
          blk.assign(BAND_SW + BAND_RW * st.rows.size(), 0);
          blk[BS_FLAGS] = st.flags | ((int32_t)st.rows.size() << 8);
          blk[BS_BORDER] = st.border;
          for (int q = 0; q < 3; q++) blk[BS_DG0 + q] = dir == 0 ? off(st.dg[q]) : ZB;
          blk[BS_RHO] = dir == 0 ? off(st.rho) : ZB;
          if (dir == 0 && st.rho < 0) blk[BS_FLAGS] |= 1 << 16;   // no rho slot: the ladder's rho is not applied here
          for (int q = 0; q < 2 * BAND_HW; q++) blk[BS_OD + q] = dir == 0 ? off(st.od[q]) : ZB;
          blk[BS_BC0] = dir == 0 ? off(st.bc[0]) : ZB; blk[BS_BC1] = dir == 0 ? off(st.bc[1]) : ZB;
          blk[BS_RX] = dir == 0 ? off(st.rx) : ZB;
          if (dir == 0) {
            const int32_t lbh = u < uh ? lbase1 : lbase2;
            blk[BS_LB] = st.lev_b >= 0 ? (BAND_LOUT_OFF + st.lev_b * BAND_LREC - lbh) * 8 : ZB;
            blk[BS_LX] = st.lev_x >= 0 ? (BAND_LOUT_OFF + st.lev_x * BAND_LREC - lbh) * 8 : ZB;
            blk[BS_DX] = ZB;
          } else {
            blk[BS_LB] = st.lev_b >= 0 ? pk.off(2, st.lev_b * BAND_LREC) : ZB;
            blk[BS_LX] = st.lev_x >= 0 ? pk.off(2, st.lev_x * BAND_LREC) : ZB;
            // a factor record must not straddle two pieces that are not adjacent in LDS: pieces of one array are consecutive and
            // contiguous in the array unless clamped at its end, where they may overlap — check every element
            for (int32_t ev : {st.lev_b, st.lev_x})
              if (ev >= 0)
                for (int i = 0; i < BAND_LREC; i++)
                  if (pk.off(2, ev * BAND_LREC + i) != pk.off(2, ev * BAND_LREC) + 8 * i) return no("internal: factor record not contiguous in LDS");
            blk[BS_DX] = (st.flags & BF_PIVOT_X) ? (BAND_DX_OFF + st.xpiv - x_lo) * 8 : ZB;
          }
          for (size_t i = 0; i < st.rows.size(); i++) {
            const RowOp& ro = st.rows[i];
            int32_t* rb = blk.data() + BAND_SW + BAND_RW * i;
            rb[BR_DI] = off(ro.di); rb[BR_RR] = off(ro.rr);
            for (int s = 0; s < BAND_NB; s++) rb[BR_J0 + s] = off(ro.j[s]);
            rb[BR_DR] = (BAND_DR_OFF + ro.r - r_lo) * 8;
            if (rb[BR_DI] < 0 || rb[BR_RR] < 0) return no("internal: operand without a piece");
            for (int s = 0; s < BAND_NB; s++) if (rb[BR_J0 + s] < 0) return no("internal: operand without a piece");
          }
          for (int q = BS_DG0; q <= BS_RX; q++) if (blk[q] < 0) return no("internal: operand without a piece");
        }
      }
      return true;
    };
    for (int32_t u0 = 0; u0 < nsteps;) {
      std::vector<int32_t> E(BAND_EW, 0);
      // every epoch but the last has exactly BAND_EPOCH steps: the kernels' step code is specialised by step number modulo
      // BAND_EPOCH (= the number of window slots), so an epoch starts at slot 0
      const int32_t cnt = std::min<int32_t>(BAND_EPOCH, nsteps - u0);
      if (!try_epoch(u0, u0 + cnt, E.data())) return no(why_not);
      E[BE_NSTEP] = cnt;
      Q.epochs.insert(Q.epochs.end(), E.begin(), E.end());
      u0 += cnt;
    }
    Q.nepochs = (int32_t)(Q.epochs.size() / BAND_EW);
    {
      int32_t fo = 0;
      for (int32_t e = 0; e < Q.nepochs; e++) { Q.epochs[(size_t)e * BAND_EW + BE_FOFF] = fo; fo += Q.epochs[(size_t)e * BAND_EW + BE_OPLEN]; }
      int32_t bo = 0;
      for (int32_t e = Q.nepochs - 1; e >= 0; e--) { Q.epochs[(size_t)e * BAND_EW + BE_BOFF] = bo; bo += Q.epochs[(size_t)e * BAND_EW + BE_OPLEN]; }
    }
    for (int32_t u = 0; u < nsteps; u++) Q.fops.insert(Q.fops.end(), fblocks[u].begin(), fblocks[u].end());
    // backward order: steps reversed, the rows of a step reversed
    for (int32_t u = nsteps - 1; u >= 0; u--) {
      const std::vector<int32_t>& blk = bblocks[u];
      Q.bops.insert(Q.bops.end(), blk.begin(), blk.begin() + BAND_SW);
      const int nr = (int)((blk.size() - BAND_SW) / BAND_RW);
      for (int i = nr - 1; i >= 0; i--) Q.bops.insert(Q.bops.end(), blk.begin() + BAND_SW + BAND_RW * i, blk.begin() + BAND_SW + BAND_RW * (i + 1));
    }
    // slack for the kernels' prefetch of the next block
    Q.fops.resize(Q.fops.size() + BAND_REC_MAX + 64, 0);
    Q.bops.resize(Q.bops.size() + BAND_REC_MAX + 64, 0);
  }
  B.lsize = loff + 8;
  B.ok = true;
}

}  // namespace cnl
