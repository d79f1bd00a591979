// analysis.cpp — symbolic analysis: ordering, elimination tree, supernodes,
// multifrontal index maps and the static LDS stack layout (see plan.h).
//
// Reference behaviour being replaced (not translated): LDLFactStruct's ctor,
// /root/reference/src/solver_types.jl:61-65, which merges duplicate COO
// entries (`sparse`), keeps one triangle (`triu`) and calls `ldl_analyze`
// (AMD + etree + column counts in LDLFactorizations.jl).  Here the ordering
// is constrained by the KKT block structure of src/CaNNOLeS.jl:282 —
// residual nodes (pivot -1) first, multipliers (pivot -delta) last — so that
// every x pivot is an entry of a Schur complement of H + rho I + J'J and
// "factorisation completes" coincides with "inertia is (nvar, nequ+ncon, 0)".
#include "plan.h"

#include <algorithm>
#include <array>
#include <atomic>
#include <functional>
#include <thread>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <unordered_map>

namespace cnl {
namespace {

typedef std::vector<int32_t> ivec;

inline int64_t tri(int64_t i) { return i * (i + 1) / 2; }

struct Graph {  // symmetric adjacency, no diagonal, sorted neighbour lists
  int32_t n = 0;
  ivec ptr, idx;
  int32_t deg(int32_t v) const { return ptr[v + 1] - ptr[v]; }
};

// ---------------------------------------------------------------------------
// Minimum degree on an explicit elimination graph (exact external degree, no
// quotient graph).  Used on ND leaves and, as one candidate, on the whole x
// graph.  `nodes` are vertices of g; adjacency outside `nodes` is ignored.
// A dense tail (remaining graph nearly complete) is emitted in natural order.
void min_degree(const Graph& g, const ivec& nodes, ivec& out) {
  const int32_t m = (int32_t)nodes.size();
  if (m <= 2) { out.insert(out.end(), nodes.begin(), nodes.end()); return; }
  std::unordered_map<int32_t, int32_t> loc;  // global -> local (small graphs only) or dense map
  ivec locv;
  bool dense_map = (int64_t)m * 4 > g.n;
  if (dense_map) { locv.assign(g.n, -1); for (int32_t i = 0; i < m; i++) locv[nodes[i]] = i; }
  else { loc.reserve(m * 2); for (int32_t i = 0; i < m; i++) loc[nodes[i]] = i; }
  auto local = [&](int32_t v) -> int32_t {
    if (dense_map) return locv[v];
    auto it = loc.find(v);
    return it == loc.end() ? -1 : it->second;
  };
  std::vector<ivec> adj(m);
  for (int32_t i = 0; i < m; i++) {
    int32_t v = nodes[i];
    for (int32_t p = g.ptr[v]; p < g.ptr[v + 1]; p++) { int32_t l = local(g.idx[p]); if (l >= 0) adj[i].push_back(l); }
    std::sort(adj[i].begin(), adj[i].end());
  }
  // bucket lists by degree
  ivec degv(m), next(m, -1), prev(m, -1), head(m + 1, -1);
  std::vector<char> done(m, 0);
  auto bucket_insert = [&](int32_t v) {
    int32_t d = degv[v]; next[v] = head[d]; prev[v] = -1; if (head[d] >= 0) prev[head[d]] = v; head[d] = v;
  };
  auto bucket_remove = [&](int32_t v) {
    int32_t d = degv[v];
    if (prev[v] >= 0) next[prev[v]] = next[v]; else head[d] = next[v];
    if (next[v] >= 0) prev[next[v]] = prev[v];
  };
  for (int32_t i = 0; i < m; i++) { degv[i] = (int32_t)adj[i].size(); }
  for (int32_t i = m - 1; i >= 0; i--) bucket_insert(i);  // lower index at the head: deterministic ties
  int32_t mind = 0, left = m;
  ivec tmp;
  while (left > 0) {
    while (mind <= m && head[mind] < 0) mind++;
    int32_t v = head[mind];
    // dense tail: everything left is (almost) a clique
    if (mind >= left - 1) {
      for (int32_t i = 0; i < m; i++) if (!done[i]) { out.push_back(nodes[i]); }
      return;
    }
    bucket_remove(v); done[v] = 1; left--; out.push_back(nodes[v]);
    const ivec& nv = adj[v];
    for (int32_t u : nv) {
      // adj[u] = (adj[u] U nv) \ {u, v}
      tmp.clear();
      const ivec& au = adj[u];
      size_t a = 0, b = 0;
      while (a < au.size() || b < nv.size()) {
        int32_t x;
        if (b >= nv.size() || (a < au.size() && au[a] < nv[b])) x = au[a++];
        else if (a >= au.size() || nv[b] < au[a]) x = nv[b++];
        else { x = au[a]; a++; b++; }
        if (x != u && x != v) tmp.push_back(x);
      }
      bucket_remove(u);
      adj[u].swap(tmp);
      degv[u] = (int32_t)adj[u].size();
      bucket_insert(u);
      if (degv[u] < mind) mind = degv[u];
    }
    ivec().swap(adj[v]);
  }
}

// ---------------------------------------------------------------------------
// Nested dissection by BFS level structures (George's automatic ND) on the
// subgraph induced by `nodes`.  Leaves are ordered by min_degree.
struct NDWork {
  const Graph* g;
  ivec part;     // part id per vertex (-1 = not in play)
  ivec level, queue;
  int32_t leaf;
  int32_t next_part = 1;
  bool canon_leaves = false;  // leaves keep the caller's (canonical) order instead of minimum degree: chain-like inside a part
};

void bfs(NDWork& w, int32_t root, int32_t pid, ivec& order_out, int32_t& nlev) {
  // BFS inside part pid; level[] filled; returns visited vertices in BFS order
  order_out.clear();
  w.level[root] = 0; order_out.push_back(root);
  size_t h = 0; nlev = 1;
  // mark visited by temporarily negating part: use a separate stamp via level>=0
  while (h < order_out.size()) {
    int32_t v = order_out[h++];
    for (int32_t p = w.g->ptr[v]; p < w.g->ptr[v + 1]; p++) {
      int32_t u = w.g->idx[p];
      if (w.part[u] == pid && w.level[u] < 0) { w.level[u] = w.level[v] + 1; nlev = w.level[u] + 1; order_out.push_back(u); }
    }
  }
}

void nd_rec(NDWork& w, ivec nodes, ivec& out) {
  // iterative on an explicit stack of tasks to avoid deep recursion
  struct Task { ivec nodes; bool emit; bool presorted = false; };
  std::vector<Task> st;
  st.push_back({std::move(nodes), false});
  ivec comp, comp2;
  while (!st.empty()) {
    Task t = std::move(st.back()); st.pop_back();
    if (t.emit) { out.insert(out.end(), t.nodes.begin(), t.nodes.end()); continue; }
    ivec& nd = t.nodes;
    if ((int32_t)nd.size() <= w.leaf) {
      if (w.canon_leaves) { if (!t.presorted) std::sort(nd.begin(), nd.end()); out.insert(out.end(), nd.begin(), nd.end()); }
      else min_degree(*w.g, nd, out);
      continue;
    }
    int32_t pid = w.next_part++;
    for (int32_t v : nd) { w.part[v] = pid; w.level[v] = -1; }
    // connected component of the first vertex
    int32_t nlev;
    bfs(w, nd[0], pid, comp, nlev);
    if (comp.size() < nd.size()) {
      // disconnected: order the component found, then the rest (independent)
      ivec rest; rest.reserve(nd.size() - comp.size());
      for (int32_t v : nd) if (w.level[v] < 0) rest.push_back(v);
      for (int32_t v : nd) w.part[v] = -1;
      st.push_back({std::move(rest), false});
      st.push_back({comp, false});
      continue;
    }
    // pseudo-peripheral root: repeat BFS from a min-degree vertex of the last level
    for (int it = 0; it < 4; it++) {
      int32_t last = comp.back(), best = last, bd = 1 << 30;
      for (size_t k = comp.size(); k-- > 0 && w.level[comp[k]] == w.level[last];) {
        int32_t d = w.g->deg(comp[k]); if (d < bd) { bd = d; best = comp[k]; }
      }
      for (int32_t v : nd) w.level[v] = -1;
      int32_t nl2; bfs(w, best, pid, comp2, nl2);
      bool better = nl2 > nlev;
      comp.swap(comp2); nlev = nl2;
      if (!better) break;
    }
    if (nlev < 3) { for (int32_t v : nd) w.part[v] = -1; min_degree(*w.g, nd, out); continue; }
    // choose the separator level: among levels 1..nlev-2 the one minimising
    // |S| * imbalance penalty around the median
    ivec cnt(nlev, 0);
    for (int32_t v : comp) cnt[w.level[v]]++;
    int64_t total = (int64_t)comp.size(), acc = cnt[0];
    int32_t bestl = 1; double bestscore = 1e300;
    for (int32_t l = 1; l < nlev - 1; l++) {
      int64_t a = acc, b = total - acc - cnt[l];
      acc += cnt[l];
      if (a == 0 || b == 0) continue;
      double bal = (double)std::max(a, b) / (double)(a + b);  // 0.5 .. 1
      if (bal > 0.75) continue;
      double score = cnt[l] * (1.0 + 4.0 * (bal - 0.5));
      if (score < bestscore) { bestscore = score; bestl = l; }
    }
    if (bestscore == 1e300) bestl = nlev / 2;
    ivec A, B, S;
    for (int32_t v : comp) {
      int32_t l = w.level[v];
      if (l < bestl) A.push_back(v); else if (l > bestl) B.push_back(v); else S.push_back(v);
    }
    // separator nodes with no neighbour in B move to A (thinner separator)
    {
      ivec S2;
      for (int32_t v : S) {
        bool touchB = false;
        for (int32_t p = w.g->ptr[v]; p < w.g->ptr[v + 1] && !touchB; p++) {
          int32_t u = w.g->idx[p];
          if (w.part[u] == pid && w.level[u] == bestl + 1) touchB = true;
        }
        if (touchB) S2.push_back(v); else A.push_back(v);
      }
      S.swap(S2);
    }
    for (int32_t v : nd) w.part[v] = -1;
    std::sort(S.begin(), S.end());
    if (w.canon_leaves) {
      // a part that becomes a leaf is eliminated FROM ITS FAR END TOWARDS THIS SEPARATOR (BFS levels of the split: A lies
      // below the separator level, B above): the chain of fronts then never carries the separator's variables along.  In the
      // canonical order the part on the far side of the root would start next to the separator and drag it through every
      // front (two thirds more fronts on the band patterns).
      // (inside a level: in the direction of travel, judged by the distance of the labels from the root's — any tie-break is
      //  valid, this one keeps a band's chain monotone)
      const int32_t root = comp[0];
      auto dist = [&](int32_t v) { return v > root ? v - root : root - v; };
      std::sort(A.begin(), A.end(), [&](int32_t x, int32_t y) { return w.level[x] != w.level[y] ? w.level[x] < w.level[y] : dist(x) < dist(y); });
      std::sort(B.begin(), B.end(), [&](int32_t x, int32_t y) { return w.level[x] != w.level[y] ? w.level[x] > w.level[y] : dist(x) > dist(y); });
    } else {
      std::sort(A.begin(), A.end());
      std::sort(B.begin(), B.end());
    }
    st.push_back({std::move(S), true});   // emitted last
    st.push_back({std::move(B), false, w.canon_leaves});
    st.push_back({std::move(A), false, w.canon_leaves});  // processed first
  }
}

// ---------------------------------------------------------------------------
struct Symbolic {
  ivec perm, iperm;            // postordered elimination order
  ivec parent;                 // etree
  std::vector<ivec> lstruct;   // strictly-below structure of every column of L (sorted ascending)
  int64_t nnzL = 0;
};

// permuted adjacency split: lower neighbours (elim idx < j) for every j
void perm_lower_adj(const Graph& g, const ivec& iperm, std::vector<ivec>& low, std::vector<ivec>* up) {
  int32_t n = g.n;
  low.assign(n, ivec());
  if (up) up->assign(n, ivec());
  for (int32_t v = 0; v < n; v++) {
    int32_t a = iperm[v];
    for (int32_t p = g.ptr[v]; p < g.ptr[v + 1]; p++) {
      int32_t b = iperm[g.idx[p]];
      if (b < a) low[a].push_back(b); else if (up) (*up)[a].push_back(b);
    }
  }
}

void etree(const std::vector<ivec>& low, ivec& parent) {
  int32_t n = (int32_t)low.size();
  parent.assign(n, -1);
  ivec anc(n, -1);
  for (int32_t j = 0; j < n; j++)
    for (int32_t i0 : low[j]) {
      int32_t i = i0;
      while (i != -1 && i < j) {
        int32_t nx = anc[i];
        anc[i] = j;
        if (nx == -1) parent[i] = j;
        i = nx;
      }
    }
}

void postorder(const ivec& parent, ivec& post) {
  int32_t n = (int32_t)parent.size();
  ivec head(n, -1), next(n, -1);
  for (int32_t j = n - 1; j >= 0; j--) if (parent[j] >= 0) { next[j] = head[parent[j]]; head[parent[j]] = j; }
  post.clear(); post.reserve(n);
  ivec stack;
  for (int32_t r = 0; r < n; r++) {
    if (parent[r] >= 0) continue;
    stack.push_back(r);
    while (!stack.empty()) {
      int32_t v = stack.back();
      int32_t c = head[v];
      if (c >= 0) { head[v] = next[c]; stack.push_back(c); }
      else { post.push_back(v); stack.pop_back(); }
    }
  }
}

// full symbolic factorisation for an elimination order (returned postordered)
void symbolic(const Graph& g, const ivec& perm_in, Symbolic& S) {
  int32_t n = g.n;
  ivec iperm(n);
  for (int32_t k = 0; k < n; k++) iperm[perm_in[k]] = k;
  std::vector<ivec> low;
  perm_lower_adj(g, iperm, low, nullptr);
  ivec parent; etree(low, parent);
  ivec post; postorder(parent, post);
  S.perm.resize(n); S.iperm.resize(n);
  for (int32_t k = 0; k < n; k++) { S.perm[k] = perm_in[post[k]]; S.iperm[S.perm[k]] = k; }
  std::vector<ivec> up;
  perm_lower_adj(g, S.iperm, low, &up);
  etree(low, S.parent);
  S.lstruct.assign(n, ivec());
  std::vector<ivec> kids(n);
  ivec mark(n, -1);
  S.nnzL = 0;
  for (int32_t j = 0; j < n; j++) {
    ivec& L = S.lstruct[j];
    mark[j] = j;
    for (int32_t i : up[j]) if (mark[i] != j) { mark[i] = j; L.push_back(i); }
    for (int32_t c : kids[j])
      for (int32_t i : S.lstruct[c]) if (mark[i] != j) { mark[i] = j; L.push_back(i); }
    std::sort(L.begin(), L.end());
    if (!L.empty()) kids[L[0]].push_back(j);
    S.nnzL += (int64_t)L.size();
  }
}

// ---------------------------------------------------------------------------
// kernel cost model (wave-instructions, one 64-lane wave per problem)
inline double chunks(int64_t n) { return (double)((n + 63) / 64); }
double front_cost(int64_t npiv, int64_t nupd, int64_t indep = 0) {
  int64_t f = 1 + nupd + npiv;
  double c = 24.0 + 2.0 * chunks(tri(f)) + 3.0 * chunks(tri(f) - tri(1 + nupd)) + 3.0 * chunks(tri(1 + nupd));
  c += (double)indep * (12.0 + 6.0 * chunks(tri(f - indep)));
  for (int64_t i = f - indep - 1; i > nupd; i--) c += 12.0 + 6.0 * chunks(tri(i));
  // backward solve: panel load + per pivot dot/reduce
  c += 10.0 + 2.0 * chunks(tri(f) - tri(1 + nupd)) + 14.0 * (double)npiv;
  return c;
}
double extend_cost(int64_t nupd_child) { return 6.0 + 5.0 * chunks(tri(1 + nupd_child)); }

// ---- cost model of the register-front kernel (kernels2.hip), in cycles per wavefront (= 4 problems) ----
inline int64_t ceil4(int64_t x) { return (x + 3) & ~(int64_t)3; }
double front_cost2(int64_t npiv, int64_t nupd, int64_t /*indep*/ = 0) {
  const int64_t f = 1 + nupd + npiv;
  const int64_t TE = f <= 16 ? 16 : (f <= 32 ? 32 : 64);
  const double passes = (double)(TE / 16);
  double c = 1500.0 + 12.0 * (double)((tri(f) + 15) / 16) * (TE == 64 ? 4.0 : 1.0) + 50.0 * (double)((7 * npiv + 15) / 16);
  double el = (double)TE * 10.0 + (double)(nupd + 1) * 14.0;
  for (int64_t i = f - 1; i > nupd; i--) el += 260.0 + 35.0 * (double)ceil4(std::min<int64_t>(i - 1, TE - 1));
  c += passes * el * (TE == 64 ? 1.5 : 1.0);
  c += passes * (350.0 + 220.0 * (double)npiv + 4.0 * (double)TE);
  return c;
}
double extend_cost2(int64_t nupd_child) { return 120.0 + 35.0 * (double)((tri(1 + nupd_child) + 15) / 16); }

struct SNode {
  ivec icols;       // leading mutually independent pivots (merged single-pivot leaves)
  ivec cols;        // dependent pivot columns in elimination order (postordered labels)
  int32_t nupd = 0; // update rows
  int32_t parent = -1;
  ivec kids;
  bool alive = true;
  int64_t np() const { return (int64_t)icols.size() + (int64_t)cols.size(); }
};

}  // namespace

// ===========================================================================
int build_plan(Plan& P, int64_t N64, int64_t nnz, const int64_t* rows1, const int64_t* cols1,
               int64_t nvar, int64_t nequ, int64_t ncon, const Options& opt_in, std::string& msg) {
  const Options& opt = opt_in;
  const bool verbose = opt.verbose != 0 || getenv("CNL_VERBOSE") != nullptr;  // logging only: never changes a decision
  if (N64 <= 0 || nvar < 0 || nequ < 0 || ncon < 0 || nvar + nequ + ncon != N64) { msg = "bad dimensions: N != nvar+nequ+ncon"; return 2; }
  if (N64 + nnz >= (int64_t)1 << 30) { msg = "problem too large for 32-bit plan indices"; return 2; }
  if (nnz < nvar) { msg = "nnz < nvar: the last nvar COO entries must be the rho slots"; return 2; }
  const int32_t N = (int32_t)N64;
  P = Plan();
  P.N = N; P.nnz = nnz; P.nvar = nvar; P.nequ = nequ; P.ncon = ncon;
  P.rho_begin = (int32_t)(nnz - nvar);

  // ---- unique lower-triangular pattern + COO -> slot map -------------------
  std::vector<int64_t> key(nnz);
  for (int64_t k = 0; k < nnz; k++) {
    int64_t i = rows1[k] - 1, j = cols1[k] - 1;
    if (i < 0 || i >= N || j < 0 || j >= N) { msg = "COO index out of range at entry " + std::to_string(k); return 3; }
    if (i < j) { msg = "COO entry " + std::to_string(k) + " is in the upper triangle (rows < cols); the KKT pattern must be lower triangular"; return 3; }
    key[k] = j * (int64_t)N + i;
  }
  std::vector<int64_t> ukey(key);
  std::sort(ukey.begin(), ukey.end());
  ukey.erase(std::unique(ukey.begin(), ukey.end()), ukey.end());
  const int64_t nK = (int64_t)ukey.size();
  P.nnzK = nK;
  ivec slot(nnz);
  for (int64_t k = 0; k < nnz; k++) slot[k] = (int32_t)(std::lower_bound(ukey.begin(), ukey.end(), key[k]) - ukey.begin());
  ivec Ki(nK), Kj(nK);
  for (int64_t u = 0; u < nK; u++) { Kj[u] = (int32_t)(ukey[u] / N); Ki[u] = (int32_t)(ukey[u] % N); }

  // ---- symmetric graph -----------------------------------------------------
  Graph g; g.n = N; g.ptr.assign(N + 1, 0);
  for (int64_t u = 0; u < nK; u++) if (Ki[u] != Kj[u]) { g.ptr[Ki[u] + 1]++; g.ptr[Kj[u] + 1]++; }
  for (int32_t v = 0; v < N; v++) g.ptr[v + 1] += g.ptr[v];
  g.idx.resize(g.ptr[N]);
  {
    ivec nx(g.ptr.begin(), g.ptr.end() - 1);
    for (int64_t u = 0; u < nK; u++) if (Ki[u] != Kj[u]) { g.idx[nx[Ki[u]]++] = Kj[u]; g.idx[nx[Kj[u]]++] = Ki[u]; }
    for (int32_t v = 0; v < N; v++) std::sort(g.idx.begin() + g.ptr[v], g.idx.begin() + g.ptr[v + 1]);
  }

  // ---- constrained ordering --------------------------------------------------
  // stage A: residual nodes (pivot -1, mutually independent) first, except
  //          "dense" rows (AMD-style threshold) which are postponed;
  // stage B: x nodes, ordered on the Schur graph S = pattern(H + J'J);
  // stage C: multipliers and postponed residual rows.
  const int32_t nx_ = (int32_t)nvar, nr_ = (int32_t)nequ;
  const int32_t dense_thr = std::max(16, (int32_t)(10.0 * std::sqrt((double)N)));
  ivec stageA, stageC;
  std::vector<char> firstR(N, 0);
  for (int32_t r = nx_; r < nx_ + nr_; r++) {
    bool only_x = true;
    for (int32_t p = g.ptr[r]; p < g.ptr[r + 1]; p++) if (g.idx[p] >= nx_) { only_x = false; break; }
    if (only_x && g.deg(r) <= dense_thr) { stageA.push_back(r); firstR[r] = 1; }
    else stageC.push_back(r);
  }
  for (int32_t v = nx_ + nr_; v < N; v++) stageC.push_back(v);

  // Schur graph on x
  Graph sg; sg.n = nx_; sg.ptr.assign(nx_ + 1, 0);
  {
    // dedupe identical residual rows (e.g. a dense Jacobian): same neighbour list -> one clique
    std::unordered_map<uint64_t, std::vector<int32_t>> buckets;
    ivec rep;  // representative rows
    for (int32_t r : stageA) {
      uint64_t h = 1469598103934665603ull;
      for (int32_t p = g.ptr[r]; p < g.ptr[r + 1]; p++) { h ^= (uint64_t)g.idx[p] + 0x9e3779b97f4a7c15ull; h *= 1099511628211ull; }
      auto& b = buckets[h];
      bool dup = false;
      for (int32_t q : b) {
        if (g.deg(q) == g.deg(r) && std::equal(g.idx.begin() + g.ptr[r], g.idx.begin() + g.ptr[r + 1], g.idx.begin() + g.ptr[q])) { dup = true; break; }
      }
      if (!dup) { b.push_back(r); rep.push_back(r); }
    }
    std::vector<ivec> rows_of_x(nx_);
    for (int32_t r : rep) for (int32_t p = g.ptr[r]; p < g.ptr[r + 1]; p++) rows_of_x[g.idx[p]].push_back(r);
    ivec mark(nx_, -1);
    std::vector<ivec> adj(nx_);
    for (int32_t j = 0; j < nx_; j++) {
      mark[j] = j;
      for (int32_t p = g.ptr[j]; p < g.ptr[j + 1]; p++) { int32_t u = g.idx[p]; if (u < nx_ && mark[u] != j) { mark[u] = j; adj[j].push_back(u); } }
      for (int32_t r : rows_of_x[j])
        for (int32_t p = g.ptr[r]; p < g.ptr[r + 1]; p++) { int32_t u = g.idx[p]; if (mark[u] != j) { mark[u] = j; adj[j].push_back(u); } }
      std::sort(adj[j].begin(), adj[j].end());
      sg.ptr[j + 1] = sg.ptr[j] + (int32_t)adj[j].size();
    }
    sg.idx.resize(sg.ptr[nx_]);
    for (int32_t j = 0; j < nx_; j++) std::copy(adj[j].begin(), adj[j].end(), sg.idx.begin() + sg.ptr[j]);
  }

  // hubs of the x graph (degree far above the median) are ordered last among x
  ivec xs_all(nx_);
  std::iota(xs_all.begin(), xs_all.end(), 0);

  auto assemble_perm = [&](const ivec& xorder) {
    ivec perm; perm.reserve(N);
    perm.insert(perm.end(), stageA.begin(), stageA.end());
    perm.insert(perm.end(), xorder.begin(), xorder.end());
    perm.insert(perm.end(), stageC.begin(), stageC.end());
    return perm;
  };

  // Variant of an x order: every late node (multiplier / kept residual row) whose neighbours are all variables goes
  // right behind the last of them instead of to the very end.  Any symmetric permutation has the same inertia
  // (Sylvester), and the pivots of such a node are -delta minus a positive term, never zero when the x block is
  // definite; what changes is the fill: a multiplier that covers a contiguous block of a banded problem then stops
  // riding along in every later update matrix, so the fronts stay small up to the root.
  auto assemble_perm_early = [&](const ivec& xorder) {
    ivec posx(nx_, -1);
    for (int32_t i = 0; i < (int32_t)xorder.size(); i++) posx[xorder[i]] = i;
    std::vector<ivec> after(xorder.size());
    ivec tail;
    for (int32_t v : stageC) {
      int32_t last = -1;
      bool ok = g.deg(v) > 0;
      for (int32_t p = g.ptr[v]; p < g.ptr[v + 1] && ok; p++) {
        const int32_t u = g.idx[p];
        if (u >= nx_ || posx[u] < 0) ok = false; else last = std::max(last, posx[u]);
      }
      if (ok && last >= 0) after[last].push_back(v); else tail.push_back(v);
    }
    ivec perm; perm.reserve(N);
    perm.insert(perm.end(), stageA.begin(), stageA.end());
    for (int32_t i = 0; i < (int32_t)xorder.size(); i++) {
      perm.push_back(xorder[i]);
      perm.insert(perm.end(), after[i].begin(), after[i].end());
    }
    perm.insert(perm.end(), tail.begin(), tail.end());
    return perm;
  };

  const int relax = opt.relax >= 0 ? opt.relax : 4;

  // supernodes + amalgamation + cost for a candidate; returns the final
  // supernode column lists (in final elimination order) through `sn_out`
  struct Cand { std::string name; ivec perm; ivec sn_first, sn_indep; double cost = 0, cpath = 0; int64_t nnzL = 0, nnzL_exact = 0; bool m2 = false; };
  const int64_t merge_tri_cap = (int64_t)20000;  // do not grow LDS-sized fronts past this by relaxation
  auto evaluate = [&](const ivec& perm0, Cand& c) {
    Symbolic S; symbolic(g, perm0, S);
    c.nnzL_exact = S.nnzL;
    // maximal supernodes
    std::vector<SNode> sn;
    ivec sn_of(N);
    for (int32_t j = 0; j < N; j++) {
      bool join = j > 0 && S.parent[j - 1] == j && S.lstruct[j - 1].size() == S.lstruct[j].size() + 1;
      if (!join) { sn.emplace_back(); }
      sn.back().cols.push_back(j);
      sn_of[j] = (int32_t)sn.size() - 1;
    }
    int32_t ns = (int32_t)sn.size();
    for (int32_t s = 0; s < ns; s++) {
      int32_t last = sn[s].cols.back();
      sn[s].nupd = (int32_t)S.lstruct[last].size();
      sn[s].parent = S.parent[last] >= 0 ? sn_of[S.parent[last]] : -1;
      if (sn[s].parent >= 0) sn[sn[s].parent].kids.push_back(s);
    }
    // kernel selection: the register-front kernel needs every front of order <= 64
    int64_t fund_fmax = 0;
    for (int32_t s = 0; s < ns; s++) fund_fmax = std::max<int64_t>(fund_fmax, 1 + sn[s].nupd + sn[s].np());
    const bool m2 = fund_fmax <= 64 && opt.register_front;
    const int64_t fcap = m2 ? 64 : ((int64_t)1 << 40);
    auto fcost = [&](int64_t np_, int64_t nu_, int64_t ind_) { return m2 ? front_cost2(np_, nu_, ind_) : front_cost(np_, nu_, ind_); };
    auto ecost = [&](int64_t nu_) { return m2 ? extend_cost2(nu_) : extend_cost(nu_); };
    auto cost_of = [&](const SNode& x) { return fcost(x.np(), x.nupd, (int64_t)x.icols.size()); };
    // relaxed amalgamation, children before parents (supernodes are in postorder)
    for (int32_t p = 0; p < ns; p++) {
      if (!sn[p].alive) continue;
      // (1) single-pivot leaf children become leading independent pivots of p
      {
        ivec keep;
        for (int32_t cidx : sn[p].kids) {
          SNode& ch = sn[cidx];
          bool leaf1 = ch.kids.empty() && ch.icols.empty() && ch.cols.size() == 1;
          bool merged = false;
          if (leaf1) {
            int64_t fdep = 1 + sn[p].nupd + (int64_t)sn[p].cols.size();  // rows an independent pivot updates
            double before = fcost(1, ch.nupd, 0) + ecost(ch.nupd) + cost_of(sn[p]);
            double after = fcost(sn[p].np() + 1, sn[p].nupd, (int64_t)sn[p].icols.size() + 1);
            bool fits = (tri(1 + sn[p].nupd + sn[p].np() + 1) <= merge_tri_cap || (fdep - 1 - ch.nupd) * 4 <= fdep) &&
                        1 + sn[p].nupd + sn[p].np() + 1 <= fcap;
            if (after <= before && fits) {
              sn[p].icols.push_back(ch.cols[0]);
              ch.alive = false; merged = true;
            }
          }
          if (!merged) keep.push_back(cidx);
        }
        sn[p].kids.swap(keep);
      }
      // (2) dependent merges
      bool changed = true;
      while (changed) {
        changed = false;
        ivec ks = sn[p].kids;
        std::sort(ks.begin(), ks.end(), [&](int32_t a, int32_t b) { return sn[a].nupd > sn[b].nupd; });
        for (int32_t cidx : ks) {
          SNode& ch = sn[cidx];
          int64_t npc = ch.np(), npp = sn[p].np();
          int64_t extra = (int64_t)sn[p].cols.size() + sn[p].nupd - ch.nupd;  // explicit zeros per child column
          double before = cost_of(ch) + cost_of(sn[p]) + ecost(ch.nupd);
          double after = fcost(npc + npp, sn[p].nupd, (int64_t)(ch.icols.size() + sn[p].icols.size()));
          bool ok = m2 ? after <= before : ((extra <= relax && after <= before * 1.02) || after <= before * 0.9);
          if (tri(1 + sn[p].nupd + npc + npp) > merge_tri_cap && extra > 0) ok = false;
          if (1 + sn[p].nupd + npc + npp > fcap) ok = false;
          if (!ok) continue;
          // merge ch into p: ch's columns are eliminated right before p's
          ivec nc; nc.reserve(ch.cols.size() + sn[p].cols.size());
          nc.insert(nc.end(), ch.cols.begin(), ch.cols.end());
          nc.insert(nc.end(), sn[p].cols.begin(), sn[p].cols.end());
          sn[p].cols.swap(nc);
          sn[p].icols.insert(sn[p].icols.begin(), ch.icols.begin(), ch.icols.end());
          ivec nk;
          for (int32_t k : sn[p].kids) if (k != cidx) nk.push_back(k);
          for (int32_t k : ch.kids) { nk.push_back(k); sn[k].parent = p; }
          sn[p].kids.swap(nk);
          ch.alive = false; ch.kids.clear();
          changed = true;
          break;
        }
      }
    }
    // final order: DFS postorder over the merged tree; within a node: its columns
    ivec order; order.reserve(N);
    c.sn_first.clear(); c.sn_indep.clear();
    {
      // kids sorted so that the largest update matrix comes LAST
      std::vector<int32_t> stack; ivec it(ns, 0);
      for (int32_t s = 0; s < ns; s++) if (sn[s].alive)
        std::sort(sn[s].kids.begin(), sn[s].kids.end(), [&](int32_t a, int32_t b) {
          if (sn[a].nupd != sn[b].nupd) return sn[a].nupd < sn[b].nupd; return a < b; });
      for (int32_t r = 0; r < ns; r++) {
        if (!sn[r].alive || sn[r].parent >= 0) continue;
        stack.push_back(r);
        while (!stack.empty()) {
          int32_t v = stack.back();
          if (it[v] < (int32_t)sn[v].kids.size()) { stack.push_back(sn[v].kids[it[v]++]); }
          else {
            c.sn_first.push_back((int32_t)order.size());
            c.sn_indep.push_back((int32_t)sn[v].icols.size());
            for (int32_t col : sn[v].icols) order.push_back(S.perm[col]);
            for (int32_t col : sn[v].cols) order.push_back(S.perm[col]);
            stack.pop_back();
          }
        }
      }
      c.sn_first.push_back((int32_t)order.size());
    }
    c.perm.swap(order);
    // cost + nnzL with explicit zeros; critical path of the (merged) tree: supernodes are numbered children first
    c.cost = 0; c.nnzL = 0; c.cpath = 0; c.m2 = m2;
    std::vector<double> cp(ns, 0.0);
    for (int32_t s = 0; s < ns; s++) if (sn[s].alive) {
      int64_t np = sn[s].np(), nu = sn[s].nupd;
      c.cost += cost_of(sn[s]);
      double mine = cost_of(sn[s]);
      if (sn[s].parent >= 0) { c.cost += ecost(nu); mine += ecost(nu); }
      c.nnzL += np * nu + np * (np - 1) / 2;
      double below = 0;
      for (int32_t k : sn[s].kids) below = std::max(below, cp[k]);
      cp[s] = below + mine;
      c.cpath = std::max(c.cpath, cp[s]);
    }
  };

  // Candidates = (x order) x (multipliers last | early).  The x orders and their evaluations are independent of each other: they run
  // on host threads (round 6: the sweep was 0.3 s of a 0.65 s analysis of one cfg3-size system, serial), results land in a fixed
  // order so that the choice does not depend on the threads.  With opt.force_order only the named candidate is built.
  std::vector<Cand> cands;
  const bool try_early = !stageC.empty() && opt.early;
  struct Job { std::string name; std::function<void(ivec&)> make; };
  std::vector<Job> jobs;
  auto wanted = [&](const std::string& name) {
    return opt.force_order.empty() || opt.force_order == name || (try_early && opt.force_order == name + "+early");
  };
  auto add_job = [&](const std::string& name, std::function<void(ivec&)> make) { if (wanted(name)) jobs.push_back({name, std::move(make)}); };
  int mode = opt.order_mode;
  if (mode == 0 || mode < 0) add_job("canonical", [&](ivec& xo) { xo = xs_all; });
  double sdens = nx_ > 0 ? (double)sg.ptr[nx_] / ((double)nx_ * (double)std::max(1, nx_ - 1)) : 1.0;
  bool sparse_enough = nx_ > 8 && sdens < 0.4;
  if ((mode == 2 || mode < 0) && sparse_enough) add_job("md", [&](ivec& xo) { min_degree(sg, xs_all, xo); });
  ivec body, hubs;
  Graph bg;
  if ((mode == 1 || mode < 0) && sparse_enough) {
    // hubs: x nodes with degree > 8*median+16 are excluded from ND and go last among x
    ivec dsort(nx_);
    for (int32_t j = 0; j < nx_; j++) dsort[j] = sg.deg(j);
    std::nth_element(dsort.begin(), dsort.begin() + nx_ / 2, dsort.end());
    int32_t hub_thr = 8 * dsort[nx_ / 2] + 16;
    for (int32_t j = 0; j < nx_; j++) (sg.deg(j) > hub_thr ? hubs : body).push_back(j);
    bg = sg;
    if (!hubs.empty()) {  // drop hub edges from the working graph
      std::vector<char> ish(nx_, 0); for (int32_t h : hubs) ish[h] = 1;
      bg.ptr.assign(nx_ + 1, 0); bg.idx.clear();
      for (int32_t j = 0; j < nx_; j++) {
        if (!ish[j]) for (int32_t p = sg.ptr[j]; p < sg.ptr[j + 1]; p++) if (!ish[sg.idx[p]]) bg.idx.push_back(sg.idx[p]);
        bg.ptr[j + 1] = (int32_t)bg.idx.size();
      }
    }
    std::vector<int32_t> leaves;
    if (opt.nd_leaf > 0) leaves.push_back(opt.nd_leaf);
    else if (opt.latency) leaves = {32, 64, 96, 160, 256, 512};
    else leaves = {32, 96, 256, 768, 2048};
    for (int32_t leaf : leaves) {
      if (leaf >= (int32_t)body.size() && leaves.size() > 1 && leaf != leaves.front()) break;
      add_job("nd" + std::to_string(leaf), [&, leaf](ivec& xo) {
        NDWork w; w.g = &bg; w.part.assign(nx_, -1); w.level.assign(nx_, -1); w.leaf = leaf;
        xo.reserve(nx_);
        nd_rec(w, body, xo);
        xo.insert(xo.end(), hubs.begin(), hubs.end());
      });
    }
    // Mid-size batches (latency plans with few wavefront slots per group of problems): a handful of LARGE parts, each kept in
    // the canonical order — inside a part the tree is the chain of full fronts the throughput order has, the parts run as
    // whole-subtree tasks on different wavefronts, only the separators above them are serial.
    if (opt.latency && opt.nd_leaf <= 0 && opt.par < 512)
      for (int div : {2, 4, 8, 16, 32, 64, 128}) {
        const int32_t leaf = (int32_t)body.size() / div + 1;
        if (leaf < 64) break;
        add_job("ndc" + std::to_string(div), [&, leaf](ivec& xo) {
          NDWork w; w.g = &bg; w.part.assign(nx_, -1); w.level.assign(nx_, -1); w.leaf = leaf; w.canon_leaves = true;
          xo.reserve(nx_);
          nd_rec(w, body, xo);
          xo.insert(xo.end(), hubs.begin(), hubs.end());
        });
      }
  }
  if (jobs.empty()) jobs.push_back({"canonical", [&](ivec& xo) { xo = xs_all; }});
  {
    const size_t per = try_early ? 2 : 1;
    std::vector<Cand> slot(jobs.size() * per);
    std::vector<char> have(slot.size(), 0);
    std::atomic<size_t> next{0};
    auto work = [&]() {
      for (size_t k = next.fetch_add(1); k < jobs.size(); k = next.fetch_add(1)) {
        ivec xo;
        jobs[k].make(xo);
        if (opt.force_order.empty() || opt.force_order == jobs[k].name) {
          Cand& c = slot[k * per]; c.name = jobs[k].name;
          evaluate(assemble_perm(xo), c);
          have[k * per] = 1;
        }
        if (try_early && (opt.force_order.empty() || opt.force_order == jobs[k].name + "+early")) {
          Cand& c = slot[k * per + 1]; c.name = jobs[k].name + "+early";
          evaluate(assemble_perm_early(xo), c);
          have[k * per + 1] = 1;
        }
      }
    };
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const size_t nthr = opt.threads > 0 ? std::min<size_t>(jobs.size(), (size_t)opt.threads) : std::min<size_t>({jobs.size(), (size_t)hw, (size_t)16});
    if (nthr <= 1 || N < 2000) work();   // (small systems: a thread start costs more than their candidates)
    else {
      std::vector<std::thread> th;
      for (size_t t = 0; t + 1 < nthr; t++) th.emplace_back(work);
      work();
      for (auto& t : th) t.join();
    }
    for (size_t k = 0; k < slot.size(); k++) if (have[k]) cands.push_back(std::move(slot[k]));
  }
  if (cands.empty()) {   // (a forced name that no candidate carries: the canonical order)
    Cand c; c.name = "canonical";
    evaluate(assemble_perm(xs_all), c);
    cands.push_back(std::move(c));
  }
  // throughput plans minimise the total work; latency plans the critical path plus the work spread over the wavefront
  // slots a group of four problems can use (only orders the register-front kernel can run are staged)
  // Latency score = the longer of the critical path and the work spread over the wavefront slots of a group of problems.
  // A task costs a wavefront start per phase on top of its fronts (prologue, pipeline fill, hand-over through global memory;
  // fitted on cfg3 at 2048 problems: 495 small tasks 4.6 ms, 8 large ones 3.4 ms for the same model work), and when the parts
  // of all groups together are more than the machine holds at once, a last partial round costs a whole one (measured: eight
  // parts at 1536 problems = 1.5 rounds run at 496 k systems/s, thirty-two parts = 6 rounds at 586 k).
  auto ntasks_est = [&](const Cand& c) {
    if (c.name.rfind("ndc", 0) == 0) return 2.0 * atoi(c.name.c_str() + 3);
    return (double)(c.sn_first.size() - 1) / 3.6;
  };
  auto score = [&](const Cand& c) {
    if (!opt.latency) return c.cost;
    if (!c.m2) return 1e30 + c.cost;
    const double par = opt.slots > 0 ? std::max(1.0, opt.slots) : (double)std::max(1, opt.par);
    const double rounds = 0.5 * ntasks_est(c) / par;  // bottom tasks per slot
    const double fill = rounds > 1.0 ? std::ceil(rounds - 1e-9) / rounds : 1.0;
    // (a wavefront that has its SIMD to itself runs 1.37 times faster than two sharing one: with fewer parts than slots the
    //  critical path shrinks accordingly)
    double alone = 0.73 + 0.27 * std::min(1.0, rounds);
    // The bidirectional chain (two tasks of ~500 full fronts per group) gains more than that from a SIMD of its own, and keeps some of
    // it up to two wavefronts per SIMD (round 3, lean kernel, cfg3: 2.50 / 2.57 / 2.62 / 3.08 / 3.30 / 4.11 ms at 1024 / 1536 / 2048 /
    // 2560 / 3072 / 4096 problems = 0.48 ... 0.82 of the model's path).  The other candidates' model costs run ~10 % high against it
    // (their fronts are smaller than the model's unit), hence the handicap: the chain is taken where it wins clearly — from ~1800
    // problems on (2048: 781 k systems/s against 592 k with eight parts) instead of from 3072.
    // (only from ~0.9 wavefronts per SIMD on: below that the short chains of small systems lose to eight parts — cfg4 at 1280
    //  problems 3.9 M against 5.2 M systems/s)
    if (c.name.rfind("ndc2", 0) == 0 && rounds >= 0.43) alone = 1.10 * (0.49 + 0.33 * std::min(1.0, (rounds - 0.4) / 0.6));
    return std::max(c.cpath * alone, (c.cost + 3.5e4 * ntasks_est(c)) / par * fill) + 1e-3 * c.cost;
  };
  size_t best = 0;
  for (size_t i = 1; i < cands.size(); i++) if (score(cands[i]) < score(cands[best])) best = i;
  if (!opt.force_order.empty())  // experiments: pick a candidate by name
    for (size_t i = 0; i < cands.size(); i++) if (cands[i].name == opt.force_order) best = i;
  if (verbose) {
    for (auto& c : cands)
      fprintf(stderr, "[cnl] order %-12s cost %.3e path %.3e nnzL %lld (exact %lld) fronts %zu\n", c.name.c_str(), c.cost, c.cpath,
              (long long)c.nnzL, (long long)c.nnzL_exact, c.sn_first.size() - 1);
    fprintf(stderr, "[cnl] chose %s%s\n", cands[best].name.c_str(), opt.latency ? " (latency plan)" : "");
  }
  Cand& C = cands[best];
  if (!opt.split_positions.empty()) {
    // cut the named supernodes (a cut inside a supernode only makes its first part the child of its second part)
    ivec nf, ni;
    const int32_t ns0 = (int32_t)C.sn_first.size() - 1;
    for (int32_t s0 = 0; s0 < ns0; s0++) {
      const int32_t a = C.sn_first[s0], c = C.sn_first[s0 + 1];
      ivec cuts;
      for (int32_t q : opt.split_positions) if (q > a && q < c) cuts.push_back(q);
      std::sort(cuts.begin(), cuts.end());
      cuts.erase(std::unique(cuts.begin(), cuts.end()), cuts.end());
      int32_t start = a;
      bool first = true;
      for (size_t k = 0; k <= cuts.size(); k++) {
        const int32_t end = k < cuts.size() ? cuts[k] : c;
        nf.push_back(start);
        ni.push_back(first ? std::min<int32_t>(C.sn_indep[s0], end - start) : 0);  // independence is known for the leading pivots only
        first = false;
        start = end;
      }
    }
    nf.push_back(C.sn_first[ns0]);
    C.sn_first.swap(nf);
    C.sn_indep.swap(ni);
  }
  P.order_name = C.name; P.cost = C.cost; P.cpath = C.cpath; P.nnzL = C.nnzL; P.nnzL_exact = C.nnzL_exact;
  P.perm = C.perm;
  P.iperm.assign(N, 0);
  for (int32_t k = 0; k < N; k++) P.iperm[P.perm[k]] = k;
  const int32_t ns = (int32_t)C.sn_first.size() - 1;
  P.nsuper = ns;

  // ---- final structures on the chosen order -----------------------------------
  // exact column structures in the final labelling (the order is already a
  // postorder of its own etree up to sibling permutations; recompute directly)
  std::vector<ivec> low, up;
  perm_lower_adj(g, P.iperm, low, &up);
  ivec sn_of(N);
  for (int32_t s = 0; s < ns; s++) for (int32_t j = C.sn_first[s]; j < C.sn_first[s + 1]; j++) sn_of[j] = s;
  // front rows: union of the below-front structure of all pivot columns, built bottom-up
  std::vector<ivec> frows(ns);  // ascending elimination indices, all > last pivot
  std::vector<ivec> skids(ns);
  ivec sparent(ns, -1);
  {
    ivec mark(N, -1);
    for (int32_t s = 0; s < ns; s++) {
      int32_t lastp = C.sn_first[s + 1] - 1;
      ivec& R = frows[s];
      for (int32_t j = C.sn_first[s]; j <= lastp; j++)
        for (int32_t i : up[j]) if (i > lastp && mark[i] != s) { mark[i] = s; R.push_back(i); }
      for (int32_t c : skids[s])
        for (int32_t i : frows[c]) if (i > lastp && mark[i] != s) { mark[i] = s; R.push_back(i); }
      std::sort(R.begin(), R.end());
      if (!R.empty()) { sparent[s] = sn_of[R[0]]; skids[sparent[s]].push_back(s); }
    }
    // consistency: children must precede parents and be contiguous subtrees
    for (int32_t s = 0; s < ns; s++) if (sparent[s] >= 0 && sparent[s] <= s) { msg = "internal: supernode order is not a postorder"; return 9; }
  }
  // verify postorder contiguity (needed by the static stack layout): the
  // children of s, in index order, must tile [first descendant, s)
  {
    ivec first_desc(ns);
    for (int32_t s = 0; s < ns; s++) {
      first_desc[s] = s;
      int32_t expect = s;
      for (size_t k = skids[s].size(); k-- > 0;) {
        int32_t c = skids[s][k];
        if (c != expect - 1) { msg = "internal: supernodal tree is not postordered"; return 9; }
        expect = first_desc[c];
      }
      first_desc[s] = expect;
    }
  }

  P.fronts.assign(ns, FrontHdr());
  P.child_idx.clear(); P.rel_idx.clear();
  int64_t lptr = 0; int32_t fmax = 0, panel_max = 0; double flops = 0;
  for (int32_t s = 0; s < ns; s++) {
    FrontHdr& F = P.fronts[s];
    std::memset(&F, 0, sizeof(F));
    F.npiv = C.sn_first[s + 1] - C.sn_first[s];
    F.nupd = (int32_t)frows[s].size();
    F.first_piv = C.sn_first[s];
    F.parent = sparent[s];
    int64_t f = 1 + (int64_t)F.nupd + F.npiv;
    if (tri(f) >= ((int64_t)1 << 30)) { msg = "front too large"; return 2; }
    fmax = std::max<int32_t>(fmax, (int32_t)f);
    int64_t panel = tri(f) - tri(1 + F.nupd);
    panel_max = (int32_t)std::max<int64_t>(panel_max, panel);
    F.lptr_lo = (int32_t)(lptr & 0x7fffffff); F.lptr_hi = (int32_t)(lptr >> 31);
    lptr += panel;
    F.child_begin = (int32_t)P.child_idx.size();
    for (int32_t c : skids[s]) P.child_idx.push_back(c);
    F.child_end = (int32_t)P.child_idx.size();
    F.indep = C.sn_indep[s];
    flops += (double)F.indep * (double)tri(f - F.indep);
    for (int64_t i = f - F.indep - 1; i > F.nupd; i--) flops += (double)tri(i);
  }
  P.lsize = lptr; P.fmax = fmax; P.panel_max = panel_max; P.flops = flops;

  // local index of an elimination index inside front s
  auto local_of = [&](int32_t s, int32_t e) -> int32_t {
    const FrontHdr& F = P.fronts[s];
    int32_t f = 1 + F.nupd + F.npiv;
    if (e >= F.first_piv && e < F.first_piv + F.npiv) return f - 1 - (e - F.first_piv);
    const ivec& R = frows[s];
    auto it = std::lower_bound(R.begin(), R.end(), e);
    if (it == R.end() || *it != e) return -1;
    int32_t k = (int32_t)(it - R.begin());       // ascending position
    return 1 + (F.nupd - 1 - k);                 // descending local order
  };
  // rel maps (child -> parent)
  for (int32_t s = 0; s < ns; s++) {
    FrontHdr& F = P.fronts[s];
    F.rel_begin = (int32_t)P.rel_idx.size();
    P.rel_idx.push_back(0);
    if (F.parent >= 0) {
      for (int32_t l = 1; l <= F.nupd; l++) {
        int32_t e = frows[s][F.nupd - l];
        int32_t q = local_of(F.parent, e);
        if (q < 0) { msg = "internal: child row missing in parent front"; return 9; }
        P.rel_idx.push_back(q);
      }
    } else {
      for (int32_t l = 1; l <= F.nupd; l++) P.rel_idx.push_back(0);
    }
  }

  // ---- assembly maps -------------------------------------------------------------
  {
    // rank of every COO entry among the entries of its slot, in COO order
    ivec cnt(nK, 0), rank(nnz);
    int32_t maxrank = 0;
    for (int64_t k = 0; k < nnz; k++) { rank[k] = cnt[slot[k]]++; maxrank = std::max(maxrank, rank[k]); }
    // (front, pos) of every slot
    ivec sfront(nK), spos(nK);
    for (int64_t u = 0; u < nK; u++) {
      int32_t a = P.iperm[Ki[u]], b = P.iperm[Kj[u]];
      int32_t lo = std::min(a, b), hi = std::max(a, b);
      int32_t s = sn_of[lo];
      int32_t li = local_of(s, lo), lj = local_of(s, hi);
      if (li < 0 || lj < 0 || lj > li) { msg = "internal: assembly position"; return 9; }
      sfront[u] = s; spos[u] = (int32_t)(tri(li) + lj);
    }
    // bucket entries by (front, rank)
    struct E { int32_t front, rank, pos, src; };
    std::vector<E> es; es.reserve(nnz + N);
    for (int64_t k = 0; k < nnz; k++) es.push_back({sfront[slot[k]], rank[k], spos[slot[k]], (int32_t)k});
    if (opt.with_rhs_row)
      for (int32_t e = 0; e < N; e++) {
        int32_t s = sn_of[e];
        es.push_back({s, 0, (int32_t)tri(local_of(s, e)), (int32_t)(nnz + P.perm[e])});
      }
    std::sort(es.begin(), es.end(), [](const E& a, const E& b) {
      if (a.front != b.front) return a.front < b.front;
      if (a.rank != b.rank) return a.rank < b.rank;
      return a.src < b.src;
    });
    P.seg_ptr.clear(); P.asm_pos.resize(es.size()); P.asm_src.resize(es.size());
    size_t i = 0;
    for (int32_t s = 0; s < ns; s++) {
      P.fronts[s].seg_begin = (int32_t)P.seg_ptr.size();
      while (i < es.size() && es[i].front == s) {
        int32_t r = es[i].rank;
        P.seg_ptr.push_back((int32_t)i);
        while (i < es.size() && es[i].front == s && es[i].rank == r) { P.asm_pos[i] = es[i].pos; P.asm_src[i] = es[i].src; i++; }
      }
      P.fronts[s].seg_end = (int32_t)P.seg_ptr.size();
    }
    P.seg_ptr.push_back((int32_t)es.size());
  }

  // ---- static stack layout ----------------------------------------------------------
  {
    int64_t sp = 0, peak = 0;
    for (int32_t s = 0; s < ns; s++) {
      FrontHdr& F = P.fronts[s];
      int64_t base = sp;
      if (F.child_end > F.child_begin) base = P.fronts[P.child_idx[F.child_begin]].ubase;
      int64_t f = 1 + (int64_t)F.nupd + F.npiv;
      F.foff = (int32_t)sp;
      peak = std::max(peak, sp + tri(f));
      F.ubase = (int32_t)base;
      sp = base + tri(1 + F.nupd);
      if (peak >= ((int64_t)1 << 30)) { msg = "work stack too large"; return 2; }
    }
    P.fwd_peak = (int32_t)peak;
    int64_t bpeak = 0;
    for (int32_t s = ns - 1; s >= 0; s--) {
      FrontHdr& F = P.fronts[s];
      int64_t f = 1 + (int64_t)F.nupd + F.npiv;
      if (F.parent < 0) F.xoff = 0;
      else {
        const FrontHdr& Pp = P.fronts[F.parent];
        bool first_child = P.child_idx[Pp.child_begin] == s;
        F.xoff = first_child ? Pp.xoff : Pp.xoff + (1 + Pp.nupd + Pp.npiv);
      }
      bpeak = std::max(bpeak, (int64_t)F.xoff + f);
    }
    P.bwd_peak = (int32_t)bpeak;
  }

  // ---- v2 streams (register-front kernel) ---------------------------------------------
  P.v2_ok = opt.with_rhs_row && P.fmax <= 64 && opt.register_front;
  // ---- tasks of the staged execution: maximal subtrees of at most task_cap fronts at the bottom (postorder makes a
  //      subtree a contiguous range of fronts), every front above the cut on its own; stage = 1 + latest child stage
  ivec task_first(ns, 0), task_root(ns, 0);
  P.tasks.clear(); P.stage_ptr.clear();
  if (opt.latency && P.v2_ok && ns > 1) {
    // measured (cfg3): 8 fronts per bottom task are best for a handful of problems (every task has a wavefront slot of its own),
    // 12 from 64 problems on
    int cap = opt.task_cap > 0 ? opt.task_cap : (opt.par >= 256 ? 8 : 12);
    // orders with a few large canonical parts ("ndc"): a part is one task
    if (opt.task_cap <= 0 && C.name.rfind("ndc", 0) == 0) cap = std::max(cap, (int)(1.5 * ns / std::max(1, atoi(C.name.c_str() + 3))));
    ivec nsub(ns, 1), fdesc(ns), stage(ns, 0);
    for (int32_t s = 0; s < ns; s++) {
      fdesc[s] = s;
      for (int32_t c : skids[s]) { nsub[s] += nsub[c]; fdesc[s] = std::min(fdesc[s], fdesc[c]); }
    }
    std::vector<Task> ts;
    for (int32_t s = 0; s < ns; s++) {
      const bool bottom = nsub[s] <= cap;
      if (bottom) {
        stage[s] = 0;
        if (sparent[s] < 0 || nsub[sparent[s]] > cap) ts.push_back({0, fdesc[s], s + 1, 0, 0, sparent[s] < 0 ? 1 : 0, -1, 0});
      } else {
        int32_t st = 0;
        for (int32_t c : skids[s]) st = std::max(st, stage[c]);
        stage[s] = st + 1;
        ts.push_back({stage[s], s, s + 1, 0, 0, sparent[s] < 0 ? 1 : 0, -1, 0});
      }
    }
    if (ts.size() > 1) {
      std::stable_sort(ts.begin(), ts.end(), [](const Task& x, const Task& y) { return x.stage < y.stage; });
      P.tasks = ts;
      {  // dependencies between tasks (the dataflow execution waits on them instead of on a launch per stage)
        ivec task_of(ns, -1);
        for (size_t k = 0; k < P.tasks.size(); k++) for (int32_t f = P.tasks[k].f0; f < P.tasks[k].f1; f++) task_of[f] = (int32_t)k;
        for (Task& t : P.tasks) {
          const int32_t pf = sparent[t.f1 - 1];
          t.parent = pf >= 0 ? task_of[pf] : -1;
          if (t.parent >= 0) P.tasks[t.parent].nchild++;
        }
      }
      for (const Task& t : P.tasks) { task_first[t.f0] = 1; task_root[t.f1 - 1] = 1; }
      const int32_t nst = P.tasks.back().stage + 1;
      P.stage_ptr.assign(nst + 1, 0);
      for (const Task& t : P.tasks) P.stage_ptr[t.stage + 1]++;
      for (int32_t q = 0; q < nst; q++) P.stage_ptr[q + 1] += P.stage_ptr[q];
      if (verbose)
        fprintf(stderr, "[cnl] staged: %zu tasks in %d stages (cap %d fronts), first stage %d tasks\n", P.tasks.size(), nst, cap, P.stage_ptr[1]);
    }
  }
  const bool staged = !P.tasks.empty();
  if (P.v2_ok) {
    const int64_t ubig_thr = tri(opt.ubig > 0 ? opt.ubig : 17);  // update matrices above this size live in global scratch
    const int32_t wait_thr = opt.wait_thr >= 0 ? opt.wait_thr : 2;
    ivec uoff2(ns, 0), uglob(ns, 0), fsglob(ns, 0), fsoff2(ns, 0), cls(ns, 16);
    int64_t spL = 0, spG = 0, peakL = 0, peakG = 0, fsmax = 0;
    for (int32_t s = 0; s < ns; s++) {
      const FrontHdr& F = P.fronts[s];
      int64_t f = 1 + (int64_t)F.nupd + F.npiv;
      cls[s] = f <= 16 ? 16 : (f <= 32 ? 32 : 64);
      P.ncls[cls[s] == 16 ? 0 : (cls[s] == 32 ? 1 : 2)]++;
    }
    // fronts of order 17..32 are staged in LDS only when they are common; a few of them would
    // otherwise set the LDS footprint of every problem
    const bool fs32_lds = (int64_t)P.ncls[1] * 20 > ns;
    int64_t bumpG = 0;  // staged plans: every global slot is used once (tasks run concurrently, a stack would be shared)
    for (int32_t s = 0; s < ns; s++) {
      const FrontHdr& F = P.fronts[s];
      int64_t f = 1 + (int64_t)F.nupd + F.npiv;
      if (staged && task_first[s]) spL = 0;  // a task starts with an empty LDS stack
      int64_t baseL = spL, baseG = spG;
      bool seenL = false, seenG = false;
      for (int32_t ci = F.child_begin; ci < F.child_end; ci++) {
        int32_t c = P.child_idx[ci];
        if (uglob[c]) { if (!seenG) { baseG = uoff2[c]; seenG = true; } }
        else { if (!seenL) { baseL = uoff2[c]; seenL = true; } }
      }
      fsglob[s] = cls[s] == 64 || (cls[s] == 32 && !fs32_lds);
      int64_t tu = tri(1 + F.nupd);
      // update matrices that wait long for their parent (left siblings of big subtrees) would pin LDS:
      // they go to the global scratch; only the ones consumed within the next few fronts stay in LDS
      const int32_t wait = F.parent >= 0 ? F.parent - s - 1 : 0;
      uglob[s] = tu > ubig_thr || wait > wait_thr;
      if (staged) {
        // a task hands its root's update matrix to the parent's task through the global scratch; inside a task everything
        // that fits stays in LDS (tasks are short)
        uglob[s] = task_root[s] || tu > ubig_thr;
        if (fsglob[s]) { fsoff2[s] = (int32_t)bumpG; bumpG += (tri(f) + 1) & ~(int64_t)1; }
        else fsmax = std::max(fsmax, tri(f));
        // + class: the kernel stores an update matrix as rows of TE lanes — TE = 16, 32 or 64, the class of the front (kernels2.hip,
        // CNL_USTG / CNL_DPP_USTG: all lanes of a row store) —, so the last rows run up to TE - 1 doubles past its end.  (Until round 5 the
        // pad was 16 whatever the class: the out-of-line classes overwrote up to 47 doubles of the NEXT slot — another task's update
        // matrix or staging triangle, in use by another wavefront at the same time — and, from the last slot, memory past the problem's
        // scratch: the history-dependent wrong decisions and memory faults of profiles/HISTORY.md 4b item 8.)
        if (uglob[s]) { uoff2[s] = (int32_t)bumpG; bumpG += (tu + cls[s] + 1) & ~(int64_t)1; spL = baseL; }
        else { uoff2[s] = (int32_t)baseL; spL = baseL + tu; peakL = std::max(peakL, spL); }
        peakG = bumpG;
        if (bumpG >= ((int64_t)1 << 30)) { msg = "global scratch of the staged plan too large"; return 2; }
        continue;
      }
      if (fsglob[s]) {
        // the staging triangle must not overlap the slot its own update matrix is written to
        // (rows >= 32 of the update matrix are stored while rows < 32 are still read from staging)
        int64_t fo = uglob[s] ? std::max(spG, baseG + tu) : spG;
        fsoff2[s] = (int32_t)fo;
        peakG = std::max(peakG, fo + tri(f));
      } else fsmax = std::max(fsmax, tri(f));
      if (uglob[s]) { uoff2[s] = (int32_t)baseG; spG = baseG + tu; spL = baseL; peakG = std::max(peakG, spG); }
      else { uoff2[s] = (int32_t)baseL; spL = baseL + tu; spG = baseG; peakL = std::max(peakL, spL); }
    }
    P.u2_peak = (int32_t)((peakL + 1) & ~(int64_t)1);
    P.fs2_max = (int32_t)((fsmax + 1) & ~(int64_t)1);
    P.gs_doubles = (peakG + 1) & ~(int64_t)1;
    P.v2_cls = cls; P.v2_fsglob = fsglob; P.v2_uglob = uglob; P.v2_uoff = uoff2; P.v2_fsoff = fsoff2;
    for (int32_t s = 0; s < ns; s++) if (cls[s] == 16 && !fsglob[s]) fsmax = std::max<int64_t>(fsmax, FAST_IMG_DOUBLES);
    P.fs2_max = (int32_t)((fsmax + 1) & ~(int64_t)1);
    write_forward_records(P, nullptr);
    // backward records, reverse post-order
    P.brec.clear(); P.brec_maxlen = 0;
    for (int32_t s = ns - 1; s >= 0; s--) {
      const FrontHdr& F = P.fronts[s];
      size_t r0 = P.brec.size();
      P.brec.resize(r0 + B_HDR, 0);
      const bool px_global = staged && task_root[s] && F.parent >= 0;  // the parent is solved by another task
      if (px_global) {
        P.brec.push_back(0);
        for (int32_t l = 1; l <= F.nupd; l++) P.brec.push_back(P.perm[frows[s][F.nupd - l]]);  // solution components of the update rows
      } else
        for (int32_t l = 0; l <= F.nupd; l++) P.brec.push_back(P.rel_idx[F.rel_begin + l]);
      const int32_t f = 1 + F.nupd + F.npiv;
      for (int32_t i = F.nupd + 1; i < f; i++) P.brec.push_back(P.perm[F.first_piv + (f - 1 - i)]);
      while ((P.brec.size() - r0) % 4) P.brec.push_back(0);
      int32_t* H = P.brec.data() + r0;
      H[B_NPIV] = F.npiv; H[B_NUPD] = F.nupd; H[B_RECLEN] = (int32_t)(P.brec.size() - r0); H[B_XOFF] = F.xoff;
      H[B_PXOFF] = px_global ? (int32_t)B_PX_GLOBAL : (F.parent >= 0 ? P.fronts[F.parent].xoff : (int32_t)B_PX_NONE);
      H[B_LPTR_LO] = F.lptr_lo; H[B_LPTR_HI] = F.lptr_hi; H[B_CLS] = cls[s];
      P.brec_maxlen = std::max(P.brec_maxlen, H[B_RECLEN]);
    }
    finalize_tasks(P);
  }
  msg.clear();
  return 0;
}

void finalize_tasks(Plan& P) {
  if (P.tasks.empty()) return;
  const int32_t ns = P.nsuper;
  ivec rstart(ns + 1, 0), bstart(ns + 1, 0);
  size_t r0 = 0;
  for (int32_t s = 0; s < ns; s++) { rstart[s] = (int32_t)r0; r0 += (size_t)P.rec[r0 + R_RECLEN]; }
  size_t b0 = 0;
  for (int32_t s = ns - 1; s >= 0; s--) { bstart[s] = (int32_t)b0; b0 += (size_t)P.brec[b0 + B_RECLEN]; }
  for (Task& t : P.tasks) { t.rec_off = rstart[t.f0]; t.brec_off = bstart[t.f1 - 1]; }
}


// ---------------------------------------------------------------------------------------------
int write_forward_records(Plan& P, const DirectLists* D) {
  const int32_t ns = P.nsuper;
  const ivec &cls = P.v2_cls, &fsglob = P.v2_fsglob, &uglob = P.v2_uglob, &uoff2 = P.v2_uoff, &fsoff2 = P.v2_fsoff;
  // fast fronts (order <= 16, LDS staging): packed triangle + FAST_IMG_DOUBLES - FAST_IMG_TRI slots for the padding entries
  std::vector<int32_t> rec;
  int32_t rec_maxlen = 0;
  int64_t st_raw = 0, st_prod = 0, st_asm = 0, st_rawmax = 0, st_prodmax = 0, st_asmmax = 0;
  // every condensed residual pivot d_r is "owned" by the first front that stages it: that front counts it in the inertia
  std::unordered_map<int32_t, char> d_claimed;
  P.rows_fronts = 0; P.listprod_fronts = 0;
  P.rows_overflow.clear();
  // Structure of the update matrix every fast front leaves behind (row a: bit b set <=> entry (a, b) can be non-zero), for the
  // band form of its parents (below); fronts of the other classes count as dense.
  std::vector<std::array<uint16_t, 16>> ustruct(ns);
  std::vector<char> udense(ns, 1);
  P.band_fronts = 0;
  for (int32_t s = 0; s < ns; s++) {
    const FrontHdr& F = P.fronts[s];
    size_t r0 = rec.size();
    rec.resize(r0 + R_HDR, 0);
    const bool strided = cls[s] == 16 && !fsglob[s];
    // plain entries (pos, src) grouped in rounds: all positions of a round are distinct, rounds are applied in
    // order (COO-order sums of duplicates); every round is padded to a multiple of 16 with entries that add
    // some value to an UNUSED slot (fast fronts: the slots behind the triangle, distinct positions)
    struct PE { int32_t round, src, pos; };
    std::vector<PE> pes;
    struct PR { int32_t pos, a, b, d; };
    std::vector<PR> prs;
    {
      std::unordered_map<int32_t, int32_t> occ;  // position -> plain entries so far
      for (int32_t r = F.seg_begin; r < F.seg_end; r++)
        for (int32_t e = P.seg_ptr[r]; e < P.seg_ptr[r + 1]; e++) {
          const int32_t pos = P.asm_pos[e];
          const int32_t src = P.asm_src[e];
          if (!D) { pes.push_back({occ[pos]++, src, pos}); continue; }
          for (int32_t c = D->c_ptr[src]; c < D->c_ptr[src + 1]; c++) {
            if (D->c_b[c] < 0) pes.push_back({occ[pos]++, D->c_a[c], pos});
            else prs.push_back({pos, D->c_a[c], D->c_b[c], D->c_d[c]});
          }
        }
    }
    // entries that read the matrix values come first, entries that read the right-hand side last (each group in
    // rounds padded to 16): a round of the hot path gathers from ONE array, with a wave-uniform base address
    const int32_t nnz_thr = D ? D->nnz_outer : (int32_t)P.nnz;  // sources >= nnz_thr address the right-hand side
    ivec asrc, apos;
    int32_t nasmv = 0;
    for (int grp = 0; grp < 2; grp++) {
      std::vector<PE> sel;
      for (auto& e : pes) if ((e.src >= nnz_thr) == (grp == 1)) sel.push_back(e);
      // duplicate rounds are counted inside the group (positions of the two groups are disjoint: rhs row vs the rest)
      {
        std::unordered_map<int32_t, int32_t> occ;
        for (auto& e : sel) e.round = occ[e.pos]++;
      }
      std::stable_sort(sel.begin(), sel.end(), [](const PE& x, const PE& y) { return x.round < y.round || (x.round == y.round && x.src < y.src); });
      const int32_t dsrc = grp == 1 ? nnz_thr : 0;  // padding reads entry 0 of the group's array
      for (size_t i = 0; i < sel.size(); i++) {
        if (i > 0 && sel[i].round != sel[i - 1].round) {
          int32_t dk = 0;
          while (asrc.size() % 16) { asrc.push_back(dsrc); apos.push_back(strided ? FAST_IMG_TRI + (dk++ % 16) : 0); }
        }
        asrc.push_back(sel[i].src); apos.push_back(sel[i].pos);
      }
      { int32_t dk = 0; while (asrc.size() % 16) { asrc.push_back(dsrc); apos.push_back(strided ? FAST_IMG_TRI + (dk++ % 16) : 0); } }
      if (grp == 0) nasmv = (int32_t)asrc.size();
    }
    // products of one position go to different rounds of 16 (same-address LDS atomics of one instruction serialise):
    // order by (occurrence of the position, position)
    {
      std::unordered_map<int32_t, int32_t> occ;
      std::vector<std::pair<int64_t, size_t>> key(prs.size());
      for (size_t i = 0; i < prs.size(); i++) key[i] = {((int64_t)occ[prs[i].pos]++ << 32) | (uint32_t)prs[i].pos, i};
      std::sort(key.begin(), key.end());
      std::vector<PR> q(prs.size());
      for (size_t i = 0; i < prs.size(); i++) q[i] = prs[key[i].second];
      prs.swap(q);
    }
    // ---- band form (plan.h): symbolic elimination of the front.  S[a] = columns b <= a where the assembled front can hold a
    // non-zero: plain entries, condensation products, the children's update matrices through their extend-add maps.  Pivot i
    // (from the top) updates row a only if entry (i, a) is in S; the updates fill S[a] |= S[i] (columns <= a).  When the row of
    // every pivot lies inside [the nfix lowest columns] + [the CNL band columns right below the pivot], the record says so and
    // the kernel runs the elimination that does not contain the other updates (kernels2.hip, eliminate16_dpp<LATE, BNF>).
    int32_t band_word = 0;
    if (strided) {
      const int32_t f = 1 + F.nupd + F.npiv;
      std::array<uint16_t, 16> S{};
      auto mark = [&](int32_t pos) {
        if (pos < 0 || pos >= FAST_IMG_TRI) return;
        int32_t a = 0;
        while ((int64_t)tri(a + 1) <= pos) a++;
        S[a] |= (uint16_t)(1u << (pos - (int32_t)tri(a)));
      };
      for (auto& e : pes) mark(e.pos);
      for (auto& p_ : prs) mark(p_.pos);
      for (int32_t ci = F.child_begin; ci < F.child_end; ci++) {
        const int32_t c = P.child_idx[ci];
        const FrontHdr& C = P.fronts[c];
        const int32_t* rel = P.rel_idx.data() + C.rel_begin;
        for (int32_t a = 0; a <= C.nupd; a++)
          for (int32_t b = 0; b <= a; b++)
            if (udense[c] || (ustruct[c][a] >> b & 1)) S[rel[a]] |= (uint16_t)(1u << rel[b]);
      }
      constexpr int HW = 4;   // kernels2.hip: CNL_BAND_HW
      int32_t need_fix = 0;   // smallest nfix that covers every pivot row
      bool band_ok = true;
      for (int32_t i = f - 1; i > F.nupd; i--) {
        const uint16_t row = (uint16_t)(S[i] & ((1u << i) - 1u));
        const uint16_t bandm = (uint16_t)(((1u << i) - 1u) & ~((i - HW > 0) ? ((1u << (i - HW)) - 1u) : 0u));
        const uint16_t out = (uint16_t)(row & ~bandm);      // non-zeros outside the band: must be among the lowest columns
        if (out) { int32_t hb = 15; while (!(out >> hb & 1)) hb--; need_fix = std::max(need_fix, hb + 1); }
        for (int32_t a = 0; a < i; a++)
          if (row >> a & 1) S[a] |= (uint16_t)(row & ((2u << a) - 1u));
      }
      if (need_fix > 3) band_ok = false;
      const int32_t nfix = need_fix <= 2 ? 2 : 3;     // the kernel carries the forms with 2 and 3 fixed columns
      // worth it only when it removes updates: some pivot must have rows outside its band + fixed columns
      if (band_ok) {
        int32_t removed = 0;
        for (int32_t i = f - 1; i > F.nupd; i--) removed += std::max(0, i - HW - nfix);
        if (removed > 0 && P.band_form) band_word = nfix | (HW << 8);
      }
      for (int32_t a = 0; a <= F.nupd; a++) ustruct[s][a] = S[a];
      udense[s] = 0;
      if (band_word) P.band_fronts++;
    }
    // ---- row form (plan.h, RF_ROWS): products grouped by residual row, operands in registers --------------------------
    ivec rowsec;
    int32_t rows_own = 0, rows_n = 0;
    // The row form costs the same seven gathers and twenty atomic rounds whatever the front holds; the list form costs per
    // round of sixteen products (measured on cfg3's chain fronts, 208 products: 1.47 ms of the kernel as lists, 0.62 ms as
    // rows).  Fronts with fewer than five rounds of products — the small fronts of the bushy latency orders — keep the lists.
    if (D && strided && (int64_t)prs.size() >= P.row_min_products && P.row_products) {
      struct Row { int32_t d; ivec m; int32_t r = -1; std::vector<std::array<int32_t, 3>> pr; };  // pr: (operand a, operand b, pos)
      std::vector<Row> rows;
      std::unordered_map<int32_t, int32_t> row_of;
      bool ok = true;
      for (auto& p_ : prs) {
        if (p_.d >= nnz_thr) { ok = false; break; }
        auto it = row_of.find(p_.d);
        if (it == row_of.end()) { it = row_of.emplace(p_.d, (int32_t)rows.size()).first; rows.push_back(Row()); rows.back().d = p_.d; }
        rows[it->second].pr.push_back({p_.a, p_.b, p_.pos});
      }
      if (rows.size() > 16) { ok = false; if (F.npiv >= 2) P.rows_overflow.push_back(F.first_piv + F.npiv / 2); }
      for (auto& R : rows) {
        if (!ok) break;
        for (auto& q : R.pr)
          for (int k = 0; k < 2; k++) {
            const int32_t src = q[k];
            if (src == R.d) { ok = false; break; }
            if (src >= nnz_thr) { if (R.r >= 0 && R.r != src) ok = false; R.r = src; }
            else if (std::find(R.m.begin(), R.m.end(), src) == R.m.end()) R.m.push_back(src);
          }
        if ((int)R.m.size() > ROWS_KM) ok = false;
        std::sort(R.m.begin(), R.m.end());
      }
      // pivots must not be operands of other rows either (the old form refuses that too)
      if (ok) for (auto& R : rows) for (int32_t m : R.m) if (row_of.count(m)) ok = false;
      if (ok) {
        // rows whose pivot no earlier front has staged come first: this front counts them in the inertia
        std::vector<Row> own, rest;
        for (auto& R : rows) (d_claimed.count(R.d) ? rest : own).push_back(R);
        auto byd = [](const Row& x, const Row& y) { return x.d < y.d; };
        std::sort(own.begin(), own.end(), byd);
        std::sort(rest.begin(), rest.end(), byd);
        rows_own = (int32_t)own.size();
        rows = own;
        rows.insert(rows.end(), rest.begin(), rest.end());
        rows_n = (int32_t)rows.size();
        rowsec.assign(ROWS_WORDS, 0);
        const int32_t dummy = rows[0].d;
        for (int32_t l = 0; l < 16; l++) {
          rowsec[l] = dummy;
          for (int q = 0; q < ROWS_KM; q++) rowsec[16 * (1 + q) + l] = dummy;
          rowsec[16 * (1 + ROWS_KM) + l] = nnz_thr;  // right-hand side entry 0
        }
        std::vector<uint8_t> posb(16 * ROWS_PW * 4);
        for (int32_t l = 0; l < 16; l++) for (int k = 0; k < ROWS_PW * 4; k++) posb[(size_t)l * ROWS_PW * 4 + k] = (uint8_t)(FAST_IMG_TRI + l);
        for (int32_t l = 0; l < rows_n && ok; l++) {
          const Row& R = rows[l];
          rowsec[l] = R.d;
          for (size_t q = 0; q < R.m.size(); q++) rowsec[16 * (1 + q) + l] = R.m[q];
          if (R.r >= 0) rowsec[16 * (1 + ROWS_KM) + l] = R.r;
          std::vector<char> used(ROWS_NPAIR, 0);
          for (auto& q : R.pr) {
            int32_t a = q[0], b = q[1];
            int k;
            if (a >= nnz_thr || b >= nnz_thr) {
              if (a >= nnz_thr && b >= nnz_thr) { ok = false; break; }
              const int32_t m = a >= nnz_thr ? b : a;
              const int iq = (int)(std::find(R.m.begin(), R.m.end(), m) - R.m.begin());
              k = ROWS_KM * (ROWS_KM + 1) / 2 + iq;
            } else {
              int ia = (int)(std::find(R.m.begin(), R.m.end(), a) - R.m.begin()), ib = (int)(std::find(R.m.begin(), R.m.end(), b) - R.m.begin());
              if (ia < ib) std::swap(ia, ib);
              k = ia * (ia + 1) / 2 + ib;
            }
            if (used[k] || q[2] >= FAST_IMG_TRI) { ok = false; break; }  // one slot per operand pair
            used[k] = 1;
            posb[(size_t)l * ROWS_PW * 4 + k] = (uint8_t)q[2];
          }
        }
        if (ok) {
          for (int32_t l = 0; l < 16; l++)
            for (int g = 0; g < ROWS_PW; g++) {
              const uint8_t* pb_ = &posb[(size_t)l * ROWS_PW * 4 + 4 * g];
              rowsec[16 * (2 + ROWS_KM + g) + l] = (int32_t)((uint32_t)pb_[0] | ((uint32_t)pb_[1] << 8) | ((uint32_t)pb_[2] << 16) | ((uint32_t)pb_[3] << 24));
            }
          for (int32_t l = 0; l < rows_own; l++) d_claimed[rows[l].d] = 1;
        } else {
          rowsec.clear();
        }
      }
      if (rowsec.empty() && getenv("CNL_VERBOSE"))
        fprintf(stderr, "[cnl] row form refused: front %d (npiv %d nupd %d), %zu products, %zu rows%s\n", s, F.npiv, F.nupd, prs.size(), rows.size(),
                rows.size() > 16 ? " (> 16 rows)" : "");
    }
    const bool rowform = !rowsec.empty();
    if (rowform) prs.clear();
    // raw sources of the products: pivots d first, then the other operands
    ivec raw; int32_t nrd = 0;
    std::unordered_map<int32_t, int32_t> rawidx;
    for (auto& p_ : prs) if (!rawidx.count(p_.d)) { rawidx[p_.d] = (int32_t)raw.size(); raw.push_back(p_.d); }
    nrd = (int32_t)raw.size();
    auto why = [&](const char* w) { if (getenv("CNL_VERBOSE")) fprintf(stderr, "[cnl] direct records: front %d (order %d): %s (raw %zu, products %zu)\n", s, 1 + F.nupd + F.npiv, w, raw.size(), prs.size()); return 1; };
    for (int32_t i = 0; i < nrd; i++) if (raw[i] >= nnz_thr) return why("a pivot read from the right-hand side");
    // layout: [pivots d | other matrix values | pad to 16 | right-hand-side values | pad to 16], ascending source
    // order inside each group (neighbouring lanes gather neighbouring addresses)
    ivec rv, rr;
    {
      std::unordered_map<int32_t, int32_t> seen;
      for (int32_t i = 0; i < nrd; i++) seen[raw[i]] = 1;
      for (auto& p_ : prs)
        for (int32_t srcq : {p_.a, p_.b})
          if (!seen.count(srcq)) { seen[srcq] = 1; (srcq >= nnz_thr ? rr : rv).push_back(srcq); }
    }
    for (auto& p_ : prs) if (rawidx.count(p_.a) || rawidx.count(p_.b)) return why("a pivot is also an operand");
    // owned pivots first (each group in ascending source order)
    int32_t nrd_own = 0;
    {
      ivec own, rest;
      for (int32_t q : raw) (d_claimed.count(q) ? rest : own).push_back(q);
      for (int32_t q : own) d_claimed[q] = 1;
      std::sort(own.begin(), own.end());
      std::sort(rest.begin(), rest.end());
      nrd_own = (int32_t)own.size();
      raw = own;
      raw.insert(raw.end(), rest.begin(), rest.end());
    }
    std::sort(rv.begin(), rv.end());
    std::sort(rr.begin(), rr.end());
    raw.insert(raw.end(), rv.begin(), rv.end());
    if (!prs.empty()) while (raw.size() % 16) raw.push_back(raw[0]);
    const int32_t nrawv = (int32_t)raw.size();
    raw.insert(raw.end(), rr.begin(), rr.end());
    if (!prs.empty()) while (raw.size() % 16) raw.push_back(rr.empty() ? raw[0] : rr[0]);
    rawidx.clear();
    for (size_t i = 0; i < raw.size(); i++) if (!rawidx.count(raw[i])) rawidx[raw[i]] = (int32_t)i;
    if (raw.size() > 1023) return why("more than 1023 raw values");       // descriptor fields are 10 bits
    if (strided && raw.size() > 128) return why("fast front with more than 128 raw values");  // 7-bit fields, LDS area of 128
    if (strided && (int32_t)raw.size() - nrawv > 32) return why("fast front with more than 32 right-hand-side operands");
    ivec prod;
    {
      int32_t dk = 0;
      if (strided) {  // one word per product: pos | ia << 8 | ib << 15 | id << 22
        for (auto& p_ : prs) prod.push_back(p_.pos | (rawidx[p_.a] << 8) | (rawidx[p_.b] << 15) | (rawidx[p_.d] << 22));
        while (prod.size() % 16) prod.push_back(FAST_IMG_TRI + (dk++ % 16));
      } else {        // two words: pos, ia | ib << 10 | id << 20
        for (auto& p_ : prs) { prod.push_back(p_.pos); prod.push_back(rawidx[p_.a] | (rawidx[p_.b] << 10) | (rawidx[p_.d] << 20)); }
        while ((prod.size() / 2) % 16) { prod.push_back(0); prod.push_back(0); }
      }
      if (prs.empty()) prod.clear();
    }
    if (prs.empty()) raw.clear();
    if (rowform) { raw = rowsec; nrd = rows_n; nrd_own = rows_own; }
    int32_t asm_off = (int32_t)(rec.size() - r0);
    rec.insert(rec.end(), asrc.begin(), asrc.end());
    rec.insert(rec.end(), apos.begin(), apos.end());
    rec.insert(rec.end(), raw.begin(), raw.end());
    rec.insert(rec.end(), prod.begin(), prod.end());
    int32_t child_off = (int32_t)(rec.size() - r0);
    for (int32_t ci = F.child_begin; ci < F.child_end; ci++) {
      int32_t c = P.child_idx[ci];
      const FrontHdr& C = P.fronts[c];
      int32_t tuc = (int32_t)tri(1 + C.nupd);
      rec.push_back(uoff2[c]); rec.push_back(tuc); rec.push_back(uglob[c] ? 1 : 0); rec.push_back(0);
      const int32_t* rel = P.rel_idx.data() + C.rel_begin;
      for (int32_t a = 0; a <= C.nupd; a++)
        for (int32_t b = 0; b <= a; b++) rec.push_back((int32_t)(tri(rel[a]) + rel[b]));
      while ((rec.size() - r0) % 4) rec.push_back(0);
    }
    while ((rec.size() - r0) % 4) rec.push_back(0);
    int32_t* H = rec.data() + r0;
    H[R_NPIV] = F.npiv; H[R_NUPD] = F.nupd; H[R_RECLEN] = (int32_t)(rec.size() - r0); H[R_NASM] = (int32_t)asrc.size();
    H[R_NCHILD] = F.child_end - F.child_begin; H[R_UOFF] = uoff2[s];
    H[R_FLAGS] = (uglob[s] ? RF_U_GLOBAL : 0) | (fsglob[s] ? RF_FS_GLOBAL : 0) | (rowform ? RF_ROWS : 0) | (cls[s] << 8);
    H[R_FSOFF] = strided ? band_word : fsoff2[s];   // fast fronts stage in LDS at a fixed place: the word carries their band form
    P.rows_fronts += rowform ? 1 : 0;
    P.listprod_fronts += (!rowform && !prod.empty() && strided) ? 1 : 0;
    H[R_LPTR_LO] = F.lptr_lo; H[R_LPTR_HI] = F.lptr_hi; H[R_NASMV] = nasmv; H[R_ASM_OFF] = asm_off; H[R_CHILD_OFF] = child_off;
    const int32_t nprod_ = (int32_t)(strided ? prod.size() : prod.size() / 2);
    if (nprod_ >= 65536) return why("more than 65535 products in one front");
    H[R_NPROD] = nprod_ | (nrd_own << 16); H[R_NRAW] = (int32_t)raw.size(); H[R_NRD] = nrd | ((rowform ? 0 : nrawv) << 16);
    st_raw += H[R_NRAW]; st_prod += nprod_; st_asm += H[R_NASM];
    st_rawmax = std::max<int64_t>(st_rawmax, H[R_NRAW]); st_prodmax = std::max<int64_t>(st_prodmax, nprod_);
    st_asmmax = std::max<int64_t>(st_asmmax, H[R_NASM]);
    // globally staged fronts read their lists from the stream itself: only the header must fit the LDS buffer
    rec_maxlen = std::max(rec_maxlen, fsglob[s] ? (int32_t)R_HDR : H[R_RECLEN]);
  }
  if (getenv("CNL_VERBOSE"))
    fprintf(stderr, "[cnl] records%s: fronts %d, plain %lld (max %lld), raw %lld (max %lld), products %lld (max %lld), words %zu\n",
            D ? " (direct)" : "", ns, (long long)st_asm, (long long)st_asmmax, (long long)st_raw, (long long)st_rawmax,
            (long long)st_prod, (long long)st_prodmax, rec.size());
  P.rec.swap(rec);
  P.rec_maxlen = rec_maxlen;
  P.rec_direct = D != nullptr;
  if (D) { P.nnz_outer = D->nnz_outer; P.n_outer = D->n_outer; }
  P.d_owned = (int64_t)d_claimed.size();  // condensed pivots whose inertia the kernel counts while staging them
  return 0;
}


// ---------------------------------------------------------------------------------------------
int write_backward_rows(Plan& P, const BackRowsIn& in) {
  const int32_t ns = P.nsuper;
  if (!P.rec_direct || P.rec.empty() || P.brec.empty() || in.ncond <= 0) return 1;
  // elimination position of every local row of every front (parents first: update rows are named by the parent's rows)
  std::vector<std::array<int32_t, 17>> glob(ns);
  for (int32_t s = ns - 1; s >= 0; s--) {
    const FrontHdr& F = P.fronts[s];
    const int32_t f = 1 + F.nupd + F.npiv;
    if (f > 17) return 1;
    glob[s].fill(-1);
    for (int32_t i = F.nupd + 1; i < f; i++) glob[s][i] = F.first_piv + (f - 1 - i);
    for (int32_t l = 1; l <= F.nupd; l++) {
      if (F.parent < 0) return 1;
      const int32_t pl = P.rel_idx[F.rel_begin + l];
      if (pl < 1 || pl > 16) return 1;
      glob[s][l] = glob[F.parent][pl];
    }
  }
  std::unordered_map<int32_t, int32_t> row_of_d;  // pivot source -> condensed row
  for (int32_t q = 0; q < in.ncond; q++) row_of_d[in.r_dsrc[q]] = q;
  std::vector<ivec> sec(ns);
  std::vector<int32_t> nown(ns, 0);
  std::vector<char> covered(in.ncond, 0);
  int64_t ncov = 0;
  size_t r0 = 0;
  for (int32_t s = 0; s < ns; s++) {
    const int32_t* H = P.rec.data() + r0;
    const FrontHdr& F = P.fronts[s];
    const int32_t f = 1 + F.nupd + F.npiv;
    const int32_t flags = H[R_FLAGS];
    if ((flags >> 8) != 16 || (flags & RF_FS_GLOBAL)) return 1;
    ivec& S = sec[s];
    S.assign(BROWS_WORDS, 0);
    for (int32_t l = 0; l < 16; l++) {
      S[16 * (ROWS_KM + 1) + l] = P.nnz_outer;  // right-hand side entry 0
      S[16 * (ROWS_KM + 2) + l] = 0x11111;      // operands that do not exist: local index 1 (a finite x), coefficient 0
    }
    int32_t rows_own = 0;
    if (flags & RF_ROWS) {
      rows_own = H[R_NPROD] >> 16;
      const int32_t* rs = H + H[R_ASM_OFF] + 2 * H[R_NASM];
      if (H[R_NRAW] != ROWS_WORDS || rows_own > 16) return 1;
      for (int32_t l = 0; l < rows_own; l++) {
        auto it = row_of_d.find(rs[l]);
        if (it == row_of_d.end()) return 1;
        const int32_t q = it->second;
        if (covered[q]) return 1;
        const int32_t k0 = in.r_ptr[q], k1 = in.r_ptr[q + 1];
        if (k1 - k0 > ROWS_KM) return 1;
        int32_t iw = ((k1 - k0) << 20) | (1 << 23) | 0x11111;
        S[l] = in.r_dsrc[q];
        for (int32_t k = k0; k < k1; k++) {
          const int32_t pos = P.iperm[in.r_jx[k]];
          int32_t li = -1;
          for (int32_t t = 1; t < f; t++) if (glob[s][t] == pos) { li = t; break; }
          if (li < 0) return 1;  // a column of the row outside the front that owns it
          S[16 * (1 + (k - k0)) + l] = in.r_jsrc[k];
          iw = (iw & ~(15 << (4 * (k - k0)))) | (li << (4 * (k - k0)));
        }
        S[16 * (ROWS_KM + 1) + l] = P.nnz_outer + in.r_orig[q];
        S[16 * (ROWS_KM + 2) + l] = iw;
        covered[q] = 1;
        ncov++;
      }
    } else if ((H[R_NPROD] & 0xffff) != 0) return 1;  // products in list form: the lean kernel does not take this plan anyway
    nown[s] = rows_own;
    r0 += (size_t)H[R_RECLEN];
  }
  if (ncov != in.ncond) return 1;
  // rebuild the backward stream (reverse post-order) with the sections
  ivec br;
  br.reserve(P.brec.size() + (size_t)ns * BROWS_WORDS);
  int32_t maxlen = 0;
  size_t b0 = 0;
  for (int32_t s = ns - 1; s >= 0; s--) {
    const int32_t len = P.brec[b0 + B_RECLEN];
    const int32_t f = 1 + P.brec[b0 + B_NUPD] + P.brec[b0 + B_NPIV];
    const int32_t roff = (B_HDR + f + 3) & ~3;
    if (len != roff || (P.brec[b0 + B_CLS] >> 8) != 0) return 1;
    const size_t n0 = br.size();
    br.insert(br.end(), P.brec.begin() + b0, P.brec.begin() + b0 + len);
    br.insert(br.end(), sec[s].begin(), sec[s].end());
    br[n0 + B_RECLEN] = roff + BROWS_WORDS;
    br[n0 + B_CLS] |= B_ROWS_FLAG | (nown[s] << 16);
    maxlen = std::max(maxlen, roff + (int32_t)BROWS_WORDS);
    b0 += (size_t)len;
  }
  P.brec.swap(br);
  P.brec_maxlen = maxlen;
  P.back_rows = true;
  finalize_tasks(P);
  return 0;
}

}  // namespace cnl
