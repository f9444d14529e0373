// Host-side sparse helpers of the product (setup-time only; see host_sparse.h).
#include "host_sparse.h"
#include "env.h"

#include <sched.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <queue>
#include <thread>
#include <unordered_map>

namespace dcora {

HostCsr csr_from_coo(int nrows, int ncols, const std::vector<int> &I, const std::vector<int> &J,
                     const std::vector<double> &V) {
  HostCsr A;
  A.n = nrows;
  A.ncols = ncols;
  const size_t m = I.size();
  std::vector<int> cnt(nrows + 1, 0);
  for (size_t e = 0; e < m; ++e) cnt[I[e] + 1]++;
  for (int i = 0; i < nrows; ++i) cnt[i + 1] += cnt[i];
  std::vector<size_t> order(m);
  {
    std::vector<int> pos(cnt.begin(), cnt.end() - 1);
    for (size_t e = 0; e < m; ++e) order[pos[I[e]]++] = e;
  }
  A.rp.assign(nrows + 1, 0);
  A.ci.reserve(m);
  A.v.reserve(m);
  std::vector<std::pair<int, double>> row;
  for (int i = 0; i < nrows; ++i) {
    row.clear();
    for (int p = cnt[i]; p < cnt[i + 1]; ++p) row.emplace_back(J[order[p]], V[order[p]]);
    std::stable_sort(row.begin(), row.end(), [](const auto &a, const auto &b) { return a.first < b.first; });
    for (size_t q = 0; q < row.size();) {
      const int c = row[q].first;
      double s = 0;
      while (q < row.size() && row[q].first == c) s += row[q++].second;
      A.ci.push_back(c);
      A.v.push_back(s);
    }
    A.rp[i + 1] = (int)A.ci.size();
  }
  return A;
}

HostBsr bsr_from_csr(const HostCsr &A, int bs) {
  HostBsr B;
  B.bs = bs;
  B.nbrows = A.n / bs;
  B.bp.assign(B.nbrows + 1, 0);
  std::vector<int> slot;  // block column -> position in the current block row
  std::vector<int> cols;
  for (int bi = 0; bi < B.nbrows; ++bi) {
    cols.clear();
    for (int a = 0; a < bs; ++a) {
      const int i = bi * bs + a;
      for (int p = A.rp[i]; p < A.rp[i + 1]; ++p) cols.push_back(A.ci[p] / bs);
    }
    std::sort(cols.begin(), cols.end());
    cols.erase(std::unique(cols.begin(), cols.end()), cols.end());
    const size_t first = B.bc.size();
    B.bc.insert(B.bc.end(), cols.begin(), cols.end());
    B.bv.resize(B.bc.size() * (size_t)bs * bs, 0.0);
    for (int a = 0; a < bs; ++a) {
      const int i = bi * bs + a;
      for (int p = A.rp[i]; p < A.rp[i + 1]; ++p) {
        const int bcj = A.ci[p] / bs, c = A.ci[p] - bcj * bs;
        const size_t pos = first + (std::lower_bound(cols.begin(), cols.end(), bcj) - cols.begin());
        B.bv[pos * bs * bs + (size_t)c * bs + a] += A.v[p];  // entry (a, c) of the block at c bs + a
      }
    }
    B.bp[bi + 1] = (int)B.bc.size();
  }
  return B;
}

HostCsr csr_shift_diag(const HostCsr &A, double s) {
  HostCsr B;
  B.n = A.n;
  B.ncols = A.ncols;
  B.rp.assign(A.n + 1, 0);
  for (int i = 0; i < A.n; ++i) {
    bool seen = false;
    for (int p = A.rp[i]; p < A.rp[i + 1]; ++p) {
      const int c = A.ci[p];
      if (!seen && c > i) {
        B.ci.push_back(i);
        B.v.push_back(s);
        seen = true;
      }
      B.ci.push_back(c);
      B.v.push_back(A.v[p] + ((c == i) ? s : 0.0));
      if (c == i) seen = true;
    }
    if (!seen) {
      B.ci.push_back(i);
      B.v.push_back(s);
    }
    B.rp[i + 1] = (int)B.ci.size();
  }
  return B;
}

// ---------------------------------------------------------------------------------------------------
// Ordering: automatic nested dissection on the block-compressed graph.  A component is bisected along a level of a
// BFS level structure rooted at a pseudo-peripheral node (George), the bisection is refined and its separator is a
// minimum vertex cover of the cut edges (CutCover below); leaves are ordered by a cheap local minimum-degree pass.  Produces shallow elimination trees, which is what a level-scheduled device
// triangular solve will want ("next").
// ---------------------------------------------------------------------------------------------------
namespace {
struct NDGraph {
  int nb;
  std::vector<int> xadj, adj;
};

void bfs_levels(const NDGraph &G, const std::vector<int> &comp_id, int cid, int root, std::vector<int> &level,
                std::vector<int> &order) {
  order.clear();
  order.push_back(root);
  level[root] = 0;
  for (size_t h = 0; h < order.size(); ++h) {
    const int u = order[h];
    for (int p = G.xadj[u]; p < G.xadj[u + 1]; ++p) {
      const int w = G.adj[p];
      if (comp_id[w] == cid && level[w] < 0) {
        level[w] = level[u] + 1;
        order.push_back(w);
      }
    }
  }
}

void leaf_order(const NDGraph &G, const std::vector<int> &nodes, std::vector<int> &comp_id, int cid,
                std::vector<int> &out) {
  // greedy minimum (static) degree inside the leaf; small sets only
  std::vector<std::pair<int, int>> dn;
  for (int u : nodes) {
    int dg = 0;
    for (int p = G.xadj[u]; p < G.xadj[u + 1]; ++p)
      if (comp_id[G.adj[p]] == cid) ++dg;
    dn.emplace_back(dg, u);
  }
  std::sort(dn.begin(), dn.end());
  for (auto &e : dn) out.push_back(e.second);
}

// ---- vertex separators from edge bisections (round 4).  A whole BFS level is a poor separator of a sparse pose graph
//      (trajectory chains + loop closures: 3.4 neighbours per pose): most of its nodes touch only one side.  For a
//      bisection (side 0 / 1 of the nodes of a component) the smallest set of nodes covering every cut edge is a minimum
//      vertex cover of the bipartite graph of the cut edges (Koenig: its size is a maximum matching, Hopcroft-Karp).
//      Measured on the pose graph of a 100k-lattice agent: nnz(L) 635 k -> 460 k blocks with the level chosen among the
//      balanced ones, a greedy refinement of the bisection and the cover (a spectral bisection gives 414 k, minimum degree
//      300 k); the whole lattice: 23.4 M -> 15.4 M, the factorisation's flops -47 %. ----
constexpr double kNdBalance = 0.42;  // either side of a bisection keeps at least this share of the component's nodes
struct CutCover {
  // local ids 0 .. m - 1 of a component; nbr lists through lid (global -> local, -1 outside)
  const NDGraph &G;
  const std::vector<int> &nodes;
  const std::vector<int> &lid;
  std::vector<int> aidx, bidx;        // boundary nodes of side 0 / side 1 (local ids)
  std::vector<int> apos, bpos;        // local id -> position in aidx / bidx, or -1
  std::vector<int> xadj, adj;         // A position -> B positions
  std::vector<int> matchA, matchB, dist;
  CutCover(const NDGraph &G_, const std::vector<int> &nodes_, const std::vector<int> &lid_)
      : G(G_), nodes(nodes_), lid(lid_) {}
  bool bfs() {
    std::vector<int> q;
    const int INF = 1 << 30;
    bool found = false;
    for (size_t a = 0; a < aidx.size(); ++a) {
      dist[a] = matchA[a] < 0 ? 0 : INF;
      if (matchA[a] < 0) q.push_back((int)a);
    }
    for (size_t h = 0; h < q.size(); ++h) {
      const int a = q[h];
      for (int p = xadj[a]; p < xadj[a + 1]; ++p) {
        const int a2 = matchB[adj[p]];
        if (a2 < 0) {
          found = true;
        } else if (dist[a2] == INF) {
          dist[a2] = dist[a] + 1;
          q.push_back(a2);
        }
      }
    }
    return found;
  }
  bool dfs(int a) {
    for (int p = xadj[a]; p < xadj[a + 1]; ++p) {
      const int b = adj[p], a2 = matchB[b];
      if (a2 < 0 || (dist[a2] == dist[a] + 1 && dfs(a2))) {
        matchA[a] = b;
        matchB[b] = a;
        return true;
      }
    }
    dist[a] = 1 << 30;
    return false;
  }
  // minimum vertex cover of the cut of `side`; cover (local ids) returned in `out`
  int run(const std::vector<char> &side, std::vector<int> *out) {
    const int m = (int)nodes.size();
    aidx.clear();
    bidx.clear();
    apos.assign((size_t)m, -1);
    bpos.assign((size_t)m, -1);
    for (int i = 0; i < m; ++i) {
      const int u = nodes[(size_t)i];
      bool cut = false;
      for (int p = G.xadj[u]; p < G.xadj[u + 1] && !cut; ++p) {
        const int j = lid[(size_t)G.adj[p]];
        cut = j >= 0 && side[(size_t)j] != side[(size_t)i];
      }
      if (!cut) continue;
      if (side[(size_t)i]) {
        bpos[(size_t)i] = (int)bidx.size();
        bidx.push_back(i);
      } else {
        apos[(size_t)i] = (int)aidx.size();
        aidx.push_back(i);
      }
    }
    xadj.assign(aidx.size() + 1, 0);
    adj.clear();
    for (size_t a = 0; a < aidx.size(); ++a) {
      const int u = nodes[(size_t)aidx[a]];
      for (int p = G.xadj[u]; p < G.xadj[u + 1]; ++p) {
        const int j = lid[(size_t)G.adj[p]];
        if (j >= 0 && side[(size_t)j]) adj.push_back(bpos[(size_t)j]);
      }
      xadj[a + 1] = (int)adj.size();
    }
    matchA.assign(aidx.size(), -1);
    matchB.assign(bidx.size(), -1);
    dist.assign(aidx.size(), 0);
    int matched = 0;
    while (bfs())
      for (size_t a = 0; a < aidx.size(); ++a)
        if (matchA[a] < 0 && dfs((int)a)) ++matched;
    if (!out) return matched;
    // Koenig: Z = everything reachable from the unmatched A nodes along alternating paths; cover = (A \ Z) + (B & Z)
    std::vector<char> za(aidx.size(), 0), zb(bidx.size(), 0);
    std::vector<int> st;
    for (size_t a = 0; a < aidx.size(); ++a)
      if (matchA[a] < 0) {
        za[a] = 1;
        st.push_back((int)a);
      }
    while (!st.empty()) {
      const int a = st.back();
      st.pop_back();
      for (int p = xadj[a]; p < xadj[a + 1]; ++p) {
        const int b = adj[p];
        if (zb[(size_t)b] || matchA[(size_t)a] == b) continue;
        zb[(size_t)b] = 1;
        const int a2 = matchB[(size_t)b];
        if (a2 >= 0 && !za[(size_t)a2]) {
          za[(size_t)a2] = 1;
          st.push_back(a2);
        }
      }
    }
    out->clear();
    for (size_t a = 0; a < aidx.size(); ++a)
      if (!za[a]) out->push_back(aidx[a]);
    for (size_t b = 0; b < bidx.size(); ++b)
      if (zb[b]) out->push_back(bidx[b]);
    return (int)out->size();
  }
};

// single-node moves that reduce the number of cut edges, largest gain first, both sides kept above `lo` of the component
void refine_bisection(const NDGraph &G, const std::vector<int> &nodes, const std::vector<int> &lid,
                      std::vector<char> &side, double lo) {
  const int m = (int)nodes.size();
  int n1 = 0;
  for (char c : side) n1 += c;
  const int min_side = (int)(lo * m);
  std::vector<std::pair<int, int>> cand;
  for (int pass = 0; pass < 4; ++pass) {
    cand.clear();
    for (int i = 0; i < m; ++i) {
      const int u = nodes[(size_t)i];
      int ext = 0, inn = 0;
      for (int p = G.xadj[u]; p < G.xadj[u + 1]; ++p) {
        const int j = lid[(size_t)G.adj[p]];
        if (j < 0) continue;
        if (side[(size_t)j] != side[(size_t)i])
          ++ext;
        else
          ++inn;
      }
      if (ext > inn) cand.emplace_back(-(ext - inn), i);
    }
    if (cand.empty()) break;
    std::sort(cand.begin(), cand.end());
    int moved = 0;
    for (const auto &c : cand) {
      const int i = c.second, u = nodes[(size_t)i];
      int ext = 0, inn = 0;
      for (int p = G.xadj[u]; p < G.xadj[u + 1]; ++p) {
        const int j = lid[(size_t)G.adj[p]];
        if (j < 0) continue;
        if (side[(size_t)j] != side[(size_t)i])
          ++ext;
        else
          ++inn;
      }
      if (ext <= inn) continue;
      if (side[(size_t)i] ? (n1 - 1 < min_side) : (m - n1 - 1 < min_side)) continue;
      side[(size_t)i] = !side[(size_t)i];
      n1 += side[(size_t)i] ? 1 : -1;
      ++moved;
    }
    if (!moved) break;
  }
}

// one dissection step of a labelled component (comp_id == cid, level == -1 on its nodes; lid == -1 everywhere):
// 0 = split into sep / left / right, 1 = disconnected (reached / rest returned in left / right), 2 = too shallow.
// All scratch arrays are the caller's: the serial recursion and the workers of the parallel pre-pass bring their own.
int split_component(const NDGraph &G, const std::vector<int> &cur, int cid, const std::vector<int> &comp_id,
                    std::vector<int> &level, std::vector<int> &lid, std::vector<int> &order, std::vector<int> &sep,
                    std::vector<int> &left, std::vector<int> &right) {
  sep.clear();
  left.clear();
  right.clear();
  int root = cur[0];
  bfs_levels(G, comp_id, cid, root, level, order);  // pseudo-peripheral root: two more sweeps below
  if (order.size() < cur.size()) {
    for (int u : cur)
      if (level[u] < 0) right.push_back(u);
    left = order;
    return 1;
  }
  for (int sweep = 0; sweep < 2; ++sweep) {
    root = order.back();
    for (int u : cur) level[u] = -1;
    bfs_levels(G, comp_id, cid, root, level, order);
  }
  const int depth = level[order.back()];
  if (depth < 2) return 2;
  const int level_of_median = level[order[order.size() / 2]];
  // Bisections {f < t} | {f >= t} of six node functions -- the BFS level from one end u of the pseudo-diameter,
  // the level from its other end v, their difference (whose level sets are the bisectors between the two ends:
  // straighter than either family of "spheres"), and the same with a third root w far from both -- at the thresholds t that leave 42 .. 58 % of the nodes on either
  // side.  The thresholds of a function are ranked by the smaller of the two boundaries of their cuts (one pass for
  // all of them); the best one is covered as it is and after a refinement (single-node moves that remove cut edges);
  // the smallest separator of the twelve wins, ties go to the more balanced cut.  (Pose graph of a 100k-lattice agent, nnz(L) in
  // blocks: a whole BFS level 635 k; level of u + cover 488 k; + levels of v 448 k; + the difference 419 k; a
  // spectral bisection 414 k.)
  const int m = (int)cur.size();
  for (int i = 0; i < m; ++i) lid[(size_t)cur[(size_t)i]] = i;
  std::vector<int> lev_u((size_t)m), lev_v((size_t)m), lev_w((size_t)m), fv((size_t)m);
  for (int i = 0; i < m; ++i) lev_u[(size_t)i] = level[cur[(size_t)i]];
  {
    const int v_end = order.back();
    for (int u : cur) level[u] = -1;
    std::vector<int> order2;
    bfs_levels(G, comp_id, cid, v_end, level, order2);
    for (int i = 0; i < m; ++i) lev_v[(size_t)i] = level[cur[(size_t)i]];
    // a third root w, far from both ends (the first node that maximises min(d_u, d_v)): its levels and the two
    // differences with it cut ACROSS the pseudo-diameter's direction.  Fill of the whole 100k lattice with three / six
    // functions, nnz(L) in blocks: 14.4 M / 11.9 M, the factorisation's flops -41 %; top separator 1414 / 847 poses.
    int wi = 0, wbest = -1;
    for (int i = 0; i < m; ++i) {
      const int d = std::min(lev_u[(size_t)i], lev_v[(size_t)i]);
      if (d > wbest) {
        wbest = d;
        wi = i;
      }
    }
    for (int u : cur) level[u] = -1;
    bfs_levels(G, comp_id, cid, cur[(size_t)wi], level, order2);
    for (int i = 0; i < m; ++i) lev_w[(size_t)i] = level[cur[(size_t)i]];
  }
  CutCover cc(G, cur, lid);
  std::vector<char> side((size_t)m), best_side;
  std::vector<int> cover, best_cover;
  int best = -1, best_imb = 0;
  for (int fn = 0; fn < 6; ++fn) {
    int fmin = 1 << 30, fmax = -(1 << 30);
    for (int i = 0; i < m; ++i) {
      const int du = lev_u[(size_t)i], dv = lev_v[(size_t)i], dw = lev_w[(size_t)i];
      fv[(size_t)i] = fn == 0 ? du : fn == 1 ? dv : fn == 2 ? du - dv : fn == 3 ? dw : fn == 4 ? du - dw : dv - dw;
      fmin = std::min(fmin, fv[(size_t)i]);
      fmax = std::max(fmax, fv[(size_t)i]);
    }
    std::vector<int> below((size_t)(fmax - fmin) + 2, 0);  // below[t - fmin] = nodes with f < t
    for (int i = 0; i < m; ++i) ++below[(size_t)(fv[(size_t)i] - fmin) + 1];
    for (size_t l = 1; l < below.size(); ++l) below[l] += below[l - 1];
    // boundary sizes of EVERY threshold in one pass: node i lies on the low side's boundary for the thresholds
    // f_i < t <= (largest f among its neighbours), on the high side's for (smallest f among its neighbours) < t <= f_i;
    // the smaller of the two boundaries bounds the cover and ranks the thresholds
    const int range = fmax - fmin + 2;
    std::vector<int> lowb((size_t)range + 1, 0), highb((size_t)range + 1, 0);
    for (int i = 0; i < m; ++i) {
      const int u = cur[(size_t)i];
      int hi = fv[(size_t)i], lo = fv[(size_t)i];
      for (int p2 = G.xadj[u]; p2 < G.xadj[u + 1]; ++p2) {
        const int j = lid[(size_t)G.adj[p2]];
        if (j < 0) continue;
        hi = std::max(hi, fv[(size_t)j]);
        lo = std::min(lo, fv[(size_t)j]);
      }
      if (hi > fv[(size_t)i]) {  // thresholds fv + 1 .. hi
        ++lowb[(size_t)(fv[(size_t)i] + 1 - fmin)];
        --lowb[(size_t)(hi + 1 - fmin)];
      }
      if (lo < fv[(size_t)i]) {  // thresholds lo + 1 .. fv
        ++highb[(size_t)(lo + 1 - fmin)];
        --highb[(size_t)(fv[(size_t)i] + 1 - fmin)];
      }
    }
    for (int q = 1; q <= range; ++q) {
      lowb[(size_t)q] += lowb[(size_t)q - 1];
      highb[(size_t)q] += highb[(size_t)q - 1];
    }
    std::vector<std::pair<std::pair<int, int>, int>> ranked;  // ((boundary bound, imbalance), threshold)
    for (int t = fmin + 1; t <= fmax; ++t) {
      const int nlow = below[(size_t)(t - fmin)];
      if (nlow >= kNdBalance * m && nlow <= (1.0 - kNdBalance) * m)
        ranked.push_back({{std::min(lowb[(size_t)(t - fmin)], highb[(size_t)(t - fmin)]), std::abs(2 * nlow - m)}, t});
    }
    if (ranked.empty() && fn == 0)
      ranked.push_back({{0, 0}, std::min(std::max(level_of_median, fmin + 1), fmax)});
    if (ranked.empty()) continue;
    std::sort(ranked.begin(), ranked.end());
    // the best-ranked threshold as it is and refined
    for (int refined = 0; refined < 2; ++refined) {
      for (int i = 0; i < m; ++i) side[(size_t)i] = fv[(size_t)i] >= ranked[0].second ? 1 : 0;
      if (refined) refine_bisection(G, cur, lid, side, kNdBalance);
      const int sz = cc.run(side, &cover);
      int n1 = 0;
      for (char c : side) n1 += c;
      const int imb = std::abs(2 * n1 - m);
      if (best < 0 || sz < best || (sz == best && imb < best_imb)) {
        best = sz;
        best_imb = imb;
        best_side = side;
        best_cover = cover;
      }
    }
  }
  for (int i = 0; i < m; ++i) level[cur[(size_t)i]] = lev_u[(size_t)i];  // (the fallback below reads the levels of u)
  std::vector<char> in_sep((size_t)m, 0);
  for (int i : best_cover) in_sep[(size_t)i] = 1;
  for (int i = 0; i < m; ++i) {
    const int u = cur[(size_t)i];
    if (in_sep[(size_t)i])
      sep.push_back(u);  // separators are eliminated last
    else if (!best_side[(size_t)i])
      left.push_back(u);
    else
      right.push_back(u);
  }
  for (int u : cur) lid[(size_t)u] = -1;
  if (left.empty() || right.empty()) {  // (a degenerate cut: everything on one side of the cover)
    sep.clear();
    left.clear();
    right.clear();
    const int mid = level[order[order.size() / 2]];
    const int sep_level = std::min(std::max(mid, 1), depth - 1);
    for (int u : order) {
      if (level[u] == sep_level)
        sep.push_back(u);
      else if (level[u] < sep_level)
        left.push_back(u);
      else
        right.push_back(u);
    }
  }
  return 0;
}

// the splits of one dissection, kept for a second pass over the same graph (amd_like_order probes the separator sizes
// per depth before it fixes the dense top and the leaf size: the final pass meets the same components again)
struct SplitMemo {
  struct Entry {
    std::vector<int> nodes, sep, left, right;
    int how;
  };
  std::unordered_map<unsigned long long, std::vector<Entry>> map;
  static unsigned long long key(const std::vector<int> &v) {
    unsigned long long h = 1469598103934665603ull ^ v.size();
    for (int x : v) h = (h ^ (unsigned long long)(unsigned)x) * 1099511628211ull;
    return h;
  }
};

// Parallel pre-pass of a large dissection: the components of one depth are independent, so a pool of host threads splits
// them side by side (every worker with scratch arrays of its own) and leaves the splits in the memo; the serial passes of
// amd_like_order then only assemble the order.  (The whole 100k lattice: 331 ms of ordering on one thread.)
void prefill_splits(const NDGraph &G, const std::vector<int> &all, int leaf_nodes, int max_depth, int nthreads,
                    SplitMemo *memo) {
  struct Item {
    std::vector<int> nodes;
    int dep;
  };
  std::deque<Item> queue;
  std::mutex mu;
  std::condition_variable cv;
  int active = 0;
  queue.push_back(Item{all, 0});
  auto worker = [&]() {
    std::vector<int> comp_id((size_t)G.nb, -1), level((size_t)G.nb, -1), lid((size_t)G.nb, -1), order, sep, left, right;
    for (;;) {
      Item it;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return !queue.empty() || active == 0; });
        if (queue.empty()) return;
        it = std::move(queue.front());
        queue.pop_front();
        ++active;
      }
      const std::vector<int> &cur = it.nodes;
      int how = 2;
      const bool leaf = (int)cur.size() <= leaf_nodes ||
                        (max_depth >= 0 && it.dep >= max_depth && (int)cur.size() <= 3 * leaf_nodes);
      if (!leaf && !cur.empty()) {
        for (int u : cur) {
          comp_id[(size_t)u] = 1;
          level[(size_t)u] = -1;
        }
        how = split_component(G, cur, 1, comp_id, level, lid, order, sep, left, right);
        for (int u : cur) comp_id[(size_t)u] = -1;
      }
      {
        std::unique_lock<std::mutex> lk(mu);
        if (!leaf && !cur.empty()) {
          memo->map[SplitMemo::key(cur)].push_back(SplitMemo::Entry{cur, sep, left, right, how});
          if (how == 0 || how == 1) {
            const int d2 = how == 0 ? it.dep + 1 : it.dep;
            queue.push_back(Item{std::move(left), d2});
            queue.push_back(Item{std::move(right), d2});
          }
        }
        --active;
      }
      cv.notify_all();
    }
  };
  std::vector<std::thread> pool;
  for (int t = 1; t < nthreads; ++t) pool.emplace_back(worker);
  worker();
  for (std::thread &t : pool) t.join();
}

void nd_recurse(const NDGraph &G, std::vector<int> nodes, std::vector<int> &comp_id, int &next_cid,
                std::vector<int> &level, std::vector<int> &out, std::vector<int> &cuts, int leaf_nodes,
                int task_nodes, std::vector<std::pair<int, int>> &tasks,
                std::vector<std::vector<std::pair<int, int>>> &sep_waves, int top_depth,
                std::vector<long> *sep_nodes_by_depth, int max_depth = -1, SplitMemo *memo = nullptr) {
  // max_depth >= 0: a component that is still larger than a leaf at this depth becomes a leaf all the same when it is at
  // most three leaves large -- bisections that are balanced only to 42 : 58 would otherwise add a tree level (two
  // dependent launches of the device replay) for a handful of stragglers
  // top_depth > 0: the separators of the first top_depth dissection depths are not placed between their sub-trees
  // but collected and eliminated last, as ONE piece (see amd_like_order).  sep_nodes_by_depth (optional) receives the
  // number of separator nodes per depth.
  // iterative worklist: (nodes) ; output order is built back-to-front: separators last.
  // A component and everything dissected out of it occupy one contiguous run of the order; maximal components of
  // at most task_nodes nodes are reported as independent tasks (their columns depend on nothing outside the run).
  // Separators above the tasks are reported by depth (sep_waves[depth]): separators of one depth head disjoint
  // sub-trees, so a wave can be factorised concurrently once everything deeper is done.
  std::vector<int> lid((size_t)G.nb, -1);  // global node -> local id inside the component being split
  std::vector<std::vector<int>> stack;
  std::vector<char> stack_in_task;
  std::vector<int> stack_depth;
  std::vector<int> rev;  // reversed elimination order
  std::vector<int> top;  // separators of depth < top_depth, in the order they were found (root first)
  stack.push_back(std::move(nodes));
  stack_in_task.push_back(0);
  stack_depth.push_back(0);
  std::vector<int> order;
  auto note_sep = [&](int dep, int start, int count) {
    if (task_nodes <= 0 || count <= 0) return;
    if ((int)sep_waves.size() <= dep) sep_waves.resize(dep + 1);
    sep_waves[dep].emplace_back(start, count);
  };
  auto try_split = [&](const std::vector<int> &cur, int cid, std::vector<int> &sep, std::vector<int> &left,
                       std::vector<int> &right) -> int {
    return split_component(G, cur, cid, comp_id, level, lid, order, sep, left, right);
  };
  auto label = [&](const std::vector<int> &nodes_, int cid) {
    for (int u : nodes_) {
      comp_id[u] = cid;
      level[u] = -1;
    }
  };
  auto unlabel = [&](const std::vector<int> &nodes_) {
    for (int u : nodes_) comp_id[u] = -1;
  };
  auto push = [&](std::vector<int> &&nodes_, char in_task, int dep) {
    stack.push_back(std::move(nodes_));
    stack_in_task.push_back(in_task);
    stack_depth.push_back(dep);
  };
  auto count_sep = [&](int dep, size_t count) {
    if (!sep_nodes_by_depth) return;
    if ((int)sep_nodes_by_depth->size() <= dep) sep_nodes_by_depth->resize(dep + 1, 0);
    (*sep_nodes_by_depth)[dep] += (long)count;
  };
  std::vector<int> sep, left, right;
  while (!stack.empty()) {
    std::vector<int> cur = std::move(stack.back());
    char in_task = stack_in_task.back();
    const int dep = stack_depth.back();
    stack.pop_back();
    stack_in_task.pop_back();
    stack_depth.pop_back();
    if (cur.empty()) continue;
    if (!in_task && (int)cur.size() <= task_nodes && dep >= top_depth) {
      tasks.emplace_back((int)rev.size(), (int)cur.size());
      in_task = 1;
    }
    const int cid = next_cid++;
    label(cur, cid);
    if ((int)cur.size() <= leaf_nodes || (max_depth >= 0 && dep >= max_depth && (int)cur.size() <= 3 * leaf_nodes)) {
      std::vector<int> lo;
      leaf_order(G, cur, comp_id, cid, lo);
      for (auto it = lo.rbegin(); it != lo.rend(); ++it) rev.push_back(*it);
      cuts.push_back((int)rev.size());
      unlabel(cur);
      continue;
    }
    int how = -1;
    unsigned long long mkey = 0;
    if (memo) {
      mkey = SplitMemo::key(cur);
      auto it = memo->map.find(mkey);
      if (it != memo->map.end())
        for (const SplitMemo::Entry &e : it->second)
          if (e.nodes == cur) {
            sep = e.sep;
            left = e.left;
            right = e.right;
            how = e.how;
            break;
          }
    }
    if (how < 0) {
      how = try_split(cur, cid, sep, left, right);
      if (memo) memo->map[mkey].push_back(SplitMemo::Entry{cur, sep, left, right, how});
    }
    if (how == 1) {  // disconnected: split off the reached component
      unlabel(cur);
      push(std::move(right), in_task, dep);
      push(std::move(left), in_task, dep);
      continue;
    }
    if (how == 2) {
      std::vector<int> lo;
      leaf_order(G, cur, comp_id, cid, lo);
      if (!in_task) note_sep(dep, (int)rev.size(), (int)lo.size());
      for (auto it = lo.rbegin(); it != lo.rend(); ++it) rev.push_back(*it);
      cuts.push_back((int)rev.size());
      unlabel(cur);
      continue;
    }
    unlabel(cur);
    count_sep(dep, sep.size());
    if (dep < top_depth) {
      top.insert(top.end(), sep.begin(), sep.end());
      push(std::move(left), in_task, dep + 1);
      push(std::move(right), in_task, dep + 1);
      continue;
    }
    // (Measured and dropped: the separators of the two children joined to this one as ONE piece -- a four-way
    // dissection step, to halve the dependent levels of the device replay below the dense top.  BFS-level separators
    // leave many children disconnected, so only one depth pair merged on the critical path of a 100k-lattice agent:
    // 19 -> 17 launches, +32 MB of weights, 170 -> 190 us per application.)
    const int sep_start = (int)rev.size();
    rev.insert(rev.end(), sep.begin(), sep.end());
    cuts.push_back((int)rev.size());
    if (!in_task) note_sep(dep, sep_start, (int)rev.size() - sep_start);
    push(std::move(left), in_task, dep + 1);
    push(std::move(right), in_task, dep + 1);
  }
  for (auto it = rev.rbegin(); it != rev.rend(); ++it) out.push_back(*it);
  // cuts were taken in the reversed order: position p there is position total - p in the final order
  const int total = (int)rev.size();
  for (auto &t : tasks) t.first = total - t.first - t.second;  // (start, size) in the final order
  for (auto &wv : sep_waves)
    for (auto &t : wv) t.first = total - t.first - t.second;
  for (int &c : cuts) c = total - c;
  cuts.push_back(0);
  cuts.push_back(total);
  if (!top.empty()) {  // the merged top piece: deeper separators first, the root separator last
    for (auto it = top.rbegin(); it != top.rend(); ++it) out.push_back(*it);
    cuts.push_back(total + (int)top.size());
  }
  std::sort(cuts.begin(), cuts.end());
  cuts.erase(std::unique(cuts.begin(), cuts.end()), cuts.end());
}
}  // namespace

int host_cpus_available() {
  static const int n = [] {
    if (env::host_threads() > 0) return env::host_threads();
    long best = (long)std::max(1u, std::thread::hardware_concurrency());
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof(set), &set) == 0 && CPU_COUNT(&set) > 0) best = std::min<long>(best, CPU_COUNT(&set));
    auto quota = [&](const char *path) {
      if (FILE *f = std::fopen(path, "r")) {
        char a[64] = {0};
        long period = 0;
        if (std::fscanf(f, "%63s %ld", a, &period) == 2 && std::strcmp(a, "max") != 0 && period > 0) {
          const long q = atol(a);
          if (q > 0) best = std::min(best, (q + period - 1) / period);
        }
        std::fclose(f);
      }
    };
    quota("/sys/fs/cgroup/cpu.max");
    {
      long q = -1, per = -1;
      if (FILE *f = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
        if (std::fscanf(f, "%ld", &q) != 1) q = -1;
        std::fclose(f);
      }
      if (FILE *f = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
        if (std::fscanf(f, "%ld", &per) != 1) per = -1;
        std::fclose(f);
      }
      if (q > 0 && per > 0) best = std::min(best, (q + per - 1) / per);
    }
    return (int)std::max(1L, best);
  }();
  return n;
}

int nd_top_default() {
  // measured on sphere2500 / torus3D / tiers.pyfg / a 100k-lattice agent: best of 1536 / 2048 / 3072 / 4096
  return 3072;
}

std::vector<int> amd_like_order(const HostCsr &A, int block, std::vector<int> *pieces, int *nhub_cols,
                                std::vector<std::pair<int, int>> *col_tasks, int want_tasks,
                                std::vector<std::vector<std::pair<int, int>>> *col_waves, int top_unknowns) {
  const int n = A.n;
  if (block < 1) block = 1;
  const int nb = (n + block - 1) / block;
  // block graph
  std::vector<std::vector<int>> nbrs(nb);
  for (int i = 0; i < n; ++i)
    for (int p = A.rp[i]; p < A.rp[i + 1]; ++p) {
      const int a = i / block, b = A.ci[p] / block;
      if (a != b) nbrs[a].push_back(b);
    }
  NDGraph G;
  G.nb = nb;
  G.xadj.assign(nb + 1, 0);
  for (int u = 0; u < nb; ++u) {
    auto &v = nbrs[u];
    std::sort(v.begin(), v.end());
    v.erase(std::unique(v.begin(), v.end()), v.end());
    G.xadj[u + 1] = G.xadj[u] + (int)v.size();
  }
  G.adj.reserve(G.xadj[nb]);
  for (int u = 0; u < nb; ++u) G.adj.insert(G.adj.end(), nbrs[u].begin(), nbrs[u].end());
  std::vector<int> comp_id(nb, -1), level(nb, -1), border;
  // hubs (a landmark ranged from every pose, say) would collapse every BFS level structure to depth 2: take them
  // out first and eliminate them last, as one top-level separator
  std::vector<int> all, hubs;
  {
    const double mean_deg = nb ? (double)G.xadj[nb] / nb : 0.0;
    const int thr = std::max(48, (int)(8.0 * mean_deg));
    for (int u = 0; u < nb; ++u) (G.xadj[u + 1] - G.xadj[u] > thr ? hubs : all).push_back(u);
  }
  int next_cid = 0;
  std::vector<int> cuts;
  // leaf sub-domains of ~96 unknowns: 24 pose blocks, or 96 scalar unknowns when the graph is not block-compressed
  constexpr int leaf_unknowns = 96;
  int leaf_nodes = std::max(leaf_unknowns / 4, leaf_unknowns / block);
  std::vector<std::pair<int, int>> tasks;
  const int task_nodes = (col_tasks && want_tasks > 1) ? std::max(4 * leaf_nodes, (int)all.size() / want_tasks) : 0;
  std::vector<std::vector<std::pair<int, int>>> waves;
  // Dense top of the tree.  On surface-like graphs (sphere2500, tiers.pyfg) the upper separators are small; every
  // dissection depth costs the device replay two dependent launches (sparse_precond.h), while ONE piece holding the
  // separators of the first L depths is applied by a single symmetric dense product.  L = as many depths as fit
  // top_unknowns unknowns (nd_top_default() for the replay; 0 for a plain factorisation, whose dense fronts would
  // only get more expensive); volume-like graphs, whose root separator alone is larger, keep the plain tree.
  // the depth a perfectly balanced dissection needs (nd_recurse: max_depth)
  auto depth_cap = [&](int leaf) {
    int dcap = 1;
    while (((long)leaf << dcap) < (long)all.size()) ++dcap;
    return dcap;
  };
  SplitMemo memo;
  if ((int)all.size() >= 30000) {  // (agents of a session are set up side by side on host threads: small graphs stay serial)
    const int nthreads = std::max(1, std::min(host_cpus_available(), 16));
    if (nthreads > 1) prefill_splits(G, all, leaf_nodes, depth_cap(leaf_nodes), nthreads, &memo);
  }
  int top_depth = 0;
  bool have_plain = false;  // the probe below IS the final dissection when no dense top comes out of it
  {
    if (top_unknowns > 0 && (int)all.size() >= 16 * leaf_nodes) {
      std::vector<long> by_depth;
      std::vector<int> comp0(nb, -1), level0(nb, -1), border0, cuts0;
      std::vector<std::pair<int, int>> tasks0;
      std::vector<std::vector<std::pair<int, int>>> waves0;
      int cid0 = 0;
      nd_recurse(G, all, comp0, cid0, level0, border0, cuts0, leaf_nodes, 0, tasks0, waves0, 0, &by_depth,
                 depth_cap(leaf_nodes), &memo);
      // as many depths as fit top_unknowns -- but no more than pay: with D separator depths in all, a top of t depths
      // leaves D - t + 1 levels below it, applied in pairs (2 ceil((D - t + 1) / 2) + 1 launches of ~10 us), and costs its
      // dense symmetric product (c_t^2 doubles at ~5 TB/s); the cheapest t wins, the smaller on a tie (tiers.pyfg: an
      // eighth depth fits, saves no launch and doubles the bytes of the top)
      long sum = 0;
      int t_fit = 0;
      std::vector<long> cum;
      for (size_t dpt = 0; dpt < by_depth.size(); ++dpt) {
        sum += by_depth[dpt] * block;
        if (sum > top_unknowns) break;
        cum.push_back(sum);
        t_fit = (int)dpt + 1;
      }
      const int D = (int)by_depth.size();
      // ... and leaves twice as large take one more depth away (their dense blocks double: 8 n block^2 leaf bytes).
      // Measured per application of the replay with the BFS-level separators of round 2, leaves of 96 / 192 / 384 unknowns:
      // sphere2500 49 / 41 / 39 us, torus3D 89 / 75 / 74, tiers.pyfg 72 / 65 / 82, a lattice agent 169 / 190 / 235 -- the
      // model below reproduces those choices and follows the smaller separators of round 4.
      double best_cost = 0;
      int best_mult = 1;
      for (int mult = 1; mult <= 2; ++mult)
        for (int t = 2; t <= t_fit; ++t) {
          const int below = std::max(1, D - (mult - 1) - t + 1);
          const double bytes = 8.0 * (double)cum[(size_t)t - 1] * (double)cum[(size_t)t - 1] +
                               8.0 * (double)all.size() * block * block * leaf_nodes * mult;
          const double cost = 10.0 * (2 * ((below + 1) / 2) + 1) + bytes / 5e6;
          if (top_depth == 0 || cost < best_cost) {
            best_cost = cost;
            top_depth = t;
            best_mult = mult;
          }
        }
      leaf_nodes *= best_mult;
      if (top_depth == 0 && task_nodes == 0) {  // volume-like graph: the same recursion would run again (90 ms at k = 400 000)
        comp_id.swap(comp0);
        level.swap(level0);
        border.swap(border0);
        cuts.swap(cuts0);
        next_cid = cid0;
        have_plain = true;
      }
      if (env::init_timing()) {
        std::fprintf(stderr, "[order] %d nodes: %d separator depths, %d fit the dense top, dense top of %d depths (%ld unknowns), leaves of %d nodes\n",
                     (int)all.size(), D, t_fit, top_depth, top_depth > 0 ? cum[(size_t)top_depth - 1] : 0L, leaf_nodes);
      }
    }
  }
  if (!have_plain)
    nd_recurse(G, all, comp_id, next_cid, level, border, cuts, leaf_nodes, task_nodes, tasks, waves, top_depth, nullptr,
               depth_cap(leaf_nodes), &memo);
  if (col_tasks) {
    col_tasks->clear();
    for (const auto &t : tasks)
      col_tasks->emplace_back(std::min(n, t.first * block), std::min(n, (t.first + t.second) * block));
  }
  if (col_waves) {
    col_waves->clear();
    for (const auto &wv : waves) {
      col_waves->emplace_back();
      for (const auto &t : wv)
        col_waves->back().emplace_back(std::min(n, t.first * block), std::min(n, (t.first + t.second) * block));
    }
  }
  if (!hubs.empty()) {
    border.insert(border.end(), hubs.begin(), hubs.end());
    cuts.push_back(nb);
  }
  if (pieces) {
    pieces->clear();
    for (int c : cuts) pieces->push_back(std::min(n, c * block));
    pieces->erase(std::unique(pieces->begin(), pieces->end()), pieces->end());
  }
  std::vector<int> perm;
  perm.reserve(n);
  int hub_cols = 0;
  for (size_t q = 0; q < border.size(); ++q) {
    const int bn = border[q];
    for (int t = 0; t < block; ++t)
      if (bn * block + t < n) {
        perm.push_back(bn * block + t);
        if (q >= border.size() - hubs.size()) ++hub_cols;
      }
  }
  if (nhub_cols) *nhub_cols = hub_cols;
  return perm;
}

// ---------------------------------------------------------------------------------------------------
// Numeric factorisation: up-looking LL^T over the elimination tree
// ---------------------------------------------------------------------------------------------------
bool SparseChol::factor(const HostCsr &A, int block, int top_unknowns) {
  n_ = A.n;
  ok_ = false;
  const int n = n_;
  // sub-tree parallel numeric phase for matrices that are worth it
  int nthreads = (n >= 4096) ? std::max(1, std::min(host_cpus_available(), 16)) : 1;
  std::vector<std::pair<int, int>> tasks;
  std::vector<std::vector<std::pair<int, int>>> waves;  // separators above the tasks, by dissection depth
  perm_ = amd_like_order(A, block, &pieces_, &nhub_, &tasks, 4 * nthreads, &waves, top_unknowns);
  const bool timing = env::init_timing();
  auto tnow = [] { return std::chrono::steady_clock::now(); };
  auto tms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
  const auto T0 = tnow();
  iperm_.assign(n, 0);
  for (int i = 0; i < n; ++i) iperm_[perm_[i]] = i;
  // upper triangle of P A P^T by columns
  std::vector<int> Cp(n + 1, 0);
  for (int io = 0; io < n; ++io) {
    const int i = iperm_[io];
    for (int p = A.rp[io]; p < A.rp[io + 1]; ++p) {
      const int j = iperm_[A.ci[p]];
      if (i <= j) Cp[j + 1]++;
    }
  }
  for (int j = 0; j < n; ++j) Cp[j + 1] += Cp[j];
  std::vector<int> Ci(Cp[n]);
  std::vector<double> Cx(Cp[n]);
  {
    std::vector<int> pos(Cp.begin(), Cp.end() - 1);
    for (int io = 0; io < n; ++io) {
      const int i = iperm_[io];
      for (int p = A.rp[io]; p < A.rp[io + 1]; ++p) {
        const int j = iperm_[A.ci[p]];
        if (i <= j) {
          Ci[pos[j]] = i;
          Cx[pos[j]++] = A.v[p];
        }
      }
    }
  }
  const auto T1 = tnow();
  std::vector<int> parent(n, -1), anc(n, -1);
  for (int k = 0; k < n; ++k)
    for (int p = Cp[k]; p < Cp[k + 1]; ++p) {
      int i = Ci[p];
      while (i != -1 && i < k) {
        const int nx = anc[i];
        anc[i] = k;
        if (nx == -1) parent[i] = k;
        i = nx;
      }
    }
  std::vector<int> cnt(n, 1), mark(n, -1);
  for (int k = 0; k < n; ++k) {
    mark[k] = k;
    for (int p = Cp[k]; p < Cp[k + 1]; ++p) {
      int i = Ci[p];
      while (i < k && mark[i] != k) {
        cnt[i]++;
        mark[i] = k;
        i = parent[i];
      }
    }
  }
  const auto T2 = tnow();
  Lp_.assign(n + 1, 0);
  for (int j = 0; j < n; ++j) Lp_[j + 1] = Lp_[j] + cnt[j];
  Li_.assign(Lp_[n], 0);
  Lx_.assign(Lp_[n], 0.0);
  std::vector<int> fill(Lp_.begin(), Lp_.end() - 1);
  // Row k of L only needs the columns of its elimination sub-tree, so disjoint dissection sub-trees (col_tasks) are
  // factorised by separate threads; the separators above them follow in order on the calling thread.
  struct Work {
    std::vector<int> mark, stk, pat;
    std::vector<double> x;
    explicit Work(int n) : mark(n, -1), stk(n), pat(n), x(n, 0.0) {}
  };
  auto process_row = [&](int k, Work &w) -> bool {
    int top = n;
    w.mark[k] = k;
    double dk = 0;
    for (int p = Cp[k]; p < Cp[k + 1]; ++p) {
      int i = Ci[p];
      if (i == k) {
        dk += Cx[p];
        continue;
      }
      w.x[i] += Cx[p];
      int len = 0;
      while (w.mark[i] != k) {
        w.stk[len++] = i;
        w.mark[i] = k;
        i = parent[i];
      }
      while (len > 0) w.pat[--top] = w.stk[--len];
    }
    for (; top < n; ++top) {
      const int i = w.pat[top];
      const double lki = w.x[i] / Lx_[Lp_[i]];
      w.x[i] = 0;
      for (int p = Lp_[i] + 1; p < fill[i]; ++p) w.x[Li_[p]] -= Lx_[p] * lki;
      dk -= lki * lki;
      const int q = fill[i]++;
      Li_[q] = k;
      Lx_[q] = lki;
    }
    if (!(dk > 0)) return false;
    const int q = fill[k]++;
    Li_[q] = k;
    Lx_[q] = std::sqrt(dk);
    return true;
  };
  const auto T3 = tnow();
  std::vector<char> in_task(n, 0);
  std::atomic<bool> failed(false);
  if (tasks.size() > 1 && nthreads > 1) {
    std::sort(tasks.begin(), tasks.end(),
              [](const auto &a, const auto &b) { return a.second - a.first > b.second - b.first; });
    for (const auto &t : tasks)
      for (int k = t.first; k < t.second; ++k) in_task[k] = 1;
    std::atomic<int> next(0);
    auto worker = [&]() {
      Work w(n);
      for (;;) {
        const int ti = next.fetch_add(1);
        if (ti >= (int)tasks.size() || failed.load()) break;
        for (int k = tasks[ti].first; k < tasks[ti].second; ++k)
          if (!process_row(k, w)) {
            failed.store(true);
            break;
          }
      }
    };
    const int nt = std::min<int>(nthreads, (int)tasks.size());
    std::vector<std::thread> th;
    for (int t = 1; t < nt; ++t) th.emplace_back(worker);
    worker();
    for (auto &t : th) t.join();
    if (failed.load()) return false;
    // the separators above the tasks, deepest wave first; the separators of one wave are independent
    for (int dep = (int)waves.size() - 1; dep >= 1; --dep) {
      std::vector<std::pair<int, int>> &wv = waves[dep];
      if (wv.size() < 2) break;  // from here up the calling thread does the rest in order
      for (const auto &t : wv)
        for (int k = t.first; k < t.second; ++k) in_task[k] = 1;
      std::atomic<int> nxt(0);
      auto wworker = [&]() {
        Work w(n);
        for (;;) {
          const int ti = nxt.fetch_add(1);
          if (ti >= (int)wv.size() || failed.load()) break;
          for (int k = wv[ti].first; k < wv[ti].second; ++k)
            if (!process_row(k, w)) {
              failed.store(true);
              break;
            }
        }
      };
      const int wt = std::min<int>(nthreads, (int)wv.size());
      std::vector<std::thread> wth;
      for (int t = 1; t < wt; ++t) wth.emplace_back(wworker);
      wworker();
      for (auto &t : wth) t.join();
      if (failed.load()) return false;
    }
  }
  const auto T4 = tnow();
  {
    Work w(n);
    for (int k = 0; k < n; ++k)
      if (!in_task[k] && !process_row(k, w)) return false;
  }
  if (timing)
    std::fprintf(stderr, "[factor] n %d nnzL %d: permute %.2f, etree+counts %.2f, alloc %.2f, subtrees (%zu tasks, %d thr) %.2f, top %.2f ms\n",
                 n, Lp_[n], tms(T0, T1), tms(T1, T2), tms(T2, T3), tasks.size(), nthreads, tms(T3, T4), tms(T4, tnow()));
  ok_ = true;
  return true;
}

void SparseChol::solve_inplace(double *B, int nrhs) const {
  const int n = n_;
  for (int j = 0; j < n; ++j) {
    double *bj = B + (size_t)j * nrhs;
    const double inv = 1.0 / Lx_[Lp_[j]];
    for (int t = 0; t < nrhs; ++t) bj[t] *= inv;
    for (int p = Lp_[j] + 1; p < Lp_[j + 1]; ++p) {
      double *bi = B + (size_t)Li_[p] * nrhs;
      const double l = Lx_[p];
      for (int t = 0; t < nrhs; ++t) bi[t] -= l * bj[t];
    }
  }
  for (int j = n - 1; j >= 0; --j) {
    double *bj = B + (size_t)j * nrhs;
    for (int p = Lp_[j] + 1; p < Lp_[j + 1]; ++p) {
      const double *bi = B + (size_t)Li_[p] * nrhs;
      const double l = Lx_[p];
      for (int t = 0; t < nrhs; ++t) bj[t] -= l * bi[t];
    }
    const double inv = 1.0 / Lx_[Lp_[j]];
    for (int t = 0; t < nrhs; ++t) bj[t] *= inv;
  }
}

void SparseChol::solve_vec(const double *b, double *x) const {
  std::vector<double> y(n_);
  for (int i = 0; i < n_; ++i) y[i] = b[perm_[i]];
  solve_inplace(y.data(), 1);
  for (int i = 0; i < n_; ++i) x[perm_[i]] = y[i];
}

void SparseChol::dense_inverse(double *out, size_t ld, int nthreads) const {
  const int n = n_;
  constexpr int NB = 16;
  const int nblk = (n + NB - 1) / NB;
  nthreads = std::max(1, std::min(nthreads, nblk));
  auto work = [&](int tid) {
    std::vector<double> B((size_t)n * NB);
    for (int blk = tid; blk < nblk; blk += nthreads) {
      const int c0 = blk * NB, nc = std::min(NB, n - c0);
      std::fill(B.begin(), B.end(), 0.0);
      // columns c0..c0+nc-1 of the inverse in ORIGINAL numbering: rhs e_c -> permuted position iperm[c]
      for (int t = 0; t < nc; ++t) B[(size_t)iperm_[c0 + t] * NB + t] = 1.0;
      solve_inplace(B.data(), NB);
      for (int t = 0; t < nc; ++t) {
        double *row = out + (size_t)(c0 + t) * ld;
        for (int i = 0; i < n; ++i) row[perm_[i]] = B[(size_t)i * NB + t];
      }
    }
  };
  std::vector<std::thread> th;
  for (int t = 1; t < nthreads; ++t) th.emplace_back(work, t);
  work(0);
  for (auto &t : th) t.join();
}

}  // namespace dcora
