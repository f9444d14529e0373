// Second schedule of the partitioned inverse (sparse_precond.h): fewer dependent launches per application of
// (Q + reg I)^-1 (ref src/QuadraticProblem.cpp:70-84; the factor of ref src/Graph.cpp:1901-1917).
//
// The first schedule (host_partinv.cpp) applies  L^-1 = prod L_s^-1  and  L^-T  one tree level per launch: 2 depth - 1
// dependent launches, each a burst of a few MB that costs the launch floor plus three dependent memory round trips
// whatever it moves.  Two changes, same arithmetic (the solve is the same product of the same factors, regrouped):
//
//  1. The diagonal blocks leave the dependent chain.  With y_s the value of piece s once every piece below it has
//     been applied ("pre" value), forward and backward substitution read
//         y_q  = b_q + sum_{s below q} W_s[q] y_s                 W_s = -B_s D_s^-1  (rows of the pieces above s)
//         x_s  = M_s y_s + W_s^T x_{rows(s)}                      M_s = D_s^-T D_s^-1
//     so D_s^-1 y_s alone is never needed: the two triangular products of a piece become ONE symmetric product
//     z_s = M_s y_s (the same bytes), which nothing waits for until the backward sweep reaches s.  Those "M tiles" are
//     spread over the launches between the one that completes y_s and the one that consumes z_s so that every launch
//     streams about the same number of bytes while its dependent tiles wait for their operands.
//
//  2. Levels are merged in pairs (t, t + 1): for s on level t and p on level t + 1 the product
//         W'_s[q] = W_s[q] + sum_p W_p[q] W_s[p]                  (q above level t + 1)
//     is formed once, at set-up; the pair's forward launch then updates the rows of level t + 1 (through W_s[p]) AND all
//     rows above (through W'_s from the level-t values and through W_p from the not yet updated level-(t+1) values) at
//     the same time, and the pair's backward launch computes x_p and x_s together (x_s = z_s + W_s[p]^T z_p +
//     W'_s^T x_above).  Half the launches for the fill of W'_s (rows(s) united with rows(p)): +16-25 % stored weights on
//     the graphs measured.  A pair is merged only while its fill and its set-up products stay small next to the launch
//     they save (never on the whole 100k lattice as one problem, whose replay is bandwidth-bound anyway).
//
// The tasks of a launch now differ widely in length, so they are sorted by the entries they gather and every tile gets
// the lanes its own length asks for (SpLevel::cls, k_sp_multi).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>

#include "host_partinv_int.h"

namespace dcora {
namespace partinv {

namespace {

constexpr int RT = kSpTile;
constexpr int kBig = 384;

struct Source {        // what a piece contributes to the rows above it in one forward launch / gathers in a backward one
  int s = -1;          // piece
  const int *rows = nullptr;  // global permuted rows, ascending
  int m = 0;
  const double *V = nullptr;  // m x c row-major
};

// entries a tile gathers -> lanes it runs on (regime 0: r >= 4, 1: r < 4).  Twice sp_pick_lanes' thresholds: those were
// measured on launches of one tree level each, short bursts bound by their dependent loads; a merged launch streams
// ~30 MB, and there the RT r sums a tile reduces over its lanes (14 of 129 us on an agent of the 100k lattice) weigh more
// than the steps a lane walks.  Measured (us per application, agent of the 100k lattice, r = 5 / 7):
// x1 129.5 / 156.8, x2 114.5 / 143.7, x3 132 / 141, x4 137.7 / 148.7, x8 171 / 180.  Tiles of 8 and 4 lanes for the
// shortest lists, and lanes chosen as entries / steps for 3 .. 12 steps per lane, were measured as well: no gain
// (119-186 us) -- fewer lanes mean fewer sums to reduce but longer dependent walks, and the walk costs more.
int lanes_class(long entries, int regime) {
  static const double scale_env = [] {
    const char *e = std::getenv("DCORA_SP_CLS_SCALE");
    return e ? atof(e) : 0.0;
  }();
  // r < 4 (tiers.pyfg, r = 2: RT r = 8 sums per tile) keeps the thresholds: 14 640 tCG it/s at x1 against 14 060 at x2
  const double scale = scale_env > 0 ? scale_env : (regime == 0 ? 2.0 : 1.0);
  const double t64 = scale * (regime == 0 ? 100 : 20), t32 = scale * (regime == 0 ? 30 : 8);
  return entries >= scale * 400 ? 0 : entries >= scale * 160 ? 1 : entries >= t64 ? 2 : entries >= t32 ? 3 : 4;
}

}  // namespace

void layout_merged(const std::vector<Piece> &pc, const std::vector<const double *> &Mgiven,
                   const std::vector<int> &piece_of, int k, int nlev, int nthreads, bool timing, PartInvHost *out) {
  PartInvHost &P = *out;
  auto tnow = [] { return std::chrono::steady_clock::now(); };
  auto tms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
  const auto T0 = tnow();
  const int np = (int)pc.size();
  nthreads = std::max(1, nthreads);
  std::vector<std::vector<int>> by_level((size_t)nlev);
  for (int s = 0; s < np; ++s) by_level[pc[s].level].push_back(s);

  // ---- M_s = D_s^-T D_s^-1 of every piece (symmetric, both triangles) ----
  std::vector<std::vector<double>> Mown((size_t)np);
  std::vector<const double *> M((size_t)np, nullptr);
  {
    std::vector<int> small;
    for (int s = 0; s < np; ++s) {
      if (Mgiven[s]) {
        M[s] = Mgiven[s];
        continue;
      }
      const int c = pc[s].c;
      Mown[s].assign((size_t)c * c, 0.0);
      M[s] = Mown[s].data();
      if (c >= kBig && nthreads > 1) {
        // row a of the lower triangle by one thread: M(a, j) = sum_{i >= a} Dinv(i, a) Dinv(i, j), j <= a
        std::vector<double> &Ms = Mown[s];
        const double *Dinv = pc[s].dinv();
        parallel_for(c, nthreads, 4, [&](int a) {
          double *ma = &Ms[(size_t)a * c];
          for (int i = a; i < c; ++i) {
            const double *di = &Dinv[(size_t)i * c];
            const double v = di[a];
            if (v == 0.0) continue;
            for (int j = 0; j <= a; ++j) ma[j] += v * di[j];
          }
        });
        for (int a = 0; a < c; ++a)
          for (int j = a + 1; j < c; ++j) Ms[(size_t)a * c + j] = Ms[(size_t)j * c + a];
      } else {
        small.push_back(s);
      }
    }
    std::sort(small.begin(), small.end(), [&](int a, int b) { return pc[a].c > pc[b].c; });
    parallel_for((int)small.size(), nthreads, 1, [&](int t) {
      const int s = small[(size_t)t], c = pc[s].c;
      std::vector<double> &Ms = Mown[s];
      const double *Dinv = pc[s].dinv();
      for (int i = 0; i < c; ++i) {
        const double *di = &Dinv[(size_t)i * c];
        for (int a = 0; a <= i; ++a) {
          const double v = di[a];
          if (v == 0.0) continue;
          double *ma = &Ms[(size_t)a * c];
          for (int j = 0; j <= a; ++j) ma[j] += v * di[j];
        }
      }
      for (int a = 0; a < c; ++a)
        for (int j = a + 1; j < c; ++j) Ms[(size_t)a * c + j] = Ms[(size_t)j * c + a];
    });
  }
  const auto T1 = tnow();

  // ---- which levels merge.  Levels below the top: 0 .. nlo - 1; the top level is a launch of its own. ----
  const int nlo = nlev - 1;
  static const bool merge_on = [] {
    const char *e = std::getenv("DCORA_SP_MERGE");
    return !(e && std::strcmp(e, "0") == 0);
  }();
  // symbolic fill of a candidate pair (t, t + 1): rows W'_s gains over W_s, and the flops of the set-up products
  std::vector<std::vector<int>> merged_rows((size_t)np);  // rows of V_s for the lower pieces of CHOSEN pairs
  auto union_rows = [&](int s, int t, std::vector<int> *rows_out, double *flops) {
    const Piece &ps = pc[s];
    std::vector<int> acc(ps.rows);
    int last = -1;
    for (size_t a = 0; a < ps.rows.size(); ++a) {
      const int q = piece_of[ps.rows[a]];
      if (q == last) continue;
      last = q;
      if (pc[q].level != t + 1) continue;
      size_t b = a;
      while (b < ps.rows.size() && piece_of[ps.rows[b]] == q) ++b;
      if (flops) *flops += 2.0 * (double)pc[q].rows.size() * (double)(b - a) * ps.c;
      std::vector<int> tmp;
      tmp.reserve(acc.size() + pc[q].rows.size());
      std::set_union(acc.begin(), acc.end(), pc[q].rows.begin(), pc[q].rows.end(), std::back_inserter(tmp));
      acc.swap(tmp);
    }
    rows_out->swap(acc);
  };
  std::vector<double> pair_bytes((size_t)std::max(nlo, 1), 0.0), pair_flops((size_t)std::max(nlo, 1), 0.0);
  if (merge_on)
    for (int t = 0; t + 1 < nlo; ++t) {
      std::vector<int> rows;
      for (int s : by_level[t]) {
        union_rows(s, t, &rows, &pair_flops[t]);
        pair_bytes[t] += 16.0 * (double)(rows.size() - pc[s].rows.size()) * pc[s].c;  // both sweeps
      }
    }
  // a launch saved is worth ~6 us = ~25 MB of streaming; the products run on the host's threads at set-up
  static const double max_bytes = [] {
    const char *e = std::getenv("DCORA_SP_MERGE_MB");
    return 1e6 * (e ? atof(e) : 50.0);
  }();
  static const double max_flops = 12e9;
  auto pair_ok = [&](int t) { return merge_on && pair_bytes[t] <= max_bytes && pair_flops[t] <= max_flops; };
  // most pairs first, least fill second: best[i] = (pairs, -bytes) over levels i .. nlo - 1
  std::vector<int> group_of((size_t)std::max(nlo, 1), 0), is_pair_lo((size_t)std::max(nlo, 1), 0);
  {
    std::vector<int> cnt((size_t)nlo + 2, 0);
    std::vector<double> cost((size_t)nlo + 2, 0.0);
    std::vector<char> take((size_t)nlo + 2, 0);
    for (int i = nlo - 1; i >= 0; --i) {
      cnt[i] = cnt[i + 1];
      cost[i] = cost[i + 1];
      take[i] = 0;
      if (i + 1 < nlo && pair_ok(i)) {
        const int c2 = cnt[i + 2] + 1;
        const double b2 = cost[i + 2] + pair_bytes[i];
        if (c2 > cnt[i] || (c2 == cnt[i] && b2 < cost[i])) {
          cnt[i] = c2;
          cost[i] = b2;
          take[i] = 1;
        }
      }
    }
    for (int i = 0; i < nlo;) {
      if (take[i]) {
        is_pair_lo[i] = 1;
        i += 2;
      } else {
        i += 1;
      }
    }
  }
  struct Group {
    int lo, hi;  // hi = -1: single level
  };
  std::vector<Group> groups;
  for (int i = 0; i < nlo;) {
    if (is_pair_lo[i]) {
      groups.push_back({i, i + 1});
      group_of[i] = group_of[i + 1] = (int)groups.size() - 1;
      i += 2;
    } else {
      groups.push_back({i, -1});
      group_of[i] = (int)groups.size() - 1;
      i += 1;
    }
  }
  const int NG = (int)groups.size();
  const int NL = 2 * NG + 1;  // forward groups, the top, backward groups

  // ---- V_s of the lower pieces of merged pairs: rows(s) united with the rows of the level-(t+1) pieces s feeds ----
  std::vector<std::vector<double>> Vown((size_t)np);
  {
    std::vector<int> todo;
    for (const Group &g : groups)
      if (g.hi >= 0)
        for (int s : by_level[g.lo]) todo.push_back(s);
    std::sort(todo.begin(), todo.end(), [&](int a, int b) {
      return (double)pc[a].c * pc[a].rows.size() > (double)pc[b].c * pc[b].rows.size();
    });
    parallel_for((int)todo.size(), nthreads, 1, [&](int ti) {
      const int s = todo[(size_t)ti];
      const Piece &ps = pc[s];
      const int t = ps.level, c = ps.c, m = (int)ps.rows.size();
      std::vector<int> &R = merged_rows[s];
      union_rows(s, t, &R, nullptr);
      std::vector<double> &V = Vown[s];
      V.assign(R.size() * (size_t)c, 0.0);
      // own rows
      {
        size_t u = 0;
        for (int a = 0; a < m; ++a) {
          while (R[u] != ps.rows[a]) ++u;
          std::copy(ps.w() + (size_t)a * c, ps.w() + (size_t)a * c + c, &V[u * c]);
        }
      }
      // through the pieces of level t + 1
      for (int a = 0; a < m;) {
        const int q = piece_of[ps.rows[a]];
        int b = a;
        while (b < m && piece_of[ps.rows[b]] == q) ++b;
        if (pc[q].level == t + 1) {
          const Piece &pq = pc[q];
          const int cq = pq.c, mq = (int)pq.rows.size();
          size_t u = 0;
          for (int bb = 0; bb < mq; ++bb) {
            while (R[u] != pq.rows[bb]) ++u;
            double *dst = &V[u * c];
            const double *wq = pq.w() + (size_t)bb * cq;
            for (int aa = a; aa < b; ++aa) {
              const double coef = wq[ps.rows[aa] - pq.c0];
              if (coef == 0.0) continue;
              const double *ws = ps.w() + (size_t)aa * c;
              for (int j = 0; j < c; ++j) dst[j] += coef * ws[j];
            }
          }
        }
        a = b;
      }
    });
  }
  const auto T2 = tnow();

  // what a piece gathers from / scatters to in its group's launches
  auto source_of = [&](int s) {
    Source S;
    S.s = s;
    if (!merged_rows[s].empty() || !Vown[s].empty()) {
      S.rows = merged_rows[s].data();
      S.m = (int)merged_rows[s].size();
      S.V = Vown[s].data();
    } else {
      S.rows = pc[s].rows.data();
      S.m = (int)pc[s].rows.size();
      S.V = pc[s].w();
    }
    return S;
  };

  // ---- lay the launches out ----
  std::vector<int> cur((size_t)np, 0);    // buffer that holds the piece's current value
  std::vector<int> ybuf((size_t)np, -1);  // buffer of its pre value, once that is complete
  auto pos = [&](int b, int row) { return b * k + row; };
  std::vector<Fill> fills;
  long long cursor = 0;
  double weights = 0;
  auto reserve = [&](int kind, int nrows, int len, const double *base, int c, int a0, int m, const int *loc) {
    Fill f;
    f.off = cursor;
    f.base = base;
    f.kind = kind;
    f.nrows = nrows;
    f.len = len;
    f.c = c;
    f.a0 = a0;
    f.m = m;
    for (int q = 0; q < RT; ++q) f.loc[q] = loc ? loc[q] : 0;
    fills.push_back(f);
    cursor += (long long)len * nrows;
    weights += (double)len * nrows;
    return f.off;
  };
  struct Launch {
    std::vector<PTask> tasks;
    double bytes = 0;
  };
  std::vector<Launch> L((size_t)NL);
  auto task_entries = [&](const PTask &T) {
    long e = 0;
    for (int q = T.seg0; q < T.seg0 + T.nseg; ++q) e += P.segs[(size_t)q].len;
    return e;
  };
  auto push_task = [&](Launch &Ln, PTask &T) {
    T.nseg = (int)P.segs.size() - T.seg0;
    inline_first_segment(T, P.segs);
    Ln.bytes += 8.0 * (double)task_entries(T) * T.nrows + 64.0;
    Ln.tasks.push_back(T);
  };

  // M tiles are placed afterwards; remember between which launches each piece's tiles may run
  std::vector<int> m_first((size_t)np, 0), m_last((size_t)np, 0);

  // forward launches
  std::vector<std::vector<std::pair<int, int>>> hits((size_t)k);  // per row: (source index, local row of its V)
  std::vector<int> touched((size_t)np, -1);
  for (int g = 0; g < NG; ++g) {
    const Group &G = groups[g];
    Launch &Ln = L[(size_t)g];
    std::vector<Source> srcs;
    for (int s : by_level[G.lo]) srcs.push_back(source_of(s));
    if (G.hi >= 0)
      for (int s : by_level[G.hi]) srcs.push_back(source_of(s));
    for (int s : by_level[G.lo]) {
      ybuf[s] = cur[s];
      m_first[s] = g;
      m_last[s] = 2 * NG - g - 1;
    }
    std::vector<int> affected, hit_rows;
    for (int si = 0; si < (int)srcs.size(); ++si) {
      const Source &S = srcs[(size_t)si];
      for (int a = 0; a < S.m; ++a) {
        const int i = S.rows[a], q = piece_of[i];
        if (touched[q] != g) {
          touched[q] = g;
          affected.push_back(q);
        }
        if (hits[i].empty()) hit_rows.push_back(i);
        hits[i].emplace_back(si, a);
      }
    }
    // a level-(hi) piece nothing feeds still counts as complete after this launch: no copy needed, it stays in cur
    for (int q : affected) {
      const Piece &p = pc[q];
      for (int a0 = 0; a0 < p.c;) {
        int nrows = 1;
        const auto &h0 = hits[p.c0 + a0];
        while (nrows < RT && a0 + nrows < p.c) {
          const auto &h1 = hits[p.c0 + a0 + nrows];
          bool same = h1.size() == h0.size();
          for (size_t u = 0; same && u < h0.size(); ++u) same = (h1[u].first == h0[u].first);
          if (!same) break;
          ++nrows;
        }
        PTask T;
        T.out = pos(1 - cur[q], p.c0 + a0);
        T.carry = pos(cur[q], p.c0 + a0);
        T.nrows = nrows;
        T.seg0 = (int)P.segs.size();
        for (size_t u = 0; u < h0.size(); ++u) {
          const Source &S = srcs[(size_t)h0[u].first];
          const Piece &s = pc[S.s];
          int loc[RT] = {0};
          for (int r_ = 0; r_ < nrows; ++r_) loc[r_] = hits[p.c0 + a0 + r_][u].second;
          PSeg Sg;
          Sg.len = pad2(s.c);
          Sg.src = pos(cur[S.s], s.c0);  // a source of the launch is never one of its destinations' NEW values
          Sg.idx = 0;
          Sg.pad = 0;
          Sg.w = reserve(1, nrows, Sg.len, S.V, s.c, 0, 0, loc);
          P.segs.push_back(Sg);
        }
        push_task(Ln, T);
        a0 += nrows;
      }
    }
    for (int i : hit_rows) hits[i].clear();
    for (int q : affected) cur[q] ^= 1;
    if (G.hi >= 0)
      for (int s : by_level[G.hi]) {
        ybuf[s] = cur[s];
        m_first[s] = g + 1;
        m_last[s] = 2 * NG - g - 1;
      }
  }
  // the top level: its pieces feed nobody; their M tiles ARE the top launch
  for (int s : by_level[nlev - 1]) {
    ybuf[s] = cur[s];
    m_first[s] = m_last[s] = NG;
  }
  // pieces below the top without rows (a component of its own): x = z, any launch will do
  // backward launches
  std::vector<char> has_back((size_t)np, 0);
  for (int g = NG - 1; g >= 0; --g) {
    const Group &G = groups[g];
    Launch &Ln = L[(size_t)(2 * NG - g)];
    auto emit = [&](int s, int pair_hi_level) {
      const Piece &p = pc[s];
      const Source S = source_of(s);
      if (S.m == 0) return;
      has_back[s] = 1;
      const int idx0 = (int)P.idxs.size();
      for (int a = 0; a < S.m; ++a) {
        const int i = S.rows[a], q = piece_of[i];
        // a row of the pair's upper level: its z (the pair's launch computes x_q at the same time); else the final x
        const bool upper = pc[q].level == pair_hi_level;
        const int b = upper ? 1 - ybuf[q] : (pc[q].rows.empty() ? 1 - ybuf[q] : ybuf[q]);
        P.idxs.push_back(pos(b, i));
      }
      if (S.m & 1) P.idxs.push_back(P.idxs.back());
      for (int a0 = 0; a0 < p.c; a0 += RT) {
        const int nrows = std::min(RT, p.c - a0);
        PTask T;
        T.out = pos(ybuf[s], p.c0 + a0);
        T.carry = pos(1 - ybuf[s], p.c0 + a0);
        T.nrows = nrows;
        T.seg0 = (int)P.segs.size();
        PSeg Wt;
        Wt.len = pad2(S.m);
        Wt.src = -1;
        Wt.idx = idx0;
        Wt.pad = 0;
        Wt.w = reserve(4, nrows, Wt.len, S.V, p.c, a0, S.m, nullptr);
        P.segs.push_back(Wt);
        push_task(Ln, T);
      }
    };
    if (G.hi >= 0)
      for (int s : by_level[G.hi]) emit(s, -1);
    for (int s : by_level[G.lo]) emit(s, G.hi);
  }
  // ---- M tiles, spread: earliest deadline first, up to an even share of the bytes that are left ----
  {
    std::vector<std::vector<int>> avail_at((size_t)NL);
    for (int s = 0; s < np; ++s) avail_at[(size_t)m_first[s]].push_back(s);
    double total = 0;
    for (const Launch &Ln : L) total += Ln.bytes;
    for (int s = 0; s < np; ++s) total += 8.0 * (double)pc[s].c * pad2(pc[s].c);
    // pending pieces ordered by deadline; within a piece tiles go in row order
    std::vector<std::pair<int, int>> pending;  // (deadline, piece), kept sorted
    std::vector<int> next_row((size_t)np, 0);
    double placed_so_far = 0;
    for (int l = 0; l < NL; ++l) {
      for (int s : avail_at[(size_t)l]) pending.emplace_back(m_last[s], s);
      std::sort(pending.begin(), pending.end());
      Launch &Ln = L[(size_t)l];
      const double share = (total - placed_so_far) / (NL - l);
      size_t pi = 0;
      while (pi < pending.size()) {
        const int dl = pending[pi].first, s = pending[pi].second;
        const Piece &p = pc[s];
        const bool must = dl <= l;
        bool finished = false;
        while (next_row[s] < p.c) {
          if (!must && Ln.bytes >= share) break;
          const int a0 = next_row[s], nrows = std::min(RT, p.c - a0);
          PTask T;
          T.out = pos(1 - ybuf[s], p.c0 + a0);
          T.carry = -1;
          T.nrows = nrows;
          T.seg0 = (int)P.segs.size();
          PSeg Sg;
          Sg.len = pad2(p.c);
          Sg.src = pos(ybuf[s], p.c0);
          Sg.idx = 0;
          Sg.pad = 0;
          Sg.w = reserve(2, nrows, Sg.len, M[s], p.c, a0, 0, nullptr);
          P.segs.push_back(Sg);
          push_task(Ln, T);
          next_row[s] += nrows;
        }
        finished = next_row[s] >= p.c;
        if (finished) {
          pending.erase(pending.begin() + (long)pi);
        } else {
          break;  // the share of this launch is used up (pieces with later deadlines wait as well)
        }
      }
      placed_so_far += Ln.bytes;
    }
  }
  // ---- launches -> P.levels / P.tasks, tasks sorted by the entries they gather ----
  for (int l = 0; l < NL; ++l) {
    Launch &Ln = L[(size_t)l];
    std::vector<std::pair<long, int>> order(Ln.tasks.size());
    for (size_t i = 0; i < Ln.tasks.size(); ++i) order[i] = {-task_entries(Ln.tasks[i]), (int)i};
    std::sort(order.begin(), order.end());
    SpLevel lv;
    lv.task0 = (int)P.tasks.size();
    lv.ntasks = (int)Ln.tasks.size();
    lv.multi = 1;
    long long sum = 0;
    for (int g2 = 0; g2 < 2; ++g2)
      for (int cidx = 0; cidx < 4; ++cidx) lv.cls[g2][cidx] = lv.ntasks;
    for (size_t i = 0; i < order.size(); ++i) {
      const long e = -order[i].first;
      sum += e;
      for (int g2 = 0; g2 < 2; ++g2) {
        const int cl = lanes_class(e, g2);  // 0 .. 4: 256 .. 16 lanes per tile
        for (int cidx = 0; cidx < cl && cidx < 4; ++cidx) lv.cls[g2][cidx] = std::min(lv.cls[g2][cidx], (int)i);
      }
      P.tasks.push_back(Ln.tasks[(size_t)order[i].second]);
    }
    lv.avg_entries = lv.ntasks ? (double)sum / lv.ntasks : 1.0;
    lv.lanes = lv.avg_entries >= 160 ? 256 : lv.avg_entries >= 20 ? 64 : lv.avg_entries >= 8 ? 32 : 16;
    P.levels.push_back(lv);
  }
  P.nforward = NG;
  const auto T3 = tnow();
  P.weights_ok = write_weights(fills, cursor, nthreads, &P);
  P.out_off.resize((size_t)k);
  for (int j = 0; j < k; ++j) {
    const int s = piece_of[j];
    P.out_off[j] = pos(has_back[s] ? ybuf[s] : 1 - ybuf[s], j);
  }
  P.weights_read_per_apply = weights;
  if (timing) {
    std::fprintf(stderr, "[partinv2] %d levels -> %d groups, %d launches; M %.1f ms, merged products %.1f ms, layout %.1f ms, weights %.1f ms\n",
                 nlev, NG, NL, tms(T0, T1), tms(T1, T2), tms(T2, T3), tms(T3, tnow()));
    for (int g = 0; g < NG; ++g)
      if (groups[g].hi >= 0)
        std::fprintf(stderr, "[partinv2]   pair (%d,%d): fill %.2f MB, %.2f Gflop\n", groups[g].lo, groups[g].hi,
                     1e-6 * pair_bytes[groups[g].lo], 1e-9 * pair_flops[groups[g].lo]);
    for (size_t li = 0; li < P.levels.size(); ++li) {
      const SpLevel &lv = P.levels[li];
      long long w = 0, sg = 0;
      for (int t = lv.task0; t < lv.task0 + lv.ntasks; ++t) {
        const PTask &T = P.tasks[(size_t)t];
        for (int q = T.seg0; q < T.seg0 + T.nseg; ++q) w += (long long)P.segs[(size_t)q].len * T.nrows;
        sg += T.nseg;
      }
      std::fprintf(stderr, "[partinv2]   launch %2zu tiles %6d (256: %d, 128: %d, 64: %d, 32: %d) segments/tile %.2f weights %.2f MB\n",
                   li, lv.ntasks, lv.cls[0][0], lv.cls[0][1] - lv.cls[0][0], lv.cls[0][2] - lv.cls[0][1],
                   lv.cls[0][3] - lv.cls[0][2], lv.ntasks ? (double)sg / lv.ntasks : 0.0, 8e-6 * (double)w);
    }
  }
}

}  // namespace partinv
}  // namespace dcora
