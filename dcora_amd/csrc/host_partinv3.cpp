// Matrix-pipe schedule of the partitioned inverse (sparse_precond.h): the application of (Q + reg I)^-1
// (ref src/QuadraticProblem.cpp:70-84; the factor of ref src/Graph.cpp:1901-1917) with the sums over the gathered
// entries inside v_mfma_f64_4x4x4_4b_f64 (k_sp_mtile, sparse_precond.hip).
//
// The launches (rounds 1-3 applied L^-1 = prod L_s^-1 and L^-T one tree level per launch, 2 depth - 1 dependent launches,
// then regrouped them; the regrouped form is the only one left).  With y_s the value of piece s once every piece below
// it has been applied ("pre" value),
//     y_q = b_q + sum_{s below q} W_s[q] y_s        W_s = -B_s D_s^-1      (forward, leaves first)
//     x_s = M_s y_s + W_s^T x_{rows(s)}             M_s = D_s^-T D_s^-1    (backward, root first)
// so D_s^-1 y_s alone is never needed: the two triangular products of a piece are ONE symmetric product z_s = M_s y_s
// that nothing waits for until the backward sweep reaches s -- those M tiles are spread over the launches between the one
// that completes y_s and the one that consumes z_s, so that every launch streams about the same number of bytes.  Levels
// are merged in pairs (t, t + 1) where the fill of  W'_s[q] = W_s[q] + sum_p W_p[q] W_s[p]  (p on level t + 1, formed once
// at set-up) is cheaper than the launch it saves: the pair's forward launch updates the rows of level t + 1 AND all rows
// above at once, its backward launch computes x_p and x_s together.  A task is a tile of up to four consecutive output
// rows that gather from the same sources.  Storage and arithmetic (round 4; the tile schedule before it kept one copy of
// W per sweep laid out per tile, 4 r running sums per lane and a reduction over the lanes by DPP + LDS):
//   * every matrix is stored ONCE in 4 x 4 micro-blocks and read directly (forward: the rows of W_s a tile's rows need)
//     or transposed (backward: W_s^T; M_s, symmetric).  Stored weights of one agent of the 100k lattice: 162 MB instead
//     of 288 MB (137 MB without level pairs) -- inside the 256 MB Infinity Cache;
//   * a micro-block is one A operand of v_mfma_f64_4x4x4_4b_f64, which multiplies four independent 4 x 4 blocks per
//     instruction: a step of a tile consumes 16 gathered entries (512 bytes of weights) in ceil(r / 8) * 2
//     instructions and the sum over the entries happens in the matrix pipe;
//   * the device reads WAVE RECORDS (sparse_precond.h): the host cuts every tile's steps into contiguous shares of about
//     eight steps, one per wave, and packs eight waves per workgroup.
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
constexpr int kFillCols = 128;  // micro-block columns per fill record: 2048 weights

struct MSeg {      // one dense product of a tile: its rows x K entries of one stored matrix
  long long w;
  int ncb, ng, src, idx, kind, col0;
  int loc[4];
};
struct MTask {     // a tile: up to RT consecutive output rows that gather from the same sources
  int out, carry, nrows, steps;
  std::vector<MSeg> segs;
};

struct Source {        // what a piece contributes to the rows above it in one forward launch / gathers in a backward one
  int s = -1;          // piece
  const int *rows = nullptr;  // global permuted rows, ascending
  int m = 0;
  const double *V = nullptr;  // m x c row-major
};

// steps a wave should get: it requests the loads of up to eight steps together, so eight steps are one memory round trip
// (measured per application on a lattice agent: 4 -> 105 us, 8 -> 102, 16 -> 106; with the round-4 ordering 8 -> 90.4,
// 16 -> 89.5, sphere2500 26.5 / 28.0; the whole lattice as one problem, 3.6 GB per application: 8 / 16 / 32 -> 1.10 /
// 1.08 / 1.09 ms -- the number of waves is not what bounds the replay at either size)
constexpr int steps_per_wave() { return 8; }

}  // namespace

void layout_mpipe(const std::vector<Piece> &pc, const std::vector<const double *> &Mgiven,
                   const std::vector<int> &piece_of, int k, int nlev, int nthreads, bool timing, PartInvHost *out) {
  PartInvHost &P = *out;
  auto tnow = [] { return std::chrono::steady_clock::now(); };
  auto tms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
  const auto T0 = tnow();
  const int np = (int)pc.size();
  nthreads = std::max(1, nthreads);
  std::vector<std::vector<int>> by_level((size_t)nlev);
  for (int s = 0; s < np; ++s) by_level[pc[s].level].push_back(s);

  // ---- M_s = D_s^-T D_s^-1 of every piece the device did not deliver (symmetric, both triangles) ----
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

  // ---- which levels merge.  Levels below the top: 0 .. nlo - 1; the top level is a launch of its own.
  //      A GROUP is a run of consecutive levels [lo, hi] applied by ONE forward and ONE backward launch.  With
  //      apply_L(X) = X + sum_{q on level L} W_q X[rows of q], a piece s of the group on level l < hi contributes to
  //      everything above it through V_s = apply_hi(... apply_{l+1}(W_s)) = W_s + sum_{q in the group above s} V_q W_s[q]
  //      (V_q = W_q on level hi): the sources of the group's forward launch read their values from BEFORE the launch, the
  //      propagation inside the group is in the matrices.  Backward: x_s = M_s y_s + sum_{q in the group above s}
  //      V_s[q]^T (M_q y_q) + V_s[above]^T x_above.  Pairs (round 3) are the groups of two levels; round 4 merges on while a
  //      merge costs fewer streamed bytes than the two launches it saves are worth. ----
  const int nlo = nlev - 1;
  std::vector<std::vector<int>> merged_rows((size_t)np);  // rows of V_s for the pieces of CHOSEN groups (below their top level)
  // rows of V_s for s on level l of the group [lo, hi], given the rows of the pieces above it in the group (rows_of)
  auto union_rows = [&](int s, int hi, const std::vector<std::vector<int>> &rows_of, std::vector<int> *rows_out,
                        double *flops) {
    const Piece &ps = pc[s];
    std::vector<int> acc(ps.rows);
    int last = -1;
    for (size_t a = 0; a < ps.rows.size(); ++a) {
      const int q = piece_of[ps.rows[a]];
      if (q == last) continue;
      last = q;
      if (pc[q].level <= ps.level || pc[q].level > hi) continue;
      size_t b = a;
      while (b < ps.rows.size() && piece_of[ps.rows[b]] == q) ++b;
      const std::vector<int> &rq = rows_of[(size_t)q].empty() ? pc[q].rows : rows_of[(size_t)q];
      if (flops) *flops += 2.0 * (double)rq.size() * (double)(b - a) * ps.c;
      std::vector<int> tmp;
      tmp.reserve(acc.size() + rq.size());
      std::set_union(acc.begin(), acc.end(), rq.begin(), rq.end(), std::back_inserter(tmp));
      acc.swap(tmp);
    }
    rows_out->swap(acc);
  };
  // streamed bytes (stored once, read by both sweeps) and set-up flops a group [lo, hi] adds over its unmerged levels
  auto group_cost = [&](int lo, int hi, double *bytes, double *flops) {
    *bytes = 0;
    *flops = 0;
    std::vector<std::vector<int>> rows_of((size_t)np);
    for (int l = hi - 1; l >= lo; --l)
      for (int s : by_level[(size_t)l]) {
        std::vector<int> R;
        union_rows(s, hi, rows_of, &R, flops);
        *bytes += 16.0 * (double)(R.size() - pc[s].rows.size()) * pc[s].c;
        if (R.size() != pc[s].rows.size()) rows_of[(size_t)s].swap(R);
      }
  };
  struct Group {
    int lo, hi;  // levels lo .. hi (hi == lo: a single level)
    double bytes = 0, flops = 0;
  };
  std::vector<Group> groups;
  {
    // The grouping that minimises  launch pairs x 10 us + fill / 4 TB/s  over all splits of the levels into runs of at
    // most four (dynamic programme over the level boundaries); the fill of a group is stored once and streamed by both
    // sweeps; set-up products of a group stay below 12 Gflop.  Calibrated on a lattice agent (8 levels below the top):
    // {0} {1..4} {5..7}, 7 launches and 38 MB of fill: 89 us per application; {0,1} {2..7}, 5 launches and 109 MB: 96 us;
    // pairs, 9 launches and 36 MB: 91 us.
    const double us_per_launch_pair = 10.0, bytes_per_us = 4e6, max_flops = 12e9;
    const int kMaxRun = 4;
    std::vector<std::vector<Group>> cand((size_t)std::max(nlo, 1));  // cand[lo][len - 1]
    for (int lo = 0; lo < nlo; ++lo)
      for (int len = 1; len <= kMaxRun && lo + len <= nlo; ++len) {
        Group g{lo, lo + len - 1};
        if (len > 1) group_cost(g.lo, g.hi, &g.bytes, &g.flops);
        cand[(size_t)lo].push_back(g);
      }
    std::vector<double> best((size_t)nlo + 1, 1e300);
    std::vector<int> from((size_t)nlo + 1, -1);
    best[0] = 0;
    for (int i = 1; i <= nlo; ++i)
      for (int len = 1; len <= kMaxRun && len <= i; ++len) {
        const Group &g = cand[(size_t)(i - len)][(size_t)len - 1];
        if (g.flops > max_flops) continue;
        const double c = best[(size_t)(i - len)] + us_per_launch_pair + g.bytes / bytes_per_us;
        if (c < best[(size_t)i]) {
          best[(size_t)i] = c;
          from[(size_t)i] = len;
        }
      }
    for (int i = nlo; i > 0; i -= from[(size_t)i]) groups.push_back(cand[(size_t)(i - from[(size_t)i])][(size_t)from[(size_t)i] - 1]);
    std::reverse(groups.begin(), groups.end());
  }
  const int NG = (int)groups.size();
  const int NL = 2 * NG + 1;  // forward groups, the top, backward groups
  std::vector<int> group_of_level((size_t)std::max(nlev, 1), -1);
  for (int g = 0; g < NG; ++g)
    for (int l = groups[(size_t)g].lo; l <= groups[(size_t)g].hi; ++l) group_of_level[(size_t)l] = g;

  // ---- V_s of the pieces below the top level of their group, the highest level first ----
  std::vector<std::vector<double>> Vown((size_t)np);
  for (const Group &g : groups)
    for (int l = g.hi - 1; l >= g.lo; --l) {
      std::vector<int> todo(by_level[(size_t)l]);
      std::sort(todo.begin(), todo.end(), [&](int a, int b) {
        return (double)pc[a].c * pc[a].rows.size() > (double)pc[b].c * pc[b].rows.size();
      });
      parallel_for((int)todo.size(), nthreads, 1, [&](int ti) {
        const int s = todo[(size_t)ti];
        const Piece &ps = pc[s];
        const int c = ps.c, m = (int)ps.rows.size();
        std::vector<int> R;
        union_rows(s, g.hi, merged_rows, &R, nullptr);
        std::vector<double> &V = Vown[s];
        V.assign(R.size() * (size_t)c, 0.0);
        {
          size_t u = 0;
          for (int a = 0; a < m; ++a) {
            while (R[u] != ps.rows[a]) ++u;
            std::copy(ps.w() + (size_t)a * c, ps.w() + (size_t)a * c + c, &V[u * c]);
          }
        }
        for (int a = 0; a < m;) {
          const int q = piece_of[ps.rows[a]];
          int b = a;
          while (b < m && piece_of[ps.rows[b]] == q) ++b;
          if (pc[q].level > l && pc[q].level <= g.hi) {
            // V_q of a piece above s in the group (its own W on the group's top level)
            const Piece &pq = pc[q];
            const bool mq_merged = !merged_rows[(size_t)q].empty();
            const std::vector<int> &rq = mq_merged ? merged_rows[(size_t)q] : pq.rows;
            const double *vq = mq_merged ? Vown[(size_t)q].data() : pq.w();
            const int cq = pq.c, mq = (int)rq.size();
            size_t u = 0;
            for (int bb = 0; bb < mq; ++bb) {
              while (R[u] != rq[(size_t)bb]) ++u;
              double *dst = &V[u * c];
              const double *wq = vq + (size_t)bb * cq;
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
        merged_rows[(size_t)s].swap(R);  // (written by the one thread that owns s; pieces above s were finished a level ago)
      });
    }
  const auto T2 = tnow();

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

  // ---- storage: piece s owns W_s (or V_s) as an m x c matrix and M_s as a c x c one, both in micro-blocks ----
  std::vector<Fill> fills;
  long long cursor = 0;
  std::vector<long long> woff((size_t)np, -1), moff((size_t)np, -1);
  auto store_matrix = [&](const double *base, int m, int c) {
    const long long off = cursor;
    const int mb = pad4(m) / 4, ncb = pad4(c) / 4;
    for (int rb = 0; rb < mb; ++rb)
      for (int cb0 = 0; cb0 < ncb; cb0 += kFillCols) {
        const int cb1 = std::min(ncb, cb0 + kFillCols);
        Fill f;
        f.off = cursor;
        f.base = base;
        f.kind = 5;
        f.nrows = 4;
        f.len = (cb1 - cb0) * 4;
        f.c = c;
        f.a0 = rb * 4;
        f.m = m;
        f.loc[0] = cb0;
        f.loc[1] = cb1;
        f.loc[2] = f.loc[3] = 0;
        fills.push_back(f);
        cursor += (long long)(cb1 - cb0) * 16;
      }
    return off;
  };
  int ncb_max = 1;
  for (int s = 0; s < np; ++s) {
    const Source S = source_of(s);
    if (S.m > 0) woff[s] = store_matrix(S.V, S.m, pc[s].c);
    moff[s] = store_matrix(M[s], pc[s].c, pc[s].c);
    ncb_max = std::max(ncb_max, pad4(pc[s].c) / 4);
  }
  // the kernel reads (and zeroes) up to three groups past the end of a segment: three micro-block rows of the widest
  // matrix of zeros behind the last one
  for (long long left = 3LL * ncb_max; left > 0; left -= kFillCols) {
    Fill f;
    f.off = cursor;
    f.base = np ? M[0] : nullptr;
    f.kind = 5;
    f.nrows = 4;
    f.loc[0] = 0;
    f.loc[1] = (int)std::min<long long>(left, kFillCols);
    f.len = f.loc[1] * 4;
    f.c = 4;
    f.a0 = 0;
    f.m = 0;  // no row exists: zeros
    f.loc[2] = f.loc[3] = 0;
    fills.push_back(f);
    cursor += (long long)f.loc[1] * 16;
  }

  // ---- lay the launches out ----
  std::vector<int> cur((size_t)np, 0);    // buffer that holds the piece's current value
  std::vector<int> ybuf((size_t)np, -1);  // buffer of its pre value, once that is complete
  auto pos = [&](int b, int row) { return b * k + row; };
  double weights = 0;
  struct Launch {
    std::vector<MTask> tasks;
    double bytes = 0;
  };
  std::vector<Launch> L((size_t)NL);
  auto steps_of = [](const MSeg &S) { return (S.ng + 3) / 4; };
  auto push_task = [&](Launch &Ln, MTask T, const std::vector<MSeg> &sg) {
    T.segs = sg;
    T.steps = 0;
    for (const MSeg &S : sg) T.steps += steps_of(S);
    double b = 128.0;
    for (const MSeg &S : sg) b += 32.0 * S.ng * T.nrows;
    Ln.bytes += b;
    weights += (b - 128.0) / 8.0;
    Ln.tasks.push_back(std::move(T));
  };

  std::vector<int> m_first((size_t)np, 0), m_last((size_t)np, 0);

  // forward launches
  std::vector<std::vector<std::pair<int, int>>> hits((size_t)k);  // per row: (source index, local row of its V)
  std::vector<int> touched((size_t)np, -1);
  for (int g = 0; g < NG; ++g) {
    const Group &G = groups[g];
    Launch &Ln = L[(size_t)g];
    std::vector<Source> srcs;
    for (int l = G.lo; l <= G.hi; ++l)
      for (int s : by_level[(size_t)l]) srcs.push_back(source_of(s));
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
        // consecutive rows fed by the same pieces share a tile
        int nrows = 1;
        const auto &h0 = hits[p.c0 + a0];
        while (nrows < RT && a0 + nrows < p.c) {
          const auto &h1 = hits[p.c0 + a0 + nrows];
          bool same = h1.size() == h0.size();
          for (size_t u = 0; same && u < h0.size(); ++u) same = (h1[u].first == h0[u].first);
          if (!same) break;
          ++nrows;
        }
        MTask T;
        T.out = pos(1 - cur[q], p.c0 + a0);
        T.carry = pos(cur[q], p.c0 + a0);
        T.nrows = nrows;
        std::vector<MSeg> sg;
        for (size_t u = 0; u < h0.size(); ++u) {
          const Source &S = srcs[(size_t)h0[u].first];
          const Piece &s = pc[S.s];
          MSeg Sg;
          Sg.w = woff[S.s];
          Sg.ncb = pad4(s.c) / 4;
          Sg.ng = pad4(s.c) / 4;
          Sg.src = pos(cur[S.s], s.c0);  // a source of the launch is never one of its destinations' NEW values
          Sg.idx = 0;
          Sg.kind = 0;
          Sg.col0 = 0;
          for (int r_ = 0; r_ < RT; ++r_) Sg.loc[r_] = r_ < nrows ? hits[p.c0 + a0 + r_][u].second : -1;
          sg.push_back(Sg);
        }
        push_task(Ln, T, sg);
        a0 += nrows;
      }
    }
    for (int i : hit_rows) hits[i].clear();
    for (int q : affected) cur[q] ^= 1;
    for (int l = G.lo + 1; l <= G.hi; ++l)
      for (int s : by_level[(size_t)l]) {
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
  // backward launches
  std::vector<char> has_back((size_t)np, 0);
  for (int g = NG - 1; g >= 0; --g) {
    const Group &G = groups[g];
    Launch &Ln = L[(size_t)(2 * NG - g)];
    auto emit = [&](int s) {
      const Piece &p = pc[s];
      const Source S = source_of(s);
      if (S.m == 0) return;
      has_back[s] = 1;
      while (P.idxs.size() & 3) P.idxs.push_back(P.idxs.empty() ? 0 : P.idxs.back());
      const int idx0 = (int)P.idxs.size();
      for (int a = 0; a < S.m; ++a) {
        const int i = S.rows[a], q = piece_of[i];
        // a row of a piece above s in the SAME group: its z = M_q y_q (this launch computes x_q at the same time);
        // else the final x
        const bool upper = pc[q].level > p.level && pc[q].level <= G.hi;
        const int b = upper ? 1 - ybuf[q] : (pc[q].rows.empty() ? 1 - ybuf[q] : ybuf[q]);
        P.idxs.push_back(pos(b, i));
      }
      while (P.idxs.size() & 3) P.idxs.push_back(P.idxs.back());  // padded rows of the matrix are zero: any position
      for (int a0 = 0; a0 < p.c; a0 += RT) {
        const int nrows = std::min(RT, p.c - a0);
        MTask T;
        T.out = pos(ybuf[s], p.c0 + a0);
        T.carry = pos(1 - ybuf[s], p.c0 + a0);
        T.nrows = nrows;
        MSeg Wt;
        Wt.w = woff[s];
        Wt.ncb = pad4(p.c) / 4;
        Wt.ng = pad4(S.m) / 4;
        Wt.src = -1;
        Wt.idx = idx0;
        Wt.kind = 1;
        Wt.col0 = a0;
        for (int r_ = 0; r_ < RT; ++r_) Wt.loc[r_] = 0;
        push_task(Ln, T, std::vector<MSeg>(1, Wt));
      }
    };
    for (int l = G.hi; l >= G.lo; --l)
      for (int s : by_level[(size_t)l]) emit(s);
  }
  // ---- M tiles, spread: earliest deadline first, up to an even share of the bytes that are left ----
  {
    std::vector<std::vector<int>> avail_at((size_t)NL);
    for (int s = 0; s < np; ++s) avail_at[(size_t)m_first[s]].push_back(s);
    double total = 0;
    for (const Launch &Ln : L) total += Ln.bytes;
    for (int s = 0; s < np; ++s) total += 8.0 * (double)pc[s].c * pad4(pc[s].c);
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
        while (next_row[s] < p.c) {
          if (!must && Ln.bytes >= share) break;
          const int a0 = next_row[s], nrows = std::min(RT, p.c - a0);
          MTask T;
          T.out = pos(1 - ybuf[s], p.c0 + a0);
          T.carry = -1;
          T.nrows = nrows;
          MSeg Sg;
          Sg.w = moff[s];
          Sg.ncb = pad4(p.c) / 4;
          Sg.ng = pad4(p.c) / 4;
          Sg.src = pos(ybuf[s], p.c0);
          Sg.idx = 0;
          Sg.kind = 0;  // rows a0 .. of M: consecutive micro-blocks of one micro-block row, 512 contiguous bytes per step
          Sg.col0 = 0;
          for (int r_ = 0; r_ < RT; ++r_) Sg.loc[r_] = r_ < nrows ? a0 + r_ : -1;
          push_task(Ln, T, std::vector<MSeg>(1, Sg));
          next_row[s] += nrows;
        }
        if (next_row[s] >= p.c) {
          pending.erase(pending.begin() + (long)pi);
        } else {
          break;  // the share of this launch is used up (pieces with later deadlines wait as well)
        }
      }
      placed_so_far += Ln.bytes;
    }
  }
  // ---- launches -> wave records.  A tile of T steps gets n = ceil(T / steps_per_wave) waves (at most a workgroup's,
  //      at least enough for every wave to touch four segments at most), each a contiguous share of the tile's steps; tiles
  //      are packed into workgroups of kMtWaves waves, most waves first; a tile with more segments than its waves can
  //      hold continues in chained records ----
  const int spw = steps_per_wave();
  long long n_records = 0, n_chained = 0;
  std::vector<int> longest_chain((size_t)NL, 0), most_segs((size_t)NL, 0);
  for (int l = 0; l < NL; ++l) {
    Launch &Ln = L[(size_t)l];
    struct Plan {
      int tile, n;
    };
    std::vector<Plan> plan(Ln.tasks.size());
    for (size_t i = 0; i < Ln.tasks.size(); ++i) {
      const MTask &T = Ln.tasks[i];
      int n = (T.steps + spw - 1) / spw;
      // four runs a wave (a second, chained record) before a tile of many short runs takes another wave: measured 87.6 us
      // on a lattice agent against 89.3 with two and 88.4 with eight
      n = std::max(n, std::min(kMtWaves, ((int)T.segs.size() + 3) / 4));
      n = std::max(1, std::min(n, std::min(kMtWaves, std::max(1, T.steps))));
      plan[i] = {(int)i, n};
    }
    std::stable_sort(plan.begin(), plan.end(), [](const Plan &a, const Plan &b) { return a.n > b.n; });
    // first fit, decreasing: open workgroups by free waves
    std::vector<std::vector<int>> wgs;  // tiles of every workgroup
    std::vector<int> freew;
    std::vector<std::vector<int>> open_by_free((size_t)kMtWaves + 1);
    for (const Plan &pl : plan) {
      int wg = -1;
      for (int f = pl.n; f <= kMtWaves && wg < 0; ++f)
        if (!open_by_free[(size_t)f].empty()) {
          wg = open_by_free[(size_t)f].back();
          open_by_free[(size_t)f].pop_back();
        }
      if (wg < 0) {
        wg = (int)wgs.size();
        wgs.emplace_back();
        freew.push_back(kMtWaves);
      }
      wgs[(size_t)wg].push_back(pl.tile);
      freew[(size_t)wg] -= pl.n;
      if (freew[(size_t)wg] > 0) open_by_free[(size_t)freew[(size_t)wg]].push_back(wg);
    }
    std::vector<int> nwaves_of(Ln.tasks.size(), 1);
    for (const Plan &pl : plan) nwaves_of[(size_t)pl.tile] = pl.n;
    SpLevel lv;
    lv.task0 = (int)P.mwaves.size();
    lv.ntasks = (int)wgs.size();
    const size_t base = P.mwaves.size();
    MWave idle;
    std::memset(&idle, 0, sizeof idle);
    idle.carry = -1;
    idle.next = -1;
    idle.red_n = 1;
    P.mwaves.resize(base + wgs.size() * (size_t)kMtWaves, idle);
    std::vector<MWave> chained;  // continuation records of this launch: appended behind its workgroups
    long long sum = 0;
    for (size_t w = 0; w < wgs.size(); ++w) {
      int wave = 0;
      for (int ti : wgs[w]) {
        const MTask &T = Ln.tasks[(size_t)ti];
        const int n = nwaves_of[(size_t)ti];
        sum += T.steps;
        // the runs of the tile, in order: (segment, first step, steps); wave v takes steps [v T / n, (v + 1) T / n)
        int sg = 0, s_in = 0;  // cursor: segment and step inside it
        for (int v = 0; v < n; ++v) {
          const int t0 = (int)((long long)v * T.steps / n), t1 = (int)((long long)(v + 1) * T.steps / n);
          // the runs of this wave, two per record: the first record sits in the workgroup, the others are chained
          std::vector<MWave> recs;
          int left = t1 - t0, filled = 2;
          while (left > 0) {
            while (s_in >= steps_of(T.segs[(size_t)sg])) {
              ++sg;
              s_in = 0;
            }
            const MSeg &S = T.segs[(size_t)sg];
            const int take = std::min(left, steps_of(S) - s_in);
            if (filled == 2) {
              recs.push_back(idle);
              filled = 0;
            }
            MSub &U = filled == 0 ? recs.back().a : recs.back().b;
            U.w = S.w;
            U.ncb = S.ncb;
            U.ng = S.ng;
            U.src = S.kind == 0 ? S.src : S.idx;
            U.s0 = s_in;
            U.n = take;
            U.pad = 0;
            for (int q = 0; q < 4; ++q) U.loc[q] = S.kind == 0 ? S.loc[q] : (q == 0 ? S.col0 : 0);
            ++filled;
            s_in += take;
            left -= take;
          }
          if (recs.empty()) recs.push_back(idle);  // a tile without sources: the copy of its old values
          longest_chain[(size_t)l] = std::max(longest_chain[(size_t)l], (int)recs.size());
          most_segs[(size_t)l] = std::max(most_segs[(size_t)l], (int)T.segs.size());
          for (size_t q = 0; q < recs.size(); ++q) {
            MWave &R = recs[q];
            R.out = T.out;
            R.carry = q == 0 ? T.carry : -1;
            R.nrows = T.nrows;
            R.kind = T.segs.empty() ? 0 : T.segs[0].kind;
            R.red_first = wave;
            R.red_n = n;
            R.next = -1;
            if (q + 1 < recs.size())  // relative to the launch's first record
              R.next = (int)(wgs.size() * (size_t)kMtWaves + chained.size() + (q == 0 ? 0 : 1));
            if (q == 0) {
              P.mwaves[base + w * kMtWaves + (size_t)(wave + v)] = R;
            } else {
              chained.push_back(R);
              ++n_chained;
            }
          }
        }
        wave += n;
      }
    }
    for (size_t w = 0; w < wgs.size(); ++w) {
      int solo = 1;
      for (int q = 0; q < kMtWaves; ++q) solo &= P.mwaves[base + w * kMtWaves + (size_t)q].red_n <= 1;
      for (int q = 0; q < kMtWaves; ++q) P.mwaves[base + w * kMtWaves + (size_t)q].solo = solo;
    }
    P.mwaves.insert(P.mwaves.end(), chained.begin(), chained.end());
    n_records += (long long)wgs.size() * kMtWaves + (long long)chained.size();
    lv.avg_entries = Ln.tasks.empty() ? 1.0 : 16.0 * (double)sum / (double)Ln.tasks.size();
    lv.ntiles = (int)Ln.tasks.size();
    P.levels.push_back(lv);
  }
  for (int q = 0; q < 16; ++q) P.idxs.push_back(0);  // the kernel reads up to three groups past the end of a list
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
    std::fprintf(stderr, "[partinv3] %d levels -> %d groups, %d launches, %.1f MB stored, %.1f MB streamed per application; M %.1f ms, merged products %.1f ms, layout %.1f ms, weights %.1f ms\n",
                 nlev, NG, NL, 8e-6 * (double)cursor, 8e-6 * weights, tms(T0, T1), tms(T1, T2), tms(T2, T3),
                 tms(T3, tnow()));
    for (int g = 0; g < NG; ++g)
      if (groups[g].hi > groups[g].lo)
        std::fprintf(stderr, "[partinv3]   levels %d..%d in one launch: fill %.2f MB, %.2f Gflop\n", groups[g].lo,
                     groups[g].hi, 1e-6 * groups[g].bytes, 1e-9 * groups[g].flops);
    std::fprintf(stderr, "[partinv3]   %lld wave records (%lld chained)\n", n_records, n_chained);
    for (size_t li = 0; li < P.levels.size(); ++li) {
      const SpLevel &lv = P.levels[li];
      int busy = 0;
      for (int q = 0; q < lv.ntasks * kMtWaves; ++q) busy += P.mwaves[(size_t)lv.task0 + q].nrows > 0;
      std::fprintf(stderr, "[partinv3]   launch %2zu tiles %6d workgroups %5d waves %6d steps/tile %.1f, at most %d segments a tile, %d records a wave\n",
                   li, lv.ntiles, lv.ntasks, busy, lv.avg_entries / 16.0, most_segs[li], longest_chain[li]);
    }
  }
}

}  // namespace partinv
}  // namespace dcora
