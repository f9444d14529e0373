// Host-side construction of the partitioned inverse of a nested-dissection Cholesky factor (sparse_precond.h).
// Setup-time code: runs once per Q; the per-iteration path only replays the schedule on the device.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <thread>

#include "host_partinv_int.h"
#include "sparse_precond.h"

namespace dcora {

using namespace partinv;

namespace {

constexpr int kBigPiece = 384;  // pieces this wide are inverted by all threads together (the merged top of the tree)

int pick_lanes(double avg_entries_per_tile, int ntasks) {
  (void)ntasks;
  if (avg_entries_per_tile >= 160) return 256;
  if (avg_entries_per_tile >= 20) return 64;
  if (avg_entries_per_tile >= 8) return 32;
  return 16;
}

}  // namespace

void piecewise_from_chol(const SparseChol &chol, PiecewiseFactor *out) {
  PiecewiseFactor &F = *out;
  F = PiecewiseFactor();
  F.n = chol.n();
  F.nhub = chol.nhub();
  F.nnzL = chol.nnzL();
  F.perm = chol.perm();
  F.iperm = chol.iperm();
  const std::vector<int> &Lp = chol.Lp(), &Li = chol.Li();
  const std::vector<double> &Lx = chol.Lx();
  const std::vector<int> &cuts = chol.pieces();
  const int np = (int)cuts.size() - 1;
  F.pieces.assign((size_t)np, PieceFactor());
  std::vector<int> mark((size_t)F.n, -1), where((size_t)F.n, -1);
  for (int s = 0; s < np; ++s) {
    PieceFactor &p = F.pieces[s];
    p.c0 = cuts[s];
    p.c = cuts[s + 1] - cuts[s];
    const int hi = p.c0 + p.c, c = p.c;
    for (int j = p.c0; j < hi; ++j)
      for (int q = Lp[j] + 1; q < Lp[j + 1]; ++q) {
        const int i = Li[q];
        if (i >= hi && mark[i] != s) {
          mark[i] = s;
          p.rows.push_back(i);
        }
      }
    std::sort(p.rows.begin(), p.rows.end());
    const int m = (int)p.rows.size();
    for (int a2 = 0; a2 < m; ++a2) where[p.rows[a2]] = a2;
    p.panel.assign((size_t)(c + m) * c, 0.0);
    for (int j = 0; j < c; ++j)
      for (int q = Lp[p.c0 + j]; q < Lp[p.c0 + j + 1]; ++q) {
        const int i = Li[q];
        const int row = i < hi ? i - p.c0 : c + where[i];
        p.panel[(size_t)row * c + j] = Lx[q];
      }
  }
}

namespace {
// u <- A11^-1 u with the pieces [0, np) of the factor (the leading k x k block of L): block forward and backward
// substitution on the panels (hubs only: h right-hand sides, set-up time)
void piecewise_solve(const PiecewiseFactor &F, int np, int k, std::vector<double> &u) {
  for (int s = 0; s < np; ++s) {
    const PieceFactor &p = F.pieces[s];
    const int c = p.c, m = (int)p.rows.size();
    double *us = &u[(size_t)p.c0];
    for (int i = 0; i < c; ++i) {
      const double *di = p.pan() + (size_t)i * c;
      double sum = us[i];
      for (int l = 0; l < i; ++l) sum -= di[l] * us[l];
      us[i] = sum / di[i];
    }
    for (int a2 = 0; a2 < m; ++a2) {
      const int row = p.rows[a2];
      if (row >= k) continue;
      const double *ba = p.pan() + (size_t)(c + a2) * c;
      double sum = 0;
      for (int j = 0; j < c; ++j) sum += ba[j] * us[j];
      u[(size_t)row] -= sum;
    }
  }
  for (int s = np - 1; s >= 0; --s) {
    const PieceFactor &p = F.pieces[s];
    const int c = p.c, m = (int)p.rows.size();
    double *us = &u[(size_t)p.c0];
    for (int a2 = 0; a2 < m; ++a2) {
      const int row = p.rows[a2];
      if (row >= k) continue;
      const double ur = u[(size_t)row];
      if (ur == 0.0) continue;
      const double *ba = p.pan() + (size_t)(c + a2) * c;
      for (int j = 0; j < c; ++j) us[j] -= ba[j] * ur;
    }
    for (int i = c - 1; i >= 0; --i) {
      const double xi = us[i] / p.pan()[(size_t)i * c + i];
      us[i] = xi;
      const double *di = p.pan() + (size_t)i * c;
      for (int l = 0; l < i; ++l) us[l] -= di[l] * xi;
    }
  }
}
}  // namespace

bool build_partitioned_inverse(const HostCsr &A, int block, int nthreads, PartInvHost *out) {
  const bool timing = std::getenv("DCORA_INIT_TIMING") != nullptr;
  const auto T0 = std::chrono::steady_clock::now();
  SparseChol chol;
  if (!chol.factor(A, block, nd_top_default())) return false;
  PiecewiseFactor F;
  piecewise_from_chol(chol, &F);
  if (timing)
    std::fprintf(stderr, "[partinv] host factorisation + panels %.1f ms\n",
                 std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - T0).count());
  return build_partitioned_inverse_from(A, F, nthreads, out);
}

bool build_partitioned_inverse_from(const HostCsr &A, const PiecewiseFactor &F, int nthreads, PartInvHost *out) {
  const bool timing = std::getenv("DCORA_INIT_TIMING") != nullptr;
  auto tnow = [] { return std::chrono::steady_clock::now(); };
  auto tms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
  const auto T1 = tnow();
  const int kfull = A.n;
  // hubs are the trailing columns of the order and the last piece: they stay out of the replay (Schur complement,
  // see PartInvHub); the leading k x k block of L is the factor of A11
  const int h = (F.nhub > 0 && F.nhub <= 64 && F.nhub < kfull) ? F.nhub : 0;
  const int k = kfull - h;
  PartInvHost &P = *out;
  WeightSink *const sink = P.sink;  // the caller's choice survives the reset
  P = PartInvHost();
  P.sink = sink;
  P.mirrors = F.mirrors;
  P.sources_on_device_only = F.m_on_device_only;
  P.k = k;
  P.kfull = kfull;
  P.perm.assign(F.perm.begin(), F.perm.begin() + k);
  P.nnzL = F.nnzL;
  const int np = (int)F.pieces.size() - (h > 0 ? 1 : 0);
  P.npieces = np;
  std::vector<Piece> pc((size_t)np);
  std::vector<int> piece_of((size_t)k);
  for (int s = 0; s < np; ++s) {
    pc[s].c0 = F.pieces[s].c0;
    pc[s].c = F.pieces[s].c;
    for (int j = pc[s].c0; j < pc[s].c0 + pc[s].c; ++j) piece_of[j] = s;
  }
  // rows below each piece and dependency levels (longest path from the leaves).  Rows of the hubs are left to the
  // Schur complement; rows that a closed piece structure carries as explicit zeros (device_chol.h) are dropped.
  for (int s = 0; s < np; ++s) {
    const PieceFactor &f = F.pieces[s];
    const int c = f.c;
    for (int a2 = 0; a2 < (int)f.rows.size(); ++a2) {
      const int i = f.rows[a2];
      if (i >= k) continue;
      const double *ba = f.pan() + (size_t)(c + a2) * c;
      bool any = false;
      for (int j = 0; j < c && !any; ++j) any = ba[j] != 0.0;
      if (!any) continue;
      pc[s].rows.push_back(i);
      pc[s].src.push_back(a2);
    }
    for (int i : pc[s].rows) {
      Piece &q = pc[piece_of[i]];
      q.level = std::max(q.level, pc[s].level + 1);
    }
  }
  int nlev = 0;
  for (const Piece &p : pc) nlev = std::max(nlev, p.level + 1);
  if (const char *dump = std::getenv("DCORA_PARTINV_DUMP")) {  // structure only: c0 c level m rows...
    if (FILE *fp = std::fopen(dump, "w")) {
      for (const Piece &p : pc) {
        std::fprintf(fp, "%d %d %d %d", p.c0, p.c, p.level, (int)p.rows.size());
        for (int i : p.rows) std::fprintf(fp, " %d", i);
        std::fprintf(fp, "\n");
      }
      std::fclose(fp);
    }
  }
  // ---- numeric part: D^-1 and W = -B D^-1 of every piece.  Wide pieces first, one at a time with all threads on the
  //      columns of the inverse (column k of D^-1 is an independent forward substitution); then the many small
  //      pieces in parallel, most expensive first ----
  nthreads = std::max(1, nthreads);
  auto extract = [&](int s, std::vector<double> &D, std::vector<double> &B) {
    const PieceFactor &f = F.pieces[s];
    const Piece &p = pc[s];
    const int c = p.c, m = (int)p.rows.size();
    const double *fp = f.pan();
    D.assign(fp, fp + (size_t)c * c);
    B.resize((size_t)m * c);
    for (int a2 = 0; a2 < m; ++a2)
      std::copy(fp + (size_t)(c + p.src[a2]) * c, fp + (size_t)(c + p.src[a2]) * c + c, &B[(size_t)a2 * c]);
  };
  auto w_row = [](const double *ba, const std::vector<double> &Dinv, int c, double *wa) {
    for (int l = 0; l < c; ++l) {
      const double b = ba[l];
      if (b == 0.0) continue;
      const double *dl = &Dinv[(size_t)l * c];
      for (int j = 0; j <= l; ++j) wa[j] -= b * dl[j];
    }
  };
  {
    std::vector<double> D, B;
    for (int s = 0; s < np; ++s) {
      Piece &p = pc[s];
      if (F.pieces[s].inverted) {  // the device delivered L11^-1 over W: nothing to compute
        const PieceFactor &f = F.pieces[s];
        const int c = p.c, m = (int)p.rows.size();
        const double *fp = f.pan();
        p.Dinv_view = fp;
        bool all_rows = true;
        for (int a2 = 0; a2 < m; ++a2) all_rows = all_rows && p.src[a2] == a2;
        if (all_rows) {
          p.W_view = fp + (size_t)c * c;
        } else {
          p.W.resize((size_t)m * c);
          for (int a2 = 0; a2 < m; ++a2)
            std::copy(fp + (size_t)(c + p.src[a2]) * c, fp + (size_t)(c + p.src[a2]) * c + c, &p.W[(size_t)a2 * c]);
        }
        continue;
      }
      if (p.c < kBigPiece || nthreads < 2) continue;
      const int c = p.c, m = (int)p.rows.size();
      extract(s, D, B);
      std::vector<double> &Dinv = p.Dinv;
      Dinv.assign((size_t)c * c, 0.0);
      // column kc of D^-1 by forward substitution in axpy form over the rows of D^T (contiguous, vectorisable:
      // a dot-product form would be a serial chain of dependent additions)
      std::vector<double> DT((size_t)c * c);
      parallel_for(c, nthreads, 16, [&](int i) {
        for (int l = 0; l <= i; ++l) DT[(size_t)l * c + i] = D[(size_t)i * c + l];
      });
      parallel_for(c, nthreads, 4, [&](int kc) {
        std::vector<double> x((size_t)c, 0.0);
        x[kc] = 1.0;
        for (int l = kc; l < c; ++l) {
          const double *dl = &DT[(size_t)l * c];
          const double xl = x[l] / dl[l];
          x[l] = xl;
          if (xl == 0.0) continue;
          for (int i = l + 1; i < c; ++i) x[i] -= dl[i] * xl;
        }
        for (int i = kc; i < c; ++i) Dinv[(size_t)i * c + kc] = x[i];
      });
      p.W.assign((size_t)m * c, 0.0);
      parallel_for(m, nthreads, 8, [&](int a) { w_row(&B[(size_t)a * c], Dinv, c, &p.W[(size_t)a * c]); });
    }
  }
  {
    std::vector<int> order;
    for (int s = 0; s < np; ++s)
      if (!pc[s].dinv()) order.push_back(s);
    std::sort(order.begin(), order.end(), [&](int a, int b) {
      const double wa = (double)pc[a].c * pc[a].c * (pc[a].c + 3.0 * pc[a].rows.size());
      const double wb = (double)pc[b].c * pc[b].c * (pc[b].c + 3.0 * pc[b].rows.size());
      return wa > wb;
    });
    const int nord = (int)order.size();
    std::atomic<int> next(0);
    auto work = [&]() {
      std::vector<double> D, B;
      for (;;) {
        const int t = next.fetch_add(1);
        if (t >= nord) break;
        Piece &p = pc[order[t]];
        const int c = p.c, m = (int)p.rows.size();
        extract(order[t], D, B);
        // Dinv = D^-1 (lower triangular), built row by row: row_i = (e_i - sum_{l<i} D_il row_l) / D_ii
        std::vector<double> &Dinv = p.Dinv;
        Dinv.assign((size_t)c * c, 0.0);
        for (int i = 0; i < c; ++i) {
          const double *di = &D[(size_t)i * c];
          double *ri = &Dinv[(size_t)i * c];
          for (int l = 0; l < i; ++l) {
            const double coef = di[l];
            if (coef == 0.0) continue;
            const double *rl = &Dinv[(size_t)l * c];
            for (int j = 0; j <= l; ++j) ri[j] += coef * rl[j];
          }
          const double inv = 1.0 / di[i];
          for (int j = 0; j < i; ++j) ri[j] = -ri[j] * inv;
          ri[i] = inv;
        }
        p.W.assign((size_t)m * c, 0.0);
        for (int a = 0; a < m; ++a) w_row(&B[(size_t)a * c], Dinv, c, &p.W[(size_t)a * c]);
      }
    };
    const int nt = std::max(1, std::min(nthreads, nord));
    std::vector<std::thread> th;
    for (int t = 1; t < nt; ++t) th.emplace_back(work);
    work();
    for (auto &t : th) t.join();
  }
  const auto T2 = tnow();
  // ---- schedule: merged levels (host_partinv2.cpp) unless DCORA_SP_SCHEDULE=v1 asks for one launch per tree level ----
  static const int schedule = [] {  // 3: panels (default), 2: merged 4-row tiles, 1: one launch per tree level
    const char *e = std::getenv("DCORA_SP_SCHEDULE");
    return e && std::strcmp(e, "v1") == 0 ? 1 : e && std::strcmp(e, "v2") == 0 ? 2 : 3;
  }();
  const bool merged = schedule != 1;
  if (merged) {
    std::vector<const double *> Mgiven((size_t)np, nullptr);
    for (int s = 0; s < np; ++s)
      if (F.pieces[s].mtop()) Mgiven[s] = F.pieces[s].mtop();
    if (schedule == 3)
      layout_mpipe(pc, Mgiven, piece_of, k, nlev, nthreads, timing, &P);
    else
      layout_merged(pc, Mgiven, piece_of, k, nlev, nthreads, timing, &P);
    if (timing)
      std::fprintf(stderr, "[partinv] k %d pieces %d levels %d: piece inverses %.1f, %s schedule %.1f ms\n", k, np, nlev,
                   tms(T1, T2), schedule == 3 ? "matrix-pipe" : "merged", tms(T2, tnow()));
  } else {
  // ---- schedule.  Which buffer holds a piece's current value is static; start: everything in buffer 0.
  // A task is a tile of up to kSpTile consecutive output rows that gather from the same sources; the weights of
  // a segment are stored entry-major over the tile's rows:  [entry j][row q]. ----
  constexpr int RT = kSpTile;
  std::vector<int> bit((size_t)np, 0);
  std::vector<std::vector<int>> by_level((size_t)nlev);
  for (int s = 0; s < np; ++s) by_level[pc[s].level].push_back(s);
  auto pos = [&](int s_bit, int row) { return s_bit * k + row; };
  std::vector<int> touched_stamp((size_t)np, -1);
  std::vector<std::vector<std::pair<int, int>>> hits((size_t)k);  // per row: (piece, local row) of this level
  double weights = 0;
  // The weights of a segment ([entry j][tile row q], j < len) are only RESERVED while the schedule is laid out; the
  // (by far larger) job of writing them -- 0.8 G doubles for the whole 100k lattice -- is done afterwards by all threads
  std::vector<Fill> fills;
  std::vector<std::vector<double>> top_blocks;  // D^-T D^-1 of the top pieces: alive until the fill
  long long cursor = 0;
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
    for (int q = 0; q < kSpTile; ++q) f.loc[q] = loc ? loc[q] : 0;
    fills.push_back(f);
    cursor += (long long)len * nrows;
    weights += (double)len * nrows;
    return f.off;
  };
  // forward: y <- L_t^-1 y, leaves first.  The top level is skipped here: its pieces feed nobody (no rows below),
  // so their forward and backward steps are adjacent and are applied together as the symmetric D^-T D^-1 below
  for (int t = 0; t + 1 < nlev; ++t) {
    SpLevel lv;
    lv.task0 = (int)P.tasks.size();
    std::vector<int> affected;
    for (int s : by_level[t]) {
      touched_stamp[s] = t;
      affected.push_back(s);
    }
    std::vector<int> hit_rows;
    for (int s : by_level[t]) {
      const std::vector<int> &rows = pc[s].rows;
      for (int a = 0; a < (int)rows.size(); ++a) {
        const int i = rows[a], q = piece_of[i];
        if (touched_stamp[q] != t) {
          touched_stamp[q] = t;
          affected.push_back(q);
        }
        if (hits[i].empty()) hit_rows.push_back(i);
        hits[i].emplace_back(s, a);
      }
    }
    long long seg_len_sum = 0, seg_cnt = 0;
    for (int q : affected) {
      const Piece &p = pc[q];
      const bool own = (p.level == t);
      for (int a0 = 0; a0 < p.c;) {
        int nrows = 1;
        if (own) {
          nrows = std::min(RT, p.c - a0);
        } else {
          // consecutive rows fed by the same pieces share a tile
          const auto &h0 = hits[p.c0 + a0];
          while (nrows < RT && a0 + nrows < p.c) {
            const auto &h1 = hits[p.c0 + a0 + nrows];
            bool same = h1.size() == h0.size();
            for (size_t u = 0; same && u < h0.size(); ++u) same = (h1[u].first == h0[u].first);
            if (!same) break;
            ++nrows;
          }
        }
        PTask T;
        T.out = pos(1 - bit[q], p.c0 + a0);
        T.nrows = nrows;
        T.seg0 = (int)P.segs.size();
        if (own) {
          T.carry = -1;
          PSeg S;
          S.len = pad2(a0 + nrows);
          S.src = pos(bit[q], p.c0);
          S.idx = 0;
          S.pad = 0;
          const int c = p.c;
          S.w = reserve(0, nrows, S.len, p.dinv(), c, a0, 0, nullptr);
          P.segs.push_back(S);
        } else {
          T.carry = pos(bit[q], p.c0 + a0);
          const auto &h0 = hits[p.c0 + a0];
          for (size_t u = 0; u < h0.size(); ++u) {
            const int sid = h0[u].first;
            const Piece &s = pc[sid];
            int loc[RT] = {0};
            for (int r_ = 0; r_ < nrows; ++r_) loc[r_] = hits[p.c0 + a0 + r_][u].second;
            PSeg S;
            S.len = pad2(s.c);
            S.src = pos(bit[sid], s.c0);
            S.idx = 0;
            S.pad = 0;
            const int c = s.c;
            S.w = reserve(1, nrows, S.len, s.w(), c, 0, 0, loc);
            P.segs.push_back(S);
          }
        }
        T.nseg = (int)P.segs.size() - T.seg0;
        for (int q2 = T.seg0; q2 < T.seg0 + T.nseg; ++q2) {
          seg_len_sum += P.segs[q2].len;
          ++seg_cnt;
        }
        inline_first_segment(T, P.segs);
        P.tasks.push_back(T);
        a0 += nrows;
      }
    }
    for (int i : hit_rows) hits[i].clear();
    for (int q : affected) bit[q] ^= 1;
    lv.ntasks = (int)P.tasks.size() - lv.task0;
    lv.avg_entries = lv.ntasks ? (double)seg_len_sum / lv.ntasks : 1.0;
    lv.lanes = pick_lanes(lv.avg_entries, lv.ntasks);
    P.levels.push_back(lv);
  }
  P.nforward = nlev - 1;
  // backward: x <- L_t^-T x, root first.  Only the piece's own rows change; they gather from their own old
  // values and from the (final) values of the rows below.
  for (int t = nlev - 1; t >= 0; --t) {
    SpLevel lv;
    lv.task0 = (int)P.tasks.size();
    long long seg_len_sum = 0, seg_cnt = 0;
    for (int s : by_level[t]) {
      const Piece &p = pc[s];
      const int m = (int)p.rows.size(), c = p.c;
      const int idx0 = (int)P.idxs.size();
      for (int i : p.rows) P.idxs.push_back(pos(bit[piece_of[i]], i));
      if (m & 1) P.idxs.push_back(P.idxs.back());  // padded pair: weight 0, any valid position
      // top level only: D^-T D^-1 (c x c, symmetric), kept until the weights are written
      const double *Mtop = nullptr;
      if (t == nlev - 1) {
        if (F.pieces[s].mtop()) {  // formed on the device
          Mtop = F.pieces[s].mtop();
        } else {
        top_blocks.emplace_back((size_t)c * c, 0.0);
        std::vector<double> &M = top_blocks.back();
        Mtop = M.data();
        if (c >= kBigPiece && nthreads > 1) {
          // row a of the lower triangle by one thread: M(a, j) = sum_{i >= a} Dinv(i, a) Dinv(i, j), j <= a
          parallel_for(c, nthreads, 4, [&](int a) {
            double *ma = &M[(size_t)a * c];
            for (int i = a; i < c; ++i) {
              const double *di = p.dinv() + (size_t)i * c;
              const double v = di[a];
              if (v == 0.0) continue;
              for (int j = 0; j <= a; ++j) ma[j] += v * di[j];
            }
          });
          for (int a = 0; a < c; ++a)
            for (int j = a + 1; j < c; ++j) M[(size_t)a * c + j] = M[(size_t)j * c + a];
        } else {
          for (int i = 0; i < c; ++i) {
            const double *di = p.dinv() + (size_t)i * c;
            for (int a = 0; a <= i; ++a) {
              const double v = di[a];
              if (v == 0.0) continue;
              double *ma = &M[(size_t)a * c];
              for (int j = 0; j <= i; ++j) ma[j] += v * di[j];
            }
          }
        }
        }
      }
      for (int a0 = 0; a0 < c; a0 += RT) {
        const int nrows = std::min(RT, c - a0);
        PTask T;
        T.out = pos(1 - bit[s], p.c0 + a0);
        T.carry = -1;
        T.nrows = nrows;
        T.seg0 = (int)P.segs.size();
        PSeg S;
        S.idx = 0;
        S.pad = 0;
        if (t == nlev - 1) {
          S.len = pad2(c);
          S.src = pos(bit[s], p.c0);
          S.w = reserve(2, nrows, S.len, Mtop, c, a0, 0, nullptr);
        } else {
          S.len = pad2(c - a0);
          S.src = pos(bit[s], p.c0 + a0);
          // (D^-T)(a0 + q, a0 + j) = Dinv(a0 + j, a0 + q), zero below the diagonal of the transpose
          S.w = reserve(3, nrows, S.len, p.dinv(), c, a0, 0, nullptr);
        }
        P.segs.push_back(S);
        seg_len_sum += S.len;
        ++seg_cnt;
        if (m > 0) {
          PSeg Wt;
          Wt.len = pad2(m);
          Wt.src = -1;
          Wt.idx = idx0;
          Wt.pad = 0;
          Wt.w = reserve(4, nrows, Wt.len, p.w(), c, a0, m, nullptr);
          P.segs.push_back(Wt);
          seg_len_sum += Wt.len;
          ++seg_cnt;
        }
        T.nseg = (int)P.segs.size() - T.seg0;
        inline_first_segment(T, P.segs);
        P.tasks.push_back(T);
      }
    }
    for (int s : by_level[t]) bit[s] ^= 1;
    lv.ntasks = (int)P.tasks.size() - lv.task0;
    lv.avg_entries = lv.ntasks ? (double)seg_len_sum / lv.ntasks : 1.0;
    lv.lanes = pick_lanes(lv.avg_entries, lv.ntasks);
    P.levels.push_back(lv);
  }
  // ---- write the weights ----
  P.weights_ok = write_weights(fills, cursor, nthreads, &P);
  top_blocks.clear();
  P.out_off.resize((size_t)k);
  for (int j = 0; j < k; ++j) P.out_off[j] = pos(bit[piece_of[j]], j);
  P.weights_read_per_apply = weights;
  if (timing) {
    int cmax = 0;
    for (const Piece &p : pc) cmax = std::max(cmax, p.c);
    std::fprintf(stderr, "[partinv] k %d pieces %d levels %d widest piece %d: piece inverses %.1f, schedule %.1f ms\n",
                 k, np, nlev, cmax, tms(T1, T2), tms(T2, tnow()));
    for (size_t li = 0; li < P.levels.size(); ++li) {
      const SpLevel &lv = P.levels[li];
      long long w = 0, sg = 0;
      for (int t = lv.task0; t < lv.task0 + lv.ntasks; ++t) {
        const PTask &T = P.tasks[(size_t)t];
        for (int q = T.seg0; q < T.seg0 + T.nseg; ++q) w += (long long)P.segs[(size_t)q].len * T.nrows;
        sg += T.nseg;
      }
      std::fprintf(stderr, "[partinv]   level %2zu%s tiles %6d lanes %3d waves %6lld segments/tile %.2f weights %.2f MB\n", li,
                   (int)li < P.nforward ? "f" : "b", lv.ntasks, lv.lanes, (long long)lv.ntasks * lv.lanes / 64,
                   lv.ntasks ? (double)sg / lv.ntasks : 0.0, 8e-6 * (double)w);
    }
  }
  }  // schedule v1
  // ---- hubs: U = A11^-1 a with the leading block of L, Sc = alpha - a^T U ----
  if (h > 0) {
    PartInvHub &H = P.hub;
    H.h = h;
    const std::vector<int> &perm = F.perm, &iperm = F.iperm;
    H.idx.assign(perm.begin() + k, perm.end());
    H.ap.assign(1, 0);
    H.U.assign((size_t)k * h, 0.0);
    std::vector<double> alpha((size_t)h * h, 0.0), u((size_t)k), acol((size_t)k);
    for (int q = 0; q < h; ++q) {
      const int hq = H.idx[q];
      std::fill(acol.begin(), acol.end(), 0.0);
      for (int p = A.rp[hq]; p < A.rp[hq + 1]; ++p) {
        const int j = iperm[A.ci[p]];
        if (j < k) {
          acol[j] = A.v[p];
          H.apos.push_back(P.out_off[j]);
          H.aval.push_back(A.v[p]);
        } else {
          alpha[(size_t)q * h + (j - k)] = A.v[p];
        }
      }
      H.ap.push_back((int)H.apos.size());
      u = acol;
      piecewise_solve(F, np, k, u);
      for (int j = 0; j < k; ++j) H.U[(size_t)j * h + q] = u[j];
    }
    // Sc = alpha - a^T U (symmetric positive definite), inverted by Gauss-Jordan (h is tiny)
    std::vector<double> Sc((size_t)h * h), Inv((size_t)h * h, 0.0);
    for (int q = 0; q < h; ++q)
      for (int q2 = 0; q2 < h; ++q2) {
        double s = alpha[(size_t)q * h + q2];
        const int hq = H.idx[q];
        for (int p = A.rp[hq]; p < A.rp[hq + 1]; ++p) {
          const int j = iperm[A.ci[p]];
          if (j < k) s -= A.v[p] * H.U[(size_t)j * h + q2];
        }
        Sc[(size_t)q * h + q2] = s;
      }
    for (int q = 0; q < h; ++q) Inv[(size_t)q * h + q] = 1.0;
    for (int c = 0; c < h; ++c) {
      const double piv = Sc[(size_t)c * h + c];
      if (!(piv > 0)) return false;
      for (int j = 0; j < h; ++j) {
        Sc[(size_t)c * h + j] /= piv;
        Inv[(size_t)c * h + j] /= piv;
      }
      for (int i = 0; i < h; ++i) {
        if (i == c) continue;
        const double fct = Sc[(size_t)i * h + c];
        if (fct == 0.0) continue;
        for (int j = 0; j < h; ++j) {
          Sc[(size_t)i * h + j] -= fct * Sc[(size_t)c * h + j];
          Inv[(size_t)i * h + j] -= fct * Inv[(size_t)c * h + j];
        }
      }
    }
    H.Sinv = Inv;
  }
  if (P.idxs.size() & 1) P.idxs.push_back(0);
  if (P.idxs.empty()) P.idxs.assign(2, 0);
  if (!P.sink && P.vals.empty()) P.vals.assign(2, 0.0);
  return P.weights_ok;
}

namespace partinv {
namespace {
// the weights of one fill; w points at weight f.off and the extent of the fill is zero
// w[j nr + q] = row_q[j] for j < hi[q]: the tile's rows are read side by side and the weights leave in storage order
inline void interleave_rows(const double *const *row, const int *hi, int nr, double *w) {
  int hmin = hi[0], hmax = hi[0];
  for (int q = 1; q < nr; ++q) {
    hmin = std::min(hmin, hi[q]);
    hmax = std::max(hmax, hi[q]);
  }
  if (nr == 4) {
    const double *r0 = row[0], *r1 = row[1], *r2 = row[2], *r3 = row[3];
    for (int j = 0; j < hmin; ++j) {
      double *o = w + (size_t)j * 4;
      o[0] = r0[j];
      o[1] = r1[j];
      o[2] = r2[j];
      o[3] = r3[j];
    }
  } else {
    for (int j = 0; j < hmin; ++j)
      for (int q = 0; q < nr; ++q) w[(size_t)j * nr + q] = row[q][j];
  }
  for (int j = std::max(hmin, 0); j < hmax; ++j)
    for (int q = 0; q < nr; ++q)
      if (j < hi[q]) w[(size_t)j * nr + q] = row[q][j];
}
inline void fill_one(const Fill &f, double *w) {
  const int nr = f.nrows, c = f.c, a0 = f.a0;
  const double *row[kSpTile] = {nullptr, nullptr, nullptr, nullptr};
  int hi[kSpTile] = {0, 0, 0, 0};
  if (nr <= 0) return;
  switch (f.kind) {
    case 0:  // rows a0 + q of a lower-triangular matrix
      for (int q = 0; q < nr; ++q) {
        row[q] = f.base + (size_t)(a0 + q) * c;
        hi[q] = std::max(0, std::min(std::min(f.len, c), a0 + q + 1));
      }
      interleave_rows(row, hi, nr, w);
      break;
    case 1:  // rows loc[q] of a matrix with c columns
      for (int q = 0; q < nr; ++q) {
        row[q] = f.base + (size_t)f.loc[q] * c;
        hi[q] = std::max(0, std::min(f.len, c));
      }
      interleave_rows(row, hi, nr, w);
      break;
    case 2:  // rows a0 + q of a full c x c matrix
      for (int q = 0; q < nr; ++q) {
        row[q] = f.base + (size_t)(a0 + q) * c;
        hi[q] = std::max(0, std::min(f.len, c));
      }
      interleave_rows(row, hi, nr, w);
      break;
    case 5: {  // micro-blocks: rows a0 .. a0 + 3, micro-block columns [loc[0], loc[1])
      const int cb0 = f.loc[0], cb1 = f.loc[1];
      for (int cb = cb0; cb < cb1; ++cb)
        for (int ee = 0; ee < 4; ++ee) {
          const int e = cb * 4 + ee;
          if (e >= c) break;
          for (int aa = 0; aa < 4 && a0 + aa < f.m; ++aa)
            w[(size_t)(cb - cb0) * 16 + ee * 4 + aa] = f.base[(size_t)(a0 + aa) * c + e];
        }
      break;
    }
    case 3:  // (D^-T)(a0 + q, a0 + j) = Dinv(a0 + j, a0 + q), j >= q
      for (int j = 0; j < f.len && a0 + j < c; ++j) {
        const double *src = f.base + (size_t)(a0 + j) * c + a0;
        for (int q = 0; q < nr && q <= j; ++q) w[(size_t)j * nr + q] = src[q];
      }
      break;
    default:  // transposed block: base(j, a0 + q)
      for (int j = 0; j < f.len && j < f.m; ++j) {
        const double *src = f.base + (size_t)j * c + a0;
        for (int q = 0; q < nr; ++q) w[(size_t)j * nr + q] = src[q];
      }
      break;
  }
}
}  // namespace

bool write_weights(const std::vector<Fill> &fills, long long total, int nthreads, PartInvHost *P) {
  P->nvals = total;
  if (P->sink && !P->mirrors.empty()) {
    // the sources are on the device (the factor's panels, the pieces' M): the sink forms the weights there
    if (P->sink->fill_on_device(fills, total, P->mirrors, nthreads)) {
      P->vals.clear();
      return true;
    }
    if (P->sources_on_device_only) return false;  // nothing on the host to fall back to
  }
  if (P->sources_on_device_only) return false;
  if (!P->sink) {
    std::vector<double> &vals = P->vals;
    vals.assign((size_t)total, 0.0);
    parallel_for((int)fills.size(), nthreads, 64, [&](int fi) { fill_one(fills[(size_t)fi], vals.data() + fills[(size_t)fi].off); });
    return true;
  }
  // Streamed: the fills are laid out in ascending order of their offsets, so a chunk is a run of consecutive fills;
  // every fill zeroes its own extent (up to the next fill) before it writes, and the chunk travels while the next one
  // is formed.
  WeightSink &sink = *P->sink;
  P->vals.clear();
  if (!sink.begin(total)) return false;
  const int nf = (int)fills.size();
  std::vector<int> order;
  for (int i = 1; i < nf; ++i)
    if (fills[(size_t)i].off < fills[(size_t)i - 1].off) {
      order.resize((size_t)nf);
      for (int q = 0; q < nf; ++q) order[(size_t)q] = q;
      std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return fills[(size_t)a].off < fills[(size_t)b].off; });
      break;
    }
  auto at = [&](int i) -> const Fill & { return fills[(size_t)(order.empty() ? i : order[(size_t)i])]; };
  auto extent_end = [&](int i) { return i + 1 < nf ? at(i + 1).off : total; };
  const long long cap = std::max<long long>(1, sink.chunk_cap());
  int i0 = 0;
  long long off0 = 0;  // start of the chunk: 0, then the offset of its first fill
  while (off0 < total) {
    int i1 = i0;
    long long end = nf ? (i0 < nf ? at(i0).off : total) : total;
    if (i0 < nf) {
      // at least one fill per chunk (an extent never exceeds the cap: a fill is a tile of a few rows)
      end = extent_end(i0);
      i1 = i0 + 1;
      while (i1 < nf && extent_end(i1) - off0 <= cap) {
        end = extent_end(i1);
        ++i1;
      }
    }
    const long long n = end - off0;
    if (n > cap) return false;
    double *buf = sink.acquire(n);
    if (!buf) return false;
    if (i0 < nf && at(i0).off > off0) std::fill(buf, buf + (at(i0).off - off0), 0.0);  // before the first fill
    parallel_for(i1 - i0, nthreads, 64, [&](int q) {
      const Fill &f = at(i0 + q);
      double *w = buf + (f.off - off0);
      std::fill(w, w + (extent_end(i0 + q) - f.off), 0.0);
      fill_one(f, w);
    });
    if (!sink.commit(off0, n)) return false;
    i0 = i1;
    off0 = end;
  }
  return sink.end();
}
}  // namespace partinv

void partitioned_inverse_apply_host(const PartInvHost &P, int r, const double *R, double *Z) {
  const int k = P.k;
  std::vector<double> y((size_t)(2 * k + 4) * r, 0.0);  // padded pairs / K steps may touch up to three unknowns past the end
  for (int j = 0; j < k; ++j)
    for (int t = 0; t < r; ++t) y[(size_t)j * r + t] = R[(size_t)P.perm[j] * r + t];
  std::vector<double> acc;
  std::vector<double> outv;
  for (const SpLevel &lv : P.levels) {
    if (lv.mpipe) {  // matrix-pipe schedule: the wave records executed by plain loops
      struct Out {
        int out, nrows;
        std::vector<double> v;
      };
      std::vector<Out> outs;
      for (int wg = 0; wg < lv.ntasks; ++wg) {
        std::vector<double> part((size_t)kMtWaves * kSpTile * r, 0.0);
        const MWave *W8 = &P.mwaves[(size_t)lv.task0 + (size_t)wg * kMtWaves];
        for (int wv = 0; wv < kMtWaves; ++wv) {
          if (W8[wv].nrows == 0) continue;
          double *pa = &part[(size_t)wv * kSpTile * r];
          for (const MWave *R = &W8[wv]; R; R = R->next >= 0 ? &P.mwaves[(size_t)lv.task0 + R->next] : nullptr) {
            for (int half = 0; half < 2; ++half) {
              const MSub &U = half == 0 ? R->a : R->b;
              const double *w = &P.vals[(size_t)U.w];
              for (int s = U.s0; s < U.s0 + U.n; ++s)
                for (int e = 16 * s; e < 16 * s + 16 && e < 4 * U.ng; ++e) {
                  const size_t u = R->kind == 0 ? (size_t)U.src + e : (size_t)P.idxs[(size_t)U.src + e];
                  for (int i = 0; i < R->nrows; ++i) {
                    double wj;
                    if (R->kind == 0) {
                      const int a = U.loc[i];
                      if (a < 0) continue;
                      wj = w[((size_t)(a >> 2) * U.ncb + (e >> 2)) * 16 + (e & 3) * 4 + (a & 3)];
                    } else {
                      const int col = U.loc[0] + i;
                      wj = w[((size_t)(e >> 2) * U.ncb + (col >> 2)) * 16 + (col & 3) * 4 + (e & 3)];
                    }
                    if (wj == 0.0) continue;
                    for (int t = 0; t < r; ++t) pa[(size_t)i * r + t] += wj * y[u * r + t];
                  }
                }
            }
          }
        }
        for (int wv = 0; wv < kMtWaves; ++wv) {
          const MWave &R = W8[wv];
          if (R.nrows == 0 || R.red_first != wv) continue;
          Out o;
          o.out = R.out;
          o.nrows = R.nrows;
          o.v.assign((size_t)R.nrows * r, 0.0);
          for (int a = 0; a < R.nrows; ++a)
            for (int t = 0; t < r; ++t) {
              double v = 0;
              for (int q = 0; q < R.red_n; ++q) v += part[((size_t)(wv + q) * kSpTile + a) * r + t];
              o.v[(size_t)a * r + t] = v + (R.carry >= 0 ? y[((size_t)R.carry + a) * r + t] : 0.0);
            }
          outs.push_back(std::move(o));
        }
      }
      for (const Out &o : outs)
        for (int a = 0; a < o.nrows; ++a)
          for (int t = 0; t < r; ++t) y[((size_t)o.out + a) * r + t] = o.v[(size_t)a * r + t];
      continue;
    }
    // every task of a level reads the state before the level: evaluate all, then store
    outv.assign((size_t)lv.ntasks * kSpTile * r, 0.0);
    for (int q = 0; q < lv.ntasks; ++q) {
      const PTask &T = P.tasks[(size_t)lv.task0 + q];
      acc.assign((size_t)T.nrows * r, 0.0);
      for (int s = T.seg0; s < T.seg0 + T.nseg; ++s) {
        const PSeg &S = P.segs[(size_t)s];
        const double *w = &P.vals[(size_t)S.w];
        for (int j = 0; j < S.len; ++j) {
          const size_t u = (S.src >= 0) ? (size_t)S.src + j : (size_t)P.idxs[(size_t)S.idx + j];
          for (int a = 0; a < T.nrows; ++a) {
            const double wj = w[(size_t)j * T.nrows + a];
            for (int t = 0; t < r; ++t) acc[(size_t)a * r + t] += wj * y[u * r + t];
          }
        }
      }
      for (int a = 0; a < T.nrows; ++a)
        for (int t = 0; t < r; ++t)
          outv[((size_t)q * kSpTile + a) * r + t] =
              acc[(size_t)a * r + t] + (T.carry >= 0 ? y[((size_t)T.carry + a) * r + t] : 0.0);
    }
    for (int q = 0; q < lv.ntasks; ++q) {
      const PTask &T = P.tasks[(size_t)lv.task0 + q];
      for (int a = 0; a < T.nrows; ++a)
        for (int t = 0; t < r; ++t) y[((size_t)T.out + a) * r + t] = outv[((size_t)q * kSpTile + a) * r + t];
    }
  }
  const PartInvHub &H = P.hub;
  std::vector<double> x2((size_t)H.h * r, 0.0);
  if (H.h > 0) {
    std::vector<double> w((size_t)H.h * r);
    for (int q = 0; q < H.h; ++q)
      for (int t = 0; t < r; ++t) {
        double s = R[(size_t)H.idx[q] * r + t];
        for (int p = H.ap[q]; p < H.ap[q + 1]; ++p) s -= H.aval[p] * y[(size_t)H.apos[p] * r + t];
        w[(size_t)q * r + t] = s;
      }
    for (int q = 0; q < H.h; ++q)
      for (int t = 0; t < r; ++t) {
        double s = 0;
        for (int q2 = 0; q2 < H.h; ++q2) s += H.Sinv[(size_t)q * H.h + q2] * w[(size_t)q2 * r + t];
        x2[(size_t)q * r + t] = s;
        Z[(size_t)H.idx[q] * r + t] = s;
      }
  }
  for (int j = 0; j < k; ++j)
    for (int t = 0; t < r; ++t) {
      double v = y[(size_t)P.out_off[j] * r + t];
      for (int q = 0; q < H.h; ++q) v -= H.U[(size_t)j * H.h + q] * x2[(size_t)q * r + t];
      Z[(size_t)P.perm[j] * r + t] = v;
    }
}

}  // namespace dcora
