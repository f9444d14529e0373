// Host-side construction of the partitioned inverse of a nested-dissection Cholesky factor (sparse_precond.h):
// pieces, their inverses, hubs; the schedule of the replay and the layout of the stored weights are host_partinv3.cpp's.
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
#include "env.h"
#include "sparse_precond.h"

namespace dcora {

using namespace partinv;

namespace {

constexpr int kBigPiece = 384;  // pieces this wide are inverted by all threads together (the merged top of the tree)

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
  const bool timing = env::init_timing();
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
  const bool timing = env::init_timing();
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
  // ---- schedule and stored weights: the matrix-pipe schedule (host_partinv3.cpp) ----
  {
    std::vector<const double *> Mgiven((size_t)np, nullptr);
    for (int s = 0; s < np; ++s)
      if (F.pieces[s].mtop()) Mgiven[s] = F.pieces[s].mtop();
    layout_mpipe(pc, Mgiven, piece_of, k, nlev, nthreads, timing, &P);
    if (timing)
      std::fprintf(stderr, "[partinv] k %d pieces %d levels %d: piece inverses %.1f, schedule %.1f ms\n", k, np, nlev,
                   tms(T1, T2), tms(T2, tnow()));
  }
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
// the weights of one fill (micro-blocks: rows a0 .. a0 + 3 of an m x c matrix, micro-block columns [loc[0], loc[1]));
// w points at weight f.off and the extent of the fill is zero
inline void fill_one(const Fill &f, double *w) {
  const int c = f.c, a0 = f.a0, cb0 = f.loc[0], cb1 = f.loc[1];
  for (int cb = cb0; cb < cb1; ++cb)
    for (int ee = 0; ee < 4; ++ee) {
      const int e = cb * 4 + ee;
      if (e >= c) break;
      for (int aa = 0; aa < 4 && a0 + aa < f.m; ++aa)
        w[(size_t)(cb - cb0) * 16 + ee * 4 + aa] = f.base[(size_t)(a0 + aa) * c + e];
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
      // at least one fill per chunk (an extent never exceeds the cap: a fill is at most 2048 weights)
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
  for (const SpLevel &lv : P.levels) {
    {  // the wave records executed by plain loops: every tile of a launch reads the state before the launch
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
