// Symbolic analysis for the multifrontal device Cholesky (device_chol.h) and a plain host executor of the same
// schedule (validation only).  Setup-time code: once per sparsity pattern.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "device_chol.h"
#include "env.h"

namespace dcora {

void chol_symbolic(const HostCsr &A, int block, CholSymbolic *out, int top_unknowns) {
  CholSymbolic &S = *out;
  S = CholSymbolic();
  const int n = A.n;
  S.n = n;
  std::vector<int> cuts;
  int nhub = 0;
  const bool timing = env::init_timing();
  auto tl = std::chrono::steady_clock::now();
  auto lap = [&](const char *what) {
    if (!timing) return;
    const auto now = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[symbolic] %-22s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - tl).count());
    tl = now;
  };
  S.perm = amd_like_order(A, block, &cuts, &nhub, nullptr, 0, nullptr, top_unknowns);
  lap("ordering");
  S.nhub = nhub;
  S.iperm.assign((size_t)n, 0);
  for (int i = 0; i < n; ++i) S.iperm[S.perm[i]] = i;
  const int np = (int)cuts.size() - 1;
  S.pieces.assign((size_t)np, CholPiece());
  std::vector<int> piece_of((size_t)n);
  for (int s = 0; s < np; ++s) {
    S.pieces[s].c0 = cuts[s];
    S.pieces[s].c = cuts[s + 1] - cuts[s];
    for (int j = cuts[s]; j < cuts[s + 1]; ++j) piece_of[j] = s;
  }
  // structure of every piece, closed under the piece tree: rows(s) = (rows of A below the piece in its columns)
  // united with the rows of its children outside the piece.  A separator taken from a BFS level is usually not
  // connected in itself, so the union over the columns of the scalar factor would not be closed; the closure only
  // adds explicit zeros.
  std::vector<std::vector<int>> children((size_t)np);
  std::vector<int> mark((size_t)n, -1), tmp;
  for (int s = 0; s < np; ++s) {
    CholPiece &P = S.pieces[s];
    const int hi = P.c0 + P.c;
    tmp.clear();
    for (int j = P.c0; j < hi; ++j) {
      const int jo = S.perm[j];
      for (int p = A.rp[jo]; p < A.rp[jo + 1]; ++p) {
        const int i = S.iperm[A.ci[p]];
        if (i >= hi && mark[i] != s) {
          mark[i] = s;
          tmp.push_back(i);
        }
      }
    }
    for (int d : children[s]) {
      const CholPiece &D = S.pieces[d];
      for (int a = 0; a < D.m; ++a) {
        const int i = S.rows[(size_t)D.rows_off + a];
        if (i >= hi && mark[i] != s) {
          mark[i] = s;
          tmp.push_back(i);
        }
      }
      P.level = std::max(P.level, D.level + 1);
    }
    std::sort(tmp.begin(), tmp.end());
    P.m = (int)tmp.size();
    P.rows_off = (int)S.rows.size();
    S.rows.insert(S.rows.end(), tmp.begin(), tmp.end());
    if (P.m > 0) {
      P.parent = piece_of[tmp[0]];
      children[P.parent].push_back(s);
    }
    const double c = P.c, m = P.m;
    S.flops += c * c * c / 3.0 + m * c * c + m * m * c;
  }
  lap("piece structures");
  // fronts
  long long off = 0;
  for (CholPiece &P : S.pieces) {
    P.off = off;
    const long long f = (long long)P.c + P.m;
    off += f * f;
    off = (off + 1) & ~1LL;
    S.nlev = std::max(S.nlev, P.level + 1);
  }
  S.arena = off;
  // position of every row of a piece inside the parent's front (both lists ascending: one merge walk)
  S.rel.assign(S.rows.size(), -1);
  for (int s = 0; s < np; ++s) {
    const CholPiece &P = S.pieces[s];
    if (P.parent < 0) continue;
    const CholPiece &Q = S.pieces[P.parent];
    int b = 0;
    for (int a = 0; a < P.m; ++a) {
      const int i = S.rows[(size_t)P.rows_off + a];
      if (i < Q.c0 + Q.c) {
        S.rel[(size_t)P.rows_off + a] = i - Q.c0;
      } else {
        while (b < Q.m && S.rows[(size_t)Q.rows_off + b] < i) ++b;
        // closure: the row is there
        S.rel[(size_t)P.rows_off + a] = Q.c + b;
      }
    }
  }
  lap("fronts + positions");
  // scatter map of the entries of the lower triangle of P A P^T
  S.a_dest.assign((size_t)A.nnz(), -1);
  {
    // every entry on its own: the rows are shared out over the host's threads (6.2 M entries for the 100k lattice)
    auto rows_range = [&](int lo, int hi) {
      for (int io = lo; io < hi; ++io) {
        const int i = S.iperm[io];
        for (int p = A.rp[io]; p < A.rp[io + 1]; ++p) {
          const int j = S.iperm[A.ci[p]];
          if (i < j) continue;
          const CholPiece &P = S.pieces[piece_of[j]];
          const long long f = (long long)P.c + P.m;
          long long li;
          if (i < P.c0 + P.c) {
            li = i - P.c0;
          } else {
            const int *r0 = &S.rows[(size_t)P.rows_off];
            li = P.c + (std::lower_bound(r0, r0 + P.m, i) - r0);
          }
          S.a_dest[(size_t)p] = P.off + li * f + (j - P.c0);
        }
      }
    };
    const int nt = n >= 20000 ? std::max(1, std::min(host_cpus_available(), 16)) : 1;
    std::vector<std::thread> th;
    for (int t = 1; t < nt; ++t)
      th.emplace_back(rows_range, (int)((long long)n * t / nt), (int)((long long)n * (t + 1) / nt));
    rows_range(0, (int)((long long)n / nt));
    for (auto &t : th) t.join();
  }
  lap("scatter map");
  // schedule: pieces by level (widest first), children by level of the parent and position among the siblings
  std::vector<std::vector<int>> by_level((size_t)S.nlev);
  for (int s = 0; s < np; ++s) by_level[S.pieces[s].level].push_back(s);
  S.level_ptr.assign(1, 0);
  S.slot_level_ptr.assign(1, 0);
  S.slot_ptr.assign(1, 0);
  for (int t = 0; t < S.nlev; ++t) {
    std::vector<int> &L = by_level[t];
    std::stable_sort(L.begin(), L.end(), [&](int a, int b) { return S.pieces[a].c > S.pieces[b].c; });
    S.level_pieces.insert(S.level_pieces.end(), L.begin(), L.end());
    S.level_ptr.push_back((int)S.level_pieces.size());
    size_t maxch = 0;
    for (int s : L) maxch = std::max(maxch, children[s].size());
    for (size_t k = 0; k < maxch; ++k) {
      for (int s : L)
        if (children[s].size() > k) S.slot_children.push_back(children[s][k]);
      S.slot_ptr.push_back((int)S.slot_children.size());
    }
    S.slot_level_ptr.push_back((int)S.slot_ptr.size() - 1);
  }
}

bool chol_numeric_host(const CholSymbolic &S, const double *vals, std::vector<double> *fronts) {
  std::vector<double> &F = *fronts;
  F.assign((size_t)S.arena, 0.0);
  for (size_t p = 0; p < S.a_dest.size(); ++p)
    if (S.a_dest[p] >= 0) F[(size_t)S.a_dest[p]] += vals[p];
  for (int t = 0; t < S.nlev; ++t) {
    for (int sl = S.slot_level_ptr[t]; sl < S.slot_level_ptr[t + 1]; ++sl)
      for (int q = S.slot_ptr[sl]; q < S.slot_ptr[sl + 1]; ++q) {
        const CholPiece &D = S.pieces[S.slot_children[q]];
        const CholPiece &P = S.pieces[D.parent];
        const long long fd = (long long)D.c + D.m, fp = (long long)P.c + P.m;
        const int *rel = &S.rel[(size_t)D.rows_off];
        for (int a = 0; a < D.m; ++a)
          for (int b = 0; b <= a; ++b)
            F[(size_t)(P.off + rel[a] * fp + rel[b])] += F[(size_t)(D.off + (D.c + a) * fd + (D.c + b))];
      }
    for (int q = S.level_ptr[t]; q < S.level_ptr[t + 1]; ++q) {
      const CholPiece &P = S.pieces[S.level_pieces[q]];
      const long long f = (long long)P.c + P.m;
      double *M = &F[(size_t)P.off];
      for (int j = 0; j < P.c; ++j) {
        const double d = M[j * f + j];
        if (!(d > 0)) return false;
        const double s = std::sqrt(d);
        M[j * f + j] = s;
        for (long long i = j + 1; i < f; ++i) M[i * f + j] /= s;
        for (long long k = j + 1; k < f; ++k) {
          const double lk = M[k * f + j];
          if (lk == 0.0) continue;
          for (long long i = k; i < f; ++i) M[i * f + k] -= M[i * f + j] * lk;
        }
      }
    }
  }
  return true;
}

}  // namespace dcora
