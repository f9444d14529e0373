// Shared between the builder of the partitioned inverse (host_partinv.cpp) and its schedule (host_partinv3.cpp).
// Set-up time code, host only.
#pragma once
#include <algorithm>
#include <atomic>
#include <thread>
#include <vector>

#include "sparse_precond.h"

namespace dcora {
namespace partinv {

struct Piece {
  int c0 = 0, c = 0;          // columns [c0, c0 + c) in permuted numbering
  std::vector<int> rows;      // rows of L below the piece, ascending
  std::vector<int> src;       // where each of them sits in the factor's panel (row c + src)
  int level = 0;
  std::vector<double> Dinv;   // D^-1, c x c row-major (lower triangular)
  std::vector<double> W;      // -B D^-1, m x c row-major
  // a piece the device delivered inverted, with no row dropped, is read where it arrived (views into the factor's
  // host block) instead of being copied: 3 GB for the whole 100k lattice
  const double *Dinv_view = nullptr, *W_view = nullptr;
  const double *dinv() const { return Dinv_view ? Dinv_view : (Dinv.empty() ? nullptr : Dinv.data()); }
  const double *w() const { return W_view ? W_view : W.data(); }
};

inline int pad4(int x) { return (x + 3) & ~3; }

// body(i) for i in [0, n) on up to nthreads threads, dynamic chunks
template <class F>
void parallel_for(int n, int nthreads, int chunk, F body) {
  nthreads = std::max(1, std::min(nthreads, (n + chunk - 1) / chunk));
  std::atomic<int> next(0);
  auto work = [&]() {
    for (;;) {
      const int i0 = next.fetch_add(chunk);
      if (i0 >= n) break;
      const int i1 = std::min(n, i0 + chunk);
      for (int i = i0; i < i1; ++i) body(i);
    }
  };
  std::vector<std::thread> th;
  for (int t = 1; t < nthreads; ++t) th.emplace_back(work);
  work();
  for (auto &t : th) t.join();
}

// The stored weights are only RESERVED while a schedule is laid out; writing them -- 0.4 G doubles for the whole 100k
// lattice -- is done afterwards, on the device where the sources are there, else by all host threads.  A fill is a
// run of micro-blocks (sparse_precond.h): rows a0 .. a0 + 3 of an m x c matrix `base` (row-major), micro-block columns
// [loc[0], loc[1]):  w[(cb - loc[0]) 16 + (e % 4) 4 + a % 4] = base(a, e), zero beyond the matrix; kind is 5 (kinds
// 0 .. 4 were the per-tile layouts of the schedules this one replaced).
struct Fill {
  long long off;
  const double *base;
  int kind, nrows, len, c, a0, m;
  int loc[kSpTile];
};
// into P->vals, or chunk by chunk into P->sink when one is set; false when the sink failed
bool write_weights(const std::vector<Fill> &fills, long long total, int nthreads, PartInvHost *P);

// The schedule builder (host_partinv3.cpp).  pc: pieces with Dinv and W; Mgiven[s] (may be null): D^-T D^-1 of piece s
// where the device delivered it.  Fills P->levels / mwaves / idxs / vals / out_off / weights_read_per_apply.
void layout_mpipe(const std::vector<Piece> &pc, const std::vector<const double *> &Mgiven,
                  const std::vector<int> &piece_of, int k, int nlev, int nthreads, bool timing, PartInvHost *P);

}  // namespace partinv
}  // namespace dcora
