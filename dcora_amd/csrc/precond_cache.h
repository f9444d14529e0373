// Cache of preconditioner images keyed on the content of Q + reg I (Ownership): the reference
// re-creates its QuadraticProblem on every Agent::updateX (ref src/Agent.cpp:1252) while Graph caches Q and its
// factor (ref src/Graph.cpp:523-533, 1901-1917); here the device-resident inverse -- dense or partitioned -- is built
// once per distinct matrix and shared (read-only) by every problem that asks for it again: the staircase levels of a
// driver (the inverse does not depend on r), problems re-created per update, sessions re-created per GNC round.
// Least-recently-used entries are dropped beyond DCORA_PRECOND_CACHE_MB (default 8192; 0 disables the cache); an
// entry in use stays alive through its shared_ptr.
#pragma once
#include <cstdint>
#include <memory>

#include "device_problem.h"

namespace dcora {

struct PrecondKey {
  uint64_t h0 = 0, h1 = 0;
  int k = 0, nnz = 0, block = 0, device = 0, sparse = 0;
  double reg = 0;
  // A hit attaches another problem's inverse: two 64-bit hashes are not left to decide that alone.  The guard holds
  // what a colliding matrix would also have to reproduce exactly: the plain sums of its values and of its column
  // indices (independent of the hash's mixing) and 24 entries sampled at fixed strides (value, column index, and the
  // row pointers at those strides).
  static constexpr int kSamples = 24;
  double vsum = 0, vabs = 0;
  long long cisum = 0;
  double vsample[kSamples] = {0};
  int cisample[kSamples] = {0}, rpsample[kSamples] = {0};
  bool operator==(const PrecondKey &o) const {
    if (!(h0 == o.h0 && h1 == o.h1 && k == o.k && nnz == o.nnz && block == o.block && device == o.device &&
          sparse == o.sparse && reg == o.reg && vsum == o.vsum && vabs == o.vabs && cisum == o.cisum))
      return false;
    for (int i = 0; i < kSamples; ++i)
      if (vsample[i] != o.vsample[i] || cisample[i] != o.cisample[i] || rpsample[i] != o.rpsample[i]) return false;
    return true;
  }
};
PrecondKey make_precond_key(const HostCsr &Q, double reg, int block, int device, bool sparse);

struct PrecondEntry {
  std::shared_ptr<const DevBuf<double>> dense;  // k x ldm symmetric inverse
  int ldm = 0;
  std::shared_ptr<const SpImage> sparse;
  long nnzL = 0;
  size_t bytes = 0;
  double build_ms = 0;
  // built ahead of the problem that needs it (precond_prebuild_dense: the agents of a session in one batch): the first
  // look-up that finds it is that problem's -- it counts as the MISS its build was, later ones as hits
  bool prebuilt = false;
};

// count = false: a look that leaves the statistics alone (the batch builder asking what is there already)
bool precond_cache_find(const PrecondKey &key, PrecondEntry *out, bool count = true);
void precond_cache_insert(const PrecondKey &key, const PrecondEntry &e);
// hits, misses, entries, bytes held
void precond_cache_stats(double *stats4);
void precond_cache_clear();

}  // namespace dcora
