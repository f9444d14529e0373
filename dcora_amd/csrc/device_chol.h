// Sparse Cholesky factorisation on the device: the numeric half of the positive-semidefiniteness test of the dual
// certificate (ref src/DCORA_utils.cpp:1737-1747, isSparseSymmetricMatrixPSD: CHOLMOD LL^T of S + eta I, PSD <=> the
// factorisation succeeds) and of every other complete factorisation the path needs.
//
// Method: multifrontal LL^T over the pieces of the nested dissection of host_sparse.cpp (leaf sub-domains and
// separators, each a contiguous column range of the permuted matrix).  A piece s with c columns and m rows below
// them owns a dense front of (c + m)^2 doubles; the fronts of ALL pieces live in one arena (sized for 288 GB of HBM:
// no stack, no reuse, so every front of a tree level can be worked on at once).  A level of the piece tree is
//   extend-add of the children's Schur complements (one launch per child slot: fixed order, no atomics),
//   then a blocked right-looking partial factorisation of the level's fronts, 64 columns at a time:
//   diagonal block (one workgroup per front: LL^T in LDS, its inverse for the panel), panel = rows * L^-T and the
//   trailing update as 64 x 64 register-tiled fp64 products.
// A pivot that is not positive sets a flag that turns every later launch into a no-op (CHOLMOD's
// quick_return_if_not_posdef).  The host does the symbolic analysis once per sparsity pattern (ordering, piece
// structures closed under the piece tree, scatter map of the matrix entries); it is cached on the pattern.
#pragma once
#include <cstddef>
#include <functional>
#include <memory>
#include <vector>

#include "host_sparse.h"

namespace dcora {

constexpr int kCholNb = 64;  // panel width of the dense partial factorisations

struct CholPiece {
  long long off = 0;  // front, row-major (c + m) x (c + m), lower triangle meaningful
  int c0 = 0, c = 0, m = 0;
  int parent = -1, level = 0;
  int rows_off = 0;   // into rows / rel: the m rows below the piece (permuted numbering, ascending)
};

struct CholSymbolic {
  int n = 0, nlev = 0;
  std::vector<int> perm, iperm;
  std::vector<CholPiece> pieces;
  std::vector<int> rows;          // concatenated row lists
  std::vector<int> rel;           // same layout: index of the row inside the PARENT's front
  std::vector<long long> a_dest;  // per stored entry of the input CSR: where it goes in the arena, -1 = upper triangle
  long long arena = 0;            // doubles
  std::vector<int> level_ptr, level_pieces;  // pieces of a level, widest (largest c) first
  std::vector<int> slot_ptr, slot_children;  // child lists; slots of level t: slot_level_ptr[t] .. slot_level_ptr[t+1]
  std::vector<int> slot_level_ptr;
  double flops = 0;
  int nhub = 0;
};

// pattern of A only (both triangles stored, diagonal present); block = unknowns ordered together; top_unknowns: see
// amd_like_order (0 = plain dissection tree, what a factorisation for its own sake wants)
void chol_symbolic(const HostCsr &A, int block, CholSymbolic *out, int top_unknowns = 0);

// the same schedule executed by plain host loops (validation of the symbolic analysis without a GPU; tests only).
// fronts receives the arena; returns false at the first non-positive pivot
bool chol_numeric_host(const CholSymbolic &S, const double *vals, std::vector<double> *fronts);

// device numeric factorisation of the matrix whose values (CSR order of the analysed pattern) are vals (host).
// *pd = every pivot positive.  Symbolic analyses are cached on the pattern.
int device_chol_is_pd(const HostCsr &A, int block, int device, bool *pd, double *info8 = nullptr);

// the factor itself, by pieces (sparse_precond.h), for the builder of the partitioned inverse: ordered with
// top_unknowns (nd_top_default() for the replay).  DCORA_ERR_NOT_PD when a pivot is not positive.
struct PiecewiseFactor;
int device_chol_piecewise_factor(const HostCsr &A, int block, int top_unknowns, int device, PiecewiseFactor *out,
                                 double *info8 = nullptr);

// partitioned inverse of A (sparse_precond.h) with the numeric factorisation on the device and the piece inverses and
// the replay schedule on the host; DCORA_FACTOR=host keeps the host factorisation (A/B measurements).
// DCORA_OK, DCORA_ERR_NOT_PD, or a HIP error.
struct PartInvHost;
int build_partitioned_inverse_auto(const HostCsr &A, int block, int nthreads, int device, PartInvHost *out);

// dense inverse of a small SPD matrix, entirely on the device: Minv (device, k rows of ldm >= k doubles, row-major,
// both triangles) <- A^-1.  *pd = false (and Minv undefined) when A is not positive definite.
int device_dense_spd_inverse(const HostCsr &A, int device, double *Minv, int ldm, bool *pd);
// the same for a batch of matrices of EQUAL size in one set of launches (bitwise the single build per matrix); pd gets one
// verdict for the whole batch (a failure anywhere: build one by one to find it)
int device_dense_spd_inverse_batch(const std::vector<const HostCsr *> &As, int device, const std::vector<double *> &Minv,
                                   int ldm, std::vector<char> *pd);

// X = A^-1 B for a sparse SPD matrix and up to 16 right-hand sides (contiguous per unknown) through the partitioned
// inverse built from the device factorisation and ONE replay on the device: the solver the chordal initialisation
// takes at sizes where the host factorisation takes minutes (host_graph.h, SpdSolve)
std::function<bool(const HostCsr &, int, int, const double *, double *)> device_spd_solver(int device);

void chol_cache_clear();

// The symbolic analysis, the device image and the arena of the factorisation of A's PATTERN (values ignored), left in
// the cache: a later device_chol_is_pd of a matrix with this pattern starts with its numeric phase.  The certificate
// S = Q - Lambda has Q's pattern, which is known before the solve starts: a driver prepares it on another host thread
// while the agents iterate (dcora_cert_prepare).
int device_chol_prepare(const HostCsr &A, int block, int device);

// recycled device scratch (at most two idle arenas per process; chol_cache_clear() drops them): callers that need a few
// temporary buffers per call take ONE block here instead of paying a hipMalloc / hipFree pair per buffer
char *scratch_acquire(int device, size_t bytes);
void scratch_release(int device, char *p, size_t bytes);
// the same for pinned HOST memory (staging of small transfers): asynchronous copies from / to pageable memory -- a
// std::vector, a variable on the stack -- stalled for 16 - 38 ms now and then on this stack (seen inside the certificate's
// clock); through a pinned buffer they take their 0.1 ms.  nullptr when the allocation fails (callers fall back).
char *pinned_acquire(size_t bytes);
void pinned_release(char *p, size_t bytes);

}  // namespace dcora
