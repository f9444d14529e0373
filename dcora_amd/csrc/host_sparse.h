// Host-side sparse helpers of the product: CSR containers, a fill-reducing ordering and a sparse Cholesky
// used (a) once per Q to build the preconditioner (Q + reg I)^-1 that is applied on the GPU
// (ref src/Graph.cpp:1901-1917) and (b) for the PSD test of the dual certificate
// (ref src/DCORA_utils.cpp:1737-1747).  Setup-time code; the per-iteration path never comes here.
#pragma once
#include <cstddef>
#include <utility>
#include <vector>

namespace dcora {

struct HostCsr {
  int n = 0;      // rows
  int ncols = 0;  // columns (== n for square)
  std::vector<int> rp, ci;
  std::vector<double> v;
  int nnz() const { return (int)ci.size(); }
};

// block-CSR with dense bs x bs blocks, pattern = union of the scalar entries' blocks.  Inside a block entry (a, c) --
// scalar row block_row * bs + a, scalar column block_col * bs + c -- sits at c * bs + a (COLUMN-major): the lane that
// owns column c of the neighbour's r x (d+1) block in the Q-apply reads the bs weights it needs as one contiguous run
struct HostBsr {
  int nbrows = 0, bs = 0;
  std::vector<int> bp, bc;
  std::vector<double> bv;
  int nblocks() const { return (int)bc.size(); }
};
HostBsr bsr_from_csr(const HostCsr &A, int bs);

// builds a CSR from (row, col, val) triplets, summing duplicates, columns sorted inside rows
HostCsr csr_from_coo(int nrows, int ncols, const std::vector<int> &I, const std::vector<int> &J,
                     const std::vector<double> &V);
HostCsr csr_shift_diag(const HostCsr &A, double s);

class SparseChol {
 public:
  // A symmetric (both triangles); block = size of the index groups that are ordered together (d+1 for pose graphs).
  // Returns false when a pivot is not positive (matrix not PD): quick return, like CHOLMOD's
  // quick_return_if_not_posdef.
  // top_unknowns > 0: the upper separators of the dissection, as many depths as fit that many unknowns, are
  // ordered last as ONE piece (the dense top the device replay of sparse_precond.h wants)
  bool factor(const HostCsr &A, int block, int top_unknowns = 0);
  bool ok() const { return ok_; }
  long nnzL() const { return (long)Li_.size(); }
  int n() const { return n_; }
  // x = A^-1 b for nrhs right-hand sides stored contiguously per unknown: B[i*nrhs + t]
  void solve_inplace(double *B, int nrhs) const;  // B in permuted order
  const std::vector<int> &perm() const { return perm_; }
  const std::vector<int> &iperm() const { return iperm_; }
  // factor in compressed columns (diagonal first, then increasing rows), permuted numbering
  const std::vector<int> &Lp() const { return Lp_; }
  const std::vector<int> &Li() const { return Li_; }
  const std::vector<double> &Lx() const { return Lx_; }
  // boundaries (permuted scalar columns, ascending, first 0, last n) of the dissection pieces: every leaf
  // sub-domain and every separator is one contiguous column range
  const std::vector<int> &pieces() const { return pieces_; }
  // hubs (unknowns coupled to a large share of all others, e.g. a landmark ranged from every pose) are ordered
  // last and form the last piece: the trailing nhub() columns
  int nhub() const { return nhub_; }
  // dense inverse written row-major with leading dimension ld (>= n), using nthreads host threads
  void dense_inverse(double *out, size_t ld, int nthreads) const;
  void solve_vec(const double *b, double *x) const;

 private:
  int n_ = 0, nhub_ = 0;
  bool ok_ = false;
  std::vector<int> perm_, iperm_, Lp_, Li_, pieces_;
  std::vector<double> Lx_;
};

// nested-dissection order; pieces (optional) receives the column boundaries of the leaves / separators
// col_tasks (optional): about want_tasks disjoint column ranges [begin, end) of the order, each closed under
// elimination dependencies (a dissection sub-tree): they can be factorised independently of each other
std::vector<int> amd_like_order(const HostCsr &A, int block, std::vector<int> *pieces = nullptr,
                                int *nhub_cols = nullptr, std::vector<std::pair<int, int>> *col_tasks = nullptr,
                                int want_tasks = 0,
                                std::vector<std::vector<std::pair<int, int>>> *col_waves = nullptr,
                                int top_unknowns = 0);
int nd_top_default();  // 3072 unknowns (a constant: 1536 / 2048 / 4096 / 6144 measured worse)

// CPUs this process may really use: the smallest of the hardware count, the affinity mask and the cgroup quota (a
// container on a 256-core host is often limited to 16: more threads than that only slow the set-up down -- measured:
// the central 400 000-unknown preconditioner 14.0 s with 16 threads, 18.5 s with 128); DCORA_HOST_THREADS overrides
int host_cpus_available();

}  // namespace dcora
