// Certification entry points of the product (see cert.hip).
#pragma once
#include <cstdint>
#include <functional>
#include <vector>

#include "device_problem.h"

namespace dcora {

struct LanczosResult {
  bool ok = false;
  double lambda = 0;
  std::vector<double> v;
  long matvecs = 0;
};

class DeviceLanczos {
 public:
  static constexpr int kMaxNcv = 20;
  int device = 0, n = 0;
  hipStream_t st = nullptr;
  bool own_stream = true;
  DevCsr Sd;
  // shift-and-invert mode: when set, a Lanczos step applies (S - sigma I)^-1 through the partitioned sparse inverse
  // instead of S - shift I
  const SparsePrecond *inverse_op = nullptr;
  // Row-block form (one process per GPU, exchange.hip): this object holds rows [lo, lo + n) of a matrix of order
  // n_global.  op applies the row block (it moves the halo entries itself); allreduce sums `count` doubles over the
  // ranks in rank order, the same result on every rank.  Start vectors are generated for the whole problem on every
  // rank (same seed) and sliced, so the run does not depend on how the rows are split.
  std::function<int(const double *v, double *w)> op;
  std::function<int(double *vals, int count)> allreduce;
  int n_global = 0, lo = 0;
  // rows of this rank that are NOT one contiguous range of the whole problem (range-aided sessions: an agent's
  // variables are scattered over the global ordering): global index of every local row; empty = [lo, lo + n)
  std::vector<int> row_map;
  DevBuf<double> V, Vtmp, w, part, small;
  ~DeviceLanczos();
  int init(const HostCsr &S, int device_);
  int init_rows(int n_local, int n_global_, int lo_, int device_, hipStream_t stream);
  int largest_magnitude(double shift, int ncv, int maxit, double tol, const double *x0, uint64_t seed,
                        LanczosResult *out);
};

std::vector<double> min_eig_second_start(const HostCsr &S, uint64_t seed);
int device_min_eig(const HostCsr &S, int maxit, double min_eig_tol, int ncv, uint64_t seed, int device,
                   LanczosResult *out);
int device_dual_certificate(const dcora_dims &dims, const double *Xh, const HostCsr &Q, int device, HostCsr *S);
int host_is_psd(const HostCsr &S, int block, bool *psd);
int device_fast_verification(const HostCsr &S, double eta, int block, int device, bool *psd, double *theta,
                             std::vector<double> *x, double *lambda_min, long *matvecs);

}  // namespace dcora
