// Certification entry points of the product (see cert.hip).
#pragma once
#include <cstdint>
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
  DevCsr Sd;
  // shift-and-invert mode: when set, a Lanczos step applies (S - sigma I)^-1 through the partitioned sparse inverse
  // instead of S - shift I
  const SparsePrecond *inverse_op = nullptr;
  DevBuf<double> V, Vtmp, w, part, small;
  ~DeviceLanczos();
  int init(const HostCsr &S, int device_);
  int largest_magnitude(double shift, int ncv, int maxit, double tol, const double *x0, uint64_t seed,
                        LanczosResult *out);
};

int device_min_eig(const HostCsr &S, int maxit, double min_eig_tol, int ncv, uint64_t seed, int device,
                   LanczosResult *out);
int device_dual_certificate(const dcora_dims &dims, const double *Xh, const HostCsr &Q, int device, HostCsr *S);
int host_is_psd(const HostCsr &S, int block, bool *psd);
int device_fast_verification(const HostCsr &S, double eta, int block, int device, bool *psd, double *theta,
                             std::vector<double> *x, double *lambda_min, long *matvecs);

}  // namespace dcora
