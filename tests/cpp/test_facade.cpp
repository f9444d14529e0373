// C++ host-side smoke test of the reference-shaped facade (dcora_amd/include/DCORA/*.h) over the C ABI.
// Mirrors ref tests/testRobust.cpp:162-226 (testPrior): RTR converges onto a pose prior within 1e-6.
// Exit code 0 = pass, 2 = no GPU (the library has no CPU fallback), 1 = failure.
#include <cmath>
#include <cstdio>

#include "DCORA/DCORA_utils.h"
#include "DCORA/QuadraticOptimizer.h"

int main() {
  if (dcora_device_count() < 1) {
    std::printf("no GPU: facade compiled and linked, compute skipped\n");
    return 2;
  }
  const int d = 3, n = 2, r = 3;
  // one odometry edge 0 -> 1 with R = I, t = 0, kappa 1e4, tau 1e2
  int ids[4] = {0, 0, 0, 1};
  double vals[9 + 3 + 3] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 10000, 100, 1};
  dcora_csr_t Qh;
  DCORA::check_status(dcora_graph_build_Q_pgo(d, n, 0, 1, ids, vals, &Qh), "build Q");
  DCORA::ProblemData pd;
  pd.r = r; pd.d = d; pd.n = n;
  int k = 0, nnz = 0;
  dcora_csr_info(Qh, &k, &nnz);
  pd.Q.n = k;
  pd.Q.rowptr.resize(k + 1);
  pd.Q.colidx.resize(nnz);
  pd.Q.vals.resize(nnz);
  dcora_csr_copy(Qh, pd.Q.rowptr.data(), pd.Q.colidx.data(), pd.Q.vals.data());
  dcora_csr_destroy(Qh);
  // prior rotation (already orthonormal to 1e-4; re-orthonormalised by two Gram-Schmidt passes)
  double P[9] = {0.7236, -0.6100, -0.3230, 0.1817, 0.6198, -0.7634, 0.6658, 0.4938, 0.5594};  // column-major
  for (int pass = 0; pass < 2; ++pass)
    for (int j = 0; j < 3; ++j) {
      for (int c = 0; c < j; ++c) {
        double s = 0;
        for (int i = 0; i < 3; ++i) s += P[3 * c + i] * P[3 * j + i];
        for (int i = 0; i < 3; ++i) P[3 * j + i] -= s * P[3 * c + i];
      }
      double nn = 0;
      for (int i = 0; i < 3; ++i) nn += P[3 * j + i] * P[3 * j + i];
      for (int i = 0; i < 3; ++i) P[3 * j + i] /= std::sqrt(nn);
    }
  pd.G = DCORA::Matrix(r, 4 * n);
  for (int c = 0; c < 3; ++c)
    for (int i = 0; i < 3; ++i) pd.G(i, 4 + c) = -P[3 * c + i] * 10000.0;  // G = -P Omega (ref src/Graph.cpp:805-816)
  DCORA::QuadraticProblem problem(pd);
  DCORA::Matrix T(r, 4 * n);
  for (int i = 0; i < 3; ++i) T(i, i) = T(i, 4 + i) = 1.0;
  DCORA::ROptParameters params;
  params.RTR_iterations = 50;
  params.RTR_tCG_iterations = 500;
  params.gradnorm_tol = 1e-5;
  DCORA::QuadraticOptimizer optimizer(&problem, params);
  DCORA::Matrix Topt = optimizer.optimize(T);
  double e0 = 0, e1 = 0;
  for (int c = 0; c < 3; ++c)
    for (int i = 0; i < 3; ++i) {
      e0 += std::pow(Topt(i, c) - P[3 * c + i], 2);
      e1 += std::pow(Topt(i, 4 + c) - P[3 * c + i], 2);
    }
  const DCORA::ROPTResult res = optimizer.getOptResult();
  std::printf("testPrior facade: err0 %.3e err1 %.3e f %.6f -> %.6f |g| %.2e\n", std::sqrt(e0), std::sqrt(e1), res.fInit,
              res.fOpt, res.gradNormOpt);
  bool ok = std::sqrt(e0) < 1e-6 && std::sqrt(e1) < 1e-6;
  // rounding through the facade (ref src/DCORA_utils.cpp:2262-2289): in the frame of pose 0 both poses are the
  // identity rotation (the edge says R = I) and the relative translation is zero
  DCORA::Matrix anchor(r, 4);
  for (int c = 0; c < 4; ++c)
    for (int i = 0; i < r; ++i) anchor(i, c) = Topt(i, c);
  const DCORA::Matrix Tr = DCORA::alignLiftedTrajectoryToFrame(Topt, anchor, d, n, true);
  double e2 = 0;
  for (int p = 0; p < n; ++p)
    for (int c = 0; c < 4; ++c)
      for (int i = 0; i < 3; ++i) e2 += std::pow(Tr(i, 4 * p + c) - ((c < 3 && i == c) ? 1.0 : 0.0), 2);
  // certificate of the (noiseless) two-pose problem without the prior: S = Q - Lambda is PSD at the optimum
  DCORA::Matrix Tid(r, 4 * n);
  for (int i = 0; i < 3; ++i) Tid(i, i) = Tid(i, 4 + i) = 1.0;
  const DCORA::SparseMatrix S = DCORA::constructDualCertificateMatrixPGO(Tid, pd.Q, d, n);
  double theta = 0;
  DCORA::Vector v;
  const bool psd = DCORA::fastVerification(S, 1e-3, &theta, &v, d + 1);
  std::printf("rounding err %.3e, certificate PSD %d\n", std::sqrt(e2), (int)psd);
  ok = ok && std::sqrt(e2) < 1e-6 && psd;
  // setProblem (ref include/DCORA/QuadraticOptimizer.h:52): the same optimizer pointed at a second problem object on
  // the same data gives the same answer
  DCORA::QuadraticProblem problem2(pd);
  optimizer.setProblem(&problem2);
  const DCORA::Matrix Topt2 = optimizer.optimize(T);
  double e3 = 0;
  for (int c = 0; c < 4 * n; ++c)
    for (int i = 0; i < r; ++i) e3 += std::pow(Topt2(i, c) - Topt(i, c), 2);
  std::printf("setProblem: difference %.3e\n", std::sqrt(e3));
  ok = ok && e3 == 0.0;
  return ok ? 0 : 1;
}
