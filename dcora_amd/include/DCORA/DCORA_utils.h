// Host-side mirror of the free functions of the reference that sit next to the hot path and are served by
// libdcora_hip (ref include/DCORA/DCORA_utils.h): certification, rounding, initialisation.  Same names and
// argument meaning as the reference; every call goes through the C ABI of include/dcora_hip.h.
#pragma once
#include <algorithm>

#include "DCORA_types.h"

namespace DCORA {

namespace detail {
inline void check(int status, const char *what) {
  if (status != DCORA_OK) throw std::runtime_error(std::string(what) + ": " + dcora_last_error());
}
inline SparseMatrix take(dcora_csr_t h) {
  SparseMatrix S;
  int nnz = 0;
  check(dcora_csr_info(h, &S.n, &nnz), "csr_info");
  S.rowptr.resize((size_t)S.n + 1);
  S.colidx.resize((size_t)std::max(nnz, 1));
  S.vals.resize((size_t)std::max(nnz, 1));
  check(dcora_csr_copy(h, S.rowptr.data(), S.colidx.data(), S.vals.data()), "csr_copy");
  S.colidx.resize((size_t)nnz);
  S.vals.resize((size_t)nnz);
  dcora_csr_destroy(h);
  return S;
}
}  // namespace detail

// ref include/DCORA/Measurements.h:765-777, src/DCORA_utils.cpp:179-375: the pose-pose measurements of a g2o file in
// global pose numbering (r1 = r2 = 0), through the library's reader
struct G2ODataset {
  unsigned dim = 0, num_poses = 0;
  std::vector<RelativePosePoseMeasurement> pose_pose_measurements;
};
inline G2ODataset read_g2o_file(const std::string &filename) {
  dcora_dataset_t ds = nullptr;
  detail::check(dcora_dataset_load_g2o(filename.c_str(), &ds), "read_g2o_file");
  int d = 0, n = 0, m = 0;
  dcora_dataset_info(ds, &d, &n, &m);
  const size_t w = (size_t)d * d + d + 3;
  std::vector<int> ids((size_t)4 * m);
  std::vector<double> vals((size_t)m * w);
  detail::check(dcora_dataset_copy(ds, ids.data(), vals.data()), "read_g2o_file");
  dcora_dataset_destroy(ds);
  G2ODataset out;
  out.dim = (unsigned)d;
  out.num_poses = (unsigned)n;
  out.pose_pose_measurements.reserve((size_t)m);
  for (int i = 0; i < m; ++i) {
    const double *v = &vals[(size_t)i * w];
    Matrix R((size_t)d, (size_t)d);
    for (int c = 0; c < d; ++c)
      for (int a = 0; a < d; ++a) R((size_t)a, (size_t)c) = v[c * d + a];
    Vector t(v + d * d, v + d * d + d);
    out.pose_pose_measurements.emplace_back((size_t)ids[4 * (size_t)i], (size_t)ids[4 * (size_t)i + 2],
                                            (size_t)ids[4 * (size_t)i + 1], (size_t)ids[4 * (size_t)i + 3], R, t,
                                            v[d * d + d], v[d * d + d + 1], false, v[d * d + d + 2]);
  }
  return out;
}

// ref src/DCORA_utils.cpp:1898-1931 / 1933-1982
inline SparseMatrix constructDualCertificateMatrixPGO(const Matrix &X, const SparseMatrix &Q, unsigned d, unsigned n) {
  dcora_dims dims{(int)X.rows(), (int)d, (int)n, 0, 0, DCORA_LAYOUT_SE};
  dcora_csr_t h = nullptr;
  detail::check(dcora_cert_dual_matrix(&dims, X.data(), Q.rowptr.data(), Q.colidx.data(), Q.vals.data(), 0, &h),
                "constructDualCertificateMatrixPGO");
  return detail::take(h);
}
inline SparseMatrix constructDualCertificateMatrixRASLAM(const Matrix &X, const SparseMatrix &Q, unsigned d, unsigned n,
                                                         unsigned l, unsigned b) {
  dcora_dims dims{(int)X.rows(), (int)d, (int)n, (int)l, (int)b, DCORA_LAYOUT_RA};
  dcora_csr_t h = nullptr;
  detail::check(dcora_cert_dual_matrix(&dims, X.data(), Q.rowptr.data(), Q.colidx.data(), Q.vals.data(), 0, &h),
                "constructDualCertificateMatrixRASLAM");
  return detail::take(h);
}

// ref src/DCORA_utils.cpp:1713-1735: true when S + eta I is positive definite; otherwise theta / min_eigenvector
// hold the minimum eigenpair
inline bool fastVerification(const SparseMatrix &S, double eta, double *theta, Vector *min_eigenvector,
                             int block = 1) {
  int psd = 0;
  double th = 0, lmin = 0;
  Vector x((size_t)S.n, 0.0);
  detail::check(dcora_cert_fast_verification(S.n, S.rowptr.data(), S.colidx.data(), S.vals.data(), eta, block, 0, &psd,
                                             &th, x.data(), &lmin),
                "fastVerification");
  if (theta) *theta = th;
  if (min_eigenvector) *min_eigenvector = x;
  return psd != 0;
}

// ref src/DCORA_utils.cpp:2262-2289.  Tw0 is the lifted anchor pose [Y0 p0] (r x (d+1)).
inline Matrix alignLiftedTrajectoryToFrame(const Matrix &liftedTrajectoryInit, const Matrix &Tw0, unsigned d,
                                           unsigned n, bool isGlobalAlignment) {
  dcora_dims dims{(int)liftedTrajectoryInit.rows(), (int)d, (int)n, 0, 0, DCORA_LAYOUT_SE};
  Matrix T(d, (size_t)(d + 1) * n);
  detail::check(dcora_round_align_trajectory(&dims, liftedTrajectoryInit.data(), Tw0.data(), isGlobalAlignment ? 1 : 0,
                                             T.data(), nullptr, nullptr, 0),
                "alignLiftedTrajectoryToFrame");
  return T;
}

// ref src/DCORA_utils.cpp:1984-2031
inline Matrix projectSolutionRASLAM(const Matrix &X, unsigned r, unsigned d, unsigned n, unsigned l, unsigned b) {
  dcora_dims dims{(int)r, (int)d, (int)n, (int)l, (int)b, DCORA_LAYOUT_RA};
  Matrix P(d, (size_t)(d + 1) * n + l + b);
  detail::check(dcora_round_project_solution_raslam(&dims, X.data(), P.data(), 0), "projectSolutionRASLAM");
  return P;
}

}  // namespace DCORA
