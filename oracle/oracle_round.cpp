// ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle.hpp).
// Rounding / solution recovery: lifted iterate (r x k) -> poses in SE(d).
#include <algorithm>
#include <cmath>

#include "oracle.hpp"

namespace orc {

// ref: src/DCORA_utils.cpp:2262-2289 alignLiftedTrajectoryToFrame (global alignment: the anchor's translation
// fixes the origin) and src/Agent.cpp:963-980 (local frame: the first pose of the trajectory itself becomes the
// origin).  X is r x (d+1) n in the SE ordering, Tw0 = [Y0 p0] is r x (d+1).
void align_lifted_trajectory_to_frame(const Mat &X, const Mat &Tw0, int d, int n, bool global, Mat &out) {
  const int r = X.rows, dh = d + 1;
  out = Mat(d, dh * n);
  // R0^T X
  for (int c = 0; c < dh * n; ++c)
    for (int a = 0; a < d; ++a) {
      double s = 0;
      for (int i = 0; i < r; ++i) s += Tw0(i, a) * X(i, c);
      out(a, c) = s;
    }
  double t0[3] = {0, 0, 0};
  if (global) {
    for (int a = 0; a < d; ++a) {
      double s = 0;
      for (int i = 0; i < r; ++i) s += Tw0(i, a) * Tw0(i, d);
      t0[a] = s;
    }
  } else {
    for (int a = 0; a < d; ++a) t0[a] = out(a, d);
  }
  for (int i = 0; i < n; ++i) {
    double blk[9], prj[9];
    for (int c = 0; c < d; ++c)
      for (int a = 0; a < d; ++a) blk[c * d + a] = out(a, i * dh + c);
    project_to_rotation_group(d, blk, prj);
    for (int c = 0; c < d; ++c)
      for (int a = 0; a < d; ++a) out(a, i * dh + c) = prj[c * d + a];
    for (int a = 0; a < d; ++a) out(a, i * dh + d) -= t0[a];
  }
}

// ref: src/Agent.cpp:950-1003 getStatesInLocalFrame.  X is r x k in the RA ordering; the trajectory comes back
// d x (d+1) n in the SE ordering (PoseArray), unit spheres d x l (rotated only), landmarks d x b (rotated and
// translated into the frame of pose 0).
void ra_states_in_local_frame(const Mat &X, const Dims &dm, Mat &traj, Mat &spheres, Mat &landmarks) {
  const int r = dm.r, d = dm.d, n = dm.n, l = dm.l, b = dm.b, dh = d + 1;
  Mat Xse(r, dh * n), Tw0(r, dh);
  for (int i = 0; i < n; ++i) {
    for (int c = 0; c < d; ++c)
      for (int a = 0; a < r; ++a) Xse(a, i * dh + c) = X(a, dm.rot_col(i) + c);
    for (int a = 0; a < r; ++a) Xse(a, i * dh + d) = X(a, d * n + l + i);
  }
  for (int c = 0; c < dh; ++c)
    for (int a = 0; a < r; ++a) Tw0(a, c) = Xse(a, c);
  // t0 before the trajectory is re-centred
  double t0[3] = {0, 0, 0};
  for (int a = 0; a < d; ++a) {
    double s = 0;
    for (int i = 0; i < r; ++i) s += Tw0(i, a) * Tw0(i, d);
    t0[a] = s;
  }
  align_lifted_trajectory_to_frame(Xse, Tw0, d, n, false, traj);
  spheres = Mat(d, l);
  for (int j = 0; j < l; ++j)
    for (int a = 0; a < d; ++a) {
      double s = 0;
      for (int i = 0; i < r; ++i) s += Tw0(i, a) * X(i, d * n + j);
      spheres(a, j) = s;
    }
  landmarks = Mat(d, b);
  for (int j = 0; j < b; ++j)
    for (int a = 0; a < d; ++a) {
      double s = 0;
      for (int i = 0; i < r; ++i) s += Tw0(i, a) * X(i, d * n + l + n + j);
      landmarks(a, j) = s - t0[a];
    }
}

// ref: src/DCORA_utils.cpp:1984-2031 projectSolutionRASLAM: rank-d truncation (U_d S_d)^T of the thin SVD of X^T,
// reflection when fewer than half of the rotation blocks have positive determinant, then SO(d) / unit-sphere
// projection block by block.  The SVD is a one-sided Jacobi on the r columns of X^T.
void project_solution_raslam(const Mat &X, const Dims &dm, Mat &out) {
  const int r = dm.r, d = dm.d, n = dm.n, l = dm.l, k = dm.k();
  // A = X^T (k x r), stored by columns: A(:, j) = row j of X
  std::vector<double> A((size_t)k * r);
  for (int j = 0; j < r; ++j)
    for (int c = 0; c < k; ++c) A[(size_t)j * k + c] = X(j, c);
  for (int sweep = 0; sweep < 100; ++sweep) {
    double off = 0;
    for (int p = 0; p < r - 1; ++p)
      for (int q = p + 1; q < r; ++q) {
        double app = 0, aqq = 0, apq = 0;
        const double *ap = &A[(size_t)p * k], *aq = &A[(size_t)q * k];
        for (int i = 0; i < k; ++i) {
          app += ap[i] * ap[i];
          aqq += aq[i] * aq[i];
          apq += ap[i] * aq[i];
        }
        if (std::fabs(apq) <= 1e-300 || std::fabs(apq) <= 1e-16 * std::sqrt(app * aqq)) continue;
        off = std::max(off, std::fabs(apq) / std::sqrt(app * aqq));
        const double zeta = (aqq - app) / (2.0 * apq);
        const double tt = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
        const double cs = 1.0 / std::sqrt(1.0 + tt * tt), sn = cs * tt;
        double *bp = &A[(size_t)p * k], *bq = &A[(size_t)q * k];
        for (int i = 0; i < k; ++i) {
          const double x = bp[i], y = bq[i];
          bp[i] = cs * x - sn * y;
          bq[i] = sn * x + cs * y;
        }
      }
    if (off < 1e-15) break;
  }
  // columns of A are now U_j sigma_j; keep the d largest
  std::vector<std::pair<double, int>> sig((size_t)r);
  for (int j = 0; j < r; ++j) {
    double s = 0;
    for (int i = 0; i < k; ++i) s += A[(size_t)j * k + i] * A[(size_t)j * k + i];
    sig[j] = {std::sqrt(s), j};
  }
  std::sort(sig.begin(), sig.end(), [](const auto &a, const auto &b) { return a.first > b.first; });
  out = Mat(d, k);
  for (int a = 0; a < d; ++a)
    for (int c = 0; c < k; ++c) out(a, c) = A[(size_t)sig[a].second * k + c];
  auto det_block = [&](int i) {
    const int c0 = dm.rot_col(i);
    if (d == 2) return out(0, c0) * out(1, c0 + 1) - out(0, c0 + 1) * out(1, c0);
    return out(0, c0) * (out(1, c0 + 1) * out(2, c0 + 2) - out(1, c0 + 2) * out(2, c0 + 1)) -
           out(0, c0 + 1) * (out(1, c0) * out(2, c0 + 2) - out(1, c0 + 2) * out(2, c0)) +
           out(0, c0 + 2) * (out(1, c0) * out(2, c0 + 1) - out(1, c0 + 1) * out(2, c0));
  };
  int npos = 0;
  for (int i = 0; i < n; ++i)
    if (det_block(i) > 0) ++npos;
  if (npos < n / 2)
    for (int c = 0; c < k; ++c) out(d - 1, c) = -out(d - 1, c);
  for (int i = 0; i < n; ++i) {
    const int c0 = dm.rot_col(i);
    double blk[9], prj[9];
    for (int c = 0; c < d; ++c)
      for (int a = 0; a < d; ++a) blk[c * d + a] = out(a, c0 + c);
    project_to_rotation_group(d, blk, prj);
    for (int c = 0; c < d; ++c)
      for (int a = 0; a < d; ++a) out(a, c0 + c) = prj[c * d + a];
  }
  for (int j = 0; j < l; ++j) {
    const int c = d * n + j;
    double s = 0;
    for (int a = 0; a < d; ++a) s += out(a, c) * out(a, c);
    s = std::sqrt(s);
    if (s > 0)
      for (int a = 0; a < d; ++a) out(a, c) /= s;
  }
}

}  // namespace orc
