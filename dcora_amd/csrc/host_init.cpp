// Chordal initialisation (ref src/DCORA_solver.cpp:218-268; B matrices src/DCORA_utils.cpp:1542-1630;
// recoverTranslations :1632-1659).  Setup-time host code of the product: the start point of the RBCD loop for the
// reference driver's InitializationMethod::Chordal (ref examples/MultiRobotExample.cpp:150-153).
//
// The reference solves two sparse linear least-squares problems with SPQR,
//     min_R sum_e kappa_e |R_j - R_i R_ij|_F^2   with R_0 = I,       min_t sum_e tau_e |t_j - t_i - R_i t_ij|^2   with t_0 = 0;
// here their normal equations are solved block-wise: the rotation system is the reduced rotation connection Laplacian
// (d x d blocks, d right-hand sides), the translation system the reduced weighted graph Laplacian.
#include <cmath>

#include "host_graph.h"

namespace dcora {

namespace {
double det_small(int d, const double *M) {
  if (d == 2) return M[0] * M[3] - M[2] * M[1];
  return M[0] * (M[4] * M[8] - M[7] * M[5]) - M[3] * (M[1] * M[8] - M[7] * M[2]) + M[6] * (M[1] * M[5] - M[4] * M[2]);
}
// projectToRotationGroup (ref src/DCORA_utils.cpp:1661-1675): U V^T with the last column of U flipped when
// det(U) det(V) < 0; SVD by two-sided use of the symmetric eigen-decomposition of M^T M (d <= 3, well-conditioned input)
void project_so(int d, const double *M, double *out) {
  // one-sided Jacobi on the columns of M
  double A[9], V[9];
  for (int i = 0; i < d * d; ++i) A[i] = M[i];
  for (int a = 0; a < d; ++a)
    for (int b = 0; b < d; ++b) V[a + b * d] = (a == b);
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0;
    for (int p = 0; p < d - 1; ++p)
      for (int q = p + 1; q < d; ++q) {
        double app = 0, aqq = 0, apq = 0;
        for (int i = 0; i < d; ++i) {
          app += A[p * d + i] * A[p * d + i];
          aqq += A[q * d + i] * A[q * d + i];
          apq += A[p * d + i] * A[q * d + i];
        }
        const double sc = std::sqrt(app * aqq);
        if (!(std::fabs(apq) > 1e-16 * sc) || std::fabs(apq) < 1e-300) continue;
        off = std::fmax(off, std::fabs(apq) / sc);
        const double zeta = (aqq - app) / (2 * apq);
        const double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1 + zeta * zeta));
        const double c = 1 / std::sqrt(1 + t * t), s = c * t;
        for (int i = 0; i < d; ++i) {
          double x = A[p * d + i], y = A[q * d + i];
          A[p * d + i] = c * x - s * y;
          A[q * d + i] = s * x + c * y;
          x = V[p * d + i];
          y = V[q * d + i];
          V[p * d + i] = c * x - s * y;
          V[q * d + i] = s * x + c * y;
        }
      }
    if (off < 1e-15) break;
  }
  int jmin = 0;
  double smin = 1e300;
  for (int j = 0; j < d; ++j) {
    double nn = 0;
    for (int i = 0; i < d; ++i) nn += A[j * d + i] * A[j * d + i];
    nn = std::sqrt(nn);
    if (nn < smin) {
      smin = nn;
      jmin = j;
    }
    if (nn > 0)
      for (int i = 0; i < d; ++i) A[j * d + i] /= nn;
  }
  if (det_small(d, A) * det_small(d, V) < 0)
    for (int i = 0; i < d; ++i) A[jmin * d + i] = -A[jmin * d + i];
  for (int c = 0; c < d; ++c)
    for (int i = 0; i < d; ++i) {
      double s = 0;
      for (int j = 0; j < d; ++j) s += A[j * d + i] * V[j * d + c];
      out[c * d + i] = s;
    }
}
}  // namespace

// projectToRotationGroup for other host code (host_robust.cpp)
void project_to_rotation_group_host(int d, const double *M, double *out) { project_so(d, M, out); }

// T (d x (d+1) n, column-major, SE ordering); returns false when a reduced Laplacian is not positive definite
// (disconnected measurement graph)
namespace {
// host solve: nested-dissection Cholesky, right-hand sides in the permuted order
bool host_spd_solve(const HostCsr &A, int block, int nrhs, const double *B, double *X) {
  SparseChol ch;
  if (!ch.factor(A, block)) return false;
  const int m = A.n;
  std::vector<double> y((size_t)m * nrhs);
  const std::vector<int> &perm = ch.perm();
  for (int q = 0; q < m; ++q)
    for (int a = 0; a < nrhs; ++a) y[(size_t)q * nrhs + a] = B[(size_t)perm[q] * nrhs + a];
  ch.solve_inplace(y.data(), nrhs);
  for (int q = 0; q < m; ++q)
    for (int a = 0; a < nrhs; ++a) X[(size_t)perm[q] * nrhs + a] = y[(size_t)q * nrhs + a];
  return true;
}
}  // namespace

bool chordal_initialization(const HostDataset &ds, std::vector<double> &T, const SpdSolve &solve_in) {
  const SpdSolve solve = solve_in ? solve_in : SpdSolve(host_spd_solve);
  const int d = ds.d, n = ds.n, dh = d + 1;
  T.assign((size_t)d * dh * n, 0.0);
  for (int a = 0; a < d; ++a) T[(size_t)a * d + a] = 1.0;
  if (n == 1) return true;
  // ---- rotations ----
  std::vector<int> I, J;
  std::vector<double> V;
  const int m = d * (n - 1);
  std::vector<double> rhs((size_t)m * d, 0.0);  // rhs[(col) * d + row]: d right-hand sides, contiguous per unknown
  for (const PoseMeas &e : ds.meas) {
    const int i = e.p1, j = e.p2;
    const double k = e.kappa;
    for (int a = 0; a < d; ++a) {
      if (i > 0) { I.push_back((i - 1) * d + a); J.push_back((i - 1) * d + a); V.push_back(k); }
      if (j > 0) { I.push_back((j - 1) * d + a); J.push_back((j - 1) * d + a); V.push_back(k); }
      for (int b = 0; b < d; ++b) {
        const double v = -k * e.R[a + b * d];  // L(i*d + a, j*d + b)
        if (i > 0 && j > 0) {
          I.push_back((i - 1) * d + a); J.push_back((j - 1) * d + b); V.push_back(v);
          I.push_back((j - 1) * d + b); J.push_back((i - 1) * d + a); V.push_back(v);
        } else if (i == 0 && j > 0) {
          rhs[(size_t)((j - 1) * d + b) * d + a] -= v;  // -(R_0 L_{0,red}), R_0 = I
        } else if (j == 0 && i > 0) {
          rhs[(size_t)((i - 1) * d + a) * d + b] -= v;
        }
      }
    }
  }
  {
    HostCsr L = csr_from_coo(m, m, I, J, V);
    std::vector<double> sol((size_t)m * d);
    if (!solve(L, d, d, rhs.data(), sol.data())) return false;
    for (int i = 1; i < n; ++i) {
      double blk[9], out[9];
      for (int c = 0; c < d; ++c)
        for (int a = 0; a < d; ++a) blk[a + c * d] = sol[(size_t)((i - 1) * d + c) * d + a];
      project_so(d, blk, out);
      for (int c = 0; c < d; ++c)
        for (int a = 0; a < d; ++a) T[(size_t)(i * dh + c) * d + a] = out[a + c * d];
    }
  }
  // ---- translations ----
  I.clear();
  J.clear();
  V.clear();
  std::vector<double> b((size_t)(n - 1) * d, 0.0);
  for (const PoseMeas &e : ds.meas) {
    const int i = e.p1, j = e.p2;
    if (i > 0) { I.push_back(i - 1); J.push_back(i - 1); V.push_back(e.tau); }
    if (j > 0) { I.push_back(j - 1); J.push_back(j - 1); V.push_back(e.tau); }
    if (i > 0 && j > 0) {
      I.push_back(i - 1); J.push_back(j - 1); V.push_back(-e.tau);
      I.push_back(j - 1); J.push_back(i - 1); V.push_back(-e.tau);
    }
    for (int a = 0; a < d; ++a) {
      double rt = 0;
      for (int c = 0; c < d; ++c) rt += T[(size_t)(i * dh + c) * d + a] * e.t[c];
      if (j > 0) b[(size_t)(j - 1) * d + a] += e.tau * rt;
      if (i > 0) b[(size_t)(i - 1) * d + a] -= e.tau * rt;
    }
  }
  HostCsr Lt = csr_from_coo(n - 1, n - 1, I, J, V);
  std::vector<double> y((size_t)(n - 1) * d);
  if (!solve(Lt, 1, d, b.data(), y.data())) return false;
  for (int q = 0; q < n - 1; ++q)
    for (int a = 0; a < d; ++a) T[(size_t)((q + 1) * dh + d) * d + a] = y[(size_t)q * d + a];
  return true;
}

}  // namespace dcora

namespace dcora {
namespace {
inline unsigned long long splitmix64_next(unsigned long long &s) {
  unsigned long long z = (s += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
}  // namespace

// Start point of the centralised CORA driver (ref examples/SingleRobotExample_RASLAM.cpp:92-150, odometryInitialization
// ref src/DCORA_solver.cpp:270-302): odometry chains anchored at their ground-truth first pose, ground-truth unit
// spheres, landmarks uniform in (-1, 1) from a seeded splitmix64 stream.  X0 is d x k, RA ordering, column-major.
void ra_odometry_initialization(const HostRADataset &ds, unsigned long long seed, std::vector<double> &X0) {
  const int d = ds.d, n = ds.n, l = ds.l, b = ds.b, k = ds.k();
  X0.assign((size_t)d * k, 0.0);
  auto X = [&](int a, int c) -> double & { return X0[(size_t)c * d + a]; };
  auto G = [&](int a, int c) { return ds.gt[(size_t)c * d + a]; };
  std::vector<int> into((size_t)n, -1);
  for (int e = 0; e < (int)ds.pose_pose.size(); ++e) {
    const PoseMeas &m = ds.pose_pose[e];
    if (m.p1 + 1 == m.p2 && into[m.p2] < 0) into[m.p2] = e;
  }
  const int ot = d * n + l;
  for (int i = 0; i < n; ++i) {
    if (into[i] < 0) {
      for (int c = 0; c < d; ++c)
        for (int a = 0; a < d; ++a) X(a, d * i + c) = G(a, d * i + c);
      for (int a = 0; a < d; ++a) X(a, ot + i) = G(a, ot + i);
      continue;
    }
    const PoseMeas &m = ds.pose_pose[into[i]];
    for (int c = 0; c < d; ++c)
      for (int a = 0; a < d; ++a) {
        double s = 0;
        for (int q = 0; q < d; ++q) s += X(a, d * (i - 1) + q) * m.R[q + c * d];
        X(a, d * i + c) = s;
      }
    for (int a = 0; a < d; ++a) {
      double s = X(a, ot + i - 1);
      for (int q = 0; q < d; ++q) s += X(a, d * (i - 1) + q) * m.t[q];
      X(a, ot + i) = s;
    }
  }
  for (int j = 0; j < l; ++j)
    for (int a = 0; a < d; ++a) X(a, d * n + j) = G(a, d * n + j);
  unsigned long long s = seed;
  for (int j = 0; j < b; ++j)
    for (int a = 0; a < d; ++a) X(a, ot + n + j) = 2.0 * ((splitmix64_next(s) >> 11) * (1.0 / 9007199254740992.0)) - 1.0;
}
}  // namespace dcora
