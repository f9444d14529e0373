// ORACLE -- test infrastructure only (see oracle.hpp).
// Manifold arithmetic, QuadraticProblem operations and the RTR/tCG local
// solver (ref: src/QuadraticProblem.cpp, src/QuadraticOptimizer.cpp,
// src/manifold/LiftedManifold.cpp; ROPTLIB behaviour per SURVEY.md 3.4).
#include <algorithm>
#include <chrono>

#include "oracle.hpp"

namespace orc {

// ---------------------------------------------------------------------------
// Tangent projection.  Stiefel: V - Y sym(Y^T V)  (ref: src/DCORA_utils.cpp:
// 1695-1711 symBlockDiagProduct, :2033-2041); sphere: v - y (y^T v)
// (:2043-2051); Euclidean: identity.
// ---------------------------------------------------------------------------
static void stiefel_tangent(int r, int d, const double *Y, const double *V,
                            double *out) {
  double P[9], S[9];
  for (int a = 0; a < d; ++a)
    for (int b = 0; b < d; ++b) {
      double s = 0;
      for (int i = 0; i < r; ++i) s += Y[a * r + i] * V[b * r + i];
      P[a + b * d] = s;  // (Y^T V)(a,b)
    }
  for (int a = 0; a < d; ++a)
    for (int b = 0; b < d; ++b) S[a + b * d] = 0.5 * (P[a + b * d] + P[b + a * d]);
  for (int b = 0; b < d; ++b)
    for (int i = 0; i < r; ++i) {
      double s = 0;
      for (int a = 0; a < d; ++a) s += Y[a * r + i] * S[a + b * d];
      out[b * r + i] = V[b * r + i] - s;
    }
}

void tangent_project(const Dims &D, const Mat &X, const Mat &V, Mat &out) {
  const int r = D.r, d = D.d;
  if (out.rows != V.rows || out.cols != V.cols) out = Mat(V.rows, V.cols);
  if (&out != &V) out.a = V.a;
  std::vector<double> tmp((size_t)r * d);
  for (int i = 0; i < D.n; ++i) {
    const int c = D.rot_col(i);
    stiefel_tangent(r, d, X.col(c), V.col(c), tmp.data());
    std::copy(tmp.begin(), tmp.end(), out.col(c));
  }
  for (int i = 0; i < D.l; ++i) {
    const int c = D.sphere_col(i);
    const double *y = X.col(c);
    const double *v = V.col(c);
    double s = 0;
    for (int t = 0; t < r; ++t) s += y[t] * v[t];
    double *o = out.col(c);
    for (int t = 0; t < r; ++t) o[t] = v[t] - y[t] * s;
  }
}

// QF retraction of one Stiefel block: thin QR of Y+V by modified Gram-Schmidt
// with a second orthogonalisation pass; MGS gives R with positive diagonal,
// which is the sign convention of ROPTLIB's qf.
static void qf_block(int r, int d, const double *A, double *Qo) {
  for (int j = 0; j < d; ++j) {
    double *q = Qo + j * r;
    for (int i = 0; i < r; ++i) q[i] = A[j * r + i];
    for (int pass = 0; pass < 2; ++pass)
      for (int c = 0; c < j; ++c) {
        const double *qc = Qo + c * r;
        double s = 0;
        for (int i = 0; i < r; ++i) s += qc[i] * q[i];
        for (int i = 0; i < r; ++i) q[i] -= s * qc[i];
      }
    double nn = 0;
    for (int i = 0; i < r; ++i) nn += q[i] * q[i];
    nn = std::sqrt(nn);
    for (int i = 0; i < r; ++i) q[i] /= nn;
  }
}

void retract(const Dims &D, const Mat &X, const Mat &V, Mat &out) {
  const int r = D.r, d = D.d;
  Mat W(X.rows, X.cols);
  for (size_t i = 0; i < W.a.size(); ++i) W.a[i] = X.a[i] + V.a[i];
  std::vector<double> tmp((size_t)r * d);
  for (int i = 0; i < D.n; ++i) {
    const int c = D.rot_col(i);
    qf_block(r, d, W.col(c), tmp.data());
    std::copy(tmp.begin(), tmp.end(), W.col(c));
  }
  for (int i = 0; i < D.l; ++i) {
    double *w = W.col(D.sphere_col(i));
    double nn = 0;
    for (int t = 0; t < r; ++t) nn += w[t] * w[t];
    nn = std::sqrt(nn);
    for (int t = 0; t < r; ++t) w[t] /= nn;
  }
  out = std::move(W);
}

// Polar factor U V^T of an r x d block by one-sided (Hestenes) Jacobi SVD
// (ref: src/DCORA_utils.cpp:1677-1683 thin JacobiSVD -> U V^T).
void polar_factor(int r, int d, const double *M, double *out) {
  double A[16 * 3], Vm[9];
  for (int i = 0; i < r * d; ++i) A[i] = M[i];
  for (int a = 0; a < d; ++a)
    for (int b = 0; b < d; ++b) Vm[a + b * d] = (a == b);
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0;
    for (int p = 0; p < d - 1; ++p)
      for (int q = p + 1; q < d; ++q) {
        double app = 0, aqq = 0, apq = 0;
        for (int i = 0; i < r; ++i) {
          app += A[p * r + i] * A[p * r + i];
          aqq += A[q * r + i] * A[q * r + i];
          apq += A[p * r + i] * A[q * r + i];
        }
        if (std::fabs(apq) <= 1e-300 || std::fabs(apq) <= 1e-16 * std::sqrt(app * aqq)) continue;
        off = std::max(off, std::fabs(apq) / std::sqrt(app * aqq));
        const double zeta = (aqq - app) / (2.0 * apq);
        const double tt = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
        const double cs = 1.0 / std::sqrt(1.0 + tt * tt), sn = cs * tt;
        for (int i = 0; i < r; ++i) {
          const double x = A[p * r + i], y = A[q * r + i];
          A[p * r + i] = cs * x - sn * y;
          A[q * r + i] = sn * x + cs * y;
        }
        for (int i = 0; i < d; ++i) {
          const double x = Vm[p * d + i], y = Vm[q * d + i];
          Vm[p * d + i] = cs * x - sn * y;
          Vm[q * d + i] = sn * x + cs * y;
        }
      }
    if (off < 1e-15) break;
  }
  // A = U Sigma (columns), polar = U V^T
  for (int j = 0; j < d; ++j) {
    double nn = 0;
    for (int i = 0; i < r; ++i) nn += A[j * r + i] * A[j * r + i];
    nn = std::sqrt(nn);
    if (nn > 0)
      for (int i = 0; i < r; ++i) A[j * r + i] /= nn;
  }
  for (int c = 0; c < d; ++c)
    for (int i = 0; i < r; ++i) {
      double s = 0;
      for (int j = 0; j < d; ++j) s += A[j * r + i] * Vm[j * d + c];  // U(i,j) V(c,j)
      out[c * r + i] = s;
    }
}

static double det_small(int d, const double *M) {
  if (d == 2) return M[0] * M[3] - M[2] * M[1];
  return M[0] * (M[4] * M[8] - M[7] * M[5]) - M[3] * (M[1] * M[8] - M[7] * M[2]) +
         M[6] * (M[1] * M[5] - M[4] * M[2]);
}

// ref: src/DCORA_utils.cpp:1661-1675 (SVD, flip last column of U when
// det(U) det(V) < 0).  Equivalent closed form: polar factor P of M; if
// det(P) < 0 reflect along the least singular direction.
void project_to_rotation_group(int d, const double *M, double *out) {
  // one-sided Jacobi again, but keeping U, Sigma, V explicitly
  double A[9], Vm[9];
  for (int i = 0; i < d * d; ++i) A[i] = M[i];
  for (int a = 0; a < d; ++a)
    for (int b = 0; b < d; ++b) Vm[a + b * d] = (a == b);
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
        if (std::fabs(apq) <= 1e-300 || std::fabs(apq) <= 1e-16 * std::sqrt(app * aqq)) continue;
        off = std::max(off, std::fabs(apq) / std::sqrt(app * aqq));
        const double zeta = (aqq - app) / (2.0 * apq);
        const double tt = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
        const double cs = 1.0 / std::sqrt(1.0 + tt * tt), sn = cs * tt;
        for (int i = 0; i < d; ++i) {
          double x = A[p * d + i], y = A[q * d + i];
          A[p * d + i] = cs * x - sn * y;
          A[q * d + i] = sn * x + cs * y;
          x = Vm[p * d + i];
          y = Vm[q * d + i];
          Vm[p * d + i] = cs * x - sn * y;
          Vm[q * d + i] = sn * x + cs * y;
        }
      }
    if (off < 1e-15) break;
  }
  double sig[3];
  int jmin = 0;
  for (int j = 0; j < d; ++j) {
    double nn = 0;
    for (int i = 0; i < d; ++i) nn += A[j * d + i] * A[j * d + i];
    sig[j] = std::sqrt(nn);
    if (sig[j] < sig[jmin]) jmin = j;
    if (sig[j] > 0)
      for (int i = 0; i < d; ++i) A[j * d + i] /= sig[j];
  }
  // (rank-deficient input: complete U by Gram-Schmidt against e_i)
  for (int j = 0; j < d; ++j)
    if (!(sig[j] > 0)) {
      for (int e = 0; e < d; ++e) {
        double v[3] = {0, 0, 0};
        v[e] = 1;
        for (int c = 0; c < d; ++c)
          if (c != j && (sig[c] > 0 || c < j)) {
            double s = 0;
            for (int i = 0; i < d; ++i) s += A[c * d + i] * v[i];
            for (int i = 0; i < d; ++i) v[i] -= s * A[c * d + i];
          }
        double nn = 0;
        for (int i = 0; i < d; ++i) nn += v[i] * v[i];
        if (nn > 1e-8) {
          nn = std::sqrt(nn);
          for (int i = 0; i < d; ++i) A[j * d + i] = v[i] / nn;
          break;
        }
      }
    }
  if (det_small(d, A) * det_small(d, Vm) < 0)
    for (int i = 0; i < d; ++i) A[jmin * d + i] = -A[jmin * d + i];
  for (int c = 0; c < d; ++c)
    for (int i = 0; i < d; ++i) {
      double s = 0;
      for (int j = 0; j < d; ++j) s += A[j * d + i] * Vm[j * d + c];
      out[c * d + i] = s;
    }
}

void project_to_manifold(const Dims &D, const Mat &M, Mat &out) {
  const int r = D.r, d = D.d;
  Mat W = M;
  std::vector<double> tmp((size_t)r * d);
  for (int i = 0; i < D.n; ++i) {
    const int c = D.rot_col(i);
    polar_factor(r, d, M.col(c), tmp.data());
    std::copy(tmp.begin(), tmp.end(), W.col(c));
  }
  for (int i = 0; i < D.l; ++i) {
    double *w = W.col(D.sphere_col(i));
    double nn = 0;
    for (int t = 0; t < r; ++t) nn += w[t] * w[t];
    nn = std::sqrt(nn);
    for (int t = 0; t < r; ++t) w[t] /= nn;
  }
  out = std::move(W);
}

// ---------------------------------------------------------------------------
// QuadraticProblem
// ---------------------------------------------------------------------------
double Problem::f(const Mat &X) const {
  Mat XQ;
  spmm_right(X, *Q, XQ);
  double s = 0.5 * dot(XQ, X);
  if (G) s += dot(X, *G);
  return s;
}
void Problem::egrad(const Mat &X, Mat &EG) const {
  spmm_right(X, *Q, EG);
  if (G) axpy(1.0, *G, EG);
}
void Problem::rgrad(const Mat &X, Mat &RG) const {
  Mat EG;
  egrad(X, EG);
  tangent_project(D, X, EG, RG);
}
double Problem::rgradnorm(const Mat &X) const {
  Mat RG;
  rgrad(X, RG);
  return norm(RG);
}
// Hess f(X)[V] = Proj_X( V Q - V_i sym(Y_i^T EG_i) ) on Stiefel blocks,
// Proj( V Q - v (y^T eg) ) on spheres (ROPTLIB Stiefel::EucHvToHv, Euclidean
// metric; EucHessianEta = V Q, ref: src/QuadraticProblem.cpp:61-68).
void Problem::hess(const Mat &X, const Mat &EG, const Mat &V, Mat &HV) const {
  const int r = D.r, d = D.d;
  Mat W;
  spmm_right(V, *Q, W);
  for (int i = 0; i < D.n; ++i) {
    const int c = D.rot_col(i);
    const double *Y = X.col(c), *E = EG.col(c), *Vi = V.col(c);
    double P[9], S[9];
    for (int a = 0; a < d; ++a)
      for (int b = 0; b < d; ++b) {
        double s = 0;
        for (int t = 0; t < r; ++t) s += Y[a * r + t] * E[b * r + t];
        P[a + b * d] = s;
      }
    for (int a = 0; a < d; ++a)
      for (int b = 0; b < d; ++b) S[a + b * d] = 0.5 * (P[a + b * d] + P[b + a * d]);
    double *Wc = W.col(c);
    for (int b = 0; b < d; ++b)
      for (int t = 0; t < r; ++t) {
        double s = 0;
        for (int a = 0; a < d; ++a) s += Vi[a * r + t] * S[a + b * d];
        Wc[b * r + t] -= s;
      }
  }
  for (int i = 0; i < D.l; ++i) {
    const int c = D.sphere_col(i);
    const double *y = X.col(c), *e = EG.col(c), *v = V.col(c);
    double s = 0;
    for (int t = 0; t < r; ++t) s += y[t] * e[t];
    double *w = W.col(c);
    for (int t = 0; t < r; ++t) w[t] -= v[t] * s;
  }
  tangent_project(D, X, W, HV);
}
void Problem::precondition(const Mat &X, const Mat &V, Mat &Z) const {
  Mat T;
  precon->solve_rows(V, T);
  tangent_project(D, X, T, Z);
}

// ---------------------------------------------------------------------------
// Truncated CG (Steihaug-Toint, preconditioned) -- ROPTLIB SolversTR::tCG_TR
// restated: theta = 1, kappa = 0.1, eta0 = 0.
// ---------------------------------------------------------------------------
struct TCGOut { int status; int iters; };
static TCGOut tcg(const Problem &P, const Mat &X, const Mat &EG, const Mat &grad,
                  double Delta, int max_inner, Mat &eta, Mat &Heta) {
  const double theta = 1.0, kappa = 0.1;
  Mat r = grad, z, delta, Hd;
  eta = Mat(grad.rows, grad.cols);
  Heta = Mat(grad.rows, grad.cols);
  double e_Pe = 0;
  double r_r = dot(r, r);
  const double norm_r0 = std::sqrt(r_r);
  P.precondition(X, r, z);
  double z_r = dot(z, r);
  double d_Pd = z_r;
  delta = z;
  for (auto &x : delta.a) x = -x;
  double e_Pd = 0;
  TCGOut out{4, 0};
  int j = 0;
  for (; j < max_inner; ++j) {
    P.hess(X, EG, delta, Hd);
    const double d_Hd = dot(delta, Hd);
    const double alpha = z_r / d_Hd;
    const double e_Pe_new = e_Pe + 2.0 * alpha * e_Pd + alpha * alpha * d_Pd;
    if (d_Hd <= 0 || e_Pe_new >= Delta * Delta) {
      const double tau = (-e_Pd + std::sqrt(e_Pd * e_Pd + d_Pd * (Delta * Delta - e_Pe))) / d_Pd;
      axpy(tau, delta, eta);
      axpy(tau, Hd, Heta);
      out.status = (d_Hd <= 0) ? 0 : 1;
      ++j;
      break;
    }
    e_Pe = e_Pe_new;
    axpy(alpha, delta, eta);
    axpy(alpha, Hd, Heta);
    axpy(alpha, Hd, r);
    r_r = dot(r, r);
    const double norm_r = std::sqrt(r_r);
    const double tempnum = std::pow(norm_r0, theta);
    if (norm_r <= norm_r0 * std::min(tempnum, kappa)) {
      out.status = (kappa < tempnum) ? 2 : 3;
      ++j;
      break;
    }
    P.precondition(X, r, z);
    const double zold_rold = z_r;
    z_r = dot(z, r);
    const double beta = z_r / zold_rold;
    for (size_t i = 0; i < delta.a.size(); ++i) delta.a[i] = -z.a[i] + beta * delta.a[i];
    e_Pd = beta * (e_Pd + alpha * d_Pd);
    d_Pd = z_r + beta * beta * d_Pd;
  }
  out.iters = j;
  return out;
}

// RTRNewton::Run restated: accept iff rho > 0.1; rho < 0.25
// => Delta /= 4; rho > 0.75 and tCG stopped at the boundary / negative
// curvature => Delta = min(2 Delta, maximum_Delta).
struct RTROut { bool last_accepted; int iters, inner, accepted, tcg_status; };
static RTROut rtr_run(const Problem &P, Mat &X, double initial_Delta, double maximum_Delta,
                      int max_iter, int max_inner, double tol, double time_bound) {
  using clk = std::chrono::steady_clock;
  const auto t0 = clk::now();
  RTROut out{false, 0, 0, 0, 4};
  Mat EG, grad, eta, Heta, X2;
  double f1 = P.f(X);
  P.egrad(X, EG);
  tangent_project(P.D, X, EG, grad);
  double ngf = norm(grad);
  double Delta = initial_Delta;
  int iter = 0;
  while (ngf >= tol && iter < max_iter) {
    if (std::chrono::duration<double>(clk::now() - t0).count() > time_bound) break;
    TCGOut tc = tcg(P, X, EG, grad, Delta, max_inner, eta, Heta);
    out.inner += tc.iters;
    out.tcg_status = tc.status;
    retract(P.D, X, eta, X2);
    const double f2 = P.f(X2);
    // rho = (f1 - f2) / ( -<eta, grad + 0.5 H eta> )
    const double denom = -(dot(eta, grad) + 0.5 * dot(eta, Heta));
    const double rho = (f1 - f2) / denom;
    if (rho > 0.75) {
      if (tc.status == 0 || tc.status == 1) Delta = std::min(2.0 * Delta, maximum_Delta);
    } else if (rho < 0.25) {
      Delta *= 0.25;
    }
    if (rho > 0.1 && std::isfinite(rho)) {
      X = X2;
      f1 = f2;
      P.egrad(X, EG);
      tangent_project(P.D, X, EG, grad);
      ngf = norm(grad);
      out.last_accepted = true;
      out.accepted++;
    } else {
      out.last_accepted = false;
    }
    ++iter;
  }
  out.iters = iter;
  return out;
}

// ref: src/QuadraticOptimizer.cpp:234-280
static bool trust_region(const Problem &P, const ROptParams &prm, Mat &X, ROptResult *res) {
  const double tb = 5.0;
  if (prm.RTR_iterations == 1) {
    double radius = prm.RTR_initial_radius;
    int total_steps = 0;
    while (true) {
      Mat Xtry = X;
      RTROut o = rtr_run(P, Xtry, radius, radius, 1, prm.RTR_tCG_iterations, prm.gradnorm_tol, tb);
      res->outer_iters += o.iters;
      res->inner_iters += o.inner;
      res->tcg_status = o.tcg_status;
      if (o.last_accepted) {
        X = Xtry;
        res->accepted += 1;
        return true;
      } else if (total_steps > 10) {
        return false;
      }
      radius /= 4;
      total_steps++;
    }
  }
  RTROut o = rtr_run(P, X, prm.RTR_initial_radius, 5 * prm.RTR_initial_radius, prm.RTR_iterations,
                     prm.RTR_tCG_iterations, prm.gradnorm_tol, tb);
  res->outer_iters = o.iters;
  res->inner_iters = o.inner;
  res->accepted = o.accepted;
  res->tcg_status = o.tcg_status;
  return true;
}

// ref: src/QuadraticOptimizer.cpp:123-150
static Mat gradient_descent(const Problem &P, const ROptParams &prm, const Mat &Y) {
  Mat RG;
  P.rgrad(Y, RG);
  if (prm.RGD_use_preconditioner) {
    Mat Z;
    P.precondition(Y, RG, Z);
    RG = Z;
  }
  for (auto &x : RG.a) x *= -prm.RGD_stepsize;
  Mat out;
  retract(P.D, Y, RG, out);
  return out;
}

Mat optimize(const Problem &P, const ROptParams &prm, const Mat &Y0, ROptResult *res) {
  ROptResult local;
  if (!res) res = &local;
  *res = ROptResult();
  res->fInit = P.f(Y0);
  res->gradNormInit = P.rgradnorm(Y0);
  const auto t0 = std::chrono::steady_clock::now();
  Mat Y = Y0;
  if (prm.method == 0) {
    // ref: src/QuadraticOptimizer.cpp:54-55 early return
    if (!(P.rgradnorm(Y0) < prm.gradnorm_tol)) {
      Mat X = Y0;
      if (trust_region(P, prm, X, res)) Y = X;
    }
  } else {
    Y = gradient_descent(P, prm, Y0);
  }
  res->elapsedMs = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  res->fOpt = P.f(Y);
  res->gradNormOpt = P.rgradnorm(Y);
  res->success = 1;
  return Y;
}

}  // namespace orc
