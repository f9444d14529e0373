// Robust estimation, host side (see host_robust.h).
#include "host_robust.h"

#include <algorithm>
#include <cmath>

namespace dcora {

// ---- chi-square quantile: the reference calls boost::math::quantile(chi_squared(dof), q); here the regularised
// lower incomplete gamma function (series / Lentz continued fraction) inverted by bisection ----
namespace {
double gamma_p(double a, double x) {
  if (x <= 0) return 0.0;
  const double lg = std::lgamma(a);
  if (x < a + 1.0) {
    double ap = a, sum = 1.0 / a, del = sum;
    for (int n = 0; n < 1000; ++n) {
      ap += 1.0;
      del *= x / ap;
      sum += del;
      if (std::fabs(del) < std::fabs(sum) * 1e-16) break;
    }
    return sum * std::exp(-x + a * std::log(x) - lg);
  }
  double b = x + 1.0 - a, c = 1.0 / 1e-300, d = 1.0 / b, h = d;
  for (int i = 1; i < 1000; ++i) {
    const double an = -i * (i - a);
    b += 2.0;
    d = an * d + b;
    if (std::fabs(d) < 1e-300) d = 1e-300;
    c = b + an / c;
    if (std::fabs(c) < 1e-300) c = 1e-300;
    d = 1.0 / d;
    const double del = d * c;
    h *= del;
    if (std::fabs(del - 1.0) < 1e-16) break;
  }
  return 1.0 - std::exp(-x + a * std::log(x) - lg) * h;
}
}  // namespace

double chi2inv(double quantile, int dof) {
  const double a = 0.5 * dof;
  double lo = 0.0, hi = std::max(4.0 * dof, 10.0);
  while (gamma_p(a, 0.5 * hi) < quantile) hi *= 2.0;
  for (int it = 0; it < 200; ++it) {
    const double mid = 0.5 * (lo + hi);
    (gamma_p(a, 0.5 * mid) < quantile ? lo : hi) = mid;
    if (hi - lo < 1e-14 * std::max(1.0, hi)) break;
  }
  return 0.5 * (lo + hi);
}

bool error_threshold_at_quantile(double quantile, int dimension, double *out) {
  if (dimension != 3 || !(quantile > 0)) return false;  // CHECKs of the reference
  *out = quantile < 1 ? std::sqrt(chi2inv(quantile, 6)) : 1e5;
  return true;
}

double RobustCost::weight(double r) const {
  switch (p_.cost_type) {
    case DCORA_ROBUST_L2: return 1;
    case DCORA_ROBUST_L1: return 1 / r;
    case DCORA_ROBUST_HUBER: return r < p_.HuberThreshold ? 1 : p_.HuberThreshold / r;
    case DCORA_ROBUST_TLS: return r < p_.TLSThreshold ? 1 : 0;
    case DCORA_ROBUST_GM: {
      const double a = 1 + r * r;
      return 1 / (a * a);
    }
    case DCORA_ROBUST_GNC_TLS: {  // eq. (14) of the GNC paper
      const double rSq = r * r, bSq = p_.GNCBarc * p_.GNCBarc;
      const double ub = (mu_ + 1) / mu_ * bSq, lb = mu_ / (mu_ + 1) * bSq;
      if (rSq >= ub) return 0;
      if (rSq <= lb) return 1;
      return std::sqrt(bSq * mu_ * (mu_ + 1) / rSq) - mu_;
    }
  }
  return 1;
}
void RobustCost::reset() {
  if (p_.cost_type == DCORA_ROBUST_GNC_TLS) {
    mu_ = p_.GNCInitMu;
    iteration_ = 0;
  }
}
void RobustCost::update() {
  if (p_.cost_type != DCORA_ROBUST_GNC_TLS) return;
  iteration_++;
  if (iteration_ > p_.GNCMaxNumIters) return;
  mu_ = p_.GNCMuStep * mu_;
}

// ---- averaging ----
namespace {
double sqdist(int m, const double *a, const double *b) {
  double s = 0;
  for (int e = 0; e < m; ++e) s += (a[e] - b[e]) * (a[e] - b[e]);
  return s;
}
// singleRotationAveraging / singleTranslationAveraging (ref src/DCORA_solver.cpp:28-72) and the GNC loop shared by
// robustSingleRotationAveraging (:76-141) and robustSinglePoseAveraging (:143-216)
void robust_average(int d, int n, const double *R, const double *t, const double *kappa, const double *tau,
                    double barc, int max_iters, double *Ropt, double *topt, std::vector<int> &inliers) {
  const double w_tol = 1e-8;
  std::vector<double> w((size_t)n, 1.0);
  auto solve = [&]() {
    double M[9] = {0}, s[3] = {0, 0, 0}, ws = 0;
    for (int i = 0; i < n; ++i) {
      const double kw = kappa[i] * w[i];
      for (int e = 0; e < d * d; ++e) M[e] += kw * R[(size_t)i * d * d + e];
      if (t) {
        const double tw = tau[i] * w[i];
        for (int a = 0; a < d; ++a) s[a] += tw * t[(size_t)i * d + a];
        ws += tw;
      }
    }
    if (t)
      for (int a = 0; a < d; ++a) topt[a] = s[a] / ws;
    project_to_rotation_group_host(d, M, Ropt);
  };
  auto rsq = [&](int i) {
    double s = kappa[i] * sqdist(d * d, Ropt, R + (size_t)i * d * d);
    if (t) s += tau[i] * sqdist(d, topt, t + (size_t)i * d);
    return s;
  };
  solve();
  double rmax = 0;
  for (int i = 0; i < n; ++i) rmax = std::max(rmax, rsq(i));
  const double barcSq = barc * barc;
  double muInit = barcSq / (2 * rmax - barcSq);
  muInit = std::min(muInit, 1e-5);
  if (muInit > 0) {  // negative: residuals already small, GNC is skipped
    dcora_robust_params prm;
    dcora_robust_params_default(&prm);
    prm.cost_type = DCORA_ROBUST_GNC_TLS;
    prm.GNCBarc = barc;
    prm.GNCMaxNumIters = max_iters;
    prm.GNCInitMu = muInit;
    RobustCost cost(prm);
    for (int iter = 0; iter < max_iters; ++iter) {
      solve();
      int nc = 0;
      for (int i = 0; i < n; ++i) {
        const double wi = cost.weight(std::sqrt(rsq(i)));
        if (wi < w_tol || wi > 1 - w_tol) nc++;
        w[i] = wi;
      }
      if (nc == n) break;
      cost.update();
    }
  }
  inliers.clear();
  for (int i = 0; i < n; ++i)
    if (w[i] > 1 - w_tol) inliers.push_back(i);
}
}  // namespace

void robust_single_rotation_averaging(int d, int n, const double *R, const double *kappa, double threshold,
                                      double *Ropt, std::vector<int> &inliers) {
  std::vector<double> k1((size_t)n, 1.0);
  robust_average(d, n, R, nullptr, kappa ? kappa : k1.data(), nullptr, threshold, 1000, Ropt, nullptr, inliers);
}
void robust_single_pose_averaging(int d, int n, const double *R, const double *t, const double *kappa,
                                  const double *tau, double threshold, double *Ropt, double *topt,
                                  std::vector<int> &inliers) {
  std::vector<double> k1((size_t)n, 10000.0), t1((size_t)n, 100.0);
  robust_average(d, n, R, t, kappa ? kappa : k1.data(), tau ? tau : t1.data(), threshold, 10000, Ropt, topt, inliers);
}

// ---- cross-robot frame alignment ----
namespace {
// C = A * B for poses [R t] (d x (d+1), column-major)
void pose_mul(int d, const double *A, const double *B, double *C) {
  double out[12];
  for (int c = 0; c <= d; ++c)
    for (int a = 0; a < d; ++a) {
      double s = (c == d) ? A[a + d * d] : 0.0;
      for (int q = 0; q < d; ++q) s += A[a + q * d] * B[q + c * d];
      out[a + c * d] = s;
    }
  for (int e = 0; e < d * (d + 1); ++e) C[e] = out[e];
}
void pose_inv(int d, const double *A, double *C) {
  double out[12];
  for (int c = 0; c < d; ++c)
    for (int a = 0; a < d; ++a) out[a + c * d] = A[c + a * d];
  for (int a = 0; a < d; ++a) {
    double s = 0;
    for (int q = 0; q < d; ++q) s += A[q + a * d] * A[q + d * d];
    out[a + d * d] = -s;
  }
  for (int e = 0; e < d * (d + 1); ++e) C[e] = out[e];
}
}  // namespace

void neighbor_transform(int d, bool incoming, const double *Rm, const double *tm, const double *T_w2_f2,
                        const double *T_w1_f1, double *T_out) {
  double dT[12], f1f2[12], tmp[12], inv[12];
  for (int e = 0; e < d * d; ++e) dT[e] = Rm[e];
  for (int a = 0; a < d; ++a) dT[d * d + a] = tm[a];
  if (incoming) {
    pose_inv(d, dT, f1f2);
  } else {
    for (int e = 0; e < d * (d + 1); ++e) f1f2[e] = dT[e];
  }
  pose_inv(d, f1f2, inv);
  pose_mul(d, T_w2_f2, inv, tmp);      // T_world2_frame1
  pose_inv(d, T_w1_f1, inv);
  pose_mul(d, tmp, inv, T_out);        // T_world2_world1
}

bool robust_neighbor_transform(int d, int m, const double *cand, bool two_stage, int min_inliers, double *T,
                               int *num_inliers) {
  if (num_inliers) *num_inliers = 0;
  if (m < 1) return false;
  const int ps = d * (d + 1);
  std::vector<double> R((size_t)m * d * d), t((size_t)m * d);
  for (int i = 0; i < m; ++i) {
    for (int e = 0; e < d * d; ++e) R[(size_t)i * d * d + e] = cand[(size_t)i * ps + e];
    for (int a = 0; a < d; ++a) t[(size_t)i * d + a] = cand[(size_t)i * ps + d * d + a];
  }
  std::vector<int> in;
  double Ropt[9], topt[3] = {0, 0, 0};
  if (two_stage) {
    std::vector<double> kappa((size_t)m, 1.0);
    const double max_rot_err = 2.0 * std::sqrt(2.0) * std::sin(0.5 / 2.0);  // angular2ChordalSO3(0.5)
    robust_single_rotation_averaging(d, m, R.data(), kappa.data(), max_rot_err, Ropt, in);
    if (num_inliers) *num_inliers = (int)in.size();
    if ((int)in.size() < min_inliers) return false;
    for (int i : in)
      for (int a = 0; a < d; ++a) topt[a] += t[(size_t)i * d + a];
    for (int a = 0; a < d; ++a) topt[a] /= (double)in.size();
  } else {
    std::vector<double> kappa((size_t)m, 1.82), tau((size_t)m, 0.01);
    double cbar = 0;
    if (!error_threshold_at_quantile(0.9, 3, &cbar)) return false;
    robust_single_pose_averaging(d, m, R.data(), t.data(), kappa.data(), tau.data(), cbar, Ropt, topt, in);
    if (num_inliers) *num_inliers = (int)in.size();
    if ((int)in.size() < min_inliers) return false;
  }
  for (int e = 0; e < d * d; ++e) T[e] = Ropt[e];
  for (int a = 0; a < d; ++a) T[d * d + a] = topt[a];
  return true;
}

void fixed_stiefel_variable(int r, int d, double *Y) {
  unsigned long long s = 1;  // the reference seeds std::srand(1)
  auto next = [&]() {
    unsigned long long z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return 2.0 * ((z >> 11) * (1.0 / 9007199254740992.0)) - 1.0;
  };
  for (int c = 0; c < d; ++c) {
    double *y = Y + (size_t)c * r;
    double nrm = 0;
    while (nrm < 1e-3) {  // a draw (numerically) inside the span of the earlier columns is redrawn
      for (int a = 0; a < r; ++a) y[a] = next();
      for (int pass = 0; pass < 2; ++pass)
        for (int p = 0; p < c; ++p) {
          const double *q = Y + (size_t)p * r;
          double dot = 0;
          for (int a = 0; a < r; ++a) dot += q[a] * y[a];
          for (int a = 0; a < r; ++a) y[a] -= dot * q[a];
        }
      nrm = 0;
      for (int a = 0; a < r; ++a) nrm += y[a] * y[a];
      nrm = std::sqrt(nrm);
    }
    for (int a = 0; a < r; ++a) y[a] /= nrm;
  }
}

void initialize_in_global_frame(int r, int d, int n, int l, int b, bool se, const double *Twr, const double *Tlocal,
                                const double *YLift, double *X) {
  const int dh = d + 1;
  const int k = dh * n + l + b;
  auto lifted = [&](int col, const double *g) {  // X(:, col) = YLift g
    for (int a = 0; a < r; ++a) {
      double s = 0;
      for (int q = 0; q < d; ++q) s += YLift[a + q * r] * g[q];
      X[(size_t)col * r + a] = s;
    }
  };
  auto apply = [&](const double *v, bool point, double *g) {  // g = R_wr v (+ t_wr)
    for (int a = 0; a < d; ++a) {
      double s = point ? Twr[d * d + a] : 0.0;
      for (int q = 0; q < d; ++q) s += Twr[a + q * d] * v[q];
      g[a] = s;
    }
  };
  double g[3];
  for (int col = 0; col < k; ++col) {
    bool point;
    if (se)
      point = (col % dh) == d;
    else
      point = col >= d * n + l;  // translations and landmarks move with the frame, rotations / unit spheres turn
    apply(Tlocal + (size_t)col * d, point, g);
    lifted(col, g);
  }
}

}  // namespace dcora
