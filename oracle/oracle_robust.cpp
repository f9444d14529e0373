// ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle.hpp).
// Robust estimation around the hot path: RobustCost (M-estimators + GNC-TLS), single rotation / pose averaging with
// GNC, measurement residuals, solvePGO / solveRobustPGO.
#include <algorithm>
#include <cmath>
#include <stdexcept>

#include "oracle.hpp"

namespace orc {

// ---- chi-square quantile (boost::math::quantile(chi_squared) in the reference, src/DCORA_utils.cpp:2103-2106) ----
// regularised lower incomplete gamma P(a, x) by series / continued fraction, inverted by bisection + Newton
static double gamma_p(double a, double x) {
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
double chi2inv(double quantile, int dof) {
  const double a = 0.5 * dof;
  double lo = 0.0, hi = std::max(4.0 * dof, 10.0);
  while (gamma_p(a, 0.5 * hi) < quantile) hi *= 2.0;
  for (int it = 0; it < 200; ++it) {
    const double mid = 0.5 * (lo + hi);
    if (gamma_p(a, 0.5 * mid) < quantile)
      lo = mid;
    else
      hi = mid;
    if (hi - lo < 1e-14 * std::max(1.0, hi)) break;
  }
  return 0.5 * (lo + hi);
}

// ---- RobustCost (ref: src/DCORA_robust.cpp:51-148, include/DCORA/DCORA_robust.h:25-140) ----
double RobustCost::weight(double r) const {
  switch (p.type) {
    case RobustType::L2: return 1;
    case RobustType::L1: return 1 / r;
    case RobustType::Huber: return r < p.HuberThreshold ? 1 : p.HuberThreshold / r;
    case RobustType::TLS: return r < p.TLSThreshold ? 1 : 0;
    case RobustType::GM: {
      const double a = 1 + r * r;
      return 1 / (a * a);
    }
    case RobustType::GNC_TLS: {
      const double rSq = r * r, bSq = p.GNCBarc * p.GNCBarc;
      const double ub = (mu + 1) / mu * bSq, lb = mu / (mu + 1) * bSq;
      if (rSq >= ub) return 0;
      if (rSq <= lb) return 1;
      return std::sqrt(bSq * mu * (mu + 1) / rSq) - mu;
    }
  }
  throw std::runtime_error("weight function not implemented");
}
void RobustCost::reset() {
  if (p.type == RobustType::GNC_TLS) {
    mu = p.GNCInitMu;
    iteration = 0;
  }
}
void RobustCost::update() {
  if (p.type != RobustType::GNC_TLS) return;
  iteration++;
  if (iteration > p.GNCMaxNumIters) return;  // "GNC: reached maximum iterations."
  mu = p.GNCMuStep * mu;
}
double error_threshold_at_quantile(double quantile, int dimension) {
  if (dimension != 3) throw std::runtime_error("quantile function currently only supports 3D problem");
  return quantile < 1 ? std::sqrt(chi2inv(quantile, 6)) : 1e5;
}

// ---- averaging (ref: src/DCORA_solver.cpp:28-216).  Rotations d x d column-major, concatenated ----
static void rotation_average(int d, int n, const double *R, const double *w, double *Ropt) {
  double M[9] = {0};
  for (int i = 0; i < n; ++i)
    for (int e = 0; e < d * d; ++e) M[e] += w[i] * R[(size_t)i * d * d + e];
  project_to_rotation_group(d, M, Ropt);
}
static void translation_average(int d, int n, const double *t, const double *w, double *topt) {
  double s[3] = {0, 0, 0}, ws = 0;
  for (int i = 0; i < n; ++i) {
    for (int a = 0; a < d; ++a) s[a] += w[i] * t[(size_t)i * d + a];
    ws += w[i];
  }
  for (int a = 0; a < d; ++a) topt[a] = s[a] / ws;
}
static double sqdist(int m, const double *a, const double *b) {
  double s = 0;
  for (int e = 0; e < m; ++e) s += (a[e] - b[e]) * (a[e] - b[e]);
  return s;
}

// shared GNC loop of robustSingleRotationAveraging (:76-141) and robustSinglePoseAveraging (:143-216)
static void robust_average(int d, int n, const double *R, const double *t, const double *kappa, const double *tau,
                           double barc, int max_iters, double *Ropt, double *topt, std::vector<int> &inliers) {
  const double w_tol = 1e-8;
  std::vector<double> w((size_t)n, 1.0), kw((size_t)n), tw((size_t)n);
  auto solve = [&]() {
    for (int i = 0; i < n; ++i) {
      kw[i] = kappa[i] * w[i];
      if (t) tw[i] = tau[i] * w[i];
    }
    if (t) translation_average(d, n, t, tw.data(), topt);
    rotation_average(d, n, R, kw.data(), Ropt);
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
  if (muInit > 0) {
    RobustParams prm;
    prm.type = RobustType::GNC_TLS;
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

// ---- residuals and the centralised robust solve ----
// computeMeasurementError (ref: src/DCORA_utils.cpp:2095-2101) with T r x (d+1) n in the SE ordering, r >= d
// (the reference evaluates it on lifted blocks as well, src/Agent.cpp:1342-1395)
double measurement_error(const Meas &m, int d, const Mat &T) {
  const int dh = d + 1, r = T.rows;
  double rot = 0, tr = 0;
  for (int c = 0; c < d; ++c)
    for (int a = 0; a < r; ++a) {
      double s = 0;
      for (int q = 0; q < d; ++q) s += T(a, m.p1 * dh + q) * m.R[q + c * d];
      const double e = s - T(a, m.p2 * dh + c);
      rot += e * e;
    }
  for (int a = 0; a < r; ++a) {
    double s = T(a, m.p2 * dh + d) - T(a, m.p1 * dh + d);
    for (int q = 0; q < d; ++q) s -= T(a, m.p1 * dh + q) * m.t[q];
    tr += s * s;
  }
  return m.kappa * rot + m.tau * tr;
}

// solvePGO (ref: src/DCORA_solver.cpp:304-328): chordal start unless T0 is given, one optimize() at rank d
Mat solve_pgo(const Dataset &ds, const ROptParams &prm, const Mat *T0) {
  Mat T = T0 ? *T0 : chordal_initialization(ds);
  Dims D;
  D.r = ds.d;
  D.d = ds.d;
  D.n = ds.n;
  const CSR Q = build_Q_pgo(ds.d, ds.n, ds.meas.empty() ? 0 : ds.meas[0].r1, ds.meas);
  Chol precon;
  const bool hasP = precon.factor(csr_add_diag(Q, 0.1), ds.d + 1);
  Problem P;
  P.D = D;
  P.Q = &Q;
  P.G = nullptr;
  P.precon = hasP ? &precon : nullptr;
  ROptResult res;
  return optimize(P, prm, T, &res);
}

// solveRobustPGO (ref: src/DCORA_solver.cpp:330-409); weights are written back into ds.meas
Mat solve_robust_pgo(Dataset &ds, const ROptParams &prm, const RobustParams &rp, const std::vector<char> &fixed,
                     const Mat *T0) {
  const double w_tol = 1e-8;
  const int m = (int)ds.meas.size();
  Mat T = solve_pgo(ds, prm, T0);
  double rmax = 0;
  for (int i = 0; i < m; ++i) {
    ds.meas[i].weight = 1.0;
    rmax = std::max(rmax, measurement_error(ds.meas[i], ds.d, T));
  }
  const double barcSq = rp.GNCBarc * rp.GNCBarc;
  const double muInit = barcSq / (2 * rmax - barcSq);
  if (muInit > 0) {
    RobustParams g = rp;
    g.type = RobustType::GNC_TLS;
    g.GNCInitMu = muInit;
    RobustCost cost(g);
    for (int iter = 0; iter < g.GNCMaxNumIters; ++iter) {
      T = solve_pgo(ds, prm, T0);
      int undecided = 0;
      for (int i = 0; i < m; ++i) {
        if (fixed[i]) continue;
        ds.meas[i].weight = cost.weight(std::sqrt(measurement_error(ds.meas[i], ds.d, T)));
        if (!(ds.meas[i].weight < w_tol) && !(ds.meas[i].weight > 1.0 - w_tol)) ++undecided;
      }
      if (undecided == 0) break;
      cost.update();
    }
  }
  return solve_pgo(ds, prm, T0);
}

}  // namespace orc
