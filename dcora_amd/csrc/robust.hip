// Robust pose-graph optimisation around the local solver:
//   measurement_errors : computeMeasurementError of every edge at once (ref src/DCORA_utils.cpp:2095-2101;
//                        Agent::computeMeasurementResidual, ref src/Agent.cpp:1342-1389, with lifted poses)
//   solve_pgo          : solvePGO (ref src/DCORA_solver.cpp:304-328) -- chordal start, one optimize() at rank d
//   solve_robust_pgo   : solveRobustPGO (ref :330-409) -- GNC-TLS outer loop around solve_pgo
#include <algorithm>
#include <cmath>
#include <vector>

#include "device_problem.h"
#include "host_graph.h"
#include "host_robust.h"

namespace dcora {

namespace {

struct EdgeDev {
  const int *p1, *p2;
  const double *R, *t, *kappa, *tau;  // R: d*d per edge (column-major), t: d per edge
};
// one thread per edge: kappa |Y1 R - Y2|^2 + tau |p2 - p1 - Y1 t|^2 with Y r x d, p r-vectors (SE ordering)
template <int D>
__global__ __launch_bounds__(kBlock) void k_measurement_errors(int r, int m, EdgeDev E, const double *__restrict__ X,
                                                               double *__restrict__ out) {
  const int e = blockIdx.x * kBlock + threadIdx.x;
  if (e >= m) return;
  constexpr int DH = D + 1;
  const double *X1 = X + (size_t)E.p1[e] * DH * r, *X2 = X + (size_t)E.p2[e] * DH * r;
  const double *Re = E.R + (size_t)e * D * D, *te = E.t + (size_t)e * D;
  double rot = 0, tr = 0;
  for (int q = 0; q < r; ++q) {
    double y1[D];
#pragma unroll
    for (int a = 0; a < D; ++a) y1[a] = X1[(size_t)a * r + q];
#pragma unroll
    for (int c = 0; c < D; ++c) {
      double s = 0;
#pragma unroll
      for (int a = 0; a < D; ++a) s += y1[a] * Re[c * D + a];
      const double dlt = s - X2[(size_t)c * r + q];
      rot += dlt * dlt;
    }
    double s = X2[(size_t)D * r + q] - X1[(size_t)D * r + q];
#pragma unroll
    for (int a = 0; a < D; ++a) s -= y1[a] * te[a];
    tr += s * s;
  }
  out[e] = E.kappa[e] * rot + E.tau[e] * tr;
}

int no_device() {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    set_last_error("no HIP device available: libdcora_hip has no CPU fallback");
    return DCORA_ERR_NO_DEVICE;
  }
  return DCORA_OK;
}

}  // namespace

// X: r x (d+1) n host (SE ordering, r >= d); out: one squared error per measurement (weights not applied)
int measurement_errors(const HostDataset &ds, int r, const double *X, double *out, int device) {
  int rc = no_device();
  if (rc) return rc;
  DCORA_HIP(hipSetDevice(device));
  const int d = ds.d, m = (int)ds.meas.size();
  if (m == 0) return DCORA_OK;
  std::vector<int> p1((size_t)m), p2((size_t)m);
  std::vector<double> R((size_t)m * d * d), t((size_t)m * d), ka((size_t)m), ta((size_t)m);
  for (int e = 0; e < m; ++e) {
    const PoseMeas &q = ds.meas[e];
    if (q.p1 < 0 || q.p1 >= ds.n || q.p2 < 0 || q.p2 >= ds.n) {
      set_last_error("measurement_errors: pose index out of range");
      return DCORA_ERR_BAD_ARG;
    }
    p1[e] = q.p1;
    p2[e] = q.p2;
    for (int i = 0; i < d * d; ++i) R[(size_t)e * d * d + i] = q.R[i];
    for (int i = 0; i < d; ++i) t[(size_t)e * d + i] = q.t[i];
    ka[e] = q.kappa;
    ta[e] = q.tau;
  }
  DevBuf<int> dp1, dp2;
  DevBuf<double> dR, dt, dk, dta, dX, dout;
  const size_t N = (size_t)r * (d + 1) * ds.n;
  DCORA_HIP(dp1.alloc(m));
  DCORA_HIP(dp2.alloc(m));
  DCORA_HIP(dR.alloc(R.size()));
  DCORA_HIP(dt.alloc(t.size()));
  DCORA_HIP(dk.alloc(m));
  DCORA_HIP(dta.alloc(m));
  DCORA_HIP(dX.alloc(N));
  DCORA_HIP(dout.alloc(m));
  DCORA_HIP(hipMemcpy(dp1.p, p1.data(), sizeof(int) * m, hipMemcpyHostToDevice));
  DCORA_HIP(hipMemcpy(dp2.p, p2.data(), sizeof(int) * m, hipMemcpyHostToDevice));
  DCORA_HIP(hipMemcpy(dR.p, R.data(), sizeof(double) * R.size(), hipMemcpyHostToDevice));
  DCORA_HIP(hipMemcpy(dt.p, t.data(), sizeof(double) * t.size(), hipMemcpyHostToDevice));
  DCORA_HIP(hipMemcpy(dk.p, ka.data(), sizeof(double) * m, hipMemcpyHostToDevice));
  DCORA_HIP(hipMemcpy(dta.p, ta.data(), sizeof(double) * m, hipMemcpyHostToDevice));
  DCORA_HIP(hipMemcpy(dX.p, X, sizeof(double) * N, hipMemcpyHostToDevice));
  EdgeDev E{dp1.p, dp2.p, dR.p, dt.p, dk.p, dta.p};
  const int grid = (m + kBlock - 1) / kBlock;
  if (d == 3)
    hipLaunchKernelGGL(k_measurement_errors<3>, dim3(grid), dim3(kBlock), 0, nullptr, r, m, E, dX.p, dout.p);
  else
    hipLaunchKernelGGL(k_measurement_errors<2>, dim3(grid), dim3(kBlock), 0, nullptr, r, m, E, dX.p, dout.p);
  DCORA_HIP(hipDeviceSynchronize());
  DCORA_HIP(hipMemcpy(out, dout.p, sizeof(double) * m, hipMemcpyDeviceToHost));
  return DCORA_OK;
}

// T0 (d x (d+1) n) may be null => chordal initialisation; Tout d x (d+1) n
int solve_pgo(const HostDataset &ds, const dcora_ropt_params &prm, const double *T0, double *Tout, int device,
              dcora_ropt_result *res) {
  int rc = no_device();
  if (rc) return rc;
  const int d = ds.d, n = ds.n;
  const size_t N = (size_t)d * (d + 1) * n;
  std::vector<double> T;
  if (T0) {
    T.assign(T0, T0 + N);
  } else if (!chordal_initialization(ds, T)) {
    set_last_error("solve_pgo: chordal initialisation failed (disconnected measurement graph?)");
    return DCORA_ERR_NOT_PD;
  }
  const int id = ds.meas.empty() ? 0 : ds.meas[0].r1;
  const HostCsr Q = build_Q_pgo(d, n, id, ds.meas);
  DeviceProblem P;
  dcora_dims dims{d, d, n, 0, 0};
  rc = P.init(dims, Q, nullptr, 0.1, device, nullptr);
  if (rc) return rc;
  dcora_ropt_result tmp;
  return P.optimize(prm, T.data(), Tout, res ? res : &tmp);
}

// fixed: m flags (fixedWeight of the reference); weights (in/out through ds.meas[i].weight, also copied to weights_out)
int solve_robust_pgo(HostDataset &ds, const dcora_ropt_params &prm, const dcora_robust_params &rp, const int *fixed,
                     const double *T0, double *Tout, double *weights_out, int device) {
  const double w_tol = 1e-8;
  const int m = (int)ds.meas.size(), d = ds.d;
  if (rp.cost_type != DCORA_ROBUST_GNC_TLS) {
    set_last_error("solve_robust_pgo: only GNC_TLS is supported (CHECK of the reference, src/DCORA_solver.cpp:347)");
    return DCORA_ERR_BAD_ARG;
  }
  int rc = solve_pgo(ds, prm, T0, Tout, device, nullptr);
  if (rc) return rc;
  std::vector<double> rsq((size_t)m);
  for (PoseMeas &q : ds.meas) q.weight = 1.0;
  rc = measurement_errors(ds, d, Tout, rsq.data(), device);
  if (rc) return rc;
  const double rmax = m ? *std::max_element(rsq.begin(), rsq.end()) : 0.0;
  const double barcSq = rp.GNCBarc * rp.GNCBarc;
  const double muInit = barcSq / (2 * rmax - barcSq);
  if (muInit > 0) {  // negative: small residuals, GNC skipped
    dcora_robust_params g = rp;
    g.GNCInitMu = muInit;
    RobustCost cost(g);
    for (int iter = 0; iter < g.GNCMaxNumIters; ++iter) {
      rc = solve_pgo(ds, prm, T0, Tout, device, nullptr);
      if (rc) return rc;
      rc = measurement_errors(ds, d, Tout, rsq.data(), device);
      if (rc) return rc;
      int undecided = 0;
      for (int i = 0; i < m; ++i) {
        if (fixed && fixed[i]) continue;
        const double w = cost.weight(std::sqrt(rsq[i]));
        ds.meas[i].weight = w;
        if (!(w < w_tol) && !(w > 1.0 - w_tol)) ++undecided;
      }
      if (undecided == 0) break;
      cost.update();
    }
  }
  rc = solve_pgo(ds, prm, T0, Tout, device, nullptr);
  if (rc) return rc;
  if (weights_out)
    for (int i = 0; i < m; ++i) weights_out[i] = ds.meas[i].weight;
  return DCORA_OK;
}

}  // namespace dcora
