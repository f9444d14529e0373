// The reference's synchronous multi-robot driver for range-aided SLAM with the Riemannian staircase (ref
// examples/MultiRobotExample_RASLAM.cpp) as a C++ program over the C ABI of include/dcora_hip.h and the facade's
// Matrix-level functions -- host code stays C++, every numerical step runs on the MI355X:
//
//   multi-robot-example-raslam <file.pyfg> [--rank r_min] [--iters N] [--rgrad-tol t] [--seed s] [--quiet]
//
//   X = odometry start of the CORA driver at rank d (examples/SingleRobotExample_RASLAM.cpp:92-150), lifted to r_min
//   for r = r_min, r_min + 1, ...
//     agents at rank r: one per robot of the file, each on its own poses, unit spheres and landmarks     dcora_ra_rbcd_create
//     RBCD++ with greedy block selection until |rgrad| < tol or N iterations                             dcora_ra_rbcd_iterate
//     S = Q - Lambda(X) of the merged problem; fastVerification(S, 1e-3)                                 dcora_cert_*
//     certified: done;  else escapeSaddle of the central problem into rank r + 1                         dcora_problem_escape_saddle
//
// The last line on stdout is a one-line JSON summary.  Exit code 0 = ran (certified or not), 2 = no GPU, 1 = error.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "DCORA/DCORA_utils.h"
#include "DCORA/QuadraticProblem.h"

namespace {
double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
}  // namespace

int main(int argc, char **argv) {
  if (argc < 2) {
    std::printf("usage: %s file.pyfg [--rank r_min] [--iters N] [--rgrad-tol t] [--seed s] [--quiet]\n", argv[0]);
    return 1;
  }
  const char *path = argv[1];
  int r_min = 0, numIters = 1000;
  unsigned long long seed = 20250310ull;
  double RGradNormTol = 0.1;
  const double min_eig_num_tol = 1e-3, gradient_tolerance = 1e-4, preconditioned_gradient_tolerance = 1e-4;
  bool quiet = false;
  for (int i = 2; i < argc; ++i) {
    if (!std::strcmp(argv[i], "--rank") && i + 1 < argc) r_min = std::atoi(argv[++i]);
    else if (!std::strcmp(argv[i], "--iters") && i + 1 < argc) numIters = std::atoi(argv[++i]);
    else if (!std::strcmp(argv[i], "--rgrad-tol") && i + 1 < argc) RGradNormTol = std::atof(argv[++i]);
    else if (!std::strcmp(argv[i], "--seed") && i + 1 < argc) seed = std::strtoull(argv[++i], nullptr, 10);
    else if (!std::strcmp(argv[i], "--quiet")) quiet = true;
  }
  if (dcora_device_count() < 1) {
    std::printf("no GPU: libdcora_hip has no CPU fallback\n");
    return 2;
  }
  try {
    dcora_radataset_t ds;
    DCORA::check_status(dcora_radataset_load_pyfg(path, &ds), "read_pyfg_file");
    int info[7];
    DCORA::check_status(dcora_radataset_info(ds, info), "info");
    const int d = info[0], n = info[1], l = info[2], b = info[3];
    const unsigned k = (unsigned)((d + 1) * n + l + b);
    if (r_min <= 0) r_min = d;
    const int r_max = r_min + 12;
    std::printf("Loaded %d poses, %d unit spheres, %d landmarks (d = %d): %d pose-pose, %d pose-landmark, %d range "
                "measurements\n", n, l, b, d, info[4], info[5], info[6]);

    // merged quadratic form (certificate, escape step) and its preconditioner regularisation (ref src/Graph.cpp:1921-1960)
    dcora_csr_t Qh;
    DCORA::check_status(dcora_radataset_build_Q(ds, &Qh), "constructQuadraticCostTermRASLAM");
    const DCORA::SparseMatrix Q = DCORA::detail::take(Qh);

    // start point: the odometry initialisation at rank d, lifted by zero rows
    std::vector<double> x0((size_t)d * k);
    DCORA::check_status(dcora_radataset_odometry_init(ds, seed, x0.data()), "odometryInitialization");
    DCORA::Matrix Xcurr((size_t)r_min, k);
    for (unsigned c = 0; c < k; ++c)
      for (int i = 0; i < d; ++i) Xcurr(i, c) = x0[(size_t)c * d + i];

    double setup_ms = 0, rbcd_ms = 0, cert_ms = 0, escape_ms = 0, cost2 = 0, gradnorm = 0, theta = 0, reg = -1;
    int totalIter = 0, r = r_min, levels = 0, num_agents = 0;
    bool certified = false;
    DCORA::Matrix Xopt;
    for (; r < r_max; ++r) {
      ++levels;
      double t0 = now_ms();
      dcora_rbcd_options opt;
      dcora_rbcd_options_default(&opt);
      opt.r = r;
      opt.acceleration = 1;
      opt.local.RTR_iterations = 200;        // the example's local solver: RTR 200 x 200 at 1e-4
      opt.local.RTR_tCG_iterations = 200;
      opt.local.gradnorm_tol = 1e-4;
      dcora_ra_rbcd_t s;
      DCORA::check_status(dcora_ra_rbcd_create(ds, &opt, &s), "agents");
      DCORA::check_status(dcora_ra_rbcd_info(s, &num_agents, nullptr), "agents");
      DCORA::check_status(dcora_ra_rbcd_set_X(s, Xcurr.data()), "setX");
      setup_ms += now_ms() - t0;
      t0 = now_ms();
      int selected = 0;
      for (int iter = 0; iter < numIters; ++iter) {
        int next = selected;
        DCORA::check_status(dcora_ra_rbcd_iterate(s, selected, &cost2, &gradnorm, nullptr, &next), "iterate");
        if (!quiet)
          std::printf("Iter = %d | robot = %d | cost = %.5f | gradnorm = %.5f\n", totalIter, selected, cost2, gradnorm);
        ++totalIter;
        if (gradnorm < RGradNormTol) break;
        selected = next;
      }
      Xopt = DCORA::Matrix((size_t)r, k);
      DCORA::check_status(dcora_ra_rbcd_get_X(s, Xopt.data()), "getX");
      DCORA::check_status(dcora_ra_rbcd_destroy(s), "agents");
      rbcd_ms += now_ms() - t0;
      t0 = now_ms();
      const DCORA::SparseMatrix S =
          DCORA::constructDualCertificateMatrixRASLAM(Xopt, Q, (unsigned)d, (unsigned)n, (unsigned)l, (unsigned)b);
      DCORA::Vector min_eigenvector;
      min_eigenvector.assign((size_t)k, 0.0);
      int psd = 0;
      double lmin = 0;
      DCORA::check_status(dcora_cert_fast_verification(S.n, S.rowptr.data(), S.colidx.data(), S.vals.data(), min_eig_num_tol,
                                                       1, 0, &psd, &theta, min_eigenvector.data(), &lmin),
                          "fastVerification");
      cert_ms += now_ms() - t0;
      if (psd) {
        std::printf("Z = (X*)^T(X*) is a global minimizer at rank %d (2 f = %.6f, |rgrad| = %.4g)\n", r, cost2, gradnorm);
        certified = true;
        break;
      }
      if (theta >= -min_eig_num_tol / 2) {
        std::printf("Error: escape direction computation did not converge to the desired precision\n");
        break;
      }
      std::printf("Saddle point detected at rank %d! Curvature along escape direction: %g\n", r, theta);
      t0 = now_ms();
      if (reg < 0)
        DCORA::check_status(dcora_graph_precond_regularization(Q.n, Q.rowptr.data(), Q.colidx.data(), Q.vals.data(), 0, &reg),
                            "computePreconditionerRegularization");
      DCORA::ProblemData pd;
      pd.r = (unsigned)(r + 1); pd.d = (unsigned)d; pd.n = (unsigned)n; pd.l = (unsigned)l; pd.b = (unsigned)b;
      pd.Q = Q;
      pd.precond_reg = reg;
      DCORA::QuadraticProblem problemCentralNextRank(pd);
      DCORA::Matrix X;
      const bool escape_success = problemCentralNextRank.escapeSaddle(Xopt, theta, min_eigenvector, gradient_tolerance,
                                                                     preconditioned_gradient_tolerance, &X);
      escape_ms += now_ms() - t0;
      if (!escape_success) {
        std::printf("Warning: backtracking line search failed to escape from the saddle point\n");
        break;
      }
      Xcurr = X;
    }
    dcora_radataset_destroy(ds);
    std::printf("{\"rank\": %d, \"levels\": %d, \"agents\": %d, \"iterations\": %d, \"cost_2f\": %.12g, \"gradnorm\": %.6g, "
                "\"certified\": %s, \"theta\": %.6g, \"agent_setup_ms\": %.3f, \"rbcd_ms\": %.3f, \"certification_ms\": %.3f, "
                "\"escape_ms\": %.3f}\n",
                (int)Xopt.rows(), levels, num_agents, totalIter, cost2, gradnorm, certified ? "true" : "false", theta, setup_ms,
                rbcd_ms, cert_ms, escape_ms);
    return 0;
  } catch (const std::exception &e) {
    std::printf("error: %s\n", e.what());
    return 1;
  }
}
