// The reference's centralised CORA driver for range-aided SLAM (ref examples/SingleRobotExample_RASLAM.cpp:48-283) as a
// C++ program over the facade classes and the C ABI of include/dcora_hip.h:
//
//   single-robot-example-raslam <file.pyfg> [--seed s] [--rmax r]
//
//   X = odometry start at rank d (ref :92-150)
//   for r = d, d + 1, ...
//     RTR 200 x 200 at 1e-4 on the problem of rank r                                    QuadraticOptimizer::optimize
//     S = Q - Lambda(X); fastVerification(S, 1e-4)                                      dcora_cert_*
//     certified: projectSolutionRASLAM to rank d, refine there, done (ref :223-234)     dcora_round_project_raslam
//     else escapeSaddle (second-order step) into rank r + 1                             dcora_problem_escape_saddle
//
// The last line on stdout is a one-line JSON summary.  Exit code 0 = ran, 2 = no GPU, 1 = error.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <vector>

#include "DCORA/DCORA_utils.h"
#include "DCORA/QuadraticOptimizer.h"
#include "DCORA/QuadraticProblem.h"

namespace {
double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
}  // namespace

int main(int argc, char **argv) {
  if (argc < 2) {
    std::printf("usage: %s file.pyfg [--seed s] [--rmax r]\n", argv[0]);
    return 1;
  }
  unsigned long long seed = 20250310ull;
  unsigned r_max = 20;
  for (int i = 2; i < argc; ++i) {
    if (!std::strcmp(argv[i], "--seed") && i + 1 < argc) seed = std::strtoull(argv[++i], nullptr, 10);
    else if (!std::strcmp(argv[i], "--rmax") && i + 1 < argc) r_max = (unsigned)std::atoi(argv[++i]);
  }
  if (dcora_device_count() < 1) {
    std::printf("no GPU: libdcora_hip has no CPU fallback\n");
    return 2;
  }
  try {
    const double t_start = now_ms();
    dcora_radataset_t ds;
    DCORA::check_status(dcora_radataset_load_pyfg(argv[1], &ds), "read_pyfg_file");
    int info[7];
    DCORA::check_status(dcora_radataset_info(ds, info), "info");
    const unsigned d = (unsigned)info[0], n = (unsigned)info[1], l = (unsigned)info[2], b = (unsigned)info[3];
    const unsigned k = (d + 1) * n + l + b;
    dcora_csr_t Qh;
    DCORA::check_status(dcora_radataset_build_Q(ds, &Qh), "constructQuadraticCostTermRASLAM");
    const DCORA::SparseMatrix Q = DCORA::detail::take(Qh);
    double reg = 0.1;
    DCORA::check_status(dcora_graph_precond_regularization(Q.n, Q.rowptr.data(), Q.colidx.data(), Q.vals.data(), 0, &reg),
                        "computePreconditionerRegularization");
    std::vector<double> x0((size_t)d * k);
    DCORA::check_status(dcora_radataset_odometry_init(ds, seed, x0.data()), "odometryInitialization");
    dcora_radataset_destroy(ds);
    DCORA::Matrix X(d, k);
    for (unsigned c = 0; c < k; ++c)
      for (unsigned i = 0; i < d; ++i) X(i, c) = x0[(size_t)c * d + i];

    DCORA::ROptParameters params;
    params.RTR_iterations = 200;
    params.RTR_tCG_iterations = 200;
    params.gradnorm_tol = 1e-4;
    const double min_eig_num_tol = 1e-4;
    auto problem_at = [&](unsigned r) {
      DCORA::ProblemData pd;
      pd.r = r; pd.d = d; pd.n = n; pd.l = l; pd.b = b;
      pd.Q = Q;
      pd.precond_reg = reg;
      return std::make_unique<DCORA::QuadraticProblem>(pd);
    };
    unsigned r = d, levels = 0;
    bool certified = false;
    double f = 0, gradnorm = 0, theta = 0, f_rounded = 0;
    for (; r < r_max; ++r) {
      ++levels;
      auto P = problem_at(r);
      DCORA::QuadraticOptimizer opt(P.get(), params);
      DCORA::Matrix Xopt = opt.optimize(X);
      f = opt.getOptResult().fOpt;
      gradnorm = opt.getOptResult().gradNormOpt;
      const DCORA::SparseMatrix S = DCORA::constructDualCertificateMatrixRASLAM(Xopt, Q, d, n, l, b);
      DCORA::Vector v;
      const bool psd = DCORA::fastVerification(S, min_eig_num_tol, &theta, &v);
      std::printf("rank %u: f = %.9g, |rgrad| = %.3g, %s\n", r, f, gradnorm,
                  psd ? "certified" : "saddle");
      if (psd) {
        certified = true;
        // rounding: rank-d truncation + refinement at rank d (ref :223-234)
        DCORA::Matrix Xp = r == d ? Xopt : DCORA::projectSolutionRASLAM(Xopt, r, d, n, l, b);
        auto Pd = problem_at(d);
        DCORA::QuadraticOptimizer refine(Pd.get(), params);
        const DCORA::Matrix Xr = refine.optimize(Xp);
        f_rounded = refine.getOptResult().fOpt;
        X = Xopt;
        break;
      }
      if (theta >= -min_eig_num_tol / 2) {
        std::printf("Error: escape direction computation did not converge to the desired precision\n");
        X = Xopt;
        break;
      }
      auto Pn = problem_at(r + 1);
      DCORA::Matrix Xn;
      if (!Pn->escapeSaddle(Xopt, theta, v, 1e-4, 1e-4, &Xn, /*isSecondOrder=*/true)) {
        std::printf("Warning: backtracking line search failed to escape from the saddle point\n");
        X = Xopt;
        break;
      }
      X = Xn;
    }
    std::printf("{\"rank\": %u, \"levels\": %u, \"certified\": %s, \"f\": %.12g, \"gradnorm\": %.6g, \"theta\": %.6g, "
                "\"f_rounded\": %.12g, \"ms_total\": %.3f}\n",
                (unsigned)X.rows(), levels, certified ? "true" : "false", f, gradnorm, theta, f_rounded, now_ms() - t_start);
    return 0;
  } catch (const std::exception &e) {
    std::printf("error: %s\n", e.what());
    return 1;
  }
}
