// The reference's synchronous multi-robot driver with the Riemannian staircase (ref examples/MultiRobotExample.cpp:
// 121-372) as a C++ program over the facade classes and the C ABI of include/dcora_hip.h -- host code stays C++, every
// numerical step runs on the MI355X:
//
//   multi-robot-example <num_robots> <file.g2o> [--rank r_min] [--iters N] [--rgrad-tol t] [--out trajectory.txt]
//
//   for r = r_min, r_min + 1, ...                                              reference lines
//     agents at rank r (one RBCD session), X = current point                   :172-217
//     RBCD++ with greedy block selection until |rgrad| < tol or N iterations   :223-307   dcora_rbcd_iterate
//     S = Q - Lambda(X); fastVerification(S, 1e-3)                             :320-334   dcora_cert_*
//     certified: suboptimality gap, rounding in the frame of pose 0, done      :336-350   dcora_round_align_trajectory
//     else escapeSaddle into rank r + 1                                        :352-366   dcora_problem_escape_saddle
//
// The last line on stdout is a one-line JSON summary (rank, iterations, cost 2f, gradient norm, certified, theta, gap,
// milliseconds per phase).  Exit code 0 = ran (certified or not), 2 = no GPU, 1 = error.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "DCORA/Agent.h"
#include "DCORA/DCORA_utils.h"
#include "DCORA/QuadraticProblem.h"

namespace {
double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
}  // namespace

int main(int argc, char **argv) {
  if (argc < 3) {
    std::printf("usage: %s num_robots file.g2o [--rank r_min] [--iters N] [--rgrad-tol t] [--out file] [--quiet]\n", argv[0]);
    return 1;
  }
  const unsigned num_robots = (unsigned)std::atoi(argv[1]);
  const char *path = argv[2];
  unsigned r_min = 5, r_max = 16, numIters = 1000;
  double RGradNormTol = 0.1;
  const double min_eig_num_tol = 1e-3, gradient_tolerance = 1e-6, preconditioned_gradient_tolerance = 1e-6;
  const char *out_path = nullptr;
  bool quiet = false;
  for (int i = 3; i < argc; ++i) {
    if (!std::strcmp(argv[i], "--rank") && i + 1 < argc) r_min = (unsigned)std::atoi(argv[++i]);
    else if (!std::strcmp(argv[i], "--iters") && i + 1 < argc) numIters = (unsigned)std::atoi(argv[++i]);
    else if (!std::strcmp(argv[i], "--rgrad-tol") && i + 1 < argc) RGradNormTol = std::atof(argv[++i]);
    else if (!std::strcmp(argv[i], "--out") && i + 1 < argc) out_path = argv[++i];
    else if (!std::strcmp(argv[i], "--quiet")) quiet = true;
  }
  if (dcora_device_count() < 1) {
    std::printf("no GPU: libdcora_hip has no CPU fallback\n");
    return 2;
  }
  try {
    dcora_dataset_t ds;
    DCORA::check_status(dcora_dataset_load_g2o(path, &ds), "read_g2o_file");
    int d = 0, n = 0, m = 0;
    dcora_dataset_info(ds, &d, &n, &m);
    const unsigned dh = d + 1, k = dh * n;
    std::printf("Loaded %d poses, %d measurements (d = %d); %u robots\n", n, m, d, num_robots);

    // central quadratic form (evaluation of the certificate and the escape step, :173-182)
    std::vector<int> ids((size_t)4 * m);
    std::vector<double> vals((size_t)m * (d * d + d + 3));
    DCORA::check_status(dcora_dataset_copy(ds, ids.data(), vals.data()), "measurements");
    dcora_csr_t Qh;
    DCORA::check_status(dcora_graph_build_Q_pgo(d, n, 0, m, ids.data(), vals.data(), &Qh), "constructQuadraticCostTermPGO");
    const DCORA::SparseMatrix Q = DCORA::detail::take(Qh);

    // InitializationMethod::Chordal lifted to rank r_min (:150-153, src/Agent.cpp:502)
    const double t_init0 = now_ms();
    std::vector<double> T((size_t)d * k);
    DCORA::check_status(dcora_dataset_chordal_init(ds, T.data()), "chordalInitialization");
    DCORA::Matrix Xcurr(r_min, k);
    for (unsigned c = 0; c < k; ++c)
      for (int i = 0; i < d; ++i) Xcurr(i, c) = T[(size_t)c * d + i];
    const double init_ms = now_ms() - t_init0;

    double setup_ms = 0, rbcd_ms = 0, cert_ms = 0, escape_ms = 0, cost2 = 0, gradnorm = 0, theta = 0, gap_f = 0, n_eff = 0;
    unsigned totalIter = 0, r = r_min, levels = 0;
    bool certified = false;
    DCORA::Matrix Xopt;
    // The certificate S = Q - Lambda has Q's pattern: the analysis of its PSD test (ordering, fronts, device image) runs
    // on another host thread while the agents iterate (dcora_cert_prepare, an addition of this library; the reference
    // analyses inside isSparseSymmetricMatrixPSD after the loop).  Joined before the first fastVerification.
    std::thread cert_prepare([&] {
      const dcora_dims pd{1, d, n, 0, 0};
      (void)dcora_cert_prepare(&pd, Q.rowptr.data(), Q.colidx.data(), (int)dh, 0);
    });
    struct Joiner {
      std::thread &t;
      ~Joiner() {
        if (t.joinable()) t.join();
      }
    } cert_prepare_joiner{cert_prepare};
    for (; r < r_max; ++r) {
      ++levels;
      double t0 = now_ms();
      DCORA::AgentParameters options((unsigned)d, r, num_robots);
      options.acceleration = true;
      auto team = DCORA::AgentTeam::create(ds, options);
      DCORA::check_status(dcora_rbcd_set_X(team->session(), Xcurr.data()), "setX");
      setup_ms += now_ms() - t0;
      t0 = now_ms();
      int selectedRobot = 0;
      for (unsigned iter = 0; iter < numIters; ++iter) {
        int next = selectedRobot;
        DCORA::check_status(dcora_rbcd_iterate(team->session(), selectedRobot, &cost2, &gradnorm, nullptr, &next), "iterate");
        if (!quiet)
          std::printf("Iter = %u | robot = %d | cost = %.5f | gradnorm = %.5f\n", totalIter, selectedRobot, cost2, gradnorm);
        ++totalIter;
        if (gradnorm < RGradNormTol) break;
        selectedRobot = next;
      }
      Xopt = DCORA::Matrix(r, k);
      DCORA::check_status(dcora_rbcd_get_X(team->session(), Xopt.data()), "getX");
      rbcd_ms += now_ms() - t0;
      t0 = now_ms();
      const DCORA::SparseMatrix S = DCORA::constructDualCertificateMatrixPGO(Xopt, Q, (unsigned)d, (unsigned)n);
      if (cert_prepare.joinable()) cert_prepare.join();
      DCORA::Vector min_eigenvector;
      int psd = 0;
      double lmin = 0;
      min_eigenvector.assign((size_t)k, 0.0);
      DCORA::check_status(dcora_cert_fast_verification(S.n, S.rowptr.data(), S.colidx.data(), S.vals.data(), min_eig_num_tol,
                                                       (int)dh, 0, &psd, &theta, min_eigenvector.data(), &lmin),
                          "fastVerification");
      cert_ms += now_ms() - t0;
      dcora_dims dims{(int)r, d, n, 0, 0};
      DCORA::check_status(dcora_cert_suboptimality_gap(&dims, Xopt.data(), psd ? -min_eig_num_tol : lmin - min_eig_num_tol,
                                                       &gap_f, &n_eff),
                          "suboptimality_gap");
      if (psd) {
        std::printf("Z = (X*)^T(X*) is a global minimizer at rank %u (2 f = %.6f, |rgrad| = %.4f)\n", r, cost2, gradnorm);
        certified = true;
        break;
      }
      if (theta >= -min_eig_num_tol / 2) {  // :332-334
        std::printf("Error: escape direction computation did not converge to the desired precision\n");
        break;
      }
      std::printf("Saddle point detected at rank %u! Curvature along escape direction: %g\n", r, theta);
      t0 = now_ms();
      DCORA::ProblemData pd;
      pd.r = r + 1; pd.d = (unsigned)d; pd.n = (unsigned)n;
      pd.Q = Q;
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
    // rounding: trajectory in the frame of the first pose (:336-350, src/Agent.cpp:950-1034)
    if (out_path) {
      DCORA::Matrix anchor(Xopt.rows(), dh);
      for (unsigned c = 0; c < dh; ++c)
        for (size_t i = 0; i < Xopt.rows(); ++i) anchor(i, c) = Xopt(i, c);
      const DCORA::Matrix Traj = DCORA::alignLiftedTrajectoryToFrame(Xopt, anchor, (unsigned)d, (unsigned)n, true);
      DCORA::check_status(dcora_log_trajectory(out_path, d, n, Traj.data()), "logTrajectory");
    }
    dcora_dataset_destroy(ds);
    std::printf("{\"rank\": %u, \"levels\": %u, \"iterations\": %u, \"cost_2f\": %.12g, \"gradnorm\": %.6g, \"certified\": %s, "
                "\"theta\": %.6g, \"suboptimality_gap_f\": %.6g, \"n_eff\": %.6g, \"init_ms\": %.3f, \"agent_setup_ms\": %.3f, "
                "\"rbcd_ms\": %.3f, \"certification_ms\": %.3f, \"escape_ms\": %.3f}\n",
                (unsigned)Xopt.rows(), levels, totalIter, cost2, gradnorm, certified ? "true" : "false", theta, gap_f, n_eff,
                init_ms, setup_ms, rbcd_ms, cert_ms, escape_ms);
    return 0;
  } catch (const std::exception &e) {
    std::printf("error: %s\n", e.what());
    return 1;
  }
}
