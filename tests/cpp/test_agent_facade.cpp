// The reference's multi-robot driver loop (ref examples/MultiRobotExample.cpp:184-307) written against the façade
// classes DCORA::Agent / DCORA::QuadraticProblem, next to the same run through dcora_rbcd_iterate: the two must
// produce the same block sequence and the same costs.  usage: test_agent_facade <file.g2o>
// Exit code 0 = pass, 2 = no GPU (the library has no CPU fallback), 1 = failure.
#include <cmath>
#include <cstdio>
#include <vector>

#include "DCORA/Agent.h"
#include "DCORA/QuadraticProblem.h"

int main(int argc, char **argv) {
  if (argc < 2) {
    std::printf("usage: %s file.g2o\n", argv[0]);
    return 1;
  }
  if (dcora_device_count() < 1) {
    std::printf("no GPU: facade compiled and linked, compute skipped\n");
    return 2;
  }
  dcora_dataset_t ds;
  DCORA::check_status(dcora_dataset_load_g2o(argv[1], &ds), "load");
  int d = 0, n = 0, m = 0;
  dcora_dataset_info(ds, &d, &n, &m);
  const unsigned num_robots = 5, r = 5, numIters = 40;
  const unsigned dh = d + 1, k = dh * n;

  // start point: chordal initialisation lifted to rank r (InitializationMethod::Chordal, :150-153)
  std::vector<double> T((size_t)d * k);
  DCORA::check_status(dcora_dataset_chordal_init(ds, T.data()), "chordal");
  DCORA::Matrix Xcurr(r, k);
  for (unsigned c = 0; c < k; ++c)
    for (int i = 0; i < d; ++i) Xcurr(i, c) = T[(size_t)c * d + i];

  // central problem used for evaluation (:173-177)
  std::vector<int> ids((size_t)4 * m);
  std::vector<double> vals((size_t)m * (d * d + d + 3));
  DCORA::check_status(dcora_dataset_copy(ds, ids.data(), vals.data()), "copy");
  dcora_csr_t Qh;
  DCORA::check_status(dcora_graph_build_Q_pgo(d, n, 0, m, ids.data(), vals.data(), &Qh), "Q");
  DCORA::ProblemData pd;
  pd.r = r; pd.d = d; pd.n = n; pd.precond_reg = -1.0;
  int kk = 0, nnz = 0;
  dcora_csr_info(Qh, &kk, &nnz);
  pd.Q.n = kk;
  pd.Q.rowptr.resize(kk + 1);
  pd.Q.colidx.resize(nnz);
  pd.Q.vals.resize(nnz);
  dcora_csr_copy(Qh, pd.Q.rowptr.data(), pd.Q.colidx.data(), pd.Q.vals.data());
  dcora_csr_destroy(Qh);
  DCORA::QuadraticProblem problemCentral(pd);

  // agents (:184-217)
  DCORA::AgentParameters options(d, r, num_robots);
  options.acceleration = true;
  auto team = DCORA::AgentTeam::create(ds, options);
  auto &agents = team->agents;
  const unsigned per = n / num_robots;
  std::vector<unsigned> startIdx(num_robots), endIdx(num_robots);
  for (unsigned robot = 0; robot < num_robots; ++robot) {
    startIdx[robot] = robot * per;
    endIdx[robot] = (robot == num_robots - 1) ? (unsigned)n : (robot + 1) * per;
    const unsigned cols = (endIdx[robot] - startIdx[robot]) * dh;
    DCORA::Matrix Xb(r, cols);
    for (unsigned c = 0; c < cols; ++c)
      for (unsigned i = 0; i < r; ++i) Xb(i, c) = Xcurr(i, startIdx[robot] * dh + c);
    agents[robot]->setX(Xb);
  }

  // the loop (:223-307)
  std::vector<double> cost_facade;
  std::vector<unsigned> sel_facade;
  unsigned selectedRobot = 0;
  DCORA::Matrix Xopt(r, k);
  for (unsigned iter = 0; iter < numIters; ++iter) {
    auto &selectedRobotPtr = agents[selectedRobot];
    for (auto &robotPtr : agents) {
      if (robotPtr->iteration_number() != iter) {
        std::printf("iteration_number %u != %u\n", robotPtr->iteration_number(), iter);
        return 1;
      }
      if (robotPtr->getID() != selectedRobot) robotPtr->iterate(false);
    }
    for (auto &robotPtr : agents) {
      if (robotPtr->getID() == selectedRobot) continue;
      DCORA::PoseDict sharedPoses;
      if (!robotPtr->getSharedStateDicts(&sharedPoses)) continue;
      selectedRobotPtr->updateNeighborStates(robotPtr->getID(), sharedPoses);
      selectedRobotPtr->updateNeighborStates(robotPtr->getID(), sharedPoses, true);  // auxiliary poses (:246-258)
    }
    selectedRobotPtr->iterate(true);
    for (unsigned robot = 0; robot < num_robots; ++robot) {
      DCORA::Matrix XRobot;
      agents[robot]->getX(&XRobot);
      for (unsigned c = 0; c < XRobot.cols(); ++c)
        for (unsigned i = 0; i < r; ++i) Xopt(i, startIdx[robot] * dh + c) = XRobot(i, c);
    }
    const DCORA::Matrix RGrad = problemCentral.RieGrad(Xopt);
    cost_facade.push_back(2 * problemCentral.f(Xopt));
    sel_facade.push_back(selectedRobot);
    // greedy selection (:288-305)
    double best = -1;
    unsigned arg = 0;
    for (unsigned robot = 0; robot < num_robots; ++robot) {
      double s = 0;
      for (unsigned c = startIdx[robot] * dh; c < endIdx[robot] * dh; ++c)
        for (unsigned i = 0; i < r; ++i) s += RGrad(i, c) * RGrad(i, c);
      if (std::sqrt(s) > best) {
        best = std::sqrt(s);
        arg = robot;
      }
    }
    selectedRobot = arg;
  }

  // the same run through the session's own loop body
  dcora_rbcd_options o;
  dcora_rbcd_options_default(&o);
  o.num_robots = (int)num_robots;
  o.r = (int)r;
  o.acceleration = 1;
  dcora_rbcd_t s;
  DCORA::check_status(dcora_rbcd_create(ds, &o, &s), "session");
  DCORA::check_status(dcora_rbcd_set_X(s, Xcurr.data()), "set_X");
  int selected = 0;
  bool ok = true;
  double worst = 0;
  for (unsigned iter = 0; iter < numIters; ++iter) {
    double c2 = 0, gn = 0;
    int nxt = 0;
    DCORA::check_status(dcora_rbcd_iterate(s, selected, &c2, &gn, nullptr, &nxt), "iterate");
    if ((unsigned)selected != sel_facade[iter]) {
      std::printf("iteration %u: block %d vs %u\n", iter, selected, sel_facade[iter]);
      ok = false;
      break;
    }
    worst = std::fmax(worst, std::fabs(c2 - cost_facade[iter]) / std::fabs(c2));
    selected = nxt;
  }
  dcora_rbcd_destroy(s);
  dcora_dataset_destroy(ds);
  std::printf("agent facade: %u iterations, 2f %.6f -> %.6f, max relative cost difference to dcora_rbcd_iterate %.2e\n",
              numIters, cost_facade.front(), cost_facade.back(), worst);
  ok = ok && worst < 1e-9 && cost_facade.back() < cost_facade.front();
  // a second iterate(true) of the same agent inside one round is refused (lockstep)
  agents[0]->iterate(true);
  bool refused = false;
  try {
    agents[0]->iterate(true);
  } catch (const std::exception &e) {
    refused = true;
  }
  ok = ok && refused;
  // poses that are not the neighbour's current state are refused, not silently ignored (ref src/Agent.cpp:844-906:
  // the reference optimises against what it was handed; this facade optimises against the shared mirror)
  {
    DCORA::PoseDict shared;
    agents[1]->getSharedStateDicts(&shared);
    agents[0]->updateNeighborStates(1, shared);  // the current state: accepted
    shared.begin()->second(0, 0) += 1e-3;
    bool stale_refused = false;
    try {
      agents[0]->updateNeighborStates(1, shared);
    } catch (const std::runtime_error &e) {
      stale_refused = true;
    }
    ok = ok && stale_refused;
  }
  return ok ? 0 : 1;
}
