// The reference's multi-robot driver (ref examples/MultiRobotExample.cpp:44-307) written against the façade classes, in
// both shapes the façade offers, next to the same run through dcora_rbcd_iterate:
//   A  a team created from the dataset (DCORA::AgentTeam::create), the loop body of :223-307;
//   B  the driver's OWN construction lines: read_g2o_file, the partition into odometry / private / shared lists, a
//      central std::make_shared<DCORA::Graph>(0, r, d) + QuadraticProblem(graph), `new DCORA::Agent(robot, options)`,
//      setLiftingMatrix, setMeasurements, initialize, setX (:44-217);
// all three must produce the same block sequence and the same costs.  Then the hand-over semantics of
// Agent::updateNeighborStates (ref src/Agent.cpp:844-906): stale poses are used as handed, an agent that was handed
// only part of what it needs skips its optimisation.
// usage: test_agent_facade <file.g2o>.  Exit code 0 = pass, 2 = no GPU (the library has no CPU fallback), 1 = failure.
#include <cmath>
#include <cstdio>
#include <map>
#include <set>
#include <vector>

#include "DCORA/Agent.h"
#include "DCORA/DCORA_utils.h"
#include "DCORA/QuadraticProblem.h"

namespace {

struct Trace {
  std::vector<double> cost;
  std::vector<unsigned> selected;
};

// the loop body of the driver (:223-307) over any container of agent pointers
template <class Agents>
bool run_loop(Agents &agents, DCORA::QuadraticProblem &problemCentral, unsigned num_robots, unsigned n, unsigned d,
              unsigned r, unsigned numIters, Trace *tr) {
  const unsigned dh = d + 1, k = dh * n, per = n / num_robots;
  std::vector<unsigned> startIdx(num_robots), endIdx(num_robots);
  for (unsigned robot = 0; robot < num_robots; ++robot) {
    startIdx[robot] = robot * per;
    endIdx[robot] = (robot == num_robots - 1) ? n : (robot + 1) * per;
  }
  unsigned selectedRobot = 0;
  DCORA::Matrix Xopt(r, k);
  for (unsigned iter = 0; iter < numIters; ++iter) {
    auto &selectedRobotPtr = agents[selectedRobot];
    for (auto &robotPtr : agents) {
      if (robotPtr->iteration_number() != iter) {
        std::printf("iteration_number %u != %u\n", robotPtr->iteration_number(), iter);
        return false;
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
    if (!selectedRobotPtr->iterate(true)) {
      std::printf("iteration %u: the selected agent skipped its optimisation\n", iter);
      return false;
    }
    for (unsigned robot = 0; robot < num_robots; ++robot) {
      DCORA::Matrix XRobot;
      agents[robot]->getX(&XRobot);
      for (unsigned c = 0; c < XRobot.cols(); ++c)
        for (unsigned i = 0; i < r; ++i) Xopt(i, startIdx[robot] * dh + c) = XRobot(i, c);
    }
    const DCORA::Matrix RGrad = problemCentral.RieGrad(Xopt);
    tr->cost.push_back(2 * problemCentral.f(Xopt));
    tr->selected.push_back(selectedRobot);
    double best = -1;
    unsigned arg = 0;
    for (unsigned robot = 0; robot < num_robots; ++robot) {  // greedy selection (:288-305)
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
  return true;
}

DCORA::Matrix block_of(const DCORA::Matrix &X, unsigned r, unsigned c0, unsigned cols) {
  DCORA::Matrix Xb(r, cols);
  for (unsigned c = 0; c < cols; ++c)
    for (unsigned i = 0; i < r; ++i) Xb(i, c) = X(i, c0 + c);
  return Xb;
}

}  // namespace

int main(int argc, char **argv) {
  if (argc < 2) {
    std::printf("usage: %s file.g2o\n", argv[0]);
    return 1;
  }
  if (dcora_device_count() < 1) {
    std::printf("no GPU: facade compiled and linked, compute skipped\n");
    return 2;
  }
  // ---- the driver's preamble (:44-118) ----
  const DCORA::G2ODataset dataset = DCORA::read_g2o_file(argv[1]);
  const std::vector<DCORA::RelativePosePoseMeasurement> &measurements = dataset.pose_pose_measurements;
  const unsigned d = dataset.dim, n = dataset.num_poses;
  const unsigned num_robots = 5, r = 5, numIters = 40;
  const unsigned dh = d + 1, k = dh * n;
  const unsigned num_poses_per_robot = n / num_robots;
  std::set<unsigned> robot_IDs;
  for (unsigned i = 0; i < num_robots; ++i) robot_IDs.insert(i);
  std::map<unsigned, DCORA::PoseID> PoseMap;
  for (unsigned robot = 0; robot < num_robots; ++robot) {
    const unsigned startIdx = robot * num_poses_per_robot;
    const unsigned endIdx = (robot == num_robots - 1) ? n : (robot + 1) * num_poses_per_robot;
    for (unsigned idx = startIdx; idx < endIdx; ++idx) PoseMap[idx] = DCORA::PoseID(robot, idx - startIdx);
  }
  std::vector<std::vector<DCORA::RelativePosePoseMeasurement>> odometry(num_robots), private_loop_closures(num_robots),
      shared_loop_closure(num_robots);
  for (const auto &mIn : measurements) {
    const DCORA::PoseID src = PoseMap[(unsigned)mIn.p1], dst = PoseMap[(unsigned)mIn.p2];
    DCORA::RelativePosePoseMeasurement m(src.robot_id, dst.robot_id, src.frame_id, dst.frame_id, mIn.R, mIn.t, mIn.kappa,
                                         mIn.tau);
    if (src.robot_id == dst.robot_id) {
      if (src.frame_id + 1 == dst.frame_id)
        odometry[src.robot_id].push_back(m);
      else
        private_loop_closures[src.robot_id].push_back(m);
    } else {
      shared_loop_closure[src.robot_id].push_back(m);
      shared_loop_closure[dst.robot_id].push_back(m);
    }
  }
  // start point: chordal initialisation lifted to rank r (InitializationMethod::Chordal, :150-153)
  dcora_dataset_t ds;
  DCORA::check_status(dcora_dataset_load_g2o(argv[1], &ds), "load");
  std::vector<double> T((size_t)d * k);
  DCORA::check_status(dcora_dataset_chordal_init(ds, T.data()), "chordal");
  DCORA::Matrix Xcurr(r, k);
  for (unsigned c = 0; c < k; ++c)
    for (unsigned i = 0; i < d; ++i) Xcurr(i, c) = T[(size_t)c * d + i];

  // ---- the central problem, as the driver builds it (:173-177) ----
  std::shared_ptr<DCORA::Graph> poseGraphCurrRank = std::make_shared<DCORA::Graph>(0, r, d);
  poseGraphCurrRank->setMeasurements(measurements);
  DCORA::QuadraticProblem problemCentralCurrRank(poseGraphCurrRank);
  bool ok = poseGraphCurrRank->n() == n;

  // ---- A: team from the dataset ----
  Trace trA;
  auto setX_all = [&](auto &agents) {
    for (unsigned robot = 0; robot < num_robots; ++robot) {
      const unsigned startIdx = robot * num_poses_per_robot;
      const unsigned endIdx = (robot == num_robots - 1) ? n : (robot + 1) * num_poses_per_robot;
      agents[robot]->setX(block_of(Xcurr, r, startIdx * dh, (endIdx - startIdx) * dh));
    }
  };
  DCORA::AgentParameters optionsA(d, r, num_robots);
  optionsA.acceleration = true;
  auto team = DCORA::AgentTeam::create(ds, optionsA);
  setX_all(team->agents);
  ok = run_loop(team->agents, problemCentralCurrRank, num_robots, n, d, r, numIters, &trA) && ok;

  // ---- B: the driver's construction lines (:184-217) ----
  Trace trB;
  std::vector<DCORA::Agent *> agents;
  const DCORA::AgentTeamHandle handle = DCORA::makeAgentTeam();  // the one line the facade adds: no process-wide registry
  for (unsigned robot = 0; robot < num_robots; ++robot) {
    DCORA::AgentParameters options(d, r, robot_IDs);
    options.team = handle;
    options.acceleration = true;
    options.verbose = false;
    auto *agent = new DCORA::Agent(robot, options);
    if (robot > 0) {
      DCORA::Matrix M;
      agents[0]->getLiftingMatrix(&M);
      agent->setLiftingMatrix(M);
    }
    agent->setMeasurements(odometry[robot], private_loop_closures[robot], shared_loop_closure[robot]);
    agent->initialize();
    agents.push_back(agent);
  }
  setX_all(agents);
  ok = run_loop(agents, problemCentralCurrRank, num_robots, n, d, r, numIters, &trB) && ok;

  // ---- the same run through the session's own loop body ----
  dcora_rbcd_options o;
  dcora_rbcd_options_default(&o);
  o.num_robots = (int)num_robots;
  o.r = (int)r;
  o.acceleration = 1;
  dcora_rbcd_t s;
  DCORA::check_status(dcora_rbcd_create(ds, &o, &s), "session");
  DCORA::check_status(dcora_rbcd_set_X(s, Xcurr.data()), "set_X");
  int selected = 0;
  double worstA = 0, worstB = 0;
  for (unsigned iter = 0; ok && iter < numIters; ++iter) {
    double c2 = 0, gn = 0;
    int nxt = 0;
    DCORA::check_status(dcora_rbcd_iterate(s, selected, &c2, &gn, nullptr, &nxt), "iterate");
    if ((unsigned)selected != trA.selected[iter] || (unsigned)selected != trB.selected[iter]) {
      std::printf("iteration %u: block %d vs %u (team) / %u (agents)\n", iter, selected, trA.selected[iter],
                  trB.selected[iter]);
      ok = false;
      break;
    }
    worstA = std::fmax(worstA, std::fabs(c2 - trA.cost[iter]) / std::fabs(c2));
    worstB = std::fmax(worstB, std::fabs(c2 - trB.cost[iter]) / std::fabs(c2));
    selected = nxt;
  }
  dcora_rbcd_destroy(s);
  std::printf("agent facade: %u iterations, 2f %.6f -> %.6f, max relative cost difference to dcora_rbcd_iterate %.2e "
              "(team), %.2e (agents constructed one by one)\n",
              numIters, trA.cost.front(), trA.cost.back(), worstA, worstB);
  ok = ok && worstA < 1e-9 && worstB < 1e-9 && trA.cost.back() < trA.cost.front();

  // a second iterate(true) of the same agent inside one round is refused (lockstep)
  team->agents[0]->iterate(true);
  bool refused = false;
  try {
    team->agents[0]->iterate(true);
  } catch (const std::exception &e) {
    refused = true;
  }
  ok = ok && refused;

  // ---- hand-over semantics (ref src/Agent.cpp:844-906, 1234-1249) on fresh teams ----
  auto fresh = [&]() {
    DCORA::AgentParameters p(d, r, num_robots);  // non-accelerated: one cache in play
    auto t = DCORA::AgentTeam::create(ds, p);
    setX_all(t->agents);
    return t;
  };
  auto handover = [&](std::shared_ptr<DCORA::AgentTeam> &t, double bump, bool only_half) {
    for (unsigned nb = 0; nb < num_robots; ++nb) {
      if (nb == 1) continue;
      DCORA::PoseDict shared;
      t->agents[nb]->getSharedStateDicts(&shared);
      if (only_half && nb == 0) {
        size_t keep = shared.size() / 2;
        for (auto it = shared.begin(); it != shared.end();)
          it = (keep-- > 0) ? std::next(it) : shared.erase(it);
      }
      for (auto &kv : shared) kv.second(0, 0) += bump;
      t->agents[1]->updateNeighborStates(nb, shared);
    }
  };
  auto X_of = [&](std::shared_ptr<DCORA::AgentTeam> &t) {
    DCORA::Matrix X;
    t->agents[1]->getX(&X);
    return X;
  };
  auto diff = [](const DCORA::Matrix &A, const DCORA::Matrix &B) {
    double s = 0;
    for (size_t j = 0; j < A.cols(); ++j)
      for (size_t i = 0; i < A.rows(); ++i) s = std::fmax(s, std::fabs(A(i, j) - B(i, j)));
    return s;
  };
  {
    auto tCur = fresh(), tStale = fresh(), tMirror = fresh(), tHalf = fresh();
    handover(tCur, 0.0, false);    // the neighbours' current poses
    handover(tStale, 0.05, false); // altered ("stale") poses: used as handed, no complaint
    handover(tHalf, 0.0, true);    // agent 0's poses only in part: the cache of agent 1 stays incomplete
    const DCORA::Matrix X0 = X_of(tHalf);
    const bool okCur = tCur->agents[1]->iterate(true), okStale = tStale->agents[1]->iterate(true);
    const bool okMirror = tMirror->agents[1]->iterate(true);  // never handed anything: reads the shared mirror
    const bool okHalf = tHalf->agents[1]->iterate(true);
    const double dMirror = diff(X_of(tCur), X_of(tMirror)), dStale = diff(X_of(tCur), X_of(tStale)),
                 dHalf = diff(X0, X_of(tHalf));
    std::printf("hand-over: current vs mirror %.2e, current vs stale %.2e, partial hand-over: iterate %s, moved %.2e\n",
                dMirror, dStale, okHalf ? "ran" : "skipped", dHalf);
    ok = ok && okCur && okStale && okMirror && !okHalf && dMirror == 0.0 && dStale > 1e-6 && dHalf == 0.0;
    // poses of another robot / of the wrong shape are the reference's CHECK failures
    bool threw = false;
    try {
      DCORA::PoseDict bad;
      bad[DCORA::PoseID(2, 0)] = DCORA::Matrix(r, dh);
      tCur->agents[1]->updateNeighborStates(0, bad);
    } catch (const std::invalid_argument &e) {
      threw = true;
    }
    ok = ok && threw;
  }
  for (DCORA::Agent *a : agents) delete a;
  dcora_dataset_destroy(ds);
  return ok ? 0 : 1;
}
