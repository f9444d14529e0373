// Agent with the reference's per-robot interface (ref include/DCORA/Agent.h:256-650) on top of the RBCD session of
// include/dcora_hip.h, so that the reference's driver loop (examples/MultiRobotExample.cpp:184-307) runs unchanged:
//
//     auto team = DCORA::AgentTeam::create(dataset_handle, params);      // Agents of this process + their device state
//     for (auto &a : team->agents) a->setX(block of Xcurr);
//     for (iter ...) {
//       for (auto &a : team->agents) if (a->getID() != selected) a->iterate(false);
//       ... getSharedStateDicts / updateNeighborStates ...
//       team->agents[selected]->iterate(true);
//       ... getX of every agent, central evaluation, greedy selection ...
//     }
//
// What differs from the reference object model: the agents of one process share one device-resident mirror of the
// lifted variable (the session), so a neighbour's public poses are visible to the selected agent without a copy --
// getSharedStateDicts still hands out the poses (for callers that ship them to other processes), updateNeighborStates
// checks what it is given against the mirror's layout and stores it.  Measurements come from the dataset the team is
// created from (the contiguous partition of the driver, :56-118) instead of three setMeasurements lists.
#pragma once
#include <map>
#include <memory>
#include <utility>

#include "DCORA_types.h"

namespace DCORA {

// ref include/DCORA/Agent.h:40-147 (the fields the RBCD loop reads)
struct AgentParameters {
  unsigned d = 3, r = 5, numRobots = 1;
  ROptParameters localOptimizationParams;
  bool acceleration = false;
  unsigned restartInterval = 30;
  int device = 0;
  AgentParameters(unsigned dIn, unsigned rIn, unsigned numRobotsIn) : d(dIn), r(rIn), numRobots(numRobotsIn) {}
};

// ref include/DCORA/DCORA_types.h (PoseID = (robot, frame)); a lifted pose is r x (d+1)
using PoseID = std::pair<unsigned, unsigned>;
using PoseDict = std::map<PoseID, Matrix>;

class Agent;

// the Agents hosted by this process and the session that holds their state on the GPU
class AgentTeam : public std::enable_shared_from_this<AgentTeam> {
 public:
  static std::shared_ptr<AgentTeam> create(dcora_dataset_t dataset, const AgentParameters &params) {
    std::shared_ptr<AgentTeam> t(new AgentTeam(params));
    dcora_rbcd_options o;
    dcora_rbcd_options_default(&o);
    o.num_robots = (int)params.numRobots;
    o.r = (int)params.r;
    o.acceleration = params.acceleration ? 1 : 0;
    o.restart_interval = (int)params.restartInterval;
    o.local = params.localOptimizationParams.c();
    o.device = params.device;
    check_status(dcora_rbcd_create(dataset, &o, &t->session_), "AgentTeam");
    t->build_agents();
    return t;
  }
  ~AgentTeam() { dcora_rbcd_destroy(session_); }
  AgentTeam(const AgentTeam &) = delete;
  AgentTeam &operator=(const AgentTeam &) = delete;

  std::vector<std::shared_ptr<Agent>> agents;
  const AgentParameters &params() const { return params_; }
  dcora_rbcd_t session() const { return session_; }

 private:
  explicit AgentTeam(const AgentParameters &p) : params_(p) {}
  void build_agents();
  AgentParameters params_;
  dcora_rbcd_t session_ = nullptr;
};

class Agent {
 public:
  Agent(unsigned ID, const std::shared_ptr<AgentTeam> &team) : mID(ID), team_(team) {
    int np = 0, first = 0;
    check_status(dcora_rbcd_agent_info(team->session(), (int)ID, &np, &first, nullptr), "Agent");
    n_ = (unsigned)np;
    first_pose_ = (unsigned)first;
  }
  unsigned getID() const { return mID; }
  unsigned relaxation_rank() const { return team()->params().r; }
  unsigned dimension() const { return team()->params().d; }
  unsigned num_poses() const { return n_; }
  unsigned problem_dimension() const { return (dimension() + 1) * n_; }
  unsigned instance_number() const { return 0; }
  unsigned iteration_number() const {
    int it = 0;
    check_status(dcora_rbcd_agent_info(team()->session(), (int)mID, nullptr, nullptr, &it), "iteration_number");
    return (unsigned)it;
  }
  // ref src/Agent.cpp:64-77 (also re-initialises the acceleration, :1178-1187)
  void setX(const Matrix &Xin) {
    if (Xin.rows() != relaxation_rank() || Xin.cols() != problem_dimension())
      throw std::invalid_argument("Agent::setX: expected r x (d+1) n");
    check_status(dcora_rbcd_agent_set_X(team()->session(), (int)mID, Xin.data()), "setX");
  }
  // ref src/Agent.cpp:98-105
  bool getX(Matrix *Mout) {
    *Mout = Matrix(relaxation_rank(), problem_dimension());
    return dcora_rbcd_agent_get_X(team()->session(), (int)mID, Mout->data()) == DCORA_OK;
  }
  // ref src/Agent.cpp:535-596
  bool iterate(bool doOptimization = true) {
    check_status(dcora_rbcd_agent_iterate(team()->session(), (int)mID, doOptimization ? 1 : 0), "iterate");
    return true;
  }
  // ref src/Agent.cpp:113-152: my public poses (those with an inter-robot measurement), keyed (robot, local frame)
  bool getSharedStateDicts(PoseDict *poseDict) {
    int cnt = 0;
    check_status(dcora_rbcd_public_count(team()->session(), (int)mID, &cnt), "getSharedStateDicts");
    std::vector<int> idx((size_t)(cnt > 0 ? cnt : 1));
    check_status(dcora_rbcd_public_indices(team()->session(), (int)mID, idx.data()), "getSharedStateDicts");
    Matrix X;
    if (!getX(&X)) return false;
    const unsigned r = relaxation_rank(), dh = dimension() + 1;
    poseDict->clear();
    for (int q = 0; q < cnt; ++q) {
      const unsigned local = (unsigned)idx[(size_t)q] - first_pose_;
      Matrix P(r, dh);
      for (unsigned c = 0; c < dh; ++c)
        for (unsigned i = 0; i < r; ++i) P(i, c) = X(i, local * dh + c);
      (*poseDict)[PoseID(mID, local)] = P;
    }
    return true;
  }
  // ref src/Agent.cpp:844-906.  Restriction of this facade: the agents of a team share ONE device-resident mirror of
  // X, and the selected agent's linear term is built from that mirror -- not from a per-agent copy of what it was
  // handed.  The hand-over is therefore CHECKED against the mirror: poses that are stale, altered or belong to
  // another robot are refused (std::runtime_error) instead of being silently ignored; frames that are left out are
  // still read from the mirror.  (Ranks in different processes exchange through dcora_exchange_* instead.)
  void updateNeighborStates(unsigned neighborID, const PoseDict &poseDict, bool areNeighborStatesAux = false) {
    (void)areNeighborStatesAux;  // the reference driver hands over X in both calls (examples/MultiRobotExample.cpp:252)
    if (poseDict.empty()) return;
    int np = 0;
    check_status(dcora_rbcd_agent_info(team()->session(), (int)neighborID, &np, nullptr, nullptr),
                 "updateNeighborStates");
    const unsigned r = relaxation_rank(), dh = dimension() + 1;
    Matrix Xn(r, dh * (unsigned)np);
    check_status(dcora_rbcd_agent_get_X(team()->session(), (int)neighborID, Xn.data()), "updateNeighborStates");
    for (const auto &kv : poseDict) {
      if (kv.first.first != neighborID) throw std::invalid_argument("updateNeighborStates: pose of another robot");
      if (kv.second.rows() != r || kv.second.cols() != dh)
        throw std::invalid_argument("updateNeighborStates: expected r x (d+1) poses");
      const unsigned frame = kv.first.second;
      if (frame >= (unsigned)np) throw std::invalid_argument("updateNeighborStates: frame out of range");
      for (unsigned c = 0; c < dh; ++c)
        for (unsigned i = 0; i < r; ++i)
          if (kv.second(i, c) != Xn(i, frame * dh + c))
            throw std::runtime_error(
                "updateNeighborStates: the poses handed over differ from the neighbour's current state; agents of one "
                "AgentTeam optimise against the shared device mirror and cannot be given stale or altered poses");
    }
  }
  // ref src/Agent.cpp:535 getSharedPose(index): pose `index` of this agent, r x (d+1)
  bool getSharedPose(unsigned index, Matrix *Mout) {
    Matrix X;
    if (index >= n_ || !getX(&X)) return false;
    const unsigned r = relaxation_rank(), dh = dimension() + 1;
    *Mout = Matrix(r, dh);
    for (unsigned c = 0; c < dh; ++c)
      for (unsigned i = 0; i < r; ++i) (*Mout)(i, c) = X(i, index * dh + c);
    return true;
  }

 private:
  std::shared_ptr<AgentTeam> team() const {
    std::shared_ptr<AgentTeam> t = team_.lock();
    if (!t) throw std::runtime_error("Agent outlived its AgentTeam");
    return t;
  }
  unsigned mID, n_ = 0, first_pose_ = 0;
  std::weak_ptr<AgentTeam> team_;
};

inline void AgentTeam::build_agents() {
  for (unsigned id = 0; id < params_.numRobots; ++id) agents.push_back(std::make_shared<Agent>(id, shared_from_this()));
}

}  // namespace DCORA
