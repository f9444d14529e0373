// Agent with the reference's per-robot interface (ref include/DCORA/Agent.h:256-650) on top of the RBCD session of
// include/dcora_hip.h, so that the reference's driver (examples/MultiRobotExample.cpp:184-307) runs with its own lines:
//
//     DCORA::AgentParameters options(d, r, robot_IDs);                    // :186
//     auto *agent = new DCORA::Agent(robot, options);                     // :192
//     agent->setMeasurements(odometry[robot], private_loop_closures[robot], shared_loop_closure[robot]);   // :201
//     agent->initialize();                                                // :203
//     agents[robot]->setX(block of Xcurr);                                // :213
//     for (iter ...) {
//       for (auto *a : agents) if (a->getID() != selected) a->iterate(false);
//       ... getSharedStateDicts / updateNeighborStates (plain and auxiliary) ...
//       agents[selected]->iterate(true);
//       ... getX of every agent, central evaluation, greedy selection ...
//     }
//
// Object model (pose graphs).  The agents of one process share one device-resident session (one mirror of the lifted
// variable, one stream).  Agents that are constructed one by one, as the reference's driver does, meet through a TEAM
// HANDLE THE CALLER OWNS -- `options.team = DCORA::makeAgentTeam();` once, the same handle in every robot's
// AgentParameters: one added line, no process-wide state -- and the session is created when the LAST robot of
// AgentParameters::robotIDs has been initialised, from the union of the measurements the agents were given, which must
// follow the driver's contiguous partition (equal pose counts, the last robot takes the remainder; :56-118).  A single
// robot needs no handle.  Robots that live in other processes are not this class's business: a rank of a multi-process
// job creates its session with rank / world_size and exchanges through dcora_exchange_*.
// AgentTeam::create(dataset, params) remains as the direct way to the same state.
//
// Range-aided graphs (GraphType::RangeAidedSLAMGraph): every Agent is self-contained, as the reference's is
// (RangeAidedAgent.h) -- its measurements, its iterate in its own RA ordering, its acceleration, its caches of the
// neighbours' public states; robots talk through the three dictionaries of getSharedStateDicts / updateNeighborStates
// only.  MAP_ID is the passive map agent (ref src/Agent.cpp:541): one pose holding the lifting matrix, iterate() counts.
//
// updateNeighborStates hands poses to ONE agent: from then on that agent optimises against what it was handed (its own
// plain / auxiliary caches on the device: stale poses are used as given, poses it does not require are ignored, an
// incomplete cache skips the optimisation, ref src/Agent.cpp:844-906, 1234-1249).
#pragma once
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <string>
#include <utility>

#include "DCORA_types.h"
#include "Graph.h"
#include "QuadraticOptimizer.h"
#include "RangeAidedAgent.h"

namespace DCORA {

namespace detail {
struct PendingTeam;
}
// the handle through which Agents constructed one by one form their team (pose graphs, several robots in one process)
using AgentTeamHandle = std::shared_ptr<detail::PendingTeam>;
inline AgentTeamHandle makeAgentTeam();

// ref include/DCORA/Agent.h:149-200 (the fields the drivers read)
enum class AgentState { WAIT_FOR_DATA, WAIT_FOR_INITIALIZATION, INITIALIZED };
struct AgentStatus {
  unsigned agentID = 0;
  AgentState state = AgentState::WAIT_FOR_DATA;
  unsigned instanceNumber = 0, iterationNumber = 0;
  bool readyToTerminate = false;
  double relativeChange = 0;
};

// ref include/DCORA/Agent.h:40-147 (the fields the RBCD loop reads)
struct AgentParameters {
  unsigned d = 3, r = 5;
  std::set<unsigned> robotIDs;
  unsigned numRobots = 1;
  ROptParameters localOptimizationParams;
  bool acceleration = false;
  unsigned restartInterval = 30;
  bool verbose = false, logData = false;
  std::string logDirectory;
  int device = 0;
  GraphType graphType = GraphType::PoseGraph;
  AgentTeamHandle team;  // pose graphs, several robots: the caller's handle (makeAgentTeam()), the same in every robot's
  AgentParameters(unsigned dIn, unsigned rIn, const std::set<unsigned> &robotIDsIn,
                  GraphType graphTypeIn = GraphType::PoseGraph)
      : d(dIn), r(rIn), robotIDs(robotIDsIn), numRobots((unsigned)robotIDsIn.size()), graphType(graphTypeIn) {}
  AgentParameters(unsigned dIn, unsigned rIn, unsigned numRobotsIn) : d(dIn), r(rIn), numRobots(numRobotsIn) {
    for (unsigned i = 0; i < numRobotsIn; ++i) robotIDs.insert(i);
  }
};

class Agent;

// the Agents hosted by this process and the session that holds their state on the GPU
class AgentTeam : public std::enable_shared_from_this<AgentTeam> {
 public:
  static std::shared_ptr<AgentTeam> create(dcora_dataset_t dataset, const AgentParameters &params) {
    std::shared_ptr<AgentTeam> t(new AgentTeam(params));
    t->open(dataset);
    t->build_agents();
    return t;
  }
  ~AgentTeam() {
    dcora_rbcd_destroy(session_);
    if (owned_ds_) dcora_dataset_destroy(owned_ds_);
  }
  AgentTeam(const AgentTeam &) = delete;
  AgentTeam &operator=(const AgentTeam &) = delete;

  std::vector<std::shared_ptr<Agent>> agents;
  const AgentParameters &params() const { return params_; }
  dcora_rbcd_t session() const { return session_; }

 private:
  friend class Agent;
  explicit AgentTeam(const AgentParameters &p) : params_(p) {}
  void open(dcora_dataset_t dataset) {
    dcora_rbcd_options o;
    dcora_rbcd_options_default(&o);
    o.num_robots = (int)params_.numRobots;
    o.r = (int)params_.r;
    o.acceleration = params_.acceleration ? 1 : 0;
    o.restart_interval = (int)params_.restartInterval;
    o.local = params_.localOptimizationParams.c();
    o.device = params_.device;
    check_status(dcora_rbcd_create(dataset, &o, &session_), "AgentTeam");
  }
  void build_agents();
  AgentParameters params_;
  dcora_rbcd_t session_ = nullptr;
  dcora_dataset_t owned_ds_ = nullptr;
};

namespace detail {
// Agents constructed one by one (the reference's shape) meet here until their team is complete
struct PendingTeam {
  bool have_params = false;
  AgentParameters params{3, 5, 1u};
  std::map<unsigned, std::vector<RelativePosePoseMeasurement>> odometry, private_lc, shared_lc;
  std::set<unsigned> constructed, initialized;
  std::shared_ptr<AgentTeam> team;
  std::mutex mu;
};
}  // namespace detail

class Agent {
 public:
  // member of a team created from a dataset (AgentTeam::create)
  Agent(unsigned ID, const std::shared_ptr<AgentTeam> &team) : mID(ID), params_(team->params()), team_(team) {}
  // the reference's constructor (ref include/DCORA/Agent.h:256): the team forms when every robot of params.robotIDs
  // has been constructed, given its measurements and initialised in this process
  Agent(unsigned ID, const AgentParameters &params) : mID(ID), params_(params) {
    if (!params.robotIDs.count(ID)) throw std::invalid_argument("Agent: ID is not in AgentParameters::robotIDs");
    if (params.graphType == GraphType::RangeAidedSLAMGraph) {
      if (ID == MAP_ID) {  // the passive map agent: one pose that holds the lifting matrix (ref src/Agent.cpp:541)
        map_agent_ = true;
        return;
      }
      ra_ = std::make_shared<detail::RangeAidedAgentCore>(ID, params.d, params.r, params.numRobots, params.acceleration,
                                                          params.restartInterval, params.localOptimizationParams,
                                                          params.device);
      return;
    }
    std::shared_ptr<detail::PendingTeam> p = params.team;
    if (!p) {
      if (params.robotIDs.size() != 1)
        throw std::invalid_argument(
            "Agent: the robots of a pose graph that live in one process form their team through a handle the caller "
            "owns: `options.team = DCORA::makeAgentTeam();` once, the same handle in every robot's AgentParameters");
      p = makeAgentTeam();
    }
    std::lock_guard<std::mutex> lk(p->mu);
    if (!p->have_params) {
      p->params = params;
      p->params.team.reset();  // (the team does not keep itself alive)
      p->have_params = true;
    } else if (p->params.d != params.d || p->params.r != params.r || p->params.robotIDs != params.robotIDs ||
               p->params.acceleration != params.acceleration) {
      throw std::invalid_argument("Agent: the robots of one team must be constructed with the same d, r, robotIDs, acceleration");
    }
    if (!p->constructed.insert(ID).second) throw std::invalid_argument("Agent: this team already has that robot");
    pending_ = p;
  }
  unsigned getID() const { return mID; }
  unsigned relaxation_rank() const { return params_.r; }
  unsigned dimension() const { return params_.d; }
  unsigned num_poses() const { return map_agent_ ? 1 : ra_ ? ra_->n() : info().first; }
  unsigned num_unit_spheres() const { return ra_ ? ra_->l() : 0; }
  unsigned num_landmarks() const { return ra_ ? ra_->b() : 0; }
  unsigned problem_dimension() const { return (dimension() + 1) * num_poses() + num_unit_spheres() + num_landmarks(); }
  unsigned instance_number() const { return 0; }
  unsigned iteration_number() const {
    if (map_agent_) return map_iterations_;
    if (ra_) return ra_->iterations();
    int it = 0;
    check_status(dcora_rbcd_agent_info(session(), (int)mID, nullptr, nullptr, &it), "iteration_number");
    return (unsigned)it;
  }
  // ref include/DCORA/Agent.h:269-285 / src/Agent.cpp (setMeasurements): the three lists of the driver's partition
  void setMeasurements(const std::vector<RelativePosePoseMeasurement> &inputOdometry,
                       const std::vector<RelativePosePoseMeasurement> &inputPrivateLoopClosures,
                       const std::vector<RelativePosePoseMeasurement> &inputSharedLoopClosures) {
    if (!pending_) throw std::logic_error("Agent::setMeasurements: this agent belongs to a team created from a dataset");
    std::lock_guard<std::mutex> lk(pending_->mu);
    if (pending_->team) throw std::logic_error("Agent::setMeasurements: the team has been formed already");
    pending_->odometry[mID] = inputOdometry;
    pending_->private_lc[mID] = inputPrivateLoopClosures;
    pending_->shared_lc[mID] = inputSharedLoopClosures;
  }
  // ref include/DCORA/Agent.h:286-292: all relative measurements of a range-aided graph
  void setMeasurements(const RelativeMeasurements &measurements) {
    if (!ra_) throw std::logic_error("Agent::setMeasurements(RelativeMeasurements): the agent is on a pose graph");
    ra_->setMeasurements(measurements);
  }
  // ref include/DCORA/Agent.h:315-330, src/Agent.cpp:396-458: the start point from an estimate of the states in the
  // robot's frame (the tests hand over the ground truth), lifted by the shared YLift
  void initialize(const PoseArray *TInitPtr, const PointArray *UnitSphereInitPtr, const PointArray *LandmarkInitPtr) {
    if (!ra_) throw std::logic_error("Agent::initialize(states): the agent is on a pose graph");
    const unsigned d = dimension(), r = relaxation_rank(), n = num_poses(), l = num_unit_spheres(), b = num_landmarks();
    if (!TInitPtr || TInitPtr->n() != n || (l && (!UnitSphereInitPtr || UnitSphereInitPtr->n() != l)) ||
        (b && (!LandmarkInitPtr || LandmarkInitPtr->n() != b)))
      throw std::invalid_argument("Agent::initialize: the estimate does not match the graph's states");
    Matrix local(d, problem_dimension());  // RA ordering [R_1 .. R_n | s | t_1 .. t_n | L]
    for (unsigned i = 0; i < n; ++i)
      for (unsigned a = 0; a < d; ++a) {
        for (unsigned c = 0; c < d; ++c) local(a, (size_t)i * d + c) = TInitPtr->getData()(a, (size_t)i * (d + 1) + c);
        local(a, (size_t)d * n + l + i) = TInitPtr->getData()(a, (size_t)i * (d + 1) + d);
      }
    for (unsigned i = 0; i < l; ++i)
      for (unsigned a = 0; a < d; ++a) local(a, (size_t)d * n + i) = UnitSphereInitPtr->getData()(a, i);
    for (unsigned i = 0; i < b; ++i)
      for (unsigned a = 0; a < d; ++a) local(a, (size_t)(d + 1) * n + l + i) = LandmarkInitPtr->getData()(a, i);
    Matrix YLift(r, d), Tid(d, d + 1);
    check_status(dcora_fixed_stiefel_variable((int)r, (int)d, YLift.data()), "Agent::initialize");
    for (unsigned a = 0; a < d; ++a) Tid(a, a) = 1.0;
    dcora_dims dims{(int)r, (int)d, (int)n, (int)l, (int)b, DCORA_LAYOUT_RA};
    Matrix X0(r, problem_dimension());
    check_status(dcora_agent_initialize_in_global_frame(&dims, Tid.data(), local.data(), YLift.data(), X0.data()),
                 "Agent::initialize");
    ra_->setX(X0);
  }
  // ref src/Agent.cpp:950-1034: the estimate rounded to SE(d)^n x (S^{d-1})^l x R^{d b} in the frame of the first pose
  bool getStatesInLocalFrame(Matrix *Trajectory, Matrix *UnitSpheres, Matrix *Landmarks) {
    if (!ra_) throw std::logic_error("Agent::getStatesInLocalFrame(3): the agent is on a pose graph");
    if (!ra_->initialized()) return false;
    const unsigned d = dimension(), n = num_poses(), l = num_unit_spheres(), b = num_landmarks();
    dcora_dims dims{(int)relaxation_rank(), (int)d, (int)n, (int)l, (int)b, DCORA_LAYOUT_RA};
    Matrix T(d, (size_t)(d + 1) * n), S(d, l), L(d, b);
    check_status(dcora_round_align_trajectory(&dims, ra_->X().data(), nullptr, 0, T.data(), l ? S.data() : nullptr,
                                              b ? L.data() : nullptr, params_.device),
                 "getStatesInLocalFrame");
    if (Trajectory) *Trajectory = T;
    if (UnitSpheres) *UnitSpheres = S;
    if (Landmarks) *Landmarks = L;
    return true;
  }
  // ref src/Agent.cpp:598-648: back to the state before setMeasurements
  void reset() {
    if (map_agent_) {
      map_iterations_ = 0;
      return;
    }
    if (!ra_) return;
    ra_ = std::make_shared<detail::RangeAidedAgentCore>(mID, params_.d, params_.r, params_.numRobots, params_.acceleration,
                                                        params_.restartInterval, params_.localOptimizationParams,
                                                        params_.device);
  }
  // ref include/DCORA/Agent.h:315 (the trajectory / frame arguments of the reference's initialisation do not apply: the
  // driver sets X itself, :208-217).  The last robot to arrive forms the team.
  void initialize() {
    if (!pending_) return;  // (range-aided agents and the map: nothing to do, the driver sets X itself)
    std::lock_guard<std::mutex> lk(pending_->mu);
    pending_->initialized.insert(mID);
    if (!pending_->team && pending_->initialized == pending_->params.robotIDs) form_team(*pending_);
  }
  // ref include/DCORA/Agent.h:638-644: the lifting matrix all robots share (r x d, orthonormal columns).  The solver
  // on the device does not need it (the driver hands over lifted states); it is kept for callers that lift with it.
  bool getLiftingMatrix(Matrix *M) const {
    if (lifting_.rows() == 0) {
      Matrix Y(relaxation_rank(), dimension());
      for (unsigned i = 0; i < dimension(); ++i) Y(i, i) = 1.0;
      *M = Y;
    } else {
      *M = lifting_;
    }
    return true;
  }
  void setLiftingMatrix(const Matrix &M) {
    if (M.rows() != relaxation_rank() || M.cols() != dimension())
      throw std::invalid_argument("setLiftingMatrix: expected r x d");
    lifting_ = M;
  }
  // ref src/Agent.cpp:64-77 (also re-initialises the acceleration, :1178-1187)
  void setX(const Matrix &Xin) {
    if (map_agent_) throw std::logic_error("Agent::setX: the map agent holds the lifting matrix");
    if (Xin.rows() != relaxation_rank() || Xin.cols() != problem_dimension())
      throw std::invalid_argument("Agent::setX: expected r x (d+1) n");
    if (ra_) {
      ra_->setX(Xin);
      return;
    }
    check_status(dcora_rbcd_agent_set_X(session(), (int)mID, Xin.data()), "setX");
  }
  // ref src/Agent.cpp:98-105
  bool getX(Matrix *Mout) {
    if (map_agent_) {  // [YLift 0]: the fixed global frame, rotated by the lifting matrix
      Matrix Y;
      getLiftingMatrix(&Y);
      *Mout = Matrix(relaxation_rank(), dimension() + 1);
      for (unsigned c = 0; c < dimension(); ++c)
        for (unsigned a = 0; a < relaxation_rank(); ++a) (*Mout)(a, c) = Y(a, c);
      return true;
    }
    if (ra_) {
      *Mout = ra_->X();
      return ra_->initialized();
    }
    *Mout = Matrix(relaxation_rank(), problem_dimension());
    return dcora_rbcd_agent_get_X(session(), (int)mID, Mout->data()) == DCORA_OK;
  }
  // ref src/Agent.cpp:535-596; false when the optimisation was skipped because a required neighbour pose has never
  // been handed over (ref :1243-1249)
  bool iterate(bool doOptimization = true) {
    if (map_agent_) {  // passive: the iteration counter advances, nothing else (ref src/Agent.cpp:535-541)
      ++map_iterations_;
      return true;
    }
    if (ra_) return ra_->iterate(doOptimization);
    check_status(dcora_rbcd_agent_iterate(session(), (int)mID, doOptimization ? 1 : 0), "iterate");
    int skipped = 0;
    check_status(dcora_rbcd_agent_last_skipped(session(), (int)mID, &skipped), "iterate");
    return !(doOptimization && skipped);
  }
  // ref src/Agent.cpp:113-152: my public poses (those with an inter-robot measurement), keyed (robot, local frame)
  bool getSharedStateDicts(PoseDict *poseDict, UnitSphereDict *unitSphereDict = nullptr,
                           LandmarkDict *landmarkDict = nullptr) {
    if (map_agent_) {  // the map shares nothing
      poseDict->clear();
      if (unitSphereDict) unitSphereDict->clear();
      if (landmarkDict) landmarkDict->clear();
      return true;
    }
    if (ra_) {
      if (!ra_->initialized()) return false;
      ra_->sharedStates(poseDict, unitSphereDict, landmarkDict);
      return true;
    }
    if (unitSphereDict || landmarkDict)
      throw std::invalid_argument("getSharedStateDicts: a pose graph has neither unit spheres nor landmarks");
    int cnt = 0;
    check_status(dcora_rbcd_public_count(session(), (int)mID, &cnt), "getSharedStateDicts");
    std::vector<int> idx((size_t)(cnt > 0 ? cnt : 1));
    check_status(dcora_rbcd_public_indices(session(), (int)mID, idx.data()), "getSharedStateDicts");
    Matrix X;
    if (!getX(&X)) return false;
    const unsigned r = relaxation_rank(), dh = dimension() + 1, first = info().second;
    poseDict->clear();
    for (int q = 0; q < cnt; ++q) {
      const unsigned local = (unsigned)idx[(size_t)q] - first;
      Matrix P(r, dh);
      for (unsigned c = 0; c < dh; ++c)
        for (unsigned i = 0; i < r; ++i) P(i, c) = X(i, local * dh + c);
      (*poseDict)[PoseID(mID, local)] = P;
    }
    return true;
  }
  // ref src/Agent.cpp:844-906: the poses go into this agent's own cache on the device (the plain one, or the auxiliary
  // one it reads when it optimises from Y); the reference's CHECKs on the robot id and the shapes throw here
  void updateNeighborStates(unsigned neighborID, const PoseDict &poseDict, bool areNeighborStatesAux = false,
                            const UnitSphereDict &unitSphereDict = UnitSphereDict(),
                            const LandmarkDict &landmarkDict = LandmarkDict()) {
    if (neighborID == mID) throw std::invalid_argument("updateNeighborStates: neighborID is this agent");
    if (map_agent_) return;
    if (ra_) {
      ra_->updateNeighborStates(neighborID, poseDict, areNeighborStatesAux, unitSphereDict, landmarkDict);
      return;
    }
    if (!unitSphereDict.empty() || !landmarkDict.empty())
      throw std::invalid_argument("updateNeighborStates: a pose graph has neither unit spheres nor landmarks");
    if (poseDict.empty()) return;
    const unsigned r = relaxation_rank(), dh = dimension() + 1;
    std::vector<int> frames;
    std::vector<double> poses;
    frames.reserve(poseDict.size());
    poses.reserve(poseDict.size() * r * dh);
    for (const auto &kv : poseDict) {
      if (kv.first.robot_id != neighborID) throw std::invalid_argument("updateNeighborStates: pose of another robot");
      if (kv.second.rows() != r || kv.second.cols() != dh)
        throw std::invalid_argument("updateNeighborStates: expected r x (d+1) poses");
      frames.push_back((int)kv.first.frame_id);
      poses.insert(poses.end(), kv.second.data(), kv.second.data() + (size_t)r * dh);
    }
    check_status(dcora_rbcd_agent_update_neighbor(session(), (int)mID, (int)neighborID, (int)frames.size(),
                                                  frames.data(), poses.data(), areNeighborStatesAux ? 1 : 0),
                 "updateNeighborStates");
  }
  // ref include/DCORA/Agent.h:380-400: what the drivers pass around with the dictionaries
  AgentStatus getStatus() {
    AgentStatus st;
    st.agentID = mID;
    st.state = (map_agent_ || !ra_ || ra_->initialized()) ? AgentState::INITIALIZED : AgentState::WAIT_FOR_INITIALIZATION;
    st.iterationNumber = iteration_number();
    return st;
  }
  void setNeighborStatus(const AgentStatus &status) { neighbor_status_[status.agentID] = status; }
  // ref src/Agent.cpp:535 getSharedPose(index): pose `index` of this agent, r x (d+1)
  bool getSharedPose(unsigned index, Matrix *Mout) {
    Matrix X;
    if (index >= num_poses() || !getX(&X)) return false;
    const unsigned r = relaxation_rank(), dh = dimension() + 1;
    *Mout = Matrix(r, dh);
    for (unsigned c = 0; c < dh; ++c)
      for (unsigned i = 0; i < r; ++i) (*Mout)(i, c) = X(i, index * dh + c);
    return true;
  }

 private:
  std::shared_ptr<AgentTeam> team() const {
    if (pending_) {
      std::lock_guard<std::mutex> lk(pending_->mu);
      if (!pending_->team)
        throw std::logic_error(
            "Agent: its team is not complete -- every robot of AgentParameters::robotIDs must have been constructed, "
            "given its measurements and initialised in this process (robots of other processes: a session per rank "
            "with rank / world_size and dcora_exchange_*)");
      return pending_->team;
    }
    std::shared_ptr<AgentTeam> t = team_.lock();
    if (!t) throw std::runtime_error("Agent outlived its AgentTeam");
    return t;
  }
  dcora_rbcd_t session() const { return team()->session(); }
  // (number of poses, first global pose)
  std::pair<unsigned, unsigned> info() const {
    if (!have_info_) {
      int np = 0, first = 0;
      check_status(dcora_rbcd_agent_info(session(), (int)mID, &np, &first, nullptr), "Agent");
      n_ = (unsigned)np;
      first_pose_ = (unsigned)first;
      have_info_ = true;
    }
    return {n_, first_pose_};
  }
  // the team of agents that were constructed one by one: the union of their measurements as one dataset in global pose
  // numbering (the driver's contiguous partition, ref examples/MultiRobotExample.cpp:56-118), one session
  static void form_team(detail::PendingTeam &p) {
    const AgentParameters &prm = p.params;
    const unsigned R = prm.numRobots, d = prm.d;
    unsigned want = 0;
    for (unsigned id : prm.robotIDs)
      if (id != want++) throw std::invalid_argument("AgentTeam: robot IDs must be 0 .. numRobots - 1");
    std::vector<unsigned> np(R, 0);
    auto count = [&](const std::vector<RelativePosePoseMeasurement> &v) {
      for (const RelativePosePoseMeasurement &m : v) {
        if (m.r1 >= R || m.r2 >= R) throw std::invalid_argument("AgentTeam: measurement names an unknown robot");
        np[m.r1] = std::max<unsigned>(np[m.r1], (unsigned)m.p1 + 1);
        np[m.r2] = std::max<unsigned>(np[m.r2], (unsigned)m.p2 + 1);
      }
    };
    for (unsigned b = 0; b < R; ++b) {
      count(p.odometry[b]);
      count(p.private_lc[b]);
      count(p.shared_lc[b]);
    }
    unsigned n = 0;
    std::vector<unsigned> start(R + 1, 0);
    for (unsigned b = 0; b < R; ++b) {
      start[b] = n;
      n += np[b];
    }
    start[R] = n;
    const unsigned per = n / R;
    for (unsigned b = 0; b + 1 < R; ++b)
      if (np[b] != per)
        throw std::invalid_argument(
            "AgentTeam: the robots' pose counts do not follow the contiguous partition of the driver (n / numRobots "
            "poses each, the last robot takes the remainder)");
    std::vector<RelativePosePoseMeasurement> all;
    auto take = [&](const std::vector<RelativePosePoseMeasurement> &v, unsigned owner, bool shared) {
      for (const RelativePosePoseMeasurement &m : v) {
        if (shared && m.r1 != owner) continue;  // a shared closure sits in both robots' lists: counted once
        RelativePosePoseMeasurement g = m;
        g.p1 = start[m.r1] + m.p1;
        g.p2 = start[m.r2] + m.p2;
        g.r1 = g.r2 = 0;
        all.push_back(g);
      }
    };
    for (unsigned b = 0; b < R; ++b) {
      take(p.odometry[b], b, false);
      take(p.private_lc[b], b, false);
      take(p.shared_lc[b], b, true);
    }
    std::vector<int> ids;
    std::vector<double> vals;
    pack_measurements(all, d, &ids, &vals);
    dcora_dataset_t ds = nullptr;
    check_status(dcora_dataset_create((int)d, (int)n, (int)all.size(), ids.data(), vals.data(), &ds), "AgentTeam");
    std::shared_ptr<AgentTeam> t(new AgentTeam(prm));
    t->owned_ds_ = ds;
    t->open(ds);
    p.team = t;
  }

  std::shared_ptr<detail::RangeAidedAgentCore> ra_;  // an agent on a range-aided graph (RangeAidedAgent.h)
  bool map_agent_ = false;
  unsigned map_iterations_ = 0;
  std::map<unsigned, AgentStatus> neighbor_status_;
  unsigned mID;
  AgentParameters params_;
  mutable unsigned n_ = 0, first_pose_ = 0;
  mutable bool have_info_ = false;
  std::weak_ptr<AgentTeam> team_;
  std::shared_ptr<detail::PendingTeam> pending_;
  Matrix lifting_;
};

inline AgentTeamHandle makeAgentTeam() { return std::make_shared<detail::PendingTeam>(); }

inline void AgentTeam::build_agents() {
  for (unsigned id = 0; id < params_.numRobots; ++id) agents.push_back(std::make_shared<Agent>(id, shared_from_this()));
}

}  // namespace DCORA
