// RBCD++ session: the agents of one process, their device-resident state and the synchronous driver loop
// (replaces Agent::iterate / updateX / getSharedStateDicts / updateNeighborStates, ref src/Agent.cpp:113-152,
// 535-596, 844-906, 1158-1278, and the loop body of examples/MultiRobotExample.cpp:223-307).
#pragma once
#include <memory>
#include <string>
#include <vector>

#include "device_problem.h"
#include "exchange_session.h"
#include "host_graph.h"

namespace dcora {

struct AgentDev {
  int id = 0, n = 0, col0 = 0;  // poses, first global column
  bool hosted = false;
  bool v_feasible = false;  // V is the output of a projection since the agent's last setX
  std::unique_ptr<DeviceProblem> prob;  // Q_bb, (Q_bb + 0.1 I)^-1, solver workspace (hosted agents only)
  DevCsr coupling;                      // rows: local columns, cols: global columns (hosted agents only)
  std::vector<int> public_poses;        // global pose indices of my public poses (all agents)
  DevBuf<int> public_cols;              // their global columns, (d+1) per pose (all agents)
  std::vector<int> neighbors;           // agents sharing a measurement with me (all agents)
  hipStream_t own = nullptr;            // stream of my solve when several agents update at once (hosted agents only;
  hipEvent_t done = nullptr;            // both owned by the session)
  // Agent::updateNeighborStates (ref src/Agent.cpp:844-906): once an agent has been HANDED neighbour poses it
  // optimises against what it was handed -- its own cache of them (neighborPoseDict / neighborAuxPoseDict), stale or
  // not -- instead of the session's shared mirror.  required: the global poses of other agents my measurements
  // reach (Graph::requireNeighborPose); nbr[0 / 1]: my copies (plain / auxiliary), laid out like the mirror; got:
  // which required poses each cache holds.  An optimisation whose cache misses a required pose is skipped
  // (constructDataMatrices fails, ref src/Agent.cpp:1243-1249).
  std::vector<int> required;
  bool detached = false;
  DevBuf<double> nbr[2];
  std::vector<char> got[2];
  bool last_skipped = false;
};

class RbcdSession : public ExchangeSession {
 public:
  int d = 0, r = 0, n = 0, R = 1;
  Partition P;
  dcora_rbcd_options opt{};
  hipStream_t st = nullptr;
  ManiDesc mg{};  // global manifold (n poses)
  std::vector<AgentDev> agents;
  std::unique_ptr<DeviceProblem> central;  // global Q (evaluation); world_size == 1 only
  DevBuf<double> Xg, Vg, Yg, XPrevg;       // r x (d+1) n global mirrors
  DevBuf<int> col_start;                   // R + 1 global column offsets
  DevBuf<double> evalbuf, posenorm, eval_split;
  DevBuf<int> pose_start;          // R + 1 global pose offsets
  EvalOut *eval_host = nullptr;    // host-mapped results of the evaluation epilogue
  double *x_stage = nullptr;       // pinned staging buffer of get_X
  EvalOut *eval_dev = nullptr;
  int eval_seq = 0;
  DeviceProblem *last_solver = nullptr;
  double gamma = 0, alpha = 0;
  int iteration = 0;
  dcora_ropt_result last{};
  int last_result(dcora_ropt_result *res);
  double setup_ms = 0;

  ~RbcdSession();
  int init(const HostDataset &ds, const dcora_rbcd_options &o);
  int set_X(const double *Xh);
  int get_X(double *Xh);
  int set_acceleration(bool on);
  int phase_nonselected(int selected);
  int phase_selected(int selected);
  int evaluate_central(double *cost2, double *gradnorm, double *block_norms, int *next_selected);
  int phase_evaluate_dev(double *out_dev);
  int iterate(int selected, double *cost2, double *gradnorm, double *block_norms, int *next_selected);
  // simultaneous Agent::iterate(true) of a set of agents from one snapshot of the neighbour states
  int iterate_set(const int *set, int count, int allow_adjacent);
  // Agent::iterate(doOptimization) of one agent; the agents of a session advance in lockstep (one call per agent
  // and round, as the reference driver makes them)
  int agent_iterate(int agent, bool do_optimization);
  int agent_get_X(int agent, double *Xh);
  int agent_set_X(int agent, const double *Xh);
  // count poses of `neighbor` (frames local to it, each r x (d+1) column-major in `poses`) handed to `agent`
  int agent_update_neighbor(int agent, int neighbor, int count, const int *frames, const double *poses, bool aux);
  std::vector<int> agent_it;  // Agent::iteration_number() of every agent
  // greedy colouring of the agent graph: agents of one colour share no measurement
  int agent_colours(int *colours, int *ncolours) const;
  int pack_public(int agent, double *packed_dev);
  int unpack_public(int agent, const double *packed_dev);

  // ExchangeSession
  int x_num_agents() const override { return R; }
  int x_rank_r() const override { return r; }
  long x_num_cols() const override { return (long)(d + 1) * n; }
  int x_rank() const override { return opt.rank; }
  int x_world() const override { return opt.world_size; }
  int x_device() const override { return opt.device; }
  hipStream_t x_stream() const override { return st; }
  double *x_mirror() override { return Xg.p; }
  XAgentView x_agent(int a) const override {
    const AgentDev &ag = agents[(size_t)a];
    XAgentView v;
    v.hosted = ag.hosted;
    v.ncols = (int)ag.public_poses.size() * (d + 1);
    v.cols_dev = ag.public_cols.p;
    v.neighbors = &ag.neighbors;
    return v;
  }
  int x_phase_nonselected(int selected) override { return phase_nonselected(selected); }
  int x_phase_selected(int selected) override { return phase_selected(selected); }
  int x_phase_evaluate_dev(double *out_dev) override { return phase_evaluate_dev(out_dev); }
  int x_iterate_set(const int *set, int count, int allow_adjacent) override { return iterate_set(set, count, allow_adjacent); }
  int x_set_X(const double *Xh) override { return set_X(Xh); }
  int x_stage_hosted(double *host_area) override;

 private:
  bool restart_now() const { return opt.acceleration && ((iteration + 1) % opt.restart_interval == 0); }
  void advance_sequences();
  bool seq_advanced_ = false;
  int staged_selected_ = -1;  // agent whose Nesterov step rode in the non-selected agents' launch of this round
  int staged_iteration_ = -1; // the round it was staged in: honoured by update_selected_agent in that round only
  bool own_stream_ = true;
  bool pending_reset_ = false;  // gamma = alpha = 0 after a restart round, applied when the next round begins
  std::vector<char> set_marks_;  // agents that received Agent::setX since the last round
  int update_nonselected_agent(AgentDev &a, bool restart);
  int update_selected_agent(AgentDev &a, bool restart);
  hipEvent_t fork_ev_ = nullptr;
  int solve_block(AgentDev &a, std::string *err, bool serial = false);
};

}  // namespace dcora
