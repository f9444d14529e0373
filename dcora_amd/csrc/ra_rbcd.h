// RBCD++ session for multi-robot range-aided SLAM: the agents of a merged pyfg problem, their device-resident state
// and the synchronous driver loop (replaces Agent::iterate / updateX / getSharedStateDicts / updateNeighborStates on
// the RangeAidedSLAMGraph, ref src/Agent.cpp:535-596, 1158-1278, src/Graph.cpp:824-1772, and the loop body of
// examples/MultiRobotExample_RASLAM.cpp).
//
// Variables are owned as the reference assigns them (poses by robot symbol, landmarks by symbol, unit spheres by the
// source robot of their range); an agent's columns are scattered over the global RA ordering, so each agent keeps its
// X / V / Y / XPrev in its own RA ordering [rotations | unit spheres | translations | landmarks] and the global
// iterate is a mirror that the coupling products (G_a = X_global C_a^T) and the central evaluation read.
#pragma once
#include <memory>
#include <vector>

#include "device_problem.h"
#include "exchange_session.h"
#include "host_graph.h"

namespace dcora {

struct RaAgentDev {
  int robot = 0;                 // robot id ('A' = 0, ...)
  int n = 0, l = 0, b = 0, k = 0;
  bool hosted = true;            // lives on this rank (agent i on rank i / ceil(R / world_size))
  DevBuf<int> public_cols;       // my columns of the global ordering that OTHER agents' measurements reach
  int n_public = 0;
  std::vector<int> neighbors;    // agents sharing a measurement with me
  std::vector<int> own_host;     // my columns in the global ordering (host copy: gather of the whole X)
  std::unique_ptr<DeviceProblem> prob;  // Q_aa, its preconditioner, solver workspace
  DevCsr coupling;                      // rows: my columns (my ordering), cols: global columns
  DevBuf<int> own;                      // my columns in the global ordering
  DevBuf<double> X, V, Y, XPrev, tmp;   // r x k, my ordering
  double reg = 0;
};

class RaRbcdSession : public ExchangeSession {
 public:
  int d = 0, r = 0, n = 0, l = 0, b = 0, k = 0, R = 0;
  dcora_rbcd_options opt{};
  hipStream_t st = nullptr;
  int device_of_stream_ = 0;
  std::vector<RaAgentDev> agents;
  std::unique_ptr<DeviceProblem> central;  // global Q: cost and Riemannian gradient of the merged problem
  DevBuf<double> Xg;                       // r x k global mirror (RA ordering)
  DevBuf<double> evalbuf;
  double gamma = 0, alpha = 0;
  int iteration = 0;
  DeviceProblem *last_solver = nullptr;
  double setup_ms = 0;

  ~RaRbcdSession();
  int init(const HostRADataset &ds, const dcora_rbcd_options &o);
  int set_X(const double *Xh);  // r x k global; V = Y = XPrev = X for every agent
  int get_X(double *Xh);
  // one pass of the driver's loop body with agent index `selected` (position in the sorted robot list)
  int iterate(int selected, double *cost2, double *gradnorm, double *block_norms, int *next_selected);
  int evaluate(double *cost2, double *gradnorm, double *block_norms, int *next_selected);
  int last_result(dcora_ropt_result *res);
  // the loop body in the phases the exchange interleaves with its posts and waits (one process per GPU)
  int phase_nonselected(int selected);
  int phase_selected(int selected);
  int phase_evaluate_dev(double *out_dev);

  // ExchangeSession
  int x_num_agents() const override { return R; }
  int x_rank_r() const override { return r; }
  long x_num_cols() const override { return k; }
  int x_rank() const override { return opt.rank; }
  int x_world() const override { return opt.world_size; }
  int x_device() const override { return opt.device; }
  hipStream_t x_stream() const override { return st; }
  double *x_mirror() override { return Xg.p; }
  XAgentView x_agent(int a) const override {
    const RaAgentDev &ag = agents[(size_t)a];
    XAgentView v;
    v.hosted = ag.hosted;
    v.ncols = ag.n_public;
    v.cols_dev = ag.public_cols.p;
    v.neighbors = &ag.neighbors;
    return v;
  }
  int x_phase_nonselected(int selected) override { return phase_nonselected(selected); }
  int x_phase_selected(int selected) override { return phase_selected(selected); }
  int x_phase_evaluate_dev(double *out_dev) override { return phase_evaluate_dev(out_dev); }
  int x_iterate_set(const int *, int, int) override;
  int x_set_X(const double *Xh) override { return set_X(Xh); }
  int x_stage_hosted(double *host_area) override;

 private:
  bool restart_now() const { return opt.acceleration && ((iteration + 1) % opt.restart_interval == 0); }
  int scatter(RaAgentDev &a);  // my X into the global mirror
  int solve(RaAgentDev &a, const double *start, double **result);
};

int device_precond_regularization(const HostCsr &Q, int device, double *reg);

}  // namespace dcora
