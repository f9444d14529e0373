// What the in-library exchange (exchange.h) needs from a session: the pose-graph session (rbcd.h) and the range-aided
// one (ra_rbcd.h) both provide it, so the same transport -- peer stores into IPC-mapped halo buffers, flag words and
// evaluation scalars in the shared segment -- carries both (ref src/Agent.cpp:113-152, 844-906: getSharedStateDicts /
// updateNeighborStates are the same calls on either graph type).
#pragma once
#include <hip/hip_runtime.h>

#include <vector>

namespace dcora {

struct XAgentView {
  bool hosted = false;            // lives on this rank
  int ncols = 0;                  // its PUBLIC columns of the mirror: the variables other agents' measurements reach
  const int *cols_dev = nullptr;  // their global column indices (device)
  const std::vector<int> *neighbors = nullptr;
};

class ExchangeSession {
 public:
  virtual ~ExchangeSession() {}
  virtual int x_num_agents() const = 0;
  virtual int x_rank_r() const = 0;     // relaxation rank: rows of the mirror
  virtual long x_num_cols() const = 0;  // columns of the mirror (the whole problem)
  virtual int x_rank() const = 0;
  virtual int x_world() const = 0;
  virtual int x_device() const = 0;
  virtual hipStream_t x_stream() const = 0;
  virtual double *x_mirror() = 0;  // r x cols, column-major: own columns current, the neighbours' public ones after wait
  virtual XAgentView x_agent(int a) const = 0;
  virtual int x_phase_nonselected(int selected) = 0;  // Agent::iterate(false) of the hosted non-selected agents
  virtual int x_phase_selected(int selected) = 0;     // Agent::iterate(true) where the selected agent lives
  // per hosted agent a: out[2a] = |Proj(X_a Q_aa + G_a)|^2, out[2a + 1] = <X_a, X_a Q_aa + G_a> (device, 2 R doubles)
  virtual int x_phase_evaluate_dev(double *out_dev) = 0;
  virtual int x_iterate_set(const int *set, int count, int allow_adjacent) = 0;
  virtual int x_set_X(const double *Xh) = 0;
  // the hosted agents' columns of the mirror into host_area (laid out like the mirror); synchronises
  virtual int x_stage_hosted(double *host_area) = 0;
};

}  // namespace dcora
