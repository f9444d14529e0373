// RBCD++ session for multi-robot range-aided SLAM on the device (see ra_rbcd.h).
#include "ra_rbcd.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <set>

#include "cert.h"

namespace dcora {

int device_precond_regularization(const HostCsr &Q, int device, double *reg) {
  *reg = 1e-1;  // default when the eigenvalue computation is unsuccessful (ref src/Graph.cpp:1923, 1932-1939)
  DeviceLanczos L;
  int rc = L.init(Q, device);
  if (rc) return rc;
  LanczosResult e;
  rc = L.largest_magnitude(0.0, std::min(6, Q.n), 10000, 1e-3, nullptr, 1, &e);
  if (rc) return rc;
  if (e.ok && e.lambda > 0) *reg = e.lambda / (1e6 - 1);
  return DCORA_OK;
}

RaRbcdSession::~RaRbcdSession() {
  agents.clear();
  central.reset();
  if (st) stream_release(device_of_stream_, st);
}

int RaRbcdSession::init(const HostRADataset &ds, const dcora_rbcd_options &o) {
  const auto t0 = std::chrono::steady_clock::now();
  opt = o;
  d = ds.d;
  n = ds.n;
  l = ds.l;
  b = ds.b;
  k = ds.k();
  r = o.r;
  if (o.world_size < 1 || o.rank < 0 || o.rank >= o.world_size) {
    set_last_error("ra_rbcd: bad rank / world_size");
    return DCORA_ERR_BAD_ARG;
  }
  if (r < d || r > 16) {
    set_last_error("ra_rbcd: need d <= r <= 16");
    return DCORA_ERR_BAD_ARG;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    set_last_error("no HIP device available: libdcora_hip has no CPU fallback");
    return DCORA_ERR_NO_DEVICE;
  }
  DCORA_HIP(hipSetDevice(o.device));
  // agents = the robots that own poses, in id order; every unit sphere and landmark must belong to one of them
  // (the reference's map agent is passive, src/Agent.cpp:541: variables it would own are never optimised)
  std::set<int> robots(ds.pose_robot.begin(), ds.pose_robot.end());
  for (int q : ds.sphere_robot)
    if (!robots.count(q)) {
      set_last_error("ra_rbcd: a unit sphere is owned by a robot without poses (map agent): not optimisable here");
      return DCORA_ERR_UNSUPPORTED;
    }
  for (int q : ds.landmark_robot)
    if (!robots.count(q)) {
      set_last_error("ra_rbcd: a landmark is owned by a robot without poses (map agent): not optimisable here");
      return DCORA_ERR_UNSUPPORTED;
    }
  R = (int)robots.size();
  if (R < 1 || R > kMaxAgents) {
    set_last_error("ra_rbcd: bad number of robots");
    return DCORA_ERR_UNSUPPORTED;
  }
  {
    const int rcs = stream_acquire(o.device, &st);
    if (rcs) return rcs;
    device_of_stream_ = o.device;
  }
  const HostCsr Q = build_Q_ra(ds);
  const size_t N = (size_t)r * k;
  DCORA_HIP(Xg.alloc(N));
  DCORA_HIP(hipMemset(Xg.p, 0, sizeof(double) * N));
  DCORA_HIP(evalbuf.alloc(R + 8));
  agents.resize(R);
  const int per = (R + o.world_size - 1) / o.world_size;  // consecutive agents share a rank, as in the pose-graph session
  std::vector<int> owner_of((size_t)k, -1);               // global column -> agent
  std::vector<std::set<int>> pub((size_t)R), nbr((size_t)R);
  int idx = 0;
  for (int robot : robots) {
    RaAgentDev &a = agents[idx];
    a.robot = robot;
    a.hosted = (idx / per) == o.rank;
    int dims3[3];
    ra_agent_columns(ds, robot, dims3, a.own_host);
    for (int c : a.own_host) owner_of[(size_t)c] = idx;
    ++idx;
  }
  idx = 0;
  for (int robot : robots) {
    RaAgentDev &a = agents[idx];
    const int me = idx++;
    int dims3[3];
    std::vector<int> own;
    ra_agent_columns(ds, robot, dims3, own);
    a.n = dims3[0];
    a.l = dims3[1];
    a.b = dims3[2];
    a.k = (int)own.size();
    HostCsr Qaa, C;
    extract_agent_blocks(Q, own, &Qaa, &C);
    // the columns of OTHER agents my coupling block reaches are their public variables; their owners my neighbours
    for (int c : C.ci) {
      const int q = owner_of[(size_t)c];
      if (q >= 0 && q != me) {
        pub[(size_t)q].insert(c);
        nbr[(size_t)me].insert(q);
        nbr[(size_t)q].insert(me);
      }
    }
    if (!a.hosted) continue;  // agents of other ranks: their columns and public variables are all this rank keeps
    int rc = device_precond_regularization(Qaa, o.device, &a.reg);  // ref src/Graph.cpp:1901-1960, per agent
    if (rc) return rc;
    a.prob.reset(new DeviceProblem);
    dcora_dims dims{r, d, a.n, a.l, a.b, DCORA_LAYOUT_RA};  // an agent without ranges / landmarks keeps the RA ordering
    rc = a.prob->init(dims, Qaa, nullptr, a.reg, o.device, st);
    if (rc) return rc;
    rc = a.coupling.upload(C);
    if (rc) return rc;
    DCORA_HIP(a.own.alloc(own.size()));
    DCORA_HIP(hipMemcpy(a.own.p, own.data(), sizeof(int) * own.size(), hipMemcpyHostToDevice));
    const size_t Na = (size_t)r * a.k;
    DCORA_HIP(a.X.alloc(Na));
    DCORA_HIP(a.V.alloc(Na));
    DCORA_HIP(a.Y.alloc(Na));
    DCORA_HIP(a.XPrev.alloc(Na));
    DCORA_HIP(a.tmp.alloc(Na));
  }
  for (int i = 0; i < R; ++i) {
    RaAgentDev &a = agents[(size_t)i];
    a.neighbors.assign(nbr[(size_t)i].begin(), nbr[(size_t)i].end());
    std::vector<int> pc(pub[(size_t)i].begin(), pub[(size_t)i].end());
    a.n_public = (int)pc.size();
    DCORA_HIP(a.public_cols.alloc(std::max<size_t>(pc.size(), 1)));
    if (!pc.empty()) DCORA_HIP(hipMemcpy(a.public_cols.p, pc.data(), sizeof(int) * pc.size(), hipMemcpyHostToDevice));
  }
  if (o.world_size == 1) {  // the central evaluation of the one-process loop; ranks evaluate agent by agent
    central.reset(new DeviceProblem);
    dcora_dims dims{r, d, n, l, b, DCORA_LAYOUT_RA};
    int rc = central->init(dims, Q, nullptr, -1.0, o.device, st);
    if (rc) return rc;
  }
  iteration = 0;
  gamma = alpha = 0;
  setup_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  return DCORA_OK;
}

int RaRbcdSession::set_X(const double *Xh) {
  DCORA_HIP(hipSetDevice(opt.device));
  const size_t B = sizeof(double) * (size_t)r * k;
  DCORA_HIP(hipMemcpyAsync(Xg.p, Xh, B, hipMemcpyHostToDevice, st));
  for (RaAgentDev &a : agents) {
    if (!a.hosted) continue;
    const size_t Ba = sizeof(double) * (size_t)r * a.k;
    launch_gather_cols(st, r, a.k, a.own.p, Xg.p, a.X.p);
    DCORA_HIP(hipMemcpyAsync(a.V.p, a.X.p, Ba, hipMemcpyDeviceToDevice, st));
    DCORA_HIP(hipMemcpyAsync(a.Y.p, a.X.p, Ba, hipMemcpyDeviceToDevice, st));
    DCORA_HIP(hipMemcpyAsync(a.XPrev.p, a.X.p, Ba, hipMemcpyDeviceToDevice, st));
  }
  DCORA_HIP(hipStreamSynchronize(st));
  gamma = alpha = 0;
  iteration = 0;
  return DCORA_OK;
}

int RaRbcdSession::get_X(double *Xh) {
  DCORA_HIP(hipSetDevice(opt.device));
  DCORA_HIP(hipMemcpyAsync(Xh, Xg.p, sizeof(double) * (size_t)r * k, hipMemcpyDeviceToHost, st));
  DCORA_HIP(hipStreamSynchronize(st));
  return DCORA_OK;
}

int RaRbcdSession::scatter(RaAgentDev &a) {
  launch_scatter_cols(st, r, a.k, a.own.p, a.X.p, Xg.p);
  return DCORA_OK;
}

// updateX(doOptimization = true): G from the neighbours' public states in the mirror, local solve from `start`
// (ref src/Agent.cpp:1216-1278, src/Graph.cpp:1190-1772)
int RaRbcdSession::solve(RaAgentDev &a, const double *start, double **result) {
  DeviceProblem &pb = *a.prob;
  launch_spmm(st, r, a.coupling.view(), buf1(Xg.p), 0, nullptr, buf1(pb.G.p), 0, nullptr, Gate{});
  pb.has_G = true;
  DCORA_HIP(hipMemcpyAsync(pb.X0.p, start, sizeof(double) * (size_t)r * a.k, hipMemcpyDeviceToDevice, st));
  Buf2 Xres{{nullptr, nullptr}};
  const SolverCtl *cs = nullptr;
  int rc = pb.optimize_dev(opt.local, &Xres, &cs);
  if (rc) return rc;
  if (cs) {  // a solver that keeps its choice of buffer on the device: read it back
    dcora_ropt_result tmp;
    rc = pb.fetch_result(&tmp);
    if (rc) return rc;
    *result = pb.result_index() ? pb.X1.p : pb.X0.p;
  } else {
    *result = Xres.p[0];
  }
  last_solver = &pb;
  return DCORA_OK;
}

// Agent::iterate(false) of the hosted non-selected agents (ref src/Agent.cpp:535-551, 1202-1214): the shared Nesterov
// sequences advance once per round on every rank alike
int RaRbcdSession::phase_nonselected(int selected) {
  if (selected < 0 || selected >= R) {
    set_last_error("ra_rbcd: selected agent out of range");
    return DCORA_ERR_BAD_ARG;
  }
  DCORA_HIP(hipSetDevice(opt.device));
  iteration++;
  const bool accel = opt.acceleration != 0;
  if (accel) {  // updateGamma / updateAlpha (ref src/Agent.cpp:1189-1200)
    gamma = (1 + std::sqrt(1 + 4.0 * R * R * gamma * gamma)) / (2.0 * R);
    alpha = 1.0 / (gamma * R);
  }
  const bool restart = restart_now();
  // Y = proj((1 - alpha) X + alpha V), X = Y, V = proj(V)
  for (int i = 0; i < R; ++i) {
    if (i == selected) continue;
    RaAgentDev &a = agents[i];
    if (!a.hosted) continue;
    const ManiDesc &m = a.prob->m;
    const size_t Ba = sizeof(double) * (size_t)r * a.k;
    DCORA_HIP(hipMemcpyAsync(a.XPrev.p, a.X.p, Ba, hipMemcpyDeviceToDevice, st));
    if (!accel) continue;
    if (restart) {  // restartNesterovAcceleration(false): X = XPrev, V = Y = X
      DCORA_HIP(hipMemcpyAsync(a.V.p, a.X.p, Ba, hipMemcpyDeviceToDevice, st));
      DCORA_HIP(hipMemcpyAsync(a.Y.p, a.X.p, Ba, hipMemcpyDeviceToDevice, st));
    } else {
      launch_polar(st, m, 1.0 - alpha, a.X.p, alpha, a.V.p, 0.0, nullptr, a.Y.p);
      DCORA_HIP(hipMemcpyAsync(a.X.p, a.Y.p, Ba, hipMemcpyDeviceToDevice, st));
      launch_polar(st, m, 1.0, a.V.p, 0.0, nullptr, 0.0, nullptr, a.V.p);
      scatter(a);
    }
  }
  return DCORA_OK;
}

// Agent::iterate(true) of the selected agent, where it lives
int RaRbcdSession::phase_selected(int selected) {
  if (selected < 0 || selected >= R) {
    set_last_error("ra_rbcd: selected agent out of range");
    return DCORA_ERR_BAD_ARG;
  }
  DCORA_HIP(hipSetDevice(opt.device));
  const bool accel = opt.acceleration != 0;
  const bool restart = restart_now();
  RaAgentDev &a = agents[selected];
  if (a.hosted) {
    const ManiDesc &m = a.prob->m;
    const size_t Ba = sizeof(double) * (size_t)r * a.k;
    DCORA_HIP(hipMemcpyAsync(a.XPrev.p, a.X.p, Ba, hipMemcpyDeviceToDevice, st));
    double *res = nullptr;
    if (accel) {
      launch_polar(st, m, 1.0 - alpha, a.X.p, alpha, a.V.p, 0.0, nullptr, a.Y.p);
      int rc = solve(a, a.Y.p, &res);
      if (rc) return rc;
      DCORA_HIP(hipMemcpyAsync(a.X.p, res, Ba, hipMemcpyDeviceToDevice, st));
      launch_polar(st, m, 1.0, a.V.p, gamma, a.X.p, -gamma, a.Y.p, a.V.p);  // V = proj(V + gamma (X - Y))
      if (restart) {  // X = XPrev; updateX(true, false); V = Y = X
        rc = solve(a, a.XPrev.p, &res);
        if (rc) return rc;
        DCORA_HIP(hipMemcpyAsync(a.X.p, res, Ba, hipMemcpyDeviceToDevice, st));
        DCORA_HIP(hipMemcpyAsync(a.V.p, a.X.p, Ba, hipMemcpyDeviceToDevice, st));
        DCORA_HIP(hipMemcpyAsync(a.Y.p, a.X.p, Ba, hipMemcpyDeviceToDevice, st));
      }
    } else {
      int rc = solve(a, a.X.p, &res);
      if (rc) return rc;
      DCORA_HIP(hipMemcpyAsync(a.X.p, res, Ba, hipMemcpyDeviceToDevice, st));
    }
    scatter(a);
  }
  if (restart) gamma = alpha = 0;
  return DCORA_OK;
}

int RaRbcdSession::iterate(int selected, double *cost2, double *gradnorm, double *block_norms, int *next_selected) {
  if (opt.world_size != 1) {
    set_last_error("ra_rbcd: with one process per GPU the loop body is dcora_exchange_rbcd_iterate");
    return DCORA_ERR_UNSUPPORTED;
  }
  int rc = phase_nonselected(selected);
  if (rc) return rc;
  rc = phase_selected(selected);
  if (rc) return rc;
  int nxt = selected;
  rc = evaluate(cost2, gradnorm, block_norms, &nxt);
  if (rc) return rc;
  if (next_selected) *next_selected = (agents[selected].coupling.nnz > 0) ? nxt : selected;
  return DCORA_OK;
}

// distributed form of the evaluation: per hosted agent |Proj(X_a Q_aa + G_a)|^2 and <X_a, X_a Q_aa + G_a>, G_a from the
// neighbours' public variables in the mirror
int RaRbcdSession::phase_evaluate_dev(double *out_dev) {
  DCORA_HIP(hipSetDevice(opt.device));
  DCORA_HIP(hipMemsetAsync(out_dev, 0, sizeof(double) * 2 * R, st));
  for (int i = 0; i < R; ++i) {
    RaAgentDev &a = agents[i];
    if (!a.hosted) continue;
    DeviceProblem &pb = *a.prob;
    launch_spmm(st, r, a.coupling.view(), buf1(Xg.p), 0, nullptr, buf1(pb.G.p), 0, nullptr, Gate{});
    pb.has_G = true;
    pb.enqueue_egrad(a.X.p, pb.EG1.p, nullptr);
    launch_rgrad(st, pb.m, buf1(a.X.p), buf1(pb.EG1.p), buf1(pb.RG1.p), Buf2{{nullptr, nullptr}}, 0, pb.pB.p, Gate{});
    launch_sum_partials(st, pb.pB.p, pb.npPose(), 1, 1, out_dev + 2 * i);
    launch_dot(st, pb.nelem(), a.X.p, pb.EG1.p, pb.p3.p);
    launch_sum_partials(st, pb.p3.p, pb.npVec(), 1, 1, out_dev + 2 * i + 1);
  }
  return DCORA_OK;
}

int RaRbcdSession::x_iterate_set(const int *, int, int) {
  set_last_error("ra_rbcd: simultaneous updates are a pose-graph session feature");
  return DCORA_ERR_UNSUPPORTED;
}

int RaRbcdSession::x_stage_hosted(double *host_area) {
  DCORA_HIP(hipSetDevice(opt.device));
  std::vector<double> whole((size_t)r * k);
  DCORA_HIP(hipMemcpyAsync(whole.data(), Xg.p, sizeof(double) * whole.size(), hipMemcpyDeviceToHost, st));
  DCORA_HIP(hipStreamSynchronize(st));
  for (const RaAgentDev &a : agents) {
    if (!a.hosted) continue;
    for (int c : a.own_host)
      std::copy(&whole[(size_t)c * r], &whole[(size_t)c * r] + r, host_area + (size_t)c * r);
  }
  return DCORA_OK;
}

// central evaluation of the driver: 2 f, |rgrad| of the merged problem, per-agent |rgrad_a|, greedy selection
int RaRbcdSession::evaluate(double *cost2, double *gradnorm, double *block_norms, int *next_selected) {
  if (!central) {
    set_last_error("ra_rbcd: the central evaluation needs world_size == 1 (ranks: dcora_exchange_evaluate)");
    return DCORA_ERR_UNSUPPORTED;
  }
  DCORA_HIP(hipSetDevice(opt.device));
  DeviceProblem &c = *central;
  c.enqueue_egrad(Xg.p, c.EG0.p, c.pA.p);
  launch_rgrad(st, c.m, buf1(Xg.p), buf1(c.EG0.p), buf1(c.RG0.p), Buf2{{nullptr, nullptr}}, 0, c.pB.p, Gate{});
  launch_sum_partials(st, c.pA.p, c.npA(), 2, 2, evalbuf.p);
  launch_sum_partials(st, c.pB.p, c.npPose(), 1, 1, evalbuf.p + 2);
  for (int i = 0; i < R; ++i) {
    RaAgentDev &a = agents[i];
    const long Na = (long)r * a.k;
    launch_gather_cols(st, r, a.k, a.own.p, c.RG0.p, a.tmp.p);
    launch_dot(st, Na, a.tmp.p, a.tmp.p, c.p3.p);
    launch_sum_partials(st, c.p3.p, vec_grid(Na), 1, 1, evalbuf.p + 3 + i);
  }
  std::vector<double> h(R + 3);
  DCORA_HIP(hipMemcpyAsync(h.data(), evalbuf.p, sizeof(double) * (R + 3), hipMemcpyDeviceToHost, st));
  DCORA_HIP(hipStreamSynchronize(st));
  double best = -1;
  int arg = 0;
  for (int i = 0; i < R; ++i) {
    const double nb = std::sqrt(h[3 + i]);
    if (block_norms) block_norms[i] = nb;
    if (nb > best) {
      best = nb;
      arg = i;
    }
  }
  if (cost2) *cost2 = 2.0 * (0.5 * h[0] + h[1]);
  if (gradnorm) *gradnorm = std::sqrt(h[2]);
  if (next_selected) *next_selected = arg;
  return DCORA_OK;
}

int RaRbcdSession::last_result(dcora_ropt_result *res) {
  if (!last_solver) {
    set_last_error("ra_rbcd: no local solve yet");
    return DCORA_ERR_BAD_ARG;
  }
  return last_solver->fetch_result(res);
}

}  // namespace dcora
