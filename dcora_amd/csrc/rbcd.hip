// RBCD++ session on the device (see rbcd.h).
#include "rbcd.h"
#include "env.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <set>
#include <thread>

namespace dcora {

static bool group_kernels(const ManiDesc &m) {
  return group_supported(m) && !env::generic_solver();
}

static void nesterov(hipStream_t st, const ManiDesc &m, int mode, int restart, int skip_lo, int skip_hi, double alpha,
                     double gamma, double *X, double *V, double *Y, double *XPrev, double *Yloc, Buf2 Xloc,
                     const SolverCtl *ctl) {
  if (group_kernels(m))
    launch_g_nesterov(st, m, mode, restart, skip_lo, skip_hi, alpha, gamma, X, V, Y, XPrev, Yloc, Xloc, ctl);
  else  // thread-per-pose kernels: the caller resolved the result pointer (ctl == nullptr)
    launch_nesterov(st, m, mode, restart & 1, skip_lo, skip_hi, alpha, gamma, X, V, Y, XPrev, Yloc, Xloc.p[0]);
}

// The thread-per-pose Nesterov kernels take one result pointer: when the solver left its choice of buffer on the
// device (ctl), read it back first.  (The 8-lanes-per-pose kernels pick on the device.)
static int resolve_pick(DeviceProblem &pb, Buf2 *Xres, const SolverCtl **cs) {
  if (*cs == nullptr || group_kernels(pb.m)) return DCORA_OK;
  dcora_ropt_result tmp;
  const int rc = pb.fetch_result(&tmp);
  if (rc) return rc;
  double *picked = pb.result_index() ? pb.X1.p : pb.X0.p;
  *Xres = Buf2{{picked, picked}};
  *cs = nullptr;
  return DCORA_OK;
}

RbcdSession::~RbcdSession() {
  if (eval_host) (void)hipHostFree((void *)eval_host);
  if (x_stage) (void)hipHostFree((void *)x_stage);
  for (AgentDev &a : agents) {
    if (a.own) stream_release(opt.device, a.own);
    if (a.done) (void)hipEventDestroy(a.done);
  }
  if (fork_ev_) (void)hipEventDestroy(fork_ev_);
  agents.clear();
  central.reset();
  if (st && own_stream_) stream_release(opt.device, st);
}

int RbcdSession::init(const HostDataset &ds, const dcora_rbcd_options &o) {
  const auto t0 = std::chrono::steady_clock::now();
  opt = o;
  d = ds.d;
  n = ds.n;
  r = o.r;
  R = o.num_robots;
  if (R < 1 || n / R < 1 || o.world_size < 1 || o.rank < 0 || o.rank >= o.world_size) {
    set_last_error("rbcd: bad num_robots / rank / world_size");
    return DCORA_ERR_BAD_ARG;
  }
  if (o.acceleration && o.restart_interval < 1) {
    set_last_error("rbcd: acceleration needs restart_interval >= 1 (AgentParameters::restartInterval)");
    return DCORA_ERR_BAD_ARG;
  }
  if (r < d || r > 16) {
    set_last_error("rbcd: relaxation rank must satisfy d <= r <= 16");
    return DCORA_ERR_BAD_ARG;
  }
  P.R = R;
  P.n = n;
  P.per = n / R;
  const int dh = d + 1;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    set_last_error("no HIP device available: libdcora_hip has no CPU fallback");
    return DCORA_ERR_NO_DEVICE;
  }
  DCORA_HIP(hipSetDevice(o.device));
  if (o.stream) {
    st = (hipStream_t)o.stream;
    own_stream_ = false;
  } else {
    const int rcs = stream_acquire(o.device, &st);
    if (rcs) return rcs;
  }
  mg = make_mani(r, d, n, 0, 0);
  const size_t N = (size_t)r * dh * n;
  DCORA_HIP(Xg.alloc(N));
  DCORA_HIP(Vg.alloc(N));
  DCORA_HIP(Yg.alloc(N));
  DCORA_HIP(XPrevg.alloc(N));
  if (R > kMaxAgents) {
    set_last_error("rbcd: more agents than kMaxAgents");
    return DCORA_ERR_UNSUPPORTED;
  }
  DCORA_HIP(posenorm.alloc(n));
  DCORA_HIP(eval_split.alloc((size_t)eval_split_doubles()));
  DCORA_HIP(hipHostMalloc((void **)&eval_host, sizeof(EvalOut), hipHostMallocMapped));
  std::memset((void *)eval_host, 0, sizeof(EvalOut));
  DCORA_HIP(hipHostGetDevicePointer((void **)&eval_dev, (void *)eval_host, 0));
  DCORA_HIP(hipHostMalloc((void **)&x_stage, sizeof(double) * N, hipHostMallocDefault));
  // the first device-to-host copy of a process pays ~8 ms of runtime set-up: pay it here, not in get_X
  DCORA_HIP(hipMemcpyAsync(x_stage, Xg.p, sizeof(double) * N, hipMemcpyDeviceToHost, st));
  DCORA_HIP(hipStreamSynchronize(st));
  DCORA_HIP(evalbuf.alloc(2 * R + 16));
  DCORA_HIP(hipMemset(evalbuf.p, 0, sizeof(double) * (2 * R + 16)));

  if (env::init_timing())
    fprintf(stderr, "[session] buffers after %.1f ms\n",
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
  // partition (ref examples/MultiRobotExample.cpp:56-118)
  std::vector<std::vector<PoseMeas>> touching(R);
  std::vector<std::set<int>> pub(R), nb(R), req(R);
  for (const PoseMeas &mi : ds.meas) {
    PoseMeas e = mi;
    e.r1 = P.robot_of(mi.p1);
    e.r2 = P.robot_of(mi.p2);
    e.p1 = mi.p1 - P.start(e.r1);
    e.p2 = mi.p2 - P.start(e.r2);
    touching[e.r1].push_back(e);
    if (e.r2 != e.r1) {
      touching[e.r2].push_back(e);
      pub[e.r1].insert(mi.p1);
      pub[e.r2].insert(mi.p2);
      req[e.r1].insert(mi.p2);
      req[e.r2].insert(mi.p1);
      nb[e.r1].insert(e.r2);
      nb[e.r2].insert(e.r1);
    }
  }
  std::vector<PoseMeas> global = ds.meas;
  for (PoseMeas &e : global) e.r1 = e.r2 = 0;

  agents.resize(R);
  std::vector<int> cs(R + 1);
  std::vector<int> hosted_ids;
  for (int b = 0; b < R; ++b) {
    AgentDev &a = agents[b];
    a.id = b;
    a.n = P.end(b) - P.start(b);
    a.col0 = P.start(b) * dh;
    cs[b] = a.col0;
    a.hosted = (b / ((R + o.world_size - 1) / o.world_size)) == o.rank;  // consecutive agents share a rank
    a.public_poses.assign(pub[b].begin(), pub[b].end());
    a.neighbors.assign(nb[b].begin(), nb[b].end());
    a.required.assign(req[b].begin(), req[b].end());
    std::vector<int> cols;
    for (int p : a.public_poses)
      for (int c = 0; c < dh; ++c) cols.push_back(p * dh + c);
    DCORA_HIP(a.public_cols.alloc(std::max<size_t>(cols.size(), 1)));
    if (!cols.empty())
      DCORA_HIP(hipMemcpy(a.public_cols.p, cols.data(), sizeof(int) * cols.size(), hipMemcpyHostToDevice));
    if (a.hosted) hosted_ids.push_back(b);
  }
  // the pose / column offsets of the agents: uploaded before the builds (a copy queued behind them waited 16 ms)
  cs[R] = dh * n;
  {
    std::vector<int> ps(R + 1);
    for (int b = 0; b <= R; ++b) ps[b] = cs[b] / dh;
    DCORA_HIP(pose_start.alloc(R + 1));
    DCORA_HIP(col_start.alloc(R + 1));
    DCORA_HIP(hipMemcpyAsync(pose_start.p, ps.data(), sizeof(int) * (R + 1), hipMemcpyHostToDevice, st));
    DCORA_HIP(hipMemcpyAsync(col_start.p, cs.data(), sizeof(int) * (R + 1), hipMemcpyHostToDevice, st));
    DCORA_HIP(hipStreamSynchronize(st));
  }
  DCORA_HIP(hipEventCreateWithFlags(&fork_ev_, hipEventDisableTiming));
  // The hosted agents' problems (Q_bb, its preconditioner: host factorisation + inverse image, coupling block) are
  // built side by side on host threads: the factorisation of one block is partly serial (the separators above the
  // sub-trees), so eight blocks of the 100k lattice take 9 s one after the other and 2-3 s together.
  // Blocks small enough for the dense inverse: all hosted agents' matrices first, their inverses in ONE batch of launches
  // (five builds on five streams do not overlap: precond_prebuild_dense), then the problems attach to the cached images
  std::vector<HostCsr> Qbs((size_t)R);
  if ((long)(n / R + 1) * dh <= kDensePrecondMaxK && hosted_ids.size() > 1) {
    std::vector<const HostCsr *> ptrs;
    {
      std::vector<std::thread> th;  // (a matrix takes a millisecond of host work: side by side)
      for (int b : hosted_ids) th.emplace_back([&, b] { Qbs[(size_t)b] = build_Q_pgo(d, agents[b].n, b, touching[b]); });
      for (std::thread &t : th) t.join();
    }
    for (int b : hosted_ids) ptrs.push_back(&Qbs[(size_t)b]);
    const int prc = precond_prebuild_dense(ptrs, 0.1, dh, o.device);
    if (prc) return prc;
    if (env::init_timing())
      fprintf(stderr, "[session] dense inverses prebuilt after %.1f ms\n",
              std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
  }
  auto build_agent = [&](int b, std::string *err) -> int {
    AgentDev &a = agents[b];
    if (hipSetDevice(o.device) != hipSuccess) return DCORA_ERR_HIP;
    HostCsr Qb = Qbs[(size_t)b].n > 0 ? std::move(Qbs[(size_t)b]) : build_Q_pgo(d, a.n, b, touching[b]);
    a.prob.reset(new DeviceProblem);
    dcora_dims dims{r, d, a.n, 0, 0};
    int rc = a.prob->init(dims, Qb, nullptr, 0.1, o.device, st);  // reg = 1e-1, ref src/Graph.cpp:1906
    if (!rc) {
      HostCsr C = build_coupling_pgo(d, P, b, global);
      rc = a.coupling.upload(C);
    }
    if (!rc) rc = stream_acquire(o.device, &a.own);
    if (!rc && hipEventCreateWithFlags(&a.done, hipEventDisableTiming) != hipSuccess) rc = DCORA_ERR_HIP;
    if (rc && err) *err = dcora_last_error();
    return rc;
  };
  // the whole-graph problem of the evaluation (Q of all poses, no preconditioner) depends on none of the agents: its
  // assembly (0.15 s of one host thread for the 100k lattice) and upload run beside the agents' builds
  int central_rc = DCORA_OK;
  std::string central_err;
  auto build_central = [&]() {
    if (o.world_size != 1) return;
    if (hipSetDevice(o.device) != hipSuccess) {
      central_rc = DCORA_ERR_HIP;
      central_err = "hipSetDevice failed";
      return;
    }
    HostCsr Qc = build_Q_pgo(d, n, 0, global);
    central.reset(new DeviceProblem);
    dcora_dims dims{r, d, n, 0, 0};
    central_rc = central->init(dims, Qc, nullptr, -1.0, o.device, st);
    if (central_rc) central_err = dcora_last_error();
  };
  {
    const size_t nh = hosted_ids.size();
    std::vector<int> rcs(nh, DCORA_OK);
    std::vector<std::string> errs(nh);
    constexpr bool serial = false;
    // small blocks build in well under a millisecond of host work each: threads only pay off for large ones
    if (nh > 1 && !serial && (long)(n / R) * dh >= 1024) {
      std::atomic<size_t> next(0);
      auto worker = [&] {
        for (;;) {
          const size_t i = next.fetch_add(1);
          if (i >= nh) break;
          rcs[i] = build_agent(hosted_ids[i], &errs[i]);
        }
      };
      std::vector<std::thread> th;
      th.emplace_back(build_central);
      for (size_t t = 1; t < std::min<size_t>(nh, 8); ++t) th.emplace_back(worker);
      worker();
      for (std::thread &t : th) t.join();
    } else {
      for (size_t i = 0; i < nh; ++i) rcs[i] = build_agent(hosted_ids[i], &errs[i]);
      build_central();
    }
    for (size_t i = 0; i < nh; ++i)
      if (rcs[i]) {
        set_last_error(errs[i]);
        return rcs[i];
      }
  }
  const bool init_timing = env::init_timing();
  if (init_timing)
    fprintf(stderr, "[session] agents built after %.1f ms\n",
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
  if (central_rc) {
    set_last_error(central_err);
    return central_rc;
  }
  if (init_timing)
    fprintf(stderr, "[session] ready after %.1f ms\n",
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
  iteration = 0;
  gamma = alpha = 0;
  setup_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  return DCORA_OK;
}

// Agent::setX + initializeAcceleration for every agent (ref src/Agent.cpp:64-77, 1178-1187)
int RbcdSession::set_X(const double *Xh) {
  staged_selected_ = -1;  // a step staged by an interrupted round does not survive a new start point
  DCORA_HIP(hipSetDevice(opt.device));
  const size_t B = sizeof(double) * (size_t)r * (d + 1) * n;
  DCORA_HIP(hipMemcpyAsync(Xg.p, Xh, B, hipMemcpyHostToDevice, st));
  DCORA_HIP(hipMemcpyAsync(Vg.p, Xg.p, B, hipMemcpyDeviceToDevice, st));
  DCORA_HIP(hipMemcpyAsync(Yg.p, Xg.p, B, hipMemcpyDeviceToDevice, st));
  DCORA_HIP(hipMemcpyAsync(XPrevg.p, Xg.p, B, hipMemcpyDeviceToDevice, st));
  DCORA_HIP(hipStreamSynchronize(st));
  gamma = alpha = 0;
  iteration = 0;
  seq_advanced_ = false;
  pending_reset_ = false;
  agent_it.assign(R, 0);
  set_marks_.assign(R, 0);
  for (AgentDev &a : agents) a.v_feasible = false;  // V = X as handed over: projected in the next round
  return DCORA_OK;
}
// initializeAcceleration / acceleration off for every agent (ref src/Agent.cpp:1178-1187)
int RbcdSession::set_acceleration(bool on) {
  staged_selected_ = -1;
  DCORA_HIP(hipSetDevice(opt.device));
  opt.acceleration = on ? 1 : 0;
  const size_t B = sizeof(double) * (size_t)r * (d + 1) * n;
  DCORA_HIP(hipMemcpyAsync(Vg.p, Xg.p, B, hipMemcpyDeviceToDevice, st));
  DCORA_HIP(hipMemcpyAsync(Yg.p, Xg.p, B, hipMemcpyDeviceToDevice, st));
  DCORA_HIP(hipMemcpyAsync(XPrevg.p, Xg.p, B, hipMemcpyDeviceToDevice, st));
  gamma = alpha = 0;
  iteration = 0;
  seq_advanced_ = false;
  pending_reset_ = false;
  agent_it.assign(R, 0);
  set_marks_.assign(R, 0);
  for (AgentDev &a : agents) a.v_feasible = false;
  return DCORA_OK;
}
int RbcdSession::get_X(double *Xh) {
  DCORA_HIP(hipSetDevice(opt.device));
  const size_t B = sizeof(double) * (size_t)r * (d + 1) * n;
  // through a pinned staging buffer: an asynchronous copy into pageable memory takes milliseconds on this runtime
  if (!x_stage) DCORA_HIP(hipHostMalloc((void **)&x_stage, B, hipHostMallocDefault));
  DCORA_HIP(hipMemcpyAsync(x_stage, Xg.p, B, hipMemcpyDeviceToHost, st));
  DCORA_HIP(hipStreamSynchronize(st));
  std::memcpy(Xh, x_stage, B);
  return DCORA_OK;
}

// updateGamma / updateAlpha (ref src/Agent.cpp:1189-1200); the sequences are data-independent and identical for
// every agent, so they live on the host
void RbcdSession::advance_sequences() {
  iteration++;
  if (opt.acceleration) {
    gamma = (1 + std::sqrt(1 + 4.0 * R * R * gamma * gamma)) / (2.0 * R);
    alpha = 1.0 / (gamma * R);
  }
  seq_advanced_ = true;
}

// Agent::iterate(false) for every hosted agent except `selected`
int RbcdSession::phase_nonselected(int selected) {
  if (selected < 0 || selected >= R) {
    set_last_error("rbcd: selected agent out of range");
    return DCORA_ERR_BAD_ARG;
  }
  DCORA_HIP(hipSetDevice(opt.device));
  advance_sequences();
  set_marks_.assign(R, 0);
  staged_selected_ = -1;
  if (!opt.acceleration) return DCORA_OK;
  // bit 1: after the first round every V is the output of a projection (or a copy of X): skip its re-projection.
  // An Agent::setX of a single agent (which may hand over an X that is not exactly feasible) clears the flag until
  // the next round has projected every V again.
  bool all_feasible = true;
  for (const AgentDev &a : agents)
    if (a.id != selected) all_feasible = all_feasible && a.v_feasible;
  const int restart = (restart_now() ? 1 : 0) | (all_feasible ? 2 : 0);
  if (opt.world_size == 1) {
    // one launch over the whole graph; the selected agent's poses take their own step of the same kernel (Y and the
    // local solver's start point <- proj((1 - alpha) X + alpha V), what update_selected_agent would launch next)
    if (group_kernels(mg) && agents[selected].hosted && agents[selected].prob) {
      launch_g_nesterov(st, mg, 0, restart, P.start(selected), P.end(selected), alpha, gamma, Xg.p, Vg.p, Yg.p, XPrevg.p,
                        nullptr, Buf2{{nullptr, nullptr}}, nullptr, agents[selected].prob->X0.p);
      staged_selected_ = selected;
      staged_iteration_ = iteration;
    } else {
      nesterov(st, mg, 0, restart, P.start(selected), P.end(selected), alpha, gamma, Xg.p, Vg.p, Yg.p, XPrevg.p, nullptr,
               Buf2{{nullptr, nullptr}}, nullptr);
    }
    for (AgentDev &a : agents)
      if (a.id != selected) a.v_feasible = true;
  } else {
    for (AgentDev &a : agents) {
      if (!a.hosted || a.id == selected) continue;
      const int rc = update_nonselected_agent(a, (restart & 1) != 0);
      if (rc) return rc;
    }
  }
  return DCORA_OK;
}

int RbcdSession::update_nonselected_agent(AgentDev &a, bool restart) {
  const size_t off = (size_t)a.col0 * r;
  // bit 1: V is known to be feasible (output of a projection since the agent's last setX): skip its re-projection
  nesterov(st, a.prob->m, 0, (restart ? 1 : 0) | (a.v_feasible ? 2 : 0), -1, -1, alpha, gamma, Xg.p + off, Vg.p + off,
           Yg.p + off, XPrevg.p + off, nullptr, Buf2{{nullptr, nullptr}}, nullptr);
  a.v_feasible = true;
  return DCORA_OK;
}

// Agent::iterate(true) for `selected` when hosted here (ref src/Agent.cpp:535-551, 1158-1176, 1216-1278)
int RbcdSession::phase_selected(int selected) {
  if (selected < 0 || selected >= R) {
    set_last_error("rbcd: selected agent out of range");
    return DCORA_ERR_BAD_ARG;
  }
  DCORA_HIP(hipSetDevice(opt.device));
  if (!seq_advanced_) advance_sequences();
  seq_advanced_ = false;
  const bool restart = restart_now();
  AgentDev &a = agents[selected];
  if (a.hosted) {
    const int rc = update_selected_agent(a, restart);
    if (rc) return rc;
  }
  if (restart) gamma = alpha = 0;
  return DCORA_OK;
}

int RbcdSession::update_selected_agent(AgentDev &a, bool restart) {
  int rc = DCORA_OK;
  {
    DeviceProblem &pb = *a.prob;
    const size_t off = (size_t)a.col0 * r;
    const size_t B = sizeof(double) * (size_t)pb.nelem();
    // Graph::constructLinearCostTermPGO: G_b = sum_c X_c Q_cb from the neighbours' public poses
    // (ref src/Graph.cpp:685-822); the mirror Xg holds them after the pull / unpack -- or, for an agent that has been
    // handed poses through Agent::updateNeighborStates, its own cache of them: the auxiliary one when it optimises
    // from Y (ref src/Agent.cpp:1234-1240)
    a.last_skipped = false;
    auto cache_complete = [&](int which) {
      for (char c : a.got[which])
        if (!c) return false;
      return true;
    };
    if (a.detached && !cache_complete(opt.acceleration ? 1 : 0)) {
      // "cannot construct data matrices... Skip optimization" (ref src/Agent.cpp:1243-1249): only updateX is skipped;
      // updateGamma / updateAlpha / updateY have run before it and updateV and the restart follow (Agent::iterate, ref
      // src/Agent.cpp:535-596) -- X stays, Y <- proj((1 - alpha) X + alpha V), V <- proj(V + gamma (X - Y))
      a.last_skipped = true;
      if (opt.acceleration) {
        if (staged_selected_ != a.id || staged_iteration_ != iteration)
          nesterov(st, pb.m, 1, 0, -1, -1, alpha, gamma, Xg.p + off, Vg.p + off, Yg.p + off, XPrevg.p + off, pb.X0.p,
                   Buf2{{nullptr, nullptr}}, nullptr);
        staged_selected_ = -1;
        const Buf2 Xself{{Xg.p + off, Xg.p + off}};
        nesterov(st, pb.m, 2, 0, -1, -1, alpha, gamma, Xg.p + off, Vg.p + off, Yg.p + off, XPrevg.p + off, nullptr, Xself,
                 nullptr);
        a.v_feasible = true;
        if (restart) {
          // restartNesterovAcceleration: X = XPrev; updateX(true, false) reads the PLAIN cache; V = X; Y = X
          const Buf2 Xprev{{XPrevg.p + off, XPrevg.p + off}};
          bool solved = false;
          if (cache_complete(0)) {
            launch_spmm(st, r, a.coupling.view(), buf1(a.nbr[0].p), 0, nullptr, buf1(pb.G.p), 0, nullptr, Gate{});
            pb.has_G = true;
            last_solver = &pb;
            Buf2 Xres{{nullptr, nullptr}};
            const SolverCtl *cs = nullptr;
            DCORA_HIP(hipMemcpyAsync(pb.X0.p, XPrevg.p + off, B, hipMemcpyDeviceToDevice, st));
            rc = pb.optimize_dev(opt.local, &Xres, &cs);
            if (rc) return rc;
            rc = resolve_pick(pb, &Xres, &cs);
            if (rc) return rc;
            nesterov(st, pb.m, 3, 0, -1, -1, alpha, gamma, Xg.p + off, Vg.p + off, Yg.p + off, XPrevg.p + off, nullptr,
                     Xres, cs);
            // last_skipped stays true: `success` of Agent::iterate is the FIRST updateX's result, the restart's is
            // discarded (ref src/Agent.cpp:548-553), so iterate() returns false and readyToTerminate stays false
            solved = true;
          }
          if (!solved)
            nesterov(st, pb.m, 3, 0, -1, -1, alpha, gamma, Xg.p + off, Vg.p + off, Yg.p + off, XPrevg.p + off, nullptr,
                     Xprev, nullptr);
        }
      }
      return DCORA_OK;
    }
    const double *nsrc = a.detached ? a.nbr[opt.acceleration ? 1 : 0].p : Xg.p;
    launch_spmm(st, r, a.coupling.view(), buf1(nsrc), 0, nullptr, buf1(pb.G.p), 0, nullptr, Gate{});
    pb.has_G = true;
    Buf2 Xres{{nullptr, nullptr}};
    const SolverCtl *cs = nullptr;
    last_solver = &pb;
    if (opt.acceleration) {
      // (skipped only when the non-selected agents' launch of THIS round has taken the step already)
      if (staged_selected_ != a.id || staged_iteration_ != iteration)
        nesterov(st, pb.m, 1, 0, -1, -1, alpha, gamma, Xg.p + off, Vg.p + off, Yg.p + off, XPrevg.p + off, pb.X0.p,
                 Buf2{{nullptr, nullptr}}, nullptr);
      staged_selected_ = -1;
      rc = pb.optimize_dev(opt.local, &Xres, &cs);
      if (rc) return rc;
      rc = resolve_pick(pb, &Xres, &cs);
      if (rc) return rc;
      nesterov(st, pb.m, 2, 0, -1, -1, alpha, gamma, Xg.p + off, Vg.p + off, Yg.p + off, XPrevg.p + off, nullptr, Xres,
               cs);
      a.v_feasible = true;  // V = proj(V + gamma (X - Y))
      if (restart) {
        // restartNesterovAcceleration: X = XPrev; updateX(true, false); V = X; Y = X
        bool plain_ok = true;
        if (a.detached) {  // updateX(.., false) reads the PLAIN cache (ref src/Agent.cpp:1237-1240)
          if (cache_complete(0)) {
            launch_spmm(st, r, a.coupling.view(), buf1(a.nbr[0].p), 0, nullptr, buf1(pb.G.p), 0, nullptr, Gate{});
          } else {
            // no G can be built from an incomplete cache: X = XPrev without the solve.  last_skipped is NOT raised: the
            // first updateX of this iterate succeeded and that is what iterate() reports (ref src/Agent.cpp:548-553)
            plain_ok = false;
          }
        }
        if (plain_ok) {
          DCORA_HIP(hipMemcpyAsync(pb.X0.p, XPrevg.p + off, B, hipMemcpyDeviceToDevice, st));
          rc = pb.optimize_dev(opt.local, &Xres, &cs);
          if (rc) return rc;
          rc = resolve_pick(pb, &Xres, &cs);
          if (rc) return rc;
          nesterov(st, pb.m, 3, 0, -1, -1, alpha, gamma, Xg.p + off, Vg.p + off, Yg.p + off, XPrevg.p + off, nullptr,
                   Xres, cs);
        } else {
          const Buf2 Xprev{{XPrevg.p + off, XPrevg.p + off}};
          nesterov(st, pb.m, 3, 0, -1, -1, alpha, gamma, Xg.p + off, Vg.p + off, Yg.p + off, XPrevg.p + off, nullptr,
                   Xprev, nullptr);
        }
      }
    } else {
      DCORA_HIP(hipMemcpyAsync(XPrevg.p + off, Xg.p + off, B, hipMemcpyDeviceToDevice, st));
      DCORA_HIP(hipMemcpyAsync(pb.X0.p, Xg.p + off, B, hipMemcpyDeviceToDevice, st));
      rc = pb.optimize_dev(opt.local, &Xres, &cs);
      if (rc) return rc;
      if (cs) {  // resolve the device-side pick on the host (non-accelerated path is not latency critical)
        dcora_ropt_result tmp;
        rc = pb.fetch_result(&tmp);
        if (rc) return rc;
        Xres.p[0] = tmp.success && pb.result_index() ? pb.X1.p : pb.X0.p;
      }
      DCORA_HIP(hipMemcpyAsync(Xg.p + off, Xres.p[0], B, hipMemcpyDeviceToDevice, st));
    }
  }
  return DCORA_OK;
}

// Agent::iterate(doOptimization) of one agent (ref src/Agent.cpp:535-596).  The driver calls every agent once per
// round (examples/MultiRobotExample.cpp:223-262): the first call of a round advances the shared gamma / alpha
// sequences (identical for all agents, :1189-1200), a restart round zeroes them when the next round begins.
int RbcdSession::agent_iterate(int agent, bool do_optimization) {
  staged_selected_ = -1;  // the per-agent API never rides in a whole-graph launch
  if (agent < 0 || agent >= R) {
    set_last_error("rbcd: agent out of range");
    return DCORA_ERR_BAD_ARG;
  }
  AgentDev &a = agents[agent];
  if (!a.hosted) {
    set_last_error("rbcd: agent is hosted by another rank");
    return DCORA_ERR_BAD_ARG;
  }
  DCORA_HIP(hipSetDevice(opt.device));
  set_marks_.assign(R, 0);
  // Agent::iteration_number() of every agent; between rounds they are all equal to the session's round counter
  // (also after the session-level calls, which advance whole rounds)
  bool level = (int)agent_it.size() == R;
  for (int q = 0; level && q < R; ++q) level = !agents[q].hosted || agent_it[q] == agent_it[agent];
  if ((int)agent_it.size() != R || (level && agent_it[agent] != iteration)) agent_it.assign(R, iteration);
  if (agent_it[agent] == iteration) {  // this agent opens round iteration + 1: everybody has finished the last one
    for (int q = 0; q < R; ++q)
      if (agents[q].hosted && agent_it[q] != iteration) {
        set_last_error("rbcd: agent " + std::to_string(agent) + " starts a new round before agent " +
                       std::to_string(q) + " has iterated in the current one (agents advance in lockstep)");
        return DCORA_ERR_BAD_ARG;
      }
    if (pending_reset_) gamma = alpha = 0;
    pending_reset_ = false;
    advance_sequences();
    seq_advanced_ = false;
    for (int q = 0; q < R; ++q)
      if (!agents[q].hosted) agent_it[q] = iteration;
  } else if (agent_it[agent] != iteration - 1) {
    set_last_error("rbcd: agents advance in lockstep");
    return DCORA_ERR_BAD_ARG;
  }
  agent_it[agent] = iteration;
  const bool restart = restart_now();
  int rc;
  if (do_optimization) {
    rc = update_selected_agent(a, restart);
  } else {
    rc = opt.acceleration ? update_nonselected_agent(a, restart) : DCORA_OK;
  }
  if (restart) pending_reset_ = true;
  return rc;
}

// Agent::updateNeighborStates (ref src/Agent.cpp:844-906): poses the agent does not require are ignored
// (Graph::requireNeighborPose); the others go into its plain or auxiliary cache
int RbcdSession::agent_update_neighbor(int agent, int neighbor, int count, const int *frames, const double *poses,
                                       bool aux) {
  if (agent < 0 || agent >= R || neighbor < 0 || neighbor >= R || neighbor == agent || count < 0) {
    set_last_error("rbcd: bad agent / neighbour");
    return DCORA_ERR_BAD_ARG;
  }
  AgentDev &a = agents[agent];
  if (!a.hosted) {
    set_last_error("rbcd: agent is hosted by another rank");
    return DCORA_ERR_BAD_ARG;
  }
  DCORA_HIP(hipSetDevice(opt.device));
  const int dh = d + 1;
  const size_t blk = (size_t)r * dh;
  if (!a.detached) {
    const size_t N = (size_t)r * dh * n;
    for (int w = 0; w < 2; ++w) {
      DCORA_HIP(a.nbr[w].alloc(N));
      DCORA_HIP(hipMemsetAsync(a.nbr[w].p, 0, sizeof(double) * N, st));
      a.got[w].assign(a.required.size(), 0);
    }
    a.detached = true;
  }
  const int which = aux ? 1 : 0;
  std::vector<double> packed;
  std::vector<int> cols;
  for (int q = 0; q < count; ++q) {
    const int f = frames[q];
    if (f < 0 || f >= agents[neighbor].n) {
      set_last_error("rbcd: frame out of range");
      return DCORA_ERR_BAD_ARG;
    }
    const int gp = P.start(neighbor) + f;
    const auto it = std::lower_bound(a.required.begin(), a.required.end(), gp);
    if (it == a.required.end() || *it != gp) continue;  // not required: ignored, as in the reference
    a.got[which][(size_t)(it - a.required.begin())] = 1;
    packed.insert(packed.end(), poses + (size_t)q * blk, poses + (size_t)(q + 1) * blk);
    for (int c = 0; c < dh; ++c) cols.push_back(gp * dh + c);
  }
  if (!cols.empty()) {
    DevBuf<double> dp;
    DevBuf<int> dc;
    DCORA_HIP(dp.alloc(packed.size()));
    DCORA_HIP(dc.alloc(cols.size()));
    DCORA_HIP(hipMemcpyAsync(dp.p, packed.data(), sizeof(double) * packed.size(), hipMemcpyHostToDevice, st));
    DCORA_HIP(hipMemcpyAsync(dc.p, cols.data(), sizeof(int) * cols.size(), hipMemcpyHostToDevice, st));
    launch_scatter_cols(st, r, (int)cols.size(), dc.p, dp.p, a.nbr[which].p);
    DCORA_HIP(hipStreamSynchronize(st));  // the staging buffers leave scope
  }
  return DCORA_OK;
}

int RbcdSession::agent_get_X(int agent, double *Xh) {
  if (agent < 0 || agent >= R) {
    set_last_error("rbcd: agent out of range");
    return DCORA_ERR_BAD_ARG;
  }
  DCORA_HIP(hipSetDevice(opt.device));
  const AgentDev &a = agents[agent];
  const size_t B = sizeof(double) * (size_t)r * (d + 1) * a.n;
  DCORA_HIP(hipMemcpyAsync(Xh, Xg.p + (size_t)a.col0 * r, B, hipMemcpyDeviceToHost, st));
  DCORA_HIP(hipStreamSynchronize(st));
  return DCORA_OK;
}

// Agent::setX + initializeAcceleration (ref src/Agent.cpp:64-77, 1178-1187)
int RbcdSession::agent_set_X(int agent, const double *Xh) {
  staged_selected_ = -1;
  if (agent < 0 || agent >= R) {
    set_last_error("rbcd: agent out of range");
    return DCORA_ERR_BAD_ARG;
  }
  DCORA_HIP(hipSetDevice(opt.device));
  const AgentDev &a = agents[agent];
  const size_t off = (size_t)a.col0 * r;
  const size_t B = sizeof(double) * (size_t)r * (d + 1) * a.n;
  DCORA_HIP(hipMemcpyAsync(Xg.p + off, Xh, B, hipMemcpyHostToDevice, st));
  DCORA_HIP(hipMemcpyAsync(Vg.p + off, Xg.p + off, B, hipMemcpyDeviceToDevice, st));
  DCORA_HIP(hipMemcpyAsync(Yg.p + off, Xg.p + off, B, hipMemcpyDeviceToDevice, st));
  DCORA_HIP(hipMemcpyAsync(XPrevg.p + off, Xg.p + off, B, hipMemcpyDeviceToDevice, st));
  DCORA_HIP(hipStreamSynchronize(st));
  // initializeAcceleration of this agent: its V is the X it was handed (not necessarily feasible) until the next
  // projection; once EVERY agent has been set since the last round, the shared sequences restart as after set_X
  agents[agent].v_feasible = false;
  if ((int)set_marks_.size() != R) set_marks_.assign(R, 0);
  set_marks_[agent] = 1;
  bool all = true;
  for (int q = 0; q < R; ++q) all = all && (set_marks_[q] || !agents[q].hosted);
  if (all) {
    gamma = alpha = 0;
    iteration = 0;
    seq_advanced_ = false;
    pending_reset_ = false;
    agent_it.assign(R, 0);
    set_marks_.assign(R, 0);
  }
  return DCORA_OK;
}

// central evaluation of the driver (ref examples/MultiRobotExample.cpp:264-305): 2 f, |rgrad|, greedy selection
int RbcdSession::evaluate_central(double *cost2, double *gradnorm, double *block_norms, int *next_selected) {
  if (!central) {
    set_last_error("central evaluation needs world_size == 1");
    return DCORA_ERR_UNSUPPORTED;
  }
  DeviceProblem &c = *central;
  const bool gf = c.group && c.fused && !c.has_bsr && c.Q.n_long == 0;
  const bool gfb = c.group && c.has_bsr;  // the same on the block structure of Q (large graphs: any number of poses)
  if (!gf && !gfb) c.enqueue_egrad(Xg.p, c.EG0.p, c.pA.p);
  if (c.group) {
    int nA = c.npA();
    int wpa = 0;  // > 0: the evaluation dealt its workgroups to the agents and pB holds their sums (BsrGradOut)
    if (gf)  // X Q, the cost dots, the Riemannian gradient and the per-pose norms in one launch
      nA = launch_fused_grad(st, c.m, c.Q.view(), buf1(Xg.p), nullptr, buf1(c.EG0.p), buf1(c.RG0.p),
                             Buf2{{nullptr, nullptr}}, 0, c.pA.p, c.pB.p, posenorm.p, Gate{});
    else if (gfb)
      // (only the dots and the per-pose norms leave the kernel: writing EG and RG of the whole graph, 2 x 16 MB on the
      // 100k lattice, would be written back at the kernel's end for nobody)
      nA = launch_fused_grad_bsr(st, c.m.r, c.m.d, c.Qb.view(), buf1(Xg.p), nullptr, Buf2{{nullptr, nullptr}},
                                 Buf2{{nullptr, nullptr}}, Buf2{{nullptr, nullptr}}, 0, c.pA.p, c.pB.p, posenorm.p,
                                 Gate{}, pose_start.p, R, &wpa);
    else
      c.enq_rgrad(buf1(Xg.p), buf1(c.EG0.p), buf1(c.RG0.p), Buf2{{nullptr, nullptr}}, 0, c.pB.p, Gate{}, posenorm.p);
    const int want = ++eval_seq;
    launch_eval_finish(st, R, pose_start.p, posenorm.p, c.pA.p, nA, eval_dev, want, eval_split.p, n,
                       wpa > 0 ? c.pB.p : nullptr, wpa);
    const auto t0 = std::chrono::steady_clock::now();
    unsigned spins = 0;
    while (eval_host->seq != want) {
      if ((++spins & 4095u) == 0 &&
          std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 20.0) {
        set_last_error("rbcd: evaluation epilogue did not complete (spin timeout)");
        return DCORA_ERR_HIP;
      }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    if (block_norms)
      for (int b = 0; b < R; ++b) block_norms[b] = eval_host->block_norms[b];
    if (cost2) *cost2 = eval_host->cost2;
    if (gradnorm) *gradnorm = eval_host->gradnorm;
    if (next_selected) *next_selected = eval_host->next;
    return DCORA_OK;
  }
  launch_rgrad(st, mg, buf1(Xg.p), buf1(c.EG0.p), buf1(c.RG0.p), Buf2{{nullptr, nullptr}}, 0, c.pB.p, Gate{});
  launch_block_dots(st, r, R, col_start.p, c.RG0.p, nullptr, evalbuf.p);
  launch_sum_partials(st, c.pA.p, c.npA(), 2, 2, evalbuf.p + 2 * R);
  std::vector<double> h(2 * R + 2);
  DCORA_HIP(hipMemcpyAsync(h.data(), evalbuf.p, sizeof(double) * (2 * R + 2), hipMemcpyDeviceToHost, st));
  DCORA_HIP(hipStreamSynchronize(st));
  double g2 = 0, best = -1;
  int arg = 0;
  for (int b = 0; b < R; ++b) {
    const double nb = std::sqrt(h[2 * b]);
    if (block_norms) block_norms[b] = nb;
    g2 += h[2 * b];
    if (nb > best) {
      best = nb;
      arg = b;
    }
  }
  if (cost2) *cost2 = 2.0 * (0.5 * h[2 * R] + h[2 * R + 1]);
  if (gradnorm) *gradnorm = std::sqrt(g2);
  if (next_selected) *next_selected = arg;
  return DCORA_OK;
}

int RbcdSession::last_result(dcora_ropt_result *res) {
  if (last_solver) return last_solver->fetch_result(res);
  *res = last;
  return DCORA_OK;
}

// distributed form of the same evaluation: per hosted agent |Proj(X_b Q_bb + G_b)|^2 and <X_b, X_b Q_bb + G_b>
int RbcdSession::phase_evaluate_dev(double *out_dev) {
  DCORA_HIP(hipSetDevice(opt.device));
  DCORA_HIP(hipMemsetAsync(out_dev, 0, sizeof(double) * 2 * R, st));
  for (AgentDev &a : agents) {
    if (!a.hosted) continue;
    DeviceProblem &pb = *a.prob;
    const double *Xb = Xg.p + (size_t)a.col0 * r;
    launch_spmm(st, r, a.coupling.view(), buf1(Xg.p), 0, nullptr, buf1(pb.G.p), 0, nullptr, Gate{});
    pb.has_G = true;
    pb.enqueue_egrad(Xb, pb.EG1.p, nullptr);
    launch_rgrad(st, pb.m, buf1(Xb), buf1(pb.EG1.p), buf1(pb.RG1.p), Buf2{{nullptr, nullptr}}, 0, pb.pB.p, Gate{});
    launch_sum_partials(st, pb.pB.p, pb.npPose(), 1, 1, out_dev + 2 * a.id);
    launch_dot(st, pb.nelem(), Xb, pb.EG1.p, pb.p3.p);
    launch_sum_partials(st, pb.p3.p, pb.npVec(), 1, 1, out_dev + 2 * a.id + 1);
  }
  return DCORA_OK;
}

int RbcdSession::iterate(int selected, double *cost2, double *gradnorm, double *block_norms, int *next_selected) {
  if (selected < 0 || selected >= R) {
    set_last_error("rbcd: selected agent out of range");
    return DCORA_ERR_BAD_ARG;
  }
  int rc = phase_nonselected(selected);
  if (rc) return rc;
  // world_size == 1: the "pull" of public poses (ref examples/MultiRobotExample.cpp:236-258) is the identity,
  // all agents' blocks live in the same mirror Xg
  rc = phase_selected(selected);
  if (rc) return rc;
  int nxt = selected;
  rc = evaluate_central(cost2, gradnorm, block_norms, &nxt);
  if (rc) return rc;
  // greedy selection only when the selected agent has neighbours (:290-292)
  if (next_selected) *next_selected = (agents[selected].coupling.nnz > 0) ? nxt : selected;
  return DCORA_OK;
}

// Greedy colouring in agent order (smallest colour not used by a neighbour).  Agents of one colour share no
// measurement, so their simultaneous updates equal the same updates done one after the other.
int RbcdSession::agent_colours(int *colours, int *ncolours) const {
  int nc = 0;
  for (int b = 0; b < R; ++b) {
    int c = 0;
    for (bool clash = true; clash; c += clash) {
      clash = false;
      for (int q : agents[b].neighbors)
        if (q < b && colours[q] == c) clash = true;
    }
    colours[b] = c;
    nc = std::max(nc, c + 1);
  }
  if (ncolours) *ncolours = nc;
  return DCORA_OK;
}

// local solve of one agent on its own stream, from the G / X0 staged by iterate_set; the accepted iterate goes
// back into the global mirror without a host round trip when the solver keeps its choice on the device
int RbcdSession::solve_block(AgentDev &a, std::string *err, bool serial) {
  auto fail = [&](int rc) {
    if (err) *err = dcora_last_error();
    return rc;
  };
  if (hipSetDevice(opt.device) != hipSuccess) return fail(DCORA_ERR_HIP);
  DeviceProblem &pb = *a.prob;
  const size_t off = (size_t)a.col0 * r;
  const size_t B = sizeof(double) * (size_t)pb.nelem();
  hipStream_t keep = pb.st;
  // serial: the set's solves one after the other on the session's stream (each may then run its tCG runs as ONE launch,
  // k_tcg_run); otherwise side by side on the agents' own streams, on the launches per iteration
  hipStream_t run_on = serial ? st : a.own;
  pb.st = run_on;
  Buf2 Xres{{nullptr, nullptr}};
  const SolverCtl *cs = nullptr;
  pb.concurrent_solves = !serial;  // several solves share the device: no co-resident one-launch tCG run
  int rc = pb.optimize_dev(opt.local, &Xres, &cs);
  pb.concurrent_solves = false;
  if (!rc) {
    if (cs && group_kernels(pb.m)) {
      nesterov(run_on, pb.m, 3, 0, -1, -1, 0.0, 0.0, Xg.p + off, Vg.p + off, Yg.p + off, XPrevg.p + off, nullptr, Xres,
               cs);
    } else {
      if (cs) {
        dcora_ropt_result tmp;
        rc = pb.fetch_result(&tmp);
        Xres.p[0] = tmp.success && pb.result_index() ? pb.X1.p : pb.X0.p;
      }
      if (!rc && hipMemcpyAsync(Xg.p + off, Xres.p[0], B, hipMemcpyDeviceToDevice, run_on) != hipSuccess)
        rc = DCORA_ERR_HIP;
    }
  }
  if (!rc && !serial && hipEventRecord(a.done, a.own) != hipSuccess) rc = DCORA_ERR_HIP;
  pb.st = keep;
  return rc ? fail(rc) : DCORA_OK;
}

// One tick in which the agents of `set` run Agent::iterate(true) at the same time, every one of them seeing the
// neighbour states as they were when the tick began (what concurrently firing agents of the asynchronous mode see,
// ref src/Agent.cpp:650-678; non-accelerated like that mode, :651-653).  With a set of mutually non-adjacent
// agents (one colour of agent_colours) the result equals updating them one after the other.
int RbcdSession::iterate_set(const int *set, int count, int allow_adjacent) {
  staged_selected_ = -1;
  if (opt.acceleration) {
    set_last_error("rbcd: simultaneous updates need acceleration off (ref src/Agent.cpp:651-653)");
    return DCORA_ERR_UNSUPPORTED;
  }
  if (!set || count < 1 || count > R) {
    set_last_error("rbcd: bad agent set");
    return DCORA_ERR_BAD_ARG;
  }
  std::vector<char> in(R, 0);
  for (int i = 0; i < count; ++i) {
    if (set[i] < 0 || set[i] >= R || in[set[i]]) {
      set_last_error("rbcd: agent set has an id out of range or twice");
      return DCORA_ERR_BAD_ARG;
    }
    in[set[i]] = 1;
  }
  if (!allow_adjacent)
    for (int i = 0; i < count; ++i)
      for (int q : agents[set[i]].neighbors)
        if (in[q]) {
          set_last_error("rbcd: agents " + std::to_string(set[i]) + " and " + std::to_string(q) +
                         " share measurements; pass allow_adjacent to update them from one snapshot anyway");
          return DCORA_ERR_BAD_ARG;
        }
  DCORA_HIP(hipSetDevice(opt.device));
  iteration++;
  seq_advanced_ = false;
  set_marks_.assign(R, 0);
  std::vector<AgentDev *> work;
  for (int i = 0; i < count; ++i)
    if (agents[set[i]].hosted) work.push_back(&agents[set[i]]);
  if (work.empty()) return DCORA_OK;
  // snapshot: every G and every start point is taken before any block is written back
  for (AgentDev *a : work) {
    DeviceProblem &pb = *a->prob;
    const size_t off = (size_t)a->col0 * r;
    const size_t B = sizeof(double) * (size_t)pb.nelem();
    launch_spmm(st, r, a->coupling.view(), buf1(Xg.p), 0, nullptr, buf1(pb.G.p), 0, nullptr, Gate{});
    pb.has_G = true;
    DCORA_HIP(hipMemcpyAsync(XPrevg.p + off, Xg.p + off, B, hipMemcpyDeviceToDevice, st));
    DCORA_HIP(hipMemcpyAsync(pb.X0.p, Xg.p + off, B, hipMemcpyDeviceToDevice, st));
  }
  // Blocks whose tCG runs fit ONE launch (dense preconditioner, n / 2 co-resident workgroups): one after the other on the
  // session's stream -- 3 launches per RTR iteration each -- is faster on one device than side by side on the launches per
  // iteration (sphere2500 / 5 agents: 2420 -> see DESIGN.md block updates/s); the staged G / start points make the order
  // immaterial, and the two forms give the same bits.
  bool serial = true;
  for (AgentDev *a : work) serial = serial && a->prob->tcg_run_ok && a->prob->use_pc();
  std::vector<int> rcs(work.size(), DCORA_OK);
  std::vector<std::string> errs(work.size());
  if (serial) {
    for (size_t i = 0; i < work.size(); ++i) {
      rcs[i] = solve_block(*work[i], &errs[i], true);
      if (rcs[i]) {
        set_last_error(errs[i]);
        return rcs[i];
      }
    }
    last_solver = work.back()->prob.get();
    return DCORA_OK;
  }
  DCORA_HIP(hipEventRecord(fork_ev_, st));
  for (AgentDev *a : work) DCORA_HIP(hipStreamWaitEvent(a->own, fork_ev_, 0));
  if (work.size() == 1) {
    rcs[0] = solve_block(*work[0], &errs[0]);
  } else {
    // the solver paces each solve from the host (device_problem.hip): one host thread per concurrent solve
    std::vector<std::thread> th;
    for (size_t i = 1; i < work.size(); ++i)
      th.emplace_back([&, i] { rcs[i] = solve_block(*work[i], &errs[i]); });
    rcs[0] = solve_block(*work[0], &errs[0]);
    for (std::thread &t : th) t.join();
  }
  last_solver = work.back()->prob.get();
  for (size_t i = 0; i < work.size(); ++i) {
    if (rcs[i]) {
      set_last_error(errs[i]);
      return rcs[i];
    }
    DCORA_HIP(hipStreamWaitEvent(st, work[i]->done, 0));
  }
  return DCORA_OK;
}

int RbcdSession::pack_public(int agent, double *packed_dev) {
  AgentDev &a = agents[agent];
  launch_gather_cols(st, r, (int)a.public_poses.size() * (d + 1), a.public_cols.p, Xg.p, packed_dev);
  return DCORA_OK;
}
int RbcdSession::unpack_public(int agent, const double *packed_dev) {
  AgentDev &a = agents[agent];
  launch_scatter_cols(st, r, (int)a.public_poses.size() * (d + 1), a.public_cols.p, packed_dev, Xg.p);
  return DCORA_OK;
}

int RbcdSession::x_stage_hosted(double *host_area) {
  DCORA_HIP(hipSetDevice(opt.device));
  for (const AgentDev &a : agents) {
    if (!a.hosted) continue;
    const size_t off = (size_t)a.col0 * r;
    DCORA_HIP(hipMemcpyAsync(host_area + off, Xg.p + off, sizeof(double) * (size_t)r * (d + 1) * a.n,
                             hipMemcpyDeviceToHost, st));
  }
  DCORA_HIP(hipStreamSynchronize(st));
  return DCORA_OK;
}

}  // namespace dcora
