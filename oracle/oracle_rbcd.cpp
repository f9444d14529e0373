// ORACLE -- test infrastructure only (see oracle.hpp).
// Agent (RBCD++ state: Nesterov sequences, restart, neighbour caches) and the
// synchronous multi-robot driver with the Riemannian staircase
// (ref: src/Agent.cpp:64-152, 535-596, 844-906, 1158-1278;
//  examples/MultiRobotExample.cpp:56-364).
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <set>

#include <atomic>
#include <thread>

#include "oracle.hpp"

namespace orc {

using clk = std::chrono::steady_clock;
static double secs(clk::time_point a, clk::time_point b) { return std::chrono::duration<double>(b - a).count(); }

void Agent::setup(int id_, int R_, int r, int d, int n, const std::vector<Meas> &touching) {
  id = id_;
  R = R_;
  D = Dims{r, d, n, 0, 0};
  mine = touching;
  shared.clear();
  std::set<int> pub;
  for (const Meas &e : mine)
    if (e.r1 != e.r2) {
      shared.push_back(e);
      if (e.r1 == id) pub.insert(e.p1);
      if (e.r2 == id) pub.insert(e.p2);
    }
  public_ids.assign(pub.begin(), pub.end());
  Q = build_Q_pgo(d, n, id, mine);
  CSR M = csr_add_diag(Q, 0.1);  // ref: src/Graph.cpp:1904-1909 (PGO reg = 1e-1)
  precon.factor(M, d + 1);
  iteration = 0;
}

// ref: src/Agent.cpp:64-77, 1178-1187
void Agent::setX(const Mat &Xin) {
  X = Xin;
  if (acceleration) {
    XPrev = X;
    gamma = 0;
    alpha = 0;
    V = X;
    Y = X;
  }
}

void Agent::shared_dict(PoseDict &out) const {
  out.clear();
  const int dh = D.d + 1;
  for (int f : public_ids) {
    std::vector<double> p((size_t)D.r * dh);
    std::copy(X.col(f * dh), X.col(f * dh) + (size_t)D.r * dh, p.begin());
    out[{id, f}] = std::move(p);
  }
}

void Agent::update_neighbor(int nid, const PoseDict &dict, bool aux) {
  for (const auto &kv : dict) {
    if (kv.first.first != nid) continue;
    // ref: src/Agent.cpp:876-877 requireNeighborPose: keep only poses that
    // appear in my shared loop closures
    bool need = false;
    for (const Meas &e : shared)
      if ((e.r1 == nid && e.p1 == kv.first.second && e.r2 == id) ||
          (e.r2 == nid && e.p2 == kv.first.second && e.r1 == id)) {
        need = true;
        break;
      }
    if (!need) continue;
    (aux ? nbr_aux : nbr)[kv.first] = kv.second;
  }
}

// ref: src/Agent.cpp:1216-1278
bool Agent::updateX(bool doOptimization, bool accel) {
  if (!doOptimization) {
    if (accel) X = Y;
    return true;
  }
  if (!build_G_pgo(D.r, D.d, D.n, id, shared, accel ? nbr_aux : nbr, G)) {
    last = ROptResult();
    return false;
  }
  Problem P;
  P.D = D;
  P.Q = &Q;
  P.G = &G;
  P.precon = &precon;
  const Mat &X0 = accel ? Y : X;
  X = optimize(P, opt, X0, &last);
  return true;
}

// ref: src/Agent.cpp:535-596 with 1158-1214
bool Agent::iterate(bool doOptimization) {
  iteration++;
  XPrev = X;
  bool success;
  if (acceleration) {
    gamma = (1 + std::sqrt(1 + 4.0 * R * R * gamma * gamma)) / (2.0 * R);  // updateGamma
    alpha = 1.0 / (gamma * R);                                             // updateAlpha
    {                                                                      // updateY
      Mat M(X.rows, X.cols);
      for (size_t i = 0; i < M.a.size(); ++i) M.a[i] = (1 - alpha) * X.a[i] + alpha * V.a[i];
      project_to_manifold(D, M, Y);
    }
    success = updateX(doOptimization, true);
    {  // updateV
      Mat M(X.rows, X.cols);
      for (size_t i = 0; i < M.a.size(); ++i) M.a[i] = V.a[i] + gamma * (X.a[i] - Y.a[i]);
      project_to_manifold(D, M, V);
    }
    if ((iteration + 1) % restart_interval == 0) {  // shouldRestart / restartNesterovAcceleration
      X = XPrev;
      updateX(doOptimization, false);
      V = X;
      Y = X;
      gamma = 0;
      alpha = 0;
    }
  } else {
    success = updateX(doOptimization, false);
  }
  return success;
}

// ref: examples/MultiRobotExample.cpp
RBCDTrace run_rbcd(const Dataset &ds, const RBCDOptions &o, const Mat &X0) {
  RBCDTrace tr;
  const int d = ds.d, n = ds.n, dh = d + 1, Rn = o.num_robots;
  const int per = n / Rn;
  auto robot_of = [&](int idx) { return std::min(idx / per, Rn - 1); };
  auto start_of = [&](int rb) { return rb * per; };
  auto end_of = [&](int rb) { return rb == Rn - 1 ? n : (rb + 1) * per; };

  const auto t_setup0 = clk::now();
  // partition (ref: :56-118)
  std::vector<std::vector<Meas>> touching(Rn);
  for (const Meas &mi : ds.meas) {
    Meas m = mi;
    m.r1 = robot_of(mi.p1);
    m.r2 = robot_of(mi.p2);
    m.p1 = mi.p1 - start_of(m.r1);
    m.p2 = mi.p2 - start_of(m.r2);
    touching[m.r1].push_back(m);
    if (m.r2 != m.r1) touching[m.r2].push_back(m);
  }
  // central problem (evaluation)
  std::vector<Meas> central = ds.meas;
  for (Meas &m : central) m.r1 = m.r2 = 0;
  CSR Qc = build_Q_pgo(d, n, 0, central);
  Chol preconC;
  bool preconC_ready = false;
  tr.setup_seconds += secs(t_setup0, clk::now());

  Mat Xcurr(o.r_max, n * dh);
  for (int j = 0; j < X0.cols; ++j)
    for (int t = 0; t < X0.rows; ++t) Xcurr(t, j) = X0(t, j);

  int totalIter = 0;
  for (int r = o.r_min; r < o.r_max; ++r) {
    const auto ts0 = clk::now();
    std::vector<Agent> agents(Rn);
    for (int rb = 0; rb < Rn; ++rb) {
      agents[rb].acceleration = o.acceleration;
      agents[rb].opt = o.opt;
      agents[rb].setup(rb, Rn, r, d, end_of(rb) - start_of(rb), touching[rb]);
      Mat Xb(r, (end_of(rb) - start_of(rb)) * dh);
      for (int j = 0; j < Xb.cols; ++j)
        for (int t = 0; t < r; ++t) Xb(t, j) = Xcurr(t, start_of(rb) * dh + j);
      agents[rb].setX(Xb);
    }
    Problem Pc;
    Pc.D = Dims{r, d, n, 0, 0};
    Pc.Q = &Qc;
    tr.setup_seconds += secs(ts0, clk::now());

    const auto tl0 = clk::now();
    Mat Xopt(r, n * dh), RG;
    int selected = 0;
    for (int iter = 0; iter < o.max_iters; ++iter) {
      if (o.threads > 1) {  // the agents are independent here: each touches its own state only
        std::vector<std::thread> th;
        std::atomic<int> next(0);
        auto work = [&]() {
          for (;;) {
            const int rb = next.fetch_add(1);
            if (rb >= Rn) break;
            if (rb != selected) agents[rb].iterate(false);
          }
        };
        for (int t = 1; t < std::min(o.threads, Rn); ++t) th.emplace_back(work);
        work();
        for (auto &t : th) t.join();
      } else {
        for (int rb = 0; rb < Rn; ++rb)
          if (rb != selected) agents[rb].iterate(false);
      }
      for (int rb = 0; rb < Rn; ++rb) {
        if (rb == selected) continue;
        PoseDict dct;
        agents[rb].shared_dict(dct);
        agents[selected].update_neighbor(rb, dct, false);
      }
      if (o.acceleration)
        for (int rb = 0; rb < Rn; ++rb) {
          if (rb == selected) continue;
          PoseDict dct;
          agents[rb].shared_dict(dct);  // NB: X again, as the reference driver (:252)
          agents[selected].update_neighbor(rb, dct, true);
        }
      agents[selected].iterate(true);
      for (int rb = 0; rb < Rn; ++rb) {
        const Mat &Xr = agents[rb].X;
        std::copy(Xr.a.begin(), Xr.a.end(), Xopt.col(start_of(rb) * dh));
      }
      Pc.rgrad(Xopt, RG);
      const double gn = norm(RG);
      const double cost2 = 2 * Pc.f(Xopt);
      tr.cost.push_back(cost2);
      tr.gradnorm.push_back(gn);
      tr.selected.push_back(selected);
      tr.rank.push_back(r);
      tr.seconds.push_back(tr.rbcd_seconds + secs(tl0, clk::now()));
      if (o.verbose) std::printf("Iter = %d | robot = %d | cost = %.6f | gradnorm = %.6f\n", totalIter, selected, cost2, gn);
      if (gn < o.rgrad_tol) break;
      // greedy selection (:289-305); every agent here has neighbours
      bool has_nbr = !agents[selected].shared.empty();
      if (has_nbr) {
        double best = -1;
        int arg = 0;
        for (int rb = 0; rb < Rn; ++rb) {
          double s = 0;
          for (int j = start_of(rb) * dh; j < end_of(rb) * dh; ++j)
            for (int t = 0; t < r; ++t) s += RG(t, j) * RG(t, j);
          s = std::sqrt(s);
          if (s > best) {
            best = s;
            arg = rb;
          }
        }
        selected = arg;
      }
      totalIter++;
    }
    tr.rbcd_seconds += secs(tl0, clk::now());
    tr.total_iters = (int)tr.cost.size();
    tr.final_rank = r;
    tr.Xfinal = Xopt;
    if (!o.staircase) break;

    const auto tc0 = clk::now();
    CSR S = dual_certificate(Pc.D, Xopt, Qc);
    double theta = 0, lmin = 0;
    std::vector<double> v;
    long mv = 0;
    const bool opt_ok = fast_verification(S, o.min_eig_tol, dh, &theta, &v, &lmin, &mv);
    tr.theta = theta;
    tr.lambda_min = lmin;
    if (opt_ok) {
      tr.certified = 1;
      tr.cert_seconds += secs(tc0, clk::now());
      break;
    }
    if (theta >= -o.min_eig_tol / 2) {  // :332-334 (LOG(FATAL) in the reference)
      tr.certified = -1;
      tr.cert_seconds += secs(tc0, clk::now());
      break;
    }
    if (!preconC_ready) {
      CSR Mc = csr_add_diag(Qc, 0.1);
      preconC.factor(Mc, dh);
      preconC_ready = true;
    }
    Problem Pn;
    Pn.D = Dims{r + 1, d, n, 0, 0};
    Pn.Q = &Qc;
    Pn.precon = &preconC;
    Mat Xn;
    const bool esc = escape_saddle(Pn, Xopt, theta, v, 1e-6, 1e-6, Xn);
    tr.cert_seconds += secs(tc0, clk::now());
    if (!esc) break;
    for (int j = 0; j < Xn.cols; ++j)
      for (int t = 0; t < r + 1; ++t) Xcurr(t, j) = Xn(t, j);
  }
  return tr;
}


// Coloured simultaneous updates (a separate mode; the CPU figure that sits beside the product's dcora_rbcd_iterate_set):
// the agents of one colour -- no two share a measurement -- run Agent::iterate(true) AT THE SAME TIME, one host thread
// per agent as the reference's asynchronous mode starts them (ref src/Agent.cpp:650-678), non-accelerated as that mode
// is; every updating agent first pulls its neighbours' public poses.  A sweep = one tick per colour; the central cost
// is evaluated once per sweep.  Same partition and problem set-up as run_rbcd.
RBCDTrace run_coloured(const Dataset &ds, const RBCDOptions &o, const Mat &X0, int sweeps) {
  RBCDTrace tr;
  const int d = ds.d, n = ds.n, dh = d + 1, Rn = o.num_robots, r = o.r_min;
  const int per = n / Rn;
  auto robot_of = [&](int idx) { return std::min(idx / per, Rn - 1); };
  auto start_of = [&](int rb) { return rb * per; };
  auto end_of = [&](int rb) { return rb == Rn - 1 ? n : (rb + 1) * per; };
  const auto t_setup0 = clk::now();
  std::vector<std::vector<Meas>> touching(Rn);
  std::vector<std::set<int>> adj(Rn);
  for (const Meas &mi : ds.meas) {
    Meas m = mi;
    m.r1 = robot_of(mi.p1);
    m.r2 = robot_of(mi.p2);
    m.p1 = mi.p1 - start_of(m.r1);
    m.p2 = mi.p2 - start_of(m.r2);
    touching[m.r1].push_back(m);
    if (m.r2 != m.r1) {
      touching[m.r2].push_back(m);
      adj[m.r1].insert(m.r2);
      adj[m.r2].insert(m.r1);
    }
  }
  std::vector<int> colour(Rn, -1);
  int ncol = 0;
  for (int a = 0; a < Rn; ++a) {  // greedy, in index order
    std::set<int> used;
    for (int b : adj[a])
      if (colour[b] >= 0) used.insert(colour[b]);
    int c = 0;
    while (used.count(c)) ++c;
    colour[a] = c;
    ncol = std::max(ncol, c + 1);
  }
  std::vector<Meas> central = ds.meas;
  for (Meas &m : central) m.r1 = m.r2 = 0;
  CSR Qc = build_Q_pgo(d, n, 0, central);
  std::vector<Agent> agents(Rn);
  {
    std::vector<std::thread> th;
    std::atomic<int> next(0);
    auto work = [&]() {
      for (;;) {
        const int rb = next.fetch_add(1);
        if (rb >= Rn) break;
        agents[rb].acceleration = 0;
        agents[rb].opt = o.opt;
        agents[rb].setup(rb, Rn, r, d, end_of(rb) - start_of(rb), touching[rb]);
        Mat Xb(r, (end_of(rb) - start_of(rb)) * dh);
        for (int j = 0; j < Xb.cols; ++j)
          for (int t = 0; t < r; ++t) Xb(t, j) = X0(t, start_of(rb) * dh + j);
        agents[rb].setX(Xb);
      }
    };
    for (int t = 1; t < std::min(std::max(o.threads, 1), Rn); ++t) th.emplace_back(work);
    work();
    for (auto &t : th) t.join();
  }
  Problem Pc;
  Pc.D = Dims{r, d, n, 0, 0};
  Pc.Q = &Qc;
  tr.setup_seconds = secs(t_setup0, clk::now());
  const auto tl0 = clk::now();
  Mat Xopt(r, n * dh), RG;
  for (int sw = 0; sw < sweeps; ++sw) {
    for (int c = 0; c < ncol; ++c) {
      std::vector<int> set;
      for (int a = 0; a < Rn; ++a)
        if (colour[a] == c) set.push_back(a);
      // every member pulls the public poses of its neighbours (none of which updates in this tick)
      for (int a : set)
        for (int b : adj[a]) {
          PoseDict dct;
          agents[b].shared_dict(dct);
          agents[a].update_neighbor(b, dct, false);
        }
      if (o.threads > 1) {
        std::vector<std::thread> th;
        for (size_t i = 1; i < set.size(); ++i) th.emplace_back([&, i]() { agents[set[i]].iterate(true); });
        if (!set.empty()) agents[set[0]].iterate(true);
        for (auto &t : th) t.join();
      } else {
        for (int a : set) agents[a].iterate(true);
      }
    }
    for (int rb = 0; rb < Rn; ++rb) {
      const Mat &Xr = agents[rb].X;
      std::copy(Xr.a.begin(), Xr.a.end(), Xopt.col(start_of(rb) * dh));
    }
    Pc.rgrad(Xopt, RG);
    tr.cost.push_back(2 * Pc.f(Xopt));
    tr.gradnorm.push_back(norm(RG));
    tr.selected.push_back(ncol);
    tr.rank.push_back(r);
    tr.seconds.push_back(secs(tl0, clk::now()));
  }
  tr.rbcd_seconds = secs(tl0, clk::now());
  tr.total_iters = (int)tr.cost.size();
  tr.final_rank = r;
  tr.Xfinal = Xopt;
  return tr;
}

}  // namespace orc
