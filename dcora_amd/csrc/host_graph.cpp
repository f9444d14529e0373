// Host-side data feed of the product (see host_graph.h).  Q is assembled directly from the closed-form
// (d+1)x(d+1) blocks of every edge e = (i -> j), T = [R t; 0 1], Omega = diag(kappa I_d, tau):
//   Q_ii += T Omega T^T,  Q_jj += Omega,  Q_ij += -T Omega,  Q_ji += -Omega T^T
// which is what AbT * Omega * AbT^T of ref src/Graph.cpp:579-683 expands to.
#include "host_graph.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <sstream>

namespace dcora {

namespace {
double trace_inverse_sym(int d, const double *A) {  // A row-major symmetric d x d
  if (d == 2) {
    const double det = A[0] * A[3] - A[1] * A[2];
    return (A[0] + A[3]) / det;
  }
  const double m00 = A[4] * A[8] - A[5] * A[7];
  const double m11 = A[0] * A[8] - A[2] * A[6];
  const double m22 = A[0] * A[4] - A[1] * A[3];
  const double det = A[0] * m00 - A[1] * (A[3] * A[8] - A[5] * A[6]) + A[2] * (A[3] * A[7] - A[4] * A[6]);
  return (m00 + m11 + m22) / det;
}
}  // namespace

// ref src/DCORA_utils.cpp:179-375
bool load_g2o(const std::string &path, HostDataset &ds, std::string &err) {
  std::ifstream in(path);
  if (!in) {
    err = "cannot open " + path;
    return false;
  }
  ds = HostDataset();
  std::string line, tok;
  int maxid = -1;
  while (std::getline(in, line)) {
    std::istringstream ss(line);
    if (!(ss >> tok)) continue;
    if (tok == "VERTEX_SE2") {
      if (!ds.d) ds.d = 2;
      continue;
    }
    if (tok == "VERTEX_SE3:QUAT") {
      if (!ds.d) ds.d = 3;
      continue;
    }
    PoseMeas m;
    if (tok == "EDGE_SE2") {
      if (!ds.d) ds.d = 2;
      long i, j;
      double dx, dy, th, I11, I12, I13, I22, I23, I33;
      ss >> i >> j >> dx >> dy >> th >> I11 >> I12 >> I13 >> I22 >> I23 >> I33;
      m.p1 = (int)i;
      m.p2 = (int)j;
      m.t[0] = dx;
      m.t[1] = dy;
      const double c = std::cos(th), s = std::sin(th);
      m.R[0] = c;
      m.R[1] = s;
      m.R[2] = -s;
      m.R[3] = c;
      const double Tc[4] = {I11, I12, I12, I22};
      m.tau = 2.0 / trace_inverse_sym(2, Tc);
      m.kappa = I33;
    } else if (tok == "EDGE_SE3:QUAT") {
      if (!ds.d) ds.d = 3;
      long i, j;
      double dx, dy, dz, x, y, z, w, I[21];
      ss >> i >> j >> dx >> dy >> dz >> x >> y >> z >> w;
      for (double &q : I) ss >> q;
      m.p1 = (int)i;
      m.p2 = (int)j;
      m.t[0] = dx;
      m.t[1] = dy;
      m.t[2] = dz;
      // rotation matrix of the (unnormalised) quaternion, column-major
      m.R[0] = 1 - 2 * (y * y + z * z);
      m.R[1] = 2 * (x * y + z * w);
      m.R[2] = 2 * (x * z - y * w);
      m.R[3] = 2 * (x * y - z * w);
      m.R[4] = 1 - 2 * (x * x + z * z);
      m.R[5] = 2 * (y * z + x * w);
      m.R[6] = 2 * (x * z + y * w);
      m.R[7] = 2 * (y * z - x * w);
      m.R[8] = 1 - 2 * (x * x + y * y);
      const double Tc[9] = {I[0], I[1], I[2], I[1], I[6], I[7], I[2], I[7], I[11]};
      const double Rc[9] = {I[15], I[16], I[17], I[16], I[18], I[19], I[17], I[19], I[20]};
      m.tau = 3.0 / trace_inverse_sym(3, Tc);
      m.kappa = 3.0 / (2.0 * trace_inverse_sym(3, Rc));
    } else {
      err = "unrecognized g2o record: " + tok;
      return false;
    }
    maxid = std::max(maxid, std::max(m.p1, m.p2));
    ds.meas.push_back(m);
  }
  ds.n = maxid + 1;
  if (ds.d != 2 && ds.d != 3) {
    err = "no SE2/SE3 records in " + path;
    return false;
  }
  return true;
}

namespace {
struct EdgeBlocks {
  double TOT[16], TO[16], O[4];  // T Omega T^T, T Omega (row-major dh x dh), diag Omega
};
void edge_blocks(int d, const PoseMeas &e, EdgeBlocks &B) {
  const int dh = d + 1;
  double T[16] = {0};
  for (int a = 0; a < d; ++a) {
    for (int c = 0; c < d; ++c) T[a * dh + c] = e.R[a + c * d];
    T[a * dh + d] = e.t[a];
  }
  T[d * dh + d] = 1;
  for (int a = 0; a < d; ++a) B.O[a] = e.weight * e.kappa;
  B.O[d] = e.weight * e.tau;
  for (int a = 0; a < dh; ++a)
    for (int c = 0; c < dh; ++c) B.TO[a * dh + c] = T[a * dh + c] * B.O[c];
  for (int a = 0; a < dh; ++a)
    for (int c = 0; c < dh; ++c) {
      double s = 0;
      for (int q = 0; q < dh; ++q) s += B.TO[a * dh + q] * T[c * dh + q];
      B.TOT[a * dh + c] = s;
    }
}
}  // namespace

HostCsr build_Q_pgo(int d, int n, int id, const std::vector<PoseMeas> &meas) {
  const int dh = d + 1;
  std::vector<int> I, J;
  std::vector<double> V;
  I.reserve(meas.size() * 4 * dh * dh);
  J.reserve(I.capacity());
  V.reserve(I.capacity());
  auto put = [&](int bi, int bj, int a, int c, double v) {
    if (v == 0.0) return;  // structural zero of T Omega (matches the incidence-product pattern)
    I.push_back(bi * dh + a);
    J.push_back(bj * dh + c);
    V.push_back(v);
  };
  EdgeBlocks B;
  for (const PoseMeas &e : meas) {
    const int i = (e.r1 == id) ? e.p1 : -1;
    const int j = (e.r2 == id) ? e.p2 : -1;
    if (i < 0 && j < 0) continue;
    edge_blocks(d, e, B);
    if (i >= 0)
      for (int a = 0; a < dh; ++a)
        for (int c = 0; c < dh; ++c) put(i, i, a, c, B.TOT[a * dh + c]);
    if (j >= 0)
      for (int a = 0; a < dh; ++a) put(j, j, a, a, B.O[a]);
    if (i >= 0 && j >= 0)
      for (int a = 0; a < dh; ++a)
        for (int c = 0; c < dh; ++c) {
          put(i, j, a, c, -B.TO[a * dh + c]);
          put(j, i, c, a, -B.TO[a * dh + c]);
        }
  }
  // make sure every diagonal entry exists structurally
  for (int q = 0; q < dh * n; ++q) {
    I.push_back(q);
    J.push_back(q);
    V.push_back(0.0);
  }
  return csr_from_coo(dh * n, dh * n, I, J, V);
}

HostCsr build_coupling_pgo(int d, const Partition &P, int b, const std::vector<PoseMeas> &global_meas) {
  const int dh = d + 1;
  const int nb = P.end(b) - P.start(b);
  std::vector<int> I, J;
  std::vector<double> V;
  EdgeBlocks B;
  for (const PoseMeas &e : global_meas) {
    const int ri = P.robot_of(e.p1), rj = P.robot_of(e.p2);
    if (ri == rj) continue;
    if (ri != b && rj != b) continue;
    edge_blocks(d, e, B);
    if (ri == b) {  // row block = local source pose, column block = global destination pose: -T Omega
      const int li = e.p1 - P.start(b);
      for (int a = 0; a < dh; ++a)
        for (int c = 0; c < dh; ++c)
          if (B.TO[a * dh + c] != 0.0) {
            I.push_back(li * dh + a);
            J.push_back(e.p2 * dh + c);
            V.push_back(-B.TO[a * dh + c]);
          }
    } else {  // row block = local destination pose, column block = global source pose: -(T Omega)^T
      const int lj = e.p2 - P.start(b);
      for (int a = 0; a < dh; ++a)
        for (int c = 0; c < dh; ++c)
          if (B.TO[c * dh + a] != 0.0) {
            I.push_back(lj * dh + a);
            J.push_back(e.p1 * dh + c);
            V.push_back(-B.TO[c * dh + a]);
          }
    }
  }
  return csr_from_coo(dh * nb, dh * P.n, I, J, V);
}

}  // namespace dcora

// ---------------------------------------------------------------------------------------------------------------
// Range-aided SLAM (centralised agent)
// ---------------------------------------------------------------------------------------------------------------
#include <map>
#include <set>
#include <tuple>

namespace dcora {
namespace {
struct SymH {
  int type, robot, id;
};
// 'A'.. = robots, 'L' + upper-case = that robot's landmark, 'L' + digits = map landmark (robot 'M')
// (ref src/DCORA_utils.cpp:584-616)
bool parse_sym(const std::string &s, SymH &o) {
  if (s.empty()) return false;
  o = SymH{0, 0, 0};
  if (s[0] == 'L') {
    o.type = 1;
    if (s.size() > 1 && std::isupper((unsigned char)s[1])) {
      o.robot = s[1] - 'A';
      o.id = std::atoi(s.c_str() + 2);
    } else {
      o.robot = 'M' - 'A';
      o.id = std::atoi(s.c_str() + 1);
    }
    return true;
  }
  if (std::isupper((unsigned char)s[0])) {
    o.robot = s[0] - 'A';
    o.id = std::atoi(s.c_str() + 1);
    return true;
  }
  return false;
}
void quat_to_R(double x, double y, double z, double w, double *R) {
  R[0] = 1 - 2 * (y * y + z * z);
  R[1] = 2 * (x * y + z * w);
  R[2] = 2 * (x * z - y * w);
  R[3] = 2 * (x * y - z * w);
  R[4] = 1 - 2 * (x * x + z * z);
  R[5] = 2 * (y * z + x * w);
  R[6] = 2 * (x * z + y * w);
  R[7] = 2 * (y * z - x * w);
  R[8] = 1 - 2 * (x * x + y * y);
}
}  // namespace

bool load_pyfg(const std::string &path, HostRADataset &ds, std::string &err) {
  std::ifstream in(path);
  if (!in) {
    err = "cannot open " + path;
    return false;
  }
  ds = HostRADataset();
  using Key = std::pair<int, int>;
  std::map<Key, std::vector<double>> poses, lms;  // (robot, id) -> [R (d*d, col-major) | t] resp. [t]
  struct EPP { SymH a, b; PoseMeas m; };
  struct EPL { SymH a, b; PoseLandmarkMeasH m; };
  struct ERG { SymH a, b; RangeMeasH m; int robot_l; };
  std::vector<EPP> pps;
  std::vector<EPL> pls;
  std::vector<ERG> rgs;
  std::map<int, int> nsph;
  std::set<std::tuple<int, int, int, int, int, int>> seen;
  std::string line, tok, s1, s2;
  double ts;
  while (std::getline(in, line)) {
    std::istringstream ss(line);
    if (!(ss >> tok)) continue;
    SymH a;
    if (tok == "VERTEX_SE2" || tok == "VERTEX_SE3:QUAT") {
      const int d = tok == "VERTEX_SE2" ? 2 : 3;
      ds.d = d;
      ss >> ts >> s1;
      if (!parse_sym(s1, a)) { err = "bad symbol " + s1; return false; }
      std::vector<double> rec(d * d + d);
      if (d == 2) {
        double x, y, th;
        ss >> x >> y >> th;
        rec = {std::cos(th), std::sin(th), -std::sin(th), std::cos(th), x, y};
      } else {
        double x, y, z, qx, qy, qz, qw;
        ss >> x >> y >> z >> qx >> qy >> qz >> qw;
        quat_to_R(qx, qy, qz, qw, rec.data());
        rec[9] = x; rec[10] = y; rec[11] = z;
      }
      poses[{a.robot, a.id}] = rec;
    } else if (tok == "VERTEX_XY" || tok == "VERTEX_XYZ") {
      const int d = tok == "VERTEX_XY" ? 2 : 3;
      ss >> s1;
      if (!parse_sym(s1, a)) { err = "bad symbol " + s1; return false; }
      std::vector<double> t(d);
      for (double &q : t) ss >> q;
      lms[{a.robot, a.id}] = t;
    } else if (tok == "EDGE_SE2" || tok == "EDGE_SE3:QUAT") {
      const int d = tok == "EDGE_SE2" ? 2 : 3;
      EPP e;
      ss >> ts >> s1 >> s2;
      if (!parse_sym(s1, e.a) || !parse_sym(s2, e.b)) { err = "bad symbol in " + line; return false; }
      if (d == 2) {
        double x, y, th, c[6];
        ss >> x >> y >> th;
        for (double &q : c) ss >> q;
        e.m.t[0] = x; e.m.t[1] = y;
        e.m.R[0] = std::cos(th); e.m.R[1] = std::sin(th); e.m.R[2] = -std::sin(th); e.m.R[3] = std::cos(th);
        e.m.tau = 2.0 / (c[0] + c[3]);   // 2 / trace(cov_t)  (ref :546-557)
        e.m.kappa = 1.0 / c[5];          // 1 / cov_theta     (ref :566-573)
      } else {
        double x, y, z, qx, qy, qz, qw, c[21];
        ss >> x >> y >> z >> qx >> qy >> qz >> qw;
        for (double &q : c) ss >> q;
        e.m.t[0] = x; e.m.t[1] = y; e.m.t[2] = z;
        quat_to_R(qx, qy, qz, qw, e.m.R);
        e.m.tau = 3.0 / (c[0] + c[6] + c[11]);
        e.m.kappa = 3.0 / (2.0 * (c[15] + c[18] + c[20]));
      }
      pps.push_back(e);
    } else if (tok == "EDGE_SE2_XY" || tok == "EDGE_SE3_XYZ") {
      const int d = tok == "EDGE_SE2_XY" ? 2 : 3;
      EPL e;
      ss >> ts >> s1 >> s2;
      if (!parse_sym(s1, e.a) || !parse_sym(s2, e.b)) { err = "bad symbol in " + line; return false; }
      for (int i = 0; i < d; ++i) ss >> e.m.t[i];
      std::vector<double> c(d * (d + 1) / 2);
      for (double &q : c) ss >> q;
      e.m.tau = d / (d == 2 ? c[0] + c[2] : c[0] + c[3] + c[5]);
      pls.push_back(e);
    } else if (tok == "EDGE_RANGE") {
      ERG e;
      double range, cov;
      ss >> ts >> s1 >> s2 >> range >> cov;
      if (!parse_sym(s1, e.a) || !parse_sym(s2, e.b)) { err = "bad symbol in " + line; return false; }
      if (!seen.insert(std::make_tuple(e.a.type, e.a.robot, e.a.id, e.b.type, e.b.robot, e.b.id)).second) continue;
      e.m.range = range;
      e.m.precision = 1.0 / cov;
      e.robot_l = nsph[e.a.robot]++;  // unit sphere owned by the source robot (ref :1092-1097)
      rgs.push_back(e);
    }
  }
  if (ds.d != 2 && ds.d != 3) {
    err = "no pose vertices in " + path;
    return false;
  }
  const int d = ds.d;
  std::map<Key, int> pidx, lidx;
  for (auto &kv : poses) { const int i = (int)pidx.size(); pidx[kv.first] = i; }
  for (auto &kv : lms) { const int i = (int)lidx.size(); lidx[kv.first] = i; }
  std::map<int, int> sbase;
  int acc = 0;
  for (auto &kv : nsph) {
    sbase[kv.first] = acc;
    acc += kv.second;
  }
  ds.n = (int)pidx.size();
  ds.b = (int)lidx.size();
  ds.l = acc;
  auto state = [&](const SymH &s, int &idx) {
    auto &mp = s.type ? lidx : pidx;
    auto it = mp.find({s.robot, s.id});
    if (it == mp.end()) return false;
    idx = it->second;
    return true;
  };
  for (auto &e : pps) {
    PoseMeas m = e.m;
    m.r1 = m.r2 = 0;
    m.weight = 1;
    if (!state(e.a, m.p1) || !state(e.b, m.p2)) { err = "edge refers to an unknown vertex"; return false; }
    ds.pose_pose.push_back(m);
  }
  for (auto &e : pls) {
    PoseLandmarkMeasH m = e.m;
    if (!state(e.a, m.i) || !state(e.b, m.j)) { err = "edge refers to an unknown vertex"; return false; }
    ds.pose_landmark.push_back(m);
  }
  for (auto &e : rgs) {
    RangeMeasH m = e.m;
    m.type1 = e.a.type;
    m.type2 = e.b.type;
    if (!state(e.a, m.i) || !state(e.b, m.j)) { err = "edge refers to an unknown vertex"; return false; }
    m.l = sbase[e.a.robot] + e.robot_l;
    ds.ranges.push_back(m);
  }
  // ownership (ref src/DCORA_utils.cpp:1370-1512, src/Graph.cpp:584-616, 1092-1097): a pose belongs to the robot
  // of its symbol, a landmark to the robot named in its symbol ('M' = the map), a unit sphere to the SOURCE robot
  // of its range measurement
  ds.pose_robot.assign(ds.n, 0);
  ds.landmark_robot.assign(ds.b, 0);
  ds.sphere_robot.assign(ds.l, 0);
  for (auto &kv : pidx) ds.pose_robot[kv.second] = kv.first.first;
  for (auto &kv : lidx) ds.landmark_robot[kv.second] = kv.first.first;
  for (auto &kv : nsph)
    for (int q = 0; q < kv.second; ++q) ds.sphere_robot[sbase[kv.first] + q] = kv.first;
  const int k = ds.k();
  ds.gt.assign((size_t)d * k, 0.0);
  std::vector<const std::vector<double> *> prec(ds.n), lrec(ds.b);
  for (auto &kv : poses) prec[pidx[kv.first]] = &kv.second;
  for (auto &kv : lms) lrec[lidx[kv.first]] = &kv.second;
  for (int i = 0; i < ds.n; ++i) {
    const std::vector<double> &r = *prec[i];
    for (int c = 0; c < d; ++c)
      for (int a = 0; a < d; ++a) ds.gt[(size_t)(i * d + c) * d + a] = r[a + c * d];
    for (int a = 0; a < d; ++a) ds.gt[(size_t)(d * ds.n + ds.l + i) * d + a] = r[d * d + a];
  }
  for (int i = 0; i < ds.b; ++i)
    for (int a = 0; a < d; ++a) ds.gt[(size_t)(d * ds.n + ds.l + ds.n + i) * d + a] = (*lrec[i])[a];
  for (const RangeMeasH &m : ds.ranges) {
    double v[3] = {0, 0, 0}, nn = 0;
    for (int a = 0; a < d; ++a) {
      const double t1 = m.type1 ? (*lrec[m.i])[a] : (*prec[m.i])[d * d + a];
      const double t2 = m.type2 ? (*lrec[m.j])[a] : (*prec[m.j])[d * d + a];
      v[a] = t1 - t2;
      nn += v[a] * v[a];
    }
    nn = std::sqrt(nn);
    for (int a = 0; a < d; ++a) ds.gt[(size_t)(d * ds.n + m.l) * d + a] = nn > 0 ? v[a] / nn : 0.0;
  }
  return true;
}

// Closed-form blocks per factor (X = [Y | s | p | L]):
//   pose-pose (i -> j):      1/2 kappa |Y_j - Y_i R|^2 + 1/2 tau |p_j - p_i - Y_i t|^2
//   pose-landmark (i -> L):  1/2 tau |L - p_i - Y_i t|^2
//   range (x_i, x_j, s):     1/2 omega |x_j - x_i + rho s|^2      (s = direction from x_j to x_i)
HostCsr build_Q_ra(const HostRADataset &ds) {
  const int d = ds.d, n = ds.n, l = ds.l;
  const int oS = d * n, oP = d * n + l, oL = d * n + l + n;
  std::vector<int> I, J;
  std::vector<double> V;
  auto put = [&](int i, int j, double v) {
    I.push_back(i);
    J.push_back(j);
    V.push_back(v);
  };
  auto put_sym = [&](int i, int j, double v) {
    put(i, j, v);
    put(j, i, v);
  };
  auto trans_terms = [&](int yi, const double *t, double tau, int ci, int cj) {
    // tau |x_cj - x_ci - Y_i t|^2 expanded
    for (int a = 0; a < d; ++a)
      for (int b = 0; b < d; ++b) put(yi * d + a, yi * d + b, tau * t[a] * t[b]);
    for (int a = 0; a < d; ++a) {
      put_sym(yi * d + a, ci, tau * t[a]);
      put_sym(yi * d + a, cj, -tau * t[a]);
    }
    put(ci, ci, tau);
    put(cj, cj, tau);
    put_sym(ci, cj, -tau);
  };
  for (const PoseMeas &e : ds.pose_pose) {
    const int i = e.p1, j = e.p2;
    const double kap = e.weight * e.kappa, tau = e.weight * e.tau;
    for (int a = 0; a < d; ++a)
      for (int b = 0; b < d; ++b) {
        double rrt = 0;
        for (int c = 0; c < d; ++c) rrt += e.R[a + c * d] * e.R[b + c * d];
        put(i * d + a, i * d + b, kap * rrt);
        put_sym(i * d + a, j * d + b, -kap * e.R[a + b * d]);
      }
    for (int a = 0; a < d; ++a) put(j * d + a, j * d + a, kap);
    trans_terms(i, e.t, tau, oP + i, oP + j);
  }
  for (const PoseLandmarkMeasH &e : ds.pose_landmark) trans_terms(e.i, e.t, e.weight * e.tau, oP + e.i, oL + e.j);
  for (const RangeMeasH &e : ds.ranges) {
    const double w = e.weight * e.precision, rho = e.range;
    const int ci = (e.type1 ? oL : oP) + e.i, cj = (e.type2 ? oL : oP) + e.j, cs = oS + e.l;
    put(cs, cs, w * rho * rho);
    put_sym(cs, cj, w * rho);
    put_sym(cs, ci, -w * rho);
    put(ci, ci, w);
    put(cj, cj, w);
    put_sym(ci, cj, -w);
  }
  const int k = ds.k();
  for (int q = 0; q < k; ++q) put(q, q, 0.0);
  return csr_from_coo(k, k, I, J, V);
}

}  // namespace dcora

namespace dcora {
void ra_agent_columns(const HostRADataset &ds, int robot, int dims3[3], std::vector<int> &own) {
  const int d = ds.d, n = ds.n, l = ds.l;
  own.clear();
  int na = 0, la = 0, ba = 0;
  for (int i = 0; i < n; ++i)
    if (ds.pose_robot[i] == robot) {
      for (int c = 0; c < d; ++c) own.push_back(d * i + c);
      ++na;
    }
  for (int s = 0; s < l; ++s)
    if (ds.sphere_robot[s] == robot) {
      own.push_back(d * n + s);
      ++la;
    }
  for (int i = 0; i < n; ++i)
    if (ds.pose_robot[i] == robot) own.push_back(d * n + l + i);
  for (int j = 0; j < ds.b; ++j)
    if (ds.landmark_robot[j] == robot) {
      own.push_back(d * n + l + n + j);
      ++ba;
    }
  dims3[0] = na;
  dims3[1] = la;
  dims3[2] = ba;
  if (la == 0 && ba == 0) {
    // this ABI reads l = b = 0 as the SE ordering [Y1 p1 ... Yn pn]: list the columns that way
    own.clear();
    for (int i = 0; i < n; ++i)
      if (ds.pose_robot[i] == robot) {
        for (int c = 0; c < d; ++c) own.push_back(d * i + c);
        own.push_back(d * n + l + i);
      }
  }
}

void extract_agent_blocks(const HostCsr &Q, const std::vector<int> &own, HostCsr *Qaa, HostCsr *C) {
  const int ka = (int)own.size();
  std::vector<int> local((size_t)Q.n, -1);
  for (int a = 0; a < ka; ++a) local[own[a]] = a;
  std::vector<int> I1, J1, I2, J2;
  std::vector<double> V1, V2;
  for (int a = 0; a < ka; ++a) {
    const int i = own[a];
    for (int p = Q.rp[i]; p < Q.rp[i + 1]; ++p) {
      const int j = Q.ci[p];
      if (local[j] >= 0) {
        I1.push_back(a);
        J1.push_back(local[j]);
        V1.push_back(Q.v[p]);
      } else {
        I2.push_back(a);
        J2.push_back(j);
        V2.push_back(Q.v[p]);
      }
    }
  }
  if (Qaa) *Qaa = csr_from_coo(ka, ka, I1, J1, V1);
  if (C) *C = csr_from_coo(ka, Q.n, I2, J2, V2);
}
}  // namespace dcora
