// ORACLE -- test infrastructure only (see oracle.hpp).
// Range-aided SLAM data feed for the centralised (single-agent) problem: pyfg reader with the reference's global
// re-indexing (ref: src/DCORA_utils.cpp:437-1167, 1169-1365) and Q = Q_p + Q_r built through the incidence /
// data / selection matrices exactly as Graph::constructQuadraticCostTermRASLAM does (ref: src/Graph.cpp:824-1188).
#include <algorithm>
#include <fstream>
#include <set>
#include <sstream>
#include <stdexcept>

#include "oracle.hpp"

namespace orc {

namespace {
struct Sym {
  int type;  // 0 pose, 1 landmark
  int robot, id;
};
// ref: src/DCORA_utils.cpp:584-616 getRobotAndStateIDFromSymbol ('A' = robot 0, 'L' = landmark, 'M' = map)
Sym parse_symbol(const std::string &s) {
  Sym o{0, 0, 0};
  if (s[0] == 'L') {
    o.type = 1;
    if (std::isupper((unsigned char)s[1])) {
      o.robot = s[1] - 'A';
      o.id = std::stoi(s.substr(2));
    } else {
      o.robot = 'M' - 'A';
      o.id = std::stoi(s.substr(1));
    }
  } else if (std::isupper((unsigned char)s[0])) {
    o.robot = s[0] - 'A';
    o.id = std::stoi(s.substr(1));
  } else {
    throw std::runtime_error("bad pyfg symbol " + s);
  }
  return o;
}
void quat_R(double x, double y, double z, double w, double R[9]) {
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
// rows of a sparse matrix as (col, val) lists; returns A diag(w) B^T as triplets with offsets
using Rows = std::vector<std::vector<std::pair<int, double>>>;
void AwBt(const Rows &A, const std::vector<double> &w, const Rows &B, int ncols, int ro, int co, std::vector<int> &I,
          std::vector<int> &J, std::vector<double> &V, bool also_transposed) {
  Rows Ac(ncols), Bc(ncols);
  for (int a = 0; a < (int)A.size(); ++a)
    for (auto &e : A[a]) Ac[e.first].emplace_back(a, e.second);
  for (int b = 0; b < (int)B.size(); ++b)
    for (auto &e : B[b]) Bc[e.first].emplace_back(b, e.second);
  for (int c = 0; c < ncols; ++c)
    for (auto &ea : Ac[c])
      for (auto &eb : Bc[c]) {
        const double v = ea.second * w[c] * eb.second;
        I.push_back(ro + ea.first);
        J.push_back(co + eb.first);
        V.push_back(v);
        if (also_transposed) {
          I.push_back(co + eb.first);
          J.push_back(ro + ea.first);
          V.push_back(v);
        }
      }
}
}  // namespace

RADataset read_pyfg(const std::string &path) {
  std::ifstream in(path);
  if (!in) throw std::runtime_error("cannot open " + path);
  struct PoseRec { int robot, id; double R[9], t[3]; };
  struct LmRec { int robot, id; double t[3]; };
  std::vector<PoseRec> poses;
  std::vector<LmRec> lms;
  struct PP { Sym a, b; Meas m; };
  struct PL { Sym a, b; PoseLandmarkMeas m; };
  struct RG { Sym a, b; RangeMeas m; int robot_l; };
  std::vector<PP> pps;
  std::vector<PL> pls;
  std::vector<RG> rgs;
  std::map<int, int> sphere_count;
  std::set<std::pair<std::pair<int, std::pair<int, int>>, std::pair<int, std::pair<int, int>>>> seen_ranges;
  RADataset ds;
  std::string line, tok, s1, s2;
  double ts;
  while (std::getline(in, line)) {
    std::istringstream ss(line);
    if (!(ss >> tok)) continue;
    if (tok == "VERTEX_SE2") {
      ds.d = 2;
      double x, y, th;
      ss >> ts >> s1 >> x >> y >> th;
      Sym a = parse_symbol(s1);
      PoseRec p{a.robot, a.id, {std::cos(th), std::sin(th), -std::sin(th), std::cos(th)}, {x, y, 0}};
      poses.push_back(p);
    } else if (tok == "VERTEX_SE3:QUAT") {
      ds.d = 3;
      double x, y, z, qx, qy, qz, qw;
      ss >> ts >> s1 >> x >> y >> z >> qx >> qy >> qz >> qw;
      Sym a = parse_symbol(s1);
      PoseRec p{a.robot, a.id, {0}, {x, y, z}};
      quat_R(qx, qy, qz, qw, p.R);
      poses.push_back(p);
    } else if (tok == "VERTEX_XY" || tok == "VERTEX_XYZ") {
      const int d = tok == "VERTEX_XY" ? 2 : 3;
      ss >> s1;
      Sym a = parse_symbol(s1);
      LmRec l{a.robot, a.id, {0, 0, 0}};
      for (int i = 0; i < d; ++i) ss >> l.t[i];
      lms.push_back(l);
    } else if (tok == "EDGE_SE2") {
      double x, y, th, c[6];
      ss >> ts >> s1 >> s2 >> x >> y >> th;
      for (double &q : c) ss >> q;  // cov11 cov12 cov13 cov22 cov23 cov33
      PP e;
      e.a = parse_symbol(s1);
      e.b = parse_symbol(s2);
      e.m.t[0] = x;
      e.m.t[1] = y;
      e.m.R[0] = std::cos(th);
      e.m.R[1] = std::sin(th);
      e.m.R[2] = -std::sin(th);
      e.m.R[3] = std::cos(th);
      e.m.tau = 2.0 / (c[0] + c[3]);  // getTau: 2 / trace(cov_t)   (:546-557)
      e.m.kappa = 1.0 / c[5];         // getKappa 2D: 1 / cov_R     (:566-573)
      pps.push_back(e);
    } else if (tok == "EDGE_SE3:QUAT") {
      double x, y, z, qx, qy, qz, qw, c[21];
      ss >> ts >> s1 >> s2 >> x >> y >> z >> qx >> qy >> qz >> qw;
      for (double &q : c) ss >> q;
      PP e;
      e.a = parse_symbol(s1);
      e.b = parse_symbol(s2);
      e.m.t[0] = x;
      e.m.t[1] = y;
      e.m.t[2] = z;
      quat_R(qx, qy, qz, qw, e.m.R);
      e.m.tau = 3.0 / (c[0] + c[6] + c[11]);           // 3 / trace(cov_t)
      e.m.kappa = 3.0 / (2.0 * (c[15] + c[18] + c[20]));  // 3 / (2 trace(cov_R))
      pps.push_back(e);
    } else if (tok == "EDGE_SE2_XY" || tok == "EDGE_SE3_XYZ") {
      const int d = tok == "EDGE_SE2_XY" ? 2 : 3;
      PL e;
      ss >> ts >> s1 >> s2;
      e.a = parse_symbol(s1);
      e.b = parse_symbol(s2);
      for (int i = 0; i < d; ++i) ss >> e.m.t[i];
      std::vector<double> c(d * (d + 1) / 2);
      for (double &q : c) ss >> q;
      const double tr = d == 2 ? c[0] + c[2] : c[0] + c[3] + c[5];
      e.m.tau = d / tr;
      pls.push_back(e);
    } else if (tok == "EDGE_RANGE") {
      RG e;
      double range, cov;
      ss >> ts >> s1 >> s2 >> range >> cov;
      e.a = parse_symbol(s1);
      e.b = parse_symbol(s2);
      auto key = std::make_pair(std::make_pair(e.a.type, std::make_pair(e.a.robot, e.a.id)),
                                std::make_pair(e.b.type, std::make_pair(e.b.robot, e.b.id)));
      if (!seen_ranges.insert(key).second) continue;  // duplicate range edges are skipped (:1078-1086)
      e.m.range = range;
      e.m.precision = 1.0 / cov;
      e.robot_l = sphere_count[e.a.robot]++;  // the source robot owns the unit sphere (:1092-1097)
      rgs.push_back(e);
    }  // priors and other records are not used by the centralised flow
  }
  // global indices (std::map order of the reference's ground-truth dictionaries)
  std::map<std::pair<int, int>, int> pidx, lidx;
  std::sort(poses.begin(), poses.end(), [](const PoseRec &a, const PoseRec &b) {
    return std::make_pair(a.robot, a.id) < std::make_pair(b.robot, b.id);
  });
  std::sort(lms.begin(), lms.end(), [](const LmRec &a, const LmRec &b) {
    return std::make_pair(a.robot, a.id) < std::make_pair(b.robot, b.id);
  });
  for (size_t i = 0; i < poses.size(); ++i) pidx[{poses[i].robot, poses[i].id}] = (int)i;
  for (size_t i = 0; i < lms.size(); ++i) lidx[{lms[i].robot, lms[i].id}] = (int)i;
  std::map<int, int> sphere_base;
  {
    int acc = 0;
    for (auto &kv : sphere_count) {
      sphere_base[kv.first] = acc;
      acc += kv.second;
    }
    ds.l = acc;
  }
  ds.n = (int)poses.size();
  ds.b = (int)lms.size();
  for (auto &e : pps) {
    Meas m = e.m;
    m.r1 = m.r2 = 0;
    m.p1 = pidx.at({e.a.robot, e.a.id});
    m.p2 = pidx.at({e.b.robot, e.b.id});
    m.weight = 1;
    ds.pose_pose.push_back(m);
  }
  for (auto &e : pls) {
    PoseLandmarkMeas m = e.m;
    m.i = pidx.at({e.a.robot, e.a.id});
    m.j = lidx.at({e.b.robot, e.b.id});
    ds.pose_landmark.push_back(m);
  }
  for (auto &e : rgs) {
    RangeMeas m = e.m;
    m.type1 = e.a.type;
    m.type2 = e.b.type;
    m.i = e.a.type ? lidx.at({e.a.robot, e.a.id}) : pidx.at({e.a.robot, e.a.id});
    m.j = e.b.type ? lidx.at({e.b.robot, e.b.id}) : pidx.at({e.b.robot, e.b.id});
    m.l = sphere_base.at(e.a.robot) + e.robot_l;
    ds.ranges.push_back(m);
  }
  // ground truth in RA ordering
  const int d = ds.d, k = (d + 1) * ds.n + ds.l + ds.b;
  ds.gt = Mat(d, k);
  for (int i = 0; i < ds.n; ++i) {
    for (int c = 0; c < d; ++c)
      for (int a = 0; a < d; ++a) ds.gt(a, i * d + c) = poses[i].R[a + c * d];
    for (int a = 0; a < d; ++a) ds.gt(a, d * ds.n + ds.l + i) = poses[i].t[a];
  }
  for (int i = 0; i < ds.b; ++i)
    for (int a = 0; a < d; ++a) ds.gt(a, d * ds.n + ds.l + ds.n + i) = lms[i].t[a];
  for (const RangeMeas &m : ds.ranges) {
    double v[3], nn = 0;
    for (int a = 0; a < d; ++a) {
      const double t1 = m.type1 ? lms[m.i].t[a] : poses[m.i].t[a];
      const double t2 = m.type2 ? lms[m.j].t[a] : poses[m.j].t[a];
      v[a] = t1 - t2;  // (state1 - state2).normalized()  (:1138-1140)
      nn += v[a] * v[a];
    }
    nn = std::sqrt(nn);
    for (int a = 0; a < d; ++a) ds.gt(a, d * ds.n + m.l) = v[a] / nn;
  }
  return ds;
}

// ref: src/Graph.cpp:824-1188 with every state owned by this agent
CSR build_Q_ra(const RADataset &ds) {
  const int d = ds.d, n = ds.n, l = ds.l, b = ds.b;
  const int mPP = (int)ds.pose_pose.size(), mPL = (int)ds.pose_landmark.size(), mR = (int)ds.ranges.size();
  const int mPose = mPP + mPL;
  Rows ARhoT(d * n), ATauT(n + b), TT(d * n), CT(n + b), PT(l);
  std::vector<double> oRho((size_t)d * mPP), oTau(mPose), oRange(mR), oRangeD2(mR), oRangeD(mR);
  for (int k = 0; k < mPP; ++k) {
    const Meas &e = ds.pose_pose[k];
    for (int a = 0; a < d; ++a) oRho[(size_t)k * d + a] = e.weight * e.kappa;
    oTau[k] = e.weight * e.tau;
    const int i = e.p1, j = e.p2;
    for (int c = 0; c < d; ++c)
      for (int a = 0; a < d; ++a) ARhoT[i * d + a].emplace_back(k * d + c, -e.R[a + c * d]);
    for (int a = 0; a < d; ++a) TT[i * d + a].emplace_back(k, -e.t[a]);
    ATauT[i].emplace_back(k, -1.0);
    for (int a = 0; a < d; ++a) ARhoT[j * d + a].emplace_back(k * d + a, 1.0);
    ATauT[j].emplace_back(k, 1.0);
  }
  for (int q = 0; q < mPL; ++q) {
    const PoseLandmarkMeas &e = ds.pose_landmark[q];
    const int k = mPP + q;
    oTau[k] = e.weight * e.tau;
    for (int a = 0; a < d; ++a) TT[e.i * d + a].emplace_back(k, -e.t[a]);
    ATauT[e.i].emplace_back(k, -1.0);
    ATauT[n + e.j].emplace_back(k, 1.0);
  }
  for (int k = 0; k < mR; ++k) {
    const RangeMeas &e = ds.ranges[k];
    oRange[k] = e.weight * e.precision;
    oRangeD[k] = e.range * oRange[k];
    oRangeD2[k] = e.range * e.range * oRange[k];
    PT[e.l].emplace_back(k, 1.0);
    CT[(e.type1 ? n : 0) + e.i].emplace_back(k, -1.0);
    CT[(e.type2 ? n : 0) + e.j].emplace_back(k, 1.0);
  }
  std::vector<int> I, J;
  std::vector<double> V;
  const int o2 = d * n, o3 = d * n + l;
  AwBt(ARhoT, oRho, ARhoT, d * mPP, 0, 0, I, J, V, false);   // Q11 = ARhoT OmegaRho ARho
  AwBt(TT, oTau, TT, mPose, 0, 0, I, J, V, false);           //     + TT OmegaTau T
  AwBt(TT, oTau, ATauT, mPose, 0, o3, I, J, V, true);        // Q13 = TT OmegaTau ATau (+ transpose)
  AwBt(PT, oRangeD2, PT, mR, o2, o2, I, J, V, false);        // Q22 = PT OmegaRange D D P
  AwBt(PT, oRangeD, CT, mR, o2, o3, I, J, V, true);          // Q23 = PT D OmegaRange C (+ transpose)
  AwBt(ATauT, oTau, ATauT, mPose, o3, o3, I, J, V, false);   // Q33 = ATauT OmegaTau ATau
  AwBt(CT, oRange, CT, mR, o3, o3, I, J, V, false);          //     + CT OmegaRange C
  return csr_from_triplets((d + 1) * n + l + b, I, J, V);
}

}  // namespace orc

namespace orc {
// Odometry initialisation of the centralised CORA driver (ref: examples/SingleRobotExample_RASLAM.cpp:92-150 with
// odometryInitialization, src/DCORA_solver.cpp:270-302): every chain of consecutive-index pose-pose edges starts at
// its ground-truth first pose and is propagated T_dst = T_src * T_meas; unit spheres = ground truth; landmarks =
// uniform(-1, 1) (Matrix::Random in the reference; here a seeded splitmix64 stream so the start point is
// reproducible).  Returns d x k in the RA ordering.  The pyfg reader merges the robots into one index range, so a
// chain starts wherever pose i has no edge (i-1 -> i).
Mat ra_odometry_initialization(const RADataset &ds, uint64_t seed) {
  const int d = ds.d, n = ds.n, l = ds.l, b = ds.b;
  Mat X(d, (d + 1) * n + l + b);
  std::vector<int> into((size_t)n, -1);
  for (int e = 0; e < (int)ds.pose_pose.size(); ++e) {
    const Meas &m = ds.pose_pose[e];
    if (m.p1 + 1 == m.p2 && into[m.p2] < 0) into[m.p2] = e;
  }
  const int ot = d * n + l;
  for (int i = 0; i < n; ++i) {
    if (into[i] < 0) {
      for (int c = 0; c < d; ++c)
        for (int a = 0; a < d; ++a) X(a, d * i + c) = ds.gt(a, d * i + c);
      for (int a = 0; a < d; ++a) X(a, ot + i) = ds.gt(a, ot + i);
      continue;
    }
    const Meas &m = ds.pose_pose[into[i]];
    for (int c = 0; c < d; ++c)
      for (int a = 0; a < d; ++a) {
        double s = 0;
        for (int q = 0; q < d; ++q) s += X(a, d * (i - 1) + q) * m.R[q + c * d];
        X(a, d * i + c) = s;
      }
    for (int a = 0; a < d; ++a) {
      double s = X(a, ot + i - 1);
      for (int q = 0; q < d; ++q) s += X(a, d * (i - 1) + q) * m.t[q];
      X(a, ot + i) = s;
    }
  }
  for (int j = 0; j < l; ++j)
    for (int a = 0; a < d; ++a) X(a, d * n + j) = ds.gt(a, d * n + j);
  uint64_t s = seed;
  for (int j = 0; j < b; ++j)
    for (int a = 0; a < d; ++a) X(a, ot + n + j) = 2.0 * u01(s) - 1.0;
  return X;
}
}  // namespace orc
