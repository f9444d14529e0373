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
