// ORACLE -- test infrastructure only (see oracle.hpp).
// Data feed: g2o reader (ref: src/DCORA_utils.cpp:179-375) and the PGO data
// matrices Q = AbT Omega AbT^T (ref: src/Graph.cpp:579-683) and G (ref:
// src/Graph.cpp:685-822), built literally through the incidence matrix.
#include <fstream>
#include <sstream>
#include <stdexcept>

#include "oracle.hpp"

namespace orc {

static void inv3_trace(const double A[9], int d, double *trace_inv) {
  // trace of the inverse of a symmetric d x d matrix (d = 2 or 3)
  if (d == 2) {
    const double det = A[0] * A[3] - A[1] * A[2];
    *trace_inv = (A[3] + A[0]) / det;
    return;
  }
  const double a = A[0], b = A[3], c = A[6], d1 = A[1], e = A[4], f = A[7], g = A[2], h = A[5], i = A[8];
  const double det = a * (e * i - f * h) - b * (d1 * i - f * g) + c * (d1 * h - e * g);
  const double c00 = (e * i - f * h), c11 = (a * i - c * g), c22 = (a * e - b * d1);
  *trace_inv = (c00 + c11 + c22) / det;
}

static void quat_to_R(double qx, double qy, double qz, double qw, double R[9]) {
  // Eigen::Quaterniond::toRotationMatrix (no normalisation), column-major out
  const double tx = 2 * qx, ty = 2 * qy, tz = 2 * qz;
  const double twx = tx * qw, twy = ty * qw, twz = tz * qw;
  const double txx = tx * qx, txy = ty * qx, txz = tz * qx;
  const double tyy = ty * qy, tyz = tz * qy, tzz = tz * qz;
  R[0] = 1 - (tyy + tzz);
  R[3] = txy - twz;
  R[6] = txz + twy;
  R[1] = txy + twz;
  R[4] = 1 - (txx + tzz);
  R[7] = tyz - twx;
  R[2] = txz - twy;
  R[5] = tyz + twx;
  R[8] = 1 - (txx + tyy);
}

Dataset read_g2o(const std::string &path) {
  std::ifstream in(path);
  if (!in) throw std::runtime_error("cannot open " + path);
  Dataset ds;
  std::string line, tok;
  unsigned maxid = 0;
  while (std::getline(in, line)) {
    std::stringstream ss(line);
    if (!(ss >> tok)) continue;
    if (tok == "VERTEX_SE2") {
      if (!ds.d) ds.d = 2;
      continue;
    }
    if (tok == "VERTEX_SE3:QUAT") {
      if (!ds.d) ds.d = 3;
      continue;
    }
    Meas m;
    if (tok == "EDGE_SE2") {
      if (!ds.d) ds.d = 2;
      double dx, dy, dth, I11, I12, I13, I22, I23, I33;
      size_t i, j;
      ss >> i >> j >> dx >> dy >> dth >> I11 >> I12 >> I13 >> I22 >> I23 >> I33;
      m.p1 = (int)i;
      m.p2 = (int)j;
      m.t[0] = dx;
      m.t[1] = dy;
      m.R[0] = std::cos(dth);
      m.R[1] = std::sin(dth);
      m.R[2] = -std::sin(dth);
      m.R[3] = std::cos(dth);
      const double T[9] = {I11, I12, I12, I22};
      double tr;
      inv3_trace(T, 2, &tr);
      m.tau = 2 / tr;      // :288-290
      m.kappa = I33;       // :292
    } else if (tok == "EDGE_SE3:QUAT") {
      if (!ds.d) ds.d = 3;
      double dx, dy, dz, qx, qy, qz, qw;
      double I[21];
      size_t i, j;
      ss >> i >> j >> dx >> dy >> dz >> qx >> qy >> qz >> qw;
      for (double &x : I) ss >> x;
      // order: I11 I12 I13 I14 I15 I16 I22 I23 I24 I25 I26 I33 I34 I35 I36 I44 I45 I46 I55 I56 I66
      m.p1 = (int)i;
      m.p2 = (int)j;
      m.t[0] = dx;
      m.t[1] = dy;
      m.t[2] = dz;
      quat_to_R(qx, qy, qz, qw, m.R);
      const double T[9] = {I[0], I[1], I[2], I[1], I[6], I[7], I[2], I[7], I[11]};
      double tr;
      inv3_trace(T, 3, &tr);
      m.tau = 3 / tr;  // :337-339
      const double Rc[9] = {I[15], I[16], I[17], I[16], I[18], I[19], I[17], I[19], I[20]};
      inv3_trace(Rc, 3, &tr);
      m.kappa = 3 / (2 * tr);  // :344-346
    } else {
      throw std::runtime_error("unrecognized g2o token: " + tok);
    }
    maxid = std::max<unsigned>(maxid, std::max(m.p1, m.p2));
    ds.meas.push_back(m);
  }
  ds.n = (int)maxid + 1;
  return ds;
}

// ref: src/Graph.cpp:579-683.  AbT is (dh n) x (dh m); for measurement k the
// source pose i owned by this agent gets the block -T (T = [R t; 0 1]) and the
// destination pose j owned by this agent gets +I.  Q = AbT Omega AbT^T.
CSR build_Q_pgo(int d, int n, int id, const std::vector<Meas> &all) {
  const int dh = d + 1;
  const int m = (int)all.size();
  // rows of AbT stored as (col, val) lists
  std::vector<std::vector<std::pair<int, double>>> rows((size_t)dh * n);
  std::vector<double> omega((size_t)dh * m);
  for (int k = 0; k < m; ++k) {
    const Meas &e = all[k];
    for (int a = 0; a < d; ++a) omega[(size_t)k * dh + a] = e.weight * e.kappa;
    omega[(size_t)k * dh + d] = e.weight * e.tau;
    const int i = (e.r1 == id) ? e.p1 : -1;
    const int j = (e.r2 == id) ? e.p2 : -1;
    if (i >= 0) {
      for (int c = 0; c < d; ++c)
        for (int a = 0; a < d; ++a) rows[(size_t)i * dh + a].emplace_back(k * dh + c, -e.R[a + c * d]);
      for (int a = 0; a < d; ++a) rows[(size_t)i * dh + a].emplace_back(k * dh + d, -e.t[a]);
      rows[(size_t)i * dh + d].emplace_back(k * dh + d, -1.0);
    }
    if (j >= 0)
      for (int a = 0; a < dh; ++a) rows[(size_t)j * dh + a].emplace_back(k * dh + a, 1.0);
  }
  // Q(a,b) = sum_c AbT(a,c) omega(c) AbT(b,c): go through columns
  std::vector<std::vector<std::pair<int, double>>> cols((size_t)dh * m);
  for (int a = 0; a < dh * n; ++a)
    for (auto &e : rows[a]) cols[e.first].emplace_back(a, e.second);
  std::vector<int> I, J;
  std::vector<double> V;
  for (int c = 0; c < dh * m; ++c)
    for (auto &ea : cols[c])
      for (auto &eb : cols[c]) {
        I.push_back(ea.first);
        J.push_back(eb.first);
        V.push_back(ea.second * omega[c] * eb.second);
      }
  return csr_from_triplets(dh * n, I, J, V);
}

// ref: src/Graph.cpp:685-822 (without priors).  For a shared edge whose source
// i is mine: G_i += Xc * I * Omega * (-T)^T ; whose destination j is mine:
// G_j += Xc * (-T) * Omega * I.
bool build_G_pgo(int r, int d, int n, int id, const std::vector<Meas> &shared, const PoseDict &nbr,
                 Mat &G) {
  const int dh = d + 1;
  G = Mat(r, dh * n);
  std::vector<double> T((size_t)dh * dh), Om((size_t)dh), W((size_t)dh * dh);
  for (const Meas &e : shared) {
    std::fill(T.begin(), T.end(), 0.0);
    for (int c = 0; c < d; ++c)
      for (int a = 0; a < d; ++a) T[a + c * dh] = e.R[a + c * d];
    for (int a = 0; a < d; ++a) T[a + d * dh] = e.t[a];
    T[d + d * dh] = 1;
    for (int a = 0; a < d; ++a) Om[a] = e.weight * e.kappa;
    Om[d] = e.weight * e.tau;
    if (e.r1 == id && e.r2 != id) {
      auto it = nbr.find({e.r2, e.p2});
      if (it == nbr.end()) return false;
      const double *Xc = it->second.data();  // r x dh
      // W = Omega * (-T)^T  => W(a,b) = -Om[a] * T(b,a)
      for (int a = 0; a < dh; ++a)
        for (int b = 0; b < dh; ++b) W[a + b * dh] = -Om[a] * T[b + a * dh];
      double *Gi = G.col(e.p1 * dh);
      for (int b = 0; b < dh; ++b)
        for (int a = 0; a < dh; ++a) {
          const double w = W[a + b * dh];
          if (w != 0)
            for (int t = 0; t < r; ++t) Gi[b * r + t] += Xc[a * r + t] * w;
        }
    } else if (e.r2 == id && e.r1 != id) {
      auto it = nbr.find({e.r1, e.p1});
      if (it == nbr.end()) return false;
      const double *Xc = it->second.data();
      // W = (-T) * Omega => W(a,b) = -T(a,b) Om[b]
      for (int a = 0; a < dh; ++a)
        for (int b = 0; b < dh; ++b) W[a + b * dh] = -T[a + b * dh] * Om[b];
      double *Gj = G.col(e.p2 * dh);
      for (int b = 0; b < dh; ++b)
        for (int a = 0; a < dh; ++a) {
          const double w = W[a + b * dh];
          if (w != 0)
            for (int t = 0; t < r; ++t) Gj[b * r + t] += Xc[a * r + t] * w;
        }
    }
  }
  return true;
}

}  // namespace orc

namespace orc {
// chordalInitialization (ref: src/DCORA_solver.cpp:218-268, B matrices src/DCORA_utils.cpp:1542-1630,
// recoverTranslations :1632-1659).  The two linear least-squares problems
//   min sum_e kappa_e |R_j - R_i R_ij|_F^2   (R_0 = I)       and      min sum_e tau_e |t_j - t_i - R_i t_ij|^2  (t_0 = 0)
// are solved through their normal equations (reduced rotation / translation Laplacians) with the sparse Cholesky of
// this oracle instead of SPQR; the minimisers are the same.
Mat chordal_initialization(const Dataset &ds) {
  const int d = ds.d, n = ds.n, dh = d + 1;
  // rotation Laplacian (d n x d n), row-vector convention: cost = tr(R L R^T), R = [R_0 ... R_{n-1}] (d x d n)
  std::vector<int> I, J;
  std::vector<double> V;
  for (const Meas &e : ds.meas) {
    const int i = e.p1, j = e.p2;
    const double k = e.kappa;
    for (int a = 0; a < d; ++a) {
      I.push_back(i * d + a); J.push_back(i * d + a); V.push_back(k);
      I.push_back(j * d + a); J.push_back(j * d + a); V.push_back(k);
      for (int b = 0; b < d; ++b) {
        I.push_back(i * d + a); J.push_back(j * d + b); V.push_back(-k * e.R[a + b * d]);
        I.push_back(j * d + b); J.push_back(i * d + a); V.push_back(-k * e.R[a + b * d]);
      }
    }
  }
  CSR L = csr_from_triplets(d * n, I, J, V);
  // reduced system: unknown columns d..dn-1; rhs = -(R_0 L_{0,red}) with R_0 = I
  const int m = d * (n - 1);
  std::vector<int> I2, J2;
  std::vector<double> V2;
  Mat rhs(d, m);
  for (int row = 0; row < d * n; ++row)
    for (int p = L.rp[row]; p < L.rp[row + 1]; ++p) {
      const int c = L.ci[p];
      if (row >= d && c >= d) {
        I2.push_back(row - d); J2.push_back(c - d); V2.push_back(L.v[p]);
      } else if (row < d && c >= d) {
        rhs(row, c - d) -= L.v[p];  // R_0 = I: row `row` of R_0 has a 1 in column `row`
      }
    }
  CSR Lr = csr_from_triplets(m, I2, J2, V2);
  Chol ch;
  if (!ch.factor(Lr, d)) return Mat();
  Mat Rred;
  ch.solve_rows(rhs, Rred);
  Mat T(d, dh * n);
  for (int a = 0; a < d; ++a) T(a, a) = 1;
  for (int i = 1; i < n; ++i) {
    double blk[9], out[9];
    for (int c = 0; c < d; ++c)
      for (int a = 0; a < d; ++a) blk[a + c * d] = Rred(a, (i - 1) * d + c);
    project_to_rotation_group(d, blk, out);
    for (int c = 0; c < d; ++c)
      for (int a = 0; a < d; ++a) T(a, i * dh + c) = out[a + c * d];
  }
  // translations: weighted graph Laplacian, t_0 = 0
  I.clear(); J.clear(); V.clear();
  Mat b(d, n);
  for (const Meas &e : ds.meas) {
    const int i = e.p1, j = e.p2;
    I.push_back(i); J.push_back(i); V.push_back(e.tau);
    I.push_back(j); J.push_back(j); V.push_back(e.tau);
    I.push_back(i); J.push_back(j); V.push_back(-e.tau);
    I.push_back(j); J.push_back(i); V.push_back(-e.tau);
    for (int a = 0; a < d; ++a) {
      double rt = 0;
      for (int c = 0; c < d; ++c) rt += T(a, i * dh + c) * e.t[c];
      b(a, j) += e.tau * rt;
      b(a, i) -= e.tau * rt;
    }
  }
  CSR Lt = csr_from_triplets(n, I, J, V);
  I2.clear(); J2.clear(); V2.clear();
  for (int row = 1; row < n; ++row)
    for (int p = Lt.rp[row]; p < Lt.rp[row + 1]; ++p)
      if (Lt.ci[p] >= 1) {
        I2.push_back(row - 1); J2.push_back(Lt.ci[p] - 1); V2.push_back(Lt.v[p]);
      }
  CSR Ltr = csr_from_triplets(n - 1, I2, J2, V2);
  Mat bred(d, n - 1), tred;
  for (int j = 1; j < n; ++j)
    for (int a = 0; a < d; ++a) bred(a, j - 1) = b(a, j);
  Chol ct;
  if (!ct.factor(Ltr, 1)) return Mat();
  ct.solve_rows(bred, tred);
  for (int j = 1; j < n; ++j)
    for (int a = 0; a < d; ++a) T(a, j * dh + d) = tred(a, j - 1);
  return T;
}
}  // namespace orc
