// Graph with the reference's construction shape (ref include/DCORA/Graph.h:57-147): a holder of the measurements of
// one robot (or of the whole team, robot 0) from which QuadraticProblem takes (r, d, n) and the connection Laplacian
// Q = A Omega A^T (ref src/Graph.cpp:579-683), built by the library (dcora_graph_build_Q_pgo).  Pose graphs only; the
// range-aided graphs are created from files (dcora_radataset_*).
#pragma once
#include <algorithm>
#include <memory>
#include <vector>

#include "DCORA_types.h"
#include "DCORA_utils.h"

namespace DCORA {

enum class GraphType { PoseGraph, RangeAidedSLAMGraph };

// the measurement arrays of the C ABI: ids m x 4 (r1, p1, r2, p2), vals m x (d d + d + 3) (R column-major, t, kappa,
// tau, weight)
inline void pack_measurements(const std::vector<RelativePosePoseMeasurement> &ms, unsigned d, std::vector<int> *ids,
                              std::vector<double> *vals) {
  const size_t w = (size_t)d * d + d + 3;
  ids->assign(ms.size() * 4, 0);
  vals->assign(ms.size() * w, 0.0);
  for (size_t i = 0; i < ms.size(); ++i) {
    const RelativePosePoseMeasurement &m = ms[i];
    if (m.R.rows() != d || m.R.cols() != d || m.t.size() != d)
      throw std::invalid_argument("measurement of the wrong dimension");
    (*ids)[4 * i] = (int)m.r1;
    (*ids)[4 * i + 1] = (int)m.p1;
    (*ids)[4 * i + 2] = (int)m.r2;
    (*ids)[4 * i + 3] = (int)m.p2;
    double *v = &(*vals)[i * w];
    for (unsigned c = 0; c < d; ++c)
      for (unsigned a = 0; a < d; ++a) v[c * d + a] = m.R(a, c);
    for (unsigned a = 0; a < d; ++a) v[d * d + a] = m.t[a];
    v[d * d + d] = m.kappa;
    v[d * d + d + 1] = m.tau;
    v[d * d + d + 2] = m.weight;
  }
}

class Graph {
 public:
  Graph(unsigned id, unsigned r, unsigned d, GraphType graphType = GraphType::PoseGraph) : id_(id), r_(r), d_(d) {
    if (graphType != GraphType::PoseGraph)
      throw std::invalid_argument("Graph: range-aided graphs are created from pyfg files (dcora_radataset_*)");
  }
  unsigned id() const { return id_; }
  unsigned r() const { return r_; }
  unsigned d() const { return d_; }
  unsigned n() const { return n_; }
  // ref src/Graph.cpp (setMeasurements): the poses of this robot are those the measurements name
  void setMeasurements(const std::vector<RelativePosePoseMeasurement> &measurements) {
    meas_ = measurements;
    n_ = 0;
    for (const RelativePosePoseMeasurement &m : meas_) {
      if (m.r1 == id_) n_ = std::max<unsigned>(n_, (unsigned)m.p1 + 1);
      if (m.r2 == id_) n_ = std::max<unsigned>(n_, (unsigned)m.p2 + 1);
    }
    built_ = false;
  }
  const std::vector<RelativePosePoseMeasurement> &measurements() const { return meas_; }
  // Graph::quadraticMatrix() (ref src/Graph.cpp:523-533: built lazily, kept)
  const SparseMatrix &quadraticMatrix() {
    if (!built_) {
      std::vector<int> ids;
      std::vector<double> vals;
      pack_measurements(meas_, d_, &ids, &vals);
      dcora_csr_t h = nullptr;
      detail::check(dcora_graph_build_Q_pgo((int)d_, (int)n_, (int)id_, (int)meas_.size(), ids.data(), vals.data(), &h),
                    "Graph::quadraticMatrix");
      Q_ = detail::take(h);
      built_ = true;
    }
    return Q_;
  }

 private:
  unsigned id_, r_, d_, n_ = 0;
  std::vector<RelativePosePoseMeasurement> meas_;
  SparseMatrix Q_;
  bool built_ = false;
};

}  // namespace DCORA
