// Graph with the reference's construction shape (ref include/DCORA/Graph.h:57-147): a holder of the measurements of
// one robot (or of the whole team, robot 0) from which QuadraticProblem takes (r, d, n) and the connection Laplacian
// Q = A Omega A^T (ref src/Graph.cpp:579-683, 824-1188), built by the library (dcora_graph_build_Q_pgo for pose graphs,
// dcora_radataset_create + dcora_radataset_build_Q for range-aided graphs: poses, landmarks, one unit sphere per range
// measurement, RA ordering).  l(), b(), linearMatrix() and the preconditioner's regularisation are what
// QuadraticProblem(shared_ptr<Graph>) reads besides (ref src/QuadraticProblem.cpp:19-34, 42, 58, 78-79).
#pragma once
#include <algorithm>
#include <memory>
#include <vector>

#include "DCORA_types.h"
#include "DCORA_utils.h"
#include "RangeAided.h"

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
  Graph(unsigned id, unsigned r, unsigned d, GraphType graphType = GraphType::PoseGraph)
      : id_(id), r_(r), d_(d), type_(graphType) {}
  unsigned id() const { return id_; }
  unsigned r() const { return r_; }
  unsigned d() const { return d_; }
  unsigned n() const { return n_; }
  unsigned l() const { return l_; }  // unit spheres: one per range measurement
  unsigned b() const { return b_; }  // landmarks
  unsigned k() const { return (d_ + 1) * n_ + l_ + b_; }
  GraphType graphType() const { return type_; }
  // ref src/Graph.cpp:68-75: by graph TYPE -- a range-aided graph that holds neither ranges nor landmarks still lives on
  // the RA manifold (RA column ordering)
  bool isPGOCompatible() const { return type_ == GraphType::PoseGraph; }
  int layout() const { return type_ == GraphType::PoseGraph ? DCORA_LAYOUT_SE : DCORA_LAYOUT_RA; }
  // ref src/Graph.cpp (setMeasurements): the poses of this robot are those the measurements name
  void setMeasurements(const std::vector<RelativePosePoseMeasurement> &measurements) {
    meas_ = measurements;
    pose_landmark_.clear();
    ranges_.clear();
    count_states();
  }
  // ref src/Graph.cpp:374-470: all three kinds (updateNumPosesAndLandmarks, updateNumUnitSpheres)
  void setMeasurements(const RelativeMeasurements &measurements) {
    if (type_ != GraphType::RangeAidedSLAMGraph && (!measurements.GetRelativePoseLandmarkMeasurements().empty() ||
                                                    !measurements.GetRangeMeasurements().empty()))
      throw std::invalid_argument("Graph: landmark / range measurements need GraphType::RangeAidedSLAMGraph");
    meas_ = measurements.GetRelativePosePoseMeasurements();
    pose_landmark_ = measurements.GetRelativePoseLandmarkMeasurements();
    ranges_ = measurements.GetRangeMeasurements();
    count_states();
  }
  const std::vector<RelativePosePoseMeasurement> &measurements() const { return meas_; }
  // Graph::quadraticMatrix() (ref src/Graph.cpp:523-533: built lazily, kept)
  const SparseMatrix &quadraticMatrix() {
    if (!built_) {
      dcora_csr_t h = nullptr;
      if (type_ == GraphType::PoseGraph) {
        std::vector<int> ids;
        std::vector<double> vals;
        pack_measurements(meas_, d_, &ids, &vals);
        detail::check(dcora_graph_build_Q_pgo((int)d_, (int)n_, (int)id_, (int)meas_.size(), ids.data(), vals.data(), &h),
                      "Graph::quadraticMatrix");
      } else {
        dcora_radataset_t ds = ra_dataset();
        const int rc = dcora_radataset_build_Q(ds, &h);
        dcora_radataset_destroy(ds);
        detail::check(rc, "Graph::quadraticMatrix");
      }
      Q_ = detail::take(h);
      built_ = true;
    }
    return Q_;
  }
  // Graph::linearMatrix() (ref src/Graph.cpp:535-540): the coupling with the neighbours' public states.  A graph
  // that holds all measurements of its states (the centralised agent; an agent of the device-resident session gets its
  // coupling there) has none: r x k zeros.
  Matrix linearMatrix() const { return Matrix(r_, k()); }
  // ref src/Graph.cpp:1901-1960: 0.1 for pose graphs; computePreconditionerRegularization for range-aided ones
  double preconditionerRegularization(int device = 0) {
    if (type_ == GraphType::PoseGraph) return 0.1;
    const SparseMatrix &Q = quadraticMatrix();
    double reg = 0.1;
    detail::check(dcora_graph_precond_regularization(Q.n, Q.rowptr.data(), Q.colidx.data(), Q.vals.data(), device, &reg),
                  "Graph::preconditionerRegularization");
    return reg;
  }
  // the measurements as a library dataset (the caller destroys it): odometry initialisation, agents' sessions
  dcora_radataset_t ra_dataset(const Matrix *ground_truth = nullptr) const {
    const size_t w = (size_t)d_ * d_ + d_ + 3;
    std::vector<int> ppi(2 * meas_.size() + 2), pli(2 * pose_landmark_.size() + 2), rgi(5 * ranges_.size() + 5);
    std::vector<double> ppv(w * meas_.size() + 1), plv((size_t)(d_ + 2) * pose_landmark_.size() + 1),
        rgv(3 * ranges_.size() + 1);
    for (size_t i = 0; i < meas_.size(); ++i) {
      const RelativePosePoseMeasurement &m = meas_[i];
      ppi[2 * i] = (int)m.p1;
      ppi[2 * i + 1] = (int)m.p2;
      double *v = &ppv[i * w];
      for (unsigned c = 0; c < d_; ++c)
        for (unsigned a = 0; a < d_; ++a) v[c * d_ + a] = m.R(a, c);
      for (unsigned a = 0; a < d_; ++a) v[d_ * d_ + a] = m.t[a];
      v[d_ * d_ + d_] = m.kappa;
      v[d_ * d_ + d_ + 1] = m.tau;
      v[d_ * d_ + d_ + 2] = m.weight;
    }
    for (size_t i = 0; i < pose_landmark_.size(); ++i) {
      const RelativePoseLandmarkMeasurement &m = pose_landmark_[i];
      pli[2 * i] = (int)m.p1;
      pli[2 * i + 1] = (int)m.p2;
      double *v = &plv[i * (d_ + 2)];
      for (unsigned a = 0; a < d_; ++a) v[a] = m.t[a];
      v[d_] = m.tau;
      v[d_ + 1] = m.weight;
    }
    for (size_t i = 0; i < ranges_.size(); ++i) {
      const RangeMeasurement &m = ranges_[i];
      rgi[5 * i] = m.stateType1 == StateType::Landmark ? 1 : 0;
      rgi[5 * i + 1] = (int)m.p1;
      rgi[5 * i + 2] = m.stateType2 == StateType::Landmark ? 1 : 0;
      rgi[5 * i + 3] = (int)m.p2;
      rgi[5 * i + 4] = (int)m.l;
      rgv[3 * i] = m.range;
      rgv[3 * i + 1] = m.precision;
      rgv[3 * i + 2] = m.weight;
    }
    dcora_radataset_t ds = nullptr;
    detail::check(dcora_radataset_create((int)d_, (int)n_, (int)l_, (int)b_, (int)meas_.size(), ppi.data(), ppv.data(),
                                         (int)pose_landmark_.size(), pli.data(), plv.data(), (int)ranges_.size(),
                                         rgi.data(), rgv.data(), ground_truth ? ground_truth->data() : nullptr, &ds),
                  "Graph: range-aided dataset");
    return ds;
  }

 private:
  void count_states() {
    n_ = l_ = b_ = 0;
    auto pose = [&](size_t robot, size_t p) {
      if (robot == id_) n_ = std::max<unsigned>(n_, (unsigned)p + 1);
    };
    auto landmark = [&](size_t robot, size_t p) {
      if (robot == id_) b_ = std::max<unsigned>(b_, (unsigned)p + 1);
    };
    for (const RelativePosePoseMeasurement &m : meas_) {
      pose(m.r1, m.p1);
      pose(m.r2, m.p2);
    }
    for (const RelativePoseLandmarkMeasurement &m : pose_landmark_) {
      pose(m.r1, m.p1);
      landmark(m.r2, m.p2);
    }
    for (const RangeMeasurement &m : ranges_) {
      if (m.stateType1 == StateType::Landmark) landmark(m.r1, m.p1); else pose(m.r1, m.p1);
      if (m.stateType2 == StateType::Landmark) landmark(m.r2, m.p2); else pose(m.r2, m.p2);
      if (m.r1 == id_) l_ = std::max<unsigned>(l_, (unsigned)m.l + 1);  // the unit sphere belongs to the source's robot
    }
    built_ = false;
  }
  unsigned id_, r_, d_, n_ = 0, l_ = 0, b_ = 0;
  GraphType type_;
  std::vector<RelativePosePoseMeasurement> meas_;
  std::vector<RelativePoseLandmarkMeasurement> pose_landmark_;
  std::vector<RangeMeasurement> ranges_;
  SparseMatrix Q_;
  bool built_ = false;
};

}  // namespace DCORA
