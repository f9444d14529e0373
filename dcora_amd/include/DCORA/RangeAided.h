// Range-aided SLAM types of the reference that cross the hot-path boundary (ref include/DCORA/Measurements.h:186-520,
// 650-672, 818-870; include/DCORA/manifold/Elements.h:184-470; include/DCORA/DCORA_utils.h:97-116, 517-533): the three
// kinds of relative measurements and their container, the arrays of poses / points a ground truth or an estimate comes
// in, the pyfg dataset, and the frame alignments the reference's tests compare with.  Header-only; the reader and all
// numerics are libdcora_hip's (include/dcora_hip.h, dcora_radataset_*).
//
// Numbering.  The reference's getGlobalMeasurements (src/DCORA_utils.cpp:1169-1365) re-indexes a multi-robot file to
// ONE robot (CENTRALIZED_AGENT_ID) with consecutive poses, landmarks and one unit sphere per range measurement; the
// library's reader does the same, so the measurements below carry r1 = r2 = CENTRALIZED_AGENT_ID and global indices.
#pragma once
#include <cmath>
#include <map>
#include <memory>
#include <set>
#include <string>
#include <vector>

#include "DCORA_types.h"

namespace DCORA {

constexpr unsigned int CENTRALIZED_AGENT_ID = 0;  // ref include/DCORA/DCORA_types.h:42
constexpr unsigned int MAP_ID = 'M' - 'A';        // ref include/DCORA/DCORA_types.h:43-44: the passive map agent
// ref include/DCORA/DCORA_types.h:236-308: states are named (robot, frame); one key type per kind of state
struct LandmarkID : PoseID {
  LandmarkID() = default;
  LandmarkID(unsigned robot, unsigned frame) : PoseID(robot, frame) {}
};
struct UnitSphereID : PoseID {
  UnitSphereID() = default;
  UnitSphereID(unsigned robot, unsigned frame) : PoseID(robot, frame) {}
};
using PoseDict = std::map<PoseID, Matrix>;              // lifted pose, r x (d+1)
using UnitSphereDict = std::map<UnitSphereID, Matrix>;  // lifted point, r x 1
using LandmarkDict = std::map<LandmarkID, Matrix>;      // lifted point, r x 1

enum class StateType { None, Pose, Landmark, UnitSphere };
enum class MeasurementType { PosePrior, LandmarkPrior, PosePose, PoseLandmark, Range };

// ref include/DCORA/Measurements.h:186-270
struct RelativeMeasurement {
  MeasurementType measurementType = MeasurementType::PosePose;
  size_t r1 = 0, r2 = 0, p1 = 0, p2 = 0;
  StateType stateType1 = StateType::Pose, stateType2 = StateType::Pose;
  bool fixedWeight = false;
  double weight = 1.0;
};

// ref include/DCORA/Measurements.h:350-410: pose p1 of robot r1 observes landmark p2 of robot r2 at t (in the pose's frame)
struct RelativePoseLandmarkMeasurement : RelativeMeasurement {
  Vector t;
  double tau = 0;
  RelativePoseLandmarkMeasurement() {
    measurementType = MeasurementType::PoseLandmark;
    stateType2 = StateType::Landmark;
  }
  RelativePoseLandmarkMeasurement(size_t firstRobot, size_t secondRobot, size_t firstPose, size_t secondLandmark,
                                  const Vector &relativeTranslation, double translationalPrecision,
                                  bool fixedWeightIn = false, double weightIn = 1.0)
      : RelativePoseLandmarkMeasurement() {
    r1 = firstRobot;
    r2 = secondRobot;
    p1 = firstPose;
    p2 = secondLandmark;
    t = relativeTranslation;
    tau = translationalPrecision;
    fixedWeight = fixedWeightIn;
    weight = weightIn;
  }
};

// ref include/DCORA/Measurements.h:416-496: a range between two states (poses or landmarks) with its unit-sphere variable l
struct RangeMeasurement : RelativeMeasurement {
  size_t l = 0;
  double range = 0, precision = 0;
  RangeMeasurement() {
    measurementType = MeasurementType::Range;
    stateType1 = stateType2 = StateType::None;
  }
  RangeMeasurement(size_t firstRobot, size_t secondRobot, size_t firstState, size_t secondState, size_t unitSphereVarIdx,
                   double rangeMeasurement, double rangePrecision, StateType stateType1In, StateType stateType2In,
                   bool fixedWeightIn = false, double weightIn = 1.0)
      : RangeMeasurement() {
    r1 = firstRobot;
    r2 = secondRobot;
    p1 = firstState;
    p2 = secondState;
    l = unitSphereVarIdx;
    range = rangeMeasurement;
    precision = rangePrecision;
    stateType1 = stateType1In;
    stateType2 = stateType2In;
    fixedWeight = fixedWeightIn;
    weight = weightIn;
  }
};

// ref include/DCORA/Measurements.h:505-640 (the reference keeps one vector of variants; the getters are its interface)
class RelativeMeasurements {
 public:
  void push_back(const RelativePosePoseMeasurement &m) { pose_pose_.push_back(m); }
  void push_back(const RelativePoseLandmarkMeasurement &m) { pose_landmark_.push_back(m); }
  void push_back(const RangeMeasurement &m) { ranges_.push_back(m); }
  std::vector<RelativePosePoseMeasurement> GetRelativePosePoseMeasurements() const { return pose_pose_; }
  std::vector<RelativePoseLandmarkMeasurement> GetRelativePoseLandmarkMeasurements() const { return pose_landmark_; }
  std::vector<RangeMeasurement> GetRangeMeasurements() const { return ranges_; }
  size_t size() const { return pose_pose_.size() + pose_landmark_.size() + ranges_.size(); }

 private:
  std::vector<RelativePosePoseMeasurement> pose_pose_;
  std::vector<RelativePoseLandmarkMeasurement> pose_landmark_;
  std::vector<RangeMeasurement> ranges_;
};

// ---- arrays (ref include/DCORA/manifold/Elements.h:115-470), d x ... column-major as the reference's getData() ----
class Pose {  // [R t], d x (d+1)
 public:
  Pose() = default;
  explicit Pose(const Matrix &T) : T_(T) {}
  unsigned d() const { return (unsigned)T_.rows(); }
  const Matrix &getData() const { return T_; }

 private:
  Matrix T_;
};
class PoseArray {  // n poses [R_i t_i] side by side: d x (d+1) n
 public:
  PoseArray(unsigned d, unsigned n) : d_(d), n_(n), X_(d, (size_t)(d + 1) * n) {}
  unsigned d() const { return d_; }
  unsigned n() const { return n_; }
  Matrix pose(unsigned i) const {
    Matrix T(d_, d_ + 1);
    for (unsigned c = 0; c <= d_; ++c)
      for (unsigned a = 0; a < d_; ++a) T(a, c) = X_(a, (size_t)i * (d_ + 1) + c);
    return T;
  }
  const Matrix &getData() const { return X_; }
  Matrix &data() { return X_; }

 private:
  unsigned d_, n_;
  Matrix X_;
};
class PointArray {  // n points: d x n
 public:
  PointArray(unsigned d, unsigned n) : d_(d), n_(n), X_(d, n) {}
  unsigned d() const { return d_; }
  unsigned n() const { return n_; }
  const Matrix &getData() const { return X_; }
  Matrix &data() { return X_; }

 private:
  unsigned d_, n_;
  Matrix X_;
};
// all states of a range-aided problem in the RA ordering [R_1 .. R_n | s_1 .. s_l | t_1 .. t_n | L_1 .. L_b]: d x k
class RangeAidedArray {
 public:
  RangeAidedArray(unsigned d, unsigned n, unsigned l, unsigned b)
      : d_(d), n_(n), l_(l), b_(b), X_(d, (size_t)(d + 1) * n + l + b) {}
  unsigned d() const { return d_; }
  unsigned n() const { return n_; }
  unsigned l() const { return l_; }
  unsigned b() const { return b_; }
  const Matrix &getData() const { return X_; }
  Matrix &data() { return X_; }
  PoseArray getPoseArray() const {
    PoseArray T(d_, n_);
    for (unsigned i = 0; i < n_; ++i)
      for (unsigned a = 0; a < d_; ++a) {
        for (unsigned c = 0; c < d_; ++c) T.data()(a, (size_t)i * (d_ + 1) + c) = X_(a, (size_t)i * d_ + c);
        T.data()(a, (size_t)i * (d_ + 1) + d_) = X_(a, (size_t)d_ * n_ + l_ + i);
      }
    return T;
  }
  PointArray getUnitSphereArray() const {
    PointArray S(d_, l_);
    for (unsigned i = 0; i < l_; ++i)
      for (unsigned a = 0; a < d_; ++a) S.data()(a, i) = X_(a, (size_t)d_ * n_ + i);
    return S;
  }
  PointArray getLandmarkArray() const {
    PointArray L(d_, b_);
    for (unsigned i = 0; i < b_; ++i)
      for (unsigned a = 0; a < d_; ++a) L.data()(a, i) = X_(a, (size_t)(d_ + 1) * n_ + l_ + i);
    return L;
  }
  // single states (ref include/DCORA/manifold/Elements.h:330-470): pose i as [R_i t_i], points as columns
  unsigned rows() const { return (unsigned)X_.rows(); }
  Matrix pose(unsigned i) const {
    Matrix T(rows(), d_ + 1);
    for (unsigned a = 0; a < rows(); ++a) {
      for (unsigned c = 0; c < d_; ++c) T(a, c) = X_(a, (size_t)i * d_ + c);
      T(a, d_) = X_(a, (size_t)d_ * n_ + l_ + i);
    }
    return T;
  }
  Matrix unitSphere(unsigned i) const { return column((size_t)d_ * n_ + i); }
  Matrix landmark(unsigned i) const { return column((size_t)(d_ + 1) * n_ + l_ + i); }
  void setData(const Matrix &X) {
    if (X.rows() != X_.rows() || X.cols() != X_.cols()) throw std::invalid_argument("RangeAidedArray::setData: shape");
    X_ = X;
  }

 protected:
  RangeAidedArray(unsigned rows, unsigned d, unsigned n, unsigned l, unsigned b)
      : d_(d), n_(n), l_(l), b_(b), X_(rows, (size_t)(d + 1) * n + l + b) {}
  Matrix column(size_t j) const {
    Matrix v(rows(), 1);
    for (unsigned a = 0; a < rows(); ++a) v(a, 0) = X_(a, j);
    return v;
  }

 public:

 private:
  unsigned d_, n_, l_, b_;
  Matrix X_;
};

// the lifted counterpart (ref include/DCORA/manifold/Elements.h LiftedRangeAidedArray): r x k in the RA ordering
class LiftedRangeAidedArray : public RangeAidedArray {
 public:
  LiftedRangeAidedArray(unsigned r, unsigned d, unsigned n, unsigned l, unsigned b) : RangeAidedArray(r, d, n, l, b) {}
};

// ref include/DCORA/Measurements.h:650-672, 818-870 (the members the drivers and tests read)
struct Measurements {
  RelativeMeasurements relative_measurements;
  std::shared_ptr<RangeAidedArray> ground_truth_init;
};
using RobotMeasurements = std::map<unsigned int, Measurements>;  // ref include/DCORA/Measurements.h:687
struct LocalToGlobalStateDicts {                                  // ref include/DCORA/Measurements.h:700-706
  std::map<PoseID, PoseID> poses;
  std::map<LandmarkID, LandmarkID> landmarks;
  std::map<UnitSphereID, UnitSphereID> unit_spheres;
};
struct PyFGDataset {
  unsigned int dim = 0;
  std::set<unsigned int> robot_IDs;
  Measurements measurements;  // in the numbering of the merged (centralised) problem
  // owner robot of every pose / unit sphere / landmark of the merged problem ('A' = 0, ..., MAP_ID the map), as the
  // reference splits a multi-robot file (ref src/DCORA_utils.cpp:1370-1512, src/Graph.cpp:584-616, 1092-1097)
  std::vector<unsigned> pose_robot, sphere_robot, landmark_robot;
};

// ref src/DCORA_utils.cpp:437-1167: the file through the library's reader (dcora_radataset_load_pyfg)
inline PyFGDataset read_pyfg_file(const std::string &filename) {
  dcora_radataset_t h = nullptr;
  check_status(dcora_radataset_load_pyfg(filename.c_str(), &h), "read_pyfg_file");
  int info[7];
  dcora_radataset_info(h, info);
  const int d = info[0], n = info[1], l = info[2], b = info[3], mpp = info[4], mpl = info[5], mrg = info[6];
  const size_t w = (size_t)d * d + d + 3;
  std::vector<int> ppi((size_t)2 * mpp + 2), pli((size_t)2 * mpl + 2), rgi((size_t)5 * mrg + 5);
  std::vector<double> ppv(w * mpp + 1), plv((size_t)(d + 2) * mpl + 1), rgv((size_t)3 * mrg + 1);
  check_status(dcora_radataset_copy(h, ppi.data(), ppv.data(), pli.data(), plv.data(), rgi.data(), rgv.data()),
               "read_pyfg_file");
  PyFGDataset out;
  out.dim = (unsigned)d;
  RelativeMeasurements &rm = out.measurements.relative_measurements;
  const size_t id = CENTRALIZED_AGENT_ID;
  for (int i = 0; i < mpp; ++i) {
    const double *v = &ppv[(size_t)i * w];
    Matrix R((size_t)d, (size_t)d);
    for (int c = 0; c < d; ++c)
      for (int a = 0; a < d; ++a) R((size_t)a, (size_t)c) = v[c * d + a];
    rm.push_back(RelativePosePoseMeasurement(id, id, (size_t)ppi[2 * (size_t)i], (size_t)ppi[2 * (size_t)i + 1], R,
                                             Vector(v + d * d, v + d * d + d), v[d * d + d], v[d * d + d + 1], false,
                                             v[d * d + d + 2]));
  }
  for (int i = 0; i < mpl; ++i) {
    const double *v = &plv[(size_t)i * (d + 2)];
    rm.push_back(RelativePoseLandmarkMeasurement(id, id, (size_t)pli[2 * (size_t)i], (size_t)pli[2 * (size_t)i + 1],
                                                 Vector(v, v + d), v[d], false, v[d + 1]));
  }
  for (int i = 0; i < mrg; ++i) {
    const int *q = &rgi[(size_t)5 * i];
    const double *v = &rgv[(size_t)3 * i];
    rm.push_back(RangeMeasurement(id, id, (size_t)q[1], (size_t)q[3], (size_t)q[4], v[0], v[1],
                                  q[0] == 0 ? StateType::Pose : StateType::Landmark,
                                  q[2] == 0 ? StateType::Pose : StateType::Landmark, false, v[2]));
  }
  auto gt = std::make_shared<RangeAidedArray>((unsigned)d, (unsigned)n, (unsigned)l, (unsigned)b);
  check_status(dcora_radataset_ground_truth(h, gt->data().data()), "read_pyfg_file");
  out.measurements.ground_truth_init = gt;
  std::vector<int> pr((size_t)n + 1), sr((size_t)l + 1), lr((size_t)b + 1);
  dcora_radataset_ownership(h, pr.data(), sr.data(), lr.data());
  for (int i = 0; i < n; ++i) out.robot_IDs.insert((unsigned)pr[(size_t)i]);
  out.pose_robot.assign(pr.begin(), pr.begin() + n);
  out.sphere_robot.assign(sr.begin(), sr.begin() + l);
  out.landmark_robot.assign(lr.begin(), lr.begin() + b);
  for (int j = 0; j < b; ++j)
    if ((unsigned)lr[(size_t)j] == MAP_ID) out.robot_IDs.insert(MAP_ID);  // landmarks without a robot letter: the map's
  dcora_radataset_destroy(h);
  return out;
}
namespace detail {
// position of state i among the states of its owner (the reference re-indexes every robot's states from zero)
inline std::vector<unsigned> local_index(const std::vector<unsigned> &owner) {
  std::map<unsigned, unsigned> next;
  std::vector<unsigned> loc(owner.size());
  for (size_t i = 0; i < owner.size(); ++i) loc[i] = next[owner[i]]++;
  return loc;
}
}  // namespace detail
// ref src/DCORA_utils.cpp:1370-1512: every robot's share of a multi-robot file -- the measurements that touch one of
// its states, with (robot, local index) names, and its ground truth in ITS RA ordering
inline RobotMeasurements getRobotMeasurements(const PyFGDataset &ds) {
  const unsigned d = ds.dim;
  const RangeAidedArray &gt = *ds.measurements.ground_truth_init;
  const unsigned n = gt.n(), l = gt.l(), b = gt.b();
  const std::vector<unsigned> lp = detail::local_index(ds.pose_robot), ls = detail::local_index(ds.sphere_robot),
                              ll = detail::local_index(ds.landmark_robot);
  auto owner = [&](StateType t, size_t i) { return t == StateType::Landmark ? ds.landmark_robot[i] : ds.pose_robot[i]; };
  auto local = [&](StateType t, size_t i) { return t == StateType::Landmark ? ll[i] : lp[i]; };
  RobotMeasurements out;
  for (unsigned robot : ds.robot_IDs) {
    Measurements &M = out[robot];
    const RelativeMeasurements &rm = ds.measurements.relative_measurements;
    for (RelativePosePoseMeasurement m : rm.GetRelativePosePoseMeasurements()) {
      const unsigned o1 = ds.pose_robot[m.p1], o2 = ds.pose_robot[m.p2];
      if (o1 != robot && o2 != robot) continue;
      m.r1 = o1, m.r2 = o2, m.p1 = lp[m.p1], m.p2 = lp[m.p2];
      M.relative_measurements.push_back(m);
    }
    for (RelativePoseLandmarkMeasurement m : rm.GetRelativePoseLandmarkMeasurements()) {
      const unsigned o1 = ds.pose_robot[m.p1], o2 = ds.landmark_robot[m.p2];
      if (o1 != robot && o2 != robot) continue;
      m.r1 = o1, m.r2 = o2, m.p1 = lp[m.p1], m.p2 = ll[m.p2];
      M.relative_measurements.push_back(m);
    }
    for (RangeMeasurement m : rm.GetRangeMeasurements()) {
      const unsigned o1 = owner(m.stateType1, m.p1), o2 = owner(m.stateType2, m.p2);
      if (o1 != robot && o2 != robot) continue;
      m.r1 = o1, m.r2 = o2, m.p1 = local(m.stateType1, m.p1), m.p2 = local(m.stateType2, m.p2);
      m.l = ls[m.l];  // the unit sphere belongs to the source's robot
      M.relative_measurements.push_back(m);
    }
    unsigned na = 0, la = 0, ba = 0;
    for (unsigned o : ds.pose_robot) na += o == robot;
    for (unsigned o : ds.sphere_robot) la += o == robot;
    for (unsigned o : ds.landmark_robot) ba += o == robot;
    auto g = std::make_shared<RangeAidedArray>(d, na, la, ba);
    const Matrix &G = gt.getData();
    for (unsigned a = 0; a < d; ++a) {
      for (unsigned i = 0; i < n; ++i)
        if (ds.pose_robot[i] == robot) {
          for (unsigned c = 0; c < d; ++c) g->data()(a, (size_t)lp[i] * d + c) = G(a, (size_t)i * d + c);
          g->data()(a, (size_t)d * na + la + lp[i]) = G(a, (size_t)d * n + l + i);
        }
      for (unsigned s = 0; s < l; ++s)
        if (ds.sphere_robot[s] == robot) g->data()(a, (size_t)d * na + ls[s]) = G(a, (size_t)d * n + s);
      for (unsigned j = 0; j < b; ++j)
        if (ds.landmark_robot[j] == robot)
          g->data()(a, (size_t)(d + 1) * na + la + ll[j]) = G(a, (size_t)(d + 1) * n + l + j);
    }
    M.ground_truth_init = g;
  }
  return out;
}
// ref include/DCORA/DCORA_utils.h getLocalToGlobalStateMapping: (robot, local index) -> (CENTRALIZED_AGENT_ID, index in
// the merged problem), the numbering getGlobalMeasurements uses
inline LocalToGlobalStateDicts getLocalToGlobalStateMapping(const PyFGDataset &ds, bool /*reindex*/ = true) {
  const std::vector<unsigned> lp = detail::local_index(ds.pose_robot), ls = detail::local_index(ds.sphere_robot),
                              ll = detail::local_index(ds.landmark_robot);
  LocalToGlobalStateDicts out;
  for (size_t i = 0; i < lp.size(); ++i) out.poses[PoseID(ds.pose_robot[i], lp[i])] = PoseID(CENTRALIZED_AGENT_ID, (unsigned)i);
  for (size_t i = 0; i < ls.size(); ++i)
    out.unit_spheres[UnitSphereID(ds.sphere_robot[i], ls[i])] = UnitSphereID(CENTRALIZED_AGENT_ID, (unsigned)i);
  for (size_t i = 0; i < ll.size(); ++i)
    out.landmarks[LandmarkID(ds.landmark_robot[i], ll[i])] = LandmarkID(CENTRALIZED_AGENT_ID, (unsigned)i);
  return out;
}
// ref src/DCORA_utils.cpp:1169-1365: the reader above already numbers the file as one robot's problem
inline Measurements getGlobalMeasurements(const PyFGDataset &pyfg_dataset) { return pyfg_dataset.measurements; }

// ---- frame alignment of an estimate (ref src/DCORA_utils.cpp:2222-2260): into the frame of the pose Tw0 ----
inline PoseArray alignTrajectoryToFrame(PoseArray T, const Pose &Tw0) {
  const unsigned d = T.d();
  const Matrix &A = Tw0.getData();
  PoseArray out(d, T.n());
  for (unsigned i = 0; i < T.n(); ++i)
    for (unsigned c = 0; c <= d; ++c)
      for (unsigned a = 0; a < d; ++a) {
        double s = 0;  // R0^T (column c of [R_i  t_i - t0])
        for (unsigned q = 0; q < d; ++q)
          s += A(q, a) * (T.getData()(q, (size_t)i * (d + 1) + c) - (c == d ? A(q, d) : 0.0));
        out.data()(a, (size_t)i * (d + 1) + c) = s;
      }
  return out;
}
inline PointArray alignUnitSpheresToFrame(PointArray S, const Pose &Tw0) {  // directions rotate only
  const unsigned d = S.d();
  const Matrix &A = Tw0.getData();
  PointArray out(d, S.n());
  for (unsigned i = 0; i < S.n(); ++i)
    for (unsigned a = 0; a < d; ++a) {
      double s = 0;
      for (unsigned q = 0; q < d; ++q) s += A(q, a) * S.getData()(q, i);
      out.data()(a, i) = s;
    }
  return out;
}
inline PointArray alignLandmarksToFrame(PointArray L, const Pose &Tw0) {
  const unsigned d = L.d();
  const Matrix &A = Tw0.getData();
  PointArray out(d, L.n());
  for (unsigned i = 0; i < L.n(); ++i)
    for (unsigned a = 0; a < d; ++a) {
      double s = 0;
      for (unsigned q = 0; q < d; ++q) s += A(q, a) * (L.getData()(q, i) - A(q, d));
      out.data()(a, i) = s;
    }
  return out;
}

}  // namespace DCORA
