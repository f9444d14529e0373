// Host-side mirror of the reference types that cross the hot-path boundary
// (ref include/DCORA/DCORA_types.h:34-37, 152-233).  Header-only, no third-party dependency: when Eigen is
// available the reference's own `Matrix = Eigen::MatrixXd` is layout-identical (column-major, tight) to the
// raw buffers used here, so DCORA::Matrix below can be swapped for it without touching the C ABI.
#pragma once
#include <cstddef>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../../include/dcora_hip.h"

namespace DCORA {

// column-major dense matrix (ref DCORA_types.h:34)
class Matrix {
 public:
  Matrix() = default;
  Matrix(size_t rows, size_t cols) : r_(rows), c_(cols), a_(rows * cols, 0.0) {}
  static Matrix Zero(size_t rows, size_t cols) { return Matrix(rows, cols); }
  size_t rows() const { return r_; }
  size_t cols() const { return c_; }
  double &operator()(size_t i, size_t j) { return a_[j * r_ + i]; }
  double operator()(size_t i, size_t j) const { return a_[j * r_ + i]; }
  double *data() { return a_.data(); }
  const double *data() const { return a_.data(); }
  double norm() const;

 private:
  size_t r_ = 0, c_ = 0;
  std::vector<double> a_;
};
inline double Matrix::norm() const {
  double s = 0;
  for (double x : a_) s += x * x;
  return __builtin_sqrt(s);
}
using Vector = std::vector<double>;

// row-major CSR, both triangles (ref DCORA_types.h:36)
struct SparseMatrix {
  int n = 0;
  std::vector<int> rowptr, colidx;
  std::vector<double> vals;
};

// ref DCORA_types.h:152-200
class ROptParameters {
 public:
  enum class ROptMethod { RTR, RGD };
  ROptMethod method = ROptMethod::RTR;
  bool verbose = false;
  double gradnorm_tol = 1e-2;
  double RGD_stepsize = 1e-3;
  bool RGD_use_preconditioner = true;
  int RTR_iterations = 3;
  int RTR_tCG_iterations = 50;
  double RTR_initial_radius = 100;
  dcora_ropt_params c() const {
    dcora_ropt_params p;
    p.method = method == ROptMethod::RTR ? 0 : 1;
    p.verbose = verbose;
    p.gradnorm_tol = gradnorm_tol;
    p.RGD_stepsize = RGD_stepsize;
    p.RGD_use_preconditioner = RGD_use_preconditioner;
    p.RTR_iterations = RTR_iterations;
    p.RTR_tCG_iterations = RTR_tCG_iterations;
    p.RTR_initial_radius = RTR_initial_radius;
    return p;
  }
};

// ref DCORA_types.h:203-233
struct ROPTResult {
  bool success = false;
  double fInit = 0, gradNormInit = 0, fOpt = 0, gradNormOpt = 0, elapsedMs = 0;
  int tCGStatus = 4;
};

// ref include/DCORA/DCORA_types.h (PoseID): a pose is named by (robot, frame)
struct PoseID {
  unsigned robot_id = 0, frame_id = 0;
  PoseID() = default;
  PoseID(unsigned robot, unsigned frame) : robot_id(robot), frame_id(frame) {}
  bool operator<(const PoseID &o) const {
    return robot_id != o.robot_id ? robot_id < o.robot_id : frame_id < o.frame_id;
  }
  bool operator==(const PoseID &o) const { return robot_id == o.robot_id && frame_id == o.frame_id; }
};

// ref include/DCORA/Measurements.h:275-335 (the fields and the basic constructor the drivers use)
struct RelativePosePoseMeasurement {
  size_t r1 = 0, r2 = 0, p1 = 0, p2 = 0;
  Matrix R;  // d x d
  Vector t;  // d
  double kappa = 0, tau = 0;
  bool fixedWeight = false;
  double weight = 1.0;
  RelativePosePoseMeasurement() = default;
  RelativePosePoseMeasurement(size_t firstRobot, size_t secondRobot, size_t firstPose, size_t secondPose,
                              const Matrix &relativeRotation, const Vector &relativeTranslation,
                              double rotationalPrecision, double translationalPrecision, bool fixedWeightIn = false,
                              double weightIn = 1.0)
      : r1(firstRobot), r2(secondRobot), p1(firstPose), p2(secondPose), R(relativeRotation), t(relativeTranslation),
        kappa(rotationalPrecision), tau(translationalPrecision), fixedWeight(fixedWeightIn), weight(weightIn) {}
};

// glog CHECK stand-in: invariants abort in the reference (ref src/QuadraticProblem.cpp:39-40); here they throw on
// the host side of the ABI (never across it)
inline void check_status(int st, const char *what) {
  if (st != DCORA_OK) throw std::runtime_error(std::string(what) + ": " + dcora_last_error());
}

}  // namespace DCORA
