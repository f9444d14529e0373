// Host-side data feed of the product: measurement list, g2o reader, and direct (closed-form, per-edge block)
// assembly of the PGO data matrices that the GPU path consumes:
//   Q   (ref src/Graph.cpp:579-683)     -- connection Laplacian of one agent / of the whole graph
//   C   (ref src/Graph.cpp:685-822)     -- the coupling blocks: G_b = X_global * C_b, i.e. the linear term of
//                                          agent b as a sparse product with the neighbours' public poses
#pragma once
#include <functional>
#include <string>
#include <vector>

#include "host_sparse.h"

namespace dcora {

// ref include/DCORA/Measurements.h RelativePosePoseMeasurement
struct PoseMeas {
  int r1 = 0, p1 = 0, r2 = 0, p2 = 0;
  double R[9] = {0}, t[3] = {0};
  double kappa = 0, tau = 0, weight = 1;
};
struct HostDataset {
  int d = 0, n = 0;
  std::vector<PoseMeas> meas;
};

bool load_g2o(const std::string &path, HostDataset &out, std::string &err);

// Q of agent `id` over its n local poses from every measurement that touches it
HostCsr build_Q_pgo(int d, int n, int id, const std::vector<PoseMeas> &meas);

// contiguous partition of the reference driver (ref examples/MultiRobotExample.cpp:56-83)
struct Partition {
  int R = 1, n = 0, per = 0;
  int robot_of(int idx) const { return std::min(idx / per, R - 1); }
  int start(int rb) const { return rb * per; }
  int end(int rb) const { return rb == R - 1 ? n : (rb + 1) * per; }
};

// Coupling matrix of agent b with GLOBAL column indices: rows = (d+1) n_b local columns of G_b, columns =
// (d+1) n global columns of X.  G_b = X * C_b^T in the SpMM convention Y(:, j) = sum_c A(j, c) X(:, c).
HostCsr build_coupling_pgo(int d, const Partition &P, int b, const std::vector<PoseMeas> &global_meas);

}  // namespace dcora

namespace dcora {
// ---- range-aided SLAM, centralised agent (ref src/Graph.cpp:824-1188; src/DCORA_utils.cpp:437-1167, 1169-1365) ----
struct PoseLandmarkMeasH {
  int i = 0, j = 0;
  double t[3] = {0}, tau = 0, weight = 1;
};
struct RangeMeasH {
  int type1 = 0, i = 0, type2 = 0, j = 0, l = 0;  // type: 0 pose, 1 landmark
  double range = 0, precision = 0, weight = 1;
};
struct HostRADataset {
  int d = 0, n = 0, l = 0, b = 0;
  std::vector<PoseMeas> pose_pose;  // global pose indices in p1 / p2
  std::vector<PoseLandmarkMeasH> pose_landmark;
  std::vector<RangeMeasH> ranges;
  std::vector<double> gt;  // d x k ground truth, RA ordering, column-major
  // owner robot ('A' = 0, ..., 'M' = 12 the map) of every pose, unit sphere and landmark of the merged problem
  std::vector<int> pose_robot, sphere_robot, landmark_robot;
  int k() const { return (d + 1) * n + l + b; }
};
// Columns (global RA ordering) owned by `robot`, listed in that agent's own RA ordering
// [rotations of its poses | its unit spheres | translations of its poses | its landmarks]; dims3 = {n_a, l_a, b_a}
void ra_agent_columns(const HostRADataset &ds, int robot, int dims3[3], std::vector<int> &own);
// The agent's share of a global quadratic form: Qaa = Q[own, own] in the agent's ordering and the coupling
// C = Q[own, everything else] with GLOBAL column indices, so that the agent's linear term is G_a = X_global C^T
// (the same restriction the reference assembles edge by edge, ref src/Graph.cpp:824-1772)
void extract_agent_blocks(const HostCsr &Q, const std::vector<int> &own, HostCsr *Qaa, HostCsr *C);
// chordalInitialization (ref src/DCORA_solver.cpp:218-268): T is d x (d+1) n column-major, pose 0 = identity
// solve(A, block, nrhs, B, X): X = A^-1 B for a sparse SPD A, right-hand sides contiguous per unknown (B[i * nrhs + t]);
// false when A is not positive definite.  nullptr: sparse Cholesky on the host.  The device variant is
// device_spd_solver(device) of device_chol.h (the host factorisation of the 100k lattice's systems takes minutes).
using SpdSolve = std::function<bool(const HostCsr &, int, int, const double *, double *)>;
bool chordal_initialization(const HostDataset &ds, std::vector<double> &T, const SpdSolve &solve = nullptr);
bool load_pyfg(const std::string &path, HostRADataset &out, std::string &err);
// start point of the centralised CORA driver (ref examples/SingleRobotExample_RASLAM.cpp:92-150): odometry chains
// anchored at their ground-truth first pose, ground-truth unit spheres, seeded uniform(-1, 1) landmarks; d x k
void ra_odometry_initialization(const HostRADataset &ds, unsigned long long seed, std::vector<double> &X0);
// Q in the RA ordering [Y1..Yn | s1..sl | p1..pn | L1..Lb], assembled from the closed-form blocks of each factor
HostCsr build_Q_ra(const HostRADataset &ds);
}  // namespace dcora
