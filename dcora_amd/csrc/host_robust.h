// Robust estimation around the hot path (host side): RobustCost (M-estimators and GNC-TLS), chi-square quantile,
// single rotation / pose averaging with GNC (ref include/DCORA/DCORA_robust.h, src/DCORA_robust.cpp,
// src/DCORA_solver.cpp:28-216).  Scalar / d x d arithmetic: host code in the reference, host code here.
#pragma once
#include <vector>

#include "../../include/dcora_hip.h"

namespace dcora {

class RobustCost {
 public:
  explicit RobustCost(const dcora_robust_params &p) : p_(p), mu_(p.GNCInitMu) { reset(); }
  double weight(double r) const;  // ref src/DCORA_robust.cpp:56-100
  void reset();                   // :102-114
  void update();                  // :116-136
  double mu() const { return mu_; }

 private:
  dcora_robust_params p_;
  double mu_;
  int iteration_ = 0;
};

double chi2inv(double quantile, int dof);                                        // ref src/DCORA_utils.cpp:2103-2106
bool error_threshold_at_quantile(double quantile, int dimension, double *out);   // ref src/DCORA_robust.cpp:138-148
void project_to_rotation_group_host(int d, const double *M, double *out);        // host_init.cpp

// R: n rotations d x d column-major; t: n translations; kappa / tau may be null (defaults of the reference)
void robust_single_rotation_averaging(int d, int n, const double *R, const double *kappa, double threshold,
                                      double *Ropt, std::vector<int> &inliers);
void robust_single_pose_averaging(int d, int n, const double *R, const double *t, const double *kappa,
                                  const double *tau, double threshold, double *Ropt, double *topt,
                                  std::vector<int> &inliers);

// ---- cross-robot frame alignment (host logic around the solver, ref src/Agent.cpp:460-520, 694-833) ----
// poses are d x (d+1) column-major [R t]
// Agent::computeNeighborTransform: the transform world2 <- world1 implied by one inter-robot loop closure, from the
// neighbour's pose of that closure in world2 and my pose of it in world1; incoming: I am the measurement's p2
void neighbor_transform(int d, bool incoming, const double *Rm, const double *tm, const double *T_w2_f2,
                        const double *T_w1_f1, double *T_out);
// computeRobustNeighborTransform (two_stage = false: GNC pose averaging, kappa 1.82, tau 0.01, 90 % quantile) and
// computeRobustNeighborTransformTwoStage (GNC rotation averaging at ~30 deg, then the inliers' mean translation);
// false when fewer than min_inliers candidates agree
bool robust_neighbor_transform(int d, int m, const double *cand, bool two_stage, int min_inliers, double *T,
                               int *num_inliers);
// fixedStiefelVariable (ref src/DCORA_utils.cpp:2053-2056): the same r x d orthonormal frame on every call and every
// process (the lifting matrix all agents share); seeded splitmix64 entries, modified Gram-Schmidt
void fixed_stiefel_variable(int r, int d, double *Y);
// Agent::initializeInGlobalFrame: X = YLift * (T_world_robot applied to the local estimate); Tlocal d x k in this
// ABI's ordering (SE when l = b = 0, RA otherwise), YLift r x d, X r x k
void initialize_in_global_frame(int r, int d, int n, int l, int b, bool se, const double *T_world_robot, const double *Tlocal,
                                const double *YLift, double *X);

}  // namespace dcora
