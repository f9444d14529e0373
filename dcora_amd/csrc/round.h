// Rounding / solution recovery (round.hip)
#pragma once
#include "../../include/dcora_hip.h"

namespace dcora {
int round_align(const dcora_dims &dims, const double *X, const double *anchor, int global, double *traj,
                double *spheres, double *landmarks, int device);
int round_project_solution(const dcora_dims &dims, const double *X, double *out, int device);
}  // namespace dcora
