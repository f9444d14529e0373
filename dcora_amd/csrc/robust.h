// Robust pose-graph optimisation drivers (robust.hip)
#pragma once
#include "../../include/dcora_hip.h"
#include "host_graph.h"

namespace dcora {
int measurement_errors(const HostDataset &ds, int r, const double *X, double *out, int device);
int solve_pgo(const HostDataset &ds, const dcora_ropt_params &prm, const double *T0, double *Tout, int device,
              dcora_ropt_result *res);
int solve_robust_pgo(HostDataset &ds, const dcora_ropt_params &prm, const dcora_robust_params &rp, const int *fixed,
                     const double *T0, double *Tout, double *weights_out, int device);
}  // namespace dcora
