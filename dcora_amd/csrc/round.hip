// Rounding / solution recovery on the device: lifted iterate (r x k) -> poses in SE(d).
//   align_trajectory : alignLiftedTrajectoryToFrame (ref src/DCORA_utils.cpp:2262-2289) and the local-frame
//                      variants of Agent (ref src/Agent.cpp:950-1003): T_i = [ Proj_SO(d)(R0^T Y_i) | R0^T p_i - t0 ]
//   project_solution : projectSolutionRASLAM (ref src/DCORA_utils.cpp:1984-2031): rank-d truncation of X through the
//                      r x r Gram matrix (X^T = U S V^T  =>  (U_d S_d)^T = V_d^T X), reflection test, block projection
// One thread per pose / point: this runs once per solve, after the loop.
#include <algorithm>
#include <cmath>
#include <vector>

#include "device_problem.h"

namespace dcora {

namespace {

// Projection of a D x D matrix (column-major) onto SO(D): one-sided Jacobi SVD M V = U S, result U V^T with the
// least singular direction of U flipped when det(U) det(V) < 0  (ref src/DCORA_utils.cpp:1661-1675)
template <int D>
__device__ void so_project(double (&A)[D * D]) {
  double V[D * D];
#pragma unroll
  for (int a = 0; a < D; ++a)
#pragma unroll
    for (int b = 0; b < D; ++b) V[b * D + a] = (a == b) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0;
#pragma unroll
    for (int p = 0; p < D - 1; ++p)
#pragma unroll
      for (int q = p + 1; q < D; ++q) {
        double app = 0, aqq = 0, apq = 0;
#pragma unroll
        for (int i = 0; i < D; ++i) {
          app += A[p * D + i] * A[p * D + i];
          aqq += A[q * D + i] * A[q * D + i];
          apq += A[p * D + i] * A[q * D + i];
        }
        if (fabs(apq) <= 1e-300 || fabs(apq) <= 1e-16 * sqrt(app * aqq)) continue;
        off = fmax(off, fabs(apq) / sqrt(app * aqq));
        const double zeta = (aqq - app) / (2.0 * apq);
        const double tt = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        const double cs = 1.0 / sqrt(1.0 + tt * tt), sn = cs * tt;
#pragma unroll
        for (int i = 0; i < D; ++i) {
          double x = A[p * D + i], y = A[q * D + i];
          A[p * D + i] = cs * x - sn * y;
          A[q * D + i] = sn * x + cs * y;
          x = V[p * D + i];
          y = V[q * D + i];
          V[p * D + i] = cs * x - sn * y;
          V[q * D + i] = sn * x + cs * y;
        }
      }
    if (off < 1e-15) break;
  }
  double sig[D];
  int jmin = 0;
#pragma unroll
  for (int j = 0; j < D; ++j) {
    double nn = 0;
#pragma unroll
    for (int i = 0; i < D; ++i) nn += A[j * D + i] * A[j * D + i];
    sig[j] = sqrt(nn);
  }
#pragma unroll
  for (int j = 1; j < D; ++j)
    if (sig[j] < sig[jmin]) jmin = j;
  // rank-deficient blocks do not occur on iterates of the solver (Y_i has orthonormal columns and R0 is a
  // d-frame); a zero column is left as it is
#pragma unroll
  for (int j = 0; j < D; ++j)
    if (sig[j] > 0) {
      const double inv = 1.0 / sig[j];
#pragma unroll
      for (int i = 0; i < D; ++i) A[j * D + i] *= inv;
    }
  double P[D * D];
#pragma unroll
  for (int c = 0; c < D; ++c)
#pragma unroll
    for (int i = 0; i < D; ++i) {
      double s = 0;
#pragma unroll
      for (int j = 0; j < D; ++j) s += A[j * D + i] * V[j * D + c];
      P[c * D + i] = s;
    }
  double det;
  if (D == 2)
    det = P[0] * P[3] - P[2] * P[1];
  else
    det = P[0] * (P[4] * P[8] - P[7] * P[5]) - P[3] * (P[1] * P[8] - P[7] * P[2]) + P[6] * (P[1] * P[5] - P[4] * P[2]);
  if (det < 0) {
    // flip the least singular direction: P -= 2 u_min v_min^T
#pragma unroll
    for (int c = 0; c < D; ++c)
#pragma unroll
      for (int i = 0; i < D; ++i) {
        double um = 0, vm = 0;
#pragma unroll
        for (int j = 0; j < D; ++j)
          if (j == jmin) {
            um = A[j * D + i];
            vm = V[j * D + c];
          }
        P[c * D + i] -= 2.0 * um * vm;
      }
  }
#pragma unroll
  for (int i = 0; i < D * D; ++i) A[i] = P[i];
}

struct Frame {
  double R0[16 * 3];  // r x d, column-major
  double t0[3];
};

// trajectory in the SE ordering d x (d+1) n from X in the layout described by m
template <int D>
__global__ __launch_bounds__(kBlock) void k_align_poses(ManiDesc m, const double *__restrict__ X, Frame f,
                                                        double *__restrict__ out) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= m.n) return;
  const int r = m.r;
  double M[D * D];
  const double *Y = X + (size_t)m.rot_col(i) * r;
#pragma unroll
  for (int c = 0; c < D; ++c)
#pragma unroll
    for (int a = 0; a < D; ++a) {
      double s = 0;
      for (int q = 0; q < r; ++q) s += f.R0[a * r + q] * Y[(size_t)c * r + q];
      M[c * D + a] = s;
    }
  so_project<D>(M);
  double *o = out + (size_t)i * (D + 1) * D;
#pragma unroll
  for (int e = 0; e < D * D; ++e) o[e] = M[e];
  const double *p = X + (size_t)m.euc_col(i) * r;
#pragma unroll
  for (int a = 0; a < D; ++a) {
    double s = 0;
    for (int q = 0; q < r; ++q) s += f.R0[a * r + q] * p[q];
    o[D * D + a] = s - f.t0[a];
  }
}

// points: out(:, j) = R0^T X(:, col0 + j) - shift t0
__global__ __launch_bounds__(kBlock) void k_align_points(int r, int d, int count, const double *__restrict__ X,
                                                         Frame f, int shift, double *__restrict__ out) {
  const int j = blockIdx.x * kBlock + threadIdx.x;
  if (j >= count) return;
  for (int a = 0; a < d; ++a) {
    double s = 0;
    for (int q = 0; q < r; ++q) s += f.R0[a * r + q] * X[(size_t)j * r + q];
    out[(size_t)j * d + a] = s - (shift ? f.t0[a] : 0.0);
  }
}

// G(a, b) = sum_c X(a, c) X(b, c): one block per entry
__global__ __launch_bounds__(kBlock) void k_gram(int r, int k, const double *__restrict__ X, double *__restrict__ G) {
  __shared__ double sm[kBlock / 64];
  const int a = blockIdx.x / r, b = blockIdx.x - a * r;
  double s = 0;
  for (int c = threadIdx.x; c < k; c += kBlock) s += X[(size_t)c * r + a] * X[(size_t)c * r + b];
  s = wave_sum_dpp(s);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0;
    for (int w = 0; w < kBlock / 64; ++w) t += sm[w];
    G[blockIdx.x] = t;
  }
}

struct Vd {
  double v[16 * 3];  // r x d, column-major: top-d right singular vectors
};
// out(:, c) = Vd^T X(:, c)
__global__ __launch_bounds__(kBlock) void k_truncate(int r, int d, int k, const double *__restrict__ X, Vd V,
                                                     double *__restrict__ out) {
  const int c = blockIdx.x * kBlock + threadIdx.x;
  if (c >= k) return;
  for (int a = 0; a < d; ++a) {
    double s = 0;
    for (int q = 0; q < r; ++q) s += V.v[a * r + q] * X[(size_t)c * r + q];
    out[(size_t)c * d + a] = s;
  }
}
template <int D>
__device__ __forceinline__ double det_of(const double *P) {
  if (D == 2) return P[0] * P[3] - P[2] * P[1];
  return P[0] * (P[4] * P[8] - P[7] * P[5]) - P[3] * (P[1] * P[8] - P[7] * P[2]) + P[6] * (P[1] * P[5] - P[4] * P[2]);
}
// number of rotation blocks of the d x k matrix (layout m with r = d) with positive determinant
template <int D>
__global__ __launch_bounds__(kBlock) void k_count_positive(ManiDesc m, const double *__restrict__ P, int *count) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  int pos = 0;
  if (i < m.n) pos = det_of<D>(P + (size_t)m.rot_col(i) * D) > 0 ? 1 : 0;
  const unsigned long long b = __ballot(pos);
  if ((threadIdx.x & 63) == 0 && b) atomicAdd(count, __popcll(b));
}
// optional reflection of the last row, then SO(d) / unit-sphere projection block by block (in place)
template <int D>
__global__ __launch_bounds__(kBlock) void k_project_blocks(ManiDesc m, int reflect, double *__restrict__ P) {
  const int c = blockIdx.x * kBlock + threadIdx.x;
  if (c >= m.k) return;
  if (reflect) P[(size_t)c * D + (D - 1)] = -P[(size_t)c * D + (D - 1)];
  // sphere columns
  if (!m.se && c >= D * m.n && c < D * m.n + m.l) {
    double s = 0;
#pragma unroll
    for (int a = 0; a < D; ++a) s += P[(size_t)c * D + a] * P[(size_t)c * D + a];
    s = sqrt(s);
    if (s > 0)
#pragma unroll
      for (int a = 0; a < D; ++a) P[(size_t)c * D + a] /= s;
  }
}
template <int D>
__global__ __launch_bounds__(kBlock) void k_project_rotations(ManiDesc m, double *__restrict__ P) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= m.n) return;
  double M[D * D];
  double *blk = P + (size_t)m.rot_col(i) * D;
#pragma unroll
  for (int e = 0; e < D * D; ++e) M[e] = blk[e];
  so_project<D>(M);
#pragma unroll
  for (int e = 0; e < D * D; ++e) blk[e] = M[e];
}

// cyclic Jacobi eigen-decomposition of a small symmetric matrix (host): G = V diag(w) V^T
void jacobi_eig(int n, std::vector<double> &G, std::vector<double> &V) {
  V.assign((size_t)n * n, 0.0);
  for (int i = 0; i < n; ++i) V[(size_t)i * n + i] = 1.0;
  for (int sweep = 0; sweep < 100; ++sweep) {
    double off = 0;
    for (int p = 0; p < n - 1; ++p)
      for (int q = p + 1; q < n; ++q) off = std::max(off, std::fabs(G[(size_t)q * n + p]));
    double diag = 0;
    for (int i = 0; i < n; ++i) diag = std::max(diag, std::fabs(G[(size_t)i * n + i]));
    if (off <= 1e-16 * std::max(diag, 1e-300)) break;
    for (int p = 0; p < n - 1; ++p)
      for (int q = p + 1; q < n; ++q) {
        const double apq = G[(size_t)q * n + p];
        if (std::fabs(apq) <= 1e-300) continue;
        const double app = G[(size_t)p * n + p], aqq = G[(size_t)q * n + q];
        const double zeta = (aqq - app) / (2.0 * apq);
        const double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
        const double c = 1.0 / std::sqrt(1.0 + t * t), s = c * t;
        for (int i = 0; i < n; ++i) {  // columns p, q
          const double x = G[(size_t)p * n + i], y = G[(size_t)q * n + i];
          G[(size_t)p * n + i] = c * x - s * y;
          G[(size_t)q * n + i] = s * x + c * y;
        }
        for (int i = 0; i < n; ++i) {  // rows p, q
          const double x = G[(size_t)i * n + p], y = G[(size_t)i * n + q];
          G[(size_t)i * n + p] = c * x - s * y;
          G[(size_t)i * n + q] = s * x + c * y;
        }
        for (int i = 0; i < n; ++i) {
          const double x = V[(size_t)p * n + i], y = V[(size_t)q * n + i];
          V[(size_t)p * n + i] = c * x - s * y;
          V[(size_t)q * n + i] = s * x + c * y;
        }
      }
  }
}

int check_device(int device) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    set_last_error("no HIP device available: libdcora_hip has no CPU fallback");
    return DCORA_ERR_NO_DEVICE;
  }
  DCORA_HIP(hipSetDevice(device));
  return DCORA_OK;
}

}  // namespace

// X: r x k host, layout dims; anchor: r x (d+1) host or null (=> pose 0 of X, local frame)
int round_align(const dcora_dims &dims, const double *X, const double *anchor, int global, double *traj,
                double *spheres, double *landmarks, int device) {
  int rc = check_device(device);
  if (rc) return rc;
  if (dims.r < dims.d || dims.r > 16 || (dims.d != 2 && dims.d != 3) || dims.n < 1) {
    set_last_error("round_align: need d <= r <= 16, d in {2,3}, n >= 1");
    return DCORA_ERR_BAD_ARG;
  }
  const ManiDesc m = make_mani(dims);
  const int r = m.r, d = m.d;
  Frame f;
  std::vector<double> a((size_t)r * (d + 1));
  if (anchor) {
    std::copy(anchor, anchor + a.size(), a.begin());
  } else {
    for (int c = 0; c < d; ++c)
      for (int q = 0; q < r; ++q) a[(size_t)c * r + q] = X[((size_t)m.rot_col(0) + c) * r + q];
    for (int q = 0; q < r; ++q) a[(size_t)d * r + q] = X[(size_t)m.euc_col(0) * r + q];
  }
  for (int c = 0; c < d; ++c)
    for (int q = 0; q < r; ++q) f.R0[c * r + q] = a[(size_t)c * r + q];
  // global: the anchor's translation is the origin; local: the (rotated) translation of pose 0 of X
  const double *porg = (global && anchor) ? &a[(size_t)d * r] : X + (size_t)m.euc_col(0) * r;
  for (int c = 0; c < d; ++c) {
    double s = 0;
    for (int q = 0; q < r; ++q) s += f.R0[c * r + q] * porg[q];
    f.t0[c] = s;
  }
  DevBuf<double> dX, dT, dS, dL;
  const size_t N = (size_t)r * m.k;
  DCORA_HIP(dX.alloc(N));
  DCORA_HIP(hipMemcpy(dX.p, X, N * sizeof(double), hipMemcpyHostToDevice));
  DCORA_HIP(dT.alloc((size_t)d * (d + 1) * m.n));
  const int grid = (m.n + kBlock - 1) / kBlock;
  if (d == 3)
    hipLaunchKernelGGL(k_align_poses<3>, dim3(grid), dim3(kBlock), 0, nullptr, m, dX.p, f, dT.p);
  else
    hipLaunchKernelGGL(k_align_poses<2>, dim3(grid), dim3(kBlock), 0, nullptr, m, dX.p, f, dT.p);
  if (spheres && m.l > 0) {
    DCORA_HIP(dS.alloc((size_t)d * m.l));
    hipLaunchKernelGGL(k_align_points, dim3((m.l + kBlock - 1) / kBlock), dim3(kBlock), 0, nullptr, r, d, m.l,
                       dX.p + (size_t)m.sphere_col(0) * r, f, 0, dS.p);
  }
  if (landmarks && m.b > 0) {
    DCORA_HIP(dL.alloc((size_t)d * m.b));
    hipLaunchKernelGGL(k_align_points, dim3((m.b + kBlock - 1) / kBlock), dim3(kBlock), 0, nullptr, r, d, m.b,
                       dX.p + (size_t)(d * m.n + m.l + m.n) * r, f, 1, dL.p);
  }
  DCORA_HIP(hipDeviceSynchronize());
  DCORA_HIP(hipMemcpy(traj, dT.p, (size_t)d * (d + 1) * m.n * sizeof(double), hipMemcpyDeviceToHost));
  if (spheres && m.l > 0) DCORA_HIP(hipMemcpy(spheres, dS.p, (size_t)d * m.l * sizeof(double), hipMemcpyDeviceToHost));
  if (landmarks && m.b > 0)
    DCORA_HIP(hipMemcpy(landmarks, dL.p, (size_t)d * m.b * sizeof(double), hipMemcpyDeviceToHost));
  return DCORA_OK;
}

int round_project_solution(const dcora_dims &dims, const double *X, double *out, int device) {
  int rc = check_device(device);
  if (rc) return rc;
  if (dims.r < dims.d || dims.r > 16 || (dims.d != 2 && dims.d != 3) || dims.n < 1) {
    set_last_error("round_project_solution: need d <= r <= 16, d in {2,3}, n >= 1");
    return DCORA_ERR_BAD_ARG;
  }
  const ManiDesc m = make_mani(dims);
  const int r = m.r, d = m.d, k = m.k;
  DevBuf<double> dX, dG, dP;
  DevBuf<int> dcnt;
  DCORA_HIP(dX.alloc((size_t)r * k));
  DCORA_HIP(hipMemcpy(dX.p, X, (size_t)r * k * sizeof(double), hipMemcpyHostToDevice));
  DCORA_HIP(dG.alloc((size_t)r * r));
  hipLaunchKernelGGL(k_gram, dim3(r * r), dim3(kBlock), 0, nullptr, r, k, dX.p, dG.p);
  std::vector<double> G((size_t)r * r), V;
  DCORA_HIP(hipMemcpy(G.data(), dG.p, G.size() * sizeof(double), hipMemcpyDeviceToHost));
  jacobi_eig(r, G, V);
  std::vector<std::pair<double, int>> w((size_t)r);
  for (int j = 0; j < r; ++j) w[j] = {G[(size_t)j * r + j], j};
  std::sort(w.begin(), w.end(), [](const auto &a, const auto &b) { return a.first > b.first; });
  Vd vd;
  for (int a = 0; a < d; ++a)
    for (int q = 0; q < r; ++q) vd.v[a * r + q] = V[(size_t)w[a].second * r + q];
  DCORA_HIP(dP.alloc((size_t)d * k));
  hipLaunchKernelGGL(k_truncate, dim3((k + kBlock - 1) / kBlock), dim3(kBlock), 0, nullptr, r, d, k, dX.p, vd, dP.p);
  // the truncated matrix has the same column layout with r = d
  ManiDesc md = m;
  md.r = d;
  DCORA_HIP(dcnt.alloc(1));
  DCORA_HIP(hipMemset(dcnt.p, 0, sizeof(int)));
  const int gp = (m.n + kBlock - 1) / kBlock;
  if (d == 3)
    hipLaunchKernelGGL(k_count_positive<3>, dim3(gp), dim3(kBlock), 0, nullptr, md, dP.p, dcnt.p);
  else
    hipLaunchKernelGGL(k_count_positive<2>, dim3(gp), dim3(kBlock), 0, nullptr, md, dP.p, dcnt.p);
  int npos = 0;
  DCORA_HIP(hipMemcpy(&npos, dcnt.p, sizeof(int), hipMemcpyDeviceToHost));
  const int reflect = (npos < m.n / 2) ? 1 : 0;
  const int gk = (k + kBlock - 1) / kBlock;
  if (d == 3) {
    hipLaunchKernelGGL(k_project_blocks<3>, dim3(gk), dim3(kBlock), 0, nullptr, md, reflect, dP.p);
    hipLaunchKernelGGL(k_project_rotations<3>, dim3(gp), dim3(kBlock), 0, nullptr, md, dP.p);
  } else {
    hipLaunchKernelGGL(k_project_blocks<2>, dim3(gk), dim3(kBlock), 0, nullptr, md, reflect, dP.p);
    hipLaunchKernelGGL(k_project_rotations<2>, dim3(gp), dim3(kBlock), 0, nullptr, md, dP.p);
  }
  DCORA_HIP(hipDeviceSynchronize());
  DCORA_HIP(hipMemcpy(out, dP.p, (size_t)d * k * sizeof(double), hipMemcpyDeviceToHost));
  return DCORA_OK;
}

}  // namespace dcora
