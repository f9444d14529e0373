// scratch: v_mfma_f64_4x4x4_4b_f64 on this GPU: (1) which lane holds which element of A, B and D, found by unit
// vectors; (2) its issue rate next to v_mfma_f64_16x16x4_f64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v4f64 __attribute__((ext_vector_type(4)));
// trial t = la * 64 + lb: A = e_la, B = e_lb; out[t][lane] = D
__global__ __launch_bounds__(64) void k_probe(double *out) {
  const int t = blockIdx.x, la = t >> 6, lb = t & 63, lane = threadIdx.x;
  const double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
  const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
  out[(size_t)t * 64 + lane] = d;
}
template <int NACC>
__global__ __launch_bounds__(256) void k_rate4(int iters, double *out) {
  double c[NACC];
  for (int i = 0; i < NACC; ++i) c[i] = 0;
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) c[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += c[i];
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(256) void k_rate16(int iters, double *out) {
  v4f64 c[NACC];
  for (int i = 0; i < NACC; ++i) c[i] = v4f64{0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) c[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}
template <class K>
void time_it(const char *name, K kern, int wgs, double flop_per_wave_iter, double *out) {
  const int iters = 20000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), 0, 0, 100, out);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), 0, 0, iters, out);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
  const double inst = (double)iters * 4 * wgs;  // wave-iterations
  // cycles per instruction per SIMD at 2.4 GHz nominal, if every SIMD holds wgs * 4 / 1024 waves
  printf("%-28s workgroups %4d (%.1f waves/SIMD): %.2f ms, %.1f Tflop/s, %.1f ns per wave-iteration\n", name, wgs,
         wgs * 4.0 / 1024.0, ms, flop_per_wave_iter * inst / ms / 1e9, ms * 1e6 / iters);
}
int main() {
  double *out; (void)hipMalloc(&out, sizeof(double) * 64 * 4096);
  hipLaunchKernelGGL(k_probe, dim3(4096), dim3(64), 0, 0, out);
  std::vector<double> h((size_t)4096 * 64);
  (void)hipMemcpy(h.data(), out, h.size() * sizeof(double), hipMemcpyDeviceToHost);
  // for every (la, lb) with a non-zero result: which lanes of D
  int shown = 0;
  for (int la = 0; la < 64 && shown < 400; ++la)
    for (int lb = 0; lb < 64; ++lb) {
      for (int l = 0; l < 64; ++l)
        if (h[((size_t)la * 64 + lb) * 64 + l] != 0.0) {
          if (la < 20 || la % 16 == 0) printf("A lane %2d x B lane %2d -> D lane %2d\n", la, lb, l);
          ++shown;
        }
    }
  time_it("4x4x4_4b, 1 acc", k_rate4<1>, 256, 512.0 * 1, out);
  time_it("4x4x4_4b, 2 acc", k_rate4<2>, 256, 512.0 * 2, out);
  time_it("4x4x4_4b, 4 acc", k_rate4<4>, 256, 512.0 * 4, out);
  time_it("4x4x4_4b, 8 acc", k_rate4<8>, 256, 512.0 * 8, out);
  time_it("4x4x4_4b, 4 acc", k_rate4<4>, 1024, 512.0 * 4, out);
  time_it("4x4x4_4b, 8 acc", k_rate4<8>, 512, 512.0 * 8, out);
  time_it("16x16x4, 1 acc", k_rate16<1>, 256, 2048.0 * 1, out);
  time_it("16x16x4, 4 acc", k_rate16<4>, 256, 2048.0 * 4, out);
  time_it("16x16x4, 4 acc", k_rate16<4>, 1024, 2048.0 * 4, out);
  return 0;
}
