// scratch: issue rate of v_mfma_f64_16x16x4_f64 on this GPU (independent accumulators, operands in registers)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4f64 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k_mfma(int iters, double *out) {
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
template <int NACC>
void run(int wgs, double *out) {
  const int iters = 20000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k_mfma<NACC>, dim3(wgs), dim3(256), 0, 0, 100, out);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k_mfma<NACC>, dim3(wgs), dim3(256), 0, 0, iters, out);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
  const double flop = 2048.0 * NACC * (double)iters * 4 * wgs;
  printf("accumulators %d, workgroups %d (%.1f waves/SIMD): %.2f ms, %.1f Tflop/s\n", NACC, wgs, wgs * 4.0 / 1024.0, ms, flop / ms / 1e9);
}
// the vector pipe for comparison: 16 independent v_fma_f64 chains per lane
__global__ __launch_bounds__(256) void k_fma(int iters, double *out) {
  double c[16];
  for (int i = 0; i < 16; ++i) c[i] = threadIdx.x * 1e-6 + i;
  const double a = 1.0000001, b = 1e-9;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) c[i] = fma(c[i], a, b);
  }
  double s = 0;
  for (int i = 0; i < 16; ++i) s += c[i];
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}
void run_fma(int wgs, double *out) {
  const int iters = 20000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k_fma, dim3(wgs), dim3(256), 0, 0, 100, out);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k_fma, dim3(wgs), dim3(256), 0, 0, iters, out);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
  const double flop = 2.0 * 16 * (double)iters * 256 * wgs;
  printf("v_fma_f64, workgroups %d (%.1f waves/SIMD): %.2f ms, %.1f Tflop/s\n", wgs, wgs * 4.0 / 1024.0, ms, flop / ms / 1e9);
}
int main() {
  double *out; (void)hipMalloc(&out, sizeof(double) * 256 * 4096);
  run<1>(256, out); run<2>(256, out); run<4>(256, out); run<8>(256, out);
  run<4>(512, out); run<4>(1024, out); run<8>(512, out);
  run_fma(512, out); run_fma(1024, out); run_fma(2048, out);
  return 0;
}
