// scratch: read bandwidth of a buffer that fits the Infinity Cache (160 MB) against one that does not (2 GB), for
// 8- and 16-byte loads per lane, 512-byte runs per wave instruction (the access shape of k_sp_mtile's weight loads)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int W>  // doubles per lane per load
__global__ __launch_bounds__(512) void k_read(const double *__restrict__ a, size_t n, double *out, int per_wave) {
  // every wave reads `per_wave` consecutive runs of 64 W doubles
  const size_t wave = ((size_t)blockIdx.x * 512 + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  const size_t base = wave * (size_t)per_wave * 64 * W;
  double s = 0;
  if (base + (size_t)per_wave * 64 * W <= n) {
#pragma unroll 8
    for (int u = 0; u < per_wave; ++u) {
      const double *p = a + base + (size_t)u * 64 * W + (size_t)lane * W;
      if (W == 1) s += p[0];
      else { const double2 v = *reinterpret_cast<const double2 *>(p); s += v.x + v.y; }
    }
  }
  if (s == 12345.678) out[0] = s;
}
template <int W>
void run(const char *name, const double *a, size_t n, double *out, int per_wave) {
  const size_t waves = n / ((size_t)per_wave * 64 * W);
  const int wgs = (int)((waves + 7) / 8);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k_read<W>, dim3(wgs), dim3(512), 0, 0, a, n, out, per_wave);
  (void)hipEventRecord(e0);
  const int reps = 20;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k_read<W>, dim3(wgs), dim3(512), 0, 0, a, n, out, per_wave);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%-10s %5.0f MB, %d-byte loads, %2d runs per wave: %7.1f us per pass, %.2f TB/s\n", name, n * 8e-6, 8 * W, per_wave,
         ms * 1e3 / reps, n * 8.0 * reps / ms / 1e9);
}
int main() {
  const size_t nbig = (size_t)256 << 20, nsmall = (size_t)20 << 20, n30 = (size_t)30e6 / 8;  // 2 GB, 160 MB, 30 MB
  double *a, *out; (void)hipMalloc(&a, nbig * 8); (void)hipMalloc(&out, 64);
  (void)hipMemset(a, 0, nbig * 8);
  for (int pw : {8, 16, 64}) {
    run<1>("fits", a, nsmall, out, pw); run<2>("fits", a, nsmall, out, pw);
    run<1>("hbm", a, nbig, out, pw); run<2>("hbm", a, nbig, out, pw);
    run<1>("30MB", a, n30, out, pw); run<2>("30MB", a, n30, out, pw);
  }
  return 0;
}
