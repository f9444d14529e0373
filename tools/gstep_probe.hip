// scratch: what does one grid-wide step cost among G resident workgroups of NT threads (G beyond one per CU)?  The step is
// run_grid_step of solver_fused.hip (sharded arrival counters, replicated done words, bounded spin).
//   hipcc --offload-arch=gfx950 -O3 tools/gstep_probe.hip -o tools/bin/gstep_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
constexpr int kShards = 32, kCopies = 64, kStride = 32;
__device__ inline bool grid_step(unsigned *sync, unsigned step, int *s_ok) {
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned *shard = sync, *top = sync + kShards * kStride, *done = top + kStride, *abort_w = done + kCopies * kStride;
    const int i = blockIdx.x, G = gridDim.x, sh = i % kShards;
    const unsigned in_shard = (unsigned)((G - sh + kShards - 1) / kShards), want = step + 1;
    const unsigned a = __hip_atomic_fetch_add(shard + sh * kStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (a + 1 == want * in_shard) {
      const unsigned used = (unsigned)(G < kShards ? G : kShards);
      const unsigned b = __hip_atomic_fetch_add(top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (b + 1 == want * used)
        for (int c = 0; c < kCopies; ++c) __hip_atomic_store(done + c * kStride, want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const long long t0 = wall_clock64();
    const unsigned *p = done + (i % kCopies) * kStride;
    int ok = 1;
    unsigned spins = 0;
    while (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
      __builtin_amdgcn_s_sleep(1);
      if ((++spins & 63u) == 0) {
        if (__hip_atomic_load(abort_w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { ok = 0; break; }
        if (wall_clock64() - t0 > 200000) { __hip_atomic_store(abort_w, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ok = 0; break; }
      }
    }
    *s_ok = ok;
  }
  __syncthreads();
  return *s_ok != 0;
}
template <int NT>
__global__ __launch_bounds__(NT) void k_steps(int steps, unsigned *sync, int *fail, double *v, int work) {
  __shared__ int s_ok;
  double acc = 0;
  for (int s = 0; s < steps; ++s) {
    // a little exchanged data per step: every thread stores one value through the coherent level and reads a neighbour's
    if (work) {
      const size_t n = (size_t)gridDim.x * NT;
      const size_t me = (size_t)blockIdx.x * NT + threadIdx.x;
      __hip_atomic_store(v + (s & 1) * n + me, acc + 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (!grid_step(sync, (unsigned)s, &s_ok)) { if (threadIdx.x == 0) *fail = 1; return; }
    if (work) {
      const size_t n = (size_t)gridDim.x * NT;
      const size_t nb = ((size_t)(blockIdx.x + 7) % gridDim.x) * NT + threadIdx.x;
      acc += __hip_atomic_load(v + (s & 1) * n + nb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (acc == -1.0) v[0] = acc;
}
int main() {
  unsigned *sync; int *fail; double *v;
  const int words = (kShards + 1 + kCopies + 1) * kStride;
  hipMalloc(&sync, words * 4); hipMalloc(&fail, 4); hipMalloc(&v, 2 * 2048 * 512 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int steps = 2000;
  for (int nt : {256, 512})
    for (int work = 0; work < 2; ++work)
      for (int G : {256, 512, 768, 1024, 1536, 2048}) {
        if ((long)G * nt > 256L * 2048) continue;
        float best = 1e9;
        int hf = 0;
        for (int rep = 0; rep < 3; ++rep) {
          hipMemset(sync, 0, words * 4); hipMemset(fail, 0, 4);
          hipEventRecord(e0);
          if (nt == 256) hipLaunchKernelGGL(k_steps<256>, dim3(G), dim3(256), 0, 0, steps, sync, fail, v, work);
          else hipLaunchKernelGGL(k_steps<512>, dim3(G), dim3(512), 0, 0, steps, sync, fail, v, work);
          hipEventRecord(e1); hipEventSynchronize(e1);
          float ms; hipEventElapsedTime(&ms, e0, e1);
          hipMemcpy(&hf, fail, 4, hipMemcpyDeviceToHost);
          if (ms < best) best = ms;
        }
        printf("NT %d G %4d work %d: %.2f us per step%s\n", nt, G, work, best * 1e3 / steps, hf ? "  (GAVE UP)" : "");
      }
  return 0;
}
