// scratch: what would ONE launch per tCG run cost at the headline size (k = 2000, r = 5: 250 workgroups of 2 poses)?
// A persistent tCG iteration has two grid-wide dependencies -- the Hessian product gathers the NEIGHBOURS' new direction,
// the dense preconditioner needs the WHOLE new residual in every workgroup (z_j = sum_i Minv[j, i] r_i) -- and inside one
// launch the data crosses the XCDs' private L2s only through the coherent level (write-through stores, loads past L2).
// Per iteration and workgroup this probe does what the data path of such a kernel would do, and nothing else:
//   phase A: gather NB x 40 doubles written by other workgroups in phase B of the iteration before, store 40 doubles,
//            arrive + wait (sharded counters, replicated done words: tools/chain_probe.hip mode 3)
//   phase B: read ALL 10 000 doubles the workgroups stored in phase A (the all-gather of H delta), sum them through LDS,
//            store 40 doubles, arrive + wait
// Modes: 0 barriers only (no data); 1 + neighbour gather; 2 + all-gather by 8-byte agent-scope loads; 3 + all-gather by
// 16-byte loads with sc1 (inline asm).  Spins are bounded (fail flag).  build:
//   hipcc --offload-arch=gfx950 -O3 tools/tcg_probe.hip -o tools/bin/tcg_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef unsigned v4u __attribute__((ext_vector_type(4)));
constexpr int kShards = 32, kCopies = 64, kStride = 32, kWG = 250, kPer = 40, kN = kWG * kPer, kNB = 17;

struct Sync {
  unsigned *shard, *top, *done;  // [2 phases][...] * kStride, epochs count up
};
__device__ inline unsigned ldu(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ inline bool grid_step(const Sync &s, int phase, unsigned epoch, int *fail) {
  // every wave has drained its stores (s_waitcnt) before the barrier below
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    const int i = blockIdx.x, sh = i % kShards;
    const unsigned in_shard = (unsigned)((kWG - sh + kShards - 1) / kShards);
    const unsigned a = __hip_atomic_fetch_add(s.shard + ((size_t)phase * kShards + sh) * kStride, 1u, __ATOMIC_RELAXED,
                                              __HIP_MEMORY_SCOPE_AGENT);
    if (a + 1 == epoch * in_shard) {
      const unsigned b = __hip_atomic_fetch_add(s.top + (size_t)phase * kStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (b + 1 == epoch * (unsigned)kShards)
        for (int c = 0; c < kCopies; ++c)
          __hip_atomic_store(s.done + ((size_t)phase * kCopies + c) * kStride, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const long long t0 = wall_clock64();
    const unsigned *p = s.done + ((size_t)phase * kCopies + (i % kCopies)) * kStride;
    while (ldu(p) < epoch) {
      __builtin_amdgcn_s_sleep(1);
      if (wall_clock64() - t0 > 200000) {  // 2 ms at 100 MHz: somebody is not resident
        *fail = 1;
        ok = false;
        break;
      }
    }
  }
  __syncthreads();
  return ok;
}

template <int MODE>
__global__ __launch_bounds__(256) void k_persist(int iters, double *h, double *z, Sync s, unsigned epoch0, int *fail,
                                                 double *out) {
  __shared__ double s_img[kN];
  __shared__ double s_red[4];
  const int tid = threadIdx.x, wg = blockIdx.x;
  double carry = 0;
  for (int it = 0; it < iters; ++it) {
    const unsigned epoch = epoch0 + (unsigned)it + 1;
    // ---- phase A ----
    double g = 0;
    if (MODE >= 1) {
      for (int e = tid; e < kNB * kPer; e += 256) {
        const int nb = (wg + 1 + (e / kPer) * 13) % kWG;
        g += __hip_atomic_load(z + (size_t)nb * kPer + (e % kPer), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    // (the value every reader must see this iteration: it + 1 in every slot -- a stale line would show the iteration before)
    if (tid < kPer) __hip_atomic_store(h + (size_t)wg * kPer + tid, (g != 0.123 ? 0.0 : 1.0) + it + 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!grid_step(s, 0, epoch, fail)) return;
    // ---- phase B ----
    double acc = 0;
    if (MODE == 2) {
      for (int e = tid; e < kN; e += 256) {
        const double v = __hip_atomic_load(h + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_img[e] = v;
        acc += v;
      }
    } else if (MODE == 3) {
      // 20 loads of 16 bytes per thread, five in flight at a time (the waitcnt names the registers, so that no use moves
      // above it)
      for (int b = 0; b < 4; ++b) {
        v4u v[5];
#pragma unroll
        for (int u = 0; u < 5; ++u) {
          const int e = (tid + (b * 5 + u) * 256) * 2;
          const double *p = h + (e < kN ? e : 0);
          asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v[u]) : "v"(p) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4])::"memory");
#pragma unroll
        for (int u = 0; u < 5; ++u) {
          const int e = (tid + (b * 5 + u) * 256) * 2;
          if (e < kN) {
            const double2 d = __builtin_bit_cast(double2, v[u]);
            s_img[e] = d.x;
            s_img[e + 1] = d.y;
            acc += d.x + d.y;
          }
        }
      }
    }
    else if (MODE == 4) {
      // plain 16-byte loads behind ONE invalidate of the caches per wave (what a kernel boundary does for free)
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      for (int e = tid * 2; e < kN; e += 512) {
        const double2 v = *reinterpret_cast<const double2 *>(h + e);
        s_img[e] = v.x;
        s_img[e + 1] = v.y;
        acc += v.x + v.y;
      }
    }
    __syncthreads();
    if (MODE >= 2) {
      // every slot of h must hold it + 1: the block sum of the image says whether a stale value was read
      double part = 0;
      for (int e = tid; e < kN; e += 256) part += s_img[e];
      if (fabs(part - (double)(it + 1) * ((kN - tid + 255) / 256)) > 1e-6) *fail = 2;
    }
    if (MODE >= 2) acc += s_img[(tid * 37) % kN];
    if (tid < kPer) __hip_atomic_store(z + (size_t)wg * kPer + tid, acc * 1e-9 + it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    carry += acc;
    if (!grid_step(s, 1, epoch, fail)) return;
  }
  if (carry == 12345.678) out[0] = carry;
}

template <int MODE>
void run(const char *name, int iters) {
  double *h, *z, *out;
  int *fail;
  (void)hipMalloc(&h, kN * 8);
  (void)hipMalloc(&z, kN * 8);
  (void)hipMemset(h, 0, kN * 8);
  (void)hipMemset(z, 0, kN * 8);
  (void)hipMalloc(&out, 64);
  (void)hipMalloc(&fail, 4);
  (void)hipMemset(fail, 0, 4);
  const size_t words = (size_t)2 * (kShards + 1 + kCopies) * kStride;
  unsigned *pool;
  (void)hipExtMallocWithFlags((void **)&pool, words * 4, hipDeviceMallocFinegrained);
  (void)hipMemset(pool, 0, words * 4);
  Sync s;
  s.shard = pool;
  s.top = s.shard + (size_t)2 * kShards * kStride;
  s.done = s.top + (size_t)2 * kStride;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  unsigned epoch = 0;
  hipLaunchKernelGGL(k_persist<MODE>, dim3(kWG), dim3(256), 0, 0, iters, h, z, s, epoch, fail, out);
  epoch += iters;
  (void)hipDeviceSynchronize();
  const int reps = 10;
  (void)hipEventRecord(e0);
  for (int r = 0; r < reps; ++r) {
    hipLaunchKernelGGL(k_persist<MODE>, dim3(kWG), dim3(256), 0, 0, iters, h, z, s, epoch, fail, out);
    epoch += iters;
  }
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  int f = 0;
  (void)hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost);
  printf("%-46s %3d iterations per launch: %7.1f us per launch, %6.2f us per tCG iteration (two grid steps)%s\n", name, iters,
         ms * 1e3 / reps, ms * 1e3 / reps / iters, f == 1 ? "  TIMED OUT" : f == 2 ? "  STALE VALUES READ" : "");
  (void)hipFree(h); (void)hipFree(z); (void)hipFree(out); (void)hipFree(fail); (void)hipFree(pool);
}

int main() {
  for (int iters : {1, 7, 50}) {
    run<0>("barriers only", iters);
    run<1>("+ neighbour gather (17 x 40 doubles)", iters);
    run<2>("+ all-gather of 80 KB, 8-byte agent loads", iters);
    run<3>("+ all-gather of 80 KB, 16-byte sc1 loads", iters);
    run<4>("+ all-gather, acquire fence + plain loads", iters);
  }
  return 0;
}
