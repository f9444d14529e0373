// scratch: what does a dependent LEVEL cost when the levels of the sparse replay are chained INSIDE one launch instead
// of by kernel boundaries?  L levels of G workgroups (512 threads); a workgroup of level l prefetches its "weights",
// waits until all of level l - 1 has arrived, gathers values level l - 1 wrote (fine-grained memory: written through,
// visible to the other XCDs), writes its own and arrives.  Workgroups are dispatched in blockIdx order, so every
// producer is resident or finished before its consumers spin; spins are bounded (fail flag) all the same.
//   mode 0: no waiting at all (floor: launch + L generations of work, wrong values)
//   mode 1: one counter per level (agent-scope atomic add), every consumer polls it
//   mode 2: 32 shard counters per level -> top counter -> the last arrival stores 64 replicated "done" words, consumer
//           i polls word i % 64
//   mode 3 / 4: as 2 / 1 WITHOUT fences: the values travel as agent-scope relaxed atomic stores / loads (write-through,
//           read past the L2 of the XCD), a wave waits for its stores to be acknowledged (s_waitcnt) before the arrival
// build: hipcc --offload-arch=gfx950 -O3 tools/chain_probe.hip -o tools/bin/chain_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int kShards = 32, kCopies = 64, kStride = 32;  // words 128 B apart: one per channel line

struct Sync {
  unsigned *shard;  // [L][kShards] * kStride
  unsigned *top;    // [L] * kStride
  unsigned *done;   // [L][kCopies] * kStride
  unsigned *single; // [L] * kStride
};

__device__ inline unsigned ld(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <int MODEX>
__global__ __launch_bounds__(512) void k_chain(int L, int G, double *y, const double *w, int wsteps, Sync s,
                                               unsigned epoch, int *fail, double *sink) {
  constexpr bool FENCE = MODEX <= 2;
  constexpr int MODE = MODEX == 3 ? 2 : MODEX == 4 ? 1 : MODEX;
  const int level = blockIdx.x / G, i = blockIdx.x % G, tid = threadIdx.x;
  const int N = G * 512;
  double acc = 0;
  const double *wp = w + ((size_t)blockIdx.x * 512 + tid) * (size_t)wsteps;
  for (int u = 0; u < wsteps; ++u) acc += wp[u];  // requested before the wait, consumed after it
  if (MODE != 0 && level > 0) {
    if (tid == 0) {
      const long long t0 = wall_clock64();
      const unsigned *p = MODE == 1 ? s.single + (size_t)(level - 1) * kStride
                                    : s.done + ((size_t)(level - 1) * kCopies + (i % kCopies)) * kStride;
      const unsigned want = MODE == 1 ? epoch * (unsigned)G : epoch;
      while (ld(p) < want) {
        __builtin_amdgcn_s_sleep(2);
        if (wall_clock64() - t0 > 400000) {  // 4 ms at 100 MHz
          *fail = 1;
          break;
        }
      }
      if (FENCE) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
  }
  double v;
  if (level == 0) {
    v = epoch * 100.0;
  } else {
    const int idx = (int)(((long long)tid * 977 + (long long)i * 131071) % N);
    if (FENCE)
      v = __builtin_nontemporal_load(y + (size_t)(level - 1) * N + idx);
    else
      v = __hip_atomic_load(y + (size_t)(level - 1) * N + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  const double out = v + 1.0 + (acc == 12345.678 ? 1.0 : 0.0);
  if (FENCE)
    y[(size_t)level * N + (size_t)i * 512 + tid] = out;
  else
    __hip_atomic_store(y + (size_t)level * N + (size_t)i * 512 + tid, out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (MODE != 0) {
    if (FENCE)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");  // every wave drains its own stores
    else
      __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (tid == 0) {
      if (MODE == 1) {
        __hip_atomic_fetch_add(s.single + (size_t)level * kStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        const int sh = i % kShards;
        const unsigned in_shard = (unsigned)((G - sh + kShards - 1) / kShards);
        const unsigned a = __hip_atomic_fetch_add(s.shard + ((size_t)level * kShards + sh) * kStride, 1u,
                                                  __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (a + 1 == epoch * in_shard) {
          const unsigned b = __hip_atomic_fetch_add(s.top + (size_t)level * kStride, 1u, __ATOMIC_RELAXED,
                                                    __HIP_MEMORY_SCOPE_AGENT);
          const unsigned shards_used = (unsigned)(G < kShards ? G : kShards);
          if (b + 1 == epoch * shards_used)
            for (int c = 0; c < kCopies; ++c)
              __hip_atomic_store(s.done + ((size_t)level * kCopies + c) * kStride, epoch, __ATOMIC_RELAXED,
                                 __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
  }
  if (acc == 777.25) sink[0] = acc;
}

template <int MODE>
void run(const char *name, int L, int G, bool fine, int wsteps) {
  const int N = G * 512;
  double *y, *w, *sink;
  int *fail;
  Sync s;
  const size_t ybytes = (size_t)(L + 1) * N * 8;
  if (fine)
    (void)hipExtMallocWithFlags((void **)&y, ybytes, hipDeviceMallocFinegrained);
  else
    (void)hipMalloc(&y, ybytes);
  const size_t wn = (size_t)L * G * 512 * (size_t)(wsteps > 0 ? wsteps : 1);
  (void)hipMalloc(&w, wn * 8);
  (void)hipMemset(w, 0, wn * 8);
  (void)hipMalloc(&sink, 64);
  (void)hipMalloc(&fail, 4);
  (void)hipMemset(fail, 0, 4);
  const size_t words = (size_t)L * (kShards + 1 + kCopies + 1) * kStride;
  unsigned *pool;
  (void)hipExtMallocWithFlags((void **)&pool, words * 4, hipDeviceMallocFinegrained);
  (void)hipMemset(pool, 0, words * 4);
  s.shard = pool;
  s.top = s.shard + (size_t)L * kShards * kStride;
  s.done = s.top + (size_t)L * kStride;
  s.single = s.done + (size_t)L * kCopies * kStride;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  unsigned epoch = 0;
  for (int r = 0; r < 3; ++r)
    hipLaunchKernelGGL(k_chain<MODE>, dim3(L * G), dim3(512), 0, 0, L, G, y, w, wsteps, s, ++epoch, fail, sink);
  (void)hipEventRecord(e0);
  const int reps = 20;
  for (int r = 0; r < reps; ++r)
    hipLaunchKernelGGL(k_chain<MODE>, dim3(L * G), dim3(512), 0, 0, L, G, y, w, wsteps, s, ++epoch, fail, sink);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  int f = 0;
  (void)hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost);
  std::vector<double> last(N);
  (void)hipMemcpy(last.data(), y + (size_t)(L - 1) * N, (size_t)N * 8, hipMemcpyDeviceToHost);
  long bad = 0;
  for (int k = 0; k < N; ++k) bad += last[k] != epoch * 100.0 + L;
  printf("%-28s L=%2d G=%4d %s y, %3d KB of weights per workgroup: %7.1f us per launch, %5.2f us per level%s%s\n", name,
         L, G, fine ? "fine-grained" : "ordinary    ", wsteps * 4, ms * 1e3 / reps, ms * 1e3 / reps / L,
         f ? "  TIMED OUT" : "", (MODE != 0 && bad) ? "  WRONG VALUES" : "");
  (void)hipFree(y); (void)hipFree(w); (void)hipFree(sink); (void)hipFree(fail); (void)hipFree(pool);
}

int main() {
  for (int G : {64, 250, 600}) {
    for (int ws : {0, 8}) {
      run<0>("no waiting", 9, G, true, ws);
      run<1>("one counter per level", 9, G, true, ws);
      run<2>("shards + replicated done", 9, G, true, ws);
      run<2>("shards + replicated done", 9, G, false, ws);
      run<3>("shards, no fences", 9, G, true, ws);
      run<3>("shards, no fences", 9, G, false, ws);
      run<4>("one counter, no fences", 9, G, true, ws);
    }
  }
  run<3>("shards, no fences", 18, 300, true, 8);
  return 0;
}
