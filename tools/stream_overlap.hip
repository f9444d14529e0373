// scratch: do N HIP streams overlap their chains of small dependent kernels on this part?  Each stream runs `len` launches
// of a one-workgroup kernel that spins ~`us` microseconds; one host thread per stream (threads = 1) or one thread enqueuing
// all of them (threads = 0).  Prints wall time per configuration; GPU_MAX_HW_QUEUES may be set from outside.
// build: hipcc --offload-arch=gfx950 -O3 tools/stream_overlap.hip -o tools/bin/stream_overlap -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>

__global__ void k_spin(long long ticks, int *sink) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) {
  }
  if (ticks < 0) *sink = 1;
}

static double run(int nstreams, int len, int us, bool threaded, bool nonblocking) {
  std::vector<hipStream_t> st(nstreams);
  for (auto &s : st) (void)hipStreamCreateWithFlags(&s, nonblocking ? hipStreamNonBlocking : hipStreamDefault);
  int *sink;
  (void)hipMalloc(&sink, 4);
  auto enqueue = [&](int i) {
    for (int q = 0; q < len; ++q) hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, st[i], (long long)us * 100, sink);
  };
  // warm-up
  for (int i = 0; i < nstreams; ++i) hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, st[i], 100LL, sink);
  (void)hipDeviceSynchronize();
  const auto t0 = std::chrono::steady_clock::now();
  if (threaded) {
    std::vector<std::thread> th;
    for (int i = 0; i < nstreams; ++i)
      th.emplace_back([&, i] {
        enqueue(i);
        (void)hipStreamSynchronize(st[i]);
      });
    for (auto &t : th) t.join();
  } else {
    for (int i = 0; i < nstreams; ++i) enqueue(i);
    for (int i = 0; i < nstreams; ++i) (void)hipStreamSynchronize(st[i]);
  }
  const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  for (auto &s : st) (void)hipStreamDestroy(s);
  (void)hipFree(sink);
  return ms;
}

int main() {
  const int len = 150, us = 20;
  printf("chains of %d kernels of %d us: one stream alone = %.2f ms\n", len, us, run(1, len, us, false, true));
  for (int n : {2, 4, 5, 8, 16}) {
    printf("%2d streams: one host thread %.2f ms, a thread per stream %.2f ms, (blocking streams, threads) %.2f ms\n", n,
           run(n, len, us, false, true), run(n, len, us, true, true), run(n, len, us, true, false));
  }
  return 0;
}
