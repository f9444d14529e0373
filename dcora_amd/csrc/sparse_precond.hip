// Device side of the sparse preconditioner (sparse_precond.h): uploads the partitioned inverse and replays its
// schedule, one launch of k_sp_mtile per group of tree levels.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "device_problem.h"
#include "env.h"
#include "host_partinv_int.h"

namespace dcora {

namespace {

__device__ __forceinline__ bool sp_gated(const SolverCtl *ctl, int seq, int gate) {
  if (gate == 0 || ctl == nullptr) return false;
  if (seq > ctl->outer_done_stamp) return true;
  if (gate == 2 && seq > ctl->tcg_done_stamp) return true;
  return false;
}

// y[0][j] = R[perm[j]]  (k unknowns of r values each)
__global__ __launch_bounds__(kBlock) void k_sp_permute_in(int r, int k, const int *__restrict__ perm, Buf2 Rb,
                                                          double *__restrict__ y, Gate g) {
  if (sp_gated(g.ctl, g.seq, g.gate)) return;
  const double *__restrict__ R = Rb.p[g.ctl ? (g.ctl->cur & 1) : 0];
  const long n = (long)r * k;
  for (long e = (long)blockIdx.x * kBlock + threadIdx.x; e < n; e += (long)gridDim.x * kBlock) {
    const int j = (int)(e / r), t = (int)(e - (long)j * r);
    y[e] = R[(size_t)perm[j] * r + t];
  }
}
// Z[perm[j]] = y[out_off[j]]
__global__ __launch_bounds__(kBlock) void k_sp_permute_out(int r, int k, const int *__restrict__ perm,
                                                           const int *__restrict__ out_off,
                                                           const double *__restrict__ y, double *__restrict__ Z,
                                                           Gate g) {
  if (sp_gated(g.ctl, g.seq, g.gate)) return;
  const long n = (long)r * k;
  for (long e = (long)blockIdx.x * kBlock + threadIdx.x; e < n; e += (long)gridDim.x * kBlock) {
    const int j = (int)(e / r), t = (int)(e - (long)j * r);
    Z[(size_t)perm[j] * r + t] = y[(size_t)out_off[j] * r + t];
  }
}

// ---- hubs (PartInvHub): Schur complement on top of the replay ----
struct HubDev {
  int h;
  const int *idx;
  const double *U;
  const double *x2;  // h x r: Sinv (R(hub) - U^T r1), formed by hub_x2 below
};
// w(q, :) = R(hub_q, :) - a_q^T y1 with y1 = A11^-1 r1, and a_q^T A11^-1 r1 = U(:, q)^T r1 (U = A11^-1 a is stored for
// the correction anyway, A11 symmetric): the dot runs over the INPUT image of the replay -- dense, contiguous, and it
// needs no level of the replay, so its workgroups ride in the replay's first launch (they were a launch of their own
// after the last one: 5.1 us + a gap per tCG iteration on tiers.pyfg).  The sum over the k rows is split over kHubSplit
// workgroups per hub; each writes its partial to w[q][slice][t] and the consumer adds the slices in a fixed order.
constexpr int kHubSplit = 32;
struct HubIn {
  int h = 0, k = 0;
  int stage = 0;  // 1: the slices of U^T r1 (replay's first launch); 2: x2 from the slices (one workgroup, second launch)
  const double *U = nullptr;
  double *w = nullptr;  // h x kHubSplit x r slices
  const int *idx = nullptr;
  const double *Sinv = nullptr;
  Buf2 R{{nullptr, nullptr}};
  double *x2 = nullptr;
};
template <int NT>
__device__ __forceinline__ void hub_slice(int r, const HubIn &H, const double *__restrict__ y0, int b, double *s_part) {
  const int q = b / kHubSplit, sl = b - q * kHubSplit;
  const int RB = NT / r;
  const int lj = threadIdx.x / r, t = threadIdx.x - lj * r;
  const int per = (H.k + kHubSplit - 1) / kHubSplit;
  const int lo = sl * per, hi = min(H.k, lo + per);
  double acc = 0;
  if (lj < RB)
    for (int j = lo + lj; j < hi; j += RB) acc = fma(H.U[(size_t)j * H.h + q], y0[(size_t)j * r + t], acc);
  s_part[threadIdx.x] = (lj < RB) ? acc : 0.0;
  __syncthreads();
  if ((int)threadIdx.x < r) {
    double s = 0;
    for (int u = 0; u < RB; ++u) s += s_part[u * r + threadIdx.x];
    H.w[((size_t)q * kHubSplit + sl) * r + threadIdx.x] = s;
  }
}
// x2 = Sinv w with w(q, :) = R(hub_q, :) - the slices in slice order: h r values, ONE workgroup, riding in the replay's
// second launch (the consumers -- k_tangent of the generic solver, k_sp_permute_out_hub -- rebuilt it in every workgroup:
// two dependent rounds of loads and two barriers in front of their real work).  s_w: h * r doubles of LDS.
template <int NT>
__device__ __forceinline__ void hub_x2(int r, const HubIn &H, const SolverCtl *ctl, double *s_w) {
  const double *__restrict__ R = H.R.p[ctl ? (ctl->cur & 1) : 0];
  for (int e = threadIdx.x; e < H.h * r; e += NT) {
    const int q = e / r, t = e - q * r;
    double s = 0;
    for (int sl = 0; sl < kHubSplit; ++sl) s += H.w[((size_t)q * kHubSplit + sl) * r + t];
    s_w[e] = R[(size_t)H.idx[q] * r + t] - s;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < H.h * r; e += NT) {
    const int q = e / r, t = e - q * r;
    double s = 0;
    for (int q2 = 0; q2 < H.h; ++q2) s += H.Sinv[(size_t)q * H.h + q2] * s_w[q2 * r + t];
    H.x2[e] = s;
  }
}
// (a replay of fewer than two launches: the stage that found no launch to ride in)
__global__ __launch_bounds__(kBlock) void k_sp_hub_stage(int r, HubIn H, const double *__restrict__ y0, Gate g) {
  if (sp_gated(g.ctl, g.seq, g.gate)) return;
  __shared__ double s_part[64 * 16];
  if (H.stage == 1)
    hub_slice<kBlock>(r, H, y0, blockIdx.x, s_part);
  else
    hub_x2<kBlock>(r, H, g.ctl, s_part);
}
// Z[perm[j]] = y1[j] - U(j, :) x2;  Z[hub_q] = x2(q, :)
__global__ __launch_bounds__(kBlock) void k_sp_permute_out_hub(int r, int k, const int *__restrict__ perm,
                                                               const int *__restrict__ out_off,
                                                               const double *__restrict__ y,
                                                               double *__restrict__ Z, HubDev H, Gate g) {
  if (sp_gated(g.ctl, g.seq, g.gate)) return;
  __shared__ double s_x2[64 * 16];
  const int h = H.h;
  for (int e = threadIdx.x; e < h * r; e += kBlock) {
    const double s = H.x2[e];
    s_x2[e] = s;
    if (blockIdx.x == 0) Z[(size_t)H.idx[e / r] * r + (e - (e / r) * r)] = s;
  }
  __syncthreads();
  const long n = (long)r * k;
  for (long e = (long)blockIdx.x * kBlock + threadIdx.x; e < n; e += (long)gridDim.x * kBlock) {
    const int j = (int)(e / r), t = (int)(e - (long)j * r);
    double v = y[(size_t)out_off[j] * r + t];
    for (int q = 0; q < h; ++q) v -= H.U[(size_t)j * h + q] * s_x2[q * r + t];
    Z[(size_t)perm[j] * r + t] = v;
  }
}

// ---- matrix-pipe schedule (sparse_precond.h, host_partinv3.cpp): one launch of tiles on v_mfma_f64_4x4x4_4b_f64 ----
// The instruction multiplies four independent 4 x 4 blocks.  Measured on MI355X (tools/mfma_f64_4x4.hip): lane
// l = 16 k + 4 b + i holds A_b[i][k], lane 16 k + 4 b + j holds B_b[k][j], lane 16 i + 4 b + j receives D_b[i][j];
// 65-67 Tflop/s with two waves per SIMD (v_mfma_f64_16x16x4: 47), about 17 clocks per instruction.
// A step of a tile: block b takes the group g = 4 s + b of four K entries -- one micro-block of the stored matrix
// (128 bytes; direct: micro-block (row / 4, g), transposed: micro-block (g, col0 / 4) read across) -- and the r values of
// its four entries, two columns per lane by one 16-byte load (lane j: columns 2 j and 2 j + 1; r > 8: a second load for
// 8 + 2 j, 9 + 2 j), so a step is 512 bytes of weights, one weight load, NC vector loads, 2 NC MFMAs.  The four blocks
// hold partial sums over different entries: two DPP row rotations add them at the end of the tile.
// A wave executes one record (MWave): up to two runs of steps; the loads of U steps are requested together, whichever
// run they belong to (uniform selects, straight-line code); the waves of a tile add their partial sums through LDS in
// wave order: fixed summation order, bitwise reproducible.
typedef double sp_v2f64u __attribute__((ext_vector_type(2), aligned(8)));  // a pair of vector values: 8-byte aligned
constexpr int kMtBlock = kMtWaves * 64;

template <typename T>
__device__ __forceinline__ const T *mt_at(const void *base, unsigned byte_offset) {
  return reinterpret_cast<const T *>(reinterpret_cast<const char *>(base) + byte_offset);
}
// what a lane needs to address one run
struct MtRun {
  const double *W;
  const int *ix;   // KIND 1: the index list
  int abase, gstride, ng, src, s0, n;
  bool valid;
};
template <int KIND>
__device__ __forceinline__ MtRun mt_run(const MSub &U, const double *__restrict__ vals, const int *__restrict__ idxs,
                                        int kq, int li, int nrows) {
  MtRun R;
  R.W = vals + U.w;
  R.ng = U.ng;
  R.s0 = U.s0;
  R.n = U.n;
  if (KIND == 0) {
    const int a = li == 0 ? U.loc[0] : li == 1 ? U.loc[1] : li == 2 ? U.loc[2] : U.loc[3];
    R.valid = a >= 0;
    R.abase = (a >> 2) * U.ncb * 16 + kq * 4 + (a & 3);
    R.gstride = 16;
    R.src = U.src;
    R.ix = nullptr;
  } else {
    const int col = U.loc[0] + li;
    R.valid = li < nrows;
    R.abase = (col >> 2) * 16 + (col & 3) * 4 + kq;
    R.gstride = U.ncb * 16;
    R.src = 0;
    R.ix = idxs + U.src;
  }
  if (!R.valid) R.abase = 0;
  return R;
}

// All steps of a record: flattened step f < a.n belongs to run a, the others to run b; the loads of U steps are requested
// together whichever run they belong to (indices, then weights, then vector values; uniform selects, straight-line
// code), then their MFMAs issue.  Measured against this form on a lattice agent (102 us per application): a rolling
// window that refills every slot right after its MFMAs 123 us; U = 12 / 16 131 / 145 us; per-run linear pointers with
// immediate offsets, four slots per run, 126 us (108 with two) -- SQ counters put 63 % of the wave cycles into waiting
// for memory and 230 vector instructions into a wave: the kernel is bound by the three dependent round trips of a wave
// (record, indices, operands) inside launches of 30 MB, not by its instruction stream.
template <int KIND, int NC, int U, bool HALF>
__device__ __forceinline__ void mt_record(const MtRun &a, const MtRun &b, int kq, int blk, int jj, int r,
                                          const double *__restrict__ y, double (&d)[NC][2][2]) {
  const int ntot = a.n + b.n;
  for (int f0 = 0; f0 < ntot; f0 += U) {
    int p[U], ao[U];
    const double *wp[U];
    bool ok[U];
    double av[U];
    sp_v2f64u bv[U][NC];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int f = f0 + u;
      const bool ina = f < a.n;  // uniform
      const int s = ina ? a.s0 + f : b.s0 + (f - a.n);
      const int ng = ina ? a.ng : b.ng;
      const int g = 4 * s + blk;
      ok[u] = f < ntot && g < ng && (ina ? a.valid : b.valid);
      const int gc = (f < ntot && g < ng) ? g : 0;
      wp[u] = ina ? a.W : b.W;
      ao[u] = (ina ? a.abase : b.abase) + gc * (ina ? a.gstride : b.gstride);
      if (KIND == 0)
        p[u] = (ina ? a.src : b.src) + 4 * gc + kq;
      else
        p[u] = *mt_at<int>(ina ? a.ix : b.ix, 4u * (unsigned)(4 * gc + kq));
    }
    // (uniform base + 32-bit byte offset: the address arithmetic of a load is one 32-bit multiply-add instead of a
    // 64-bit one -- a wave spends most of its vector instructions on addresses)
#pragma unroll
    for (int u = 0; u < U; ++u) av[u] = ok[u] ? *mt_at<double>(wp[u], 8u * (unsigned)ao[u]) : 0.0;
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        if (HALF)  // r <= 4: lane j holds column j, one 8-byte load
          bv[u][c].x = *mt_at<double>(y, 8u * (unsigned)(p[u] * r + jj));
        else
          bv[u][c] = *mt_at<sp_v2f64u>(y, 8u * (unsigned)(p[u] * r + jj + 8 * c));
      }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        d[c][0][u & 1] = __builtin_amdgcn_mfma_f64_4x4x4f64(av[u], bv[u][c].x, d[c][0][u & 1], 0, 0, 0);
        if (!HALF) d[c][1][u & 1] = __builtin_amdgcn_mfma_f64_4x4x4f64(av[u], bv[u][c].y, d[c][1][u & 1], 0, 0, 0);
      }
  }
}

// a wave-uniform record, read through the scalar cache (constant address space) into SGPRs, field by field (a union
// or a struct copy through this address space sends the record through scratch memory)
__device__ __forceinline__ void mt_load_sub(const __attribute__((address_space(4))) int *p, MSub &U) {
  U.w = (long long)(((unsigned long long)(unsigned)p[1] << 32) | (unsigned long long)(unsigned)p[0]);
  U.ncb = p[2];
  U.ng = p[3];
  U.src = p[4];
  U.s0 = p[5];
  U.n = p[6];
  U.pad = 0;
  U.loc[0] = p[8];
  U.loc[1] = p[9];
  U.loc[2] = p[10];
  U.loc[3] = p[11];
}

// HALF (r <= 4): one column per lane instead of a pair -- half the vector bytes, half the MFMAs, 16 registers fewer (8
// waves per SIMD instead of 6): the same sums in the same order as the pair form
template <int NC, int U, bool HALF>
__global__ __launch_bounds__(kMtBlock) void k_sp_mtile(const MWave *__restrict__ recs, const double *__restrict__ vals,
                                                        const int *__restrict__ idxs, double *__restrict__ y, int r,
                                                        Gate g, int nhubwg, HubIn hub) {
  __shared__ double s_part[kMtWaves][NC * 2][64];
  // the hubs' dots over the input image (first launch only: nothing has written it yet); they are long chains of few
  // waves, so they take the grid's first workgroups (at its end they were a 3 us tail)
  if ((int)blockIdx.x < nhubwg) {
    if (!sp_gated(g.ctl, g.seq, g.gate)) {
      if (hub.stage == 1)
        hub_slice<kMtBlock>(r, hub, y, (int)blockIdx.x, &s_part[0][0][0]);
      else
        hub_x2<kMtBlock>(r, hub, g.ctl, &s_part[0][0][0]);
    }
    return;
  }
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  typedef const __attribute__((address_space(4))) int *ConstInts;
  ConstInts rp = (ConstInts)(recs + ((size_t)((int)blockIdx.x - nhubwg) * kMtWaves + wave));
  const int t_out = rp[0], t_carry = rp[1], t_nrows = rp[2], t_kind = rp[3], t_first = rp[4], t_n = rp[5], t_solo = rp[7];
  // the gate is tested AFTER the record has been requested: the two loads travel together
  if (sp_gated(g.ctl, g.seq, g.gate)) return;
  const int kq = lane >> 4, blk = (lane >> 2) & 3, li = lane & 3;  // operand roles: K entry, block, row (A) / pair (B)
  constexpr int NH = HALF ? 1 : 2;  // columns a lane holds per 8-column group
  const int jj = HALF ? li : 2 * li;
  // the old values this tile adds to (result roles: row kq, block 0, columns 8 c + 2 li + h); first wave of the tile only
  const bool writer = t_nrows > 0 && t_first == wave && blk == 0 && kq < t_nrows;
  double cv[NC][2];
#pragma unroll
  for (int c = 0; c < NC; ++c)
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      const int col = 8 * c + jj + h;
      cv[c][h] = (writer && t_carry >= 0 && col < r) ? y[(size_t)(t_carry + kq) * r + col] : 0.0;
    }
  double d[NC][2][2];
#pragma unroll
  for (int c = 0; c < NC; ++c)
#pragma unroll
    for (int h = 0; h < 2; ++h) d[c][h][0] = d[c][h][1] = 0.0;
  for (;;) {
    MSub sa, sb;
    mt_load_sub(rp + 8, sa);
    mt_load_sub(rp + 20, sb);
    const int next = rp[6];
    {
      if (t_kind == 0)
        mt_record<0, NC, U, HALF>(mt_run<0>(sa, vals, idxs, kq, li, t_nrows), mt_run<0>(sb, vals, idxs, kq, li, t_nrows),
                                  kq, blk, jj, r, y, d);
      else
        mt_record<1, NC, U, HALF>(mt_run<1>(sa, vals, idxs, kq, li, t_nrows), mt_run<1>(sb, vals, idxs, kq, li, t_nrows),
                                  kq, blk, jj, r, y, d);
    }
    if (next < 0) break;
    rp = (ConstInts)(recs + next);  // tiles of many short segments: the wave's work continues in a chained record
  }
  double sum[NC][2];
#pragma unroll
  for (int c = 0; c < NC; ++c)
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      double v = d[c][h][0] + d[c][h][1];
      v += dpp_move<0x124>(v);  // row_ror:4
      v += dpp_move<0x128>(v);  // row_ror:8: the four blocks of a 16-lane row summed
      sum[c][h] = v;
    }
  if (!t_solo) {  // uniform over the workgroup
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int h = 0; h < NH; ++h) s_part[wave][c * 2 + h][lane] = sum[c][h];
    __syncthreads();
  }
  if (!writer) return;
#pragma unroll
  for (int c = 0; c < NC; ++c)
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      double t = sum[c][h];
      for (int w = 1; w < t_n; ++w) t += s_part[wave + w][c * 2 + h][lane];
      const int col = 8 * c + jj + h;
      if (col < r) y[(size_t)(t_out + kq) * r + col] = cv[c][h] + t;
    }
}

bool launch_mtile(hipStream_t st, int r, const SpLevel &lv, const MWave *recs, const double *vals, const int *idxs,
                  double *y, Gate g, const HubIn &hub = HubIn()) {
  if (r < 1 || r > 16) return false;
  const int extra = hub.stage == 1 ? hub.h * kHubSplit : (hub.stage == 2 ? 1 : 0);
  if (lv.ntasks + extra == 0) return true;
  const MWave *rp = recs + lv.task0;
  const dim3 grid(lv.ntasks + extra);
  if (r <= 4)
    hipLaunchKernelGGL((k_sp_mtile<1, 8, true>), grid, dim3(kMtBlock), 0, st, rp, vals, idxs, y, r, g, extra, hub);
  else if (r <= 8)
    hipLaunchKernelGGL((k_sp_mtile<1, 8, false>), grid, dim3(kMtBlock), 0, st, rp, vals, idxs, y, r, g, extra, hub);
  else
    hipLaunchKernelGGL((k_sp_mtile<2, 4, false>), grid, dim3(kMtBlock), 0, st, rp, vals, idxs, y, r, g, extra, hub);
  return true;
}

}  // namespace

namespace {
// pinned chunk buffers, recycled: pinning 64 MB costs milliseconds, and the agents of a session build side by side
std::mutex g_pin_mu;
std::vector<double *> g_pin_free;
double *pin_acquire() {
  {
    std::lock_guard<std::mutex> lk(g_pin_mu);
    if (!g_pin_free.empty()) {
      double *p = g_pin_free.back();
      g_pin_free.pop_back();
      return p;
    }
  }
  double *p = nullptr;
  if (hipHostMalloc((void **)&p, (size_t)DeviceWeightSink::kChunk * sizeof(double), hipHostMallocDefault) != hipSuccess)
    return nullptr;
  return p;
}
void pin_release(double *p) {
  if (!p) return;
  std::lock_guard<std::mutex> lk(g_pin_mu);
  if (g_pin_free.size() < 6) {
    g_pin_free.push_back(p);
    return;
  }
  (void)hipHostFree(p);
}
}  // namespace

// ---- the stored weights formed ON THE DEVICE: a fill record names a run of micro-blocks of a source matrix
//      (host_partinv_int.h); a wave per record writes the record's whole extent, zeros included ----
namespace {
struct DFill {
  long long off, extent;
  const double *src;
  int c, a0, m, cb0, ncb;
};
// micro-blocks (host_partinv_int.h): e = (cb - cb0) 16 + (col % 4) 4 + row % 4
__global__ __launch_bounds__(256) void k_fill_weights(long long nf, const DFill *__restrict__ fills,
                                                      double *__restrict__ vals) {
  const long long fi = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (fi >= nf) return;
  const DFill F = fills[fi];
  const int lane = threadIdx.x & 63;
  double *__restrict__ w = vals + F.off;
  const long long n = (long long)F.ncb * 16;
  for (long long e = lane; e < F.extent; e += 64) {
    const int cb = F.cb0 + (int)(e >> 4), col = cb * 4 + (int)((e >> 2) & 3), row = F.a0 + (int)(e & 3);
    w[e] = (e < n && row < F.m && col < F.c) ? F.src[(size_t)row * F.c + col] : 0.0;
  }
}
}  // namespace

bool DeviceWeightSink::wants_device_sources() const {
  return !env::fill_on_host();
}

bool DeviceWeightSink::fill_on_device(const std::vector<partinv::Fill> &fills, long long total,
                                      const std::vector<MirrorRange> &mirrors, int nthreads) {
  using partinv::Fill;
  if (!wants_device_sources() || mirrors.empty()) return false;
  if (hipSetDevice(device) != hipSuccess) return false;
  const long long nf = (long long)fills.size();
  // ascending order of the offsets (the builders lay the fills out that way; checked, not assumed)
  std::vector<long long> order;
  for (long long i = 1; i < nf; ++i)
    if (fills[(size_t)i].off < fills[(size_t)i - 1].off) {
      order.resize((size_t)nf);
      for (long long q = 0; q < nf; ++q) order[(size_t)q] = q;
      std::stable_sort(order.begin(), order.end(),
                       [&](long long a, long long b) { return fills[(size_t)a].off < fills[(size_t)b].off; });
      break;
    }
  auto at = [&](long long i) -> const Fill & { return fills[(size_t)(order.empty() ? i : order[(size_t)i])]; };
  // sources: inside a mirrored range -> the device copy; anything else (merged products, compacted W of pieces with
  // dropped rows) is staged: rows touched x columns of each distinct source, one upload
  auto mirrored = [&](const double *b) -> const double * {
    for (const MirrorRange &m : mirrors)
      if (b >= m.host && b < m.host + m.n) return m.dev + (b - m.host);
    return nullptr;
  };
  auto rows_touched = [](const Fill &f) -> long long { return std::min<long long>((long long)f.a0 + 4, f.m); };
  struct Staged {
    long long doubles = 0, offset = 0;
  };
  std::unordered_map<const double *, Staged> staged;
  for (const Fill &f : fills) {
    if (f.len <= 0 || f.nrows <= 0 || mirrored(f.base)) continue;
    Staged &sg = staged[f.base];
    sg.doubles = std::max(sg.doubles, rows_touched(f) * (long long)f.c);
  }
  long long stage_total = 0;
  std::vector<std::pair<const double *, Staged *>> slist;
  for (auto &kv : staged) {
    kv.second.offset = stage_total;
    stage_total += (kv.second.doubles + 1) & ~1LL;
    slist.emplace_back(kv.first, &kv.second);
  }
  DevBuf<double> dstage;
  if (stage_total > 0) {
    std::vector<double> hstage((size_t)stage_total);
    partinv::parallel_for((int)slist.size(), std::max(1, nthreads), 4, [&](int i) {
      std::copy(slist[(size_t)i].first, slist[(size_t)i].first + slist[(size_t)i].second->doubles,
                hstage.begin() + slist[(size_t)i].second->offset);
    });
    if (dstage.alloc((size_t)stage_total) != hipSuccess ||
        hipMemcpy(dstage.p, hstage.data(), (size_t)stage_total * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) {
      set_last_error("sparse preconditioner: staging the host-side sources of the weights failed");
      return false;
    }
  }
  std::vector<DFill> df((size_t)nf);
  for (long long i = 0; i < nf; ++i) {
    const Fill &f = at(i);
    DFill &d = df[(size_t)i];
    d.off = f.off;
    d.extent = (i + 1 < nf ? at(i + 1).off : total) - f.off;
    const double *dev = (f.len > 0 && f.nrows > 0) ? mirrored(f.base) : nullptr;
    if (!dev && f.len > 0 && f.nrows > 0) dev = dstage.p + staged[f.base].offset;
    d.src = dev;
    d.c = f.c;
    d.a0 = f.a0;
    d.m = dev ? f.m : 0;
    d.cb0 = f.loc[0];
    d.ncb = f.loc[1] - f.loc[0];
  }
  if (vals.alloc((size_t)std::max<long long>(2, total)) != hipSuccess) {
    set_last_error("sparse preconditioner: no device memory for the stored weights");
    return false;
  }
  const long long head = nf ? df[0].off : std::max<long long>(2, total);  // weights in front of the first fill
  DevBuf<DFill> ddf;
  if (nf > 0 && (ddf.alloc((size_t)nf) != hipSuccess ||
                 hipMemcpy(ddf.p, df.data(), (size_t)nf * sizeof(DFill), hipMemcpyHostToDevice) != hipSuccess))
    return false;
  hipStream_t fs = nullptr;
  if (stream_acquire(device, &fs) != DCORA_OK) return false;
  bool ok = true;
  if (head > 0) ok = hipMemsetAsync(vals.p, 0, (size_t)head * sizeof(double), fs) == hipSuccess;
  if (ok && nf > 0) {
    hipLaunchKernelGGL(k_fill_weights, dim3((unsigned)((nf + 3) / 4)), dim3(256), 0, fs, nf, ddf.p, vals.p);
    ok = hipGetLastError() == hipSuccess;
  }
  ok = ok && hipStreamSynchronize(fs) == hipSuccess;
  stream_release(device, fs);
  if (!ok) set_last_error("sparse preconditioner: forming the stored weights on the device failed");
  return ok;
}

DeviceWeightSink::~DeviceWeightSink() {
  if (st) {
    (void)hipStreamSynchronize(st);
    stream_release(device, st);
  }
  for (int i = 0; i < kBuffers; ++i) {
    if (ev[i]) (void)hipEventDestroy(ev[i]);
    pin_release(pin[i]);
  }
}
bool DeviceWeightSink::begin(long long total) {
  if (hipSetDevice(device) != hipSuccess) return false;
  if (vals.alloc((size_t)std::max<long long>(2, total)) != hipSuccess) {
    set_last_error("sparse preconditioner: no device memory for the stored weights");
    return false;
  }
  if (total < 2 && hipMemset(vals.p, 0, 2 * sizeof(double)) != hipSuccess) return false;
  if (stream_acquire(device, &st) != DCORA_OK) return false;
  for (int i = 0; i < kBuffers; ++i) {
    if (hipEventCreateWithFlags(&ev[i], hipEventDisableTiming) != hipSuccess) return false;
  }
  return true;
}
double *DeviceWeightSink::acquire(long long n) {
  if (n > kChunk) return nullptr;
  cur = (cur + 1) % kBuffers;
  if (!pin[cur]) {
    pin[cur] = pin_acquire();
    if (!pin[cur]) {
      set_last_error("sparse preconditioner: no pinned host memory for the weight chunks");
      return nullptr;
    }
  }
  if (busy[cur]) {
    if (hipEventSynchronize(ev[cur]) != hipSuccess) return nullptr;
    busy[cur] = false;
  }
  return pin[cur];
}
bool DeviceWeightSink::commit(long long off, long long n) {
  if (hipMemcpyAsync(vals.p + off, pin[cur], (size_t)n * sizeof(double), hipMemcpyHostToDevice, st) != hipSuccess ||
      hipEventRecord(ev[cur], st) != hipSuccess) {
    set_last_error("sparse preconditioner: copying a chunk of stored weights to the device failed");
    return false;
  }
  busy[cur] = true;
  return true;
}
bool DeviceWeightSink::end() { return hipStreamSynchronize(st) == hipSuccess; }

int SpImage::upload(const PartInvHost &P, DeviceWeightSink *streamed) {
  k = P.k;
  levels = P.levels;
  nnzL = P.nnzL;
  npieces = P.npieces;
  weights_per_apply = P.weights_read_per_apply;
  // (the matrix-pipe kernel may read up to three groups past the end of a matrix / an index list: P.vals and P.idxs end
  // with that much padding, host_partinv3.cpp)
  if (streamed && streamed->vals.p) {
    vals = std::move(streamed->vals);
  } else {
    DCORA_HIP(vals.alloc(P.vals.size()));
    DCORA_HIP(hipMemcpy(vals.p, P.vals.data(), P.vals.size() * sizeof(double), hipMemcpyHostToDevice));
  }
  DCORA_HIP(idxs.alloc(P.idxs.size()));
  DCORA_HIP(hipMemcpy(idxs.p, P.idxs.data(), P.idxs.size() * sizeof(int), hipMemcpyHostToDevice));
  DCORA_HIP(mwaves.alloc(std::max<size_t>(P.mwaves.size(), 1)));
  if (!P.mwaves.empty())
    DCORA_HIP(hipMemcpy(mwaves.p, P.mwaves.data(), P.mwaves.size() * sizeof(MWave), hipMemcpyHostToDevice));
  DCORA_HIP(perm.alloc(P.perm.size()));
  DCORA_HIP(hipMemcpy(perm.p, P.perm.data(), P.perm.size() * sizeof(int), hipMemcpyHostToDevice));
  DCORA_HIP(out_off.alloc(P.out_off.size()));
  DCORA_HIP(hipMemcpy(out_off.p, P.out_off.data(), P.out_off.size() * sizeof(int), hipMemcpyHostToDevice));
  rows_total = 0;
  nmwaves_total = (long)P.mwaves.size();
  for (const SpLevel &lv : P.levels)
    for (int q = 0; q < lv.ntasks * kMtWaves; ++q) {
      const MWave &t = P.mwaves[(size_t)lv.task0 + q];
      if (t.nrows > 0 && t.red_first == q % kMtWaves) rows_total += t.nrows;
    }
  nhub = P.hub.h;
  hub_nnz = (long)P.hub.aval.size();
  {
    // original unknown -> position in image 0 / position of the final value; -1 on a hub unknown
    std::vector<int> ip((size_t)P.kfull, -1), op((size_t)P.kfull, -1);
    for (int j = 0; j < k; ++j) {
      ip[(size_t)P.perm[j]] = j;
      op[(size_t)P.perm[j]] = P.out_off[j];
    }
    DCORA_HIP(in_pos.alloc(ip.size()));
    DCORA_HIP(hipMemcpy(in_pos.p, ip.data(), ip.size() * sizeof(int), hipMemcpyHostToDevice));
    DCORA_HIP(out_pos.alloc(op.size()));
    DCORA_HIP(hipMemcpy(out_pos.p, op.data(), op.size() * sizeof(int), hipMemcpyHostToDevice));
  }
  if (nhub > 0) {
    const PartInvHub &H = P.hub;
    DCORA_HIP(hub_idx.alloc(H.idx.size()));
    DCORA_HIP(hipMemcpy(hub_idx.p, H.idx.data(), H.idx.size() * sizeof(int), hipMemcpyHostToDevice));
    DCORA_HIP(hub_U.alloc(H.U.size()));
    DCORA_HIP(hipMemcpy(hub_U.p, H.U.data(), H.U.size() * sizeof(double), hipMemcpyHostToDevice));
    DCORA_HIP(hub_Sinv.alloc(H.Sinv.size()));
    DCORA_HIP(hipMemcpy(hub_Sinv.p, H.Sinv.data(), H.Sinv.size() * sizeof(double), hipMemcpyHostToDevice));
  }
  return DCORA_OK;
}

size_t SpImage::device_bytes() const {
  return vals.n * sizeof(double) + (idxs.n + perm.n + out_off.n + in_pos.n + out_pos.n) * sizeof(int) +
         mwaves.n * sizeof(MWave) +
         (hub_U.n + hub_Sinv.n) * sizeof(double) + hub_idx.n * sizeof(int);
}

int SparsePrecond::attach(std::shared_ptr<const SpImage> image, int rcap_) {
  im = std::move(image);
  rcap = rcap_;
  weights_per_apply = im->weights_per_apply;
  const int k = im->k;
  // two ping-pong images of the vector; padded pairs / K steps may touch up to three unknowns past the end
  // (the matrix-pipe kernel: a step past the end of a run of the vector, 16 unknowns, and seven values past a pair)
  DCORA_HIP(y.alloc((size_t)(2 * k + 20) * rcap + 8));
  DCORA_HIP(hipMemset(y.p, 0, ((size_t)(2 * k + 20) * rcap + 8) * sizeof(double)));
  if (im->nhub > 0) {
    DCORA_HIP(hub_w.alloc((size_t)im->nhub * kHubSplit * rcap));
    DCORA_HIP(hipMemset(hub_w.p, 0, (size_t)im->nhub * kHubSplit * rcap * sizeof(double)));
    DCORA_HIP(hub_x2.alloc((size_t)im->nhub * rcap));
    DCORA_HIP(hipMemset(hub_x2.p, 0, (size_t)im->nhub * rcap * sizeof(double)));
  }
  return DCORA_OK;
}

void SparsePrecond::apply(hipStream_t st, int r, Buf2 R, double *Z, Gate g, bool levels_only,
                          const std::function<bool()> *after_first) const {
  const SpImage &I = *im;
  const int k = I.k, nhub = I.nhub;
  const DevBuf<int> &perm = I.perm, &out_off = I.out_off, &hub_idx = I.hub_idx, &idxs = I.idxs;
  const DevBuf<double> &vals = I.vals, &hub_U = I.hub_U, &hub_Sinv = I.hub_Sinv;
  const DevBuf<double> &hub_x2 = this->hub_x2;
  const std::vector<SpLevel> &levels = I.levels;
  const long n = (long)r * k;
  const int grid = (int)std::min<long>((n + kBlock - 1) / kBlock, 2048);
  levels_only = levels_only && im->in_pos.p != nullptr;
  bool asked = false;
  auto go_on = [&]() {
    if (asked || !after_first) return true;
    asked = true;
    return (*after_first)();
  };
  if (!levels_only) {
    hipLaunchKernelGGL(k_sp_permute_in, dim3(grid), dim3(kBlock), 0, st, r, k, perm.p, R, y.p, g);
    if (!go_on()) return;
  }
  // hubs: stage 1 (the slices of U^T r1 over the input image) rides in the first launch, stage 2 (x2 from the slices)
  // in the second; a stage that finds no launch gets one of its own
  HubIn hin;
  if (nhub > 0) {
    hin.h = nhub;
    hin.k = k;
    hin.U = hub_U.p;
    hin.w = hub_w.p;
    hin.idx = hub_idx.p;
    hin.Sinv = hub_Sinv.p;
    hin.R = R;
    hin.x2 = hub_x2.p;
  }
  auto stage = [&](int s) {
    HubIn hs = hin;
    hs.stage = s;
    return hs;
  };
  if (levels.empty() && nhub > 0)
    hipLaunchKernelGGL(k_sp_hub_stage, dim3(nhub * kHubSplit), dim3(kBlock), 0, st, r, stage(1), y.p, g);
  int li = 0;
  for (const SpLevel &lv : levels) {
    launch_mtile(st, r, lv, I.mwaves.p, vals.p, idxs.p, y.p, g, (nhub > 0 && li < 2) ? stage(li + 1) : HubIn());
    ++li;
    if (!go_on()) return;
  }
  if (nhub > 0 && levels.size() < 2) hipLaunchKernelGGL(k_sp_hub_stage, dim3(1), dim3(kBlock), 0, st, r, stage(2), y.p, g);
  if (levels_only) return;  // the caller's kernel applies the hub correction while it reads y (fold_generic())
  if (nhub > 0) {
    HubDev H{nhub, hub_idx.p, hub_U.p, hub_x2.p};
    hipLaunchKernelGGL(k_sp_permute_out_hub, dim3(grid), dim3(kBlock), 0, st, r, k, perm.p, out_off.p, y.p, Z, H, g);
  } else {
    hipLaunchKernelGGL(k_sp_permute_out, dim3(grid), dim3(kBlock), 0, st, r, k, perm.p, out_off.p, y.p, Z, g);
  }
}

SpFold SparsePrecond::fold_generic() const {
  SpFold f;
  if (!im || !im->in_pos.p) return f;
  f.y = y.p;
  f.in_pos = im->in_pos.p;
  f.out_pos = im->out_pos.p;
  f.h = im->nhub;
  f.hub_idx = im->hub_idx.p;
  f.hub_U = im->hub_U.p;
  f.hub_x2 = hub_x2.p;
  return f;
}

double SparsePrecond::bytes_per_apply(int r) const {
  // the weights a solve streams (W twice, M once), the wave records, the vector in and out of every tile, permutes,
  // hub terms
  const SpImage &I = *im;
  return 8.0 * I.weights_per_apply + 128.0 * I.nmwaves_total +
         16.0 * r * I.rows_total +
         32.0 * r * (double)I.k + 16.0 * (double)I.nhub * I.k + 8.0 * r * (double)I.nhub * I.k;
}

}  // namespace dcora
