// see device_chol.h
#include "device_chol.h"
#include "env.h"

#include <sys/mman.h>

#include <hip/hip_runtime.h>

#include <chrono>
#include <unistd.h>
#include <cstdlib>
#include <cstring>
#include <list>
#include <mutex>

#include "device_problem.h"
#include "sparse_precond.h"

namespace dcora {

namespace {

constexpr int NB = kCholNb;
constexpr int kEaRows = 8;  // rows of a child's Schur complement per workgroup of the extend-add

// one piece of a batched inversion (blockIdx.z): offsets into the arrays the kernel is given
struct InvJob {
  int c, m;
  long long l11, f;    // L11 inside the front arena, leading dimension of the front
  long long yt, tinv;  // (L11^-1)^T (c x c) and the inverses of its 64 x 64 diagonal blocks
  long long pk, mt;    // the packed panel; M = L11^-T L11^-1 (c x c) or -1
};

struct PieceDev {
  long long off;
  int c, m;
  int parent, rel_off;
};

// F[dest[p]] = v[p] for the entries of the lower triangle
__global__ __launch_bounds__(256) void k_chol_scatter(long long nnz, const long long *__restrict__ dest,
                                                      const double *__restrict__ v, double *__restrict__ F) {
  const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  if (p >= nnz) return;
  const long long q = dest[p];
  if (q >= 0) F[q] = v[p];
}

// parent front += Schur complement of one child (children of one slot have distinct parents: no two workgroups of a
// launch touch the same front entry)
__global__ __launch_bounds__(256) void k_chol_extend_add(const PieceDev *__restrict__ pieces,
                                                         const int *__restrict__ children,
                                                         const int *__restrict__ rel, double *__restrict__ F,
                                                         const int *__restrict__ fail) {
  if (*fail) return;
  const PieceDev D = pieces[children[blockIdx.y]];
  const int a0 = blockIdx.x * kEaRows;
  if (a0 >= D.m) return;
  const PieceDev P = pieces[D.parent];
  const long long fd = (long long)D.c + D.m, fp = (long long)P.c + P.m;
  const int *__restrict__ rl = rel + D.rel_off;
  const double *__restrict__ U = F + D.off + (long long)D.c * fd + D.c;
  double *__restrict__ T = F + P.off;
  const int a1 = min(D.m, a0 + kEaRows);
  for (int a = a0; a < a1; ++a) {
    const long long ra = rl[a];
    for (int b = threadIdx.x; b <= a; b += 256) T[ra * fp + rl[b]] += U[(long long)a * fd + b];
  }
}

// ---- 64 x 64 Cholesky and triangular inverse of a workgroup, blocked by 16 (round 3).  The column-by-column form of
//      round 2 (deleted) paid a workgroup barrier per column and a 64-step dependent walk per column of the inverse:
//      83 us per diagonal block, 13 of the 21 ms of the headline's agent set-up and half of
//      sphere2500's PSD test.  Blocked: the 16 x 16 diagonal block is factored by ONE wave (LDS operations of a wave
//      stay in order: no workgroup barrier inside its 16 steps), the panel below it is a 16-step substitution per row,
//      the trailing update a rank-16 product over all threads; the inverse is four 16 x 16 inversions side by side
//      (one per wave) and three rounds of 16 x 16 block products. ----
// L: lower triangle in LDS, identity beyond the matrix' order; returns false (uniformly) on a non-positive pivot
// the value lane K of every 16-lane row holds, in all lanes of that row: DPP row_share (two 32-bit moves; a v_readlane
// goes through an SGPR and pays the VALU -> SGPR -> VALU hazard on every use)
template <int K>
__device__ __forceinline__ double row_share_k(double v) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)b, 0x150 + K, 0xF, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), 0x150 + K, 0xF, 0xF, false);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double row_share(double v, int k) {  // k: a constant once the caller's loops are unrolled
  switch (k & 15) {
    case 0: return row_share_k<0>(v);
    case 1: return row_share_k<1>(v);
    case 2: return row_share_k<2>(v);
    case 3: return row_share_k<3>(v);
    case 4: return row_share_k<4>(v);
    case 5: return row_share_k<5>(v);
    case 6: return row_share_k<6>(v);
    case 7: return row_share_k<7>(v);
    case 8: return row_share_k<8>(v);
    case 9: return row_share_k<9>(v);
    case 10: return row_share_k<10>(v);
    case 11: return row_share_k<11>(v);
    case 12: return row_share_k<12>(v);
    case 13: return row_share_k<13>(v);
    case 14: return row_share_k<14>(v);
    default: return row_share_k<15>(v);
  }
}
// 1 / sqrt(d) for a positive pivot: hardware estimate + two Newton steps (1-2 ulp, a dozen dependent instructions; the
// IEEE division and square root are chains of about fifty each and sat on the critical path of every pivot)
__device__ __forceinline__ double fast_rsqrt(double d) {
  double y = __builtin_amdgcn_rsq(d);
  y = y * fma(-0.5 * d * y, y, 1.5);
  y = y * fma(-0.5 * d * y, y, 1.5);
  return y;
}
// 16 x 16 x 16 product on the matrix pipe (v_mfma_f64_16x16x4_f64, four steps): D += A B with A(i, k) = fa(i, k) and
// B(k, j) = fb(k, j); at step s lane l feeds A(l & 15, 4 s + (l >> 4)) and B(4 s + (l >> 4), l & 15) and it receives
// D((l >> 4) + 4 v, l & 15) in d[v].  A result can be the B operand of the next product as it stands: step s wants rows
// 4 s + (l >> 4) of B, which is d[s].
typedef double v4f64 __attribute__((ext_vector_type(4)));
template <class FA, class FB>
__device__ __forceinline__ v4f64 mma16(v4f64 d, FA fa, FB fb) {
  const int lane = threadIdx.x & 63, lo = lane & 15, hi = lane >> 4;
#pragma unroll
  for (int s = 0; s < 4; ++s) d = __builtin_amdgcn_mfma_f64_16x16x4f64(fa(lo, 4 * s + hi), fb(4 * s + hi, lo), d, 0, 0, 0);
  return d;
}
__device__ __forceinline__ bool blocked_potrf64(double (*L)[NB + 1], int *s_bad, double *rd, double (*Dv)[16][17]) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  if (tid == 0) *s_bad = 0;
  __syncthreads();
#pragma unroll 1
  for (int j0 = 0; j0 < NB; j0 += 16) {
    if (wave == 0) {
      // L D L^T elimination of the 16 x 16 block IN REGISTERS: lane i of every 16-lane row holds row i, the pivot and
      // the entries of column j travel by DPP row_share (the loops are unrolled: every register and lane index is a
      // constant) -- no LDS round trip, no barrier and no IEEE division per column.  The columns keep l_ij d_j until
      // the end, then the scaling to L L^T.  All four rows of the wave compute the same thing; the first stores.
      const int i = lane & 15;
      double x[16], rdv[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) x[k] = (k <= i) ? L[j0 + i][j0 + k] : 0.0;
      bool bad = false;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const double d = row_share(x[j], j);
        bad = bad || !(d > 0.0);  // uniform
        rdv[j] = fast_rsqrt(d);
        const double lij = x[j] * (rdv[j] * rdv[j]);
#pragma unroll
        for (int k = j + 1; k < 16; ++k) x[k] -= lij * row_share(x[j], k);
      }
      if (bad) {
        if (lane == 0) *s_bad = 1;
      } else {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
          // row i of the Cholesky factor of the block: l_ik = (l_ik d_k) / sqrt(d_k), l_ii = d_i / sqrt(d_i)
          x[k] = (k <= i) ? x[k] * rdv[k] : 0.0;
          if (lane == 0) rd[j0 + k] = rdv[k];
          if (k <= i && lane < 16) L[j0 + i][j0 + k] = x[k];
        }
        // the inverse of the block, column `i` by forward substitution in registers; factor(r_, l) is row r_ of lane r_.
        // Kept for the panel below and for blocked_trtri64.
        double y[16];
#pragma unroll
        for (int r_ = 0; r_ < 16; ++r_) {
          double sacc = 0;
#pragma unroll
          for (int l = 0; l < 16; ++l)
            if (l < r_) {
              const double lrl = row_share(x[l], r_);
              sacc += (l >= i) ? lrl * y[l] : 0.0;
            }
          y[r_] = (r_ == i) ? rdv[r_] : (r_ > i ? -sacc * rdv[r_] : 0.0);
        }
#pragma unroll
        for (int r_ = 0; r_ < 16; ++r_)
          if (lane < 16) Dv[j0 >> 4][r_][i] = y[r_];
      }
    }
    __syncthreads();
    if (*s_bad) return false;
    const int nrem = NB - j0 - 16;
    if (nrem > 0) {
      // Round 5: panel and trailing update on the matrix pipe, a 16 x 16 block per wave (the FMA forms of round 4 -- four
      // threads per panel row, a 4 x 4 block of the update per thread -- sat on the LDS port: 6.7 us of the 45 us launch
      // were the update, 14.6 the scalar LDS loops of update + inverse).
      const int lo = lane & 15, hi = lane >> 4, r0 = j0 + 16, nbk = nrem >> 4;
      // panel: rows below times the transposed inverse of the block, X(a, k) = sum_{l <= k} B(a, l) Dinv(k, l); a wave
      // reads only the 16 rows it writes, and writes them after its products
      if (wave < nbk) {
        const int rw = r0 + 16 * wave, bq = j0 >> 4;
        const v4f64 d = mma16(
            v4f64{0, 0, 0, 0}, [&](int i, int k) { return L[rw + i][j0 + k]; },
            [&](int k, int j) { return k <= j ? Dv[bq][j][k] : 0.0; });
#pragma unroll
        for (int v = 0; v < 4; ++v) L[rw + hi + 4 * v][j0 + lo] = d[v];
      }
      __syncthreads();
      // trailing update of the lower triangle by the rank-16 product of the panel with itself: the 6 / 3 / 1 lower blocks
      // of the three steps dealt to the four waves (they read columns j0 .. j0 + 15 and write columns beyond)
      for (int tb = wave; tb < nbk * (nbk + 1) / 2; tb += 4) {
        const int bi = tb < 1 ? 0 : (tb < 3 ? 1 : 2), bj = tb - bi * (bi + 1) / 2;
        const int ia = r0 + 16 * bi, ib = r0 + 16 * bj;
        const v4f64 d = mma16(
            v4f64{0, 0, 0, 0}, [&](int i, int k) { return L[ia + i][j0 + k]; },
            [&](int k, int j) { return L[ib + j][j0 + k]; });
#pragma unroll
        for (int v = 0; v < 4; ++v)
          if (ib + lo <= ia + hi + 4 * v) L[ia + hi + 4 * v][ib + lo] -= d[v];
      }
      __syncthreads();
    }
  }
  return true;
}
// Li = L^-1 (both lower triangular in LDS)
__device__ __forceinline__ void blocked_trtri64(double (*L)[NB + 1], double (*Li)[NB + 1],
                                                const double (*Dv)[16][17] /* inverses of the diagonal blocks */) {
  const int tid = threadIdx.x;
  for (int e = tid; e < NB * NB; e += 256) {
    const int r_ = e >> 6, c_ = e & 63;
    if (c_ > r_ || (r_ >> 4) != (c_ >> 4)) Li[r_][c_] = 0.0;  // above the diagonal and the off-diagonal blocks
  }
  // the diagonal blocks' inverses come from the factorisation (blocked_potrf64)
  for (int e = tid; e < 4 * 256; e += 256) {
    const int bq = e >> 8, a = (e >> 4) & 15, c_ = e & 15;
    if (c_ <= a) Li[16 * bq + a][16 * bq + c_] = Dv[bq][a][c_];
  }
  __syncthreads();
  // block (bi, bj), bi - bj = dist:  Li_ij = -Li_ii (sum_{bk = bj}^{bi - 1} L_i,bk Li_bk,j): one wave per block, both
  // products on the matrix pipe; the inner sum stays in registers (a result is the next product's B operand as it
  // stands), so a distance costs one barrier where the FMA form of round 4 paid two and 32 LDS reads per output
  const int wave = tid >> 6, lo = tid & 15, hi = (tid & 63) >> 4;
  for (int dist = 1; dist < 4; ++dist) {
    if (wave < 4 - dist) {
      const int bj = wave, bi = wave + dist;
      v4f64 t{0, 0, 0, 0};
      for (int bk = bj; bk < bi; ++bk)
        t = mma16(
            t, [&](int i, int k) { return L[16 * bi + i][16 * bk + k]; },
            [&](int k, int j) { return Li[16 * bk + k][16 * bj + j]; });
      v4f64 d{0, 0, 0, 0};
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int k = 4 * s + hi;
        d = __builtin_amdgcn_mfma_f64_16x16x4f64(k <= lo ? Li[16 * bi + lo][16 * bi + k] : 0.0, t[s], d, 0, 0, 0);
      }
#pragma unroll
      for (int v = 0; v < 4; ++v) Li[16 * bi + hi + 4 * v][16 * bj + lo] = -d[v];
    }
    __syncthreads();
  }
}

// diagonal block of the current panel, blocked form (see above): LL^T of up to 64 columns, written back in place, and
// L^-1 for the panel below; a pivot that is not positive raises *fail
__global__ __launch_bounds__(256) void k_chol_potrf(const PieceDev *__restrict__ pieces, const int *__restrict__ list,
                                                    int j0, double *__restrict__ F, double *__restrict__ Linv,
                                                    int *__restrict__ fail, double *__restrict__ logdet,
                                                    int always_inv) {
  if (*fail) return;
  const PieceDev P = pieces[list[blockIdx.x]];
  const int jb = min(NB, P.c - j0);
  const long long f = (long long)P.c + P.m;
  double *__restrict__ M = F + P.off + (long long)j0 * f + j0;
  __shared__ double L[NB][NB + 1];
  __shared__ double Li[NB][NB + 1];
  __shared__ double rd[NB];
  __shared__ double Dv[4][16][17];
  __shared__ int s_bad;
  const int tid = threadIdx.x;
  for (int e = tid; e < NB * NB; e += 256) {
    const int i = e >> 6, j = e & 63;
    L[i][j] = (i < jb && j <= i) ? M[(long long)i * f + j] : (i == j ? 1.0 : 0.0);
  }
  if (!blocked_potrf64(L, &s_bad, rd, Dv)) {
    if (tid == 0) *fail = 1;
    return;
  }
  if (tid < NB) {
    // log det of the factored matrix (a cross-check of the whole factorisation for the tests; order of the sum free)
    double lg = tid < jb ? 2.0 * log(L[tid][tid]) : 0.0;
    for (int o = 32; o > 0; o >>= 1) lg += __shfl_xor(lg, o);
    if (tid == 0) atomicAdd(logdet, lg);
  }
  for (int e = tid; e < jb * jb; e += 256) {
    const int r_ = e / jb, j = e - r_ * jb;
    if (j <= r_) M[(long long)r_ * f + j] = L[r_][j];
  }
  if (!always_inv && P.m == 0 && j0 + jb >= P.c) return;  // nothing below the last panel of a root
  blocked_trtri64(L, Li, Dv);
  double *__restrict__ O = Linv + (size_t)blockIdx.x * NB * NB;
  for (int e = tid; e < NB * NB; e += 256) O[e] = Li[e >> 6][e & 63];
}



// ---- the 64 x 64 x 64 product of two LDS tiles, As[k][i] and Bs[k][j]: acc(i, j) += sum_k As[k][i] Bs[k][j] ----
// MMA = false: 4 x 4 outputs per thread by FMAs (8 LDS reads per 16 FMAs: the LDS port is the bound, 128 clocks per k
// for the workgroup against 64 of FMA).  MMA = true: v_mfma_f64_16x16x4_f64, every wave owns a 32 x 32 quadrant as
// 2 x 2 blocks; 4 LDS reads per 4 MFMAs (8192 flop), the matrix pipe is the bound.  The two forms keep their 16 sums
// per thread in acc[4][4] under different maps (acc_row / acc_col).
template <bool MMA>
__device__ __forceinline__ int acc_row(int u, int v) {
  if (MMA) return (int)(threadIdx.x >> 7) * 32 + (u >> 1) * 16 + (int)((threadIdx.x & 63) >> 4) + 4 * v;
  return (int)(threadIdx.x >> 4) + 16 * u;
}
template <bool MMA>
__device__ __forceinline__ int acc_col(int u, int v) {
  if (MMA) return (int)((threadIdx.x >> 6) & 1) * 32 + (u & 1) * 16 + (int)(threadIdx.x & 15);
  return (int)(threadIdx.x & 15) + 16 * v;
}
template <bool MMA>
__device__ __forceinline__ void tile_inner(const double (*As)[NB + 1], const double (*Bs)[NB + 1], double acc[4][4]) {
  if (MMA) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int r0 = (w >> 1) * 32 + (lane & 15), c0 = (w & 1) * 32 + (lane & 15), kq = lane >> 4;
    v4f64 c[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) c[u] = v4f64{acc[u][0], acc[u][1], acc[u][2], acc[u][3]};
#pragma unroll 4
    for (int k0 = 0; k0 < NB; k0 += 4) {
      const double a0 = As[k0 + kq][r0], a1 = As[k0 + kq][r0 + 16];
      const double b0 = Bs[k0 + kq][c0], b1 = Bs[k0 + kq][c0 + 16];
      c[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, c[0], 0, 0, 0);
      c[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, c[1], 0, 0, 0);
      c[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, c[2], 0, 0, 0);
      c[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, c[3], 0, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int v = 0; v < 4; ++v) acc[u][v] = c[u][v];
  } else {
    const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
#pragma unroll 8
    for (int k = 0; k < NB; ++k) {
      double a[4], b[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) a[u] = As[k][ty + 16 * u];
#pragma unroll
      for (int v = 0; v < 4; ++v) b[v] = Bs[k][tx + 16 * v];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) acc[u][v] = fma(a[u], b[v], acc[u][v]);
    }
  }
}
constexpr int chol_superpanel_blocks() { return 8; }  // measured: 4 -> 157 ms, 8 -> 99 ms (1: a trailing update per block, 217)
// the tile products run on the matrix pipe (MMA = true); the 4 x 4-per-thread FMA form of the same templates measured
// 126 ms against 98 for the factorisation of the 100k lattice and is not instantiated
#define DCORA_LAUNCH_MMA(KERNEL, grid, st, ...) hipLaunchKernelGGL((KERNEL<true>), grid, dim3(256), 0, st, __VA_ARGS__)

// acc(i, j) = sum_k A[i][k] B[j][k]; A and B are row-major with k contiguous; rows beyond arows / brows and k beyond K
// read as zero
template <bool MMA>
__device__ __forceinline__ void chol_tile_product(const double *__restrict__ Ag, long long lda, int arows,
                                                  const double *__restrict__ Bg, long long ldb, int brows, int K,
                                                  double (*As)[NB + 1], double (*Bs)[NB + 1], double acc[4][4]) {
  const int tid = threadIdx.x;
  for (int e = tid; e < NB * NB; e += 256) {
    const int i = e >> 6, k = e & 63;
    As[k][i] = (i < arows && k < K) ? Ag[(long long)i * lda + k] : 0.0;
    Bs[k][i] = (i < brows && k < K) ? Bg[(long long)i * ldb + k] : 0.0;
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int v = 0; v < 4; ++v) acc[u][v] = 0;
  tile_inner<MMA>(As, Bs, acc);
}

// panel below the diagonal block:  X = B L^-T, 64 rows per workgroup, in place
template <bool MMA>
__global__ __launch_bounds__(256) void k_chol_trsm(const PieceDev *__restrict__ pieces, const int *__restrict__ list,
                                                   int j0, double *__restrict__ F, const double *__restrict__ Linv,
                                                   const int *__restrict__ fail) {
  if (*fail) return;
  const PieceDev P = pieces[list[blockIdx.y]];
  const int jb = min(NB, P.c - j0);
  const int f = P.c + P.m;
  const int r0 = j0 + jb + NB * blockIdx.x;
  if (r0 >= f) return;
  __shared__ double As[NB][NB + 1];
  __shared__ double Bs[NB][NB + 1];
  double *__restrict__ A = F + P.off + (long long)r0 * f + j0;
  double acc[4][4];
  chol_tile_product<MMA>(A, f, min(NB, f - r0), Linv + (size_t)blockIdx.y * NB * NB, NB, jb, jb, As, Bs, acc);
  __syncthreads();  // the tile is overwritten in place: every read of it (into LDS) is behind us
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int i = acc_row<MMA>(u, v), j = acc_col<MMA>(u, v);
      if (r0 + i < f && j < jb) A[(long long)i * f + j] = acc[u][v];
    }
}

__global__ __launch_bounds__(256) void k_count_nonzero(long long n, const double *__restrict__ v,
                                                       unsigned long long *__restrict__ out) {
  unsigned long long c = 0;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) c += v[i] != 0.0;
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, c);
}

// the first c columns of every front (diagonal block over the rows below), packed piece after piece
__global__ __launch_bounds__(256) void k_chol_pack(const PieceDev *__restrict__ pieces,
                                                   const long long *__restrict__ panel_off,
                                                   const double *__restrict__ F, double *__restrict__ out) {
  const PieceDev P = pieces[blockIdx.y];
  const int f = P.c + P.m, c = P.c;
  const long long n = (long long)f * c;
  const double *__restrict__ M = F + P.off;
  double *__restrict__ O = out + panel_off[blockIdx.y];
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) {
    const int i = (int)(e / c), j = (int)(e - (long long)i * c);
    O[e] = (i >= c || j <= i) ? M[(long long)i * f + j] : 0.0;
  }
}

// ---------------------------------------------------------------------------------------------------------
// Dense inverse of a small SPD matrix on the device (the dense preconditioner of blocks up to 8000 unknowns,
// ref src/QuadraticProblem.cpp:70-84 applied as an explicit inverse): A = L L^T by the panel kernels above (the whole
// matrix is one front), Y = L^-1 right-looking by block rows, A^-1 = Y^T Y.  Y is kept transposed (YT, upper block
// triangular) so that every product below runs over contiguous rows of both operands.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_dense_scatter(int k, const int *__restrict__ rp, const int *__restrict__ ci,
                                                       const double *__restrict__ v, double *__restrict__ A) {
  const int i = blockIdx.x;
  for (int p = rp[i] + threadIdx.x; p < rp[i + 1]; p += 256) {
    const int j = ci[p];
    if (j <= i) A[(size_t)i * k + j] = v[p];
  }
}

// acc += A-rows x B-rows^T over K chunks of 64: A(ia, l) and B(ib, l), l in [l0, l1)
template <bool MMA>
__device__ __forceinline__ void tile_product_range(const double *__restrict__ Arow, const double *__restrict__ Brow,
                                                   long long ld, int arows, int brows, int l0, int l1,
                                                   double (*As)[NB + 1], double (*Bs)[NB + 1], double acc[4][4],
                                                   long long ldb = -1) {
  if (ldb < 0) ldb = ld;
  const int tid = threadIdx.x;
  // software pipeline over the 64-column steps: the operands of step l + 1 are requested (into registers) before the
  // products of step l, so their latency hides behind 64 MFMAs per wave instead of standing between two steps
  constexpr int PER = NB * NB / 256;  // entries of each operand per thread
  const int kk = tid & 63, ib = tid >> 6;  // entry q of a thread: row ib + 4 q, column kk of the step
  double ra[PER], rb[PER];
  auto fetch = [&](int l) {
    const int K = min(NB, l1 - l);
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const int i = ib + 4 * q;
      ra[q] = (i < arows && kk < K) ? Arow[(long long)i * ld + l + kk] : 0.0;
      rb[q] = (i < brows && kk < K) ? Brow[(long long)i * ldb + l + kk] : 0.0;
    }
  };
  if (l0 < l1) fetch(l0);
  for (int l = l0; l < l1; l += NB) {
    __syncthreads();
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      As[kk][ib + 4 * q] = ra[q];
      Bs[kk][ib + 4 * q] = rb[q];
    }
    __syncthreads();
    if (l + NB < l1) fetch(l + NB);
    tile_inner<MMA>(As, Bs, acc);
  }
}

// trailing update: C(ti, tj) -= sum_{l in [k0, base)} X(ti, l) X(tj, l) over 64 x 64 tiles of the lower triangle behind
// base = min(kcap, c).  Two uses (build_image):
//   inside a super-panel (ncb > 0): K = one 64-column block, only the ncb column blocks of the super-panel that are
//     still to be factored are updated (columns < min(ccap, c)), blockIdx.x = ti ncb + tj;
//   behind a super-panel (ncb = 0): K = the whole super-panel (up to 256 columns in 64-column steps through LDS), every
//     tile of the trailing matrix, blockIdx.x = triangular index.  A rank-256 update reads and writes the trailing
//     matrix once where four rank-64 updates did four times: the update is bound by exactly that traffic.
template <bool MMA>
__global__ __launch_bounds__(256) void k_chol_syrk(const PieceDev *__restrict__ pieces, const int *__restrict__ list,
                                                   int k0, int kcap, int ccap, int ncb, double *__restrict__ F,
                                                   const int *__restrict__ fail) {
  if (*fail) return;
  const PieceDev P = pieces[list[blockIdx.y]];
  const int f = P.c + P.m;
  const int base = min(kcap, P.c);
  if (base <= k0) return;
  const int colend = ccap >= 0 ? min(ccap, P.c) : f;
  int ti, tj;
  if (ncb > 0) {
    ti = blockIdx.x / ncb;
    tj = blockIdx.x - ti * ncb;
    if (tj > ti) return;
  } else {
    ti = (int)((sqrtf(8.0f * (float)blockIdx.x + 1.0f) - 1.0f) * 0.5f);
    while ((ti + 1) * (ti + 2) / 2 <= (int)blockIdx.x) ++ti;
    while (ti * (ti + 1) / 2 > (int)blockIdx.x) --ti;
    tj = blockIdx.x - ti * (ti + 1) / 2;
  }
  const int ri0 = base + NB * ti, cj0 = base + NB * tj;
  if (ri0 >= f || cj0 >= colend) return;
  __shared__ double As[NB][NB + 1];
  __shared__ double Bs[NB][NB + 1];
  double *__restrict__ M = F + P.off;
  double acc[4][4];
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int v = 0; v < 4; ++v) acc[u][v] = 0;
  tile_product_range<MMA>(M + (long long)ri0 * f, M + (long long)cj0 * f, f, min(NB, f - ri0), min(NB, f - cj0), k0, base,
                          As, Bs, acc);
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int r = ri0 + acc_row<MMA>(u, v), c = cj0 + acc_col<MMA>(u, v);
      if (r < f && c <= r && c < colend) M[(long long)r * f + c] -= acc[u][v];
    }
}


// Y = L^-1 right-looking, one block row of L at a time; Y is kept transposed (YT(j, i) = Y(i, j)^T).  Step ib:
//   finish:  YT(j, ib) = -TT(j, ib) Linv_ib^T for the block rows j < ib (TT accumulated in place), YT(ib, ib) = Linv_ib^T
//   update:  TT(j, i) += YT(j, ib) L(i, ib)^T for every block row i > ib and j <= ib  -- (nb - ib - 1)(ib + 1) tiles
template <bool MMA>
__global__ __launch_bounds__(256) void k_dense_trtri_finish(int k, int ib, double *__restrict__ YT,
                                                            const double *__restrict__ Linv,
                                                            const InvJob *__restrict__ jobs = nullptr,
                                                            int tstride = 1) {
  if (jobs) {
    const InvJob J = jobs[blockIdx.z];
    k = J.c;
    YT += J.yt;
    Linv += J.tinv;
    if (ib * NB >= k) return;
  }
  __shared__ double As[NB][NB + 1];
  __shared__ double Bs[NB][NB + 1];
  const int tid = threadIdx.x;
  const int i0 = ib * NB, ni = min(NB, k - i0);
  // (tstride: 64 x 64 blocks between the inverses of consecutive panels -- 1, or the batch size when the potrf of a batch
  // of equal matrices stored them panel by panel)
  const double *__restrict__ Li = Linv + (size_t)ib * NB * NB * tstride;
  if ((int)blockIdx.x == ib) {  // diagonal block: the transpose of the inverse of the diagonal block of L
    for (int e = tid; e < NB * NB; e += 256) {
      const int a = e >> 6, b = e & 63;
      if (a < ni && b < ni) YT[(size_t)(i0 + a) * k + i0 + b] = Li[b * NB + a];
    }
    return;
  }
  const int j0 = blockIdx.x * NB;
  for (int e = tid; e < NB * NB; e += 256) {
    const int a = e >> 6, mm = e & 63;
    As[mm][a] = (mm < ni) ? YT[(size_t)(j0 + a) * k + i0 + mm] : 0.0;  // TT(a, m)
    Bs[mm][a] = Li[a * NB + mm];                                       // Linv(b = a, m)
  }
  __syncthreads();
  double acc[4][4];
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int v = 0; v < 4; ++v) acc[u][v] = 0;
  tile_inner<MMA>(As, Bs, acc);
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int b = acc_col<MMA>(u, v);
      if (b < ni) YT[(size_t)(j0 + acc_row<MMA>(u, v)) * k + i0 + b] = -acc[u][v];
    }
}
template <bool MMA>
__global__ __launch_bounds__(256) void k_dense_trtri_update(int k, int ib, const double *__restrict__ L, long long ldl,
                                                            double *__restrict__ YT,
                                                            const InvJob *__restrict__ jobs = nullptr) {
  if (jobs) {
    const InvJob J = jobs[blockIdx.z];
    k = J.c;
    L += J.l11;
    ldl = J.f;
    YT += J.yt;
    if ((ib + 1 + (int)blockIdx.x) * NB >= k) return;
  }
  __shared__ double As[NB][NB + 1];
  __shared__ double Bs[NB][NB + 1];
  const int i0 = (ib + 1 + blockIdx.x) * NB, j0 = blockIdx.y * NB, l0 = ib * NB;
  const int ni = min(NB, k - i0);
  double acc[4][4];
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int v = 0; v < 4; ++v) acc[u][v] = 0;
  tile_product_range<MMA>(YT + (size_t)j0 * k, L + (size_t)i0 * ldl, k, NB, ni, l0, l0 + NB, As, Bs, acc, ldl);
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int mm = acc_col<MMA>(u, v);
      if (mm < ni) YT[(size_t)(j0 + acc_row<MMA>(u, v)) * k + i0 + mm] += acc[u][v];
    }
}

// M = Y^T Y = YT YT^T: tile (ta, tb), tb <= ta, sums over the columns l >= block ta; both triangles are written
template <bool MMA>
__global__ __launch_bounds__(256) void k_dense_lauum(int k, const double *__restrict__ YT, double *__restrict__ M,
                                                     int ldm, const InvJob *__restrict__ jobs = nullptr) {
  if (jobs) {
    const InvJob J = jobs[blockIdx.z];
    if (J.mt < 0) return;
    k = J.c;
    if (ldm <= 0) ldm = J.c;  // (a batch of dense inverses passes its padded leading dimension)
    YT += J.yt;
    M += J.mt;
  }
  __shared__ double As[NB][NB + 1];
  __shared__ double Bs[NB][NB + 1];
  int ta = (int)((sqrtf(8.0f * (float)blockIdx.x + 1.0f) - 1.0f) * 0.5f);
  while ((ta + 1) * (ta + 2) / 2 <= (int)blockIdx.x) ++ta;
  while (ta * (ta + 1) / 2 > (int)blockIdx.x) --ta;
  const int tb = blockIdx.x - ta * (ta + 1) / 2;
  const int a0 = ta * NB, b0 = tb * NB;
  if (a0 >= k) return;  // (batched: the grid is sized for the widest piece)
  const int na = min(NB, k - a0), nb = min(NB, k - b0);
  double acc[4][4];
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int v = 0; v < 4; ++v) acc[u][v] = 0;
  tile_product_range<MMA>(YT + (size_t)a0 * k, YT + (size_t)b0 * k, k, na, nb, a0, k, As, Bs, acc);
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int a = a0 + acc_row<MMA>(u, v), b = b0 + acc_col<MMA>(u, v);
      if (a < k && b < k) {
        M[(size_t)a * ldm + b] = acc[u][v];
        M[(size_t)b * ldm + a] = acc[u][v];
      }
    }
}

// ---- inverses of a WIDE dissection piece on the device (the builder of the partitioned inverse, sparse_precond.h):
//      from the factored front [L11; L21] (leading dimension f): YT = (L11^-1)^T with the right-looking kernels above,
//      W = -L21 L11^-1, and for pieces without rows below M = L11^-T L11^-1 ----
// inverse of the 64 x 64 diagonal blocks of a lower-triangular matrix, one workgroup per block
__global__ __launch_bounds__(256) void k_tri_inv64(int c, const double *__restrict__ L, long long ldl,
                                                   double *__restrict__ Linv,
                                                   const InvJob *__restrict__ jobs = nullptr) {
  if (jobs) {
    const InvJob J = jobs[blockIdx.z];
    c = J.c;
    L += J.l11;
    ldl = J.f;
    Linv += J.tinv;
    if ((int)blockIdx.x * NB >= c) return;
  }
  __shared__ double T[NB][NB + 1];
  __shared__ double Li[NB][NB + 1];
  const int b0 = blockIdx.x * NB, nb = min(NB, c - b0);
  const int tid = threadIdx.x;
  for (int e = tid; e < NB * NB; e += 256) {
    const int i = e >> 6, j = e & 63;
    T[i][j] = (i < nb && j <= i) ? L[(long long)(b0 + i) * ldl + b0 + j] : (i == j ? 1.0 : 0.0);
  }
  __syncthreads();
  const int k = tid >> 2, kq = tid & 3;
  for (int r = kq; r < k; r += 4) Li[r][k] = 0.0;
  if (kq == 0) Li[k][k] = 1.0 / T[k][k];
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  for (int r = k + 1; r < NB; ++r) {
    double s = 0;
    for (int l = k + kq; l < r; l += 4) s += T[r][l] * Li[l][k];
    s += __shfl_xor(s, 1);
    s += __shfl_xor(s, 2);
    if (kq == 0) Li[r][k] = -s / T[r][r];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();
  double *__restrict__ O = Linv + (size_t)blockIdx.x * NB * NB;
  for (int e = tid; e < NB * NB; e += 256) O[e] = Li[e >> 6][e & 63];
}
// W(a, j) = -sum_{l >= j0} B(a, l) YT(j, l): tile (ta over the m rows below, tj over the c columns)
template <bool MMA>
__global__ __launch_bounds__(256) void k_piece_w(int c, int m, const double *__restrict__ B, long long ldb,
                                                 const double *__restrict__ YT, double *__restrict__ W,
                                                 const InvJob *__restrict__ jobs = nullptr) {
  if (jobs) {  // B: the front arena, W: the packed panels (rows [c, c + m) of the piece's panel are W)
    const InvJob J = jobs[blockIdx.z];
    c = J.c;
    m = J.m;
    B += J.l11 + (long long)J.c * J.f;
    ldb = J.f;
    YT += J.yt;
    W += J.pk + (long long)J.c * J.c;
    if ((int)blockIdx.x * NB >= m || (int)blockIdx.y * NB >= c) return;
  }
  __shared__ double As[NB][NB + 1];
  __shared__ double Bs[NB][NB + 1];
  const int a0 = blockIdx.x * NB, j0 = blockIdx.y * NB;
  const int na = min(NB, m - a0), nj = min(NB, c - j0);
  double acc[4][4];
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int v = 0; v < 4; ++v) acc[u][v] = 0;
  tile_product_range<MMA>(B + (long long)a0 * ldb, YT + (long long)j0 * c, ldb, na, nj, j0, c, As, Bs, acc, c);
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int a = a0 + acc_row<MMA>(u, v), j = j0 + acc_col<MMA>(u, v);
      if (a < m && j < c) W[(long long)a * c + j] = -acc[u][v];
    }
}
// panel of an inverted piece: rows [0, c) = L11^-1 (lower; = YT transposed), rows [c, c + m) = W
__global__ __launch_bounds__(256) void k_piece_pack_inverted(int c, int m, const double *__restrict__ YT,
                                                             const double *__restrict__ W, double *__restrict__ out,
                                                             const InvJob *__restrict__ jobs = nullptr) {
  if (jobs) {  // batched: W is already in place (k_piece_w wrote it into the panel), only L11^-1 is transposed in
    const InvJob J = jobs[blockIdx.z];
    c = J.c;
    m = 0;
    YT += J.yt;
    out += J.pk;
  }
  const long long n = (long long)(c + m) * c;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) {
    const int i = (int)(e / c), j = (int)(e - (long long)i * c);
    out[e] = i < c ? (j <= i ? YT[(long long)j * c + i] : 0.0) : W[(long long)(i - c) * c + j];
  }
}

// ---- the NARROW pieces (c <= 64: the leaves and the small separators, thousands of them) all at once ----
// rows [0, c) of the packed panel <- L11^-1 (lower triangle): one workgroup per piece, column k of the inverse by lane
// group k as in k_tri_inv64
__global__ __launch_bounds__(256) void k_small_inv(const PieceDev *__restrict__ pieces, const int *__restrict__ list,
                                                   const long long *__restrict__ panel_off,
                                                   const double *__restrict__ F, double *__restrict__ out,
                                                   const long long *__restrict__ m_off, double *__restrict__ Mout) {
  __shared__ double T[NB][NB + 1];
  __shared__ double Li[NB][NB + 1];
  const int s = list[blockIdx.x];
  const PieceDev P = pieces[s];
  const int c = P.c;
  const long long f = (long long)P.c + P.m;
  const double *__restrict__ L = F + P.off;
  const int tid = threadIdx.x;
  for (int e = tid; e < NB * NB; e += 256) {
    const int i = e >> 6, j = e & 63;
    T[i][j] = (i < c && j <= i) ? L[(long long)i * f + j] : (i == j ? 1.0 : 0.0);
  }
  __syncthreads();
  const int k = tid >> 2, kq = tid & 3;
  for (int r = kq; r < k; r += 4) Li[r][k] = 0.0;
  if (kq == 0) Li[k][k] = 1.0 / T[k][k];
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  for (int r = k + 1; r < NB; ++r) {
    double sum = 0;
    for (int l = k + kq; l < r; l += 4) sum += T[r][l] * Li[l][k];
    sum += __shfl_xor(sum, 1);
    sum += __shfl_xor(sum, 2);
    if (kq == 0) Li[r][k] = -sum / T[r][r];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();
  double *__restrict__ O = out + panel_off[s];
  for (int e = tid; e < c * c; e += 256) {
    const int i = e / c, j = e - i * c;
    O[e] = j <= i ? Li[i][j] : 0.0;
  }
  if (Mout) {  // M = L11^-T L11^-1, both triangles: M(a, b) = sum_{i >= max(a, b)} Li(i, a) Li(i, b), i ascending
    double *__restrict__ Mo = Mout + m_off[s];
    for (int e = tid; e < c * c; e += 256) {
      const int a = e / c, b = e - a * c;
      double sum = 0;
      for (int i = max(a, b); i < c; ++i) sum += Li[i][a] * Li[i][b];
      Mo[e] = sum;
    }
  }
}
// rows [c, c + m) of the packed panel <- W = -L21 L11^-1: 64 rows below per workgroup (blockIdx.x), piece blockIdx.y
template <bool MMA>
__global__ __launch_bounds__(256) void k_small_w(const PieceDev *__restrict__ pieces, const int *__restrict__ list,
                                                 const long long *__restrict__ panel_off,
                                                 const double *__restrict__ F, double *__restrict__ out) {
  const int s = list[blockIdx.y];
  const PieceDev P = pieces[s];
  const int c = P.c, m = P.m, a0 = blockIdx.x * NB;
  if (a0 >= m) return;
  __shared__ double As[NB][NB + 1];
  __shared__ double Bs[NB][NB + 1];
  const long long f = (long long)c + m;
  const double *__restrict__ B = F + P.off + (long long)(c + a0) * f;
  double *__restrict__ O = out + panel_off[s];
  const int na = min(NB, m - a0);
  const int tid = threadIdx.x;
  for (int e = tid; e < NB * NB; e += 256) {
    const int i = e >> 6, kk = e & 63;
    As[kk][i] = (i < na && kk < c) ? B[(long long)i * f + kk] : 0.0;  // L21(a0 + i, kk)
    Bs[i][kk] = (i < c && kk <= i) ? O[(long long)i * c + kk] : 0.0;  // L11^-1(i, kk): [summed index][column]
  }
  __syncthreads();
  double acc[4][4];
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int v = 0; v < 4; ++v) acc[u][v] = 0;
  tile_inner<MMA>(As, Bs, acc);
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int a = acc_row<MMA>(u, v), j = acc_col<MMA>(u, v);
      if (a < na && j < c) O[(long long)(c + a0 + a) * c + j] = -acc[u][v];
    }
}

inline uint64_t mix64(uint64_t h, uint64_t w) {
  h ^= w;
  h *= 0x9E3779B97F4A7C15ull;
  h ^= h >> 29;
  return h;
}
void hash_ints(const int *p, size_t n, uint64_t *h0, uint64_t *h1) {
  size_t i = 0;
  for (; i + 2 <= n; i += 2) {
    uint64_t w;
    std::memcpy(&w, p + i, 8);
    *h0 = mix64(*h0, w);
    *h1 = mix64(*h1 + 0xD1B54A32D192ED03ull, w ^ (*h0 >> 7));
  }
  uint64_t w = (i < n) ? (uint64_t)(unsigned)p[i] : 0x5bd1e995ull;
  *h0 = mix64(*h0, w ^ n);
  *h1 = mix64(*h1, w + n);
}

struct Launch {
  int kind;  // 0 extend-add, 1 potrf, 2 trsm, 3 syrk
  int list;  // offset into the device list (children or level pieces)
  int gx, gy, j0;
  int kcap = 0, ccap = -1, ncb = 0;  // syrk: K range [j0, min(kcap, c)), column cap (-1: the whole trailing matrix), column blocks
};

struct CholImage {
  std::mutex mu;  // one factorisation at a time per image (arena, values and flags are the image's)
  int device = 0;
  CholSymbolic sym;
  DevBuf<PieceDev> pieces;
  DevBuf<int> rel, lists;  // lists = level_pieces followed by slot_children
  DevBuf<long long> dest;
  DevBuf<double> linv, arena, vals;
  DevBuf<int> fail;
  DevBuf<double> logdet;
  std::vector<Launch> plan;
  double symbolic_ms = 0;
  size_t table_bytes = 0;
  // pinned staging for the values of a factorisation: the copy from the caller's pageable array took 16 - 26 ms now and
  // then (1.3 MB at k = 10 000, inside the certificate's clock) where a copy through this buffer takes 0.1 ms
  double *vals_pinned = nullptr;
  ~CholImage() {
    if (vals_pinned) (void)hipHostFree(vals_pinned);
  }
};

struct CacheSlot {
  uint64_t h0, h1;
  int n, nnz, block, device, top;
  std::shared_ptr<CholImage> img;
};
std::mutex g_mu;
std::list<CacheSlot> g_cache;

int build_image(const HostCsr &A, int block, int top, int device, std::shared_ptr<CholImage> *out) {
  auto img = std::make_shared<CholImage>();
  img->device = device;
  const auto t0 = std::chrono::steady_clock::now();
  chol_symbolic(A, block, &img->sym, top);
  const CholSymbolic &S = img->sym;
  const int np = (int)S.pieces.size();
  std::vector<PieceDev> pd((size_t)np);
  for (int s = 0; s < np; ++s) {
    pd[s].off = S.pieces[s].off;
    pd[s].c = S.pieces[s].c;
    pd[s].m = S.pieces[s].m;
    pd[s].parent = S.pieces[s].parent;
    pd[s].rel_off = S.pieces[s].rows_off;
  }
  std::vector<int> lists(S.level_pieces);
  const int child_base = (int)lists.size();
  lists.insert(lists.end(), S.slot_children.begin(), S.slot_children.end());
  int max_level_count = 1;
  for (int t = 0; t < S.nlev; ++t) {
    max_level_count = std::max(max_level_count, S.level_ptr[t + 1] - S.level_ptr[t]);
    for (int sl = S.slot_level_ptr[t]; sl < S.slot_level_ptr[t + 1]; ++sl) {
      const int first = S.slot_ptr[sl], count = S.slot_ptr[sl + 1] - first;
      int maxm = 0;
      for (int q = 0; q < count; ++q) maxm = std::max(maxm, S.pieces[S.slot_children[first + q]].m);
      if (count > 0 && maxm > 0)
        img->plan.push_back(Launch{0, child_base + first, (maxm + kEaRows - 1) / kEaRows, count, 0});
    }
    const int lp = S.level_ptr[t], ln = S.level_ptr[t + 1] - lp;
    const int cmax = S.pieces[S.level_pieces[lp]].c;
    // super-panels of spb 64-column blocks: inside one, a factored block updates only the column blocks of the
    // super-panel that are still to come; behind it, ONE update of rank 64 spb sweeps the trailing matrix
    const int spb = chol_superpanel_blocks(), spw = spb * NB;
    for (int s0 = 0; s0 < cmax; s0 += spw) {
      int active0 = 0, maxrows2 = 0;
      while (active0 < ln && S.pieces[S.level_pieces[lp + active0]].c > s0) {
        const CholPiece &P = S.pieces[S.level_pieces[lp + active0]];
        maxrows2 = std::max(maxrows2, P.c + P.m - std::min(s0 + spw, P.c));
        ++active0;
      }
      for (int j0 = s0; j0 < std::min(cmax, s0 + spw); j0 += NB) {
        int active = 0, maxrows = 0, maxrows_block = 0;
        while (active < ln && S.pieces[S.level_pieces[lp + active]].c > j0) {
          const CholPiece &P = S.pieces[S.level_pieces[lp + active]];
          const int base = j0 + std::min(NB, P.c - j0);
          maxrows = std::max(maxrows, P.c + P.m - base);
          maxrows_block = std::max(maxrows_block, P.c + P.m - j0);
          ++active;
        }
        // left-looking inside the super-panel: before block j0 is factored, its 64 columns (all rows from j0 down) take
        // ONE update of rank j0 - s0 from the blocks of the super-panel already factored -- the same flops as a rank-64
        // update of the remaining blocks after every block, but the column block is read and written once and the K
        // loop runs pipelined over up to 7 steps (13 -> 30 Tflop/s on these launches)
        if (spb > 1 && j0 > s0)
          img->plan.push_back(Launch{3, lp, (maxrows_block + NB - 1) / NB, active, s0, j0, j0 + NB, 1});
        img->plan.push_back(Launch{1, lp, active, 1, j0});
        if (maxrows > 0) {
          const int T = (maxrows + NB - 1) / NB;
          img->plan.push_back(Launch{2, lp, T, active, j0});
          if (spb == 1) img->plan.push_back(Launch{3, lp, T * (T + 1) / 2, active, j0, j0 + NB, -1, 0});
        }
      }
      if (spb > 1 && maxrows2 > 0) {
        const int T = (maxrows2 + NB - 1) / NB;
        img->plan.push_back(Launch{3, lp, T * (T + 1) / 2, active0, s0, s0 + spw, -1, 0});
      }
    }
  }
  img->symbolic_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  DCORA_HIP(img->pieces.alloc(pd.size()));
  DCORA_HIP(hipMemcpy(img->pieces.p, pd.data(), pd.size() * sizeof(PieceDev), hipMemcpyHostToDevice));
  DCORA_HIP(img->rel.alloc(std::max<size_t>(1, S.rel.size())));
  if (!S.rel.empty()) DCORA_HIP(hipMemcpy(img->rel.p, S.rel.data(), S.rel.size() * sizeof(int), hipMemcpyHostToDevice));
  DCORA_HIP(img->lists.alloc(lists.size()));
  DCORA_HIP(hipMemcpy(img->lists.p, lists.data(), lists.size() * sizeof(int), hipMemcpyHostToDevice));
  DCORA_HIP(img->dest.alloc(S.a_dest.size()));
  DCORA_HIP(hipMemcpy(img->dest.p, S.a_dest.data(), S.a_dest.size() * sizeof(long long), hipMemcpyHostToDevice));
  DCORA_HIP(img->linv.alloc((size_t)max_level_count * NB * NB));
  DCORA_HIP(img->vals.alloc(S.a_dest.size()));
  // (small problems only: there the stall of a pageable copy is the factorisation's whole time several times over; an
  // image of the 100k lattice would pin 80 MB per cached pattern)
  if (S.a_dest.size() * sizeof(double) <= ((size_t)32 << 20) &&
      hipHostMalloc((void **)&img->vals_pinned, (S.a_dest.size() + 2) * sizeof(double), hipHostMallocDefault) != hipSuccess) {
    (void)hipGetLastError();
    img->vals_pinned = nullptr;  // the pageable copy below still works
  }
  DCORA_HIP(img->fail.alloc(1));
  DCORA_HIP(img->logdet.alloc(1));
  img->table_bytes = pd.size() * sizeof(PieceDev) + (S.rel.size() + lists.size()) * sizeof(int) +
                     S.a_dest.size() * (sizeof(long long) + sizeof(double)) + (size_t)max_level_count * NB * NB * 8;
  *out = img;
  return DCORA_OK;
}

// arenas up to this many bytes stay with the cached image (allocating and freeing tens of GB per factorisation costs
// more than the factorisation); larger ones are allocated per call.  Default: a quarter of the device's memory.
size_t arena_keep_bytes() {
  static const size_t b = [] {
    size_t fr = 0, tot = 0;
    if (hipMemGetInfo(&fr, &tot) != hipSuccess) return (size_t)2048 << 20;
    return tot / 4;
  }();
  return b;
}

}  // namespace

namespace {
void scratch_clear();  // the dense builds' recycled arenas (below)
}
void chol_cache_clear() {
  std::lock_guard<std::mutex> lk(g_mu);
  g_cache.clear();
  scratch_clear();
}

int device_chol_prepare(const HostCsr &A, int block, int device) {
  if (A.n <= 0 || (int)A.rp.size() != A.n + 1) {
    set_last_error("sparse Cholesky: empty or malformed matrix");
    return DCORA_ERR_BAD_ARG;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    set_last_error("no HIP device available: libdcora_hip has no CPU fallback");
    return DCORA_ERR_NO_DEVICE;
  }
  DCORA_HIP(hipSetDevice(device));
  uint64_t h0 = 0x243F6A8885A308D3ull, h1 = 0x13198A2E03707344ull;
  hash_ints(A.rp.data(), A.rp.size(), &h0, &h1);
  hash_ints(A.ci.data(), A.ci.size(), &h0, &h1);
  {
    std::lock_guard<std::mutex> lk(g_mu);
    for (const CacheSlot &c : g_cache)
      if (c.h0 == h0 && c.h1 == h1 && c.n == A.n && c.nnz == A.nnz() && c.block == block && c.device == device)
        return DCORA_OK;  // any order serves a verdict
  }
  std::shared_ptr<CholImage> img;
  const int rc = build_image(A, block, 0, device, &img);
  if (rc) return rc;
  const size_t arena_bytes = (size_t)img->sym.arena * sizeof(double);
  if (arena_bytes <= arena_keep_bytes() / 4 && !img->arena.p) DCORA_HIP(img->arena.alloc((size_t)img->sym.arena));
  std::lock_guard<std::mutex> lk(g_mu);
  g_cache.push_front(CacheSlot{h0, h1, A.n, A.nnz(), block, device, 0, img});
  while (g_cache.size() > 8) g_cache.pop_back();
  if (env::init_timing())
    fprintf(stderr, "[factor] prepared: n %d nnz %d block %d, %zu cached, key %016llx\n", A.n, A.nnz(), block, g_cache.size(),
            (unsigned long long)h0);
  return DCORA_OK;
}

namespace {
int factor_on_device(const HostCsr &A, int block, int top, int device, bool *pd, double *info8, PiecewiseFactor *panels) {
  double *info6 = info8;
  if (A.n <= 0 || (int)A.rp.size() != A.n + 1) {
    set_last_error("sparse Cholesky: empty or malformed matrix");
    return DCORA_ERR_BAD_ARG;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    set_last_error("no HIP device available: libdcora_hip has no CPU fallback");
    return DCORA_ERR_NO_DEVICE;
  }
  DCORA_HIP(hipSetDevice(device));
  const auto t0 = std::chrono::steady_clock::now();
  constexpr bool use_cache = true;
  uint64_t h0 = 0x243F6A8885A308D3ull, h1 = 0x13198A2E03707344ull;
  hash_ints(A.rp.data(), A.rp.size(), &h0, &h1);
  hash_ints(A.ci.data(), A.ci.size(), &h0, &h1);
  std::shared_ptr<CholImage> img;
  bool hit = false;
  if (use_cache) {
    std::lock_guard<std::mutex> lk(g_mu);
    // a verdict alone (no panels asked for) does not care about the order: an analysis of the same pattern made for
    // the preconditioner (dense top, device_chol.h) serves the PSD test of S = Q - Lambda, which has Q's pattern --
    // the certificate of the 100k lattice saves the 0.25 s analysis and a second 18 GB arena
    for (int pass = 0; pass < (panels ? 1 : 2) && !img; ++pass)
      for (auto it = g_cache.begin(); it != g_cache.end(); ++it)
        if (it->h0 == h0 && it->h1 == h1 && it->n == A.n && it->nnz == A.nnz() && it->block == block &&
            it->device == device && (it->top == top || pass == 1)) {
          g_cache.splice(g_cache.begin(), g_cache, it);
          img = g_cache.front().img;
          hit = true;
          break;
        }
  }
  if (env::init_timing())
    fprintf(stderr, "[factor] look-up: %s (n %d nnz %d block %d top %d, %zu cached, key %016llx)\n", img ? "hit" : "MISS", A.n,
            A.nnz(), block, top, g_cache.size(), (unsigned long long)h0);
  if (!img) {
    const int rc = build_image(A, block, top, device, &img);
    if (rc) return rc;
    if (use_cache) {
      std::lock_guard<std::mutex> lk(g_mu);
      g_cache.push_front(CacheSlot{h0, h1, A.n, A.nnz(), block, device, top, img});
      while (g_cache.size() > 8) g_cache.pop_back();
    }
  }
  const CholSymbolic &S = img->sym;
  std::lock_guard<std::mutex> image_lock(img->mu);
  const auto t1 = std::chrono::steady_clock::now();
  hipStream_t st = nullptr;
  int rc = stream_acquire(device, &st);
  if (rc) return rc;
  struct Rel {
    int device;
    hipStream_t st;
    ~Rel() { stream_release(device, st); }
  } rel_guard{device, st};
  DevBuf<double> local_arena;
  double *F = nullptr;
  const size_t arena_bytes = (size_t)S.arena * sizeof(double);
  if (arena_bytes <= arena_keep_bytes()) {
    if (!img->arena.p) {
      // the arenas kept by all cached images together stay within the same bound: idle images give theirs up
      std::lock_guard<std::mutex> lk(g_mu);
      size_t held = arena_bytes;
      for (CacheSlot &c : g_cache)
        if (c.img != img && c.img->arena.p) {
          held += c.img->arena.n * sizeof(double);
          if (held > arena_keep_bytes() && c.img->mu.try_lock()) {
            held -= c.img->arena.n * sizeof(double);
            c.img->arena.release();
            c.img->mu.unlock();
          }
        }
    }
    if (!img->arena.p) DCORA_HIP(img->arena.alloc((size_t)S.arena));
    F = img->arena.p;
  } else {
    DCORA_HIP(local_arena.alloc((size_t)S.arena));
    F = local_arena.p;
  }
  const long long nnz = (long long)S.a_dest.size();
  const auto t1b = std::chrono::steady_clock::now();
  const double *vsrc = A.v.data();
  if (img->vals_pinned) {
    std::memcpy(img->vals_pinned, A.v.data(), (size_t)nnz * sizeof(double));
    vsrc = img->vals_pinned;
  }
  DCORA_HIP(hipMemcpyAsync(img->vals.p, vsrc, (size_t)nnz * sizeof(double), hipMemcpyHostToDevice, st));
  const auto t1x = std::chrono::steady_clock::now();
  DCORA_HIP(hipMemsetAsync(F, 0, arena_bytes, st));
  const auto t1y = std::chrono::steady_clock::now();
  DCORA_HIP(hipMemsetAsync(img->fail.p, 0, sizeof(int), st));
  DCORA_HIP(hipMemsetAsync(img->logdet.p, 0, sizeof(double), st));
  hipLaunchKernelGGL(k_chol_scatter, dim3((unsigned)((nnz + 255) / 256)), dim3(256), 0, st, nnz, img->dest.p,
                     img->vals.p, F);
  for (const Launch &L : img->plan) {
    const int *list = img->lists.p + L.list;
    switch (L.kind) {
      case 0:
        hipLaunchKernelGGL(k_chol_extend_add, dim3(L.gx, L.gy), dim3(256), 0, st, img->pieces.p, list, img->rel.p, F,
                           img->fail.p);
        break;
      case 1:
        hipLaunchKernelGGL(k_chol_potrf, dim3(L.gx), dim3(256), 0, st, img->pieces.p, list, L.j0, F, img->linv.p,
                           img->fail.p, img->logdet.p, 0);
        break;
      case 2:
        DCORA_LAUNCH_MMA(k_chol_trsm, dim3(L.gx, L.gy), st, img->pieces.p, list, L.j0, F, img->linv.p,
                           img->fail.p);
        break;
      default:
        DCORA_LAUNCH_MMA(k_chol_syrk, dim3(L.gx, L.gy), st, img->pieces.p, list, L.j0, L.kcap, L.ccap, L.ncb, F,
                           img->fail.p);
        break;
    }
  }
  DCORA_HIP(hipGetLastError());
  const auto t1c = std::chrono::steady_clock::now();
  int failed = 0;
  double logdet = 0;
  if (img->vals_pinned) {  // (the verdict comes back through the pinned buffer's tail: see CholImage)
    double *tail = img->vals_pinned + nnz;
    DCORA_HIP(hipMemcpyAsync(tail, img->fail.p, sizeof(int), hipMemcpyDeviceToHost, st));
    DCORA_HIP(hipMemcpyAsync(tail + 1, img->logdet.p, sizeof(double), hipMemcpyDeviceToHost, st));
    DCORA_HIP(hipStreamSynchronize(st));
    std::memcpy(&failed, tail, sizeof(int));
    logdet = tail[1];
  } else {
    DCORA_HIP(hipMemcpyAsync(&failed, img->fail.p, sizeof(int), hipMemcpyDeviceToHost, st));
    DCORA_HIP(hipMemcpyAsync(&logdet, img->logdet.p, sizeof(double), hipMemcpyDeviceToHost, st));
    DCORA_HIP(hipStreamSynchronize(st));
  }
  *pd = failed == 0;
  const bool init_timing = env::init_timing();
  auto tl = std::chrono::steady_clock::now();
  auto lap = [&](const char *what) {
    if (!init_timing) return;
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "[factor] %-26s %9.1f ms\n", what, std::chrono::duration<double, std::milli>(now - tl).count());
    tl = now;
  };
  if (init_timing)
    fprintf(stderr, "[factor] %-26s %9.1f ms\n[factor] %-26s %9.1f ms (stream + arena %.1f, enqueue %.1f, wait %.1f; arena %.0f MB)\n",
            "hash + analysis / look-up", std::chrono::duration<double, std::milli>(t1 - t0).count(),
            "numeric factorisation", std::chrono::duration<double, std::milli>(tl - t1).count(),
            std::chrono::duration<double, std::milli>(t1b - t1).count(),
            std::chrono::duration<double, std::milli>(t1c - t1b).count(),
            std::chrono::duration<double, std::milli>(tl - t1c).count(), arena_bytes / 1e6);
  if (init_timing)
    fprintf(stderr, "[factor]   enqueue: copy of the values %.2f ms, arena memset %.2f ms, launches %.2f ms\n",
            std::chrono::duration<double, std::milli>(t1x - t1b).count(),
            std::chrono::duration<double, std::milli>(t1y - t1x).count(),
            std::chrono::duration<double, std::milli>(t1c - t1y).count());
  if (panels && failed == 0) {
    // hand the factor over by pieces: pack the panels on the device, one copy to the host
    const int np = (int)S.pieces.size();
    std::vector<long long> poff((size_t)np + 1, 0);
    int fmax = 1;
    for (int s2 = 0; s2 < np; ++s2) {
      const CholPiece &P = S.pieces[s2];
      poff[s2 + 1] = poff[s2] + (long long)(P.c + P.m) * P.c;
      fmax = std::max(fmax, P.c + P.m);
    }
    DevBuf<long long> dpoff;
    DevBuf<double> packed;
    DCORA_HIP(dpoff.alloc(poff.size()));
    DCORA_HIP(packed.alloc((size_t)std::max<long long>(1, poff[np])));
    DCORA_HIP(hipMemcpyAsync(dpoff.p, poff.data(), poff.size() * sizeof(long long), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_chol_pack, dim3(std::min(64, (fmax + 7) / 8), np), dim3(256), 0, st, img->pieces.p, dpoff.p, F,
                       packed.p);
    // wide pieces (no hubs): their inverses are formed here, on the device, instead of by the host's threads -- for the
    // whole 100k lattice 310 Gflop, 7.7 s on the 16 cores of the container
    // every piece arrives inverted: the narrow ones (c <= 64) by two batched launches, the ones up to 384 columns in one
    // batch of the right-looking kernels, the wide ones one after the other (on the 16 host cores of the container the
    // inverses of the whole 100k lattice, 310 Gflop, took 7.7 s)
    constexpr bool invert_on_device = true, only_wide = false;
    const int kWide = 384;
    std::vector<int> wide, narrow;
    std::vector<long long> moff((size_t)np, -1);
    long long mtotal = 0;
    // device sources (asked for by the caller whose weight sink fills on the device): every piece is inverted here, its
    // M = L11^-T L11^-1 is formed here as well and stays here, like the panels
    const bool dev_src = panels->want_device_sources && invert_on_device && S.nhub == 0 && !only_wide;
    DevBuf<double> mall;
    DevBuf<long long> dmoff;
    if (dev_src) {
      for (int s2 = 0; s2 < np; ++s2) {
        moff[s2] = mtotal;
        mtotal += (long long)S.pieces[s2].c * S.pieces[s2].c;
      }
      DCORA_HIP(mall.alloc((size_t)std::max<long long>(1, mtotal)));
      DCORA_HIP(dmoff.alloc((size_t)np));
      DCORA_HIP(hipMemcpyAsync(dmoff.p, moff.data(), (size_t)np * sizeof(long long), hipMemcpyHostToDevice, st));
    }
    if (invert_on_device && S.nhub == 0 && !only_wide) {
      int mmax = 0;
      for (int s2 = 0; s2 < np; ++s2)
        if (S.pieces[s2].c <= NB) {
          narrow.push_back(s2);
          mmax = std::max(mmax, S.pieces[s2].m);
        }
      if (!narrow.empty()) {
        DevBuf<int> dlist;
        DCORA_HIP(dlist.alloc(narrow.size()));
        DCORA_HIP(hipMemcpyAsync(dlist.p, narrow.data(), narrow.size() * sizeof(int), hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_small_inv, dim3((unsigned)narrow.size()), dim3(256), 0, st, img->pieces.p, dlist.p, dpoff.p, F,
                           packed.p, (const long long *)dmoff.p, dev_src ? mall.p : (double *)nullptr);
        if (mmax > 0)
          DCORA_LAUNCH_MMA(k_small_w, dim3((mmax + NB - 1) / NB, (unsigned)narrow.size()), st,
                             img->pieces.p, dlist.p, dpoff.p, F, packed.p);
        DCORA_HIP(hipStreamSynchronize(st));  // dlist and narrow live on this frame
      }
    }
    // the pieces in between (64 < c < 384: two thousand of them for the whole 100k lattice) in ONE batch of the
    // right-looking kernels, a piece per blockIdx.z: 15 launches for all of them
    std::vector<int> mid;
    if (invert_on_device && S.nhub == 0 && !only_wide) {
      std::vector<InvJob> jobs;
      long long yt_total = 0, tinv_total = 0;
      int nbkmax = 0, mmax = 0, cmax = 0;
      for (int s2 = 0; s2 < np; ++s2) {
        const CholPiece &P = S.pieces[s2];
        if (P.c <= NB || P.c >= kWide) continue;
        const int nbk = (P.c + NB - 1) / NB;
        InvJob J;
        J.c = P.c;
        J.m = P.m;
        J.l11 = P.off;
        J.f = (long long)P.c + P.m;
        J.yt = yt_total;
        J.tinv = tinv_total;
        J.pk = poff[s2];
        J.mt = dev_src ? moff[s2] : -1;
        yt_total += (long long)P.c * P.c;
        tinv_total += (long long)nbk * NB * NB;
        nbkmax = std::max(nbkmax, nbk);
        mmax = std::max(mmax, P.m);
        cmax = std::max(cmax, P.c);
        jobs.push_back(J);
        mid.push_back(s2);
      }
      if (!jobs.empty()) {
        const unsigned nj = (unsigned)jobs.size();
        DevBuf<InvJob> djobs;
        DevBuf<double> ytm, tinvm;
        DCORA_HIP(djobs.alloc(jobs.size()));
        DCORA_HIP(ytm.alloc((size_t)yt_total));
        DCORA_HIP(tinvm.alloc((size_t)tinv_total));
        DCORA_HIP(hipMemcpyAsync(djobs.p, jobs.data(), jobs.size() * sizeof(InvJob), hipMemcpyHostToDevice, st));
        DCORA_HIP(hipMemsetAsync(ytm.p, 0, (size_t)yt_total * sizeof(double), st));
        hipLaunchKernelGGL(k_tri_inv64, dim3(nbkmax, 1, nj), dim3(256), 0, st, 0, F, 0LL, tinvm.p, djobs.p);
        for (int ib = 0; ib < nbkmax; ++ib) {
          DCORA_LAUNCH_MMA(k_dense_trtri_finish, dim3(ib + 1, 1, nj), st, 0, ib, ytm.p, tinvm.p, djobs.p);
          if (ib + 1 < nbkmax)
            DCORA_LAUNCH_MMA(k_dense_trtri_update, dim3(nbkmax - ib - 1, ib + 1, nj), st, 0, ib, F, 0LL, ytm.p, djobs.p);
        }
        if (mmax > 0)
          DCORA_LAUNCH_MMA(k_piece_w, dim3((mmax + NB - 1) / NB, nbkmax, nj), st, 0, 0, F, 0LL, ytm.p, packed.p, djobs.p);
        hipLaunchKernelGGL(k_piece_pack_inverted, dim3(std::min(64, (cmax * cmax + 255) / 256), 1, nj), dim3(256), 0, st, 0,
                           0, ytm.p, (const double *)nullptr, packed.p, djobs.p);
        if (dev_src)
          DCORA_LAUNCH_MMA(k_dense_lauum, dim3(nbkmax * (nbkmax + 1) / 2, 1, nj), st, 0, ytm.p, mall.p, 0, djobs.p);
        DCORA_HIP(hipGetLastError());
        DCORA_HIP(hipStreamSynchronize(st));  // the job list and the scratch arenas live on this frame
      }
    }
    if (init_timing) {
      long cnt[3] = {0, 0, 0};
      double vol[3] = {0, 0, 0}, c2[3] = {0, 0, 0};
      for (int s2 = 0; s2 < np; ++s2) {
        const CholPiece &P = S.pieces[s2];
        const int b = P.c <= NB ? 0 : (P.c < kWide ? 1 : 2);
        ++cnt[b];
        vol[b] += (double)(P.c + P.m) * P.c;
        c2[b] += (double)P.c * P.c;
      }
      fprintf(stderr, "[factor] pieces: narrow %ld (%.0f MB panels, %.0f MB c^2), mid %ld (%.0f MB, %.0f MB), wide %ld (%.0f MB, %.0f MB)\n",
              cnt[0], vol[0] * 8e-6, c2[0] * 8e-6, cnt[1], vol[1] * 8e-6, c2[1] * 8e-6, cnt[2], vol[2] * 8e-6, c2[2] * 8e-6);
    }
    if (invert_on_device && S.nhub == 0)
      for (int s2 = 0; s2 < np; ++s2)
        if (S.pieces[s2].c >= kWide) {
          wide.push_back(s2);
          // L11^-T L11^-1 of every wide piece: the schedule (host_partinv3.cpp) applies the two triangular
          // products of a piece as this one symmetric product
          if (!dev_src) {
            moff[s2] = mtotal;
            mtotal += (long long)S.pieces[s2].c * S.pieces[s2].c;
          }
        }
    DevBuf<double> yt, wbuf, tinv, mtop_own;
    if (!wide.empty()) {
      long long cmax = 0, wmax = 1;
      for (int s2 : wide) {
        cmax = std::max<long long>(cmax, S.pieces[s2].c);
        wmax = std::max(wmax, (long long)S.pieces[s2].m * S.pieces[s2].c);
      }
      DCORA_HIP(yt.alloc((size_t)(cmax * cmax)));
      DCORA_HIP(wbuf.alloc((size_t)wmax));
      DCORA_HIP(tinv.alloc((size_t)((cmax + NB - 1) / NB) * NB * NB));
      if (!dev_src) DCORA_HIP(mtop_own.alloc((size_t)std::max<long long>(1, mtotal)));
      for (int s2 : wide) {
        const CholPiece &P = S.pieces[s2];
        const int c = P.c, m = P.m, nbk = (c + NB - 1) / NB;
        const long long f = (long long)c + m;
        const double *L11 = F + P.off;
        hipLaunchKernelGGL(k_tri_inv64, dim3(nbk), dim3(256), 0, st, c, L11, f, tinv.p);
        DCORA_HIP(hipMemsetAsync(yt.p, 0, (size_t)c * c * sizeof(double), st));
        for (int ib = 0; ib < nbk; ++ib) {
          DCORA_LAUNCH_MMA(k_dense_trtri_finish, dim3(ib + 1), st, c, ib, yt.p, tinv.p);
          if (ib + 1 < nbk)
            DCORA_LAUNCH_MMA(k_dense_trtri_update, dim3(nbk - ib - 1, ib + 1), st, c, ib, L11, f, yt.p);
        }
        if (m > 0)
          DCORA_LAUNCH_MMA(k_piece_w, dim3((m + NB - 1) / NB, nbk), st, c, m, L11 + (long long)c * f, f,
                             yt.p, wbuf.p);
        hipLaunchKernelGGL(k_piece_pack_inverted, dim3(std::min<long long>(4096, ((f * c) + 255) / 256)), dim3(256), 0,
                           st, c, m, yt.p, wbuf.p, packed.p + poff[s2]);
        DCORA_LAUNCH_MMA(k_dense_lauum, dim3(nbk * (nbk + 1) / 2), st, c, yt.p, (dev_src ? mall.p : mtop_own.p) + moff[s2], c);
      }
      DCORA_HIP(hipGetLastError());
    }
    DCORA_HIP(hipStreamSynchronize(st));
    lap("pack + wide inverses");
    // one host block for all panels (and the wide pieces' M): not value-initialised -- the copy below writes every
    // entry, and zeroing 3 GB first cost 0.76 s for the whole 100k lattice -- and handed to the pieces as views
    // (2 MB-aligned and advised as huge pages: first touch and release of 3 GB in 4 KB pages cost 0.4 s each)
    struct HostBlock {
      double *panels = nullptr, *M = nullptr;
      void *m_tokens = nullptr;  // reserved, unreadable addresses standing for the M that exist on the device only
      size_t m_tokens_bytes = 0;
      DevBuf<double> dev_panels, dev_M;  // device sources of a sink that fills on the device
      ~HostBlock() {
        std::free(panels);
        std::free(M);
        if (m_tokens) munmap(m_tokens, m_tokens_bytes);
      }
    };
    auto huge_alloc = [](size_t doubles) -> double * {
      const size_t two_mb = (size_t)2 << 20;
      const size_t bytes = ((std::max<size_t>(doubles, 1) * sizeof(double) + two_mb - 1) / two_mb) * two_mb;
      void *p = std::aligned_alloc(two_mb, bytes);
      if (p) (void)madvise(p, bytes, MADV_HUGEPAGE);
      return (double *)p;
    };
    auto blk = std::make_shared<HostBlock>();
    blk->panels = huge_alloc((size_t)poff[np]);
    if (dev_src) {
      blk->m_tokens_bytes = (size_t)std::max<long long>(1, mtotal) * sizeof(double);
      void *tok = mmap(nullptr, blk->m_tokens_bytes, PROT_NONE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
      if (tok == MAP_FAILED) {
        set_last_error("sparse Cholesky: could not reserve the address range of the device-resident M");
        return DCORA_ERR_HIP;
      }
      blk->m_tokens = tok;
    } else {
      blk->M = huge_alloc((size_t)mtotal);
    }
    if (!blk->panels || (!dev_src && !blk->M)) {
      set_last_error("sparse Cholesky: no host memory for the factor's panels");
      return DCORA_ERR_HIP;
    }
    double *host = blk->panels, *hostM = dev_src ? (double *)blk->m_tokens : blk->M;
    // non-zeros of the factor (reported as nnz(L)), counted where the factor is
    DevBuf<unsigned long long> dnz;
    DCORA_HIP(dnz.alloc(1));
    DCORA_HIP(hipMemsetAsync(dnz.p, 0, sizeof(unsigned long long), st));
    hipLaunchKernelGGL(k_count_nonzero, dim3(2048), dim3(256), 0, st, (long long)poff[np], packed.p, dnz.p);
    unsigned long long nz_dev = 0;
    DCORA_HIP(hipMemcpyAsync(&nz_dev, dnz.p, sizeof nz_dev, hipMemcpyDeviceToHost, st));
    lap("host buffers");
    DCORA_HIP(hipMemcpyAsync(host, packed.p, (size_t)poff[np] * sizeof(double), hipMemcpyDeviceToHost, st));
    if (mtotal > 0 && !dev_src)
      DCORA_HIP(hipMemcpyAsync(hostM, mtop_own.p, (size_t)mtotal * sizeof(double), hipMemcpyDeviceToHost, st));
    DCORA_HIP(hipStreamSynchronize(st));
    lap("download");
    std::vector<char> is_wide((size_t)np, 0);
    for (int s2 : wide) is_wide[s2] = 1;
    for (int s2 : narrow) is_wide[s2] = 1;  // narrow pieces arrive inverted as well
    for (int s2 : mid) is_wide[s2] = 1;     // and so do the ones in between
    PiecewiseFactor &W = *panels;
    W = PiecewiseFactor();
    W.n = S.n;
    W.nhub = S.nhub;
    W.perm = S.perm;
    W.iperm = S.iperm;
    W.pieces.assign((size_t)np, PieceFactor());
    W.block = blk;
    for (int s2 = 0; s2 < np; ++s2) {
      const CholPiece &P = S.pieces[s2];
      PieceFactor &pf = W.pieces[s2];
      pf.c0 = P.c0;
      pf.c = P.c;
      pf.rows.assign(S.rows.begin() + P.rows_off, S.rows.begin() + P.rows_off + P.m);
      pf.panel_view = host + poff[s2];
      pf.inverted = is_wide[s2] != 0;
      if (moff[s2] >= 0) pf.Mtop_view = hostM + moff[s2];
    }
    W.nnzL = (long)nz_dev;
    if (dev_src) {
      W.m_on_device_only = true;
      W.mirrors.push_back(MirrorRange{host, (long long)poff[np], packed.p});
      W.mirrors.push_back(MirrorRange{hostM, mtotal, mall.p});
      blk->dev_panels = std::move(packed);
      blk->dev_M = std::move(mall);
    }
    lap("per-piece copies");
  }
  if (info6) {
    const auto t2 = std::chrono::steady_clock::now();
    info6[0] = hit ? 0.0 : img->symbolic_ms;
    info6[1] = std::chrono::duration<double, std::milli>(t2 - t1).count();
    info6[2] = (double)arena_bytes;
    info6[3] = S.flops;
    info6[4] = (double)S.nlev;
    info6[5] = (double)img->plan.size();
    info6[6] = logdet;
    info6[7] = std::chrono::duration<double, std::milli>(t1 - t0).count();  // pattern hash + cache look-up / analysis
  }
  return DCORA_OK;
}
}  // namespace

int device_chol_is_pd(const HostCsr &A, int block, int device, bool *pd, double *info8) {
  return factor_on_device(A, block, 0, device, pd, info8, nullptr);
}

int device_chol_piecewise_factor(const HostCsr &A, int block, int top_unknowns, int device, PiecewiseFactor *out,
                                 double *info8) {
  bool pd = false;
  const int rc = factor_on_device(A, block, top_unknowns, device, &pd, info8, out);
  if (rc) return rc;
  if (!pd) {
    set_last_error("matrix is not positive definite");
    return DCORA_ERR_NOT_PD;
  }
  return DCORA_OK;
}

// Scratch arenas of the dense builds, recycled process-wide: hipFree of a few hundred MB takes milliseconds (and waits for
// the device), and sessions are created again and again (staircase levels, GNC rounds).  At most two idle arenas per
// process are kept; chol_cache_clear() drops them.
namespace {
std::mutex g_scratch_mu;
struct Scratch {
  int device;
  size_t bytes;
  char *p;
};
std::vector<Scratch> g_scratch_idle;
}  // namespace
char *scratch_acquire(int device, size_t bytes) {
  {
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    for (size_t i = 0; i < g_scratch_idle.size(); ++i)
      if (g_scratch_idle[i].device == device && g_scratch_idle[i].bytes >= bytes &&
          g_scratch_idle[i].bytes <= 4 * bytes + (64u << 20)) {
        char *p = g_scratch_idle[i].p;
        g_scratch_idle.erase(g_scratch_idle.begin() + (long)i);
        return p;
      }
  }
  char *p = nullptr;
  if (hipMalloc((void **)&p, bytes) != hipSuccess) return nullptr;
  return p;
}
void scratch_release(int device, char *p, size_t bytes) {
  if (!p) return;
  {
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    if (g_scratch_idle.size() < 2) {
      g_scratch_idle.push_back(Scratch{device, bytes, p});
      return;
    }
  }
  (void)hipFree(p);
}
namespace {
std::mutex g_pinned_mu;
struct Pinned {
  size_t bytes;
  char *p;
};
std::vector<Pinned> g_pinned_idle;
}  // namespace
char *pinned_acquire(size_t bytes) {
  bytes = (bytes + ((size_t)1 << 20) - 1) & ~(((size_t)1 << 20) - 1);
  {
    std::lock_guard<std::mutex> lk(g_pinned_mu);
    for (size_t i = 0; i < g_pinned_idle.size(); ++i)
      if (g_pinned_idle[i].bytes >= bytes && g_pinned_idle[i].bytes <= 4 * bytes) {
        char *p = g_pinned_idle[i].p;
        g_pinned_idle.erase(g_pinned_idle.begin() + (long)i);
        return p;
      }
  }
  char *p = nullptr;
  if (hipHostMalloc((void **)&p, bytes, hipHostMallocDefault) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  return p;
}
void pinned_release(char *p, size_t bytes) {
  if (!p) return;
  bytes = (bytes + ((size_t)1 << 20) - 1) & ~(((size_t)1 << 20) - 1);
  {
    std::lock_guard<std::mutex> lk(g_pinned_mu);
    if (bytes <= ((size_t)32 << 20) && g_pinned_idle.size() < 3) {  // (large ones go back: pinned memory is the host's)
      g_pinned_idle.push_back(Pinned{bytes, p});
      return;
    }
  }
  (void)hipHostFree(p);
}
namespace {
void scratch_clear() {
  std::lock_guard<std::mutex> lk(g_scratch_mu);
  for (Scratch &s : g_scratch_idle) (void)hipFree(s.p);
  g_scratch_idle.clear();
}
}  // namespace

int device_dense_spd_inverse(const HostCsr &A, int device, double *Minv, int ldm, bool *pd) {
  const int k = A.n;
  const auto t_in = std::chrono::steady_clock::now();
  auto lap = [&](const char *what) {
    if (env::init_timing())
      fprintf(stderr, "[dense inverse] %-10s at %7.2f ms\n", what,
              std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_in).count());
  };
  DCORA_HIP(hipSetDevice(device));
  hipStream_t st = nullptr;
  int rc = stream_acquire(device, &st);
  if (rc) return rc;
  struct Rel {
    int device;
    hipStream_t st;
    ~Rel() { stream_release(device, st); }
  } rel_guard{device, st};
  const int nb = (k + NB - 1) / NB;
  // one allocation for everything (hipMalloc / hipFree synchronise the device: agents are set up concurrently)
  const size_t nnz = A.ci.size();
  auto up16 = [](size_t bytes) { return (bytes + 15) & ~(size_t)15; };
  const size_t o_L = 0, o_YT = o_L + up16((size_t)k * k * 8), o_linv = o_YT + up16((size_t)k * k * 8),
               o_v = o_linv + up16((size_t)nb * NB * NB * 8), o_logdet = o_v + up16(std::max<size_t>(1, nnz) * 8),
               o_piece = o_logdet + 16, o_rp = o_piece + up16(sizeof(PieceDev)), o_ci = o_rp + up16(((size_t)k + 1) * 4),
               o_fail = o_ci + up16(std::max<size_t>(1, nnz) * 4), o_list = o_fail + 16, total = o_list + 16;
  struct Arena {
    int device;
    size_t bytes;
    char *p;
    ~Arena() { scratch_release(device, p, bytes); }
  } arena{device, total, scratch_acquire(device, total)};
  if (!arena.p) {
    set_last_error("dense inverse: out of device memory");
    return DCORA_ERR_HIP;
  }
  lap("arena");
  struct {
    double *p;
  } L{(double *)(arena.p + o_L)}, YT{(double *)(arena.p + o_YT)}, linv{(double *)(arena.p + o_linv)},
      v{(double *)(arena.p + o_v)}, logdet{(double *)(arena.p + o_logdet)};
  struct {
    int *p;
  } rp{(int *)(arena.p + o_rp)}, ci{(int *)(arena.p + o_ci)}, fail{(int *)(arena.p + o_fail)},
      list{(int *)(arena.p + o_list)};
  struct {
    PieceDev *p;
  } piece{(PieceDev *)(arena.p + o_piece)};
  const PieceDev one{0, k, 0, -1, 0};
  const int zero_list = 0;
  DCORA_HIP(hipMemcpyAsync(rp.p, A.rp.data(), ((size_t)k + 1) * sizeof(int), hipMemcpyHostToDevice, st));
  DCORA_HIP(hipMemcpyAsync(ci.p, A.ci.data(), nnz * sizeof(int), hipMemcpyHostToDevice, st));
  DCORA_HIP(hipMemcpyAsync(v.p, A.v.data(), nnz * sizeof(double), hipMemcpyHostToDevice, st));
  DCORA_HIP(hipMemcpyAsync(piece.p, &one, sizeof one, hipMemcpyHostToDevice, st));
  DCORA_HIP(hipMemcpyAsync(list.p, &zero_list, sizeof(int), hipMemcpyHostToDevice, st));
  DCORA_HIP(hipMemsetAsync(L.p, 0, o_linv, st));  // L and YT
  DCORA_HIP(hipMemsetAsync(fail.p, 0, sizeof(int), st));
  DCORA_HIP(hipMemsetAsync(logdet.p, 0, sizeof(double), st));
  DCORA_HIP(hipMemsetAsync(Minv, 0, (size_t)k * ldm * sizeof(double), st));
  hipLaunchKernelGGL(k_dense_scatter, dim3(k), dim3(256), 0, st, k, rp.p, ci.p, v.p, L.p);
  for (int p = 0; p < nb; ++p) {
    const int j0 = p * NB, jb = std::min(NB, k - j0), rows = k - j0 - jb;
    hipLaunchKernelGGL(k_chol_potrf, dim3(1), dim3(256), 0, st, piece.p, list.p, j0, L.p, linv.p + (size_t)p * NB * NB,
                       fail.p, logdet.p, 1);
    if (rows > 0) {
      const int T = (rows + NB - 1) / NB;
      DCORA_LAUNCH_MMA(k_chol_trsm, dim3(T, 1), st, piece.p, list.p, j0, L.p,
                         linv.p + (size_t)p * NB * NB, fail.p);
      DCORA_LAUNCH_MMA(k_chol_syrk, dim3(T * (T + 1) / 2, 1), st, piece.p, list.p, j0, j0 + NB, -1, 0, L.p, fail.p);
    }
  }
  // (a failed factorisation leaves garbage behind the flag: the products below are harmless, the caller drops Minv)
  for (int ib = 0; ib < nb; ++ib) {
    DCORA_LAUNCH_MMA(k_dense_trtri_finish, dim3(ib + 1), st, k, ib, YT.p, linv.p);
    if (ib + 1 < nb)
      DCORA_LAUNCH_MMA(k_dense_trtri_update, dim3(nb - ib - 1, ib + 1), st, k, ib, L.p, (long long)k,
                         YT.p);
  }
  DCORA_LAUNCH_MMA(k_dense_lauum, dim3(nb * (nb + 1) / 2), st, k, YT.p, Minv, ldm);
  DCORA_HIP(hipGetLastError());
  lap("enqueued");
  int failed = 0;
  DCORA_HIP(hipMemcpyAsync(&failed, fail.p, sizeof(int), hipMemcpyDeviceToHost, st));
  DCORA_HIP(hipStreamSynchronize(st));
  lap("done");
  *pd = failed == 0;
  return DCORA_OK;
}

// The same inverse for a BATCH of matrices of equal size in one set of launches (the agents of a session: five chains of
// ~130 small dependent launches on five streams do not overlap on this part -- the hardware runs two or three queues at a
// time, tools/stream_overlap.hip -- so five builds took 20 ms where one takes 3.5).  Every kernel of the single build
// takes a list / job index: potrf one workgroup per matrix, the panel kernels a grid dimension over the matrices.  The
// arithmetic per matrix is the single build's, kernel for kernel: the two give the same bits.
int device_dense_spd_inverse_batch(const std::vector<const HostCsr *> &As, int device, const std::vector<double *> &Minv,
                                   int ldm, std::vector<char> *pd) {
  const int B = (int)As.size();
  if (B == 0) return DCORA_OK;
  const int k = As[0]->n;
  for (const HostCsr *A : As)
    if (A->n != k) {
      set_last_error("dense inverse batch: matrices of different sizes");
      return DCORA_ERR_BAD_ARG;
    }
  const auto t_in = std::chrono::steady_clock::now();
  DCORA_HIP(hipSetDevice(device));
  hipStream_t st = nullptr;
  int rc = stream_acquire(device, &st);
  if (rc) return rc;
  struct Rel {
    int device;
    hipStream_t st;
    ~Rel() { stream_release(device, st); }
  } rel_guard{device, st};
  const int nb = (k + NB - 1) / NB;
  const size_t kk = (size_t)k * k;
  size_t nnz_max = 1;
  for (const HostCsr *A : As) nnz_max = std::max(nnz_max, A->ci.size());
  auto up16 = [](size_t bytes) { return (bytes + 15) & ~(size_t)15; };
  const size_t o_L = 0, o_YT = o_L + up16(B * kk * 8), o_tinv = o_YT + up16(B * kk * 8),
               o_v = o_tinv + up16((size_t)B * nb * NB * NB * 8), o_logdet = o_v + up16(B * nnz_max * 8),
               o_piece = o_logdet + 16, o_jobs = o_piece + up16(sizeof(PieceDev) * B),
               o_rp = o_jobs + up16(sizeof(InvJob) * B), o_ci = o_rp + up16((size_t)B * (k + 1) * 4),
               o_fail = o_ci + up16(B * nnz_max * 4), o_list = o_fail + 16, total = o_list + up16(4 * (size_t)B);
  struct Arena {
    int device;
    size_t bytes;
    char *p;
    ~Arena() { scratch_release(device, p, bytes); }
  } arena{device, total, scratch_acquire(device, total)};
  if (!arena.p) {
    set_last_error("dense inverse batch: out of device memory");
    return DCORA_ERR_HIP;
  }
  double *L = (double *)(arena.p + o_L), *YT = (double *)(arena.p + o_YT), *tinv = (double *)(arena.p + o_tinv),
         *v = (double *)(arena.p + o_v), *logdet = (double *)(arena.p + o_logdet);
  int *rp = (int *)(arena.p + o_rp), *ci = (int *)(arena.p + o_ci), *fail = (int *)(arena.p + o_fail),
      *list = (int *)(arena.p + o_list);
  PieceDev *piece = (PieceDev *)(arena.p + o_piece);
  InvJob *jobs = (InvJob *)(arena.p + o_jobs);
  std::vector<PieceDev> hp((size_t)B);
  std::vector<InvJob> hj((size_t)B);
  std::vector<int> hl((size_t)B);
  double *Mbase = Minv[0];
  for (double *q : Minv) Mbase = std::min(Mbase, q);
  for (int b = 0; b < B; ++b) {
    hp[(size_t)b] = PieceDev{(long long)(b * kk), k, 0, -1, 0};
    hj[(size_t)b] = InvJob{k, 0, (long long)(b * kk), (long long)k, (long long)(b * kk), (long long)b * NB * NB, 0,
                          (long long)(Minv[(size_t)b] - Mbase)};
    hl[(size_t)b] = b;
    const HostCsr &A = *As[(size_t)b];
    DCORA_HIP(hipMemcpyAsync(rp + (size_t)b * (k + 1), A.rp.data(), ((size_t)k + 1) * sizeof(int), hipMemcpyHostToDevice, st));
    DCORA_HIP(hipMemcpyAsync(ci + (size_t)b * nnz_max, A.ci.data(), A.ci.size() * sizeof(int), hipMemcpyHostToDevice, st));
    DCORA_HIP(hipMemcpyAsync(v + (size_t)b * nnz_max, A.v.data(), A.ci.size() * sizeof(double), hipMemcpyHostToDevice, st));
    DCORA_HIP(hipMemsetAsync(Minv[(size_t)b], 0, (size_t)k * ldm * sizeof(double), st));
  }
  DCORA_HIP(hipMemcpyAsync(piece, hp.data(), sizeof(PieceDev) * B, hipMemcpyHostToDevice, st));
  DCORA_HIP(hipMemcpyAsync(jobs, hj.data(), sizeof(InvJob) * B, hipMemcpyHostToDevice, st));
  DCORA_HIP(hipMemcpyAsync(list, hl.data(), sizeof(int) * B, hipMemcpyHostToDevice, st));
  DCORA_HIP(hipMemsetAsync(L, 0, o_tinv, st));  // L and YT of every matrix
  DCORA_HIP(hipMemsetAsync(fail, 0, sizeof(int), st));
  DCORA_HIP(hipMemsetAsync(logdet, 0, sizeof(double), st));
  for (int b = 0; b < B; ++b)
    hipLaunchKernelGGL(k_dense_scatter, dim3(k), dim3(256), 0, st, k, rp + (size_t)b * (k + 1), ci + (size_t)b * nnz_max,
                       v + (size_t)b * nnz_max, L + b * kk);
  for (int p = 0; p < nb; ++p) {
    const int j0 = p * NB, jb = std::min(NB, k - j0), rows = k - j0 - jb;
    double *step_inv = tinv + (size_t)p * B * NB * NB;  // panel p of matrix b at (p B + b) 64 x 64 blocks
    hipLaunchKernelGGL(k_chol_potrf, dim3(B), dim3(256), 0, st, piece, list, j0, L, step_inv, fail, logdet, 1);
    if (rows > 0) {
      const int T = (rows + NB - 1) / NB;
      DCORA_LAUNCH_MMA(k_chol_trsm, dim3(T, B), st, piece, list, j0, L, step_inv, fail);
      DCORA_LAUNCH_MMA(k_chol_syrk, dim3(T * (T + 1) / 2, B), st, piece, list, j0, j0 + NB, -1, 0, L, fail);
    }
  }
  for (int ib = 0; ib < nb; ++ib) {
    DCORA_LAUNCH_MMA(k_dense_trtri_finish, dim3(ib + 1, 1, B), st, 0, ib, YT, tinv, jobs, B);
    if (ib + 1 < nb) DCORA_LAUNCH_MMA(k_dense_trtri_update, dim3(nb - ib - 1, ib + 1, B), st, 0, ib, L, 0LL, YT, jobs);
  }
  DCORA_LAUNCH_MMA(k_dense_lauum, dim3(nb * (nb + 1) / 2, 1, B), st, 0, YT, Mbase, ldm, jobs);
  DCORA_HIP(hipGetLastError());
  int failed = 0;
  DCORA_HIP(hipMemcpyAsync(&failed, fail, sizeof(int), hipMemcpyDeviceToHost, st));
  DCORA_HIP(hipStreamSynchronize(st));
  // (one flag for the batch: a failure anywhere stops every later launch, the caller then builds one by one)
  pd->assign((size_t)B, failed == 0 ? 1 : 0);
  if (env::init_timing())
    fprintf(stderr, "[dense inverse] batch of %d (k = %d) in %.2f ms\n", B, k,
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_in).count());
  return DCORA_OK;
}

std::function<bool(const HostCsr &, int, int, const double *, double *)> device_spd_solver(int device) {
  return [device](const HostCsr &A, int block, int nrhs, const double *B, double *X) -> bool {
    if (nrhs < 1 || nrhs > 16) return false;
    if (hipSetDevice(device) != hipSuccess) return false;
    PartInvHost P;
    DeviceWeightSink sink(device);  // the stored weights stream to the device while they are formed
    P.sink = &sink;
    const int nthreads = std::max(2, host_cpus_available());
    if (build_partitioned_inverse_auto(A, block, nthreads, device, &P) != DCORA_OK) return false;
    auto img = std::make_shared<SpImage>();
    if (img->upload(P, &sink) != DCORA_OK) return false;
    P = PartInvHost();  // the host image is no longer needed
    SparsePrecond sp;
    if (sp.attach(img, nrhs) != DCORA_OK) return false;
    const size_t N = (size_t)A.n * nrhs;
    DevBuf<double> dB, dX;
    if (dB.alloc(N) != hipSuccess || dX.alloc(N) != hipSuccess) return false;
    if (hipMemcpy(dB.p, B, N * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return false;
    sp.apply(nullptr, nrhs, buf1(dB.p), dX.p, Gate{});
    if (hipMemcpy(X, dX.p, N * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return false;
    return true;
  };
}

int build_partitioned_inverse_auto(const HostCsr &A, int block, int nthreads, int device, PartInvHost *out) {
  const bool host_factor = env::factor_on_host();
  if (host_factor) {
    const bool okh = build_partitioned_inverse(A, block, nthreads, out);
    return okh ? DCORA_OK : (out->weights_ok ? DCORA_ERR_NOT_PD : DCORA_ERR_HIP);
  }
  PiecewiseFactor F;
  // a sink that forms the weights on the device wants the sources left there
  F.want_device_sources = out->sink && out->sink->wants_device_sources();
  const auto t0 = std::chrono::steady_clock::now();
  const int rc = device_chol_piecewise_factor(A, block, nd_top_default(), device, &F);
  if (rc) return rc;
  const auto t1 = std::chrono::steady_clock::now();
  const bool ok = build_partitioned_inverse_from(A, F, nthreads, out);
  if (env::init_timing())
    fprintf(stderr, "[precond] factor on the device %.1f ms, partitioned inverse on the host %.1f ms\n",
            std::chrono::duration<double, std::milli>(t1 - t0).count(),
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count());
  if (!ok && !out->weights_ok) return DCORA_ERR_HIP;  // the weight sink failed (device or pinned memory): not a verdict
  return ok ? DCORA_OK : DCORA_ERR_NOT_PD;
}

}  // namespace dcora
