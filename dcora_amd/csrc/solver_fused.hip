// Fused truncated-CG kernels for the SE layout (pose graphs), r <= 8: three launches per tCG iteration.
//
//   A  k_fused_hess     delta = -z + beta delta  (on the fly while gathering),  W = delta Q  (CSR rows staged in
//                       LDS),  H delta = Proj_X(W - delta S),  partial <delta, H delta>
//   B  k_fused_precond  alpha / boundary test, eta += a delta, H eta += a H delta, r += a H delta, partial |r|^2,
//                       Z_s = r (Q + reg I)^-1 restricted to a slice of rows (split-K over the dense inverse)
//   C  k_fused_finish   residual stopping rule, z = Proj_X(sum_s Z_s), partial <z, r>
//
// Every global reduction of the CG recurrence sits exactly on a kernel boundary, so an iteration costs three
// dependent launches instead of six.  Per-pose arithmetic uses 8 lanes per pose: lane t
// of a group owns row t of the pose's r x (d+1) block, d x d Gram matrices are reduced with 3 xor-shuffles.
#include <algorithm>
#include <stdexcept>
#include <atomic>
#include <type_traits>
#include <vector>
#include <cstdlib>
#include <cstring>

#include "kernels.h"

namespace dcora {

namespace {

constexpr int GW = 8;  // lanes per pose

__device__ __forceinline__ bool f_gated(const SolverCtl *ctl, int seq, int gate) {
  if (seq > ctl->outer_done_stamp) return true;
  if (gate == 2 && seq > ctl->tcg_done_stamp) return true;
  return false;
}
// Wave-wide sum, same value in every lane.  Data-parallel-primitive moves inside the 16-lane rows (xor 1, xor 2,
// mirror of 8, mirror of 16: no LDS crossbar round trips as with __shfl_xor / ds_bpermute), then the four row sums
// are read as scalars.  Fixed order => reproducible.
template <int CTRL>
__device__ __forceinline__ double f_dpp(double v) { return dpp_move<CTRL>(v); }
__device__ __forceinline__ double f_wave_sum(double v) {
#ifdef DCORA_WAVE_SUM_SHUFFLE  // A/B switch: the butterfly over ds_bpermute
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
#endif
  return wave_sum_dpp(v);  // kernels.h
}
__device__ __forceinline__ double f_block_sum(double v, double *sm) {
  v = f_wave_sum(v);
  const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm[w] = v;
  __syncthreads();
  double t = 0;
  for (int i = 0; i < nw; ++i) t += sm[i];
  return t;
}
// Up to 64 partials: every wave loads them itself (lane l takes partial l) and sums them with f_wave_sum, so the
// total is known in every wave without a barrier or an LDS exchange; more partials go through the block reduction.
// Up to 256 partials without a barrier: every wave loads all of them (lane l takes partials l, l + 64, l + 128, l + 192:
// four predicated loads requested together) and sums them with f_wave_sum -- __syncthreads() drains vmcnt, i.e. it
// would wait for every gather the kernel has in flight behind the partials (measured in k_fused_hess with the 250
// partials of k_fused_pc: 1.7 us at the reduction).  f_partial4_load / f_partial4_total; beyond 256 the block path.
__device__ __forceinline__ double f_partial4_load(const double *__restrict__ p, int np) {
  const int l = (int)(threadIdx.x & 63u);
  const double a = (l < np) ? p[l] : 0.0, b = (l + 64 < np) ? p[l + 64] : 0.0;
  const double c = (l + 128 < np) ? p[l + 128] : 0.0, d = (l + 192 < np) ? p[l + 192] : 0.0;
  return (a + b) + (c + d);
}
// f_partial_index is the index a thread loads, f_partial_total the matching reduction.
__device__ __forceinline__ int f_partial_index(int np) { return np <= 64 ? (int)(threadIdx.x & 63u) : (int)threadIdx.x; }
__device__ __forceinline__ double f_partial_total(double v, int np, double *sm) {
  return np <= 64 ? f_wave_sum(v) : f_block_sum(v, sm);
}
__device__ __forceinline__ void f_host_store(volatile int *p, int v) {
  __hip_atomic_store(const_cast<int *>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// sum over the 8 lanes of a pose group (every lane of the wave must take part).  The result is re-broadcast from
// the group's first lane: with FMA contraction the butterfly partial sums can differ in the last bit between
// lanes, and the Jacobi / Gram-Schmidt decisions taken from them must be identical across the group.
// DPP row operations (xor 1, xor 2 inside the quads, mirror of the 8 lanes) instead of __shfl / ds_bpermute: the
// Jacobi sweeps of row_polar take dozens of these sums per pose.  The argument is pinned as an already-rounded value
// first, so every add below combines two rounded numbers and, addition being commutative, the eight lanes end with
// bitwise the same sum (same tree as the xor butterfly it replaces).
__device__ __forceinline__ double grp_sum(double v) {
#ifdef DCORA_GRP_SUM_SHUFFLE  // A/B switch
  v += __shfl_xor(v, 1, 64);
  v += __shfl_xor(v, 2, 64);
  v += __shfl_xor(v, 4, 64);
  return __shfl(v, (int)(threadIdx.x & 63u & ~7u), 64);
#endif
  asm volatile("" : "+v"(v));
  v += f_dpp<0xB1>(v);   // quad_perm [1,0,3,2]
  asm volatile("" : "+v"(v));
  v += f_dpp<0x4E>(v);   // quad_perm [2,3,0,1]
  asm volatile("" : "+v"(v));
  v += f_dpp<0x141>(v);  // row_half_mirror: lane i <- lane 7 - i of its group of eight
  return v;
}

template <int D>
struct Row {
  double e[D + 1];  // D rotation entries + the translation entry of row t
};
template <int D>
__device__ __forceinline__ void ld_row(const double *__restrict__ p, int r, int t, bool active, Row<D> &R) {
#pragma unroll
  for (int a = 0; a <= D; ++a) R.e[a] = active ? p[a * r + t] : 0.0;
}
template <int D>
__device__ __forceinline__ void st_row(double *__restrict__ p, int r, int t, bool active, const Row<D> &R) {
  if (active)
#pragma unroll
    for (int a = 0; a <= D; ++a) p[a * r + t] = R.e[a];
}
// S = sym(Y^T E) over the rotation columns (group-wide result in every lane)
template <int D>
__device__ __forceinline__ void grp_sym_gram(const Row<D> &Y, const Row<D> &E, double (&S)[D][D]) {
#pragma unroll
  for (int a = 0; a < D; ++a)
#pragma unroll
    for (int b = a; b < D; ++b) {
      const double s = grp_sum(0.5 * (Y.e[a] * E.e[b] + Y.e[b] * E.e[a]));
      S[a][b] = s;
      S[b][a] = s;
    }
}
// V_rot <- V_rot - A_rot S
template <int D>
__device__ __forceinline__ void row_sub_AS(Row<D> &V, const Row<D> &A, const double (&S)[D][D]) {
#pragma unroll
  for (int b = 0; b < D; ++b) {
    double s = 0;
#pragma unroll
    for (int a = 0; a < D; ++a) s += A.e[a] * S[a][b];
    V.e[b] -= s;
  }
}
template <int D>
__device__ __forceinline__ void row_tangent(const Row<D> &Y, Row<D> &V) {
  double S[D][D];
  grp_sym_gram<D>(Y, V, S);
  row_sub_AS<D>(V, Y, S);
}
// QF retraction of the rotation part (modified Gram-Schmidt, one re-orthogonalisation pass)
template <int D>
__device__ __forceinline__ void row_qf(Row<D> &A) {
#pragma unroll
  for (int j = 0; j < D; ++j) {
#pragma unroll
    for (int pass = 0; pass < 2; ++pass)
#pragma unroll
      for (int c = 0; c < D; ++c)
        if (c < j) {
          const double s = grp_sum(A.e[c] * A.e[j]);
          A.e[j] -= s * A.e[c];
        }
    const double nn = grp_sum(A.e[j] * A.e[j]);
    A.e[j] *= 1.0 / sqrt(nn);
  }
}
// polar factor of the rotation part.  The RBCD++ sequences hand in blocks that are orthonormal up to the size of a
// step ((1 - alpha) x + alpha v, v + gamma (x - y): ref src/Agent.cpp:1196-1214 project them with a thin SVD), so the
// common case runs on the Gram matrix: G = A^T A by D (D + 1) / 2 group sums, Z = G^(-1/2) by the coupled Newton-Schulz
// iteration (Y <- Y T, Z <- T Z, T = (3 I - Z Y) / 2: products of D x D symmetric matrices in registers, no division, no
// square root, nothing crosses lanes), A <- A Z.  G is the same in the eight lanes of a pose, so they take the same
// steps; the iteration converges quadratically for |I - G| < 1 and the Gram form loses nothing at condition numbers
// near one.  Blocks further than kPolarGramRadius from orthonormal (set_X of a rough point, a long step) take the
// one-sided Jacobi sweeps below, as every block did before: k_g_nesterov over the 100k lattice 36 us with Jacobi for all
// (the sweeps' divisions and square roots, 12 waves deep per SIMD, not the memory), with this form see DESIGN.md.
constexpr double kPolarGramRadius = 0.25;
template <int D>
__device__ __forceinline__ void sym_mul(const double (&A)[D][D], const double (&B)[D][D], double (&C)[D][D]) {
  // the product of two commuting symmetric matrices is symmetric: upper triangle, mirrored
#pragma unroll
  for (int a = 0; a < D; ++a)
#pragma unroll
    for (int b = a; b < D; ++b) {
      double s = 0;
#pragma unroll
      for (int c = 0; c < D; ++c) s += A[a][c] * B[c][b];
      C[a][b] = s;
      C[b][a] = s;
    }
}
template <int D>
__device__ __forceinline__ void row_polar(Row<D> &A, bool live) {
  double G[D][D];
  double dist2 = 0;
#pragma unroll
  for (int a = 0; a < D; ++a)
#pragma unroll
    for (int b = a; b < D; ++b) {
      const double s = grp_sum(A.e[a] * A.e[b]);
      G[a][b] = s;
      G[b][a] = s;
      const double e = s - (a == b ? 1.0 : 0.0);
      dist2 += (a == b ? 1.0 : 2.0) * e * e;
    }
  const bool gram = live && dist2 <= kPolarGramRadius * kPolarGramRadius;  // the same in the lanes of a pose
  if (gram) {
    double Y[D][D], Z[D][D];
#pragma unroll
    for (int a = 0; a < D; ++a)
#pragma unroll
      for (int b = 0; b < D; ++b) {
        Y[a][b] = G[a][b];
        Z[a][b] = (a == b) ? 1.0 : 0.0;
      }
    for (int it = 0; it < 12; ++it) {
      double T[D][D], ZY[D][D], Yn[D][D], Zn[D][D];
      sym_mul<D>(Z, Y, ZY);
      double e2 = 0;
#pragma unroll
      for (int a = 0; a < D; ++a)
#pragma unroll
        for (int b = 0; b < D; ++b) {
          const double rr = (a == b ? 1.0 : 0.0) - ZY[a][b];
          e2 += rr * rr;
          T[a][b] = (a == b ? 1.0 : 0.0) + 0.5 * rr;
        }
      sym_mul<D>(Y, T, Yn);
      sym_mul<D>(T, Z, Zn);
#pragma unroll
      for (int a = 0; a < D; ++a)
#pragma unroll
        for (int b = 0; b < D; ++b) {
          Y[a][b] = Yn[a][b];
          Z[a][b] = Zn[a][b];
        }
      if (e2 < 1e-16) break;  // |I - Z Y| < 1e-8 before this step: below 1e-16 after it
    }
    double o[D];
#pragma unroll
    for (int c = 0; c < D; ++c) {
      double s = 0;
#pragma unroll
      for (int j = 0; j < D; ++j) s += A.e[j] * Z[j][c];
      o[c] = s;
    }
#pragma unroll
    for (int c = 0; c < D; ++c) A.e[c] = o[c];
  }
  // one-sided Jacobi for the others.  The sweep loop is wave-uniform (the group sums are cross-lane operations), but a
  // pose stops rotating once ITS sweep has converged: the result of a pose must not depend on which poses share its
  // wave (a launch over one agent's poses and a launch over the whole graph place a pose next to different neighbours).
  bool settled = !live || gram;
  if (__all(settled)) return;
  const bool jacobi = !settled;
  double Vm[D][D];
#pragma unroll
  for (int a = 0; a < D; ++a)
#pragma unroll
    for (int b = 0; b < D; ++b) Vm[a][b] = (a == b) ? 1.0 : 0.0;
  Row<D> B = A;
  for (int sweep = 0; sweep < 40; ++sweep) {
    double off = 0;
#pragma unroll
    for (int p = 0; p < D - 1; ++p)
#pragma unroll
      for (int q = p + 1; q < D; ++q) {
        const double app = grp_sum(B.e[p] * B.e[p]);
        const double aqq = grp_sum(B.e[q] * B.e[q]);
        const double apq = grp_sum(B.e[p] * B.e[q]);
        const double sc = sqrt(app * aqq);
        if (!settled && fabs(apq) > 1e-16 * sc && fabs(apq) > 1e-300) {
          off = fmax(off, fabs(apq) / sc);
          const double zeta = (aqq - app) / (2.0 * apq);
          const double tt = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
          const double cs = 1.0 / sqrt(1.0 + tt * tt), sn = cs * tt;
          const double x = B.e[p], y = B.e[q];
          B.e[p] = cs * x - sn * y;
          B.e[q] = sn * x + cs * y;
#pragma unroll
          for (int i = 0; i < D; ++i) {
            const double vx = Vm[p][i], vy = Vm[q][i];
            Vm[p][i] = cs * vx - sn * vy;
            Vm[q][i] = sn * vx + cs * vy;
          }
        }
      }
    settled = settled || off < 1e-15;
    if (__all(settled)) break;
  }
  double u[D];
#pragma unroll
  for (int j = 0; j < D; ++j) {
    const double nn = grp_sum(B.e[j] * B.e[j]);
    u[j] = B.e[j] * (nn > 0 ? 1.0 / sqrt(nn) : 0.0);
  }
  if (jacobi)
#pragma unroll
    for (int c = 0; c < D; ++c) {
      double s = 0;
#pragma unroll
      for (int j = 0; j < D; ++j) s += u[j] * Vm[j][c];
      A.e[c] = s;
    }
}

constexpr int kHessTile = 1536;              // nnz staged per pass (18 KiB of LDS)
constexpr int kPosesPerBlock = kBlock / GW;  // 32 (pure per-pose kernels)
constexpr int kBsrTile = 160;                // matrix blocks staged per pass (20 KiB at (d+1)^2 = 16)

// poses per block of the two-phase kernels: phase 1 runs one thread per output element (pose, column, row),
// phase 2 eight lanes per pose
__host__ __device__ inline int fused_pb(int r, int dh) {
  const int pb = kBlock / (dh * r);
  return pb > kPosesPerBlock ? kPosesPerBlock : pb;
}

// ------------------------------------------------------------------------------------------------------
// A: Hessian-vector product of the tCG direction, with the direction update folded into the gather.
//    Phase 1: W = delta Q, one thread per output element, CSR rows of the block staged in LDS.
//    Phase 2: H delta = Proj_X(W - delta S) with 8 lanes per pose, operands handed over through LDS.
// ------------------------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(kBlock) void k_fused_hess(ManiDesc m, CsrDev Q, const double *__restrict__ z,
                                                       const double *__restrict__ d_old,
                                                       double *__restrict__ d_new, Buf2 Xb, Buf2 Sb,
                                                       double *__restrict__ Hd, const double *__restrict__ p3,
                                                       int np3, double *__restrict__ p1, SolverCtl *ctl, int seq,
                                                       int iter) {
  // control scalars first (one batch of scalar loads); the gate itself is evaluated after the data loads below
  // have been issued, so a kernel pays one memory round trip, not two
  const int par = iter & 1;
  const int st_o = ctl->outer_done_stamp, st_t = ctl->tcg_done_stamp, cur = ctl->cur & 1;
  const double c_zr = ctl->z_r[par ^ 1], c_alpha = ctl->alpha, c_dPd = ctl->d_Pd[par ^ 1], c_ePd = ctl->e_Pd[par ^ 1],
               c_ePen = ctl->e_Pe_n;
  __shared__ int s_ci[kHessTile];
  __shared__ double s_v[kHessTile];
  __shared__ double s_W[kBlock], s_D[kBlock];
  __shared__ double s_red[16];
  constexpr int DH = D + 1;
  const int r = m.r;
  const int PB = fused_pb(r, DH);
  const int pose0 = blockIdx.x * PB;
  const int npose = min(PB, m.n - pose0);
  const int j0 = pose0 * DH, ncol = npose * DH, nout = ncol * r;
  // ---- independent loads first: first CSR tile into LDS, own entries, pose operands (latency overlaps the
  //      dependent scalar prologue below) ----
  const int e = threadIdx.x;
  const bool act = e < nout;
  const int lc = e / r, t = e - lc * r;
  const int j = j0 + lc;
  const int pbeg = Q.rp[j0], pend = Q.rp[j0 + ncol];
  const int myb = act ? Q.rp[j] : 0, mye = act ? Q.rp[j + 1] : 0;
  // <z, r> partials first (predicated loads; the loop below only runs for > 256 partials)
  const bool p3_wave = np3 <= 256;
  const int pi3 = f_partial_index(np3);
  double myp = p3_wave ? f_partial4_load(p3, np3) : ((pi3 < np3) ? p3[pi3] : 0.0);
  {
    // first tile of the matrix: all trips' loads are issued before any is stored to LDS (clamped index, straight
    // line), one memory round trip instead of one per 256 entries
    const int cnt = min(kHessTile, pend - pbeg);
    constexpr int SU = kHessTile / kBlock;
    int ci_r[SU];
    double v_r[SU];
    const int last = max(pend - 1, 0);
#pragma unroll
    for (int u = 0; u < SU; ++u) {
      const int i = min(pbeg + (int)threadIdx.x + u * kBlock, last);
      ci_r[u] = Q.ci[i];
      v_r[u] = Q.v[i];
    }
#pragma unroll
    for (int u = 0; u < SU; ++u) {
      const int i = threadIdx.x + u * kBlock;
      if (i < cnt) {
        s_ci[i] = ci_r[u];
        s_v[i] = v_r[u];
      }
    }
  }
  const size_t oown = (size_t)j * r + t;
  const double z_own = act ? z[oown] : 0.0;
  const double d_own = (act && iter > 0) ? d_old[oown] : 0.0;
  const int g = threadIdx.x >> 3, tt = threadIdx.x & (GW - 1);
  const bool pact = (g < npose) && (tt < r);
  const size_t o = (size_t)(pose0 + g) * DH * r;
  const double *__restrict__ X = Xb.p[cur];
  const double *__restrict__ Sblk = Sb.p[cur];
  Row<D> Y;
  ld_row<D>(X + o, r, tt, pact, Y);
  double S[D][D];
#pragma unroll
  for (int a = 0; a < D; ++a)
#pragma unroll
    for (int b = 0; b < D; ++b) S[a][b] = (g < npose) ? Sblk[(size_t)(pose0 + g) * D * D + a + b * D] : 0.0;
  for (int i = threadIdx.x + kBlock; i < np3; i += kBlock) myp += p3[i];
  if (seq > st_o || seq > st_t) return;  // solve or tCG already finished: no-op (uniform over the grid)
  __syncthreads();  // first tile staged
  // first batch of gathers: addresses do not depend on beta, so the loads are issued before the reduction
  constexpr int GB = 16;
  double ga[GB], gz[GB], gw[GB];
  const int lo0 = myb - pbeg;
  const int hi0 = min(mye, pbeg + kHessTile) - pbeg;
#pragma unroll
  for (int q = 0; q < GB; ++q) {
    const bool ok = act && (lo0 + q < hi0);
    const size_t oo = ok ? (size_t)s_ci[lo0 + q] * r + t : 0;
    gw[q] = ok ? s_v[lo0 + q] : 0.0;
    gz[q] = z[oo];
    ga[q] = (iter > 0) ? d_old[oo] : 0.0;
  }
  // ---- scalar recurrence (ROPTLIB tCG_TR): beta, e_Pd, d_Pd ----
  const double z_r_new = p3_wave ? f_wave_sum(myp) : f_partial_total(myp, np3, s_red);
  double beta = 0;
  if (iter > 0) beta = z_r_new / c_zr;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    if (iter == 0) {
      ctl->z_r[0] = z_r_new;
      ctl->d_Pd[0] = z_r_new;
      ctl->e_Pe[0] = 0;
      ctl->e_Pd[0] = 0;
    } else {
      ctl->z_r[par] = z_r_new;
      ctl->e_Pd[par] = beta * (c_ePd + c_alpha * c_dPd);
      ctl->d_Pd[par] = z_r_new + beta * beta * c_dPd;
      ctl->e_Pe[par] = c_ePen;
    }
  }
  // ---- phase 1 ----
  double acc = 0;
#pragma unroll
  for (int q = 0; q < GB; ++q) acc += gw[q] * (beta * ga[q] - gz[q]);
  for (int base = pbeg; base < pend; base += kHessTile) {
    const int cnt = min(kHessTile, pend - base);
    if (base != pbeg) {
      __syncthreads();
      for (int i = threadIdx.x; i < cnt; i += kBlock) {
        s_ci[i] = Q.ci[base + i];
        s_v[i] = Q.v[base + i];
      }
      __syncthreads();
    }
    int lo = max(myb, base) - base;
    const int hi = min(mye, base + cnt) - base;
    if (base == pbeg) lo += GB;  // already consumed above
    for (int p = lo; p < hi; p += 8) {
      double a8[8], b8[8], w8[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const bool ok = p + q < hi;
        const size_t oo = ok ? (size_t)s_ci[p + q] * r + t : 0;
        w8[q] = ok ? s_v[p + q] : 0.0;
        b8[q] = z[oo];
        a8[q] = (iter > 0) ? d_old[oo] : 0.0;
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) acc += w8[q] * (beta * a8[q] - b8[q]);
    }
  }
  if (act) {
    const double dn = (iter > 0) ? beta * d_own - z_own : -z_own;
    d_new[oown] = dn;
    s_W[e] = acc;
    s_D[e] = dn;
  }
  __syncthreads();
  // ---- phase 2 ----
  Row<D> V, W;
#pragma unroll
  for (int a = 0; a < DH; ++a) {
    W.e[a] = pact ? s_W[(g * DH + a) * r + tt] : 0.0;
    V.e[a] = pact ? s_D[(g * DH + a) * r + tt] : 0.0;
  }
  row_sub_AS<D>(W, V, S);
  row_tangent<D>(Y, W);
  st_row<D>(Hd + o, r, tt, pact, W);
  double dacc = 0;
#pragma unroll
  for (int a = 0; a < DH; ++a) dacc += V.e[a] * W.e[a];
  const double tot = f_block_sum(dacc, s_red);
  if (threadIdx.x == 0) p1[blockIdx.x] = tot;
}

// ------------------------------------------------------------------------------------------------------
// Cost and gradient of an RTR evaluation in ONE launch (small pose-graph blocks, CSR): EG = X Q + G with the partial
// dots {<X Q, X>, <X, G>} (what k_spmm<true> does), then RG = Proj_X(EG), S_i = sym(Y_i^T EG_i) and the partial |RG|^2
// (what k_g_rgrad does) -- phase 1 one thread per output element with the block's CSR rows staged in LDS, phase 2
// eight lanes per pose, operands handed over through LDS, exactly as k_fused_hess.  Saves one dependent launch per
// evaluation (4 per local solve, 1 per central evaluation).
// ------------------------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(kBlock) void k_fused_grad(ManiDesc m, CsrDev Q, Buf2 Xb, const double *__restrict__ G,
                                                       Buf2 EGb, Buf2 RGb, Buf2 Sb, int sel,
                                                       double *__restrict__ pA, double *__restrict__ pB,
                                                       double *__restrict__ posenorm, Gate g) {
  if (g.ctl && g.gate && f_gated(g.ctl, g.seq, g.gate)) return;
  __shared__ int s_ci[kHessTile];
  __shared__ double s_v[kHessTile];
  __shared__ double s_W[kBlock], s_X[kBlock];
  __shared__ double s_red[16];
  constexpr int DH = D + 1;
  const int idx = g.ctl ? ((g.ctl->cur ^ sel) & 1) : 0;
  const double *__restrict__ X = Xb.p[idx];
  double *__restrict__ EG = EGb.p[idx];
  double *__restrict__ RG = RGb.p[idx];
  double *__restrict__ Sblk = Sb.p[idx];
  const int r = m.r;
  const int PB = fused_pb(r, DH);
  const int pose0 = blockIdx.x * PB;
  const int npose = min(PB, m.n - pose0);
  const int j0 = pose0 * DH, ncol = npose * DH, nout = ncol * r;
  const int e = threadIdx.x;
  const bool act = e < nout;
  const int lc = e / r, t = e - lc * r;
  const int j = j0 + lc;
  const int pbeg = Q.rp[j0], pend = Q.rp[j0 + ncol];
  const int myb = act ? Q.rp[j] : 0, mye = act ? Q.rp[j + 1] : 0;
  const size_t oown = (size_t)j * r + t;
  const double x_own = act ? X[oown] : 0.0;
  const double g_own = (act && G) ? G[oown] : 0.0;
  // ---- phase 1: EG = X Q + G ----
  double acc = 0;
  for (int base = pbeg; base < pend; base += kHessTile) {
    const int cnt = min(kHessTile, pend - base);
    if (base != pbeg) __syncthreads();
    {
      constexpr int SU = kHessTile / kBlock;
      int ci_r[SU];
      double v_r[SU];
      const int last = max(pend - 1, 0);
#pragma unroll
      for (int u = 0; u < SU; ++u) {
        const int i = min(base + (int)threadIdx.x + u * kBlock, last);
        ci_r[u] = Q.ci[i];
        v_r[u] = Q.v[i];
      }
#pragma unroll
      for (int u = 0; u < SU; ++u) {
        const int i = threadIdx.x + u * kBlock;
        if (i < cnt) {
          s_ci[i] = ci_r[u];
          s_v[i] = v_r[u];
        }
      }
    }
    __syncthreads();
    const int lo = max(myb, base) - base, hi = min(mye, base + cnt) - base;
    for (int p = lo; p < hi; p += 8) {
      double b8[8], w8[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const bool ok = p + q < hi;
        const size_t oo = ok ? (size_t)s_ci[p + q] * r + t : 0;
        w8[q] = ok ? s_v[p + q] : 0.0;
        b8[q] = X[oo];
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) acc += w8[q] * b8[q];
    }
  }
  double d0 = 0, d1 = 0;
  if (act) {
    const double eg = acc + g_own;
    EG[oown] = eg;
    s_W[e] = eg;
    s_X[e] = x_own;
    d0 = acc * x_own;
    d1 = x_own * g_own;
  }
  __syncthreads();
  // ---- phase 2: RG = Proj_X(EG), S_i = sym(Y_i^T EG_i) ----
  const int gp = threadIdx.x >> 3, tt = threadIdx.x & (GW - 1);
  const bool pact = (gp < npose) && (tt < r);
  const int pose = pose0 + gp;
  const size_t o = (size_t)pose * DH * r;
  Row<D> Y, E;
#pragma unroll
  for (int a = 0; a < DH; ++a) {
    E.e[a] = pact ? s_W[(gp * DH + a) * r + tt] : 0.0;
    Y.e[a] = pact ? s_X[(gp * DH + a) * r + tt] : 0.0;
  }
  double S[D][D];
  grp_sym_gram<D>(Y, E, S);
  if (Sblk && gp < npose && tt == 0)
#pragma unroll
    for (int a = 0; a < D; ++a)
#pragma unroll
      for (int b = 0; b < D; ++b) Sblk[(size_t)pose * D * D + a + b * D] = S[a][b];
  row_sub_AS<D>(E, Y, S);
  double pa = 0;
#pragma unroll
  for (int a = 0; a < DH; ++a) pa += E.e[a] * E.e[a];
  if (posenorm) {
    const double ps = grp_sum(pa);
    if (gp < npose && tt == 0) posenorm[pose] = ps;
  }
  st_row<D>(RG + o, r, tt, pact, E);
  const double t0 = f_block_sum(d0, s_red);
  const double t1 = f_block_sum(d1, s_red);
  const double t2 = f_block_sum(pact ? pa : 0.0, s_red);
  if (threadIdx.x == 0) {
    pA[2 * blockIdx.x] = t0;
    pA[2 * blockIdx.x + 1] = t1;
    pB[blockIdx.x] = t2;
  }
}

// A on the block structure of Q (pose graphs large enough to carry the block-CSR copy): 8 lanes per pose from the
// start, so phase 1 leaves W in the lane layout phase 2 works in (no LDS hand-over), one gather of the neighbour's
// (d+1) r values per matrix block instead of (d+1)^2 scalar entries.  32 poses per workgroup.
template <int D>
__global__ __launch_bounds__(kBlock) void k_fused_hess_bsr(ManiDesc m, BsrDev A, const double *__restrict__ z,
                                                           const double *__restrict__ d_old,
                                                           double *__restrict__ d_new, Buf2 Xb, Buf2 Sb,
                                                           double *__restrict__ Hd, const double *__restrict__ p3,
                                                           int np3, double *__restrict__ p1, SolverCtl *ctl, int seq,
                                                           int iter) {
  const int par = iter & 1;
  const int st_o = ctl->outer_done_stamp, st_t = ctl->tcg_done_stamp, cur = ctl->cur & 1;
  const double c_zr = ctl->z_r[par ^ 1], c_alpha = ctl->alpha, c_dPd = ctl->d_Pd[par ^ 1], c_ePd = ctl->e_Pd[par ^ 1],
               c_ePen = ctl->e_Pe_n;
  constexpr int DH = D + 1, BS = DH * DH;
  __shared__ double s_bv[kBsrTile * BS];
  __shared__ int s_bc[kBsrTile];
  __shared__ double s_red[16];
  const int r = m.r;
  const int pose0 = blockIdx.x * kPosesPerBlock;
  const int g = threadIdx.x >> 3, tt = threadIdx.x & (GW - 1);
  const int pose = pose0 + g;
  const bool inr = pose < m.n;
  const bool pact = inr && (tt < r);
  const size_t o = (size_t)pose * DH * r;
  const int pi3 = f_partial_index(np3);
  double myp = (pi3 < np3) ? p3[pi3] : 0.0;
  const int pend_pose = min(m.n, pose0 + kPosesPerBlock);
  const int bbeg = A.bp[pose0], bend = A.bp[pend_pose];
  const int myb = inr ? A.bp[pose] : 0, mye = inr ? A.bp[pose + 1] : 0;
  // own rows of z / d_old, pose operands
  Row<D> Zo, Do, Y;
  ld_row<D>(z + o, r, tt, pact, Zo);
  if (iter > 0) {
    ld_row<D>(d_old + o, r, tt, pact, Do);
  } else {
#pragma unroll
    for (int a = 0; a < DH; ++a) Do.e[a] = 0.0;
  }
  const double *__restrict__ X = Xb.p[cur];
  const double *__restrict__ Sblk = Sb.p[cur];
  ld_row<D>(X + o, r, tt, pact, Y);
  double S[D][D];
#pragma unroll
  for (int a = 0; a < D; ++a)
#pragma unroll
    for (int b = 0; b < D; ++b) S[a][b] = inr ? Sblk[(size_t)pose * D * D + a + b * D] : 0.0;
  for (int i = threadIdx.x + kBlock; i < np3; i += kBlock) myp += p3[i];
  if (seq > st_o || seq > st_t) return;  // solve or tCG already finished: no-op (uniform over the grid)
  const double z_r_new = f_partial_total(myp, np3, s_red);
  double beta = 0;
  if (iter > 0) beta = z_r_new / c_zr;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    if (iter == 0) {
      ctl->z_r[0] = z_r_new;
      ctl->d_Pd[0] = z_r_new;
      ctl->e_Pe[0] = 0;
      ctl->e_Pd[0] = 0;
    } else {
      ctl->z_r[par] = z_r_new;
      ctl->e_Pd[par] = beta * (c_ePd + c_alpha * c_dPd);
      ctl->d_Pd[par] = z_r_new + beta * beta * c_dPd;
      ctl->e_Pe[par] = c_ePen;
    }
  }
  // ---- phase 1: W = d_new Q over the pose's block row, d_new = beta d_old - z formed in the gather ----
  Row<D> W, V;
#pragma unroll
  for (int a = 0; a < DH; ++a) W.e[a] = 0.0;
  for (int base = bbeg; base < bend; base += kBsrTile) {
    const int cnt = min(kBsrTile, bend - base);
    __syncthreads();
    {
      constexpr int SU = (kBsrTile * BS / 2 + kBlock - 1) / kBlock;
      const double2 *__restrict__ src = reinterpret_cast<const double2 *>(A.bv + (size_t)base * BS);
      const int n2 = cnt * BS / 2;
      double2 v_r[SU];
      const int bc_r = A.bc[base + min((int)threadIdx.x, cnt - 1)];
      if ((BS & 1) == 0) {
#pragma unroll
        for (int u = 0; u < SU; ++u) v_r[u] = src[min((int)threadIdx.x + u * kBlock, n2 - 1)];
#pragma unroll
        for (int u = 0; u < SU; ++u) {
          const int i = threadIdx.x + u * kBlock;
          if (i < n2) reinterpret_cast<double2 *>(s_bv)[i] = v_r[u];
        }
      } else {
        for (int i = threadIdx.x; i < cnt * BS; i += kBlock) s_bv[i] = A.bv[(size_t)base * BS + i];
      }
      if ((int)threadIdx.x < cnt) s_bc[threadIdx.x] = bc_r;
      for (int i = threadIdx.x + kBlock; i < cnt; i += kBlock) s_bc[i] = A.bc[base + i];
    }
    __syncthreads();
    const int lo = max(myb, base) - base, hi = min(mye, base + cnt) - base;
    for (int b = lo; b < hi; b += 4) {
      double xz[4][DH], xd[4][DH];
      int bb[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const bool ok = pact && (b + q < hi);
        bb[q] = (b + q < hi) ? b + q : b;
        const size_t oo = (size_t)s_bc[bb[q]] * DH * r + tt;
#pragma unroll
        for (int c = 0; c < DH; ++c) {
          xz[q][c] = ok ? z[oo + c * r] : 0.0;
          xd[q][c] = (ok && iter > 0) ? d_old[oo + c * r] : 0.0;
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const double *__restrict__ Bq = s_bv + bb[q] * BS;
#pragma unroll
        for (int a = 0; a < DH; ++a) {
          double s = 0;
#pragma unroll
          for (int c = 0; c < DH; ++c) s += Bq[c * DH + a] * (beta * xd[q][c] - xz[q][c]);
          W.e[a] += s;
        }
      }
    }
  }
#pragma unroll
  for (int a = 0; a < DH; ++a) V.e[a] = (iter > 0) ? beta * Do.e[a] - Zo.e[a] : -Zo.e[a];
  st_row<D>(d_new + o, r, tt, pact, V);
  // ---- phase 2 ----
  row_sub_AS<D>(W, V, S);
  row_tangent<D>(Y, W);
  st_row<D>(Hd + o, r, tt, pact, W);
  double dacc = 0;
#pragma unroll
  for (int a = 0; a < DH; ++a) dacc += V.e[a] * W.e[a];
  if (!pact) dacc = 0;
  const double tot = f_block_sum(dacc, s_red);
  if (threadIdx.x == 0) p1[blockIdx.x] = tot;
}

// ------------------------------------------------------------------------------------------------------
// B: step length, vector updates and the dense preconditioner product, split over row slices of Minv.
//    first != 0: start of a tCG run (res = grad, eta = H eta = 0, no step).
//    The updated residual slice is staged once in LDS; each wave then streams whole rows of the symmetric
//    inverse with 16-byte loads (lane l owns output columns 2l, 2l+1 of the block's 128-column chunk),
//    eight rows in flight per lane, and reads the residual entries as LDS broadcasts.
// ------------------------------------------------------------------------------------------------------
constexpr int kJChunk = 128;  // output columns per block
constexpr int kRowChunk = 256;  // residual rows staged per pass

template <int RM, bool HAS_M>
__global__ __launch_bounds__(kBlock) void k_fused_precond(int r, int k, int ldm, int nsplit,
                                                          const double *__restrict__ Minv, Buf2 gradb,
                                                          const double *__restrict__ delta,
                                                          const double *__restrict__ Hd, double *__restrict__ eta,
                                                          double *__restrict__ Heta,
                                                          const double *__restrict__ res_old,
                                                          double *__restrict__ res_new,
                                                          double *__restrict__ Zpart, const double *__restrict__ p1,
                                                          int np1, double *__restrict__ p2, SolverCtl *ctl,
                                                          HostFlags *hf, int seq, int iter, int first, SpFold sf) {
  const int par = iter & 1;
  const int st_o = ctl->outer_done_stamp, st_t = ctl->tcg_done_stamp, cur = ctl->cur & 1;
  const double c_zr = ctl->z_r[par], c_dPd = ctl->d_Pd[par], c_ePe = ctl->e_Pe[par], c_ePd = ctl->e_Pd[par],
               c_Delta = ctl->Delta, c_ngf = ctl->ngf;
  __shared__ double s_red[16];
  __shared__ double s_buf[(kBlock / 64) * RM * kJChunk];  // residual slice, then the cross-wave reduction
  const long N = (long)r * k;
  // ---- row slice of this block / wave, and the first kPre rows of the inverse preloaded into registers: the
  //      loads do not depend on the step length, so their latency overlaps the scalar prologue ----
  constexpr int kPre = 16;
  const int njc = (k + kJChunk - 1) / kJChunk;
  const int jc = blockIdx.x % njc, s = blockIdx.x / njc;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int rows_per_split = (k + nsplit - 1) / nsplit;
  const int c_lo = min(k, s * rows_per_split), c_hi = min(k, c_lo + rows_per_split);
  const int col = jc * kJChunk + 2 * lane;
  const int cn0 = min(kRowChunk, c_hi - c_lo);
  const int per_wave0 = (cn0 + 3) / 4;
  const int w_lo0 = min(cn0, wave * per_wave0);
  const double *__restrict__ rsrc = first ? gradb.p[cur] : res_old;
  // own element of the vector updates and own entries of the residual slice: loaded before alpha is known
  const long i0 = (long)blockIdx.x * kBlock + threadIdx.x;
  const bool own = i0 < N;
  double o_h = 0, o_d = 0, o_eta = 0, o_Heta = 0, o_r = 0;
  if (own) {
    o_r = rsrc[i0];
    if (!first) {
      o_h = Hd[i0];
      o_d = delta[i0];
      o_eta = eta[i0];
      o_Heta = Heta[i0];
    }
  }
  constexpr int kStagePre = 2;  // staged residual entries preloaded per thread
  double st_r[kStagePre], st_h[kStagePre];
#pragma unroll
  for (int u = 0; u < kStagePre; ++u) {
    const int i = threadIdx.x + u * kBlock;
    const bool ok = i < cn0 * r;
    const size_t idx = (size_t)c_lo * r + (ok ? i : 0);
    st_r[u] = ok ? rsrc[idx] : 0.0;
    st_h[u] = (ok && !first) ? Hd[idx] : 0.0;
  }
  // <d, H d> partials: one predicated load per thread (a loop would wait for its loads inside the loop)
  const int pi1 = f_partial_index(np1);
  double myp = (!first && pi1 < np1) ? p1[pi1] : 0.0;
  // The rows of the inverse are requested AFTER the few words the step length needs: vector loads retire in issue
  // order, so the scalar prologue below (two block reductions, the vector updates) waits for those words only and
  // runs while the 64 KB of the slice are still in flight, instead of behind them.
  asm volatile("" ::: "memory");
  // Straight-line loads (row index clamped, value masked afterwards) so that the compiler can count them: behind
  // per-row branches it falls back to s_waitcnt vmcnt(0) at the first use of ANY loaded value.
  double2 pre[kPre];
  if (HAS_M) {
    const int col_c = min(col, ldm - 2);
#pragma unroll
    for (int q = 0; q < kPre; ++q) {
      const int row = min(c_lo + w_lo0 + q, k - 1);
      pre[q] = *reinterpret_cast<const double2 *>(Minv + (size_t)row * ldm + col_c);
    }
  }
  asm volatile("" ::: "memory");
  if (!first)
    for (int i = threadIdx.x + kBlock; i < np1; i += kBlock) myp += p1[i];  // only blocks of > 8192 poses get here
  if (seq > st_o || (!first && seq > st_t)) return;  // finished: no-op (uniform over the grid)
  double alpha = 0, step = 0;
  bool boundary = false;
  if (!first) {
    const double d_Hd = f_partial_total(myp, np1, s_red);
    const double z_r = c_zr, d_Pd = c_dPd, e_Pe = c_ePe, e_Pd = c_ePd;
    const double Delta = c_Delta;
    alpha = z_r / d_Hd;
    const double e_Pe_new = e_Pe + 2.0 * alpha * e_Pd + alpha * alpha * d_Pd;
    boundary = (d_Hd <= 0) || (e_Pe_new >= Delta * Delta);
    step = boundary ? (-e_Pd + sqrt(e_Pd * e_Pd + d_Pd * (Delta * Delta - e_Pe))) / d_Pd : alpha;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      ctl->alpha = alpha;
      ctl->e_Pe_n = e_Pe_new;
      if (boundary) {
        ctl->tcg_status = (d_Hd <= 0) ? 0 : 1;
        ctl->tcg_iters = iter + 1;
        ctl->inner_total += iter + 1;
        ctl->tcg_done_stamp = seq;
        f_host_store(&hf->tcg_done_seq, seq);
      } else {
        // the host enqueues what follows a B that goes on only once it knows (DeviceProblem::rtr_dev_fused)
        f_host_store(&hf->go_seq, seq);
      }
    }
  }
  if (first && blockIdx.x == 0 && threadIdx.x == 0) {
    ctl->norm_r0 = c_ngf;
    ctl->tcg_status = 4;
    ctl->tcg_iters = 0;
    ctl->tcg_done_stamp = INT_MAX;
  }
  // ---- element-wise updates (each element exactly once over the grid) ----
  double acc2 = 0;
  if (own) {
    if (first) {
      eta[i0] = 0;
      Heta[i0] = 0;
      res_new[i0] = o_r;
      if (!HAS_M && sf.y) sf.y[(size_t)sf.in_pos[i0 / r] * r + (i0 % r)] = o_r;
    } else {
      eta[i0] = o_eta + step * o_d;
      Heta[i0] = o_Heta + step * o_h;
      if (!boundary) {
        const double rr = o_r + alpha * o_h;
        res_new[i0] = rr;
        if (!HAS_M && sf.y) sf.y[(size_t)sf.in_pos[i0 / r] * r + (i0 % r)] = rr;
        acc2 += rr * rr;
      }
    }
  }
  for (long i = i0 + (long)gridDim.x * kBlock; i < N; i += (long)gridDim.x * kBlock) {
    if (first) {
      eta[i] = 0;
      Heta[i] = 0;
      res_new[i] = rsrc[i];
      if (!HAS_M && sf.y) sf.y[(size_t)sf.in_pos[i / r] * r + (i % r)] = rsrc[i];
    } else {
      const double h = Hd[i];
      eta[i] += step * delta[i];
      Heta[i] += step * h;
      if (!boundary) {
        const double rr = res_old[i] + alpha * h;
        res_new[i] = rr;
        if (!HAS_M && sf.y) sf.y[(size_t)sf.in_pos[i / r] * r + (i % r)] = rr;
        acc2 += rr * rr;
      }
    }
  }
  if (!first) {
    const double tot = f_block_sum(acc2, s_red);
    if (threadIdx.x == 0) p2[blockIdx.x] = tot;
  }
  if (boundary || !HAS_M) return;
  // ---- dense product slice: Z_s(:, j) = sum_{c in slice} r(:, c) Minv(c, j) ----
  double a0[RM], a1[RM];
#pragma unroll
  for (int t = 0; t < RM; ++t) a0[t] = a1[t] = 0;
  for (int c0 = c_lo; c0 < c_hi; c0 += kRowChunk) {
    const int cn = min(kRowChunk, c_hi - c0);
    __syncthreads();
    if (c0 == c_lo) {
#pragma unroll
      for (int u = 0; u < kStagePre; ++u) {
        const int i = threadIdx.x + u * kBlock;
        if (i < cn * r) s_buf[i] = st_r[u] + alpha * st_h[u];
      }
    }
    for (int i = threadIdx.x + (c0 == c_lo ? kStagePre * kBlock : 0); i < cn * r; i += kBlock) {
      const size_t idx = (size_t)c0 * r + i;
      double x = rsrc[idx];
      if (!first) x += alpha * Hd[idx];
      s_buf[i] = x;
    }
    __syncthreads();
    const int per_wave = (cn + 3) / 4;
    const int w_lo = min(cn, wave * per_wave), w_hi = min(cn, w_lo + per_wave);
    int c = w_lo;
    if (c0 == c_lo) {
      // rows preloaded before the prologue (those past the wave's range were clamped: masked here)
#pragma unroll
      for (int q = 0; q < kPre; ++q) {
        const bool ok = (w_lo + q < w_hi);
        if (!ok) pre[q].x = pre[q].y = 0.0;
#pragma unroll
        for (int t = 0; t < RM; ++t)
          if (t < r) {
            const double x = ok ? s_buf[(w_lo + q) * r + t] : 0.0;
            a0[t] += x * pre[q].x;
            a1[t] += x * pre[q].y;
          }
      }
      c = min(w_hi, w_lo + kPre);
    }
    const double *__restrict__ mp = Minv + (size_t)(c0 + c) * ldm + col;
    for (; c + 8 <= w_hi; c += 8) {
      double2 mm[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) mm[q] = *reinterpret_cast<const double2 *>(mp + (size_t)q * ldm);
      mp += 8 * (size_t)ldm;
#pragma unroll
      for (int q = 0; q < 8; ++q)
#pragma unroll
        for (int t = 0; t < RM; ++t)
          if (t < r) {
            const double x = s_buf[(c + q) * r + t];
            a0[t] += x * mm[q].x;
            a1[t] += x * mm[q].y;
          }
    }
    for (; c < w_hi; ++c) {
      const double2 m0 = *reinterpret_cast<const double2 *>(mp);
      mp += ldm;
#pragma unroll
      for (int t = 0; t < RM; ++t)
        if (t < r) {
          const double x = s_buf[c * r + t];
          a0[t] += x * m0.x;
          a1[t] += x * m0.y;
        }
    }
  }
  __syncthreads();
#pragma unroll
  for (int t = 0; t < RM; ++t) {
    s_buf[(wave * RM + t) * kJChunk + 2 * lane] = a0[t];
    s_buf[(wave * RM + t) * kJChunk + 2 * lane + 1] = a1[t];
  }
  __syncthreads();
  const int ncol = min(kJChunk, k - jc * kJChunk);
  double *__restrict__ zp = Zpart + (size_t)s * N + (size_t)jc * kJChunk * r;
  for (int e = threadIdx.x; e < ncol * r; e += kBlock) {
    const int cc = e / r, t = e - cc * r;
    double v = 0;
#pragma unroll
    for (int w = 0; w < kBlock / 64; ++w) v += s_buf[(w * RM + t) * kJChunk + cc];
    zp[e] = v;
  }
}

// ------------------------------------------------------------------------------------------------------
// C: residual stopping rule, z = Proj_X(sum of the split-K slices), partial <z, r>
//    Phase 1 sums the slices with one thread per element (coalesced), phase 2 projects with 8 lanes per pose.
// ------------------------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(kBlock) void k_fused_finish(ManiDesc m, int nsplit, Buf2 Xb,
                                                         const double *__restrict__ Zpart,
                                                         const double *__restrict__ res, double *__restrict__ z,
                                                         const double *__restrict__ p2, int np2,
                                                         double *__restrict__ p3, SolverCtl *ctl, HostFlags *hf,
                                                         int seq, int iter, int first, SpFold sf,
                                                         double *__restrict__ zraw) {
  const int st_o = ctl->outer_done_stamp, st_t = ctl->tcg_done_stamp, cur = ctl->cur & 1;
  const double c_n0 = ctl->norm_r0;
  const int c_max_inner = ctl->max_inner;
  __shared__ double s_red[16];
  __shared__ double s_Z[kBlock], s_R[kBlock];
  constexpr int DH = D + 1;
  const int r = m.r;
  const long N = (long)r * m.k;
  const int PB = fused_pb(r, DH);
  const int pose0 = blockIdx.x * PB;
  const int npose = min(PB, m.n - pose0);
  const int nout = npose * DH * r;
  const size_t base = (size_t)pose0 * DH * r;
  // Every load of the kernel is issued before the first value is consumed: the |r|^2 partials the stopping rule
  // needs (two predicated loads per thread; a loop would wait for each trip's load inside the loop), the pose rows,
  // the residual entry and the split-K slices.  One memory round trip instead of three dependent ones.
  const double *__restrict__ X = Xb.p[cur];
  const int g = threadIdx.x >> 3, tt = threadIdx.x & (GW - 1);
  const bool pact = (g < npose) && (tt < r);
  const size_t o = base + (size_t)g * DH * r;
  const int e = threadIdx.x;
  double myp = 0, myp2 = 0;
  if (!first) {
    myp = (e < np2) ? p2[e] : 0.0;
    myp2 = (e + kBlock < np2) ? p2[e + kBlock] : 0.0;
  }
  Row<D> Y, Zr, Rr;
  ld_row<D>(X + o, r, tt, pact, Y);
  // ---- phase 1 (independent of the stopping rule): slice sum, residual entry ----
  if (e < nout) {
    const double rres = res[base + e];
    double zs = 0;
    if (sf.y) {  // sparse preconditioner folded in: the value sits in the replay vector, at its final position
      const size_t ge = base + e;
      zs = sf.y[(size_t)sf.out_pos[ge / r] * r + (ge % r)];
    } else {
      double q[32];
#pragma unroll
      for (int u = 0; u < 32; ++u) q[u] = (u < nsplit) ? Zpart[(size_t)u * N + base + e] : 0.0;
#pragma unroll
      for (int u = 0; u < 32; ++u) zs += q[u];
      for (int s = 32; s < nsplit; ++s) zs += Zpart[(size_t)s * N + base + e];
    }
    s_Z[e] = zs;
    s_R[e] = rres;
    // z0 = P grad of an RTR iteration, unprojected: a rejected step leaves the iterate and its gradient where they
    // were, and the next iteration starts from this copy instead of another application of the preconditioner
    if (zraw) zraw[base + e] = zs;
  }
  myp += myp2;
  if (!first)
    for (int i = e + 2 * kBlock; i < np2; i += kBlock) myp += p2[i];
  if (seq > st_o || (!first && seq > st_t)) return;  // finished: no-op (uniform over the grid)
  if (!first) {
    const double nr = sqrt(f_block_sum(myp, s_red));
    const double n0 = c_n0;
    const double kappa = 0.1, tempnum = n0;  // theta = 1
    if (nr <= n0 * fmin(tempnum, kappa)) {
      if (blockIdx.x == 0 && threadIdx.x == 0) {
        ctl->tcg_status = (kappa < tempnum) ? 2 : 3;
        ctl->tcg_iters = iter + 1;
        ctl->inner_total += iter + 1;
        ctl->tcg_done_stamp = seq;
        f_host_store(&hf->tcg_done_seq, seq);
      }
      return;
    }
  }
  __syncthreads();
  // ---- phase 2 ----
#pragma unroll
  for (int a = 0; a < DH; ++a) {
    Zr.e[a] = pact ? s_Z[(g * DH + a) * r + tt] : 0.0;
    Rr.e[a] = pact ? s_R[(g * DH + a) * r + tt] : 0.0;
  }
  row_tangent<D>(Y, Zr);
  st_row<D>(z + o, r, tt, pact, Zr);
  double acc = 0;
#pragma unroll
  for (int a = 0; a < DH; ++a) acc += Zr.e[a] * Rr.e[a];
  const double tot = f_block_sum(acc, s_red);
  if (threadIdx.x == 0) p3[blockIdx.x] = tot;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    if (!first && iter + 1 >= c_max_inner) {  // inner loop exhausted: status stays TR_MAXITER
      ctl->tcg_iters = iter + 1;
      ctl->inner_total += iter + 1;
      ctl->tcg_done_stamp = seq;
      f_host_store(&hf->tcg_done_seq, seq);
    }
    f_host_store(&hf->last_seq_done, seq);
  }
}

// ------------------------------------------------------------------------------------------------------
// group-style versions of the per-outer-iteration kernels (SE layout)
// ------------------------------------------------------------------------------------------------------
// B + C in ONE launch for the dense preconditioner (k <= 8000): no split-K, so no slice sum and no boundary between
// the product and its projection.  A workgroup owns PB whole poses = PB (d+1) output columns of z.  By symmetry of
// the inverse, column j of (Q + reg I)^-1 is row j: the workgroup streams its PB (d+1) rows (contiguous, 16-byte
// loads, wave w takes rows w, w + 4, ...) against the WHOLE updated residual, which every workgroup rebuilds for
// itself in LDS from r_old and H delta (2 r k doubles from L2 -- a quarter more bytes than the inverse itself, but
// from the cache level with four times the bandwidth).  Because every workgroup holds the whole residual it also
// knows |r|^2 (same summation order everywhere, bitwise the same value), so the stopping rule needs no kernel
// boundary either.  Per tCG iteration: A, then this kernel -- two dependent launches instead of three.
// ------------------------------------------------------------------------------------------------------
// 16-byte buffer load: one descriptor (SGPRs) per array, per-lane byte offset in ONE VGPR, the uniform part of the
// address in an SGPR -- instead of a 64-bit VGPR address per load in flight; out-of-range dwords read as zero
typedef unsigned pc_v4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ double2 pc_ld16(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) {
  const pc_v4u v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0);
  return __builtin_bit_cast(double2, v);
}
// sum over each 16-lane row, same value in the row's lanes (the DPP half of wave_sum_dpp)
__device__ __forceinline__ double row16_sum_dpp(double v) {
  v += dpp_move<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_move<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_move<0x141>(v);  // row_half_mirror
  v += dpp_move<0x140>(v);  // row_mirror
  return v;
}
constexpr int kPcBlock = 256;  // 4 waves: wave w takes the 128-column steps w, w + 4, ... of ALL the workgroup's rows
constexpr int kPcNW = kPcBlock / 64;
constexpr int kPcSB = 25;      // 16-byte loads of each staged operand in flight per thread and batch
constexpr int kPcLoads = 32;   // 16-byte loads of the inverse's rows in flight per lane and batch

// R: the relaxation rank when it is known at compile time (no predication in the inner loops), 0 = run-time r <= 8.
// MULTI: the residual does not fit one LDS chunk (k > chk): the loop over further chunks is compiled in.
template <int D, int PB, int R, bool MULTI>
__global__ __launch_bounds__(kPcBlock) void k_fused_pc(ManiDesc m, int ldm, int chk, const double *__restrict__ Minv,
                                                       Buf2 gradb, Buf2 Xb, const double *__restrict__ delta,
                                                       const double *__restrict__ Hd, double *__restrict__ eta,
                                                       double *__restrict__ Heta, const double *__restrict__ res_old,
                                                       double *__restrict__ res_new, double *__restrict__ z,
                                                       const double *__restrict__ p1, int np1,
                                                       double *__restrict__ p3, SolverCtl *ctl, HostFlags *hf, int seq,
                                                       int iter, int first, double *__restrict__ pC) {
  constexpr int DH = D + 1, NR = PB * DH, RM = R ? R : 8;
  constexpr int MB = kPcLoads / NR;  // steps of a wave per batch: MB * NR 16-byte loads of the inverse in flight
  extern __shared__ double s_res[];  // the residual chunk, column-major as in memory: cpad * r doubles
  __shared__ double s_P[NR * RM + 1][4 * kPcNW];  // row sums (16 lanes each) of the product columns and of |r|^2
  __shared__ double s_Z[NR * RM + 1], s_R[NR * RM];
  __shared__ double s_red[16];
  const int par = iter & 1;
  const int st_o = ctl->outer_done_stamp, st_t = ctl->tcg_done_stamp, cur = ctl->cur & 1;
  const double c_zr = ctl->z_r[par], c_dPd = ctl->d_Pd[par], c_ePe = ctl->e_Pe[par], c_ePd = ctl->e_Pd[par],
               c_Delta = ctl->Delta, c_ngf = ctl->ngf, c_n0 = ctl->norm_r0;
  const int c_max_inner = ctl->max_inner;
  const int r = R ? R : m.r, k = m.k;
  const int pose0 = blockIdx.x * PB;
  const int npose = min(PB, m.n - pose0);
  const int j0 = pose0 * DH, nrow = npose * DH;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const double *__restrict__ rsrc = first ? gradb.p[cur] : res_old;
  // ---- loads first, in the order their values are needed: the <d, H d> partials and the workgroup's own elements,
  //      the first batch of the residual operands, the first batch of the inverse's rows.  None depends on the step
  //      length: their latency overlaps the scalar prologue.
  const int e = threadIdx.x;
  const bool own = e < nrow * r;
  const size_t oown = (size_t)j0 * r + e;
  const int pi1 = (np1 <= 64) ? lane : (int)threadIdx.x;
  // np1 < 0: timing form (time_precond only): <d, H d> is taken from the control block, as it would be if the
  // PRODUCER's last workgroup had summed its partials and stored the scalar -- measured in round 3: it saves nothing
  double myp = (!first && np1 >= 0 && pi1 < np1) ? p1[pi1] : 0.0;
  double o_r = 0, o_h = 0, o_d = 0, o_eta = 0, o_Heta = 0;
  if (own) {
    o_r = rsrc[oown];
    if (!first) {
      o_h = Hd[oown];
      o_d = delta[oown];
      o_eta = eta[oown];
      o_Heta = Heta[oown];
    }
  }
  const int g = threadIdx.x >> 3, tt = threadIdx.x & (GW - 1);
  const bool pact = (g < npose) && (tt < r);
  const size_t o = (size_t)(pose0 + min(g, npose - 1)) * DH * r;
  Row<D> Y;
  ld_row<D>(Xb.p[cur] + o, r, tt, pact, Y);
  const unsigned vec_bytes = (unsigned)((size_t)r * k * sizeof(double));
  const __amdgpu_buffer_rsrc_t rs_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(rsrc), 0, vec_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_h =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(first ? rsrc : Hd), 0, vec_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_m = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<double *>(Minv), 0, (unsigned)((size_t)k * ldm * sizeof(double)), 0x00020000);
  const unsigned voff_t = threadIdx.x * 16u, voff_l = (unsigned)lane * 16u;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  double2 xr0[kPcSB], xh0[kPcSB];
#pragma unroll
  for (int u = 0; u < kPcSB; ++u) {
    xr0[u] = pc_ld16(rs_r, voff_t, (unsigned)u * kPcBlock * 16u);
    if (!first) xh0[u] = pc_ld16(rs_h, voff_t, (unsigned)u * kPcBlock * 16u);
  }
  // row q of the workgroup, columns 2 lane, 2 lane + 1 of step (wave + kPcNW * u); rows past the last pose read the
  // rows that follow (or zeros past the end of the matrix) and are never used
  double2 pre[MB][NR];
#pragma unroll
  for (int u = 0; u < MB; ++u) {
#pragma unroll
    for (int q = 0; q < NR; ++q)
      pre[u][q] = pc_ld16(rs_m, voff_l, (unsigned)(((size_t)(j0 + q) * ldm + (size_t)(wave_u + kPcNW * u) * 128) * 8));
  }
  asm volatile("" ::: "memory");
  if (!first)
    for (int i = threadIdx.x + kPcBlock; i < np1; i += kPcBlock) myp += p1[i];
  if (seq > st_o || (!first && seq > st_t)) return;  // finished: no-op (uniform over the grid)
  // The launch in which a tCG run ends (boundary, negative curvature, residual rule, iteration cap -- every workgroup
  // knows: the tests are uniform over the grid) also takes the step: X_trial = Retr_X(eta) for the workgroup's own
  // poses and the partial <eta, grad>, <eta, H eta> of the model decrease, what a launch of k_g_retract did next
  // (pC != null).  Same retraction, same entries; the partials are per workgroup of PB poses instead of 32.
  auto retract_tail = [&]() {
    if (!pC) return;
    __syncthreads();  // (the z part of wave 0 may still be reading s_R)
    double a0 = 0, a1 = 0;
    if (own) {
      const double en = eta[oown], hn = Heta[oown], gr = gradb.p[cur][oown];  // eta, H eta: this thread's own stores
      a0 = en * gr;
      a1 = en * hn;
      const int lc = e / r, t = e - lc * r;
      s_R[lc * RM + t] = en;
    }
    __syncthreads();
    if (wave == 0) {
      Row<D> Yn = Y;
#pragma unroll
      for (int a = 0; a < DH; ++a) Yn.e[a] += 1.0 * (pact ? s_R[(g * DH + a) * RM + tt] : 0.0);
      row_qf<D>(Yn);
      st_row<D>(Xb.p[cur ^ 1] + o, r, tt, pact, Yn);
    }
    const double t0 = f_block_sum(a0, s_red);
    const double t1 = f_block_sum(a1, s_red);
    if (threadIdx.x == 0) {
      pC[2 * blockIdx.x] = t0;
      pC[2 * blockIdx.x + 1] = t1;
    }
  };
  // ---- step length / trust-region boundary (ROPTLIB tCG_TR), as in B ----
  double alpha = 0, step = 0;
  bool boundary = false;
  if (!first) {
    const double d_Hd = np1 < 0 ? c_dPd + 1.0 : ((np1 <= 64) ? f_wave_sum(myp) : f_block_sum(myp, s_red));
    alpha = c_zr / d_Hd;
    const double e_Pe_new = c_ePe + 2.0 * alpha * c_ePd + alpha * alpha * c_dPd;
    boundary = (d_Hd <= 0) || (e_Pe_new >= c_Delta * c_Delta);
    step = boundary ? (-c_ePd + sqrt(c_ePd * c_ePd + c_dPd * (c_Delta * c_Delta - c_ePe))) / c_dPd : alpha;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      ctl->alpha = alpha;
      ctl->e_Pe_n = e_Pe_new;
      if (boundary) {
        ctl->tcg_status = (d_Hd <= 0) ? 0 : 1;
        ctl->tcg_iters = iter + 1;
        ctl->inner_total += iter + 1;
        ctl->tcg_done_stamp = seq;
        f_host_store(&hf->tcg_done_seq, seq);
      } else {
        // the host enqueues what follows a B that goes on only once it knows (DeviceProblem::rtr_dev_fused)
        f_host_store(&hf->go_seq, seq);
      }
    }
  }
  if (first && blockIdx.x == 0 && threadIdx.x == 0) {
    ctl->norm_r0 = c_ngf;
    ctl->tcg_status = 4;
    ctl->tcg_iters = 0;
    ctl->tcg_done_stamp = INT_MAX;
  }
  if (own) {
    const int lc = e / r, t = e - lc * r;
    if (first) {
      eta[oown] = 0;
      Heta[oown] = 0;
      res_new[oown] = o_r;
      s_R[lc * RM + t] = o_r;
    } else {
      eta[oown] = o_eta + step * o_d;
      Heta[oown] = o_Heta + step * o_h;
      if (!boundary) {
        const double rr = fma(alpha, o_h, o_r);
        res_new[oown] = rr;
        s_R[lc * RM + t] = rr;
      }
    }
  }
  if (boundary) {
    retract_tail();
    return;
  }
  // ---- the whole updated residual through LDS, chunk by chunk; product with the workgroup's rows ----
  double acc[NR][RM];
#pragma unroll
  for (int q = 0; q < NR; ++q)
#pragma unroll
    for (int t = 0; t < RM; ++t) acc[q][t] = 0;
  double nrm2 = 0;
  // one batch of staged operands into the LDS image: a straight copy of r_old + alpha H delta
  auto stage = [&](const double2 (&xr)[kPcSB], const double2 (&xh)[kPcSB], int b0, int npair) {
#pragma unroll
    for (int u = 0; u < kPcSB; ++u) {
      const int i = b0 + u * kPcBlock + (int)threadIdx.x;
      if (i < npair) {
        double2 x = xr[u];
        if (!first) {
          x.x = fma(alpha, xh[u].x, x.x);
          x.y = fma(alpha, xh[u].y, x.y);
        }
        nrm2 = fma(x.x, x.x, nrm2);
        nrm2 = fma(x.y, x.y, nrm2);
        reinterpret_cast<double2 *>(s_res)[i] = x;
      }
    }
  };
  // the wave's steps of one batch: a lane's two columns of a step are 2 r contiguous doubles of the image
  auto rows = [&](const double2 (&mm)[MB][NR], int u0, int nstep) {
#pragma unroll
    for (int u = 0; u < MB; ++u) {
      const int sidx = wave_u + kPcNW * (u0 + u);
      if (sidx < nstep) {
        const double *__restrict__ xs = s_res + (size_t)(sidx * 128 + 2 * lane) * r;
        double x0[RM], x1[RM];
#pragma unroll
        for (int t = 0; t < RM; ++t) {
          x0[t] = (R || t < r) ? xs[t] : 0.0;
          x1[t] = (R || t < r) ? xs[r + t] : 0.0;
        }
#pragma unroll
        for (int t = 0; t < RM; ++t)
#pragma unroll
          for (int q = 0; q < NR; ++q) acc[q][t] = fma(x0[t], mm[u][q].x, fma(x1[t], mm[u][q].y, acc[q][t]));
      }
    }
  };
  // One chunk of the residual.  The first chunk consumes the loads requested before the prologue; it is written out
  // as its own instance (FIRST = true) so that those registers are dead in the loop over the remaining chunks.
  auto chunk = [&](int c0, auto first_chunk) {
    constexpr bool FIRST = decltype(first_chunk)::value;
    const int cn = min(chk, k - c0);
    const int cpad = ((cn + 127) / 128) * 128;  // zero-filled up to whole 128-column steps
    const long Nc = (long)cn * r;               // flat (column-major) length of the chunk: contiguous in memory
    const int npair = (int)((Nc + 1) >> 1);     // an odd tail reads its missing half as zero (buffer bounds)
    const unsigned cbase = (unsigned)((size_t)c0 * r * sizeof(double));
    int b0 = 0;
    if constexpr (FIRST) {
      stage(xr0, xh0, 0, npair);
      b0 = kPcSB * kPcBlock;
    } else {
      __syncthreads();
    }
#pragma unroll 1
    for (; b0 < npair; b0 += kPcSB * kPcBlock) {
      double2 xr[kPcSB], xh[kPcSB];
#pragma unroll
      for (int u = 0; u < kPcSB; ++u) {
        xr[u] = pc_ld16(rs_r, voff_t, cbase + (unsigned)(b0 + u * kPcBlock) * 16u);
        if (!first) xh[u] = pc_ld16(rs_h, voff_t, cbase + (unsigned)(b0 + u * kPcBlock) * 16u);
      }
      stage(xr, xh, b0, npair);
    }
    for (long i = 2L * npair + threadIdx.x; i < (long)cpad * r; i += kPcBlock) s_res[i] = 0.0;
    __syncthreads();
    const int nstep = cpad / 128;
    const int nu = (nstep + kPcNW - 1) / kPcNW;  // steps per wave (at most)
    int u0 = 0;
    if constexpr (FIRST) {
      rows(pre, 0, nstep);
      u0 = MB;
    }
#pragma unroll 1
    for (; u0 < nu; u0 += MB) {
      double2 mm[MB][NR];
#pragma unroll
      for (int u = 0; u < MB; ++u) {
#pragma unroll
        for (int q = 0; q < NR; ++q)
          mm[u][q] = pc_ld16(rs_m, voff_l,
                             (unsigned)(((size_t)(j0 + q) * ldm + c0 + (size_t)(wave_u + kPcNW * (u0 + u)) * 128) * 8));
      }
      rows(mm, u0, nstep);
    }
  };
  chunk(0, std::true_type{});
  if constexpr (MULTI) {
#pragma unroll 1
    for (int c0 = chk; c0 < k; c0 += chk) chunk(c0, std::false_type{});
  }
  // ---- sums over the workgroup: every product column and |r|^2.  Four DPP steps give each 16-lane row its sum
  //      (no lane reads, no LDS), one lane per row stores it, NR * RM + 1 threads add the 16 row sums in a fixed order
#pragma unroll
  for (int q = 0; q < NR; ++q)
#pragma unroll
    for (int t = 0; t < RM; ++t) {
      const double v = row16_sum_dpp(acc[q][t]);
      if ((lane & 15) == 0) s_P[q * RM + t][wave * 4 + (lane >> 4)] = v;
    }
  {
    const double v = row16_sum_dpp(nrm2);
    if ((lane & 15) == 0) s_P[NR * RM][wave * 4 + (lane >> 4)] = v;
  }
  __syncthreads();
  if ((int)threadIdx.x <= NR * RM) {
    double v = 0;
#pragma unroll
    for (int w = 0; w < 4 * kPcNW; ++w) v += s_P[threadIdx.x][w];
    s_Z[threadIdx.x] = v;
  }
  __syncthreads();
  // ---- residual stopping rule: every workgroup holds |r|^2 itself (same summation order everywhere) ----
  if (!first) {
    const double nr = sqrt(s_Z[NR * RM]);
    const double kappa = 0.1, tempnum = c_n0;  // theta = 1
    if (nr <= c_n0 * fmin(tempnum, kappa)) {
      if (blockIdx.x == 0 && threadIdx.x == 0) {
        ctl->tcg_status = (kappa < tempnum) ? 2 : 3;
        ctl->tcg_iters = iter + 1;
        ctl->inner_total += iter + 1;
        ctl->tcg_done_stamp = seq;
        f_host_store(&hf->tcg_done_seq, seq);
      }
      retract_tail();
      return;
    }
  }
  // ---- z = Proj_X(columns), partial <z, r>: the per-pose lanes all sit in wave 0 (PB <= 4 poses of 8 lanes) ----
  if (wave == 0) {
    Row<D> Zr, Rr;
#pragma unroll
    for (int a = 0; a < DH; ++a) {
      Zr.e[a] = pact ? s_Z[(g * DH + a) * RM + tt] : 0.0;
      Rr.e[a] = pact ? s_R[(g * DH + a) * RM + tt] : 0.0;
    }
    row_tangent<D>(Y, Zr);
    st_row<D>(z + o, r, tt, pact, Zr);
    double zacc = 0;
#pragma unroll
    for (int a = 0; a < DH; ++a) zacc += Zr.e[a] * Rr.e[a];
    const double tot = f_wave_sum(zacc);
    if (lane == 0) p3[blockIdx.x] = tot;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    if (!first && iter + 1 >= c_max_inner) {  // inner loop exhausted: status stays TR_MAXITER
      ctl->tcg_iters = iter + 1;
      ctl->inner_total += iter + 1;
      ctl->tcg_done_stamp = seq;
      f_host_store(&hf->tcg_done_seq, seq);
    }
    f_host_store(&hf->last_seq_done, seq);
  }
  if (!first && iter + 1 >= c_max_inner) retract_tail();
}

// ------------------------------------------------------------------------------------------------------
// ONE launch per tCG run (round 5): PC in "first" mode, then [A, PC] per iteration and the retraction of the step, all
// inside one kernel of n / 2 workgroups (one per CU at the headline size: 250), with two grid-wide steps per iteration
// instead of two kernel boundaries.  What it buys (tools/tcg_probe.hip, profiles/r05_persistent_tcg.txt): a grid step
// among 250 resident workgroups costs 1.9 us where a kernel boundary costs 2.9 us of launch floor + the prologue's
// round trips; the workgroup's 8 rows of the inverse stay in REGISTERS for the whole run (64 doubles per lane: 32 MB per
// launch were streamed per iteration); the residual image stays in LDS and only H delta is gathered per iteration.
// What it costs: everything the workgroups exchange inside the launch (z, delta, H delta, the partial sums) crosses the
// XCDs' private L2s through the coherent level -- write-through stores and loads with sc1 -- and the grid must be
// co-resident: the form is used only where n / 2 <= the CU count, never by concurrent solves (the coloured mode), and
// every spin is bounded: a workgroup that waits longer than 2 ms raises an abort word, all workgroups leave, the kernels
// queued behind the run become no-ops (outer_done_stamp) and the host repeats the RTR iteration on the launch form.
// The arithmetic is the launch form's, term for term and sum for sum (same lane layouts, same partial-sum trees: the
// <delta, H delta> partials are rebuilt from the workgroups' 16-lane row sums exactly as k_fused_hess's workgroups of
// PB_A poses add them), so a run is bitwise the run of the launches: tests/test_kernel_forms_gpu.py.
// ------------------------------------------------------------------------------------------------------
constexpr int kRunShards = 32, kRunCopies = 64, kRunStride = 32;  // sync words 128 B apart
constexpr int kRunSyncWords = (kRunShards + 1 + kRunCopies + 1) * kRunStride;
constexpr int kRunQCap = 1024;  // CSR entries of a workgroup's 2 (d+1) matrix rows staged in LDS once per run
struct TcgRunArgs {
  ManiDesc m;
  int ldm;
  const double *Minv;
  CsrDev Q;
  Buf2 grad, X, S;
  double *d0, *d1, *Hd, *eta, *Heta, *z, *p1r, *p3, *pC;
  unsigned *sync;  // zeroed by the single-block kernel in front of the run (k_rtr_init / k_rtr_decide)
  SolverCtl *ctl;
  HostFlags *hf;
  int seq, pbA;    // pbA: poses per workgroup of k_fused_hess at this r (the tree its <delta, H delta> partials follow)
  int fault;       // test hook (dcora_debug_tcg_run_fault): workgroup 0 leaves before the first grid step
#ifdef DCORA_RUN_STAMPS
  long long *stamps;  // profiling build only: wall_clock64 of workgroup 0 at the phase boundaries of a run
#endif
};
#ifdef DCORA_RUN_STAMPS
#define RUN_STAMP(i)                                                            \
  do {                                                                          \
    if (a.stamps && blockIdx.x == 0 && threadIdx.x == 0 && (i) < 64) a.stamps[(i)] = wall_clock64(); \
  } while (0)
#else
#define RUN_STAMP(i) \
  do {                \
  } while (0)
#endif
__device__ __forceinline__ double ld_coh(const double *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_coh(double *p, double v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double2 pc_ld16_coh(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) {
  const pc_v4u v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 16);  // aux bit 4: sc1 (agent scope)
  return __builtin_bit_cast(double2, v);
}
// grid-wide step `step` (0, 1, 2, ...) of this launch: sharded arrival counters, the last arrival replicates the done
// word, workgroup i polls copy i % 64.  false: this workgroup (or another one) gave up.
// Giving up is decided on the SAME word that counts the completed shards (its top bit): a workgroup whose wait ran out
// sets the bit by compare-and-swap only while the count is incomplete, and the arrival that completes the count
// publishes the done word only if its own increment found the bit clear.  So either the step completes for everybody
// or nobody passes it: a workgroup that was descheduled past its 2 ms while the others completed the step does not
// abort a run the others go on to finish (seen with four ranks sharing one GPU, where waits of milliseconds are routine).
constexpr unsigned kRunAbortBit = 0x80000000u;
__device__ __forceinline__ bool run_grid_step(unsigned *sync, unsigned step, int *s_ok) {
  __builtin_amdgcn_s_waitcnt(0);  // this wave's write-through stores are acknowledged
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned *shard = sync, *top = sync + kRunShards * kRunStride, *done = top + kRunStride,
             *abort_w = done + kRunCopies * kRunStride;
    const int i = blockIdx.x, G = gridDim.x, sh = i % kRunShards;
    const unsigned in_shard = (unsigned)((G - sh + kRunShards - 1) / kRunShards), want = step + 1;
    const unsigned shards_used = (unsigned)(G < kRunShards ? G : kRunShards);
    const unsigned a = __hip_atomic_fetch_add(shard + sh * kRunStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (a + 1 == want * in_shard) {
      const unsigned b = __hip_atomic_fetch_add(top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (!(b & kRunAbortBit) && b + 1 == want * shards_used)
        for (int c = 0; c < kRunCopies; ++c)
          __hip_atomic_store(done + c * kRunStride, want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    long long t0 = wall_clock64();
    const unsigned *p = done + (i % kRunCopies) * kRunStride;
    int ok = 1;
    unsigned spins = 0;
    while (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
      __builtin_amdgcn_s_sleep(1);
      if ((++spins & 63u) == 0) {
        if (__hip_atomic_load(abort_w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
          ok = 0;
          break;
        }
        if (wall_clock64() - t0 > 200000) {  // 2 ms at 100 MHz: the grid is not co-resident
          unsigned cur = __hip_atomic_load(top, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          bool gave_up = false;
          for (;;) {
            if (cur & kRunAbortBit) {  // somebody else gave up
              gave_up = true;
              break;
            }
            if (cur >= want * shards_used) break;  // everybody has arrived: the done word is on its way
            if (__hip_atomic_compare_exchange_strong(top, &cur, cur | kRunAbortBit, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                     __HIP_MEMORY_SCOPE_AGENT)) {
              gave_up = true;
              break;
            }
          }
          if (gave_up) {
            __hip_atomic_store(abort_w, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ok = 0;
            break;
          }
          t0 = wall_clock64();
        }
      }
    }
    *s_ok = ok;
  }
  __syncthreads();
  return *s_ok != 0;
}

template <int D, int R, int NS>
__global__ __launch_bounds__(kPcBlock) void k_tcg_run(TcgRunArgs a) {
  constexpr int DH = D + 1, PB = 2, NR = PB * DH, RM = R;
  extern __shared__ double s_res[];  // the residual image, column-major as in memory: cpad * R doubles, kept for the run
  __shared__ double s_P[NR * RM + 1][4 * kPcNW];
  __shared__ double s_Z[NR * RM + 1], s_R[NR * RM], s_W[NR * RM], s_D[NR * RM], s_H[NR * RM];
  __shared__ double s_red[16];
  __shared__ int s_ci[kRunQCap];
  __shared__ double s_v[kRunQCap];
  __shared__ int s_ok;
  SolverCtl *ctl = a.ctl;
  const int seq = a.seq;
  RUN_STAMP(0);
  const int st_o = ctl->outer_done_stamp, cur = ctl->cur & 1;
  if (seq > st_o) return;  // the RTR loop has ended: no-op (uniform over the grid)
  const double c_Delta = ctl->Delta, c_ngf = ctl->ngf;
  const int c_max_inner = ctl->max_inner;
  const ManiDesc m = a.m;
  constexpr int r = R;
  const int k = m.k, ldm = a.ldm;
  const int pose0 = blockIdx.x * PB;
  const int npose = min(PB, m.n - pose0);
  const int j0 = pose0 * DH, nrow = npose * DH;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int e = threadIdx.x;
  const bool own = e < nrow * r;
  const size_t oown = (size_t)j0 * r + e;
  const int lc = e / r, t = e - lc * r;
  const int g = threadIdx.x >> 3, tt = threadIdx.x & (GW - 1);
  const bool pact = (g < npose) && (tt < r);
  const size_t o = (size_t)(pose0 + min(g, npose - 1)) * DH * r;
  const double *__restrict__ grad = a.grad.p[cur];
  const double *__restrict__ X = a.X.p[cur];
  const unsigned vec_bytes = (unsigned)((size_t)r * k * sizeof(double));
  const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(grad), 0, vec_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_h = __builtin_amdgcn_make_buffer_rsrc(a.Hd, 0, vec_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_m = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<double *>(a.Minv), 0, (unsigned)((size_t)k * ldm * sizeof(double)), 0x00020000);
  const unsigned voff_t = threadIdx.x * 16u, voff_l = (unsigned)lane * 16u;
  // ---- once per run: the workgroup's rows of the inverse (registers), its matrix rows of Q (LDS), its poses ----
  double2 mreg[NS][NR];
#pragma unroll
  for (int u = 0; u < NS; ++u)
#pragma unroll
    for (int q = 0; q < NR; ++q)
      mreg[u][q] = pc_ld16(rs_m, voff_l, (unsigned)(((size_t)(j0 + q) * ldm + (size_t)(wave_u + kPcNW * u) * 128) * 8));
  double2 xg[kPcSB];
#pragma unroll
  for (int u = 0; u < kPcSB; ++u) xg[u] = pc_ld16(rs_g, voff_t, (unsigned)u * kPcBlock * 16u);
  const int pbeg = a.Q.rp[j0], pend = a.Q.rp[j0 + nrow];
  for (int i = threadIdx.x; i < pend - pbeg; i += kPcBlock) {
    s_ci[i] = a.Q.ci[pbeg + i];
    s_v[i] = a.Q.v[pbeg + i];
  }
  const int myb = own ? a.Q.rp[j0 + lc] - pbeg : 0, mye = own ? a.Q.rp[j0 + lc + 1] - pbeg : 0;
  Row<D> Y;
  ld_row<D>(X + o, r, tt, pact, Y);
  double S[D][D];
  {
    const double *__restrict__ Sblk = a.S.p[cur];
#pragma unroll
    for (int aa = 0; aa < D; ++aa)
#pragma unroll
      for (int b = 0; b < D; ++b) S[aa][b] = (g < npose) ? Sblk[(size_t)(pose0 + g) * D * D + aa + b * D] : 0.0;
  }
  double o_r = own ? grad[oown] : 0.0, o_eta = 0, o_Heta = 0, o_d = 0, o_h = 0;
  // image geometry (one chunk: the host admits the form only where the whole residual fits)
  const int cpad = ((k + 127) / 128) * 128;
  const long Nc = (long)k * r;
  const int npair = (int)((Nc + 1) >> 1);
  const int nstep = cpad / 128;
  // tCG scalars (the control block's, kept in registers by every thread: all of them see the same sums)
  double zr = 0, dPd = 0, ePe = 0, ePd = 0, alpha = 0, ePen = 0;
  const double n0 = c_ngf;
  int status = 4, iters_done = 0;
  unsigned gstep = 0;
  double acc[NR][RM];
  double nrm2 = 0;
  // product of the workgroup's rows (registers) with the image, sums over the workgroup: s_Z
  auto product = [&]() {
#pragma unroll
    for (int q = 0; q < NR; ++q)
#pragma unroll
      for (int tq = 0; tq < RM; ++tq) acc[q][tq] = 0;
#pragma unroll
    for (int u = 0; u < NS; ++u) {
      const int sidx = wave_u + kPcNW * u;
      if (sidx < nstep) {
        const double *__restrict__ xs = s_res + (size_t)(sidx * 128 + 2 * lane) * r;
        double x0[RM], x1[RM];
#pragma unroll
        for (int tq = 0; tq < RM; ++tq) {
          x0[tq] = xs[tq];
          x1[tq] = xs[r + tq];
        }
#pragma unroll
        for (int tq = 0; tq < RM; ++tq)
#pragma unroll
          for (int q = 0; q < NR; ++q) acc[q][tq] = fma(x0[tq], mreg[u][q].x, fma(x1[tq], mreg[u][q].y, acc[q][tq]));
      }
    }
#pragma unroll
    for (int q = 0; q < NR; ++q)
#pragma unroll
      for (int tq = 0; tq < RM; ++tq) {
        const double v = row16_sum_dpp(acc[q][tq]);
        if ((lane & 15) == 0) s_P[q * RM + tq][wave * 4 + (lane >> 4)] = v;
      }
    {
      const double v = row16_sum_dpp(nrm2);
      if ((lane & 15) == 0) s_P[NR * RM][wave * 4 + (lane >> 4)] = v;
    }
    __syncthreads();
    if ((int)threadIdx.x <= NR * RM) {
      double v = 0;
#pragma unroll
      for (int w = 0; w < 4 * kPcNW; ++w) v += s_P[threadIdx.x][w];
      s_Z[threadIdx.x] = v;
    }
    __syncthreads();
  };
  // z = Proj_X(columns), partial <z, r> (wave 0 holds the per-pose lanes), published for the other workgroups
  auto project_z = [&]() {
    if (wave == 0) {
      Row<D> Zr, Rr;
#pragma unroll
      for (int aa = 0; aa < DH; ++aa) {
        Zr.e[aa] = pact ? s_Z[(g * DH + aa) * RM + tt] : 0.0;
        Rr.e[aa] = pact ? s_R[(g * DH + aa) * RM + tt] : 0.0;
      }
      row_tangent<D>(Y, Zr);
      if (pact)
#pragma unroll
        for (int aa = 0; aa < DH; ++aa) st_coh(a.z + o + aa * r + tt, Zr.e[aa]);
      double zacc = 0;
#pragma unroll
      for (int aa = 0; aa < DH; ++aa) zacc += Zr.e[aa] * Rr.e[aa];
      const double tot = f_wave_sum(zacc);
      if (lane == 0) st_coh(a.p3 + blockIdx.x, tot);
    }
  };
  // the end of a run: control block, eta / H eta, the step itself (k_fused_pc's retract_tail)
  auto finish = [&]() {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      ctl->alpha = alpha;
      ctl->e_Pe_n = ePen;
      ctl->norm_r0 = n0;
      ctl->z_r[0] = ctl->z_r[1] = zr;
      ctl->d_Pd[0] = ctl->d_Pd[1] = dPd;
      ctl->e_Pe[0] = ctl->e_Pe[1] = ePe;
      ctl->e_Pd[0] = ctl->e_Pd[1] = ePd;
      ctl->tcg_status = status;
      ctl->tcg_iters = iters_done;
      ctl->inner_total += iters_done;
      ctl->tcg_done_stamp = seq;
      f_host_store(&a.hf->tcg_done_seq, seq);
      f_host_store(&a.hf->last_seq_done, seq);
    }
    __syncthreads();
    double a0 = 0, a1 = 0;
    if (own) {
      a.eta[oown] = o_eta;
      a.Heta[oown] = o_Heta;
      const double gr = grad[oown];
      a0 = o_eta * gr;
      a1 = o_eta * o_Heta;
      s_R[lc * RM + t] = o_eta;
    }
    __syncthreads();
    if (wave == 0) {
      Row<D> Yn = Y;
#pragma unroll
      for (int aa = 0; aa < DH; ++aa) Yn.e[aa] += 1.0 * (pact ? s_R[(g * DH + aa) * RM + tt] : 0.0);
      row_qf<D>(Yn);
      st_row<D>(a.X.p[cur ^ 1] + o, r, tt, pact, Yn);
    }
    const double t0 = f_block_sum(a0, s_red);
    const double t1 = f_block_sum(a1, s_red);
    if (threadIdx.x == 0) {
      a.pC[2 * blockIdx.x] = t0;
      a.pC[2 * blockIdx.x + 1] = t1;
    }
  };
  auto give_up = [&]() {  // the grid is not co-resident: later kernels of this solve become no-ops, the host repeats
    if (threadIdx.x == 0) {
      ctl->outer_done_stamp = seq - 1;
      f_host_store(&a.hf->tcg_abort_seq, seq);
    }
  };
  // ---- PC, first: z0 = Proj_X(grad Minv), res = grad, eta = H eta = 0 ----
#pragma unroll
  for (int u = 0; u < kPcSB; ++u) {
    const int i = u * kPcBlock + (int)threadIdx.x;
    if (i < npair) {
      const double2 x = xg[u];
      nrm2 = fma(x.x, x.x, nrm2);
      nrm2 = fma(x.y, x.y, nrm2);
      reinterpret_cast<double2 *>(s_res)[i] = x;
    }
  }
  for (long i = 2L * npair + threadIdx.x; i < (long)cpad * r; i += kPcBlock) s_res[i] = 0.0;
  if (own) s_R[lc * RM + t] = o_r;
  __syncthreads();
  RUN_STAMP(1);
  product();
  project_z();
  RUN_STAMP(2);
  if (c_max_inner <= 0) {  // (no inner iterations allowed: the launch form leaves eta = 0 behind as well)
    status = 4;
    finish();
    return;
  }
  if (a.fault && blockIdx.x == 0) return;  // (test hook: the others wait in vain, give up after 2 ms and say so)
  if (!run_grid_step(a.sync, gstep++, &s_ok)) return give_up();
  RUN_STAMP(3);
  // ---- the iterations ----
  for (int iter = 0;; ++iter) {
    const int par = iter & 1;
    double *__restrict__ d_new = par ? a.d1 : a.d0;
    const double *__restrict__ d_old = par ? a.d0 : a.d1;
    // ======== A: delta = beta delta - z in the gather, H delta = Proj_X(delta Q - delta S), <delta, H delta> ========
    {
      // the gather's loads first (their addresses do not depend on beta): one round trip through the coherent level for
      // up to kRunGB entries of a matrix row, beside the partials' -- in chunks of 8 behind the partials' sum an
      // iteration paid three to four dependent round trips here
      constexpr int kRunGB = 24;
      double ga[kRunGB], gz[kRunGB];
      double z_own = 0;
      if (own) {
        z_own = ld_coh(a.z + oown);
#pragma unroll
        for (int q = 0; q < kRunGB; ++q) {
          const bool ok = myb + q < mye;
          const size_t oo = ok ? (size_t)s_ci[myb + q] * r + t : 0;
          gz[q] = ld_coh(a.z + oo);
          ga[q] = (iter > 0) ? ld_coh(d_old + oo) : 0.0;
        }
      }
      const int np3 = gridDim.x;
      const int l = lane;
      const double pa = (l < np3) ? ld_coh(a.p3 + l) : 0.0, pb = (l + 64 < np3) ? ld_coh(a.p3 + l + 64) : 0.0;
      const double pcc = (l + 128 < np3) ? ld_coh(a.p3 + l + 128) : 0.0, pd = (l + 192 < np3) ? ld_coh(a.p3 + l + 192) : 0.0;
      const double z_r_new = f_wave_sum((pa + pb) + (pcc + pd));
      double beta = 0;
      if (iter > 0) beta = z_r_new / zr;
      if (iter == 0) {
        zr = z_r_new;
        dPd = z_r_new;
        ePe = 0;
        ePd = 0;
      } else {
        const double c_ePd = ePd, c_alpha = alpha, c_dPd = dPd;
        zr = z_r_new;
        ePd = beta * (c_ePd + c_alpha * c_dPd);
        dPd = z_r_new + beta * beta * c_dPd;
        ePe = ePen;
      }
      double accw = 0, dn = 0;
      if (own) {
#pragma unroll
        for (int q = 0; q < kRunGB; ++q) {
          const double w = (myb + q < mye) ? s_v[myb + q] : 0.0;  // (the weights come from LDS when they are used)
          accw += w * (beta * ga[q] - gz[q]);
        }
        for (int p = myb + kRunGB; p < mye; p += 8) {
          double a8[8], b8[8], w8[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const bool ok = p + q < mye;
            const size_t oo = ok ? (size_t)s_ci[p + q] * r + t : 0;
            w8[q] = ok ? s_v[p + q] : 0.0;
            b8[q] = ld_coh(a.z + oo);
            a8[q] = (iter > 0) ? ld_coh(d_old + oo) : 0.0;
          }
#pragma unroll
          for (int q = 0; q < 8; ++q) accw += w8[q] * (beta * a8[q] - b8[q]);
        }
        dn = (iter > 0) ? beta * o_d - z_own : -z_own;
        st_coh(d_new + oown, dn);
        o_d = dn;
        s_W[e] = accw;
        s_D[e] = dn;
      }
      __syncthreads();
      if (wave == 0) {
        Row<D> V, W;
#pragma unroll
        for (int aa = 0; aa < DH; ++aa) {
          W.e[aa] = pact ? s_W[(g * DH + aa) * r + tt] : 0.0;
          V.e[aa] = pact ? s_D[(g * DH + aa) * r + tt] : 0.0;
        }
        row_sub_AS<D>(W, V, S);
        row_tangent<D>(Y, W);
        if (pact)
#pragma unroll
          for (int aa = 0; aa < DH; ++aa) {
            st_coh(a.Hd + o + aa * r + tt, W.e[aa]);
            s_H[(g * DH + aa) * r + tt] = W.e[aa];
          }
        double dacc = 0;
#pragma unroll
        for (int aa = 0; aa < DH; ++aa) dacc += V.e[aa] * W.e[aa];
        if (!pact) dacc = 0;
        const double rs = row16_sum_dpp(dacc);  // the 16-lane row sum k_fused_hess's block sum starts from
        if (lane == 0) st_coh(a.p1r + blockIdx.x, rs);
      }
      __syncthreads();
      if (own) o_h = s_H[e];
    }
    RUN_STAMP(4 + 7 * iter);
    if (!run_grid_step(a.sync, gstep++, &s_ok)) return give_up();
    RUN_STAMP(5 + 7 * iter);
    // ======== PC: step length, updates, z = Proj_X(res Minv), stopping rules ========
    double2 xh[kPcSB];
#pragma unroll
    for (int u = 0; u < kPcSB; ++u) xh[u] = pc_ld16_coh(rs_h, voff_t, (unsigned)u * kPcBlock * 16u);
    double d_Hd;
    {
      // k_fused_hess's partial of workgroup w: rows 2 w' .. of its pbA poses, wave by wave; then the wave sum over them
      const int rows_per = a.pbA / 2, nrows = gridDim.x, nblk = (nrows + rows_per - 1) / rows_per;
      double part = 0;
      if (lane < nblk) {
        const int b0 = lane * rows_per;
        double wsum[4] = {0, 0, 0, 0};
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          double rs4[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int ri = w * 4 + q;
            rs4[q] = (ri < rows_per && b0 + ri < nrows) ? ld_coh(a.p1r + b0 + ri) : 0.0;
          }
          wsum[w] = (rs4[0] + rs4[1]) + (rs4[2] + rs4[3]);
        }
        double tsum = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) tsum += wsum[w];
        part = tsum;
      }
      d_Hd = f_wave_sum(part);
    }
    const double c_zr = zr, c_dPd = dPd, c_ePe = ePe, c_ePd = ePd;
    alpha = c_zr / d_Hd;
    const double e_Pe_new = c_ePe + 2.0 * alpha * c_ePd + alpha * alpha * c_dPd;
    const bool boundary = (d_Hd <= 0) || (e_Pe_new >= c_Delta * c_Delta);
    const double step =
        boundary ? (-c_ePd + sqrt(c_ePd * c_ePd + c_dPd * (c_Delta * c_Delta - c_ePe))) / c_dPd : alpha;
    ePen = e_Pe_new;
    if (own) {
      o_eta = o_eta + step * o_d;
      o_Heta = o_Heta + step * o_h;
    }
    if (boundary) {
      status = (d_Hd <= 0) ? 0 : 1;
      iters_done = iter + 1;
      finish();
      return;
    }
    if (own) {
      const double rr = fma(alpha, o_h, o_r);
      o_r = rr;
      s_R[lc * RM + t] = rr;
    }
    RUN_STAMP(6 + 7 * iter);
    nrm2 = 0;
#pragma unroll
    for (int u = 0; u < kPcSB; ++u) {
      const int i = u * kPcBlock + (int)threadIdx.x;
      if (i < npair) {
        double2 x = reinterpret_cast<double2 *>(s_res)[i];
        x.x = fma(alpha, xh[u].x, x.x);
        x.y = fma(alpha, xh[u].y, x.y);
        nrm2 = fma(x.x, x.x, nrm2);
        nrm2 = fma(x.y, x.y, nrm2);
        reinterpret_cast<double2 *>(s_res)[i] = x;
      }
    }
    __syncthreads();
    RUN_STAMP(7 + 7 * iter);
    product();
    RUN_STAMP(8 + 7 * iter);
    {
      const double nr = sqrt(s_Z[NR * RM]);
      const double kappa = 0.1, tempnum = n0;  // theta = 1
      if (nr <= n0 * fmin(tempnum, kappa)) {
        status = (kappa < tempnum) ? 2 : 3;
        iters_done = iter + 1;
        finish();
        return;
      }
    }
    project_z();
    if (iter + 1 >= c_max_inner) {  // inner loop exhausted: status stays TR_MAXITER
      iters_done = iter + 1;
      finish();
      return;
    }
    RUN_STAMP(9 + 7 * iter);
    if (!run_grid_step(a.sync, gstep++, &s_ok)) return give_up();
    RUN_STAMP(10 + 7 * iter);
  }
}


// ------------------------------------------------------------------------------------------------------
// RG = Proj_X(EG), S_i = sym(Y_i^T EG_i), partial |RG|^2
template <int D>
__global__ __launch_bounds__(kBlock) void k_g_rgrad(ManiDesc m, Buf2 Xb, Buf2 EGb, Buf2 RGb, Buf2 Sb, int sel,
                                                    double *__restrict__ partials, double *__restrict__ posenorm,
                                                    Gate g) {
  if (g.ctl && g.gate && f_gated(g.ctl, g.seq, g.gate)) return;
  __shared__ double s_red[16];
  constexpr int DH = D + 1;
  const int idx = g.ctl ? ((g.ctl->cur ^ sel) & 1) : 0;
  const double *__restrict__ X = Xb.p[idx];
  const double *__restrict__ EG = EGb.p[idx];
  double *__restrict__ RG = RGb.p[idx];
  double *__restrict__ Sblk = Sb.p[idx];
  const int r = m.r;
  const int t = threadIdx.x & (GW - 1);
  double acc = 0;
  for (int pose0 = blockIdx.x * kPosesPerBlock; pose0 < m.n; pose0 += gridDim.x * kPosesPerBlock) {
    const int pose = pose0 + (threadIdx.x >> 3);
    const bool active = (pose < m.n) && (t < r);
    const size_t o = (size_t)pose * DH * r;
    Row<D> Y, E;
    ld_row<D>(X + o, r, t, active, Y);
    ld_row<D>(EG + o, r, t, active, E);
    double S[D][D];
    grp_sym_gram<D>(Y, E, S);
    if (Sblk && pose < m.n && t == 0)
#pragma unroll
      for (int a = 0; a < D; ++a)
#pragma unroll
        for (int b = 0; b < D; ++b) Sblk[(size_t)pose * D * D + a + b * D] = S[a][b];
    row_sub_AS<D>(E, Y, S);
    double pa = 0;
#pragma unroll
    for (int a = 0; a < DH; ++a) pa += E.e[a] * E.e[a];
    acc += pa;
    if (posenorm) {  // |RG_i|^2 per pose, for the per-agent block norms of the evaluation
      const double ps = grp_sum(pa);
      if (pose < m.n && t == 0) posenorm[pose] = ps;
    }
    if (RG) st_row<D>(RG + o, r, t, active, E);
  }
  const double tot = f_block_sum(acc, s_red);
  if (threadIdx.x == 0 && partials) partials[blockIdx.x] = tot;
}

// out = Retr_X(alpha V), partial {<V, grad>, <V, HV>}
template <int D>
__global__ __launch_bounds__(kBlock) void k_g_retract(ManiDesc m, Buf2 Xb, const double *__restrict__ V,
                                                      double alpha, Buf2 Ob, int selOut, Buf2 gradb,
                                                      const double *__restrict__ HV, double *__restrict__ partials,
                                                      Gate g) {
  if (g.ctl && g.gate && f_gated(g.ctl, g.seq, g.gate)) return;
  __shared__ double s_red[16];
  constexpr int DH = D + 1;
  const int cur = g.ctl ? (g.ctl->cur & 1) : 0;
  const double *__restrict__ X = Xb.p[cur];
  double *__restrict__ out = Ob.p[g.ctl ? ((cur ^ selOut) & 1) : 0];
  const double *__restrict__ grad = partials ? gradb.p[cur] : nullptr;
  const int r = m.r;
  const int t = threadIdx.x & (GW - 1);
  double a0 = 0, a1 = 0;
  for (int pose0 = blockIdx.x * kPosesPerBlock; pose0 < m.n; pose0 += gridDim.x * kPosesPerBlock) {
    const int pose = pose0 + (threadIdx.x >> 3);
    const bool active = (pose < m.n) && (t < r);
    const size_t o = (size_t)pose * DH * r;
    Row<D> Y, Vr;
    ld_row<D>(X + o, r, t, active, Y);
    ld_row<D>(V + o, r, t, active, Vr);
    if (partials) {
      Row<D> Gr, Hr;
      ld_row<D>(grad + o, r, t, active, Gr);
      ld_row<D>(HV + o, r, t, active, Hr);
#pragma unroll
      for (int a = 0; a < DH; ++a) {
        a0 += Vr.e[a] * Gr.e[a];
        a1 += Vr.e[a] * Hr.e[a];
      }
    }
#pragma unroll
    for (int a = 0; a < DH; ++a) Y.e[a] += alpha * Vr.e[a];
    row_qf<D>(Y);
    st_row<D>(out + o, r, t, active, Y);
  }
  if (partials) {
    const double t0 = f_block_sum(a0, s_red);
    const double t1 = f_block_sum(a1, s_red);
    if (threadIdx.x == 0) {
      partials[2 * blockIdx.x] = t0;
      partials[2 * blockIdx.x + 1] = t1;
    }
  }
}

// RBCD++ Nesterov bookkeeping (modes as k_nesterov in kernels.hip)
struct GNesterovArgs {
  int mode, restart, skip_lo, skip_hi;
  double alpha, gamma;
  double *X, *V, *Y, *XPrev, *Yloc;
  double *inner_Yloc;     // mode 0 only: when set, the skipped poses run mode 1 instead (Y and this buffer <- their y)
  Buf2 Xloc;              // result buffers of the local solve
  const SolverCtl *ctl;   // picks Xloc.p[ctl->cur] when non-null
};
template <int D>
__global__ __launch_bounds__(kBlock) void k_g_nesterov(ManiDesc m, GNesterovArgs a) {
  const double *__restrict__ Xloc = a.Xloc.p[a.ctl ? (a.ctl->cur & 1) : 0];
  constexpr int DH = D + 1;
  const int r = m.r;
  const int t = threadIdx.x & (GW - 1);
  for (int pose0 = blockIdx.x * kPosesPerBlock; pose0 < m.n; pose0 += gridDim.x * kPosesPerBlock) {
    const int pose = pose0 + (threadIdx.x >> 3);
    const bool skipped = pose >= a.skip_lo && pose < a.skip_hi;
    const bool inner = skipped && a.inner_Yloc != nullptr && pose < m.n;  // the selected agent's poses, staged here
    const bool inrange = (pose < m.n) && (!skipped || inner);
    const bool active = inrange && (t < r);
    const size_t o = (size_t)pose * DH * r;
    Row<D> x, v, y;
    if (a.mode <= 1) {
      ld_row<D>(a.X + o, r, t, active, x);
      ld_row<D>(a.V + o, r, t, active, v);
#pragma unroll
      for (int c = 0; c < DH; ++c) y.e[c] = (1.0 - a.alpha) * x.e[c] + a.alpha * v.e[c];
      const double yt = y.e[D];
      row_polar<D>(y, inrange);
      y.e[D] = yt;
      st_row<D>(a.XPrev + o, r, t, active, x);
      if (a.mode == 1 || inner) {
        st_row<D>(a.Y + o, r, t, active, y);
        double *yl = inner ? a.inner_Yloc + (size_t)(pose - a.skip_lo) * DH * r : a.Yloc + o;
        st_row<D>(yl, r, t, active, y);
      } else if (a.restart & 1) {
        st_row<D>(a.V + o, r, t, active, x);
        st_row<D>(a.Y + o, r, t, active, x);
        // uniform control flow for the second polar below is not needed: restart is a kernel argument
      } else {
        st_row<D>(a.Y + o, r, t, active, y);
        st_row<D>(a.X + o, r, t, active, y);
        // V <- proj(V + gamma (X - Y)) with X == Y.  Bit 1 of `restart`: V is known to be feasible (it is the output
        // of a projection or a copy of a feasible X since the last set_X), its re-projection is the identity.
        if (!(a.restart & 2)) {
          const double vt = v.e[D];
          row_polar<D>(v, inrange);
          v.e[D] = vt;
          st_row<D>(a.V + o, r, t, active, v);
        }
      }
    } else {
      ld_row<D>(Xloc + o, r, t, active, x);
      st_row<D>(a.X + o, r, t, active, x);
      if (a.mode == 2) {
        ld_row<D>(a.V + o, r, t, active, v);
        ld_row<D>(a.Y + o, r, t, active, y);
#pragma unroll
        for (int c = 0; c < DH; ++c) v.e[c] += a.gamma * (x.e[c] - y.e[c]);
        const double vt = v.e[D];
        row_polar<D>(v, inrange);
        v.e[D] = vt;
        st_row<D>(a.V + o, r, t, active, v);
      } else {
        st_row<D>(a.V + o, r, t, active, x);
        st_row<D>(a.Y + o, r, t, active, x);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// Q-apply on the block structure of the connection Laplacian: Y = X Q (+ G), Q in BSR with (d+1)^2 blocks stored
// column-major.  History of the forms (all measured on the 100k lattice, r = 5, warm / cold us): LDS-staged blocks with 8
// lanes per pose 27.4 / 33.9; the LDS-free 8-lanes-per-pose form (lane t = row t of the pose's block, four 8-byte gathers
// and two 16-byte block loads per lane and block, the block's other rows by DPP quad broadcasts) 24.6 / 33.2 -- rounds
// 2-4, unmoved by gather depth, software pipelining, non-temporal accesses, a symmetric store of the blocks, r lanes
// per pose, 16-byte gathers of row pairs, a per-pose header, locality orderings and XCD-aware grids: it was bound by
// the NUMBER of gather instructions (each serves 8 poses).  Round 5: the quad-per-pose form below, 22.6 / 27.5.
// ------------------------------------------------------------------------------------------------------
template <int A_>
__device__ __forceinline__ double quad_bcast(double v) {
  return dpp_move<A_ * 0x55>(v);  // quad_perm [A, A, A, A]
}
// GRAD: the whole evaluation of an RTR iteration in this launch -- a lane group already holds EG_i = (X Q + G)_i in the
// layout k_g_rgrad works in, so RG_i = Proj_X(EG_i), S_i = sym(Y_i^T EG_i), the partial |RG|^2 and the per-pose norms
// follow as an epilogue (the same operations in the same order as k_g_rgrad) instead of a launch of their own that
// reads EG and X back: one dependent launch less per evaluation, four per local solve of a large block.
struct BsrGradOut {
  Buf2 RG, S;
  double *pB = nullptr;        // partial |RG|^2, one per workgroup
  double *posenorm = nullptr;  // |RG_i|^2 per pose, or null
  // central evaluation of an RBCD pass: the workgroups are dealt to the agents (wg_per_agent each, a slice of the
  // agent's poses per workgroup), so pB holds every agent's |rgrad_b|^2 in wg_per_agent consecutive slots and the
  // epilogue kernel adds those -- no per-pose norms, no launch that sums them per agent
  const int *agent_start = nullptr;
  int wg_per_agent = 0;
};
// ------------------------------------------------------------------------------------------------------
// The block Q-apply (round 5): FOUR lanes per pose, lane c owns COLUMN c of the pose's r x (d+1) block.
// A column is r contiguous doubles, so the neighbour's block arrives by ceil(r / 2) 16-byte loads per lane (8-byte
// aligned; r = 5: 16 + 16 + 8 bytes) in instructions that serve 16 poses each, and the d + 1 weights lane c needs --
// column c of the (d+1)^2 block, stored column-major -- are one contiguous run (two 16-byte loads, no duplicate loads by
// a second quad, no DPP broadcast in the inner loop): 5 load instructions per 16 (pose, block) pairs where the
// 8-lanes-per-pose form issues 6 per 8, and no idle lanes at r = 5.  Lane c accumulates ITS column's contribution to
// all d + 1 output columns (r (d+1) sums in registers); the four partials of a pose meet once per pose in a quad
// reduce-scatter (lane a ends with output column a), then G, the dots and the store run on contiguous columns again.
// GRAD: the whole evaluation of an RTR iteration as the epilogue (E and Y columns handed round the quad by DPP).
// ------------------------------------------------------------------------------------------------------
typedef double q_v2f64u __attribute__((ext_vector_type(2), aligned(8)));
template <int N>
__device__ __forceinline__ void ld_run(const double *__restrict__ p, bool ok, double (&x)[N]) {
#pragma unroll
  for (int i = 0; i + 1 < N; i += 2) {
    q_v2f64u v = {0.0, 0.0};
    if (ok) v = *reinterpret_cast<const q_v2f64u *>(p + i);
    x[i] = v.x;
    x[i + 1] = v.y;
  }
  if (N & 1) x[N - 1] = ok ? p[N - 1] : 0.0;
}
template <int N>
__device__ __forceinline__ void st_run(double *__restrict__ p, bool ok, const double (&x)[N]) {
  if (!ok) return;
#pragma unroll
  for (int i = 0; i + 1 < N; i += 2) {
    q_v2f64u v = {x[i], x[i + 1]};
    *reinterpret_cast<q_v2f64u *>(p + i) = v;
  }
  if (N & 1) p[N - 1] = x[N - 1];
}
template <int A_>
__device__ __forceinline__ int quad_bcast_i(int v) {
  return __builtin_amdgcn_update_dpp(0, v, A_ * 0x55, 0xF, 0xF, true);
}
__device__ __forceinline__ double quad_sum(double v) {
  asm volatile("" : "+v"(v));
  v += f_dpp<0xB1>(v);  // quad_perm [1,0,3,2]
  asm volatile("" : "+v"(v));
  v += f_dpp<0x4E>(v);  // quad_perm [2,3,0,1]
  return v;
}
// Measured on the 100k lattice at r = 5, warm / cold us (gfx950, round 5): this form 22.6 / 27.5; its loads requested two
// or one (pose, block) pair at a time instead of four 22.6-22.8 / 27.6; 256-thread workgroups 22.7 / 28.3; the pose's
// own column of X and of G requested before the block loop (136 registers, 3 waves per SIMD) 23.6 / 27.6; half as many
// workgroups of two passes each 25.1 / 30.1; forced to 5 or 6 waves per SIMD (176 / 320 bytes of scratch per lane)
// 72 / 128 us.  The 8-lanes-per-pose form it replaces: 24.6 / 33.2.
constexpr int kQBlock = 128;  // threads per workgroup: 32 poses, as the 8-lanes-per-pose kernels' workgroups hold
template <int D, int R, bool DOTS, bool GRAD>
__global__ __launch_bounds__(kQBlock) void k_spmm_bsrq(BsrDev A, Buf2 Xb, int selX, const double *__restrict__ G, Buf2 Yb,
                                                       int selY, double *__restrict__ partials, Gate g, BsrGradOut go) {
  if (g.ctl && g.gate && f_gated(g.ctl, g.seq, g.gate)) return;
  constexpr int DH = D + 1, BS = DH * DH, PW = kQBlock / 4;
  // (pose, block) pairs whose loads a lane requests together: the four column indices of one index load, two at a time
  // at r >= 7 where four columns of the neighbours alone are 56-64 registers
  constexpr int kQGather = R >= 7 ? 2 : 4;
  __shared__ double s_red[16];
  const int cur = g.ctl ? (g.ctl->cur & 1) : 0;
  const double *__restrict__ X = Xb.p[g.ctl ? ((cur ^ selX) & 1) : 0];
  double *__restrict__ Y = Yb.p[g.ctl ? ((cur ^ selY) & 1) : 0];
  double *__restrict__ RG = GRAD ? go.RG.p[g.ctl ? ((cur ^ selY) & 1) : 0] : nullptr;
  double *__restrict__ Sblk = GRAD ? go.S.p[g.ctl ? ((cur ^ selY) & 1) : 0] : nullptr;
  const int c = threadIdx.x & 3;
  const bool lane_on = c < DH;  // d = 2: the fourth lane of a quad carries zeros
  const int cc = lane_on ? c : 0;
  const bool odd = (c & 1) != 0, upper = (c & 2) != 0;
  double d0 = 0, d1 = 0, dg = 0;
  int range_lo = (int)((long)A.nbrows * blockIdx.x / gridDim.x);
  int range_hi = (int)((long)A.nbrows * (blockIdx.x + 1) / gridDim.x);
  if (GRAD && go.agent_start) {
    const int a = blockIdx.x / go.wg_per_agent, sl = blockIdx.x - a * go.wg_per_agent;
    const int lo = go.agent_start[a], hi = go.agent_start[a + 1];
    range_lo = lo + (int)((long)(hi - lo) * sl / go.wg_per_agent);
    range_hi = lo + (int)((long)(hi - lo) * (sl + 1) / go.wg_per_agent);
  }
  const int npass = max(1, (range_hi - range_lo + PW - 1) / PW);
  const int per_pass = (range_hi - range_lo + npass - 1) / npass;
  for (int pose0 = range_lo; pose0 < range_hi; pose0 += per_pass) {
    const int pend_pose = min(range_hi, pose0 + per_pass);
    const int pose = pose0 + (threadIdx.x >> 2);
    const bool inr = pose < pend_pose;
    const bool active = inr && lane_on;
    const int myb = inr ? A.bp[pose] : 0, mye = inr ? A.bp[pose + 1] : 0;
    double acc[4][R];  // acc[a][.]: this lane's (column c's) part of output column a
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int i = 0; i < R; ++i) acc[a][i] = 0.0;
    for (int b0 = myb; b0 < mye; b0 += 4) {
      const int nb = min(4, mye - b0);
      const int mybc = (c < nb) ? A.bc[b0 + c] : 0;
#pragma unroll
      for (int h = 0; h < 4; h += kQGather) {
        double x[kQGather][R], w[kQGather][DH];
#pragma unroll
        for (int q = 0; q < kQGather; ++q) {
          const bool ok = active && (h + q < nb);
          const int col = (h + q == 0) ? quad_bcast_i<0>(mybc) : (h + q == 1) ? quad_bcast_i<1>(mybc)
                        : (h + q == 2) ? quad_bcast_i<2>(mybc) : quad_bcast_i<3>(mybc);
          ld_run<R>(X + ((size_t)col * DH + cc) * R, ok, x[q]);
          ld_run<DH>(A.bv + (size_t)(b0 + h + q) * BS + cc * DH, ok, w[q]);
        }
#pragma unroll
        for (int q = 0; q < kQGather; ++q)
#pragma unroll
          for (int a = 0; a < DH; ++a)
#pragma unroll
            for (int i = 0; i < R; ++i) acc[a][i] += w[q][a] * x[q][i];
        if (kQGather < 4) __builtin_amdgcn_sched_barrier(0);  // the next pair's loads stay behind this pair's sums
      }
    }
    // quad reduce-scatter: lane a ends with output column a = the sum of the four lanes' acc[a][.]
    double e[R];
#pragma unroll
    for (int i = 0; i < R; ++i) {
      // pairs {0,1}, {2,3}: a lane keeps the column of its own parity and sends the other to its partner
      const double keep0 = odd ? acc[1][i] : acc[0][i], send0 = odd ? acc[0][i] : acc[1][i];
      const double keep1 = odd ? acc[3][i] : acc[2][i], send1 = odd ? acc[2][i] : acc[3][i];
      const double t0 = keep0 + f_dpp<0xB1>(send0);  // columns (c & 1) over lanes c, c ^ 1
      const double t1 = keep1 + f_dpp<0xB1>(send1);  // columns 2 + (c & 1)
      const double keep = upper ? t1 : t0, send = upper ? t0 : t1;
      e[i] = keep + f_dpp<0x4E>(send);
    }
    const size_t oc = ((size_t)(inr ? pose : 0) * DH + cc) * R;
    double xo[R], gg[R];
    if (DOTS || GRAD) ld_run<R>(X + oc, active, xo);
    ld_run<R>(G + oc, active && G != nullptr, gg);
    if (DOTS || GRAD) {
#pragma unroll
      for (int i = 0; i < R; ++i) {
        d0 += e[i] * xo[i];
        d1 += xo[i] * gg[i];
      }
    }
#pragma unroll
    for (int i = 0; i < R; ++i) e[i] = active ? e[i] + gg[i] : 0.0;
    if (Y) st_run<R>(Y + oc, active, e);
    if (GRAD) {
      // columns of Y (= X_i) and E handed round the quad; every lane forms S = sym(Y^T E) over the rotation columns
      double Yc[D][R], Ec[D][R];
#pragma unroll
      for (int i = 0; i < R; ++i) {
        Yc[0][i] = quad_bcast<0>(xo[i]);
        Ec[0][i] = quad_bcast<0>(e[i]);
        Yc[1][i] = quad_bcast<1>(xo[i]);
        Ec[1][i] = quad_bcast<1>(e[i]);
        if (D == 3) {
          Yc[D - 1][i] = quad_bcast<2>(xo[i]);
          Ec[D - 1][i] = quad_bcast<2>(e[i]);
        }
      }
      double S[D][D];
#pragma unroll
      for (int a = 0; a < D; ++a)
#pragma unroll
        for (int b = a; b < D; ++b) {
          double s = 0;
#pragma unroll
          for (int i = 0; i < R; ++i) s += 0.5 * (Yc[a][i] * Ec[b][i] + Yc[b][i] * Ec[a][i]);
          S[a][b] = s;
          S[b][a] = s;
        }
      if (Sblk && inr && c < D) {  // lane b stores column b of the D x D block
#pragma unroll
        for (int a = 0; a < D; ++a) {
          const double s = (c == 0) ? S[a][0] : (c == 1) ? S[a][1] : S[a][D - 1];
          Sblk[(size_t)pose * D * D + a + c * D] = s;
        }
      }
      // RG column b = E column b - sum_a Y column a S[a][b] (rotation columns; the translation column stays)
      if (c < D) {
#pragma unroll
        for (int i = 0; i < R; ++i) {
          double s = 0;
#pragma unroll
          for (int a = 0; a < D; ++a) s += Yc[a][i] * ((c == 0) ? S[a][0] : (c == 1) ? S[a][1] : S[a][D - 1]);
          e[i] -= s;
        }
      }
      double pa = 0;
#pragma unroll
      for (int i = 0; i < R; ++i) pa += e[i] * e[i];
      if (!active) pa = 0;
      dg += pa;
      if (go.posenorm) {
        const double ps = quad_sum(pa);
        if (inr && c == 0) go.posenorm[pose] = ps;
      }
      if (RG) st_run<R>(RG + oc, active, e);
    }
  }
  if (DOTS) {
    const double a = f_block_sum(d0, s_red);
    const double b = f_block_sum(d1, s_red);
    if (threadIdx.x == 0) {
      partials[2 * blockIdx.x] = a;
      partials[2 * blockIdx.x + 1] = b;
    }
  }
  if (GRAD) {
    const double cs = f_block_sum(dg, s_red);
    if (threadIdx.x == 0) go.pB[blockIdx.x] = cs;
  }
}

// (Measured and dropped, round 3: a third form with HALF the load instructions -- lanes as (column pair, row pair) of a
// pose, two 16-byte gathers and one 16-byte block load per lane and block instead of six loads -- was slower everywhere,
// also at even r where every gather is 16-byte aligned: r = 5 28.6 / 36.7 us warm / cold against 24.6 / 33.1.  Round 4:
// a locality ordering of the poses (sub-cubes of the lattice, 4x4x2 .. 2x2x8, instead of the trajectory order) changed
// nothing: 24.0-24.3 / 32.8-33.3 against 24.0 / 32.6, tools/qapply_order.py.  The Q-apply is bound by the dependent
// chain (row pointer -> column indices -> gather) at the head of its ~3000 short-lived workgroups.)

int group_grid(int n) {
  long g = ((long)n + kPosesPerBlock - 1) / kPosesPerBlock;
  if (g < 1) g = 1;
  if (g > kMaxPartials) g = kMaxPartials;
  return (int)g;
}

}  // namespace

// start-of-solve control block, written on the device so that a solve needs no host-to-device copy
__global__ void k_ctl_init(SolverCtl *c, double tol, double Delta, double maxDelta, int max_outer, int stop_on_accept,
                           int max_inner) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  c->f1 = c->ngf = c->f2 = c->rho = c->fInit = c->gradNormInit = 0;
  c->Delta = Delta;
  c->maxDelta = maxDelta;
  c->tol = tol;
  c->cur = 0;
  c->outer_it = 0;
  c->max_outer = max_outer;
  c->accepted = 0;
  c->last_accepted = 0;
  c->stop_on_accept = stop_on_accept;
  c->outer_done_stamp = INT_MAX;
  c->alpha = c->e_Pe_n = c->norm_r0 = 0;
  c->tcg_done_stamp = INT_MAX;
  c->tcg_status = 4;
  c->tcg_iters = 0;
  c->inner_total = 0;
  c->max_inner = max_inner;
}
void launch_ctl_init(hipStream_t st, SolverCtl *c, double tol, double Delta, double maxDelta, int max_outer,
                     int stop_on_accept, int max_inner) {
  hipLaunchKernelGGL(k_ctl_init, dim3(1), dim3(64), 0, st, c, tol, Delta, maxDelta, max_outer, stop_on_accept,
                     max_inner);
}

// Evaluation epilogue of one RBCD pass (ref examples/MultiRobotExample.cpp:264-305): per-agent |rgrad_b| from the
// per-pose squared norms, 2 f from the Q-apply partials, greedy argmax; results go to host-mapped memory and are
// published by a sequence word, so the host never calls into the runtime to read them.
// Large graphs (100k poses: one workgroup walked 12 500 norms per agent in 24 dependent steps, 27 us per RBCD
// iteration): the per-agent sums are split over kEvalSplit workgroups per agent first, fixed slices, fixed order.
constexpr int kEvalSplit = 32;
__global__ __launch_bounds__(kBlock) void k_eval_partial(const int *__restrict__ pose_start,
                                                         const double *__restrict__ posenorm,
                                                         double *__restrict__ part) {
  __shared__ double s_red[16];
  const int b = blockIdx.x / kEvalSplit, sl = blockIdx.x - b * kEvalSplit;
  const int lo = pose_start[b], hi = pose_start[b + 1];
  const int per = (hi - lo + kEvalSplit - 1) / kEvalSplit;
  const int i0 = lo + sl * per, i1 = min(hi, i0 + per);
  double v = 0;
  for (int i = i0 + (int)threadIdx.x; i < i1; i += kBlock) v += posenorm[i];
  v = f_block_sum(v, s_red);
  if (threadIdx.x == 0) part[blockIdx.x] = v;
}
__global__ __launch_bounds__(kBlock) void k_eval_finish(int R, const int *__restrict__ pose_start,
                                                        const double *__restrict__ posenorm,
                                                        const double *__restrict__ pA, int npA, EvalOut *out,
                                                        int seq, const double *__restrict__ part,
                                                        const double *__restrict__ agent_partials, int wpa) {
  __shared__ double s_red[16];
  __shared__ double s_bn[kMaxAgents];
  // the cost partials first (loads in flight under the per-agent sums below)
  double q0 = ((int)threadIdx.x < npA) ? pA[2 * threadIdx.x] : 0.0;
  double q1 = ((int)threadIdx.x < npA) ? pA[2 * threadIdx.x + 1] : 0.0;
  if (agent_partials) {  // wpa consecutive partials per agent, written by the evaluation itself (BsrGradOut): one wave
                         // per agent, eight loads in flight per lane, fixed order
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = kBlock / 64;
    for (int b = w; b < R; b += nw) {
      const int lo = b * wpa, hi = lo + wpa;
      double v = 0;
      for (int i0 = lo + lane; i0 < hi; i0 += 64 * 8) {
        double t8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t8[u] = agent_partials[min(i0 + 64 * u, hi - 1)];
#pragma unroll
        for (int u = 0; u < 8; ++u) v += (i0 + 64 * u < hi) ? t8[u] : 0.0;
      }
      v = f_wave_sum(v);
      if (lane == 0) s_bn[b] = v;
    }
  } else if (part) {  // the slices of k_eval_partial, in slice order
    if ((int)threadIdx.x < R) {
      double v = 0;
      for (int u = 0; u < kEvalSplit; ++u) v += part[threadIdx.x * kEvalSplit + u];
      s_bn[threadIdx.x] = v;
    }
  } else
  // one wave per agent (waves stride over the agents): eight loads in flight per lane, a wave-level sum, no barrier
  {
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = kBlock / 64;
    for (int b = w; b < R; b += nw) {
      const int lo = pose_start[b], hi = pose_start[b + 1];
      double v = 0;
      for (int i0 = lo + lane; i0 < hi; i0 += 64 * 8) {
        double t8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t8[u] = posenorm[min(i0 + 64 * u, hi - 1)];
#pragma unroll
        for (int u = 0; u < 8; ++u) v += (i0 + 64 * u < hi) ? t8[u] : 0.0;
      }
      v = f_wave_sum(v);
      if (lane == 0) s_bn[b] = v;
    }
  }
  // (the block evaluation of a large graph leaves thousands of partials: eight trips' loads in flight at once, added in
  // the order a plain loop would add them)
  for (int i0 = threadIdx.x + kBlock; i0 < npA; i0 += 8 * kBlock) {
    double a8[8], c8[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + u * kBlock;
      a8[u] = i < npA ? pA[2 * i] : 0.0;
      c8[u] = i < npA ? pA[2 * i + 1] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      q0 += a8[u];
      q1 += c8[u];
    }
  }
  const double fq = f_block_sum(q0, s_red);  // (its barriers also publish s_bn)
  const double fg = f_block_sum(q1, s_red);
  if (threadIdx.x == 0) {
    double g2 = 0, best = -1;
    int arg = 0;
    for (int b = 0; b < R; ++b) {
      const double nb = sqrt(s_bn[b]);
      out->block_norms[b] = nb;
      g2 += s_bn[b];
      if (nb > best) {
        best = nb;
        arg = b;
      }
    }
    out->cost2 = 2.0 * (0.5 * fq + fg);
    out->gradnorm = sqrt(g2);
    out->next = arg;
    __threadfence_system();
    __hip_atomic_store(const_cast<int *>(&out->seq), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
void launch_eval_finish(hipStream_t st, int R, const int *pose_start, const double *posenorm, const double *pA,
                        int npA, EvalOut *out_dev, int seq, double *split_scratch, int nposes,
                        const double *agent_partials, int wg_per_agent) {
  const bool split = !agent_partials && split_scratch && nposes >= 16384;
  if (split)
    hipLaunchKernelGGL(k_eval_partial, dim3(R * kEvalSplit), dim3(kBlock), 0, st, pose_start, posenorm, split_scratch);
  hipLaunchKernelGGL(k_eval_finish, dim3(1), dim3(kBlock), 0, st, R, pose_start, posenorm, pA, npA, out_dev, seq,
                     split ? split_scratch : nullptr, agent_partials, wg_per_agent);
}
int eval_split_doubles() { return kMaxAgents * kEvalSplit; }

// one chunk of 32 poses per workgroup up to kBsrMaxGrid workgroups (the Q-apply partial buffer holds that many
// slots): at 100k poses more resident workgroups mean more gathers in flight (34.9 us at 1024, 30.1 us at 2048)
int spmm_bsr_grid(int nbrows) {
  // one resident round: 8 workgroups of 256 threads per CU on 256 CUs
  const int cap = kBsrMaxGrid, per = kPosesPerBlock;
  long g = ((long)nbrows + per - 1) / per;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (int)g;
}
// the quad-per-pose form for every (d, r) the block structure is built for (d <= r <= 8)
template <bool DOTS, bool GRAD>
static bool launch_bsrq(hipStream_t st, int grid, int r, int d, const BsrDev &A, Buf2 X, int selX, const double *G, Buf2 Y,
                        int selY, double *partials, Gate g, const BsrGradOut &go) {
#define DCORA_BSRQ(D_, R_)                                                                                              \
  if (d == D_ && r == R_) {                                                                                             \
    hipLaunchKernelGGL((k_spmm_bsrq<D_, R_, DOTS, GRAD>), dim3(grid), dim3(kQBlock), 0, st, A, X, selX, G, Y, selY, \
                       partials, g, go);                                                                                \
    return true;                                                                                                        \
  }
  DCORA_BSRQ(3, 3) DCORA_BSRQ(3, 4) DCORA_BSRQ(3, 5) DCORA_BSRQ(3, 6) DCORA_BSRQ(3, 7) DCORA_BSRQ(3, 8)
  DCORA_BSRQ(2, 2) DCORA_BSRQ(2, 3) DCORA_BSRQ(2, 4) DCORA_BSRQ(2, 5) DCORA_BSRQ(2, 6) DCORA_BSRQ(2, 7) DCORA_BSRQ(2, 8)
#undef DCORA_BSRQ
  return false;
}
void launch_spmm_bsr(hipStream_t st, int r, int d, const BsrDev &A, Buf2 X, int selX, const double *G, Buf2 Y,
                     int selY, double *partials, Gate g) {
  const int grid = spmm_bsr_grid(A.nbrows);
  const BsrGradOut none{};
  const bool ok = partials ? launch_bsrq<true, false>(st, grid, r, d, A, X, selX, G, Y, selY, partials, g, none)
                           : launch_bsrq<false, false>(st, grid, r, d, A, X, selX, G, Y, selY, partials, g, none);
  if (!ok) throw std::logic_error("block Q-apply: no instantiation for this (d, r); the block structure is built for d <= r <= 8 only");
}
// EG = X Q + G, RG = Proj_X(EG), S blocks, partials {<XQ,X>, <X,G>} in pA (2 per block), |RG|^2 in pB (1 per block) and
// the per-pose norms in ONE launch on the block structure of Q; returns the number of blocks
int launch_fused_grad_bsr(hipStream_t st, int r, int d, const BsrDev &A, Buf2 X, const double *G, Buf2 EG, Buf2 RG, Buf2 S,
                          int sel, double *pA, double *pB, double *posenorm, Gate g, const int *agent_start, int agents,
                          int *wg_per_agent) {
  int grid = spmm_bsr_grid(A.nbrows);
  BsrGradOut go;
  go.RG = RG;
  go.S = S;
  go.pB = pB;
  go.posenorm = posenorm;
  if (agent_start && agents > 0 && wg_per_agent) {
    const int wpa = std::max(1, grid / agents);
    grid = wpa * agents;
    go.agent_start = agent_start;
    go.wg_per_agent = wpa;
    go.posenorm = nullptr;
    *wg_per_agent = wpa;
  }
  if (!launch_bsrq<true, true>(st, grid, r, d, A, X, sel, G, EG, sel, pA, g, go))
    throw std::logic_error("block evaluation: no instantiation for this (d, r); the block structure is built for d <= r <= 8 only");
  return grid;
}

// the 8-lanes-per-pose kernels (rgrad / retract / Nesterov / BSR Q-apply) walk the poses with a capped grid: any n
bool group_supported(const ManiDesc &m) { return m.se && m.r <= GW && m.n > 0; }
bool fused_supported(const ManiDesc &m) {
  // hess / finish leave one partial slot per block in 2 * kMaxPartials-slot buffers
  return m.se && m.r <= GW && m.n > 0 && (m.n + fused_pb(m.r, m.d + 1) - 1) / fused_pb(m.r, m.d + 1) <= 2 * kMaxPartials;
}
int fused_pose_blocks(const ManiDesc &m) {
  const int pb = fused_pb(m.r, m.d + 1);
  return (m.n + pb - 1) / pb;
}
int fused_nsplit(const ManiDesc &m) {
  const int njc = (m.k + kJChunk - 1) / kJChunk;
  const int aim = 512, cap = 32;
  int ns = (aim + njc - 1) / njc;  // aim for ~512 blocks of 4 waves
  if (ns < 1) ns = 1;
  if (ns > cap) ns = cap;
  while (ns > 1 && (m.k + ns - 1) / ns < 16) --ns;
  return ns;
}
int fused_precond_grid(const ManiDesc &m) { return ((m.k + kJChunk - 1) / kJChunk) * fused_nsplit(m); }
int fused_update_grid(const ManiDesc &m) {
  const long blocks = ((long)m.r * m.k + kBlock - 1) / kBlock;
  return (int)(blocks < 1024 ? blocks : 1024);
}

int launch_fused_hess(hipStream_t st, const ManiDesc &m, const CsrDev &Q, const double *z, const double *d_old,
                      double *d_new, Buf2 X, Buf2 S, double *Hd, const double *p3, int np3, double *p1,
                      SolverCtl *ctl, int seq, int iter, const BsrDev *Ab) {
  if (Ab) {  // block structure available: 32 poses per workgroup
    const int gridb = (m.n + kPosesPerBlock - 1) / kPosesPerBlock;
    if (m.d == 3)
      hipLaunchKernelGGL(k_fused_hess_bsr<3>, dim3(gridb), dim3(kBlock), 0, st, m, *Ab, z, d_old, d_new, X, S, Hd, p3,
                         np3, p1, ctl, seq, iter);
    else
      hipLaunchKernelGGL(k_fused_hess_bsr<2>, dim3(gridb), dim3(kBlock), 0, st, m, *Ab, z, d_old, d_new, X, S, Hd, p3,
                         np3, p1, ctl, seq, iter);
    return gridb;
  }
  const int grid = fused_pose_blocks(m);
  if (m.d == 3)
    hipLaunchKernelGGL(k_fused_hess<3>, dim3(grid), dim3(kBlock), 0, st, m, Q, z, d_old, d_new, X, S, Hd, p3, np3,
                       p1, ctl, seq, iter);
  else
    hipLaunchKernelGGL(k_fused_hess<2>, dim3(grid), dim3(kBlock), 0, st, m, Q, z, d_old, d_new, X, S, Hd, p3, np3,
                       p1, ctl, seq, iter);
  return grid;
}
void launch_fused_precond(hipStream_t st, const ManiDesc &m, int ldm, const double *Minv, Buf2 grad,
                          const double *delta, const double *Hd, double *eta, double *Heta, const double *res_old,
                          double *res_new, double *Zpart, const double *p1, int np1, double *p2, SolverCtl *ctl,
                          HostFlags *hf, int seq, int iter, int first, SpFold sf) {
  const int grid = Minv ? fused_precond_grid(m) : fused_update_grid(m);
  const int ns = Minv ? fused_nsplit(m) : 1;
#define DCORA_LAUNCH_PRECOND(RM, HM)                                                                               \
  hipLaunchKernelGGL((k_fused_precond<RM, HM>), dim3(grid), dim3(kBlock), 0, st, m.r, m.k, ldm, ns, Minv, grad,     \
                     delta, Hd, eta, Heta, res_old, res_new, Zpart, p1, np1, p2, ctl, hf, seq, iter, first, sf)
  if (m.r <= 4) {
    if (Minv) DCORA_LAUNCH_PRECOND(4, true); else DCORA_LAUNCH_PRECOND(4, false);
  } else {
    if (Minv) DCORA_LAUNCH_PRECOND(8, true); else DCORA_LAUNCH_PRECOND(8, false);
  }
#undef DCORA_LAUNCH_PRECOND
}
// poses per workgroup of the one-launch B + C: about one workgroup per CU at the headline size
int fused_pc_pb(const ManiDesc &m) { return m.n <= 768 ? 2 : 4; }
int fused_pc_blocks(const ManiDesc &m) {
  const int pb = fused_pc_pb(m);
  return (m.n + pb - 1) / pb;
}
constexpr int kPcLdsCap = 128 * 1024;  // residual chunk in LDS: as many 128-column steps as fit (all of it at k = 2000)
static int pc_chunk(const ManiDesc &m, int ldm) { return std::min((kPcLdsCap / (8 * m.r)) / 128 * 128, ldm); }
// where the one-launch form wins (measured on MI355X, sphere2500 blocks): the whole residual in one LDS chunk and one
// staging batch, r <= 7 -- k = 2000: 11.0 us against 11.3 + 5.1 us for B + C at r = 5, 13.4 / 17.6 at r = 6, 18.0 /
// 18.6 at r = 7; beyond (k = 3332: 29.2 / 29.2, k = 5000: 76 / 48, r = 8: 21.0 / 20.0) the split form stays
bool fused_pc_preferred(const ManiDesc &m, int ldm) {
  return m.k <= pc_chunk(m, ldm) && (long)m.r * m.k <= 2L * kPcSB * kPcBlock && m.r <= 7;
}
template <int D, int PB, int R, bool MULTI>
static int pc_launch(hipStream_t st, const ManiDesc &m, int ldm, const double *Minv, Buf2 grad, Buf2 X,
                     const double *delta, const double *Hd, double *eta, double *Heta, const double *res_old,
                     double *res_new, double *z, const double *p1, int np1, double *p3, SolverCtl *ctl, HostFlags *hf,
                     int seq, int iter, int first, double *pC, bool prepare_only) {
  // The dynamic-LDS limit is an attribute of (function, DEVICE): it is set once per device the instantiation runs on
  // (one bit per device; a process drives R GPUs from R host threads, SURVEY 8(b) threading).  -1 = this device refuses
  // the attribute or the launch; use_pc() asks with st == nullptr before choosing the one-launch form.
  static std::atomic<unsigned long long> tried{0}, ok{0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return -1;
  const unsigned long long bit = 1ull << dev;
  if (!(tried.load(std::memory_order_acquire) & bit)) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_fused_pc<D, PB, R, MULTI>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, kPcLdsCap);
    if (e == hipSuccess)
      ok.fetch_or(bit, std::memory_order_release);
    else
      (void)hipGetLastError();
    tried.fetch_or(bit, std::memory_order_release);
  }
  const int chk = pc_chunk(m, ldm);
  const size_t lds = (size_t)chk * m.r * sizeof(double);
  if (!(ok.load(std::memory_order_acquire) & bit) && lds > 64 * 1024) return -1;
  const int grid = (m.n + PB - 1) / PB;
  if (prepare_only) return grid;
  hipLaunchKernelGGL((k_fused_pc<D, PB, R, MULTI>), dim3(grid), dim3(kPcBlock), lds, st, m, ldm, chk, Minv, grad, X,
                     delta, Hd, eta, Heta, res_old, res_new, z, p1, np1, p3, ctl, hf, seq, iter, first, pC);
  if (hipGetLastError() != hipSuccess) return -1;
  return grid;
}
static int fused_pc_dispatch(hipStream_t st, const ManiDesc &m, int ldm, const double *Minv, Buf2 grad, Buf2 X,
                             const double *delta, const double *Hd, double *eta, double *Heta, const double *res_old,
                             double *res_new, double *z, const double *p1, int np1, double *p3, SolverCtl *ctl,
                             HostFlags *hf, int seq, int iter, int first, double *pC, bool prepare_only) {
#define DCORA_PC(D_, PB_, R_, MULTI_)                                                                                \
  return pc_launch<D_, PB_, R_, MULTI_>(st, m, ldm, Minv, grad, X, delta, Hd, eta, Heta, res_old, res_new, z, p1, np1, \
                                        p3, ctl, hf, seq, iter, first, pC, prepare_only)
#define DCORA_PC_FIXED_R(D_, PB_, MULTI_)      \
  do {                                          \
    if (D_ == 3 && m.r == 3) DCORA_PC(D_, PB_, 3, MULTI_); \
    if (D_ == 3 && m.r == 4) DCORA_PC(D_, PB_, 4, MULTI_); \
    if (D_ == 3 && m.r == 5) DCORA_PC(D_, PB_, 5, MULTI_); \
    if (D_ == 3 && m.r == 6) DCORA_PC(D_, PB_, 6, MULTI_); \
    if (D_ == 3 && m.r == 7) DCORA_PC(D_, PB_, 7, MULTI_); \
  } while (0)
#define DCORA_PC_R(D_, PB_, MULTI_)             \
  do {                                          \
    DCORA_PC_FIXED_R(D_, PB_, MULTI_);          \
    DCORA_PC(D_, PB_, 0, MULTI_);               \
  } while (0)
  const int pb = fused_pc_pb(m);
  const bool multi = m.k > pc_chunk(m, ldm);
  if (m.d == 3) {
    if (pb == 2 && !multi) DCORA_PC_R(3, 2, false);
    if (pb == 2) DCORA_PC_R(3, 2, true);
    // four poses per thread with r only known at run time does not fit the register file (208 B/lane of scratch when
    // it was instantiated), and neither does r = 7 (128 B/lane): r >= 7 beyond 768 poses takes the three-launch form,
    // which fused_pc_preferred() chooses there anyway
    if (m.r == 3) DCORA_PC(3, 4, 3, true);
    if (m.r == 4) DCORA_PC(3, 4, 4, true);
    if (m.r == 5) DCORA_PC(3, 4, 5, true);
    if (m.r == 6) DCORA_PC(3, 4, 6, true);
    return -1;
  }
  if (pb == 2 && !multi) DCORA_PC(2, 2, 0, false);
  if (pb == 2) DCORA_PC(2, 2, 0, true);
  DCORA_PC(2, 4, 0, true);
#undef DCORA_PC_R
#undef DCORA_PC_FIXED_R
#undef DCORA_PC
}
// ---- the one-launch tCG run ------------------------------------------------------------------------------
constexpr int kRunNS = 4;  // 128-column steps per wave held in registers: k <= 4 * 4 * 128 = 2048
int tcg_run_sync_words() { return kRunSyncWords; }
std::atomic<int> g_tcg_run_fault{0};
std::atomic<int> g_tcg_run_fault_skip{0};
int tcg_run_max_rows_nnz(const ManiDesc &m, const int *rp) {
  const int dh = m.d + 1;
  int worst = 0;
  for (int p0 = 0; p0 < m.n; p0 += 2) {
    const int j0 = p0 * dh, j1 = std::min(m.n, p0 + 2) * dh;
    worst = std::max(worst, rp[j1] - rp[j0]);
  }
  return worst;
}
bool tcg_run_supported(const ManiDesc &m, int ldm, int cus, int max_rows_nnz) {
  if (!m.se || m.d != 3 || m.r < 4 || m.r > 6) return false;    // instantiated ranks; r = 4, 5, 6: fused_pb even
  if (fused_pb(m.r, m.d + 1) % 2 != 0) return false;               // row sums pair up with k_fused_hess's workgroups
  if (!fused_pc_preferred(m, ldm) || fused_pc_pb(m) != 2) return false;
  const int grid = (m.n + 1) / 2;
  if (grid > cus || grid > 256) return false;                      // co-resident, <z, r> partials in one wave load
  if (m.k > kRunNS * kPcNW * 128) return false;                    // the rows of the inverse fit the registers
  const int rows_per = fused_pb(m.r, m.d + 1) / 2;
  if ((grid + rows_per - 1) / rows_per > 64) return false;         // <delta, H delta> partials in one wave
  return max_rows_nnz <= kRunQCap;
}
template <int R>
static int tcg_run_launch(hipStream_t st, const TcgRunArgs &a) {
  static std::atomic<unsigned long long> tried{0}, ok{0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return -1;
  const unsigned long long bit = 1ull << dev;
  if (!(tried.load(std::memory_order_acquire) & bit)) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_tcg_run<3, R, kRunNS>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, kPcLdsCap);
    if (e == hipSuccess)
      ok.fetch_or(bit, std::memory_order_release);
    else
      (void)hipGetLastError();
    tried.fetch_or(bit, std::memory_order_release);
  }
  const int cpad = ((a.m.k + 127) / 128) * 128;
  const size_t lds = (size_t)cpad * R * sizeof(double);
  if (!(ok.load(std::memory_order_acquire) & bit) && lds > 40 * 1024) return -1;
  const int grid = (a.m.n + 1) / 2;
  hipLaunchKernelGGL((k_tcg_run<3, R, kRunNS>), dim3(grid), dim3(kPcBlock), lds, st, a);
  if (hipGetLastError() != hipSuccess) return -1;
  return grid;
}
int launch_tcg_run(hipStream_t st, const ManiDesc &m, int ldm, const double *Minv, const CsrDev &Q, Buf2 grad, Buf2 X,
                   Buf2 S, double *d0, double *d1, double *Hd, double *eta, double *Heta, double *z, double *p1r,
                   double *p3, double *pC, unsigned *sync, SolverCtl *ctl, HostFlags *hf, int seq) {
  int fault = 0;
  bool skipped = false;
  for (int sk = g_tcg_run_fault_skip.load(); sk > 0;)
    if (g_tcg_run_fault_skip.compare_exchange_weak(sk, sk - 1)) {
      skipped = true;
      break;
    }
  for (int left = skipped ? 0 : g_tcg_run_fault.load(); left > 0;)
    if (g_tcg_run_fault.compare_exchange_weak(left, left - 1)) {
      fault = 1;
      break;
    }
  TcgRunArgs a{m, ldm, Minv, Q, grad, X, S, d0, d1, Hd, eta, Heta, z, p1r, p3, pC, sync, ctl, hf, seq,
               fused_pb(m.r, m.d + 1), fault};
#ifdef DCORA_RUN_STAMPS
  {
    static long long *buf = nullptr;
    if (!buf) (void)hipHostMalloc((void **)&buf, 64 * sizeof(long long), hipHostMallocMapped);
    static int count = 0;
    if (buf && count > 0) {  // the stamps of the run before (the stream is in order; racy by a run at most: a profile)
      static double acc[64];
      static int n = 0;
      long long s0 = buf[0];
      if (buf[3] > s0 && buf[5 + 7 * 3] > 0) {
        for (int i = 0; i < 40; ++i) acc[i] += (double)(buf[i] - s0) * 0.01;
        ++n;
      }
      for (int i = 0; i < 64; ++i) buf[i] = 0;
      if (n == 500) {
        fprintf(stderr, "k_tcg_run stamps (us from start, workgroup 0, mean of %d runs with >= 4 iterations):", n);
        for (int i = 0; i < 34; ++i) fprintf(stderr, " %.2f", acc[i] / n);
        fprintf(stderr, "\n");
        n = -1000000;
      }
    }
    ++count;
    a.stamps = buf;
  }
#endif
  if (m.r == 4) return tcg_run_launch<4>(st, a);
  if (m.r == 5) return tcg_run_launch<5>(st, a);
  if (m.r == 6) return tcg_run_launch<6>(st, a);
  return -1;
}
int launch_fused_pc(hipStream_t st, const ManiDesc &m, int ldm, const double *Minv, Buf2 grad, Buf2 X,
                    const double *delta, const double *Hd, double *eta, double *Heta, const double *res_old,
                    double *res_new, double *z, const double *p1, int np1, double *p3, SolverCtl *ctl, HostFlags *hf,
                    int seq, int iter, int first, double *pC) {
  return fused_pc_dispatch(st, m, ldm, Minv, grad, X, delta, Hd, eta, Heta, res_old, res_new, z, p1, np1, p3, ctl, hf,
                           seq, iter, first, pC, false);
}
bool fused_pc_ready(const ManiDesc &m, int ldm) {
  Buf2 none{};
  return fused_pc_dispatch(nullptr, m, ldm, nullptr, none, none, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                           nullptr, nullptr, 0, nullptr, nullptr, nullptr, 0, 0, 0, nullptr, true) >= 0;
}
void launch_fused_finish(hipStream_t st, const ManiDesc &m, Buf2 X, const double *Zpart, const double *res,
                         double *z, const double *p2, int np2, double *p3, SolverCtl *ctl, HostFlags *hf, int seq,
                         int iter, int first, int nsplit, SpFold sf, double *zraw) {
  const int grid = fused_pose_blocks(m);
  const int ns = nsplit > 0 ? nsplit : fused_nsplit(m);
  if (m.d == 3)
    hipLaunchKernelGGL(k_fused_finish<3>, dim3(grid), dim3(kBlock), 0, st, m, ns, X, Zpart, res, z, p2, np2, p3, ctl,
                       hf, seq, iter, first, sf, zraw);
  else
    hipLaunchKernelGGL(k_fused_finish<2>, dim3(grid), dim3(kBlock), 0, st, m, ns, X, Zpart, res, z, p2, np2, p3, ctl,
                       hf, seq, iter, first, sf, zraw);
}

int launch_g_rgrad(hipStream_t st, const ManiDesc &m, Buf2 X, Buf2 EG, Buf2 RG, Buf2 Sblk, int sel, double *partials,
                   double *posenorm, Gate g) {
  const int grid = group_grid(m.n);
  if (m.d == 3)
    hipLaunchKernelGGL(k_g_rgrad<3>, dim3(grid), dim3(kBlock), 0, st, m, X, EG, RG, Sblk, sel, partials, posenorm, g);
  else
    hipLaunchKernelGGL(k_g_rgrad<2>, dim3(grid), dim3(kBlock), 0, st, m, X, EG, RG, Sblk, sel, partials, posenorm, g);
  return grid;
}
// EG = X Q + G, RG = Proj_X(EG), S blocks, partials {<XQ,X>, <X,G>} in pA (2 per block) and |RG|^2 in pB (1 per
// block); returns the number of blocks.  Small SE blocks without long rows only (the caller checks).
int launch_fused_grad(hipStream_t st, const ManiDesc &m, const CsrDev &Q, Buf2 X, const double *G, Buf2 EG, Buf2 RG,
                      Buf2 Sblk, int sel, double *pA, double *pB, double *posenorm, Gate g) {
  const int grid = fused_pose_blocks(m);
  if (m.d == 3)
    hipLaunchKernelGGL(k_fused_grad<3>, dim3(grid), dim3(kBlock), 0, st, m, Q, X, G, EG, RG, Sblk, sel, pA, pB, posenorm,
                       g);
  else
    hipLaunchKernelGGL(k_fused_grad<2>, dim3(grid), dim3(kBlock), 0, st, m, Q, X, G, EG, RG, Sblk, sel, pA, pB, posenorm,
                       g);
  return grid;
}
int launch_g_retract(hipStream_t st, const ManiDesc &m, Buf2 X, const double *V, double alpha, Buf2 out, int selOut,
                     Buf2 grad, const double *HV, double *partials, Gate g) {
  const int grid = group_grid(m.n);
  if (m.d == 3)
    hipLaunchKernelGGL(k_g_retract<3>, dim3(grid), dim3(kBlock), 0, st, m, X, V, alpha, out, selOut, grad, HV,
                       partials, g);
  else
    hipLaunchKernelGGL(k_g_retract<2>, dim3(grid), dim3(kBlock), 0, st, m, X, V, alpha, out, selOut, grad, HV,
                       partials, g);
  return grid;
}
void launch_g_nesterov(hipStream_t st, const ManiDesc &m, int mode, int restart, int skip_lo, int skip_hi,
                       double alpha, double gamma, double *X, double *V, double *Y, double *XPrev, double *Yloc,
                       Buf2 Xloc, const SolverCtl *ctl, double *inner_Yloc) {
  GNesterovArgs a{mode, restart, skip_lo, skip_hi, alpha, gamma, X, V, Y, XPrev, Yloc, inner_Yloc, Xloc, ctl};
  const int grid = group_grid(m.n);
  if (m.d == 3)
    hipLaunchKernelGGL(k_g_nesterov<3>, dim3(grid), dim3(kBlock), 0, st, m, a);
  else
    hipLaunchKernelGGL(k_g_nesterov<2>, dim3(grid), dim3(kBlock), 0, st, m, a);
}

}  // namespace dcora
