// Hand-written HIP kernels for gfx950 (MI355X): connection-Laplacian SpMM, per-pose manifold arithmetic
// (tangent projection, Riemannian Hessian correction, QF retraction, polar projection), dense preconditioner
// apply, and the device-resident scalar logic of the truncated-CG / trust-region solver.
//
// Design notes:
//  * wavefront = 64; all block-level reductions are wave shuffles + one LDS hop, in a fixed order;
//  * reductions leave per-block partials in HBM; the consumer kernel sums them in its prologue, so a
//    dot product costs no extra launch and the result is bitwise reproducible;
//  * solver kernels read their trust-region / tCG scalars from a SolverCtl block in HBM and are no-ops once a
//    termination stamp older than their own sequence number is set -- the host never waits for a scalar.
#include "kernels.h"

namespace dcora {

// ------------------------------------------------------------------------------------------------------
// helpers
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool gated(const SolverCtl *ctl, int seq, int gate) {
  if (ctl == nullptr || gate == 0) return false;
  if (seq > ctl->outer_done_stamp) return true;
  if (gate == 2 && seq > ctl->tcg_done_stamp) return true;
  return false;
}
// the gate's two words requested early (with a kernel's other independent loads), tested later
struct GateWords {
  int outer, tcg;
};
__device__ __forceinline__ GateWords gate_words(const SolverCtl *ctl) {
  GateWords w{0x7fffffff, 0x7fffffff};
  if (ctl) {
    w.outer = ctl->outer_done_stamp;
    w.tcg = ctl->tcg_done_stamp;
  }
  return w;
}
__device__ __forceinline__ bool gated(const GateWords &w, const SolverCtl *ctl, int seq, int gate) {
  if (ctl == nullptr || gate == 0) return false;
  if (seq > w.outer) return true;
  if (gate == 2 && seq > w.tcg) return true;
  return false;
}
__device__ __forceinline__ double *pick(const Buf2 &b, const SolverCtl *ctl, int sel) {
  return b.p[ctl ? ((ctl->cur ^ sel) & 1) : 0];
}
__device__ __forceinline__ double wave_sum(double v) { return wave_sum_dpp(v); }
// sum over the block, result broadcast to every thread; sm must hold >= 16 doubles
__device__ __forceinline__ double block_sum(double v, double *sm) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm[w] = v;
  __syncthreads();
  double t = 0;
  for (int i = 0; i < nw; ++i) t += sm[i];
  return t;
}
__device__ __forceinline__ double sum_partials(const double *p, int np, int stride, int off, double *sm) {
  double v = 0;
  for (int i = threadIdx.x; i < np; i += blockDim.x) v += p[(size_t)i * stride + off];
  return block_sum(v, sm);
}
// N sums of partial arrays at once for the single-block bookkeeping kernels: every load is issued before the first
// reduction (one memory round trip instead of one per sum) and the N wave sums share one LDS exchange (two barriers
// instead of 2 N).  p[q] has np[q] entries of stride st[q] at offset off[q]; sm must hold >= 4 N doubles.
template <int N>
__device__ __forceinline__ void sum_partials_n(const double *const (&p)[N], const int (&np)[N], const int (&st)[N],
                                               const int (&off)[N], double *sm, double (&out)[N]) {
  double v[N];
#pragma unroll
  for (int q = 0; q < N; ++q) v[q] = ((int)threadIdx.x < np[q]) ? p[q][(size_t)threadIdx.x * st[q] + off[q]] : 0.0;
#pragma unroll
  for (int q = 0; q < N; ++q)
    for (int i = threadIdx.x + blockDim.x; i < np[q]; i += blockDim.x) v[q] += p[q][(size_t)i * st[q] + off[q]];
#pragma unroll
  for (int q = 0; q < N; ++q) v[q] = wave_sum(v[q]);
  const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0)
#pragma unroll
    for (int q = 0; q < N; ++q) sm[q * 4 + w] = v[q];
  __syncthreads();
#pragma unroll
  for (int q = 0; q < N; ++q) {
    double t = 0;
    for (int i = 0; i < nw; ++i) t += sm[q * 4 + i];
    out[q] = t;
  }
}
__device__ __forceinline__ void host_store(volatile int *p, int v) {
  __hip_atomic_store(const_cast<int *>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

int vec_grid(long nelem) {
  long g = (nelem + kBlock - 1) / kBlock;
  if (g < 1) g = 1;
  if (g > kMaxPartials) g = kMaxPartials;
  return (int)g;
}

// ------------------------------------------------------------------------------------------------------
// SpMM  Y = X * A (+ G).  One thread per output element (column j, component t): the r lanes of a row read r
// contiguous doubles of X(:, c) and share one (value, column) pair.  The CSR segment of the block's rows is
// staged in LDS with fully coalesced loads (row blocks of a connection Laplacian are contiguous in CSR).
// ------------------------------------------------------------------------------------------------------
constexpr int kSpmmTile = 1536;  // nnz staged per pass: 18 KiB of LDS

int spmm_grid(int nrows, int r) {
  const int RB = kBlock / r;
  long nrb = (nrows + RB - 1) / RB;
  if (nrb < 1) nrb = 1;
  if (nrb > kMaxPartials) nrb = kMaxPartials;
  return (int)nrb;
}

template <bool DOTS>
__global__ __launch_bounds__(kBlock) void k_spmm(int r, CsrDev A, Buf2 Xb, int selX, const double *__restrict__ G,
                                                 Buf2 Yb, int selY, double *__restrict__ partials, Gate g,
                                                 int main_grid) {
  if (gated(g.ctl, g.seq, g.gate)) return;
  __shared__ int s_ci[kSpmmTile];
  __shared__ double s_v[kSpmmTile];
  __shared__ double s_red[16];
  const double *__restrict__ X = pick(Xb, g.ctl, selX);
  double *__restrict__ Y = pick(Yb, g.ctl, selY);
  const int RB = kBlock / r;
  const int nrb = (A.nrows + RB - 1) / RB;
  const int lj = threadIdx.x / r, t = threadIdx.x - lj * r;
  double d0 = 0, d1 = 0;
  if ((int)blockIdx.x >= main_grid) {
    // kLongSplit workgroups per long row (a landmark ranged from 7789 poses on tiers.pyfg: one workgroup was the
    // longest of the launch): each takes a slice, the RB entry groups stride over it, partial sums meet in LDS and
    // go to a scratch row; the last workgroup to arrive adds the slices in slice order (reproducible) and finishes
    const int li = ((int)blockIdx.x - main_grid) / kLongSplit, sl = ((int)blockIdx.x - main_grid) % kLongSplit;
    const int j = A.long_rows[li];
    const int rb0 = A.rp[j], re0 = A.rp[j + 1];
    const int per = (re0 - rb0 + kLongSplit - 1) / kLongSplit;
    const int pb = rb0 + sl * per, pe = min(re0, pb + per);
    double acc = 0;
    if (lj < RB) {
      // eight entries per step with every load in flight before the first use (index clamped, weight masked)
      for (int p = pb + lj; p < pe; p += 8 * RB) {
        int c8[8];
        double w8[8], x8[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int pp = p + q * RB;
          const bool ok = pp < pe;
          c8[q] = A.ci[ok ? pp : rb0];
          w8[q] = ok ? A.v[pp] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) x8[q] = X[(size_t)c8[q] * r + t];
#pragma unroll
        for (int q = 0; q < 8; ++q) acc += w8[q] * x8[q];
      }
    }
    double *s_part = s_v;  // kBlock doubles
    __shared__ int s_last;
    __syncthreads();
    s_part[threadIdx.x] = (lj < RB) ? acc : 0.0;
    __syncthreads();
    if ((int)threadIdx.x < r) {
      double y = 0;
      for (int q = 0; q < RB; ++q) y += s_part[q * r + threadIdx.x];
      __hip_atomic_store(A.long_part + ((size_t)li * kLongSplit + sl) * 16 + threadIdx.x, y, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
    }
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0)
      s_last = (__hip_atomic_fetch_add(A.long_cnt + li, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) ==
                kLongSplit - 1);
    __syncthreads();
    if (!s_last) return;
    if (threadIdx.x == 0) __hip_atomic_store(A.long_cnt + li, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((int)threadIdx.x < r) {
      double y = 0;
      for (int q = 0; q < kLongSplit; ++q)
        y += __hip_atomic_load(A.long_part + ((size_t)li * kLongSplit + q) * 16 + threadIdx.x, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
      const size_t o = (size_t)j * r + threadIdx.x;
      if (DOTS) {
        const double x = X[o];
        d0 = y * x;
        if (G) d1 = x * G[o];
      }
      if (G) y += G[o];
      Y[o] = y;
    }
    if (DOTS) {
      const double a = block_sum(d0, s_red);
      const double b = block_sum(d1, s_red);
      if (threadIdx.x == 0) {
        partials[2 * (main_grid + li)] = a;
        partials[2 * (main_grid + li) + 1] = b;
      }
    }
    return;
  }
  for (int rb = blockIdx.x; rb < nrb; rb += main_grid) {
    const int j0 = rb * RB;
    const int j1 = min(A.nrows, j0 + RB);
    const int j = j0 + lj;
    const bool active = (lj < RB) && (j < j1);
    const int pbeg = A.rp[j0], pend = A.rp[j1];
    int myb = active ? A.rp[j] : 0, mye = active ? A.rp[j + 1] : 0;
    const bool is_long = A.n_long > 0 && (mye - myb > kLongRow);  // served by its own block
    if (is_long) mye = myb;
    double acc = 0;
    for (int base = pbeg; base < pend; base += kSpmmTile) {
      const int cnt = min(kSpmmTile, pend - base);
      __syncthreads();
      {
        // all trips' loads of the tile are issued (clamped index, straight line) before any is stored to LDS:
        // one memory round trip per tile instead of one per 256 entries
        constexpr int SU = kSpmmTile / kBlock;
        int ci_r[SU];
        double v_r[SU];
        const int last = base + cnt - 1;
#pragma unroll
        for (int u = 0; u < SU; ++u) {
          const int i = min(base + (int)threadIdx.x + u * kBlock, last);
          ci_r[u] = A.ci[i];
          v_r[u] = A.v[i];
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
      // gathers in batches of 8 with every load issued before the first use (index clamped, weight masked)
      for (int p = lo; p < hi; p += 8) {
        double x8[8], w8[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const bool ok = p + q < hi;
          const int pp = ok ? p + q : lo;
          w8[q] = ok ? s_v[pp] : 0.0;
          x8[q] = X[(size_t)s_ci[pp] * r + t];
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) acc += w8[q] * x8[q];
      }
    }
    if (active && !is_long) {
      const size_t o = (size_t)j * r + t;
      double y = acc;
      if (DOTS) {
        const double x = X[o];
        d0 += acc * x;
        if (G) d1 += x * G[o];
      }
      if (G) y += G[o];
      Y[o] = y;
    }
  }
  if (DOTS) {
    const double a = block_sum(d0, s_red);
    const double b = block_sum(d1, s_red);
    if (threadIdx.x == 0) {
      partials[2 * blockIdx.x] = a;
      partials[2 * blockIdx.x + 1] = b;
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// Generic-layout tCG, iteration `iter`: the direction update of the previous iteration folded into the Hessian SpMM.
//   delta_new = -z + beta delta_old   (beta = <z, r>_new / <z, r>_old from the partials p3; iter 0: delta_new = -z)
//   W = delta_new Q                   (delta_new formed in the gather, written for the block's own columns)
// and the scalar recurrence of ROPTLIB's tCG_TR that k_tcg_init / k_tcg_update2 keep (block 0).  delta_old and
// delta_new are different buffers: other workgroups still gather the old direction.
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_spmm_dir(int r, CsrDev A, const double *__restrict__ z,
                                                     const double *__restrict__ d_old, double *__restrict__ d_new,
                                                     double *__restrict__ W, const double *__restrict__ p3, int np3,
                                                     SolverCtl *ctl, int seq, int iter, int main_grid) {
  if (gated(ctl, seq, 2)) return;
  __shared__ int s_ci[kSpmmTile];
  __shared__ double s_v[kSpmmTile];
  __shared__ double s_red[16];
  __shared__ int s_last;
  const int par = (iter - 1) & 1;
  const double z_r_new = sum_partials(p3, np3, 1, 0, s_red);
  const double beta = iter > 0 ? z_r_new / ctl->z_r[par] : 0.0;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    if (iter == 0) {
      ctl->z_r[0] = z_r_new;
      ctl->d_Pd[0] = z_r_new;
      ctl->e_Pe[0] = 0;
      ctl->e_Pd[0] = 0;
    } else {
      const double alpha = ctl->alpha;
      const double d_Pd = ctl->d_Pd[par], e_Pd = ctl->e_Pd[par];
      ctl->z_r[par ^ 1] = z_r_new;
      ctl->e_Pd[par ^ 1] = beta * (e_Pd + alpha * d_Pd);
      ctl->d_Pd[par ^ 1] = z_r_new + beta * beta * d_Pd;
      ctl->e_Pe[par ^ 1] = ctl->e_Pe_n;
    }
  }
  const int RB = kBlock / r;
  const int nrb = (A.nrows + RB - 1) / RB;
  const int lj = threadIdx.x / r, t = threadIdx.x - lj * r;
  auto dir = [&](size_t o) -> double { return iter > 0 ? fma(beta, d_old[o], -z[o]) : -z[o]; };
  if ((int)blockIdx.x >= main_grid) {  // a slice of a long row (see k_spmm)
    const int li = ((int)blockIdx.x - main_grid) / kLongSplit, sl = ((int)blockIdx.x - main_grid) % kLongSplit;
    const int j = A.long_rows[li];
    const int rb0 = A.rp[j], re0 = A.rp[j + 1];
    const int per = (re0 - rb0 + kLongSplit - 1) / kLongSplit;
    const int pb = rb0 + sl * per, pe = min(re0, pb + per);
    double acc = 0;
    if (lj < RB) {
      for (int p = pb + lj; p < pe; p += 8 * RB) {
        int c8[8];
        double w8[8], x8[8], y8[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int pp = p + q * RB;
          const bool ok = pp < pe;
          c8[q] = A.ci[ok ? pp : rb0];
          w8[q] = ok ? A.v[pp] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const size_t o = (size_t)c8[q] * r + t;
          x8[q] = z[o];
          y8[q] = iter > 0 ? d_old[o] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) acc += w8[q] * fma(beta, y8[q], -x8[q]);
      }
    }
    double *s_part = s_v;
    __syncthreads();
    s_part[threadIdx.x] = (lj < RB) ? acc : 0.0;
    __syncthreads();
    if ((int)threadIdx.x < r) {
      double y = 0;
      for (int q = 0; q < RB; ++q) y += s_part[q * r + threadIdx.x];
      __hip_atomic_store(A.long_part + ((size_t)li * kLongSplit + sl) * 16 + threadIdx.x, y, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
    }
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0)
      s_last = (__hip_atomic_fetch_add(A.long_cnt + li, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) ==
                kLongSplit - 1);
    __syncthreads();
    if (!s_last) return;
    if (threadIdx.x == 0) __hip_atomic_store(A.long_cnt + li, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((int)threadIdx.x < r) {
      double y = 0;
      for (int q = 0; q < kLongSplit; ++q)
        y += __hip_atomic_load(A.long_part + ((size_t)li * kLongSplit + q) * 16 + threadIdx.x, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
      const size_t o = (size_t)j * r + threadIdx.x;
      W[o] = y;
      d_new[o] = dir(o);
    }
    return;
  }
  for (int rb = blockIdx.x; rb < nrb; rb += main_grid) {
    const int j0 = rb * RB;
    const int j1 = min(A.nrows, j0 + RB);
    const int j = j0 + lj;
    const bool active = (lj < RB) && (j < j1);
    const int pbeg = A.rp[j0], pend = A.rp[j1];
    int myb = active ? A.rp[j] : 0, mye = active ? A.rp[j + 1] : 0;
    const bool is_long = A.n_long > 0 && (mye - myb > kLongRow);  // served by its own workgroups
    if (is_long) mye = myb;
    const double own = (active && !is_long) ? dir((size_t)j * r + t) : 0.0;
    double acc = 0;
    for (int base = pbeg; base < pend; base += kSpmmTile) {
      const int cnt = min(kSpmmTile, pend - base);
      __syncthreads();
      {
        constexpr int SU = kSpmmTile / kBlock;
        int ci_r[SU];
        double v_r[SU];
        const int last = base + cnt - 1;
#pragma unroll
        for (int u = 0; u < SU; ++u) {
          const int i = min(base + (int)threadIdx.x + u * kBlock, last);
          ci_r[u] = A.ci[i];
          v_r[u] = A.v[i];
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
        double x8[8], y8[8], w8[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const bool ok = p + q < hi;
          const int pp = ok ? p + q : lo;
          w8[q] = ok ? s_v[pp] : 0.0;
          const size_t o = (size_t)s_ci[pp] * r + t;
          x8[q] = z[o];
          y8[q] = iter > 0 ? d_old[o] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) acc += w8[q] * fma(beta, y8[q], -x8[q]);
      }
    }
    if (active && !is_long) {
      const size_t o = (size_t)j * r + t;
      W[o] = acc;
      d_new[o] = own;
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// The same with ROPTLIB's EucHvToHv (k_hessfix) folded in: one launch per tCG iteration instead of two.
//   delta_new = -z + beta delta_old,  W = delta_new Q,  Hd = Proj_X(W - delta_new S),  partial <delta_new, Hd>
// A workgroup owns whole manifold items: its rows-per-block count is a multiple of the rotation block's width (d,
// or d + 1 in the pose layout), so the d columns a Stiefel projection couples sit in one workgroup and meet in LDS.
// Long rows (served by their own workgroups) must be Euclidean columns (spmm_dir_fix_ok): their Hd is W itself.
// The arithmetic follows k_hessfix term by term (sub_AS, sym_gram, sub_AS); only the order in which the partial
// sums of <delta, Hd> are added differs.
// ------------------------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(kBlock) void k_spmm_dir_fix(ManiDesc m, CsrDev A, Buf2 Xb, Buf2 Sb,
                                                         const double *__restrict__ z,
                                                         const double *__restrict__ d_old, double *__restrict__ d_new,
                                                         double *__restrict__ Hd, const double *__restrict__ p3,
                                                         int np3, double *__restrict__ p1, SolverCtl *ctl, int seq,
                                                         int iter, int main_grid) {
  __shared__ int s_ci[kSpmmTile];
  __shared__ double s_v[kSpmmTile];
  __shared__ double s_red[16];
  __shared__ double s_V[kBlock], s_T[kBlock], s_Y[kBlock];
  __shared__ int s_last;
  const int r = m.r;
  const int par = (iter - 1) & 1;
  // Loads that depend on nothing are requested before the gate is looked at (one round trip instead of four in a row):
  // the partials of <z, r>, its old value, the row pointers of the workgroup's first row block and the thread's own
  // entries of z and delta.  The empty asm keeps the compiler from sinking them behind the early return.
  const GateWords gw = gate_words(ctl);
  double pv = ((int)threadIdx.x < np3) ? p3[threadIdx.x] : 0.0;
  const double zr_old = iter > 0 ? ctl->z_r[par] : 1.0;
  int rp_pre[4] = {0, 0, 0, 0};
  double z_pre = 0, d_pre = 0;
  {
    const int al_ = m.se ? D + 1 : D;
    const int RB_ = ((kBlock / r) / al_) * al_;
    const int j0 = (int)blockIdx.x * RB_, lj_ = threadIdx.x / r;
    if ((int)blockIdx.x < main_grid && j0 < A.nrows) {
      const int j1 = min(A.nrows, j0 + RB_), j = j0 + lj_;
      rp_pre[0] = A.rp[j0];
      rp_pre[1] = A.rp[j1];
      if (lj_ < RB_ && j < j1) {
        rp_pre[2] = A.rp[j];
        rp_pre[3] = A.rp[j + 1];
        const size_t o = (size_t)j * r + (threadIdx.x - lj_ * r);
        z_pre = z[o];
        d_pre = iter > 0 ? d_old[o] : 0.0;
      }
    }
  }
  asm volatile("" ::"v"(pv), "v"(zr_old), "v"(rp_pre[0]), "v"(rp_pre[1]), "v"(rp_pre[2]), "v"(rp_pre[3]), "v"(z_pre),
               "v"(d_pre), "s"(gw.outer), "s"(gw.tcg));
  if (gated(gw, ctl, seq, 2)) return;
  for (int i = threadIdx.x + blockDim.x; i < np3; i += blockDim.x) pv += p3[i];
  const double z_r_new = block_sum(pv, s_red);
  const double beta = iter > 0 ? z_r_new / zr_old : 0.0;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    if (iter == 0) {
      ctl->z_r[0] = z_r_new;
      ctl->d_Pd[0] = z_r_new;
      ctl->e_Pe[0] = 0;
      ctl->e_Pd[0] = 0;
    } else {
      const double alpha = ctl->alpha;
      const double d_Pd = ctl->d_Pd[par], e_Pd = ctl->e_Pd[par];
      ctl->z_r[par ^ 1] = z_r_new;
      ctl->e_Pd[par ^ 1] = beta * (e_Pd + alpha * d_Pd);
      ctl->d_Pd[par ^ 1] = z_r_new + beta * beta * d_Pd;
      ctl->e_Pe[par ^ 1] = ctl->e_Pe_n;
    }
  }
  const double *X = pick(Xb, ctl, 0);
  const double *Sblk = pick(Sb, ctl, 0);
  const int al = m.se ? D + 1 : D;
  const int RB = ((kBlock / r) / al) * al;
  const int nrb = (A.nrows + RB - 1) / RB;
  const int lj = threadIdx.x / r, t = threadIdx.x - lj * r;
  auto dir = [&](size_t o) -> double { return iter > 0 ? fma(beta, d_old[o], -z[o]) : -z[o]; };
  if ((int)blockIdx.x >= main_grid) {  // a slice of a long (Euclidean) row
    const int li = ((int)blockIdx.x - main_grid) / kLongSplit, sl = ((int)blockIdx.x - main_grid) % kLongSplit;
    const int j = A.long_rows[li];
    const int rb0 = A.rp[j], re0 = A.rp[j + 1];
    const int per = (re0 - rb0 + kLongSplit - 1) / kLongSplit;
    const int pb = rb0 + sl * per, pe = min(re0, pb + per);
    const int RBl = kBlock / r;
    double acc = 0;
    if (lj < RBl) {
      for (int p = pb + lj; p < pe; p += 8 * RBl) {
        int c8[8];
        double w8[8], x8[8], y8[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int pp = p + q * RBl;
          const bool ok = pp < pe;
          c8[q] = A.ci[ok ? pp : rb0];
          w8[q] = ok ? A.v[pp] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const size_t o = (size_t)c8[q] * r + t;
          x8[q] = z[o];
          y8[q] = iter > 0 ? d_old[o] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) acc += w8[q] * fma(beta, y8[q], -x8[q]);
      }
    }
    double *s_part = s_v;
    __syncthreads();
    s_part[threadIdx.x] = (lj < RBl) ? acc : 0.0;
    __syncthreads();
    if ((int)threadIdx.x < r) {
      double y = 0;
      for (int q = 0; q < RBl; ++q) y += s_part[q * r + threadIdx.x];
      __hip_atomic_store(A.long_part + ((size_t)li * kLongSplit + sl) * 16 + threadIdx.x, y, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
    }
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0)
      s_last = (__hip_atomic_fetch_add(A.long_cnt + li, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) ==
                kLongSplit - 1);
    __syncthreads();
    if (!s_last) return;
    if (threadIdx.x == 0) __hip_atomic_store(A.long_cnt + li, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (threadIdx.x < 64) {  // r <= 16: the row's r entries sit in the first wave
      double dot = 0;
      if ((int)threadIdx.x < r) {
        double y = 0;
        for (int q = 0; q < kLongSplit; ++q)
          y += __hip_atomic_load(A.long_part + ((size_t)li * kLongSplit + q) * 16 + threadIdx.x, __ATOMIC_RELAXED,
                                 __HIP_MEMORY_SCOPE_AGENT);
        const size_t o = (size_t)j * r + threadIdx.x;
        const double dn = dir(o);
        Hd[o] = y;
        d_new[o] = dn;
        dot = dn * y;
      }
      // ONE slot per long row, written by whichever slice arrives last: a slot per slice would move the row's term
      // around the partial array from run to run, and with it the order of the consumer's sum
      dot = wave_sum(dot);
      if (threadIdx.x == 0) p1[main_grid + li] = dot;
    }
    return;
  }
  const int n_rot_rows = m.se ? A.nrows : m.n * D;  // rows below this bound belong to pose items
  double dacc = 0;
  for (int rb = blockIdx.x; rb < nrb; rb += main_grid) {
    const int j0 = rb * RB;
    const int j1 = min(A.nrows, j0 + RB);
    const int j = j0 + lj;
    const bool active = (lj < RB) && (j < j1);
    const bool first = rb == (int)blockIdx.x;  // requested in the prologue
    const int pbeg = first ? rp_pre[0] : A.rp[j0], pend = first ? rp_pre[1] : A.rp[j1];
    int myb = active ? (first ? rp_pre[2] : A.rp[j]) : 0, mye = active ? (first ? rp_pre[3] : A.rp[j + 1]) : 0;
    const bool is_long = A.n_long > 0 && (mye - myb > kLongRow);  // served by its own workgroups
    if (is_long) mye = myb;
    const size_t o = (size_t)(active ? j : j0) * r + t;
    const double own = !active ? 0.0 : !first ? dir(o) : iter > 0 ? fma(beta, d_pre, -z_pre) : -z_pre;
    const double xo = active ? X[o] : 0.0;
    double acc = 0;
    for (int base = pbeg; base < pend; base += kSpmmTile) {
      const int cnt = min(kSpmmTile, pend - base);
      __syncthreads();
      {
        constexpr int SU = kSpmmTile / kBlock;
        int ci_r[SU];
        double v_r[SU];
        const int last = base + cnt - 1;
#pragma unroll
        for (int u = 0; u < SU; ++u) {
          const int i = min(base + (int)threadIdx.x + u * kBlock, last);
          ci_r[u] = A.ci[i];
          v_r[u] = A.v[i];
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
        double x8[8], y8[8], w8[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const bool ok = p + q < hi;
          const int pp = ok ? p + q : lo;
          w8[q] = ok ? s_v[pp] : 0.0;
          const size_t oc = (size_t)s_ci[pp] * r + t;
          x8[q] = z[oc];
          y8[q] = iter > 0 ? d_old[oc] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) acc += w8[q] * fma(beta, y8[q], -x8[q]);
      }
    }
    // ---- EucHvToHv on the block's own items ----
    // kind: 0 rotation column `a` of pose `it`, 1 unit-sphere column, 2 Euclidean column
    int kind = 2, a = 0, it = 0;
    if (active && j < n_rot_rows) {
      const int q = j / al;
      a = j - q * al;
      it = q;
      kind = (a < D) ? 0 : 2;
    } else if (active && !m.se && j < n_rot_rows + m.l) {
      kind = 1;
      it = j - n_rot_rows;
    }
    __syncthreads();
    s_V[threadIdx.x] = own;
    s_Y[threadIdx.x] = xo;
    __syncthreads();
    double T = acc;
    const int l0 = lj - a;  // local row of the item's first column
    if (kind == 0) {
      double sres = 0;
#pragma unroll
      for (int b = 0; b < D; ++b) sres += s_V[(l0 + b) * r + t] * Sblk[(size_t)it * D * D + b + a * D];
      T = acc - sres;
    } else if (kind == 1) {
      T = acc - own * Sblk[(size_t)m.n * D * D + it];
    }
    s_T[threadIdx.x] = T;
    __syncthreads();
    double hv = T;
    if (kind == 0) {
      // S2 = sym(Y^T T); hv = T - sum_a' Y(t, a') S2[a'][a]
      double sres = 0;
#pragma unroll
      for (int b = 0; b < D; ++b) {
        double pba = 0, pab = 0;  // P[b][a] = sum_t Y(t, b) T(t, a), P[a][b] = sum_t Y(t, a) T(t, b)
        for (int u = 0; u < r; ++u) {
          pba += s_Y[(l0 + b) * r + u] * s_T[(l0 + a) * r + u];
          pab += s_Y[(l0 + a) * r + u] * s_T[(l0 + b) * r + u];
        }
        sres += s_Y[(l0 + b) * r + t] * (0.5 * (pba + pab));
      }
      hv = T - sres;
    } else if (kind == 1) {
      double yt = 0;
      for (int u = 0; u < r; ++u) yt += s_Y[lj * r + u] * s_T[lj * r + u];
      hv = T - xo * yt;
    }
    if (active && !is_long) {
      Hd[o] = hv;
      d_new[o] = own;
      dacc += own * hv;
    }
  }
  const double tot = block_sum(dacc, s_red);
  if (threadIdx.x == 0) p1[blockIdx.x] = tot;
}

int spmm_dir_fix_grid(const ManiDesc &m, int nrows) {
  const int al = m.se ? m.d + 1 : m.d;
  const int RB = ((kBlock / m.r) / al) * al;
  if (RB < al) return 0;
  long nrb = (nrows + RB - 1) / RB;
  if (nrb < 1) nrb = 1;
  if (nrb > kMaxPartials) nrb = kMaxPartials;
  return (int)nrb;
}

int launch_spmm_dir_fix(hipStream_t st, const ManiDesc &m, const CsrDev &A, Buf2 X, Buf2 Sblk, const double *z,
                        const double *d_old, double *d_new, double *Hd, const double *p3, int np3, double *p1,
                        SolverCtl *ctl, int seq, int iter) {
  const int main_grid = spmm_dir_fix_grid(m, A.nrows);
  const int grid = main_grid + A.n_long * kLongSplit;
  if (m.d == 3)
    hipLaunchKernelGGL(k_spmm_dir_fix<3>, dim3(grid), dim3(kBlock), 0, st, m, A, X, Sblk, z, d_old, d_new, Hd, p3, np3,
                       p1, ctl, seq, iter, main_grid);
  else
    hipLaunchKernelGGL(k_spmm_dir_fix<2>, dim3(grid), dim3(kBlock), 0, st, m, A, X, Sblk, z, d_old, d_new, Hd, p3, np3,
                       p1, ctl, seq, iter, main_grid);
  return main_grid + A.n_long;
}

void launch_spmm_dir(hipStream_t st, int r, const CsrDev &A, const double *z, const double *d_old, double *d_new,
                     double *W, const double *p3, int np3, SolverCtl *ctl, int seq, int iter) {
  const int main_grid = spmm_grid(A.nrows, r);
  const int grid = main_grid + A.n_long * kLongSplit;
  hipLaunchKernelGGL(k_spmm_dir, dim3(grid), dim3(kBlock), 0, st, r, A, z, d_old, d_new, W, p3, np3, ctl, seq, iter,
                     main_grid);
}

void launch_spmm(hipStream_t st, int r, const CsrDev &A, Buf2 X, int selX, const double *G, Buf2 Y, int selY,
                 double *partials, Gate g) {
  const int main_grid = spmm_grid(A.nrows, r);
  const int grid = main_grid + A.n_long * kLongSplit;
  if (partials)
    hipLaunchKernelGGL(k_spmm<true>, dim3(grid), dim3(kBlock), 0, st, r, A, X, selX, G, Y, selY, partials, g,
                       main_grid);
  else
    hipLaunchKernelGGL(k_spmm<false>, dim3(grid), dim3(kBlock), 0, st, r, A, X, selX, G, Y, selY, partials, g,
                       main_grid);
}

// ------------------------------------------------------------------------------------------------------
// per-pose register-resident blocks: D columns of RM (>= r) rows, statically indexed
// ------------------------------------------------------------------------------------------------------
template <int D, int RM>
struct Blk {
  double a[D][RM];
};
template <int D, int RM>
__device__ __forceinline__ void ld_blk(const double *__restrict__ p, int r, Blk<D, RM> &B) {
#pragma unroll
  for (int c = 0; c < D; ++c)
#pragma unroll
    for (int t = 0; t < RM; ++t) B.a[c][t] = (t < r) ? p[c * r + t] : 0.0;
}
template <int D, int RM>
__device__ __forceinline__ void st_blk(double *__restrict__ p, int r, const Blk<D, RM> &B) {
#pragma unroll
  for (int c = 0; c < D; ++c)
#pragma unroll
    for (int t = 0; t < RM; ++t)
      if (t < r) p[c * r + t] = B.a[c][t];
}
// S = sym(Y^T E)
template <int D, int RM>
__device__ __forceinline__ void sym_gram(const Blk<D, RM> &Y, const Blk<D, RM> &E, double (&S)[D][D]) {
  double P[D][D];
#pragma unroll
  for (int a = 0; a < D; ++a)
#pragma unroll
    for (int b = 0; b < D; ++b) {
      double s = 0;
#pragma unroll
      for (int t = 0; t < RM; ++t) s += Y.a[a][t] * E.a[b][t];
      P[a][b] = s;
    }
#pragma unroll
  for (int a = 0; a < D; ++a)
#pragma unroll
    for (int b = 0; b < D; ++b) S[a][b] = 0.5 * (P[a][b] + P[b][a]);
}
// V <- V - A S   (A, V: RM x D blocks; S: D x D)
template <int D, int RM>
__device__ __forceinline__ void sub_AS(Blk<D, RM> &V, const Blk<D, RM> &A, const double (&S)[D][D]) {
#pragma unroll
  for (int b = 0; b < D; ++b)
#pragma unroll
    for (int t = 0; t < RM; ++t) {
      double s = 0;
#pragma unroll
      for (int a = 0; a < D; ++a) s += A.a[a][t] * S[a][b];
      V.a[b][t] -= s;
    }
}
template <int D, int RM>
__device__ __forceinline__ double blk_dot(const Blk<D, RM> &A, const Blk<D, RM> &B) {
  double s = 0;
#pragma unroll
  for (int c = 0; c < D; ++c)
#pragma unroll
    for (int t = 0; t < RM; ++t) s += A.a[c][t] * B.a[c][t];
  return s;
}
// thin QR by modified Gram-Schmidt with one re-orthogonalisation pass; returns Q (R has positive diagonal)
template <int D, int RM>
__device__ __forceinline__ void qf_blk(Blk<D, RM> &A) {
#pragma unroll
  for (int j = 0; j < D; ++j) {
#pragma unroll
    for (int pass = 0; pass < 2; ++pass)
#pragma unroll
      for (int c = 0; c < D; ++c)
        if (c < j) {
          double s = 0;
#pragma unroll
          for (int t = 0; t < RM; ++t) s += A.a[c][t] * A.a[j][t];
#pragma unroll
          for (int t = 0; t < RM; ++t) A.a[j][t] -= s * A.a[c][t];
        }
    double nn = 0;
#pragma unroll
    for (int t = 0; t < RM; ++t) nn += A.a[j][t] * A.a[j][t];
    const double inv = 1.0 / sqrt(nn);
#pragma unroll
    for (int t = 0; t < RM; ++t) A.a[j][t] *= inv;
  }
}
// polar factor U V^T by one-sided (Hestenes) Jacobi: rotate column pairs until mutually orthogonal
template <int D, int RM>
__device__ __forceinline__ void polar_blk(Blk<D, RM> &A) {
  double Vm[D][D];
#pragma unroll
  for (int a = 0; a < D; ++a)
#pragma unroll
    for (int b = 0; b < D; ++b) Vm[a][b] = (a == b) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 40; ++sweep) {
    double off = 0;
#pragma unroll
    for (int p = 0; p < D - 1; ++p)
#pragma unroll
      for (int q = p + 1; q < D; ++q) {
        double app = 0, aqq = 0, apq = 0;
#pragma unroll
        for (int t = 0; t < RM; ++t) {
          app += A.a[p][t] * A.a[p][t];
          aqq += A.a[q][t] * A.a[q][t];
          apq += A.a[p][t] * A.a[q][t];
        }
        const double sc = sqrt(app * aqq);
        if (fabs(apq) > 1e-16 * sc && fabs(apq) > 1e-300) {
          off = fmax(off, fabs(apq) / sc);
          const double zeta = (aqq - app) / (2.0 * apq);
          const double tt = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
          const double cs = 1.0 / sqrt(1.0 + tt * tt), sn = cs * tt;
#pragma unroll
          for (int t = 0; t < RM; ++t) {
            const double x = A.a[p][t], y = A.a[q][t];
            A.a[p][t] = cs * x - sn * y;
            A.a[q][t] = sn * x + cs * y;
          }
#pragma unroll
          for (int i = 0; i < D; ++i) {
            const double x = Vm[p][i], y = Vm[q][i];
            Vm[p][i] = cs * x - sn * y;
            Vm[q][i] = sn * x + cs * y;
          }
        }
      }
    if (off < 1e-15) break;
  }
  // columns of A are U Sigma; Vm[j][i] = V(i, j)
#pragma unroll
  for (int j = 0; j < D; ++j) {
    double nn = 0;
#pragma unroll
    for (int t = 0; t < RM; ++t) nn += A.a[j][t] * A.a[j][t];
    const double inv = nn > 0 ? 1.0 / sqrt(nn) : 0.0;
#pragma unroll
    for (int t = 0; t < RM; ++t) A.a[j][t] *= inv;
  }
  Blk<D, RM> O;
#pragma unroll
  for (int c = 0; c < D; ++c)
#pragma unroll
    for (int t = 0; t < RM; ++t) {
      double s = 0;
#pragma unroll
      for (int j = 0; j < D; ++j) s += A.a[j][t] * Vm[j][c];
      O.a[c][t] = s;
    }
  A = O;
}

int pose_grid(const ManiDesc &m) {
  const long items = (long)m.n + m.l + m.num_euc();
  long g = (items + kBlock - 1) / kBlock;
  if (g < 1) g = 1;
  if (g > kMaxPartials) g = kMaxPartials;
  return (int)g;
}

// ---- Riemannian gradient: RG = Proj_X(EG), S_i = sym(Y_i^T EG_i) ----------------------------------------
template <int D, int RM>
__global__ __launch_bounds__(kBlock) void k_rgrad(ManiDesc m, Buf2 Xb, Buf2 EGb, Buf2 RGb, Buf2 Sb, int sel,
                                                  double *__restrict__ partials, Gate g) {
  if (gated(g.ctl, g.seq, g.gate)) return;
  __shared__ double s_red[16];
  const double *X = pick(Xb, g.ctl, sel);
  const double *EG = pick(EGb, g.ctl, sel);
  double *RG = pick(RGb, g.ctl, sel);
  double *Sblk = pick(Sb, g.ctl, sel);
  const int r = m.r;
  const long items = (long)m.n + m.l + m.num_euc();
  double acc = 0;
  for (long it = (long)blockIdx.x * kBlock + threadIdx.x; it < items; it += (long)gridDim.x * kBlock) {
    if (it < m.n) {
      const size_t o = (size_t)m.rot_col((int)it) * r;
      Blk<D, RM> Y, E;
      ld_blk<D, RM>(X + o, r, Y);
      ld_blk<D, RM>(EG + o, r, E);
      double S[D][D];
      sym_gram<D, RM>(Y, E, S);
      if (Sblk)
#pragma unroll
        for (int a = 0; a < D; ++a)
#pragma unroll
          for (int b = 0; b < D; ++b) Sblk[(size_t)it * D * D + a + b * D] = S[a][b];
      sub_AS<D, RM>(E, Y, S);
      acc += blk_dot<D, RM>(E, E);
      if (RG) st_blk<D, RM>(RG + o, r, E);
    } else if (it < m.n + m.l) {
      const int i = (int)(it - m.n);
      const size_t o = (size_t)m.sphere_col(i) * r;
      double s = 0;
      for (int t = 0; t < r; ++t) s += X[o + t] * EG[o + t];
      if (Sblk) Sblk[(size_t)m.n * D * D + i] = s;
      for (int t = 0; t < r; ++t) {
        const double v = EG[o + t] - X[o + t] * s;
        acc += v * v;
        if (RG) RG[o + t] = v;
      }
    } else {
      const size_t o = (size_t)m.euc_col((int)(it - m.n - m.l)) * r;
      for (int t = 0; t < r; ++t) {
        const double v = EG[o + t];
        acc += v * v;
        if (RG) RG[o + t] = v;
      }
    }
  }
  const double tot = block_sum(acc, s_red);
  if (threadIdx.x == 0 && partials) partials[blockIdx.x] = tot;
}

// ---- out = Proj_X(V), partial <out, R>; optional tCG residual stopping rule in the prologue ---------------
template <int D, int RM>
__global__ __launch_bounds__(kBlock) void k_tangent(ManiDesc m, Buf2 Xb, const double *__restrict__ V,
                                                    double *__restrict__ out, const double *__restrict__ R,
                                                    double *__restrict__ partials, const double *__restrict__ p2,
                                                    int np2, SolverCtl *ctl, HostFlags *hf, int seq, int gate,
                                                    int iter, SpFold sf) {
  __shared__ double s_red[16];
  __shared__ double s_x2[64 * 16];
  // Loads that depend on nothing are requested before the gate is looked at -- the stopping rule's partials and |r0|,
  // the hub's index, the places of the thread's first columns in the replay's images: with the gate they are one
  // memory round trip where they were five in a row (k_tangent on tiers.pyfg: 9.9 us for 0.6 MB vectors, all of it
  // dependent round trips).  The empty asm keeps the compiler from sinking them behind the early return.
  const int r = m.r;
  const long items = (long)m.n + m.l + m.num_euc();
  const bool folded = sf.y != nullptr;
  const GateWords gw = gate_words(ctl);
  double pv = (p2 && (int)threadIdx.x < np2) ? p2[threadIdx.x] : 0.0;
  const double n0 = p2 ? ctl->norm_r0 : 0.0;
  const bool hub0 = folded && sf.h > 0 && (int)threadIdx.x < sf.h * r;
  double x2_0 = hub0 ? sf.hub_x2[threadIdx.x] : 0.0;
  const long it0 = (long)blockIdx.x * kBlock + threadIdx.x;
  int jp_pre[D], op_pre[D];
#pragma unroll
  for (int a = 0; a < D; ++a) jp_pre[a] = op_pre[a] = 0;
  if (folded && it0 < items) {
    const size_t c0 = it0 < m.n ? (size_t)m.rot_col((int)it0)
                                : it0 < m.n + m.l ? (size_t)m.sphere_col((int)(it0 - m.n))
                                                  : (size_t)m.euc_col((int)(it0 - m.n - m.l));
    const int nc0 = it0 < m.n ? D : 1;
#pragma unroll
    for (int a = 0; a < D; ++a)
      if (a < nc0) {
        jp_pre[a] = sf.in_pos[c0 + a];
        op_pre[a] = sf.out_pos[c0 + a];
      }
  }
#pragma unroll
  for (int a = 0; a < D; ++a) asm volatile("" ::"v"(jp_pre[a]), "v"(op_pre[a]));
  asm volatile("" ::"v"(pv), "v"(n0), "v"(x2_0), "s"(gw.outer), "s"(gw.tcg));
  if (gated(gw, ctl, seq, gate)) return;
  if (p2) {
    // ROPTLIB tCG_TR stopping rule (theta = 1, kappa = 0.1): |r| <= |r0| min(|r0|^theta, kappa)
    for (int i = threadIdx.x + blockDim.x; i < np2; i += blockDim.x) pv += p2[i];
    const double nr = sqrt(block_sum(pv, s_red));
    const double kappa = 0.1;
    const double tempnum = n0;  // pow(n0, theta = 1)
    if (nr <= n0 * fmin(tempnum, kappa)) {
      if (blockIdx.x == 0 && threadIdx.x == 0) {
        ctl->tcg_status = (kappa < tempnum) ? 2 : 3;
        ctl->tcg_iters = iter + 1;
        ctl->inner_total += iter + 1;
        ctl->tcg_done_stamp = seq;
        host_store(&hf->tcg_done_seq, seq);
      }
      return;
    }
  }
  const double *X = pick(Xb, ctl, 0);
  // sparse preconditioner folded in (generic layout): V(col, t) is read from where the level replay left it, with the
  // hub correction of k_sp_permute_out_hub; x2 = Sinv (R(hub) - U^T r1) comes ready from the replay's second launch
  // (round 4 rebuilt it here in every workgroup: two dependent rounds of loads and two barriers in front of the items)
  if (folded && sf.h > 0) {
    if (hub0) s_x2[threadIdx.x] = x2_0;
    for (int e = threadIdx.x + kBlock; e < sf.h * r; e += kBlock) s_x2[e] = sf.hub_x2[e];
    __syncthreads();
  }
  // NC consecutive columns starting at col0: positions first (the thread's first item: requested in the prologue), then
  // every value, straight line (a hub column -- rare -- is patched afterwards)
  auto vcols = [&](size_t col0, auto nc_tag, double (*vv)[RM], bool first) {
    constexpr int NC = decltype(nc_tag)::value;
    int jp[NC], op[NC];
#pragma unroll
    for (int a = 0; a < NC; ++a) {
      jp[a] = first ? jp_pre[a] : sf.in_pos[col0 + a];
      op[a] = first ? op_pre[a] : sf.out_pos[col0 + a];
    }
#pragma unroll
    for (int a = 0; a < NC; ++a)
#pragma unroll
      for (int t = 0; t < RM; ++t) vv[a][t] = (t < r) ? sf.y[(size_t)max(op[a], 0) * r + t] : 0.0;
    for (int q = 0; q < sf.h; ++q) {
      double u[NC];
#pragma unroll
      for (int a = 0; a < NC; ++a) u[a] = sf.hub_U[(size_t)max(jp[a], 0) * sf.h + q];
#pragma unroll
      for (int a = 0; a < NC; ++a)
#pragma unroll
        for (int t = 0; t < RM; ++t)
          if (t < r) vv[a][t] -= u[a] * s_x2[q * r + t];
    }
#pragma unroll
    for (int a = 0; a < NC; ++a)
      if (jp[a] < 0) {
#pragma unroll
        for (int t = 0; t < RM; ++t) vv[a][t] = 0.0;
        for (int q = 0; q < sf.h; ++q)
          if ((size_t)sf.hub_idx[q] == col0 + a)
#pragma unroll
            for (int t = 0; t < RM; ++t)
              if (t < r) vv[a][t] = s_x2[q * r + t];
      }
  };
  double acc = 0;
  for (long it = (long)blockIdx.x * kBlock + threadIdx.x; it < items; it += (long)gridDim.x * kBlock) {
    if (it < m.n) {
      const size_t o = (size_t)m.rot_col((int)it) * r;
      Blk<D, RM> Y, W;
      ld_blk<D, RM>(X + o, r, Y);
      if (folded) {
        vcols((size_t)m.rot_col((int)it), std::integral_constant<int, D>{}, W.a, it == it0);
      } else {
        ld_blk<D, RM>(V + o, r, W);
      }
      double S[D][D];
      sym_gram<D, RM>(Y, W, S);
      sub_AS<D, RM>(W, Y, S);
      if (R) {
        Blk<D, RM> Rr;
        ld_blk<D, RM>(R + o, r, Rr);
        acc += blk_dot<D, RM>(W, Rr);
      }
      st_blk<D, RM>(out + o, r, W);
    } else if (it < m.n + m.l) {
      const size_t col = (size_t)m.sphere_col((int)(it - m.n));
      const size_t o = col * r;
      double vv1[1][RM];
      if (folded) {
        vcols(col, std::integral_constant<int, 1>{}, vv1, it == it0);
      } else {
#pragma unroll
        for (int t = 0; t < RM; ++t) vv1[0][t] = (t < r) ? V[o + t] : 0.0;
      }
      const double *vv = vv1[0];
      double s = 0;
#pragma unroll
      for (int t = 0; t < RM; ++t)
        if (t < r) s += X[o + t] * vv[t];
#pragma unroll
      for (int t = 0; t < RM; ++t)
        if (t < r) {
          const double v = vv[t] - X[o + t] * s;
          if (R) acc += v * R[o + t];
          out[o + t] = v;
        }
    } else {
      const size_t col = (size_t)m.euc_col((int)(it - m.n - m.l));
      const size_t o = col * r;
      double ve[1][RM];
      if (folded) {
        vcols(col, std::integral_constant<int, 1>{}, ve, it == it0);
      } else {
#pragma unroll
        for (int t = 0; t < RM; ++t) ve[0][t] = (t < r) ? V[o + t] : 0.0;
      }
#pragma unroll
      for (int t = 0; t < RM; ++t)
        if (t < r) {
          const double v = ve[0][t];
          if (R) acc += v * R[o + t];
          out[o + t] = v;
        }
    }
  }
  if (partials) {
    const double tot = block_sum(acc, s_red);
    if (threadIdx.x == 0) partials[blockIdx.x] = tot;
  }
  // the last kernel of a tCG iteration when the direction update is folded into the next SpMM: paces the host
  if (hf && gate == 2 && blockIdx.x == 0 && threadIdx.x == 0) host_store(&hf->last_seq_done, seq);
}

// ---- HV = Proj_X(W - V S), partial <V, HV>  (ROPTLIB EucHvToHv for the Euclidean metric) -----------------
template <int D, int RM>
__global__ __launch_bounds__(kBlock) void k_hessfix(ManiDesc m, Buf2 Xb, Buf2 Sb, const double *__restrict__ V,
                                                    const double *__restrict__ W, double *__restrict__ HV,
                                                    double *__restrict__ partials, Gate g) {
  if (gated(g.ctl, g.seq, g.gate)) return;
  __shared__ double s_red[16];
  const double *X = pick(Xb, g.ctl, 0);
  const double *Sblk = pick(Sb, g.ctl, 0);
  const int r = m.r;
  const long items = (long)m.n + m.l + m.num_euc();
  double acc = 0;
  for (long it = (long)blockIdx.x * kBlock + threadIdx.x; it < items; it += (long)gridDim.x * kBlock) {
    if (it < m.n) {
      const size_t o = (size_t)m.rot_col((int)it) * r;
      Blk<D, RM> Y, Vb, T;
      ld_blk<D, RM>(X + o, r, Y);
      ld_blk<D, RM>(V + o, r, Vb);
      ld_blk<D, RM>(W + o, r, T);
      double S[D][D];
#pragma unroll
      for (int a = 0; a < D; ++a)
#pragma unroll
        for (int b = 0; b < D; ++b) S[a][b] = Sblk[(size_t)it * D * D + a + b * D];
      sub_AS<D, RM>(T, Vb, S);
      double S2[D][D];
      sym_gram<D, RM>(Y, T, S2);
      sub_AS<D, RM>(T, Y, S2);
      acc += blk_dot<D, RM>(Vb, T);
      st_blk<D, RM>(HV + o, r, T);
    } else if (it < m.n + m.l) {
      const int i = (int)(it - m.n);
      const size_t o = (size_t)m.sphere_col(i) * r;
      const double s = Sblk[(size_t)m.n * D * D + i];
      double yt = 0;
      for (int t = 0; t < r; ++t) yt += X[o + t] * (W[o + t] - V[o + t] * s);
      for (int t = 0; t < r; ++t) {
        const double v = (W[o + t] - V[o + t] * s) - X[o + t] * yt;
        acc += V[o + t] * v;
        HV[o + t] = v;
      }
    } else {
      const size_t o = (size_t)m.euc_col((int)(it - m.n - m.l)) * r;
      for (int t = 0; t < r; ++t) {
        const double v = W[o + t];
        acc += V[o + t] * v;
        HV[o + t] = v;
      }
    }
  }
  const double tot = block_sum(acc, s_red);
  if (threadIdx.x == 0 && partials) partials[blockIdx.x] = tot;
}

// ---- out = Retr_X(alpha V): QF on Stiefel blocks, normalise spheres, add on Euclidean columns --------------
template <int D, int RM>
__global__ __launch_bounds__(kBlock) void k_retract(ManiDesc m, Buf2 Xb, const double *__restrict__ V,
                                                    double alpha, Buf2 Ob, int selOut, Buf2 gradb,
                                                    const double *__restrict__ HV, double *__restrict__ partials,
                                                    Gate g) {
  if (gated(g.ctl, g.seq, g.gate)) return;
  __shared__ double s_red[16];
  const double *X = pick(Xb, g.ctl, 0);
  double *out = pick(Ob, g.ctl, selOut);
  const double *grad = partials ? pick(gradb, g.ctl, 0) : nullptr;
  const int r = m.r;
  const long items = (long)m.n + m.l + m.num_euc();
  double a0 = 0, a1 = 0;
  for (long it = (long)blockIdx.x * kBlock + threadIdx.x; it < items; it += (long)gridDim.x * kBlock) {
    size_t o;
    int ncol;
    if (it < m.n) {
      o = (size_t)m.rot_col((int)it) * r;
      ncol = D;
      Blk<D, RM> Y, Vb;
      ld_blk<D, RM>(X + o, r, Y);
      ld_blk<D, RM>(V + o, r, Vb);
#pragma unroll
      for (int c = 0; c < D; ++c)
#pragma unroll
        for (int t = 0; t < RM; ++t) Y.a[c][t] += alpha * Vb.a[c][t];
      qf_blk<D, RM>(Y);
      st_blk<D, RM>(out + o, r, Y);
    } else if (it < m.n + m.l) {
      o = (size_t)m.sphere_col((int)(it - m.n)) * r;
      ncol = 1;
      double nn = 0;
      for (int t = 0; t < r; ++t) {
        const double w = X[o + t] + alpha * V[o + t];
        nn += w * w;
      }
      const double inv = 1.0 / sqrt(nn);
      for (int t = 0; t < r; ++t) out[o + t] = (X[o + t] + alpha * V[o + t]) * inv;
    } else {
      o = (size_t)m.euc_col((int)(it - m.n - m.l)) * r;
      ncol = 1;
      for (int t = 0; t < r; ++t) out[o + t] = X[o + t] + alpha * V[o + t];
    }
    if (partials)
      for (int e = 0; e < ncol * r; ++e) {
        const double v = V[o + e];
        a0 += v * grad[o + e];
        a1 += v * HV[o + e];
      }
  }
  if (partials) {
    const double t0 = block_sum(a0, s_red);
    const double t1 = block_sum(a1, s_red);
    if (threadIdx.x == 0) {
      partials[2 * blockIdx.x] = t0;
      partials[2 * blockIdx.x + 1] = t1;
    }
  }
}

// ---- out = P_M(c0 A + c1 B + c2 C): polar factor per Stiefel block, normalised spheres --------------------
template <int D, int RM>
__global__ __launch_bounds__(kBlock) void k_polar(ManiDesc m, double c0, const double *__restrict__ A, double c1,
                                                  const double *__restrict__ B, double c2,
                                                  const double *__restrict__ C, double *__restrict__ out) {
  const int r = m.r;
  const long items = (long)m.n + m.l + m.num_euc();
  for (long it = (long)blockIdx.x * kBlock + threadIdx.x; it < items; it += (long)gridDim.x * kBlock) {
    if (it < m.n) {
      const size_t o = (size_t)m.rot_col((int)it) * r;
      Blk<D, RM> M, T;
      ld_blk<D, RM>(A + o, r, M);
#pragma unroll
      for (int c = 0; c < D; ++c)
#pragma unroll
        for (int t = 0; t < RM; ++t) M.a[c][t] *= c0;
      if (B) {
        ld_blk<D, RM>(B + o, r, T);
#pragma unroll
        for (int c = 0; c < D; ++c)
#pragma unroll
          for (int t = 0; t < RM; ++t) M.a[c][t] += c1 * T.a[c][t];
      }
      if (C) {
        ld_blk<D, RM>(C + o, r, T);
#pragma unroll
        for (int c = 0; c < D; ++c)
#pragma unroll
          for (int t = 0; t < RM; ++t) M.a[c][t] += c2 * T.a[c][t];
      }
      polar_blk<D, RM>(M);
      st_blk<D, RM>(out + o, r, M);
    } else {
      const bool sph = it < m.n + m.l;
      const size_t o = (size_t)(sph ? m.sphere_col((int)(it - m.n)) : m.euc_col((int)(it - m.n - m.l))) * r;
      double nn = 0;
      for (int t = 0; t < r; ++t) {
        double w = c0 * A[o + t];
        if (B) w += c1 * B[o + t];
        if (C) w += c2 * C[o + t];
        nn += w * w;
      }
      const double inv = sph ? 1.0 / sqrt(nn) : 1.0;
      for (int t = 0; t < r; ++t) {
        double w = c0 * A[o + t];
        if (B) w += c1 * B[o + t];
        if (C) w += c2 * C[o + t];
        out[o + t] = w * inv;
      }
    }
  }
}

// ---- Lambda blocks of the dual certificate -----------------------------------------------------------------
template <int D, int RM>
__global__ __launch_bounds__(kBlock) void k_lambda(ManiDesc m, const double *__restrict__ X,
                                                   const double *__restrict__ XQ, double *__restrict__ L) {
  const int r = m.r;
  const long items = (long)m.n + m.l;
  for (long it = (long)blockIdx.x * kBlock + threadIdx.x; it < items; it += (long)gridDim.x * kBlock) {
    if (it < m.n) {
      const size_t o = (size_t)m.rot_col((int)it) * r;
      Blk<D, RM> Y, E;
      ld_blk<D, RM>(X + o, r, Y);
      ld_blk<D, RM>(XQ + o, r, E);
      double S[D][D];
      sym_gram<D, RM>(E, Y, S);
#pragma unroll
      for (int a = 0; a < D; ++a)
#pragma unroll
        for (int b = 0; b < D; ++b) L[(size_t)it * D * D + a + b * D] = S[a][b];
    } else {
      const int i = (int)(it - m.n);
      const size_t o = (size_t)m.sphere_col(i) * r;
      double s = 0;
      for (int t = 0; t < r; ++t) s += X[o + t] * XQ[o + t];
      L[(size_t)m.n * D * D + i] = s;
    }
  }
}

// ---- RBCD++ Nesterov bookkeeping on a range of poses (ref src/Agent.cpp:545-551, 1158-1214) ---------------
//  mode 0 (agent not selected): XPrev = X; Y = P((1-a) X + a V); X = Y; V = P(V + g (X - Y));
//                               on restart: X = XPrev; V = X; Y = X
//  mode 1 (selected, before the local solve): XPrev = X; Y = P((1-a) X + a V); Yloc = Y
//  mode 2 (selected, after the local solve):  X = Xloc; V = P(V + g (X - Y))
//  mode 3 (selected, after the restart solve): X = Xloc; V = X; Y = X
// Global arrays (X, V, Y, XPrev) are offset to the range's first column; Yloc / Xloc are agent-local buffers.
struct NesterovArgs {
  int mode, restart, skip_lo, skip_hi;
  double alpha, gamma;
  double *X, *V, *Y, *XPrev, *Yloc;
  const double *Xloc;
};
template <int D, int RM>
__global__ __launch_bounds__(kBlock) void k_nesterov(ManiDesc m, NesterovArgs a) {
  const int r = m.r;
  const long items = (long)m.n + m.l + m.num_euc();
  for (long it = (long)blockIdx.x * kBlock + threadIdx.x; it < items; it += (long)gridDim.x * kBlock) {
    int kind;  // 0 Stiefel, 1 sphere, 2 Euclidean
    size_t o;
    long pose = -1;
    if (it < m.n) {
      kind = 0;
      o = (size_t)m.rot_col((int)it) * r;
      pose = it;
    } else if (it < m.n + m.l) {
      kind = 1;
      o = (size_t)m.sphere_col((int)(it - m.n)) * r;
    } else {
      kind = 2;
      const int e = (int)(it - m.n - m.l);
      o = (size_t)m.euc_col(e) * r;
      if (m.se) pose = e;
    }
    if (pose >= a.skip_lo && pose < a.skip_hi) continue;
    if (kind == 0) {
      Blk<D, RM> x, v, y;
      if (a.mode <= 1) {
        ld_blk<D, RM>(a.X + o, r, x);
        ld_blk<D, RM>(a.V + o, r, v);
        st_blk<D, RM>(a.XPrev + o, r, x);
#pragma unroll
        for (int c = 0; c < D; ++c)
#pragma unroll
          for (int t = 0; t < RM; ++t) y.a[c][t] = (1.0 - a.alpha) * x.a[c][t] + a.alpha * v.a[c][t];
        polar_blk<D, RM>(y);
        if (a.mode == 1) {
          st_blk<D, RM>(a.Y + o, r, y);
          st_blk<D, RM>(a.Yloc + o, r, y);
        } else if (a.restart) {
          st_blk<D, RM>(a.X + o, r, x);
          st_blk<D, RM>(a.V + o, r, x);
          st_blk<D, RM>(a.Y + o, r, x);
        } else {
          polar_blk<D, RM>(v);  // V + g (X - Y) with X == Y
          st_blk<D, RM>(a.Y + o, r, y);
          st_blk<D, RM>(a.X + o, r, y);
          st_blk<D, RM>(a.V + o, r, v);
        }
      } else {
        ld_blk<D, RM>(a.Xloc + o, r, x);
        st_blk<D, RM>(a.X + o, r, x);
        if (a.mode == 2) {
          ld_blk<D, RM>(a.V + o, r, v);
          ld_blk<D, RM>(a.Y + o, r, y);
#pragma unroll
          for (int c = 0; c < D; ++c)
#pragma unroll
            for (int t = 0; t < RM; ++t) v.a[c][t] += a.gamma * (x.a[c][t] - y.a[c][t]);
          polar_blk<D, RM>(v);
          st_blk<D, RM>(a.V + o, r, v);
        } else {
          st_blk<D, RM>(a.V + o, r, x);
          st_blk<D, RM>(a.Y + o, r, x);
        }
      }
    } else {
      // single column: sphere (normalise) or Euclidean (identity)
      const bool sph = (kind == 1);
      if (a.mode <= 1) {
        double nn = 0;
        for (int t = 0; t < r; ++t) {
          const double w = (1.0 - a.alpha) * a.X[o + t] + a.alpha * a.V[o + t];
          nn += w * w;
        }
        const double inv = sph ? 1.0 / sqrt(nn) : 1.0;
        double vn = 0;
        for (int t = 0; t < r; ++t) vn += a.V[o + t] * a.V[o + t];
        const double vinv = sph ? 1.0 / sqrt(vn) : 1.0;
        for (int t = 0; t < r; ++t) {
          const double x = a.X[o + t], v = a.V[o + t];
          const double y = ((1.0 - a.alpha) * x + a.alpha * v) * inv;
          a.XPrev[o + t] = x;
          if (a.mode == 1) {
            a.Y[o + t] = y;
            a.Yloc[o + t] = y;
          } else if (a.restart) {
            a.V[o + t] = x;
            a.Y[o + t] = x;
          } else {
            a.Y[o + t] = y;
            a.X[o + t] = y;
            a.V[o + t] = v * vinv;
          }
        }
      } else if (a.mode == 2) {
        double nn = 0;
        for (int t = 0; t < r; ++t) {
          const double w = a.V[o + t] + a.gamma * (a.Xloc[o + t] - a.Y[o + t]);
          nn += w * w;
        }
        const double inv = sph ? 1.0 / sqrt(nn) : 1.0;
        for (int t = 0; t < r; ++t) {
          const double x = a.Xloc[o + t];
          const double w = a.V[o + t] + a.gamma * (x - a.Y[o + t]);
          a.X[o + t] = x;
          a.V[o + t] = w * inv;
        }
      } else {
        for (int t = 0; t < r; ++t) {
          const double x = a.Xloc[o + t];
          a.X[o + t] = x;
          a.V[o + t] = x;
          a.Y[o + t] = x;
        }
      }
    }
  }
}

#define DCORA_DISPATCH_POSE(KERNEL, m, grid, st, ...)                                              \
  do {                                                                                             \
    const int rm_ = (m).r <= 4 ? 4 : ((m).r <= 8 ? 8 : 16);                                        \
    if ((m).d == 3) {                                                                              \
      if (rm_ == 4) hipLaunchKernelGGL((KERNEL<3, 4>), dim3(grid), dim3(kBlock), 0, st, __VA_ARGS__);       \
      else if (rm_ == 8) hipLaunchKernelGGL((KERNEL<3, 8>), dim3(grid), dim3(kBlock), 0, st, __VA_ARGS__);  \
      else hipLaunchKernelGGL((KERNEL<3, 16>), dim3(grid), dim3(kBlock), 0, st, __VA_ARGS__);               \
    } else {                                                                                       \
      if (rm_ == 4) hipLaunchKernelGGL((KERNEL<2, 4>), dim3(grid), dim3(kBlock), 0, st, __VA_ARGS__);       \
      else if (rm_ == 8) hipLaunchKernelGGL((KERNEL<2, 8>), dim3(grid), dim3(kBlock), 0, st, __VA_ARGS__);  \
      else hipLaunchKernelGGL((KERNEL<2, 16>), dim3(grid), dim3(kBlock), 0, st, __VA_ARGS__);               \
    }                                                                                              \
  } while (0)

void launch_rgrad(hipStream_t st, const ManiDesc &m, Buf2 X, Buf2 EG, Buf2 RG, Buf2 Sblk, int sel,
                  double *partials, Gate g) {
  const int grid = pose_grid(m);
  DCORA_DISPATCH_POSE(k_rgrad, m, grid, st, m, X, EG, RG, Sblk, sel, partials, g);
}
void launch_tangent(hipStream_t st, const ManiDesc &m, Buf2 X, const double *V, double *out, const double *R,
                    double *partials, const double *p2, int np2, SolverCtl *ctl, HostFlags *hf, int seq,
                    int gate, int iter, SpFold sf) {
  const int grid = pose_grid(m);
  DCORA_DISPATCH_POSE(k_tangent, m, grid, st, m, X, V, out, R, partials, p2, np2, ctl, hf, seq, gate, iter, sf);
}
void launch_hessfix(hipStream_t st, const ManiDesc &m, Buf2 X, Buf2 Sblk, const double *V, const double *W,
                    double *HV, double *partials, Gate g) {
  const int grid = pose_grid(m);
  DCORA_DISPATCH_POSE(k_hessfix, m, grid, st, m, X, Sblk, V, W, HV, partials, g);
}
void launch_retract(hipStream_t st, const ManiDesc &m, Buf2 X, const double *V, double alpha, Buf2 out,
                    int selOut, Buf2 grad, const double *HV, double *partials, Gate g) {
  const int grid = pose_grid(m);
  DCORA_DISPATCH_POSE(k_retract, m, grid, st, m, X, V, alpha, out, selOut, grad, HV, partials, g);
}
void launch_polar(hipStream_t st, const ManiDesc &m, double c0, const double *A, double c1, const double *B,
                  double c2, const double *C, double *out) {
  const int grid = pose_grid(m);
  DCORA_DISPATCH_POSE(k_polar, m, grid, st, m, c0, A, c1, B, c2, C, out);
}
void launch_nesterov(hipStream_t st, const ManiDesc &m, int mode, int restart, int skip_lo, int skip_hi, double alpha,
                     double gamma, double *X, double *V, double *Y, double *XPrev, double *Yloc,
                     const double *Xloc) {
  NesterovArgs a{mode, restart, skip_lo, skip_hi, alpha, gamma, X, V, Y, XPrev, Yloc, Xloc};
  const int grid = pose_grid(m);
  DCORA_DISPATCH_POSE(k_nesterov, m, grid, st, m, a);
}
void launch_lambda_blocks(hipStream_t st, const ManiDesc &m, const double *X, const double *XQ, double *Lblk) {
  const int grid = pose_grid(m);
  DCORA_DISPATCH_POSE(k_lambda, m, grid, st, m, X, XQ, Lblk);
}

// ------------------------------------------------------------------------------------------------------
// Dense preconditioner apply  Z = R * Minv  (Minv symmetric => row j of Minv is column j).
// One wave streams RW complete rows of Minv with 16-byte loads; lane l owns columns {2l, 2l+1} + 128 i.
// ------------------------------------------------------------------------------------------------------
template <int RM, int RW>
__global__ __launch_bounds__(kBlock) void k_dense_apply(int r, int k, int ldm, const double *__restrict__ Minv,
                                                        Buf2 Rb, double *__restrict__ Z,
                                                        const double *__restrict__ p2, int np2, Gate g) {
  if (gated(g.ctl, g.seq, g.gate)) return;
  __shared__ double s_red[16];
  if (p2) {
    // residual already below the tCG stopping threshold: the next kernel records the termination
    const double nr = sqrt(sum_partials(p2, np2, 1, 0, s_red));
    const double n0 = g.ctl->norm_r0;
    if (nr <= n0 * fmin(n0, 0.1)) return;
  }
  const double *__restrict__ R = pick(Rb, g.ctl, 0);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int j0 = (blockIdx.x * (kBlock / 64) + wave) * RW;
  if (j0 >= k) return;
  double acc[RW][RM];
#pragma unroll
  for (int a = 0; a < RW; ++a)
#pragma unroll
    for (int t = 0; t < RM; ++t) acc[a][t] = 0;
  const double *rows[RW];
#pragma unroll
  for (int a = 0; a < RW; ++a) rows[a] = Minv + (size_t)min(j0 + a, k - 1) * ldm;
  for (int c = 2 * lane; c < k; c += 128) {
    const bool two = (c + 1 < k);
    double m0[RW], m1[RW];
#pragma unroll
    for (int a = 0; a < RW; ++a) {
      const double2 mm = *reinterpret_cast<const double2 *>(rows[a] + c);  // ldm is padded: always in bounds
      m0[a] = mm.x;
      m1[a] = two ? mm.y : 0.0;
    }
    const double *rc = R + (size_t)c * r;
#pragma unroll
    for (int t = 0; t < RM; ++t)
      if (t < r) {
        const double x0 = rc[t];
        const double x1 = two ? rc[r + t] : 0.0;
#pragma unroll
        for (int a = 0; a < RW; ++a) acc[a][t] += x0 * m0[a] + x1 * m1[a];
      }
  }
#pragma unroll
  for (int a = 0; a < RW; ++a)
#pragma unroll
    for (int t = 0; t < RM; ++t)
      if (t < r) {
        const double s = wave_sum(acc[a][t]);
        if (lane == 0 && j0 + a < k) Z[(size_t)(j0 + a) * r + t] = s;
      }
}

void launch_dense_apply(hipStream_t st, int r, int k, int ldm, const double *Minv, Buf2 R, double *Z,
                        const double *p2, int np2, Gate g) {
  constexpr int RW = 2;
  const int rows_per_block = (kBlock / 64) * RW;
  const int grid = (k + rows_per_block - 1) / rows_per_block;
  if (r <= 4)
    hipLaunchKernelGGL((k_dense_apply<4, RW>), dim3(grid), dim3(kBlock), 0, st, r, k, ldm, Minv, R, Z, p2, np2, g);
  else if (r <= 8)
    hipLaunchKernelGGL((k_dense_apply<8, RW>), dim3(grid), dim3(kBlock), 0, st, r, k, ldm, Minv, R, Z, p2, np2, g);
  else
    hipLaunchKernelGGL((k_dense_apply<16, RW>), dim3(grid), dim3(kBlock), 0, st, r, k, ldm, Minv, R, Z, p2, np2, g);
}

// ------------------------------------------------------------------------------------------------------
// Trust-region / truncated-CG scalar logic (ROPTLIB RTRNewton + SolversTR::tCG_TR restated).
// Every block recomputes the same scalars from the same partials; block 0 publishes them.
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_rtr_init(const double *pA, int npA, const double *pB, int npB,
                                                     SolverCtl *ctl, HostFlags *hf, int seq, CtlInit ci,
                                                     unsigned *tcg_sync, int nsync) {
  __shared__ double s_red[16];
  for (int i = threadIdx.x; i < nsync; i += kBlock) tcg_sync[i] = 0u;
  if (ci.enable && threadIdx.x == 0) {  // start-of-solve control block (saves the separate k_ctl_init launch)
    SolverCtl *c = ctl;
    c->f2 = c->rho = 0;
    c->Delta = ci.Delta;
    c->maxDelta = ci.maxDelta;
    c->tol = ci.tol;
    c->cur = 0;
    c->outer_it = 0;
    c->max_outer = ci.max_outer;
    c->accepted = 0;
    c->last_accepted = 0;
    c->stop_on_accept = ci.stop_on_accept;
    c->outer_done_stamp = INT_MAX;
    c->alpha = c->e_Pe_n = c->norm_r0 = 0;
    c->tcg_done_stamp = INT_MAX;
    c->tcg_status = 4;
    c->tcg_iters = 0;
    c->inner_total = 0;
    c->max_inner = ci.max_inner;
  }
  const double *const ps[3] = {pA, pA, pB};
  const int nps[3] = {npA, npA, npB}, sts[3] = {2, 2, 1}, offs[3] = {0, 1, 0};
  double sums[3];
  sum_partials_n<3>(ps, nps, sts, offs, s_red, sums);
  const double fq = sums[0], fg = sums[1], g2 = sums[2];
  if (threadIdx.x == 0) {
    ctl->f1 = 0.5 * fq + fg;
    ctl->ngf = sqrt(g2);
    ctl->fInit = ctl->f1;
    ctl->gradNormInit = ctl->ngf;
    if (ctl->ngf < ctl->tol || ctl->max_outer <= 0) {  // ref src/QuadraticOptimizer.cpp:54-55
      ctl->outer_done_stamp = seq;
      host_store(&hf->outer_done_seq, seq);
    }
    host_store(&hf->last_seq_done, seq);
  }
}

// start of a tCG run: res = grad (accepted iterate), eta = H eta = 0, termination stamps re-armed
__global__ __launch_bounds__(kBlock) void k_tcg_begin(long nelem, Buf2 gradb, double *__restrict__ eta,
                                                      double *__restrict__ Heta, double *__restrict__ res,
                                                      SolverCtl *ctl, int seq) {
  if (gated(ctl, seq, 1)) return;
  const double *grad = pick(gradb, ctl, 0);
  for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < nelem; i += (long)gridDim.x * kBlock) {
    eta[i] = 0;
    Heta[i] = 0;
    res[i] = grad[i];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    ctl->norm_r0 = ctl->ngf;
    ctl->tcg_status = 4;
    ctl->tcg_iters = 0;
    ctl->tcg_done_stamp = INT_MAX;
  }
}
// delta = -z, z_r = <z, r>, d_Pd = z_r, e_Pe = e_Pd = 0
__global__ __launch_bounds__(kBlock) void k_tcg_init(long nelem, const double *__restrict__ z,
                                                     const double *__restrict__ p3, int np3,
                                                     double *__restrict__ delta, SolverCtl *ctl, int seq) {
  if (gated(ctl, seq, 1)) return;
  __shared__ double s_red[16];
  const double z_r = sum_partials(p3, np3, 1, 0, s_red);
  for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < nelem; i += (long)gridDim.x * kBlock)
    delta[i] = -z[i];
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    ctl->z_r[0] = z_r;
    ctl->d_Pd[0] = z_r;
    ctl->e_Pe[0] = 0;
    ctl->e_Pd[0] = 0;
  }
}

__global__ __launch_bounds__(kBlock) void k_tcg_update1(long nelem, const double *__restrict__ delta,
                                                        const double *__restrict__ Hd, double *__restrict__ eta,
                                                        double *__restrict__ Heta, double *__restrict__ res,
                                                        const double *__restrict__ p1, int np1,
                                                        double *__restrict__ p2, SolverCtl *ctl, HostFlags *hf,
                                                        int seq, int iter, int r, SpFold sf) {
  __shared__ double s_red[16];
  const int par = iter & 1;
  // Every load that depends on nothing is requested before the gate is looked at: control words, the partials, the
  // thread's first element of every vector and its place in the replay's image -- one memory round trip where the
  // gate, the partial sum, the control scalars and the vectors were four in a row (5.7 us on tiers.pyfg for 0.6 MB
  // vectors).  The empty asm keeps the compiler from sinking the loads behind the early return.
  const long i0 = (long)blockIdx.x * kBlock + threadIdx.x;
  const bool in0 = i0 < nelem;
  const GateWords gw = gate_words(ctl);
  double pv = ((int)threadIdx.x < np1) ? p1[threadIdx.x] : 0.0;
  const double z_r = ctl->z_r[par], d_Pd = ctl->d_Pd[par], e_Pe = ctl->e_Pe[par], e_Pd = ctl->e_Pd[par];
  const double Delta = ctl->Delta;
  double h0 = 0, dl0 = 0, et0 = 0, he0 = 0, rs0 = 0;
  int jp0 = -1;
  const long col0 = in0 ? i0 / r : 0;
  if (in0) {
    h0 = Hd[i0];
    dl0 = delta[i0];
    et0 = eta[i0];
    he0 = Heta[i0];
    rs0 = res[i0];
    if (sf.y) jp0 = sf.in_pos[col0];
  }
  asm volatile("" ::"v"(pv), "v"(h0), "v"(dl0), "v"(et0), "v"(he0), "v"(rs0), "v"(jp0), "s"(gw.outer), "s"(gw.tcg));
  if (gated(gw, ctl, seq, 2)) return;
  for (int i = threadIdx.x + blockDim.x; i < np1; i += blockDim.x) pv += p1[i];
  const double d_Hd = block_sum(pv, s_red);
  const double alpha = z_r / d_Hd;
  const double e_Pe_new = e_Pe + 2.0 * alpha * e_Pd + alpha * alpha * d_Pd;
  const bool boundary = (d_Hd <= 0) || (e_Pe_new >= Delta * Delta);
  const double step =
      boundary ? (-e_Pd + sqrt(e_Pd * e_Pd + d_Pd * (Delta * Delta - e_Pe))) / d_Pd : alpha;
  // a run that goes on says so before the vector work: the host enqueues the sparse replay behind this verdict
  // (DeviceProblem::rtr_dev); a run that stops writes tcg_done_seq below
  if (!boundary && blockIdx.x == 0 && threadIdx.x == 0) host_store(&hf->go_seq, seq);
  double acc = 0;
  if (in0) {
    eta[i0] = et0 + step * dl0;
    Heta[i0] = he0 + step * h0;
    if (!boundary) {
      const double rr = rs0 + alpha * h0;
      res[i0] = rr;
      acc += rr * rr;
      if (sf.y && jp0 >= 0) sf.y[(size_t)jp0 * r + (i0 - col0 * r)] = rr;
    }
  }
  for (long i = i0 + (long)gridDim.x * kBlock; i < nelem; i += (long)gridDim.x * kBlock) {
    const double h = Hd[i];
    eta[i] += step * delta[i];
    Heta[i] += step * h;
    if (!boundary) {
      const double rr = res[i] + alpha * h;
      res[i] = rr;
      acc += rr * rr;
      if (sf.y) {  // the replay's permute-in folded in: the new residual goes straight to its place in image 0
        const long col = i / r;
        const int jp = sf.in_pos[col];
        if (jp >= 0) sf.y[(size_t)jp * r + (i - col * r)] = rr;
      }
    }
  }
  const double tot = block_sum(acc, s_red);
  if (threadIdx.x == 0) p2[blockIdx.x] = tot;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    ctl->alpha = alpha;
    ctl->e_Pe_n = e_Pe_new;
    if (boundary) {
      ctl->tcg_status = (d_Hd <= 0) ? 0 : 1;
      ctl->tcg_iters = iter + 1;
      ctl->inner_total += iter + 1;
      ctl->tcg_done_stamp = seq;
      host_store(&hf->tcg_done_seq, seq);
    }
  }
}

__global__ __launch_bounds__(kBlock) void k_tcg_update2(long nelem, const double *__restrict__ z,
                                                        double *__restrict__ delta, const double *__restrict__ p3,
                                                        int np3, SolverCtl *ctl, HostFlags *hf, int seq, int iter) {
  if (gated(ctl, seq, 2)) return;
  __shared__ double s_red[16];
  const int par = iter & 1;
  const double z_r_new = sum_partials(p3, np3, 1, 0, s_red);
  const double z_r_old = ctl->z_r[par];
  const double beta = z_r_new / z_r_old;
  for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < nelem; i += (long)gridDim.x * kBlock)
    delta[i] = -z[i] + beta * delta[i];
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    const double alpha = ctl->alpha;
    const double d_Pd = ctl->d_Pd[par], e_Pd = ctl->e_Pd[par];
    ctl->z_r[par ^ 1] = z_r_new;
    ctl->e_Pd[par ^ 1] = beta * (e_Pd + alpha * d_Pd);
    ctl->d_Pd[par ^ 1] = z_r_new + beta * beta * d_Pd;
    ctl->e_Pe[par ^ 1] = ctl->e_Pe_n;
    if (iter + 1 >= ctl->max_inner) {  // loop exhausted: status stays TR_MAXITER
      ctl->tcg_iters = iter + 1;
      ctl->inner_total += iter + 1;
      ctl->tcg_done_stamp = seq;
      host_store(&hf->tcg_done_seq, seq);
    }
    host_store(&hf->last_seq_done, seq);
  }
}

// rho = (f1 - f2) / -(<eta, g> + 0.5 <eta, H eta>); accept iff rho > 0.1; radius update as RTRNewton
__global__ __launch_bounds__(kBlock) void k_rtr_decide(const double *pA, int npA, const double *pB, int npB,
                                                       const double *pC, int npC, SolverCtl *ctl, HostFlags *hf,
                                                       int seq, unsigned *tcg_sync, int nsync) {
  if (gated(ctl, seq, 1)) return;
  __shared__ double s_red[20];
  for (int i = threadIdx.x; i < nsync; i += kBlock) tcg_sync[i] = 0u;
  const double *const ps[5] = {pA, pA, pB, pC, pC};
  const int nps[5] = {npA, npA, npB, npC, npC}, sts[5] = {2, 2, 1, 2, 2}, offs[5] = {0, 1, 0, 0, 1};
  double sums[5];
  sum_partials_n<5>(ps, nps, sts, offs, s_red, sums);
  const double fq = sums[0], fg = sums[1], g2 = sums[2], eg = sums[3], eh = sums[4];
  if (threadIdx.x == 0) {
    const double f2 = 0.5 * fq + fg;
    const double rho = (ctl->f1 - f2) / (-(eg + 0.5 * eh));
    ctl->f2 = f2;
    ctl->rho = rho;
    if (rho > 0.75) {
      if (ctl->tcg_status == 0 || ctl->tcg_status == 1) ctl->Delta = fmin(2.0 * ctl->Delta, ctl->maxDelta);
    } else if (rho < 0.25) {
      ctl->Delta *= 0.25;
    }
    const bool accept = (rho > 0.1) && isfinite(rho);
    if (accept) {
      ctl->cur ^= 1;
      ctl->f1 = f2;
      ctl->ngf = sqrt(g2);
      ctl->accepted += 1;
      ctl->last_accepted = 1;
    } else {
      ctl->last_accepted = 0;
      host_store(&hf->reject_seq, seq);  // before last_seq_done below: the host reads it behind that word
    }
    ctl->outer_it += 1;
    if (ctl->ngf < ctl->tol || ctl->outer_it >= ctl->max_outer || (ctl->stop_on_accept && accept)) {
      ctl->outer_done_stamp = seq;
      host_store(&hf->outer_done_seq, seq);
    }
    host_store(&hf->last_seq_done, seq);
  }
}

void launch_rtr_init(hipStream_t st, const double *pA, int npA, const double *pB, int npB, SolverCtl *ctl,
                     HostFlags *hf, int seq, CtlInit ci, unsigned *tcg_sync, int nsync) {
  hipLaunchKernelGGL(k_rtr_init, dim3(1), dim3(kBlock), 0, st, pA, npA, pB, npB, ctl, hf, seq, ci, tcg_sync, nsync);
}
void launch_tcg_begin(hipStream_t st, long nelem, Buf2 grad, double *eta, double *Heta, double *res,
                      SolverCtl *ctl, int seq) {
  hipLaunchKernelGGL(k_tcg_begin, dim3(vec_grid(nelem)), dim3(kBlock), 0, st, nelem, grad, eta, Heta, res, ctl,
                     seq);
}
void launch_tcg_init(hipStream_t st, long nelem, const double *z, const double *p3, int np3, double *delta,
                     SolverCtl *ctl, int seq) {
  hipLaunchKernelGGL(k_tcg_init, dim3(vec_grid(nelem)), dim3(kBlock), 0, st, nelem, z, p3, np3, delta, ctl, seq);
}
void launch_tcg_update1(hipStream_t st, long nelem, const double *delta, const double *Hd, double *eta,
                        double *Heta, double *res, const double *p1, int np1, double *p2, SolverCtl *ctl,
                        HostFlags *hf, int seq, int iter, int r, SpFold sf) {
  hipLaunchKernelGGL(k_tcg_update1, dim3(vec_grid(nelem)), dim3(kBlock), 0, st, nelem, delta, Hd, eta, Heta, res,
                     p1, np1, p2, ctl, hf, seq, iter, r, sf);
}
void launch_tcg_update2(hipStream_t st, long nelem, const double *z, double *delta, const double *p3, int np3,
                        SolverCtl *ctl, HostFlags *hf, int seq, int iter) {
  hipLaunchKernelGGL(k_tcg_update2, dim3(vec_grid(nelem)), dim3(kBlock), 0, st, nelem, z, delta, p3, np3, ctl, hf,
                     seq, iter);
}
void launch_rtr_decide(hipStream_t st, const double *pA, int npA, const double *pB, int npB, const double *pC,
                       int npC, SolverCtl *ctl, HostFlags *hf, int seq, unsigned *tcg_sync, int nsync) {
  hipLaunchKernelGGL(k_rtr_decide, dim3(1), dim3(kBlock), 0, st, pA, npA, pB, npB, pC, npC, ctl, hf, seq, tcg_sync,
                     nsync);
}

// ------------------------------------------------------------------------------------------------------
// plain vector helpers
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_axpby(long nelem, double a, const double *__restrict__ x, double b,
                                                  const double *__restrict__ y, double *__restrict__ out) {
  for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < nelem; i += (long)gridDim.x * kBlock)
    out[i] = a * x[i] + (y ? b * y[i] : 0.0);
}
void launch_axpby(hipStream_t st, long nelem, double a, const double *x, double b, const double *y, double *out) {
  hipLaunchKernelGGL(k_axpby, dim3(vec_grid(nelem)), dim3(kBlock), 0, st, nelem, a, x, b, y, out);
}

__global__ __launch_bounds__(kBlock) void k_sum_partials(const double *partials, int np, int stride, int count,
                                                         double *out) {
  __shared__ double s_red[16];
  for (int c = 0; c < count; ++c) {
    const double s = sum_partials(partials, np, stride, c, s_red);
    if (threadIdx.x == 0) out[c] = s;
  }
}
void launch_sum_partials(hipStream_t st, const double *partials, int np, int stride, int count, double *out) {
  hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(kBlock), 0, st, partials, np, stride, count, out);
}

__global__ __launch_bounds__(kBlock) void k_dot(long nelem, const double *__restrict__ x,
                                                const double *__restrict__ y, double *__restrict__ partials) {
  __shared__ double s_red[16];
  double acc = 0;
  for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < nelem; i += (long)gridDim.x * kBlock)
    acc += x[i] * y[i];
  const double t = block_sum(acc, s_red);
  if (threadIdx.x == 0) partials[blockIdx.x] = t;
}
void launch_dot(hipStream_t st, long nelem, const double *x, const double *y, double *partials) {
  hipLaunchKernelGGL(k_dot, dim3(vec_grid(nelem)), dim3(kBlock), 0, st, nelem, x, y, partials);
}

__global__ __launch_bounds__(kBlock) void k_gather_cols(int r, int ncols, const int *__restrict__ src,
                                                        const double *__restrict__ X, double *__restrict__ out) {
  const long N = (long)ncols * r;
  for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < N; i += (long)gridDim.x * kBlock) {
    const int c = (int)(i / r), t = (int)(i - (long)c * r);
    out[i] = X[(size_t)src[c] * r + t];
  }
}
__global__ __launch_bounds__(kBlock) void k_scatter_cols(int r, int ncols, const int *__restrict__ dst,
                                                         const double *__restrict__ in, double *__restrict__ X) {
  const long N = (long)ncols * r;
  for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < N; i += (long)gridDim.x * kBlock) {
    const int c = (int)(i / r), t = (int)(i - (long)c * r);
    X[(size_t)dst[c] * r + t] = in[i];
  }
}
void launch_gather_cols(hipStream_t st, int r, int ncols, const int *src_col, const double *X, double *out) {
  if (ncols <= 0) return;
  hipLaunchKernelGGL(k_gather_cols, dim3(vec_grid((long)ncols * r)), dim3(kBlock), 0, st, r, ncols, src_col, X, out);
}
void launch_scatter_cols(hipStream_t st, int r, int ncols, const int *dst_col, const double *in, double *X) {
  if (ncols <= 0) return;
  hipLaunchKernelGGL(k_scatter_cols, dim3(vec_grid((long)ncols * r)), dim3(kBlock), 0, st, r, ncols, dst_col, in, X);
}

// one block per agent: |A_a|^2 and <A_a, B_a> over the agent's column range
__global__ __launch_bounds__(kBlock) void k_block_dots(int r, const int *__restrict__ cs,
                                                       const double *__restrict__ A, const double *__restrict__ B,
                                                       double *__restrict__ out) {
  __shared__ double s_red[16];
  const int a = blockIdx.x;
  const long lo = (long)cs[a] * r, hi = (long)cs[a + 1] * r;
  double s0 = 0, s1 = 0;
  for (long i = lo + threadIdx.x; i < hi; i += kBlock) {
    const double x = A[i];
    s0 += x * x;
    if (B) s1 += x * B[i];
  }
  const double t0 = block_sum(s0, s_red);
  const double t1 = block_sum(s1, s_red);
  if (threadIdx.x == 0) {
    out[2 * a] = t0;
    out[2 * a + 1] = t1;
  }
}
void launch_block_dots(hipStream_t st, int r, int nagents, const int *col_start, const double *A, const double *B,
                       double *out) {
  hipLaunchKernelGGL(k_block_dots, dim3(nagents), dim3(kBlock), 0, st, r, col_start, A, B, out);
}

// ------------------------------------------------------------------------------------------------------
// Lanczos helpers (certification): basis V is n x nv column-major (ld = n)
// ------------------------------------------------------------------------------------------------------
constexpr int kLanczosMaxV = 24;
__global__ __launch_bounds__(kBlock) void k_lanczos_proj(int n, int nv, const double *__restrict__ V,
                                                         const double *__restrict__ w,
                                                         double *__restrict__ partials) {
  __shared__ double s_red[16];
  double acc[kLanczosMaxV];
#pragma unroll
  for (int i = 0; i < kLanczosMaxV; ++i) acc[i] = 0;
  for (long t = (long)blockIdx.x * kBlock + threadIdx.x; t < n; t += (long)gridDim.x * kBlock) {
    const double wt = w[t];
#pragma unroll
    for (int i = 0; i < kLanczosMaxV; ++i)
      if (i < nv) acc[i] += V[(size_t)i * n + t] * wt;
  }
#pragma unroll
  for (int i = 0; i < kLanczosMaxV; ++i)
    if (i < nv) {
      const double s = block_sum(acc[i], s_red);
      if (threadIdx.x == 0) partials[(size_t)blockIdx.x * kLanczosMaxV + i] = s;
    }
}
__global__ __launch_bounds__(kBlock) void k_lanczos_sub(int n, int nv, const double *__restrict__ V,
                                                        const double *__restrict__ h, double *__restrict__ w) {
  for (long t = (long)blockIdx.x * kBlock + threadIdx.x; t < n; t += (long)gridDim.x * kBlock) {
    double s = 0;
    for (int i = 0; i < nv; ++i) s += V[(size_t)i * n + t] * h[i];
    w[t] -= s;
  }
}
void launch_lanczos_proj(hipStream_t st, int n, int nv, const double *V, const double *w, double *partials) {
  hipLaunchKernelGGL(k_lanczos_proj, dim3(vec_grid(n)), dim3(kBlock), 0, st, n, nv, V, w, partials);
}
void launch_lanczos_sub(hipStream_t st, int n, int nv, const double *V, const double *h, double *w) {
  hipLaunchKernelGGL(k_lanczos_sub, dim3(vec_grid(n)), dim3(kBlock), 0, st, n, nv, V, h, w);
}
// The same subtraction with the sum over the blocks' partial projections in its prologue (every workgroup sums the
// npart x nv partials for itself: for the problem sizes this is used at -- npart <= kLanczosFuseParts -- that is a
// few KB from L2 per workgroup and saves the k_sum_partials launch); block 0 keeps the coefficients for the host,
// which assembles the projected matrix once per restart cycle instead of once per step.
__global__ __launch_bounds__(kBlock) void k_lanczos_sub_sum(int n, int nv, const double *__restrict__ V,
                                                            const double *__restrict__ partials, int npart,
                                                            double *__restrict__ hout, double *__restrict__ w,
                                                            double *__restrict__ dotpart) {
  __shared__ double s_h[kLanczosMaxV];
  __shared__ double s_red[16];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int i = wave; i < nv; i += kBlock / 64) {
    double s = 0;
    for (int b = lane; b < npart; b += 64) s += partials[(size_t)b * kLanczosMaxV + i];
    s = wave_sum(s);
    if (lane == 0) s_h[i] = s;
  }
  __syncthreads();
  if (blockIdx.x == 0 && (int)threadIdx.x < nv) hout[threadIdx.x] = s_h[threadIdx.x];
  double ww = 0;
  for (long t = (long)blockIdx.x * kBlock + threadIdx.x; t < n; t += (long)gridDim.x * kBlock) {
    double s = 0;
    for (int i = 0; i < nv; ++i) s += V[(size_t)i * n + t] * s_h[i];
    const double wn = w[t] - s;
    w[t] = wn;
    ww += wn * wn;
  }
  if (dotpart) {  // second pass: <w, w> of the finished vector rides along (the norm's own launch saved)
    const double tot = block_sum(ww, s_red);
    if (threadIdx.x == 0) dotpart[blockIdx.x] = tot;
  }
}
// h only (large problems keep k_sum_partials + k_lanczos_sub): copies the summed coefficients to the per-step store
__global__ void k_lanczos_keep(int nv, const double *__restrict__ h, double *__restrict__ hout) {
  if ((int)threadIdx.x < nv) hout[threadIdx.x] = h[threadIdx.x];
}
// next basis vector: beta = |w| from the partial sums of <w, w> (summed by every workgroup in its prologue),
// v_next = w / beta; block 0 keeps beta for the host.  A vanishing beta (invariant subspace) raises *flag and leaves
// v_next = 0: the host redoes that step on its slow path.
__global__ __launch_bounds__(kBlock) void k_lanczos_next(int n, const double *__restrict__ partials, int npart,
                                                         double *__restrict__ beta_out, int *__restrict__ flag,
                                                         const double *__restrict__ w, double *__restrict__ vnext,
                                                         const double *__restrict__ hrow, int nv) {
  __shared__ double s_red[16];
  double s = 0;
  for (int b = threadIdx.x; b < npart; b += kBlock) s += partials[b];
  const double b2 = block_sum(s, s_red);
  const double beta = sqrt(b2);
  // |S v_j|^2 = beta^2 + sum of the squared coefficients taken out by the two passes: a remainder at the rounding level
  // of that norm is no direction (kLanczosDead, shared with the host's per-step form)
  double h2 = 0;
  for (int i = 0; i < nv; ++i) {
    const double h = hrow[i] + hrow[32 + i];
    h2 += h * h;
  }
  const bool dead = !(beta > kLanczosDead * sqrt(h2 + b2)) || !(beta >= 1e-300);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    *beta_out = dead ? 0.0 : beta;
    if (dead) *flag = 1;
  }
  const double inv = dead ? 0.0 : 1.0 / beta;
  for (long t = (long)blockIdx.x * kBlock + threadIdx.x; t < n; t += (long)gridDim.x * kBlock) vnext[t] = w[t] * inv;
}
void launch_lanczos_sub_sum(hipStream_t st, int n, int nv, const double *V, const double *partials, int npart,
                            double *hout, double *w, double *dotpart) {
  hipLaunchKernelGGL(k_lanczos_sub_sum, dim3(vec_grid(n)), dim3(kBlock), 0, st, n, nv, V, partials, npart, hout, w,
                     dotpart);
}
void launch_lanczos_keep(hipStream_t st, int nv, const double *h, double *hout) {
  hipLaunchKernelGGL(k_lanczos_keep, dim3(1), dim3(64), 0, st, nv, h, hout);
}
void launch_lanczos_next(hipStream_t st, int n, const double *partials, int npart, double *beta_out, int *flag,
                         const double *w, double *vnext, const double *hrow, int nv) {
  hipLaunchKernelGGL(k_lanczos_next, dim3(vec_grid(n)), dim3(kBlock), 0, st, n, partials, npart, beta_out, flag, w,
                     vnext, hrow, nv);
}
__global__ __launch_bounds__(kBlock) void k_scale_shift(int n, double shift, const double *__restrict__ x,
                                                        double *__restrict__ y) {
  for (long t = (long)blockIdx.x * kBlock + threadIdx.x; t < n; t += (long)gridDim.x * kBlock) y[t] -= shift * x[t];
}
void launch_scale_shift(hipStream_t st, int n, double shift, const double *x, double *y) {
  hipLaunchKernelGGL(k_scale_shift, dim3(vec_grid(n)), dim3(kBlock), 0, st, n, shift, x, y);
}
__global__ __launch_bounds__(kBlock) void k_scale(int n, const double *__restrict__ nrm2,
                                                  const double *__restrict__ w, double *__restrict__ out) {
  const double inv = 1.0 / sqrt(*nrm2);
  for (long t = (long)blockIdx.x * kBlock + threadIdx.x; t < n; t += (long)gridDim.x * kBlock) out[t] = w[t] * inv;
}
void launch_scale(hipStream_t st, int n, const double *nrm2, const double *w, double *out) {
  hipLaunchKernelGGL(k_scale, dim3(vec_grid(n)), dim3(kBlock), 0, st, n, nrm2, w, out);
}

}  // namespace dcora
