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
#include "kernels.h"

namespace dcora {

namespace {

constexpr int GW = 8;  // lanes per pose

__device__ __forceinline__ bool f_gated(const SolverCtl *ctl, int seq, int gate) {
  if (seq > ctl->outer_done_stamp) return true;
  if (gate == 2 && seq > ctl->tcg_done_stamp) return true;
  return false;
}
__device__ __forceinline__ double *f_pick(const Buf2 &b, const SolverCtl *ctl, int sel) {
  return b.p[(ctl->cur ^ sel) & 1];
}
__device__ __forceinline__ double f_wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
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
__device__ __forceinline__ double f_sum_partials(const double *p, int np, int stride, int off, double *sm) {
  double v = 0;
  for (int i = threadIdx.x; i < np; i += blockDim.x) v += p[(size_t)i * stride + off];
  return f_block_sum(v, sm);
}
__device__ __forceinline__ void f_host_store(volatile int *p, int v) {
  __hip_atomic_store(const_cast<int *>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// sum over the 8 lanes of a pose group (every lane of the wave must take part).  The result is re-broadcast from
// the group's first lane: with FMA contraction the butterfly partial sums can differ in the last bit between
// lanes, and the Jacobi / Gram-Schmidt decisions taken from them must be identical across the group.
__device__ __forceinline__ double grp_sum(double v) {
  v += __shfl_xor(v, 1, 64);
  v += __shfl_xor(v, 2, 64);
  v += __shfl_xor(v, 4, 64);
  return __shfl(v, (int)(threadIdx.x & 63u & ~7u), 64);
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
// polar factor of the rotation part by one-sided Jacobi (wave-uniform sweep loop)
template <int D>
__device__ __forceinline__ void row_polar(Row<D> &A, bool live) {
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
        const double app = grp_sum(A.e[p] * A.e[p]);
        const double aqq = grp_sum(A.e[q] * A.e[q]);
        const double apq = grp_sum(A.e[p] * A.e[q]);
        const double sc = sqrt(app * aqq);
        if (fabs(apq) > 1e-16 * sc && fabs(apq) > 1e-300) {
          off = fmax(off, fabs(apq) / sc);
          const double zeta = (aqq - app) / (2.0 * apq);
          const double tt = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
          const double cs = 1.0 / sqrt(1.0 + tt * tt), sn = cs * tt;
          const double x = A.e[p], y = A.e[q];
          A.e[p] = cs * x - sn * y;
          A.e[q] = sn * x + cs * y;
#pragma unroll
          for (int i = 0; i < D; ++i) {
            const double vx = Vm[p][i], vy = Vm[q][i];
            Vm[p][i] = cs * vx - sn * vy;
            Vm[q][i] = sn * vx + cs * vy;
          }
        }
      }
    if (__all(!live || off < 1e-15)) break;
  }
  double u[D];
#pragma unroll
  for (int j = 0; j < D; ++j) {
    const double nn = grp_sum(A.e[j] * A.e[j]);
    u[j] = A.e[j] * (nn > 0 ? 1.0 / sqrt(nn) : 0.0);
  }
#pragma unroll
  for (int c = 0; c < D; ++c) {
    double s = 0;
#pragma unroll
    for (int j = 0; j < D; ++j) s += u[j] * Vm[j][c];
    A.e[c] = s;
  }
}

constexpr int kPosesPerBlock = kBlock / GW;  // 32
constexpr int kHessTile = 2560;              // nnz staged per pass (30 KiB of LDS)

// ------------------------------------------------------------------------------------------------------
// A: Hessian-vector product of the tCG direction, with the direction update folded into the gather
// ------------------------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(kBlock) void k_fused_hess(ManiDesc m, CsrDev Q, const double *__restrict__ z,
                                                       const double *__restrict__ d_old,
                                                       double *__restrict__ d_new, Buf2 Xb, Buf2 Sb,
                                                       double *__restrict__ Hd, const double *__restrict__ p3,
                                                       int np3, double *__restrict__ p1, SolverCtl *ctl, int seq,
                                                       int iter) {
  if (f_gated(ctl, seq, 2)) return;
  __shared__ int s_ci[kHessTile];
  __shared__ double s_v[kHessTile];
  __shared__ double s_red[16];
  constexpr int DH = D + 1;
  const int r = m.r;
  // ---- scalar recurrence (ROPTLIB tCG_TR): beta, e_Pd, d_Pd ----
  const double z_r_new = f_sum_partials(p3, np3, 1, 0, s_red);
  double beta = 0;
  const int par = iter & 1;
  if (iter > 0) beta = z_r_new / ctl->z_r[par ^ 1];
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    if (iter == 0) {
      ctl->z_r[0] = z_r_new;
      ctl->d_Pd[0] = z_r_new;
      ctl->e_Pe[0] = 0;
      ctl->e_Pd[0] = 0;
    } else {
      const double alpha = ctl->alpha, d_Pd = ctl->d_Pd[par ^ 1], e_Pd = ctl->e_Pd[par ^ 1];
      ctl->z_r[par] = z_r_new;
      ctl->e_Pd[par] = beta * (e_Pd + alpha * d_Pd);
      ctl->d_Pd[par] = z_r_new + beta * beta * d_Pd;
      ctl->e_Pe[par] = ctl->e_Pe_n;
    }
  }
  const double *__restrict__ X = f_pick(Xb, ctl, 0);
  const double *__restrict__ Sblk = f_pick(Sb, ctl, 0);
  const int t = threadIdx.x & (GW - 1);
  const int pose = blockIdx.x * kPosesPerBlock + (threadIdx.x >> 3);
  const bool active = (pose < m.n) && (t < r);
  const int j0 = blockIdx.x * kPosesPerBlock * DH;
  const int j1 = min(m.k, j0 + kPosesPerBlock * DH);
  const int pbeg = Q.rp[j0], pend = Q.rp[j1];
  int rb[DH], re[DH];
#pragma unroll
  for (int a = 0; a < DH; ++a) {
    rb[a] = active ? Q.rp[pose * DH + a] : 0;
    re[a] = active ? Q.rp[pose * DH + a + 1] : 0;
  }
  Row<D> W;
#pragma unroll
  for (int a = 0; a < DH; ++a) W.e[a] = 0;
  for (int base = pbeg; base < pend; base += kHessTile) {
    const int cnt = min(kHessTile, pend - base);
    __syncthreads();
    for (int i = threadIdx.x; i < cnt; i += kBlock) {
      s_ci[i] = Q.ci[base + i];
      s_v[i] = Q.v[base + i];
    }
    __syncthreads();
#pragma unroll
    for (int a = 0; a < DH; ++a) {
      const int lo = max(rb[a], base) - base, hi = min(re[a], base + cnt) - base;
      double acc = 0;
      for (int p = lo; p < hi; ++p) {
        const size_t o = (size_t)s_ci[p] * r + t;
        acc += s_v[p] * (iter > 0 ? beta * d_old[o] - z[o] : -z[o]);
      }
      W.e[a] += acc;
    }
  }
  // own rows: direction, correction, projection, dot
  const size_t o = (size_t)pose * DH * r;
  Row<D> Y, V;
  ld_row<D>(X + o, r, t, active, Y);
#pragma unroll
  for (int a = 0; a < DH; ++a)
    V.e[a] = active ? (iter > 0 ? beta * d_old[o + a * r + t] - z[o + a * r + t] : -z[o + a * r + t]) : 0.0;
  st_row<D>(d_new + o, r, t, active, V);
  double S[D][D];
#pragma unroll
  for (int a = 0; a < D; ++a)
#pragma unroll
    for (int b = 0; b < D; ++b) S[a][b] = (pose < m.n) ? Sblk[(size_t)pose * D * D + a + b * D] : 0.0;
  row_sub_AS<D>(W, V, S);
  row_tangent<D>(Y, W);
  st_row<D>(Hd + o, r, t, active, W);
  double acc = 0;
#pragma unroll
  for (int a = 0; a < DH; ++a) acc += V.e[a] * W.e[a];
  const double tot = f_block_sum(acc, s_red);
  if (threadIdx.x == 0) p1[blockIdx.x] = tot;
}

// ------------------------------------------------------------------------------------------------------
// B: step length, vector updates and the dense preconditioner product, split over row slices of Minv.
//    first != 0: start of a tCG run (res = grad, eta = H eta = 0, no step).
// ------------------------------------------------------------------------------------------------------
constexpr int kJChunk = 128;  // output columns per block (16-byte loads: 2 columns per lane)

template <int RM>
__global__ __launch_bounds__(kBlock) void k_fused_precond(int r, int k, int ldm, int nsplit,
                                                          const double *__restrict__ Minv, Buf2 gradb,
                                                          const double *__restrict__ delta,
                                                          const double *__restrict__ Hd, double *__restrict__ eta,
                                                          double *__restrict__ Heta,
                                                          const double *__restrict__ res_old,
                                                          double *__restrict__ res_new,
                                                          double *__restrict__ Zpart, const double *__restrict__ p1,
                                                          int np1, double *__restrict__ p2, SolverCtl *ctl,
                                                          HostFlags *hf, int seq, int iter, int first) {
  if (f_gated(ctl, seq, first ? 1 : 2)) return;
  __shared__ double s_red[16];
  __shared__ double s_acc[(kBlock / 64) * RM * kJChunk];
  const long N = (long)r * k;
  double alpha = 0, step = 0;
  bool boundary = false;
  const double *__restrict__ rsrc = res_old;
  if (first) {
    rsrc = f_pick(gradb, ctl, 0);
  } else {
    const int par = iter & 1;
    const double d_Hd = f_sum_partials(p1, np1, 1, 0, s_red);
    const double z_r = ctl->z_r[par], d_Pd = ctl->d_Pd[par], e_Pe = ctl->e_Pe[par], e_Pd = ctl->e_Pd[par];
    const double Delta = ctl->Delta;
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
      }
    }
  }
  if (first && blockIdx.x == 0 && threadIdx.x == 0) {
    ctl->norm_r0 = ctl->ngf;
    ctl->tcg_status = 4;
    ctl->tcg_iters = 0;
    ctl->tcg_done_stamp = INT_MAX;
  }
  // ---- element-wise updates (each element exactly once over the grid) ----
  double acc2 = 0;
  for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < N; i += (long)gridDim.x * kBlock) {
    if (first) {
      eta[i] = 0;
      Heta[i] = 0;
      res_new[i] = rsrc[i];
    } else {
      const double h = Hd[i];
      eta[i] += step * delta[i];
      Heta[i] += step * h;
      if (!boundary) {
        const double rr = res_old[i] + alpha * h;
        res_new[i] = rr;
        acc2 += rr * rr;
      }
    }
  }
  if (!first) {
    const double tot = f_block_sum(acc2, s_red);
    if (threadIdx.x == 0) p2[blockIdx.x] = tot;
  }
  if (boundary) return;
  // ---- dense product slice: Z_s(:, j) = sum_{c in slice} r(:, c) Minv(c, j) ----
  const int njc = (k + kJChunk - 1) / kJChunk;
  const int jc = blockIdx.x % njc, s = blockIdx.x / njc;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int rows_per_split = (k + nsplit - 1) / nsplit;
  const int c_lo = s * rows_per_split, c_hi = min(k, c_lo + rows_per_split);
  const int rows_per_wave = (rows_per_split + 3) / 4;
  const int w_lo = min(c_hi, c_lo + wave * rows_per_wave), w_hi = min(c_hi, w_lo + rows_per_wave);
  const int col = jc * kJChunk + 2 * lane;
  double a0[RM], a1[RM];
#pragma unroll
  for (int t = 0; t < RM; ++t) a0[t] = a1[t] = 0;
  const double *__restrict__ mp = Minv + (size_t)w_lo * ldm + col;
  int c = w_lo;
  for (; c + 4 <= w_hi; c += 4) {
    const double2 m0 = *reinterpret_cast<const double2 *>(mp);
    const double2 m1 = *reinterpret_cast<const double2 *>(mp + ldm);
    const double2 m2 = *reinterpret_cast<const double2 *>(mp + 2 * (size_t)ldm);
    const double2 m3 = *reinterpret_cast<const double2 *>(mp + 3 * (size_t)ldm);
    mp += 4 * (size_t)ldm;
    const double *__restrict__ r0 = rsrc + (size_t)c * r;
    const double *__restrict__ h0 = Hd + (size_t)c * r;
#pragma unroll
    for (int t = 0; t < RM; ++t)
      if (t < r) {
        double x0 = r0[t], x1 = r0[r + t], x2 = r0[2 * r + t], x3 = r0[3 * r + t];
        if (!first) {
          x0 += alpha * h0[t];
          x1 += alpha * h0[r + t];
          x2 += alpha * h0[2 * r + t];
          x3 += alpha * h0[3 * r + t];
        }
        a0[t] += x0 * m0.x + x1 * m1.x + x2 * m2.x + x3 * m3.x;
        a1[t] += x0 * m0.y + x1 * m1.y + x2 * m2.y + x3 * m3.y;
      }
  }
  for (; c < w_hi; ++c) {
    const double2 m0 = *reinterpret_cast<const double2 *>(mp);
    mp += ldm;
#pragma unroll
    for (int t = 0; t < RM; ++t)
      if (t < r) {
        double x0 = rsrc[(size_t)c * r + t];
        if (!first) x0 += alpha * Hd[(size_t)c * r + t];
        a0[t] += x0 * m0.x;
        a1[t] += x0 * m0.y;
      }
  }
  __syncthreads();
#pragma unroll
  for (int t = 0; t < RM; ++t) {
    s_acc[(wave * RM + t) * kJChunk + 2 * lane] = a0[t];
    s_acc[(wave * RM + t) * kJChunk + 2 * lane + 1] = a1[t];
  }
  __syncthreads();
  const int ncol = min(kJChunk, k - jc * kJChunk);
  double *__restrict__ zp = Zpart + (size_t)s * N + (size_t)jc * kJChunk * r;
  for (int e = threadIdx.x; e < ncol * r; e += kBlock) {
    const int cc = e / r, t = e - cc * r;
    double v = 0;
#pragma unroll
    for (int w = 0; w < kBlock / 64; ++w) v += s_acc[(w * RM + t) * kJChunk + cc];
    zp[e] = v;
  }
}

// ------------------------------------------------------------------------------------------------------
// C: residual stopping rule, z = Proj_X(sum of the split-K slices), partial <z, r>
// ------------------------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(kBlock) void k_fused_finish(ManiDesc m, int nsplit, Buf2 Xb,
                                                         const double *__restrict__ Zpart,
                                                         const double *__restrict__ res, double *__restrict__ z,
                                                         const double *__restrict__ p2, int np2,
                                                         double *__restrict__ p3, SolverCtl *ctl, HostFlags *hf,
                                                         int seq, int iter, int first) {
  if (f_gated(ctl, seq, first ? 1 : 2)) return;
  __shared__ double s_red[16];
  constexpr int DH = D + 1;
  const int r = m.r;
  if (!first) {
    const double nr = sqrt(f_sum_partials(p2, np2, 1, 0, s_red));
    const double n0 = ctl->norm_r0;
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
  const double *__restrict__ X = f_pick(Xb, ctl, 0);
  const long N = (long)r * m.k;
  const int t = threadIdx.x & (GW - 1);
  const int pose = blockIdx.x * kPosesPerBlock + (threadIdx.x >> 3);
  const bool active = (pose < m.n) && (t < r);
  const size_t o = (size_t)pose * DH * r;
  Row<D> Y, Zr, Rr;
  ld_row<D>(X + o, r, t, active, Y);
  ld_row<D>(res + o, r, t, active, Rr);
#pragma unroll
  for (int a = 0; a < DH; ++a) Zr.e[a] = 0;
  if (active)
    for (int s = 0; s < nsplit; ++s) {
      const double *__restrict__ zp = Zpart + (size_t)s * N + o;
#pragma unroll
      for (int a = 0; a < DH; ++a) Zr.e[a] += zp[a * r + t];
    }
  row_tangent<D>(Y, Zr);
  st_row<D>(z + o, r, t, active, Zr);
  double acc = 0;
#pragma unroll
  for (int a = 0; a < DH; ++a) acc += Zr.e[a] * Rr.e[a];
  const double tot = f_block_sum(acc, s_red);
  if (threadIdx.x == 0) p3[blockIdx.x] = tot;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    if (!first && iter + 1 >= ctl->max_inner) {  // inner loop exhausted: status stays TR_MAXITER
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
// RG = Proj_X(EG), S_i = sym(Y_i^T EG_i), partial |RG|^2
template <int D>
__global__ __launch_bounds__(kBlock) void k_g_rgrad(ManiDesc m, Buf2 Xb, Buf2 EGb, Buf2 RGb, Buf2 Sb, int sel,
                                                    double *__restrict__ partials, Gate g) {
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
#pragma unroll
    for (int a = 0; a < DH; ++a) acc += E.e[a] * E.e[a];
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
  const double *Xloc;
};
template <int D>
__global__ __launch_bounds__(kBlock) void k_g_nesterov(ManiDesc m, GNesterovArgs a) {
  constexpr int DH = D + 1;
  const int r = m.r;
  const int t = threadIdx.x & (GW - 1);
  for (int pose0 = blockIdx.x * kPosesPerBlock; pose0 < m.n; pose0 += gridDim.x * kPosesPerBlock) {
    const int pose = pose0 + (threadIdx.x >> 3);
    const bool inrange = (pose < m.n) && !(pose >= a.skip_lo && pose < a.skip_hi);
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
      if (a.mode == 1) {
        st_row<D>(a.Y + o, r, t, active, y);
        st_row<D>(a.Yloc + o, r, t, active, y);
      } else if (a.restart) {
        st_row<D>(a.V + o, r, t, active, x);
        st_row<D>(a.Y + o, r, t, active, x);
        // uniform control flow for the second polar below is not needed: restart is a kernel argument
      } else {
        const double vt = v.e[D];
        row_polar<D>(v, inrange);  // V + g (X - Y) with X == Y
        v.e[D] = vt;
        st_row<D>(a.Y + o, r, t, active, y);
        st_row<D>(a.X + o, r, t, active, y);
        st_row<D>(a.V + o, r, t, active, v);
      }
    } else {
      ld_row<D>(a.Xloc + o, r, t, active, x);
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

int group_grid(int n) {
  long g = ((long)n + kPosesPerBlock - 1) / kPosesPerBlock;
  if (g < 1) g = 1;
  if (g > kMaxPartials) g = kMaxPartials;
  return (int)g;
}

}  // namespace

bool fused_supported(const ManiDesc &m) { return m.se && m.r <= GW && m.n > 0; }
int fused_pose_blocks(const ManiDesc &m) { return (m.n + kPosesPerBlock - 1) / kPosesPerBlock; }
int fused_nsplit(const ManiDesc &m) {
  const int njc = (m.k + kJChunk - 1) / kJChunk;
  int ns = (512 + njc - 1) / njc;  // aim for ~512 blocks of 4 waves
  if (ns < 1) ns = 1;
  if (ns > 32) ns = 32;
  while (ns > 1 && (m.k + ns - 1) / ns < 16) --ns;
  return ns;
}
int fused_precond_grid(const ManiDesc &m) { return ((m.k + kJChunk - 1) / kJChunk) * fused_nsplit(m); }

void launch_fused_hess(hipStream_t st, const ManiDesc &m, const CsrDev &Q, const double *z, const double *d_old,
                       double *d_new, Buf2 X, Buf2 S, double *Hd, const double *p3, int np3, double *p1,
                       SolverCtl *ctl, int seq, int iter) {
  const int grid = fused_pose_blocks(m);
  if (m.d == 3)
    hipLaunchKernelGGL(k_fused_hess<3>, dim3(grid), dim3(kBlock), 0, st, m, Q, z, d_old, d_new, X, S, Hd, p3, np3,
                       p1, ctl, seq, iter);
  else
    hipLaunchKernelGGL(k_fused_hess<2>, dim3(grid), dim3(kBlock), 0, st, m, Q, z, d_old, d_new, X, S, Hd, p3, np3,
                       p1, ctl, seq, iter);
}
void launch_fused_precond(hipStream_t st, const ManiDesc &m, int ldm, const double *Minv, Buf2 grad,
                          const double *delta, const double *Hd, double *eta, double *Heta, const double *res_old,
                          double *res_new, double *Zpart, const double *p1, int np1, double *p2, SolverCtl *ctl,
                          HostFlags *hf, int seq, int iter, int first) {
  const int grid = fused_precond_grid(m);
  const int ns = fused_nsplit(m);
  if (m.r <= 4)
    hipLaunchKernelGGL(k_fused_precond<4>, dim3(grid), dim3(kBlock), 0, st, m.r, m.k, ldm, ns, Minv, grad, delta, Hd,
                       eta, Heta, res_old, res_new, Zpart, p1, np1, p2, ctl, hf, seq, iter, first);
  else
    hipLaunchKernelGGL(k_fused_precond<8>, dim3(grid), dim3(kBlock), 0, st, m.r, m.k, ldm, ns, Minv, grad, delta, Hd,
                       eta, Heta, res_old, res_new, Zpart, p1, np1, p2, ctl, hf, seq, iter, first);
}
void launch_fused_finish(hipStream_t st, const ManiDesc &m, Buf2 X, const double *Zpart, const double *res,
                         double *z, const double *p2, int np2, double *p3, SolverCtl *ctl, HostFlags *hf, int seq,
                         int iter, int first) {
  const int grid = fused_pose_blocks(m);
  const int ns = fused_nsplit(m);
  if (m.d == 3)
    hipLaunchKernelGGL(k_fused_finish<3>, dim3(grid), dim3(kBlock), 0, st, m, ns, X, Zpart, res, z, p2, np2, p3, ctl,
                       hf, seq, iter, first);
  else
    hipLaunchKernelGGL(k_fused_finish<2>, dim3(grid), dim3(kBlock), 0, st, m, ns, X, Zpart, res, z, p2, np2, p3, ctl,
                       hf, seq, iter, first);
}

int launch_g_rgrad(hipStream_t st, const ManiDesc &m, Buf2 X, Buf2 EG, Buf2 RG, Buf2 Sblk, int sel, double *partials,
                   Gate g) {
  const int grid = group_grid(m.n);
  if (m.d == 3)
    hipLaunchKernelGGL(k_g_rgrad<3>, dim3(grid), dim3(kBlock), 0, st, m, X, EG, RG, Sblk, sel, partials, g);
  else
    hipLaunchKernelGGL(k_g_rgrad<2>, dim3(grid), dim3(kBlock), 0, st, m, X, EG, RG, Sblk, sel, partials, g);
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
                       const double *Xloc) {
  GNesterovArgs a{mode, restart, skip_lo, skip_hi, alpha, gamma, X, V, Y, XPrev, Yloc, Xloc};
  const int grid = group_grid(m.n);
  if (m.d == 3)
    hipLaunchKernelGGL(k_g_nesterov<3>, dim3(grid), dim3(kBlock), 0, st, m, a);
  else
    hipLaunchKernelGGL(k_g_nesterov<2>, dim3(grid), dim3(kBlock), 0, st, m, a);
}

}  // namespace dcora
