// Certification on the device (replaces src/DCORA_utils.cpp:1713-1982):
//   * dual certificate S(X) = Q - Lambda(X): Q X^T by the SpMM kernel, Lambda blocks by a per-pose kernel;
//   * minimum eigenpair: thick-restart Lanczos (nev = 1, ncv = 20, largest magnitude, spectrum shift) whose
//     mat-vecs, re-orthogonalisation GEMVs and basis updates run on the GPU; only the <= 20 x 20 projected
//     eigenproblem is solved on the host;
//   * PSD test: sparse Cholesky of S + eta I on the host (setup-class code shared with the preconditioner).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <numeric>
#include <thread>

#include "cert.h"
#include "env.h"
#include "device_chol.h"
#include "device_problem.h"

namespace dcora {

namespace {
__global__ __launch_bounds__(kBlock) void k_basis_combine(int n, int m, int keep, const double *__restrict__ V,
                                                          const double *__restrict__ Z /* m x keep, col-major */,
                                                          double *__restrict__ out) {
  for (long t = (long)blockIdx.x * kBlock + threadIdx.x; t < n; t += (long)gridDim.x * kBlock)
    for (int c = 0; c < keep; ++c) {
      double s = 0;
      for (int i = 0; i < m; ++i) s += V[(size_t)i * n + t] * Z[(size_t)c * m + i];
      out[(size_t)c * n + t] = s;
    }
}

void jacobi_eig(int m, std::vector<double> A, std::vector<double> &Z, std::vector<double> &w) {
  Z.assign((size_t)m * m, 0.0);
  for (int i = 0; i < m; ++i) Z[(size_t)i * m + i] = 1;
  for (int sweep = 0; sweep < 100; ++sweep) {
    double off = 0, dg = 0;
    for (int i = 0; i < m; ++i) {
      dg += A[(size_t)i * m + i] * A[(size_t)i * m + i];
      for (int j = i + 1; j < m; ++j) off += A[(size_t)i * m + j] * A[(size_t)i * m + j];
    }
    if (off == 0 || off <= 1e-32 * (dg + off)) break;
    for (int p = 0; p < m - 1; ++p)
      for (int q = p + 1; q < m; ++q) {
        const double apq = A[(size_t)p * m + q];
        if (apq == 0) continue;
        const double zeta = (A[(size_t)q * m + q] - A[(size_t)p * m + p]) / (2 * apq);
        const double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1 + zeta * zeta));
        const double c = 1 / std::sqrt(1 + t * t), s = c * t;
        for (int k = 0; k < m; ++k) {
          const double akp = A[(size_t)k * m + p], akq = A[(size_t)k * m + q];
          A[(size_t)k * m + p] = c * akp - s * akq;
          A[(size_t)k * m + q] = s * akp + c * akq;
        }
        for (int k = 0; k < m; ++k) {
          const double apk = A[(size_t)p * m + k], aqk = A[(size_t)q * m + k];
          A[(size_t)p * m + k] = c * apk - s * aqk;
          A[(size_t)q * m + k] = s * apk + c * aqk;
          const double zkp = Z[(size_t)k * m + p], zkq = Z[(size_t)k * m + q];
          Z[(size_t)k * m + p] = c * zkp - s * zkq;
          Z[(size_t)k * m + q] = s * zkp + c * zkq;
        }
      }
  }
  w.resize(m);
  for (int i = 0; i < m; ++i) w[i] = A[(size_t)i * m + i];
}

inline uint64_t splitmix(uint64_t &s) {
  uint64_t z = (s += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
inline double u01(uint64_t &s) { return (splitmix(s) >> 11) * (1.0 / 9007199254740992.0); }
}  // namespace

int DeviceLanczos::init(const HostCsr &S, int device_) {
  device = device_;
  n = S.n;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    set_last_error("no HIP device available: libdcora_hip has no CPU fallback");
    return DCORA_ERR_NO_DEVICE;
  }
  DCORA_HIP(hipSetDevice(device));
  DCORA_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  int rc = Sd.upload(S);
  if (rc) return rc;
  DCORA_HIP(V.alloc((size_t)n * (kMaxNcv + 1)));
  DCORA_HIP(Vtmp.alloc((size_t)n * kMaxNcv));
  DCORA_HIP(w.alloc(n));
  DCORA_HIP(part.alloc((size_t)kMaxPartials * 24));
  DCORA_HIP(small.alloc(1024));
  return DCORA_OK;
}
int DeviceLanczos::init_rows(int n_local, int n_global_, int lo_, int device_, hipStream_t stream) {
  device = device_;
  n = n_local;
  n_global = n_global_;
  lo = lo_;
  st = stream;
  own_stream = false;
  DCORA_HIP(hipSetDevice(device));
  const size_t nn = (size_t)std::max(n, 1);
  DCORA_HIP(V.alloc(nn * (kMaxNcv + 1)));
  DCORA_HIP(Vtmp.alloc(nn * kMaxNcv));
  DCORA_HIP(w.alloc(nn));
  DCORA_HIP(part.alloc((size_t)kMaxPartials * 24));
  DCORA_HIP(small.alloc(1024));
  return DCORA_OK;
}
DeviceLanczos::~DeviceLanczos() {
  if (st && own_stream) (void)hipStreamDestroy(st);
}

// Thick-restart Lanczos for the largest-magnitude eigenpair of (S - shift I).  Convergence test and the
// number of Ritz vectors kept per restart follow Spectra's SymEigsSolver for nev = 1:
// |beta_m y_m| < tol max(eps^(2/3), |theta|), keep = ncv / 2.
int DeviceLanczos::largest_magnitude(double shift, int ncv, int maxit, double tol, const double *x0, uint64_t seed,
                                     LanczosResult *out) {
  DCORA_HIP(hipSetDevice(device));
  out->ok = false;
  out->v.assign(n, 0.0);
  out->matvecs = 0;
  const bool rows = (bool)allreduce;          // row-block form: sums over the ranks after every local reduction
  const int ng = rows ? n_global : n;         // order of the whole problem
  if (ng == 0) return DCORA_OK;
  const int m = std::min(std::min(ncv, kMaxNcv), ng);
  const int keep = std::max(1, std::min(m - 1, m / 2));
  std::vector<double> H((size_t)m * m, 0.0), Z, th, host_v((size_t)std::max(n, 1)), hbuf(64);
  // a vector of the whole problem from the seeded stream (or x0, given for the whole problem), this rank's slice of it
  std::vector<int> local_of;  // row_map inverted (scattered rows only)
  if (rows && !row_map.empty()) {
    local_of.assign((size_t)ng, -1);
    for (int i = 0; i < n; ++i) local_of[(size_t)row_map[(size_t)i]] = i;
  }
  auto whole_vector = [&](uint64_t s, const double *given, double *norm) {
    double nn = 0;
    for (int i = 0; i < ng; ++i) {
      const double x = given ? given[i] : (u01(s) - 0.5);
      if (!local_of.empty()) {
        if (local_of[(size_t)i] >= 0) host_v[(size_t)local_of[(size_t)i]] = x;
      } else if (i >= lo && i < lo + n) {
        host_v[(size_t)(i - lo)] = x;
      }
      nn += x * x;
    }
    *norm = std::sqrt(nn);
  };
  if (!rows) lo = 0;
  {
    double nn = 0;
    whole_vector(seed ? seed : 1, x0, &nn);
    for (double &x : host_v) x /= nn;
    DCORA_HIP(hipMemcpyAsync(V.p, host_v.data(), sizeof(double) * n, hipMemcpyHostToDevice, st));
    DCORA_HIP(hipStreamSynchronize(st));
  }
  // sums over the ranks of `count` doubles that sit on the device at dev (no-op on one rank)
  auto reduce_dev = [&](double *dev, int count) -> int {
    if (!rows) return DCORA_OK;
    std::vector<double> tmp((size_t)count);
    DCORA_HIP(hipMemcpyAsync(tmp.data(), dev, sizeof(double) * count, hipMemcpyDeviceToHost, st));
    DCORA_HIP(hipStreamSynchronize(st));
    const int rc2 = allreduce(tmp.data(), count);
    if (rc2) return rc2;
    DCORA_HIP(hipMemcpyAsync(dev, tmp.data(), sizeof(double) * count, hipMemcpyHostToDevice, st));
    DCORA_HIP(hipStreamSynchronize(st));  // tmp leaves scope
    return DCORA_OK;
  };
  const double eps23 = std::pow(2.220446049250313e-16, 2.0 / 3.0);
  const int npart = vec_grid(n);
  const CsrDev Sv = Sd.view();
  int k = 0;
  double beta = 0;
  std::vector<int> ord(m);
  // One Lanczos step = matvec, two passes of projection against the whole basis, the norm, the next vector.  The slow
  // form reads the coefficients and the norm back after every step (needed when the sums go over ranks, and to handle a
  // vanishing norm); the fast form keeps them on the device -- the subtraction sums the projection's partials in its
  // prologue, the next vector is scaled by the kernel that sums <w, w> -- and the host reads a whole restart cycle's
  // coefficients at once: 8 launches and no host round trip per step instead of 11 launches, two copies and a wait.
  const bool lanczos_slow = env::lanczos_sync();
  const bool fast = !rows && !lanczos_slow;
  DevBuf<double> hdev, bdev;
  DevBuf<int> fdev;
  if (fast) {
    DCORA_HIP(hdev.alloc((size_t)m * 64));
    DCORA_HIP(bdev.alloc((size_t)m));
    DCORA_HIP(fdev.alloc(1));
    DCORA_HIP(hipMemsetAsync(fdev.p, 0, sizeof(int), st));
  }
  auto matvec = [&](int j) -> int {
    double *vj = V.p + (size_t)j * n;
    if (op) {
      const int orc = op(vj, w.p);
      if (orc) return orc;
      if (shift != 0) launch_scale_shift(st, n, shift, vj, w.p);
    } else if (inverse_op) {
      inverse_op->apply(st, 1, buf1(vj), w.p, Gate{});
    } else {
      launch_spmm(st, 1, Sv, buf1(vj), 0, nullptr, buf1(w.p), 0, nullptr, Gate{});
      if (shift != 0) launch_scale_shift(st, n, shift, vj, w.p);
    }
    out->matvecs++;
    return DCORA_OK;
  };
  auto slow_step = [&](int j) -> int {
    int orc = matvec(j);
    if (orc) return orc;
    const int nv = j + 1;
    for (int pass = 0; pass < 2; ++pass) {
      launch_lanczos_proj(st, n, nv, V.p, w.p, part.p);
      launch_sum_partials(st, part.p, npart, 24, nv, small.p + 32 * pass);
      const int rrc = reduce_dev(small.p + 32 * pass, nv);
      if (rrc) return rrc;
      launch_lanczos_sub(st, n, nv, V.p, small.p + 32 * pass, w.p);
    }
    launch_dot(st, n, w.p, w.p, part.p);
    launch_sum_partials(st, part.p, npart, 1, 1, small.p + 64);
    {
      const int rrc = reduce_dev(small.p + 64, 1);
      if (rrc) return rrc;
    }
    DCORA_HIP(hipMemcpyAsync(hbuf.data(), small.p, sizeof(double) * 64, hipMemcpyDeviceToHost, st));
    double b2 = 0;
    DCORA_HIP(hipMemcpyAsync(&b2, small.p + 64, sizeof(double), hipMemcpyDeviceToHost, st));
    DCORA_HIP(hipStreamSynchronize(st));
    for (int i = 0; i < nv; ++i) {
      const double h = hbuf[i] + hbuf[32 + i];
      H[(size_t)i * m + j] = h;
      H[(size_t)j * m + i] = h;
    }
    beta = std::sqrt(b2);
    double h2 = 0;
    for (int i = 0; i < nv; ++i) h2 += H[(size_t)i * m + j] * H[(size_t)i * m + j];
    if (!(beta > kLanczosDead * std::sqrt(h2 + b2)) || beta < 1e-300) {
      // invariant subspace (the remainder is rounding noise of the orthogonalisation, or exactly zero): continue with
      // a fresh random direction orthogonalised against the basis
      double unused = 0;
      whole_vector(seed + 7919 * (j + 1), nullptr, &unused);
      DCORA_HIP(hipMemcpyAsync(w.p, host_v.data(), sizeof(double) * n, hipMemcpyHostToDevice, st));
      for (int pass = 0; pass < 2; ++pass) {
        launch_lanczos_proj(st, n, nv, V.p, w.p, part.p);
        launch_sum_partials(st, part.p, npart, 24, nv, small.p);
        const int rrc = reduce_dev(small.p, nv);
        if (rrc) return rrc;
        launch_lanczos_sub(st, n, nv, V.p, small.p, w.p);
      }
      launch_dot(st, n, w.p, w.p, part.p);
      launch_sum_partials(st, part.p, npart, 1, 1, small.p + 64);
      {
        const int rrc = reduce_dev(small.p + 64, 1);
        if (rrc) return rrc;
      }
      launch_scale(st, n, small.p + 64, w.p, V.p + (size_t)(j + 1) * n);
      DCORA_HIP(hipStreamSynchronize(st));
      beta = 0;
    } else {
      launch_axpby(st, n, 1.0 / beta, w.p, 0.0, nullptr, V.p + (size_t)(j + 1) * n);
    }
    return DCORA_OK;
  };
  auto fast_step = [&](int j) -> int {
    const int orc = matvec(j);
    if (orc) return orc;
    const int nv = j + 1;
    for (int pass = 0; pass < 2; ++pass) {
      double *hout = hdev.p + (size_t)j * 64 + 32 * pass;
      launch_lanczos_proj(st, n, nv, V.p, w.p, part.p);
      if (npart <= kLanczosFuseParts) {
        // (the dot's partials go behind the projection's in `part`: this kernel still reads those)
        launch_lanczos_sub_sum(st, n, nv, V.p, part.p, npart, hout, w.p, pass == 1 ? part.p + (size_t)npart * 24 : nullptr);
      } else {
        launch_sum_partials(st, part.p, npart, 24, nv, small.p + 32 * pass);
        launch_lanczos_keep(st, nv, small.p + 32 * pass, hout);
        launch_lanczos_sub(st, n, nv, V.p, small.p + 32 * pass, w.p);
      }
    }
    if (npart <= kLanczosFuseParts) {
      launch_lanczos_next(st, n, part.p + (size_t)npart * 24, npart, bdev.p + j, fdev.p, w.p, V.p + (size_t)(j + 1) * n,
                          hdev.p + (size_t)j * 64, nv);
    } else {
      launch_dot(st, n, w.p, w.p, part.p);
      launch_lanczos_next(st, n, part.p, npart, bdev.p + j, fdev.p, w.p, V.p + (size_t)(j + 1) * n, hdev.p + (size_t)j * 64, nv);
    }
    return DCORA_OK;
  };
  for (int it = 0; it <= maxit; ++it) {
    if (!fast) {
      for (int j = k; j < m; ++j) {
        const int src = slow_step(j);
        if (src) return src;
      }
    } else {
      for (int j = k; j < m; ++j) {
        const int frc = fast_step(j);
        if (frc) return frc;
      }
      std::vector<double> hh((size_t)m * 64), bb((size_t)m);
      int dead = 0;
      DCORA_HIP(hipMemcpyAsync(hh.data(), hdev.p, sizeof(double) * hh.size(), hipMemcpyDeviceToHost, st));
      DCORA_HIP(hipMemcpyAsync(bb.data(), bdev.p, sizeof(double) * bb.size(), hipMemcpyDeviceToHost, st));
      DCORA_HIP(hipMemcpyAsync(&dead, fdev.p, sizeof(int), hipMemcpyDeviceToHost, st));
      DCORA_HIP(hipStreamSynchronize(st));
      int jdead = m;
      if (dead)
        for (int j = k; j < m; ++j)
          if (!(bb[(size_t)j] >= 1e-300)) {
            jdead = j;
            break;
          }
      for (int j = k; j < jdead; ++j) {
        for (int i = 0; i <= j; ++i) {
          const double h = hh[(size_t)j * 64 + i] + hh[(size_t)j * 64 + 32 + i];
          H[(size_t)i * m + j] = h;
          H[(size_t)j * m + i] = h;
        }
        beta = bb[(size_t)j];
      }
      if (jdead < m) {  // a norm vanished at step jdead: that step and the rest of the cycle again, on the slow path
        DCORA_HIP(hipMemsetAsync(fdev.p, 0, sizeof(int), st));
        out->matvecs -= m - jdead;
        for (int j = jdead; j < m; ++j) {
          const int src = slow_step(j);
          if (src) return src;
        }
      }
    }
    jacobi_eig(m, H, Z, th);
    std::iota(ord.begin(), ord.end(), 0);
    std::sort(ord.begin(), ord.end(), [&](int a, int b) { return std::fabs(th[a]) > std::fabs(th[b]); });
    const int b0 = ord[0];
    const double resid = std::fabs(beta * Z[(size_t)(m - 1) * m + b0]);
    const bool conv = resid < tol * std::max(eps23, std::fabs(th[b0]));
    if (conv || it == maxit || m == ng) {
      out->ok = conv || (m == ng);
      out->lambda = th[b0];
      std::vector<double> zc(m);
      for (int i = 0; i < m; ++i) zc[i] = Z[(size_t)i * m + b0];
      DCORA_HIP(hipMemcpyAsync(small.p + 128, zc.data(), sizeof(double) * m, hipMemcpyHostToDevice, st));
      hipLaunchKernelGGL(k_basis_combine, dim3(vec_grid(n)), dim3(kBlock), 0, st, n, m, 1, V.p, small.p + 128,
                         Vtmp.p);
      DCORA_HIP(hipMemcpyAsync(out->v.data(), Vtmp.p, sizeof(double) * n, hipMemcpyDeviceToHost, st));
      DCORA_HIP(hipStreamSynchronize(st));
      double nn = 0;
      for (double x : out->v) nn += x * x;
      if (rows) {
        const int rrc = allreduce(&nn, 1);
        if (rrc) return rrc;
      }
      nn = std::sqrt(nn);
      for (double &x : out->v) x /= nn;
      return DCORA_OK;
    }
    // thick restart: V[:, 0:keep] = V Z[:, ord[0:keep]], V[:, keep] = v_{m}
    std::vector<double> Zk((size_t)m * keep);
    for (int c = 0; c < keep; ++c)
      for (int i = 0; i < m; ++i) Zk[(size_t)c * m + i] = Z[(size_t)i * m + ord[c]];
    DCORA_HIP(hipMemcpyAsync(small.p + 128, Zk.data(), sizeof(double) * m * keep, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_basis_combine, dim3(vec_grid(n)), dim3(kBlock), 0, st, n, m, keep, V.p, small.p + 128,
                       Vtmp.p);
    DCORA_HIP(hipMemcpyAsync(V.p, Vtmp.p, sizeof(double) * (size_t)n * keep, hipMemcpyDeviceToDevice, st));
    DCORA_HIP(hipMemcpyAsync(V.p + (size_t)keep * n, V.p + (size_t)m * n, sizeof(double) * n,
                             hipMemcpyDeviceToDevice, st));
    DCORA_HIP(hipStreamSynchronize(st));
    std::fill(H.begin(), H.end(), 0.0);
    for (int c = 0; c < keep; ++c) H[(size_t)c * m + c] = th[ord[c]];
    k = keep;
  }
  return DCORA_OK;
}

// start vector of the spectrum-shifted run: the first row of the matrix with a ~3 % random perturbation
// (ref src/DCORA_utils.cpp:1861-1866)
std::vector<double> min_eig_second_start(const HostCsr &S, uint64_t seed) {
  const int k = S.n;
  std::vector<double> x0((size_t)k, 0.0), pert((size_t)k);
  for (int p = S.rp[0]; p < S.rp[1]; ++p) x0[S.ci[p]] = S.v[p];
  uint64_t s = seed + 17;
  double pn = 0, vn = 0;
  for (int i = 0; i < k; ++i) {
    pert[i] = 2 * u01(s) - 1;
    pn += pert[i] * pert[i];
    vn += x0[i] * x0[i];
  }
  pn = std::sqrt(pn);
  vn = std::sqrt(vn);
  for (int i = 0; i < k; ++i) x0[i] += 0.03 * vn * pert[i] / pn;
  return x0;
}

// ref src/DCORA_utils.cpp:1809-1896
int device_min_eig(const HostCsr &S, int maxit, double min_eig_tol, int ncv, uint64_t seed, int device,
                   LanczosResult *out) {
  DeviceLanczos L;
  int rc = L.init(S, device);
  if (rc) return rc;
  const int k = S.n;
  ncv = std::min(ncv, k);
  LanczosResult lm;
  rc = L.largest_magnitude(0.0, ncv, maxit, 1e-4, nullptr, seed, &lm);
  if (rc) return rc;
  if (!lm.ok) {
    set_last_error("min_eig: could not compute the largest-magnitude eigenvalue of S");
    *out = lm;
    return DCORA_ERR_NO_CONVERGENCE;
  }
  if (lm.lambda < 0) {
    *out = lm;
    return DCORA_OK;
  }
  const double lambda_lm = lm.lambda;
  if (lambda_lm == 0) {  // S = 0: every vector is an eigenvector of the eigenvalue 0
    *out = lm;
    return DCORA_OK;
  }
  const std::vector<double> x0 = min_eig_second_start(S, seed);
  LanczosResult sh;
  // The spectrum-shifted run resolves the smallest eigenvalue to min_eig_tol on a spectrum of width lambda_lm.  When that
  // ratio is hopeless for a 20-vector Krylov space (tiers.pyfg: the landmark every pose ranges to puts lambda_lm at
  // 2e6, the ratio at 5e-10) the reference's run spends its 1000 restarts -- 10 020 matvecs, 0.6 s per certificate here --
  // and then takes the shift-and-invert fallback below anyway; this goes there at once.  The eigenpair returned is the
  // fallback's either way.
  const bool hopeless = min_eig_tol / lambda_lm < 1e-8;
  if (hopeless) {
    sh.ok = false;
    sh.v.assign((size_t)k, 0.0);
    sh.lambda = 0;
  } else {
    rc = L.largest_magnitude(2 * lambda_lm, ncv, maxit, min_eig_tol / lambda_lm, x0.data(), seed, &sh);
    if (rc) return rc;
  }
  sh.matvecs += lm.matvecs;
  if (env::init_timing())
    fprintf(stderr, "[min_eig] k %d: largest-magnitude run %ld matvecs (lambda %.3e), shifted run %ld matvecs, converged %d\n", k,
            lm.matvecs, lm.lambda, sh.matvecs - lm.matvecs, (int)sh.ok);
  if (!sh.ok) {
    // Shift-and-invert fallback (ref :1878-1888 -> :1751-1805): Lanczos on (S - sigma I)^-1, sigma = -10, halved
    // on failure with the floor -2 eta.  The solve per step is the partitioned sparse inverse of the SPD matrix
    // S - sigma I replayed on the device (sparse_precond.h), one right-hand side.
    // The factorisation is a Cholesky (the reference's Spectra run factors S - sigma I by a sparse LU, which exists for
    // any shift): S - sigma I must be positive definite, i.e. sigma < lambda_min.  When the very first shift, -10, is
    // refused, lambda_min lies below it: the shift moves OUTWARD by factors of ten -- at most to -lambda_lm, below
    // which nothing of the spectrum can lie -- until the matrix factors.
    double sigma = -10.0;
    bool outward = false;
    const int nthreads = std::max(1, std::min(host_cpus_available(), 32));
    for (int i = 0; i < 24; ++i) {
      PartInvHost P;
      DeviceWeightSink sink(device);  // the stored weights stream to the device while they are formed
      P.sink = &sink;
      const int brc = build_partitioned_inverse_auto(csr_shift_diag(S, -sigma), 1, nthreads, device, &P);
      if (brc && brc != DCORA_ERR_NOT_PD) return brc;
      if (brc == DCORA_ERR_NOT_PD && (i == 0 || outward)) {
        outward = true;
        if (-sigma > 2.0 * lambda_lm + 10.0) break;  // (cannot happen in exact arithmetic)
        sigma *= 10.0;
        continue;
      }
      if (brc == DCORA_OK) {
        SparsePrecond inv;
        auto img = std::make_shared<SpImage>();
        rc = img->upload(P, &sink);
        if (rc) return rc;
        rc = inv.attach(img, 1);
        if (rc) return rc;
        L.inverse_op = &inv;
        LanczosResult si;
        rc = L.largest_magnitude(0.0, ncv, 1000, 1e-10, nullptr, seed, &si);
        L.inverse_op = nullptr;
        if (rc) return rc;
        if (env::init_timing())
          fprintf(stderr, "[min_eig] shift-and-invert at sigma %.3g: %ld solves, converged %d\n", sigma, si.matvecs, (int)si.ok);
        if (si.ok) {
          si.matvecs += sh.matvecs;
          si.lambda = sigma + 1.0 / si.lambda;
          *out = si;
          return DCORA_OK;
        }
      }
      if (outward) break;  // factored far out but Lanczos on the inverse did not converge: the shifted run below
      if (i >= 9) break;
      sigma /= 2;
      if (i == 8 || sigma > -2 * min_eig_tol) sigma = -2 * min_eig_tol;
    }
    if (hopeless) {
      // the shortcut skipped the reference's first attempt and the fallback delivered nothing: make that attempt now
      LanczosResult late;
      rc = L.largest_magnitude(2 * lambda_lm, ncv, maxit, min_eig_tol / lambda_lm, x0.data(), seed, &late);
      if (rc) return rc;
      late.matvecs += sh.matvecs;
      if (late.ok) {
        late.lambda += 2 * lambda_lm;
        *out = late;
        return DCORA_OK;
      }
      sh = late;
    }
    set_last_error("min_eig: neither the spectrum-shifted nor the shift-and-invert Lanczos run converged");
    *out = sh;
    out->lambda += 2 * lambda_lm;
    return DCORA_ERR_NO_CONVERGENCE;
  }
  sh.lambda += 2 * lambda_lm;
  *out = sh;
  return DCORA_OK;
}

// ref src/DCORA_utils.cpp:1898-1982
int device_dual_certificate(const dcora_dims &dims, const double *Xh, const HostCsr &Q, int device, HostCsr *S) {
  if (dims.r < 1 || dims.r > 16 || (dims.d != 2 && dims.d != 3) || dims.n < 0 || dims.l < 0 || dims.b < 0) {
    set_last_error("bad dims (need 1 <= r <= 16, d in {2,3})");
    return DCORA_ERR_BAD_ARG;
  }
  const ManiDesc m = make_mani(dims);
  if (Q.n != m.k) {
    set_last_error("Q dimension does not match (d+1) n + l + b");
    return DCORA_ERR_BAD_ARG;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    set_last_error("no HIP device available: libdcora_hip has no CPU fallback");
    return DCORA_ERR_NO_DEVICE;
  }
  DCORA_HIP(hipSetDevice(device));
  // only what the two kernels need: Q, X, X Q and the Lambda blocks (no solver workspace).  Q without long rows (a
  // pose graph, or a range-aided one whose landmarks are ranged from few poses) goes into ONE recycled scratch block
  // with the vectors: six allocations and six frees per certificate were 3 of its 5 ms at k = 10 000.
  const size_t N = (size_t)m.r * m.k;
  const size_t NL = (size_t)m.n * m.d * m.d + m.l;
  std::vector<double> L(NL + 1);
  bool long_rows = false;
  for (int i = 0; i < Q.n && !long_rows; ++i) long_rows = Q.rp[i + 1] - Q.rp[i] > kLongRow;
  if (!long_rows) {
    auto up8 = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t nnz = (size_t)Q.nnz();
    const size_t o_v = 0, o_x = o_v + up8(nnz * 8), o_xq = o_x + up8(N * 8), o_l = o_xq + up8(N * 8),
                 o_rp = o_l + up8((NL + 1) * 8), o_ci = o_rp + up8((size_t)(Q.n + 1) * 4), total = o_ci + up8(nnz * 4);
    struct Arena {
      int device;
      size_t bytes;
      char *p;
      ~Arena() { scratch_release(device, p, bytes); }
    } ar{device, total, scratch_acquire(device, total)};
    if (!ar.p) {
      set_last_error("dual certificate: device allocation failed");
      return DCORA_ERR_HIP;
    }
    double *dv = (double *)(ar.p + o_v), *dX = (double *)(ar.p + o_x), *dXQ = (double *)(ar.p + o_xq),
           *dL = (double *)(ar.p + o_l);
    int *drp = (int *)(ar.p + o_rp), *dci = (int *)(ar.p + o_ci);
    // the inputs travel through ONE pinned block laid out like the device block (copies from pageable memory stall
    // for tens of milliseconds now and then: device_chol.h), the Lambda blocks come back through its tail
    struct Pin {
      size_t bytes;
      char *p;
      ~Pin() { pinned_release(p, bytes); }
    } pin{total, total <= ((size_t)256 << 20) ? pinned_acquire(total) : nullptr};
    if (pin.p) {
      std::memcpy(pin.p + o_v, Q.v.data(), nnz * 8);
      std::memcpy(pin.p + o_x, Xh, sizeof(double) * N);
      std::memcpy(pin.p + o_rp, Q.rp.data(), (size_t)(Q.n + 1) * 4);
      std::memcpy(pin.p + o_ci, Q.ci.data(), nnz * 4);
      DCORA_HIP(hipMemcpyAsync(dv, pin.p + o_v, o_xq - o_v, hipMemcpyHostToDevice, nullptr));
      DCORA_HIP(hipMemcpyAsync(drp, pin.p + o_rp, total - o_rp, hipMemcpyHostToDevice, nullptr));
    } else {
      DCORA_HIP(hipMemcpyAsync(dv, Q.v.data(), nnz * 8, hipMemcpyHostToDevice, nullptr));
      DCORA_HIP(hipMemcpyAsync(drp, Q.rp.data(), (size_t)(Q.n + 1) * 4, hipMemcpyHostToDevice, nullptr));
      DCORA_HIP(hipMemcpyAsync(dci, Q.ci.data(), nnz * 4, hipMemcpyHostToDevice, nullptr));
      DCORA_HIP(hipMemcpyAsync(dX, Xh, sizeof(double) * N, hipMemcpyHostToDevice, nullptr));
    }
    CsrDev Qv;
    Qv.nrows = Q.n;
    Qv.nnz = (int)nnz;
    Qv.rp = drp;
    Qv.ci = dci;
    Qv.v = dv;
    launch_spmm(nullptr, m.r, Qv, buf1(dX), 0, nullptr, buf1(dXQ), 0, nullptr, Gate{});
    launch_lambda_blocks(nullptr, m, dX, dXQ, dL);
    if (pin.p) {
      DCORA_HIP(hipMemcpyAsync(pin.p + o_l, dL, sizeof(double) * NL, hipMemcpyDeviceToHost, nullptr));
      DCORA_HIP(hipStreamSynchronize(nullptr));
      std::memcpy(L.data(), pin.p + o_l, sizeof(double) * NL);
    } else {
      DCORA_HIP(hipMemcpy(L.data(), dL, sizeof(double) * NL, hipMemcpyDeviceToHost));
    }
  } else {
    DevCsr Qd;
    int rc = Qd.upload(Q);
    if (rc) return rc;
    DevBuf<double> dX, dXQ, dL;
    DCORA_HIP(dX.alloc(N));
    DCORA_HIP(dXQ.alloc(N));
    DCORA_HIP(dL.alloc(NL + 1));
    DCORA_HIP(hipMemcpy(dX.p, Xh, sizeof(double) * N, hipMemcpyHostToDevice));
    launch_spmm(nullptr, m.r, Qd.view(), buf1(dX.p), 0, nullptr, buf1(dXQ.p), 0, nullptr, Gate{});
    launch_lambda_blocks(nullptr, m, dX.p, dXQ.p, dL.p);
    DCORA_HIP(hipMemcpy(L.data(), dL.p, sizeof(double) * NL, hipMemcpyDeviceToHost));
  }
  // S = Q - Lambda.  Lambda is block diagonal on entries that Q's own pattern holds (the d x d rotation blocks
  // and the unit-sphere diagonal): subtract in place on a copy of Q; fall back to a merge when an entry is absent.
  {
    HostCsr T = Q;
    bool all_found = true;
    auto sub = [&](int i, int j, double val) {
      const int *lo = T.ci.data() + T.rp[i], *hi = T.ci.data() + T.rp[i + 1];
      const int *it = std::lower_bound(lo, hi, j);
      if (it == hi || *it != j) {
        all_found = false;
        return;
      }
      T.v[it - T.ci.data()] -= val;
    };
    const int d = m.d;
    for (int i = 0; i < m.n && all_found; ++i) {
      const int c = m.rot_col(i);
      for (int a = 0; a < d; ++a)
        for (int b = 0; b < d; ++b) sub(c + a, c + b, L[(size_t)i * d * d + a + b * d]);
    }
    for (int i = 0; i < m.l && all_found; ++i) sub(m.sphere_col(i), m.sphere_col(i), L[(size_t)m.n * d * d + i]);
    if (all_found) {
      *S = std::move(T);
      return DCORA_OK;
    }
  }
  std::vector<int> I, J;
  std::vector<double> V;
  I.reserve(Q.nnz() + NL);
  J.reserve(Q.nnz() + NL);
  V.reserve(Q.nnz() + NL);
  for (int i = 0; i < Q.n; ++i)
    for (int p = Q.rp[i]; p < Q.rp[i + 1]; ++p) {
      I.push_back(i);
      J.push_back(Q.ci[p]);
      V.push_back(Q.v[p]);
    }
  const int d = m.d;
  for (int i = 0; i < m.n; ++i) {
    const int c = m.rot_col(i);
    for (int a = 0; a < d; ++a)
      for (int b = 0; b < d; ++b) {
        I.push_back(c + a);
        J.push_back(c + b);
        V.push_back(-L[(size_t)i * d * d + a + b * d]);
      }
  }
  for (int i = 0; i < m.l; ++i) {
    const int c = m.sphere_col(i);
    I.push_back(c);
    J.push_back(c);
    V.push_back(-L[(size_t)m.n * d * d + i]);
  }
  *S = csr_from_coo(Q.n, Q.n, I, J, V);
  return DCORA_OK;
}

int host_is_psd(const HostCsr &S, int block, bool *psd) {
  SparseChol c;
  *psd = c.factor(S, block);
  return DCORA_OK;
}

// ref src/DCORA_utils.cpp:1713-1735
int device_fast_verification(const HostCsr &S, double eta, int block, int device, bool *psd, double *theta,
                             std::vector<double> *x, double *lambda_min, long *matvecs) {
  HostCsr M = csr_shift_diag(S, eta);
  // the PSD test: LL^T of S + eta I succeeds <=> PSD up to eta (ref src/DCORA_utils.cpp:1737-1747), factorised on the
  // device (device_chol.h)
  int rc = device_chol_is_pd(M, block, device, psd);
  if (rc) return rc;
  if (*psd) return DCORA_OK;
  LanczosResult e;
  rc = device_min_eig(M, 1000, eta, 20, 12345, device, &e);
  if (rc && rc != DCORA_ERR_NO_CONVERGENCE) return rc;
  // theta = v^T S v
  double th = 0;
  for (int i = 0; i < S.n; ++i) {
    double s = 0;
    for (int p = S.rp[i]; p < S.rp[i + 1]; ++p) s += S.v[p] * e.v[S.ci[p]];
    th += e.v[i] * s;
  }
  if (theta) *theta = th;
  if (x) *x = e.v;
  if (lambda_min) *lambda_min = e.lambda;
  if (matvecs) *matvecs = e.matvecs;
  return rc;
}

}  // namespace dcora
