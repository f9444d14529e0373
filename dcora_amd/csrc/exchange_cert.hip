// Certification across the ranks of an exchange (SURVEY 8(e) "Collective": row-block S v with the halo exchange of the
// RBCD loop, scalar all-reduces for the Lanczos recurrences; ref src/DCORA_utils.cpp:1713-1735, 1809-1896).
#include <sched.h>

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstring>
#include <string>

#include "cert.h"
#include "device_chol.h"
#include "exchange.h"

namespace dcora {

namespace {

using Clock = std::chrono::steady_clock;
inline double since(Clock::time_point t0) { return std::chrono::duration<double>(Clock::now() - t0).count(); }

// w += eta v - Lambda v on the poses of one agent: Lambda_i (d x d, column-major) acts on the d rotation columns of
// pose i (ref constructDualCertificateMatrixPGO, src/DCORA_utils.cpp:1898-1931); r = 1 vectors, (d+1) entries per pose
__global__ __launch_bounds__(256) void k_apply_lambda(int n, int d, const double *__restrict__ L, double eta,
                                                      const double *__restrict__ v, double *__restrict__ w) {
  const int dh = d + 1;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < (long)n * dh; e += (long)gridDim.x * 256) {
    const int i = (int)(e / dh), c = (int)(e - (long)i * dh);
    double acc = eta * v[e];
    if (c < d) {
      const double *__restrict__ Li = L + (size_t)i * d * d;
      for (int b = 0; b < d; ++b) acc -= Li[c + b * d] * v[(size_t)i * dh + b];
    }
    w[e] += acc;
  }
}

}  // namespace

int Exchange::allreduce_sum(double *vals, int count) {
  if (count < 0 || count > 31) return usage("allreduce_sum: at most 31 values", DCORA_ERR_BAD_ARG);
  if (world == 1) return DCORA_OK;
  const uint64_t q = ++red_seq_;
  ShmRed *slots = red_ + (size_t)(q & 1) * world;
  for (int i = 0; i < count; ++i) slots[rank].vals[i] = vals[i];
  std::atomic_thread_fence(std::memory_order_release);
  slots[rank].seq = q;
  const auto t0 = Clock::now();
  double acc[31] = {0};
  for (int p = 0; p < world; ++p) {
    unsigned spins = 0;
    while (slots[p].seq < q) {
      ++spins;
      if (spins < 4096u) {
        __builtin_ia32_pause();
      } else {
        sched_yield();
        if ((spins & 255u) == 0) {
          if (hdr_->failed.load()) return fail("another rank failed", DCORA_ERR_HIP);
          if (since(t0) > exchange_timeout_s()) return fail("rank " + std::to_string(p) + " never joined a sum", DCORA_ERR_HIP);
        }
      }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    for (int i = 0; i < count; ++i) acc[i] += slots[p].vals[i];
  }
  for (int i = 0; i < count; ++i) vals[i] = acc[i];
  return DCORA_OK;
}

int Exchange::certify(const HostCsr *Qglobal, double eta, int *certified, double *theta, double *lambda_min, double *v,
                      long long *matvecs, int *distributed) {
  RbcdSession *pgo = dynamic_cast<RbcdSession *>(s_);
  if (!pgo)
    return usage("certify: the row-block certificate operator exists for pose-graph sessions (the range-aided certificate "
                "is assembled centrally: dcora_cert_dual_matrix on the gathered X)",
                DCORA_ERR_UNSUPPORTED);
  RbcdSession &S = *pgo;
  DCORA_HIP(hipSetDevice(S.opt.device));
  const int R = S.R, r = S.r, d = S.d, dh = d + 1, ktot = dh * S.n;
  if (certified) *certified = 0;
  if (theta) *theta = 0;
  if (lambda_min) *lambda_min = 0;
  if (matvecs) *matvecs = 0;
  if (distributed) *distributed = 0;
  std::vector<int> all((size_t)R);
  for (int a = 0; a < R; ++a) all[(size_t)a] = a;
  // every rank's mirror holds its neighbours' current public poses
  int rc = post(all.data(), R);
  if (rc) return rc;
  rc = wait(all.data(), R);
  if (rc) return rc;
  // ---- the PSD test, on rank 0: S = Q - Lambda(X) from the gathered X ----
  std::vector<double> Xh((size_t)r * ktot);
  rc = gather_X(Xh.data());
  if (rc) return rc;
  HostCsr M;
  double verdict[2] = {0, 0};  // {1 = PSD, 2 = not PSD; error code}
  if (rank == 0) {
    if (!Qglobal) {
      verdict[1] = DCORA_ERR_BAD_ARG;
    } else {
      dcora_dims dims{r, d, S.n, 0, 0};
      HostCsr Sh;
      int c = device_dual_certificate(dims, Xh.data(), *Qglobal, S.opt.device, &Sh);
      bool psd = false;
      if (!c) {
        M = csr_shift_diag(Sh, eta);
        c = device_chol_is_pd(M, dh, S.opt.device, &psd);
      }
      verdict[0] = c ? 0 : (psd ? 1 : 2);
      verdict[1] = c;
    }
  }
  rc = allreduce_sum(verdict, 2);  // the other ranks add zeros: a broadcast
  if (rc) return rc;
  if (verdict[1] != 0) return fail("certify: the PSD test failed on rank 0 (global Q missing?)", (int)verdict[1]);
  if (verdict[0] == 1) {
    if (certified) *certified = 1;
    return DCORA_OK;
  }
  // ---- minimum eigenpair of M = S + eta I by all ranks ----
  int lo = -1, nloc = 0;
  for (const AgentDev &a : S.agents)
    if (a.hosted) {
      if (lo < 0) lo = a.col0;
      nloc += dh * a.n;
    }
  if (lo < 0) lo = 0;
  // Lambda blocks of the hosted agents: EG_b = X_b Q_bb + X C_b, Lambda_i = sym(X_i^T EG_i)
  std::vector<DevBuf<double>> Lblk(S.agents.size()), tmpv(S.agents.size());
  for (AgentDev &a : S.agents) {
    if (!a.hosted) continue;
    DeviceProblem &pb = *a.prob;
    const double *Xb = S.Xg.p + (size_t)a.col0 * r;
    launch_spmm(S.st, r, a.coupling.view(), buf1(S.Xg.p), 0, nullptr, buf1(pb.G.p), 0, nullptr, Gate{});
    pb.has_G = true;
    pb.enqueue_egrad(Xb, pb.EG1.p, nullptr);
    DCORA_HIP(Lblk[(size_t)a.id].alloc((size_t)a.n * d * d));
    DCORA_HIP(tmpv[(size_t)a.id].alloc((size_t)dh * a.n));
    launch_lambda_blocks(S.st, pb.m, Xb, pb.EG1.p, Lblk[(size_t)a.id].p);
  }
  DevBuf<double> vg;  // the Lanczos vector of the whole problem: own entries + the neighbours' public ones are current
  DCORA_HIP(vg.alloc((size_t)ktot));
  DCORA_HIP(hipMemsetAsync(vg.p, 0, sizeof(double) * ktot, S.st));
  DeviceLanczos L;
  rc = L.init_rows(nloc, ktot, lo, S.opt.device, S.st);
  if (rc) return rc;
  L.allreduce = [this](double *vals, int count) { return allreduce_sum(vals, count); };
  L.op = [&](const double *vj, double *w) -> int {
    if (nloc > 0)
      DCORA_HIP(hipMemcpyAsync(vg.p + lo, vj, sizeof(double) * nloc, hipMemcpyDeviceToDevice, S.st));
    int c = post_arr(all.data(), R, 1, vg.p);
    if (c) return c;
    c = wait_arr(all.data(), R, 1, vg.p);
    if (c) return c;
    for (AgentDev &a : S.agents) {
      if (!a.hosted) continue;
      DeviceProblem &pb = *a.prob;
      double *t = tmpv[(size_t)a.id].p;
      double *wa = w + (a.col0 - lo);
      launch_spmm(S.st, 1, a.coupling.view(), buf1(vg.p), 0, nullptr, buf1(t), 0, nullptr, Gate{});
      launch_spmm(S.st, 1, pb.Q.view(), buf1(vg.p + a.col0), 0, t, buf1(wa), 0, nullptr, Gate{});
      const long ne = (long)dh * a.n;
      hipLaunchKernelGGL(k_apply_lambda, dim3((int)std::min<long>((ne + 255) / 256, 1024)), dim3(256), 0, S.st, a.n, d,
                         Lblk[(size_t)a.id].p, eta, vg.p + a.col0, wa);
    }
    DCORA_HIP(hipGetLastError());
    return DCORA_OK;
  };
  const uint64_t seed = 12345;
  const int ncv = std::min(20, ktot);
  LanczosResult lm, res;
  rc = L.largest_magnitude(0.0, ncv, 1000, 1e-4, nullptr, seed, &lm);
  if (rc) return rc;
  bool ok = lm.ok;
  long mv = lm.matvecs;
  if (ok && lm.lambda < 0) {
    res = lm;
  } else if (ok) {
    // the spectrum-shifted run; its start vector comes from the first row of M, which rank 0 holds
    std::vector<double> x0((size_t)ktot, 0.0);
    if (rank == 0) x0 = min_eig_second_start(M, seed);
    std::memcpy(xarea_, x0.data(), rank == 0 ? sizeof(double) * ktot : 0);
    rc = barrier();
    if (rc) return rc;
    std::memcpy(x0.data(), xarea_, sizeof(double) * ktot);
    rc = barrier();
    if (rc) return rc;
    LanczosResult sh;
    rc = L.largest_magnitude(2 * lm.lambda, ncv, 1000, eta / lm.lambda, x0.data(), seed, &sh);
    if (rc) return rc;
    mv += sh.matvecs;
    ok = sh.ok;
    sh.lambda += 2 * lm.lambda;
    res = sh;
  }
  double flag = ok ? 0 : 1;  // (identical on all ranks: every quantity the test reads was summed in rank order)
  rc = allreduce_sum(&flag, 1);
  if (rc) return rc;
  std::vector<double> vfull((size_t)ktot, 0.0);
  if (flag != 0) {
    // neither run converged: the shift-and-invert fallback (ref :1878-1888) needs a factorisation of the whole matrix
    // and stays on rank 0, which holds it
    double out2[2] = {0, 0};
    if (rank == 0) {
      LanczosResult e;
      const int c = device_min_eig(M, 1000, eta, 20, seed, S.opt.device, &e);
      if (c && c != DCORA_ERR_NO_CONVERGENCE) out2[1] = c;
      out2[0] = e.lambda;
      mv += e.matvecs;
      if (e.v.size() == (size_t)ktot) vfull = e.v;
    }
    rc = allreduce_sum(out2, 2);
    if (rc) return rc;
    if (out2[1] != 0) return fail("certify: the minimum eigenpair could not be computed", (int)out2[1]);
    res.lambda = out2[0];
    if (rank == 0) std::memcpy(xarea_, vfull.data(), sizeof(double) * ktot);
    rc = barrier();
    if (rc) return rc;
    std::memcpy(vfull.data(), xarea_, sizeof(double) * ktot);
    rc = barrier();
    if (rc) return rc;
  } else {
    if (distributed) *distributed = 1;
    if (nloc > 0) std::memcpy(xarea_ + lo, res.v.data(), sizeof(double) * nloc);
    rc = barrier();
    if (rc) return rc;
    std::memcpy(vfull.data(), xarea_, sizeof(double) * ktot);
    rc = barrier();
    if (rc) return rc;
  }
  // theta = v^T S v = v^T M v - eta (|v| = 1), with the row-block operator
  {
    DevBuf<double> vl, wl;
    DCORA_HIP(vl.alloc((size_t)std::max(nloc, 1)));
    DCORA_HIP(wl.alloc((size_t)std::max(nloc, 1)));
    if (nloc > 0)
      DCORA_HIP(hipMemcpyAsync(vl.p, vfull.data() + lo, sizeof(double) * nloc, hipMemcpyHostToDevice, S.st));
    rc = L.op(vl.p, wl.p);
    if (rc) return rc;
    std::vector<double> wh((size_t)std::max(nloc, 1));
    if (nloc > 0) DCORA_HIP(hipMemcpyAsync(wh.data(), wl.p, sizeof(double) * nloc, hipMemcpyDeviceToHost, S.st));
    DCORA_HIP(hipStreamSynchronize(S.st));
    double dots[2] = {0, 0};
    for (int i = 0; i < nloc; ++i) {
      dots[0] += vfull[(size_t)lo + i] * wh[(size_t)i];
      dots[1] += vfull[(size_t)lo + i] * vfull[(size_t)lo + i];
    }
    rc = allreduce_sum(dots, 2);
    if (rc) return rc;
    if (theta) *theta = dots[0] / dots[1] - eta;
  }
  if (lambda_min) *lambda_min = res.lambda;
  if (matvecs) *matvecs = mv;
  if (v) std::memcpy(v, vfull.data(), sizeof(double) * ktot);
  return DCORA_OK;
}

}  // namespace dcora
