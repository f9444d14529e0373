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
#include "ra_rbcd.h"

namespace dcora {

namespace {

using Clock = std::chrono::steady_clock;
inline double since(Clock::time_point t0) { return std::chrono::duration<double>(Clock::now() - t0).count(); }

// w += eta v - Lambda v on the variables of one agent, r = 1 vectors in the agent's ordering: Lambda is block diagonal
// -- a d x d block (column-major) on the d rotation columns of every pose, a scalar on every unit sphere, nothing on
// translations and landmarks (ref constructDualCertificateMatrixPGO / ...RASLAM, src/DCORA_utils.cpp:1898-1982); L as
// k_lambda leaves it: n blocks, then l scalars.  Pose layout: d + 1 entries per pose; range-aided layout: rotations,
// unit spheres, translations, landmarks.
__global__ __launch_bounds__(256) void k_apply_lambda(ManiDesc m, const double *__restrict__ L, double eta,
                                                      const double *__restrict__ v, double *__restrict__ w) {
  const int d = m.d;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < (long)m.k; e += (long)gridDim.x * 256) {
    double acc = eta * v[e];
    if (m.se) {
      const int dh = d + 1, i = (int)(e / dh), c = (int)(e - (long)i * dh);
      if (c < d) {
        const double *__restrict__ Li = L + (size_t)i * d * d;
        for (int b = 0; b < d; ++b) acc -= Li[c + b * d] * v[(size_t)i * dh + b];
      }
    } else if (e < (long)d * m.n) {
      const int i = (int)(e / d), c = (int)(e - (long)i * d);
      const double *__restrict__ Li = L + (size_t)i * d * d;
      for (int b = 0; b < d; ++b) acc -= Li[c + b * d] * v[(size_t)i * d + b];
    } else if (e < (long)d * m.n + m.l) {
      acc -= L[(size_t)m.n * d * d + (e - (long)d * m.n)] * v[e];
    }
    w[e] += acc;
  }
}

// whole[map[i]] = local[i]
__global__ __launch_bounds__(256) void k_scatter_rows(int n, const int *__restrict__ map, const double *__restrict__ local,
                                                      double *__restrict__ whole) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) whole[map[i]] = local[i];
}

// one hosted agent of either session kind, as the row-block operator sees it
struct CertBlock {
  int k = 0;                     // its unknowns
  int loc0 = 0;                  // where they start in this rank's local vectors
  int col0 = 0;                  // pose-graph session: its first column of the global ordering (contiguous)
  const int *own_dev = nullptr;  // range-aided session: the global column of each of its columns
  DeviceProblem *prob = nullptr;
  CsrDev coupling;
  const double *X = nullptr;     // r x k, its ordering
  DevBuf<double> L, tmp;
};

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
  // the pose-graph and the range-aided session through one flow: what differs is where an agent's variables sit in the
  // global ordering (a contiguous range / scattered: CertBlock) and the layout its Lambda blocks act on
  RbcdSession *pgo = dynamic_cast<RbcdSession *>(s_);
  RaRbcdSession *ras = dynamic_cast<RaRbcdSession *>(s_);
  if (!pgo && !ras) return usage("certify: unknown session kind", DCORA_ERR_UNSUPPORTED);
  const int device = s_->x_device();
  hipStream_t st = s_->x_stream();
  DCORA_HIP(hipSetDevice(device));
  const int R = s_->x_num_agents(), r = s_->x_rank_r(), ktot = (int)s_->x_num_cols();
  const dcora_dims dims = pgo ? dcora_dims{r, pgo->d, pgo->n, 0, 0} : dcora_dims{r, ras->d, ras->n, ras->l, ras->b, DCORA_LAYOUT_RA};
  const int chol_block = pgo ? pgo->d + 1 : 1;
  double *mirror = s_->x_mirror();
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
      HostCsr Sh;
      int c = device_dual_certificate(dims, Xh.data(), *Qglobal, device, &Sh);
      bool psd = false;
      if (!c) {
        M = csr_shift_diag(Sh, eta);
        c = device_chol_is_pd(M, chol_block, device, &psd);
      }
      verdict[0] = c ? 0 : (psd ? 1 : 2);
      verdict[1] = c;
    }
  }
  rc = allreduce_sum(verdict, 2);  // the other ranks add zeros: a broadcast
  if (rc) return rc;
  // (every rank learns the code from the sum above and leaves the call together: a missing Q on rank 0 is a usage error
  // of the caller, not a failure of the job)
  if ((int)verdict[1] == DCORA_ERR_BAD_ARG) return usage("certify: rank 0 needs the global Q", DCORA_ERR_BAD_ARG);
  if (verdict[1] != 0) return fail("certify: the PSD test failed on rank 0", (int)verdict[1]);
  if (verdict[0] == 1) {
    if (certified) *certified = 1;
    return DCORA_OK;
  }
  // ---- minimum eigenpair of M = S + eta I by all ranks ----
  std::vector<CertBlock> blocks;
  int nloc = 0, lo = -1;
  std::vector<int> row_map;  // global index of every local row (range-aided sessions)
  if (pgo) {
    for (AgentDev &a : pgo->agents) {
      if (!a.hosted) continue;
      CertBlock B;
      B.k = (pgo->d + 1) * a.n;
      B.loc0 = nloc;
      B.col0 = a.col0;
      B.prob = a.prob.get();
      B.coupling = a.coupling.view();
      B.X = pgo->Xg.p + (size_t)a.col0 * r;
      if (lo < 0) lo = a.col0;
      nloc += B.k;
      blocks.push_back(std::move(B));
    }
  } else {
    for (RaAgentDev &a : ras->agents) {
      if (!a.hosted) continue;
      CertBlock B;
      B.k = a.k;
      B.loc0 = nloc;
      B.own_dev = a.own.p;
      B.prob = a.prob.get();
      B.coupling = a.coupling.view();
      B.X = a.X.p;
      row_map.insert(row_map.end(), a.own_host.begin(), a.own_host.end());
      nloc += B.k;
      blocks.push_back(std::move(B));
    }
  }
  if (lo < 0) lo = 0;
  // Lambda blocks of the hosted agents: EG_b = X_b Q_bb + X C_b, Lambda = SymBlockDiag(X_b^T EG_b)
  for (CertBlock &B : blocks) {
    DeviceProblem &pb = *B.prob;
    launch_spmm(st, r, B.coupling, buf1(mirror), 0, nullptr, buf1(pb.G.p), 0, nullptr, Gate{});
    pb.has_G = true;
    pb.enqueue_egrad(B.X, pb.EG1.p, nullptr);
    DCORA_HIP(B.L.alloc((size_t)pb.m.n * pb.m.d * pb.m.d + (size_t)std::max(pb.m.l, 1)));
    DCORA_HIP(B.tmp.alloc((size_t)B.k));
    launch_lambda_blocks(st, pb.m, B.X, pb.EG1.p, B.L.p);
  }
  DevBuf<double> vg;  // the Lanczos vector of the whole problem: own entries + the neighbours' public ones are current
  DCORA_HIP(vg.alloc((size_t)ktot));
  DCORA_HIP(hipMemsetAsync(vg.p, 0, sizeof(double) * ktot, st));
  DeviceLanczos L;
  rc = L.init_rows(nloc, ktot, lo, device, st);
  if (rc) return rc;
  L.row_map = row_map;
  L.allreduce = [this](double *vals, int count) { return allreduce_sum(vals, count); };
  L.op = [&](const double *vj, double *w) -> int {
    for (CertBlock &B : blocks) {
      if (B.own_dev)
        hipLaunchKernelGGL(k_scatter_rows, dim3(std::min((B.k + 255) / 256, 1024)), dim3(256), 0, st, B.k, B.own_dev,
                           vj + B.loc0, vg.p);
      else
        DCORA_HIP(hipMemcpyAsync(vg.p + B.col0, vj + B.loc0, sizeof(double) * B.k, hipMemcpyDeviceToDevice, st));
    }
    int c = post_arr(all.data(), R, 1, vg.p);
    if (c) return c;
    c = wait_arr(all.data(), R, 1, vg.p);
    if (c) return c;
    for (CertBlock &B : blocks) {
      DeviceProblem &pb = *B.prob;
      double *wa = w + B.loc0;
      launch_spmm(st, 1, B.coupling, buf1(vg.p), 0, nullptr, buf1(B.tmp.p), 0, nullptr, Gate{});
      launch_spmm(st, 1, pb.Q.view(), buf1(vj + B.loc0), 0, B.tmp.p, buf1(wa), 0, nullptr, Gate{});
      hipLaunchKernelGGL(k_apply_lambda, dim3(std::min((B.k + 255) / 256, 1024)), dim3(256), 0, st, pb.m, B.L.p, eta,
                         vj + B.loc0, wa);
    }
    DCORA_HIP(hipGetLastError());
    return DCORA_OK;
  };
  // this rank's rows of a vector of the whole problem, and back
  auto local_of_whole = [&](const double *whole, double *local) {
    for (int i = 0; i < nloc; ++i) local[i] = whole[row_map.empty() ? (size_t)lo + i : (size_t)row_map[(size_t)i]];
  };
  auto whole_of_local = [&](const double *local, double *whole) {
    for (int i = 0; i < nloc; ++i) whole[row_map.empty() ? (size_t)lo + i : (size_t)row_map[(size_t)i]] = local[i];
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
      const int c = device_min_eig(M, 1000, eta, 20, seed, device, &e);
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
    whole_of_local(res.v.data(), xarea_);  // every rank its own rows of the shared vector
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
    std::vector<double> vloc((size_t)std::max(nloc, 1));
    local_of_whole(vfull.data(), vloc.data());
    if (nloc > 0) DCORA_HIP(hipMemcpyAsync(vl.p, vloc.data(), sizeof(double) * nloc, hipMemcpyHostToDevice, st));
    rc = L.op(vl.p, wl.p);
    if (rc) return rc;
    std::vector<double> wh((size_t)std::max(nloc, 1));
    if (nloc > 0) DCORA_HIP(hipMemcpyAsync(wh.data(), wl.p, sizeof(double) * nloc, hipMemcpyDeviceToHost, st));
    DCORA_HIP(hipStreamSynchronize(st));
    double dots[2] = {0, 0};
    for (int i = 0; i < nloc; ++i) {
      dots[0] += vloc[(size_t)i] * wh[(size_t)i];
      dots[1] += vloc[(size_t)i] * vloc[(size_t)i];
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
