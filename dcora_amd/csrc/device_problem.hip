// DeviceProblem: QuadraticProblem / QuadraticOptimizer on the MI355X (see device_problem.h).
#include <atomic>

#include "device_problem.h"
#include "env.h"
#include "device_chol.h"
#include "precond_cache.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <thread>

namespace dcora {

static thread_local std::string g_last_error;
void set_last_error(const std::string &s) { g_last_error = s; }
const std::string &get_last_error() { return g_last_error; }
int hip_fail(hipError_t e, const char *what, const char *file, int line) {
  char buf[512];
  std::snprintf(buf, sizeof buf, "HIP error %d (%s) at %s:%d in %s", (int)e, hipGetErrorString(e), file, line, what);
  set_last_error(buf);
  return (e == hipErrorNoDevice || e == hipErrorInvalidDevice) ? DCORA_ERR_NO_DEVICE : DCORA_ERR_HIP;
}

int DevCsr::upload(int nr, int nc, const int *rph, const int *cih, const double *vh) {
  nrows = nr;
  ncols = nc;
  nnz = rph[nr];
  DCORA_HIP(rp.alloc(nr + 1));
  DCORA_HIP(ci.alloc(std::max(nnz, 1)));
  DCORA_HIP(v.alloc(std::max(nnz, 1)));
  DCORA_HIP(hipMemcpy(rp.p, rph, sizeof(int) * (nr + 1), hipMemcpyHostToDevice));
  if (nnz) {
    DCORA_HIP(hipMemcpy(ci.p, cih, sizeof(int) * nnz, hipMemcpyHostToDevice));
    DCORA_HIP(hipMemcpy(v.p, vh, sizeof(double) * nnz, hipMemcpyHostToDevice));
  }
  std::vector<int> lr;
  for (int i = 0; i < nr; ++i)
    if (rph[i + 1] - rph[i] > kLongRow) lr.push_back(i);
  n_long = (int)lr.size();
  if (n_long > kMaxPartials / 2) {  // not a hub structure any more: let the row-parallel path handle everything
    n_long = 0;
    lr.clear();
  }
  long_host = lr;
  DCORA_HIP(long_rows.alloc(std::max(n_long, 1)));
  if (n_long) DCORA_HIP(hipMemcpy(long_rows.p, lr.data(), sizeof(int) * n_long, hipMemcpyHostToDevice));
  DCORA_HIP(long_part.alloc((size_t)std::max(n_long, 1) * kLongSplit * 16));
  DCORA_HIP(long_cnt.alloc(std::max(n_long, 1)));
  DCORA_HIP(hipMemset(long_cnt.p, 0, sizeof(int) * std::max(n_long, 1)));
  return DCORA_OK;
}
int DevBsr::upload(const HostBsr &B) {
  nbrows = B.nbrows;
  nblocks = B.nblocks();
  DCORA_HIP(bp.alloc(nbrows + 1));
  DCORA_HIP(bc.alloc(std::max(nblocks, 1)));
  DCORA_HIP(bv.alloc(std::max<size_t>(B.bv.size(), 1)));
  DCORA_HIP(hipMemcpy(bp.p, B.bp.data(), sizeof(int) * (nbrows + 1), hipMemcpyHostToDevice));
  if (nblocks) {
    DCORA_HIP(hipMemcpy(bc.p, B.bc.data(), sizeof(int) * nblocks, hipMemcpyHostToDevice));
    DCORA_HIP(hipMemcpy(bv.p, B.bv.data(), sizeof(double) * B.bv.size(), hipMemcpyHostToDevice));
  }
  return DCORA_OK;
}
int DevCsr::upload(const HostCsr &A) { return upload(A.n, A.ncols, A.rp.data(), A.ci.data(), A.v.data()); }

// HostFlags words live in one host-mapped page per process, handed out slot by slot: hipHostMalloc per problem cost
// more than the rest of a cached creation
namespace {
constexpr int kFlagSlots = 1024;
constexpr size_t kFlagStride = 64;  // one line per problem: the words are written over PCIe by different streams
static_assert(sizeof(HostFlags) <= kFlagStride, "HostFlags slot");
std::mutex g_flag_mu;
HostFlags *g_flag_host = nullptr, *g_flag_dev = nullptr;
std::vector<int> g_flag_free;
}  // namespace
// HIP streams are recycled too: hipStreamCreateWithFlags took 1.7-8.5 ms of a 2-9 ms cached problem creation
// (DCORA_INIT_TIMING), hipStreamDestroy another 1-2 ms.  Idle non-blocking streams are kept per device.
namespace {
std::mutex g_stream_mu;
std::vector<std::pair<int, hipStream_t>> g_idle_streams;
}  // namespace
int stream_acquire(int device, hipStream_t *out) {
  {
    std::lock_guard<std::mutex> lk(g_stream_mu);
    for (size_t i = 0; i < g_idle_streams.size(); ++i)
      if (g_idle_streams[i].first == device) {
        *out = g_idle_streams[i].second;
        g_idle_streams.erase(g_idle_streams.begin() + (long)i);
        return DCORA_OK;
      }
  }
  DCORA_HIP(hipStreamCreateWithFlags(out, hipStreamNonBlocking));
  return DCORA_OK;
}
void stream_release(int device, hipStream_t st) {
  if (!st) return;
  (void)hipStreamSynchronize(st);
  std::lock_guard<std::mutex> lk(g_stream_mu);
  if (g_idle_streams.size() < 256) {
    g_idle_streams.emplace_back(device, st);
    return;
  }
  (void)hipStreamDestroy(st);
}

int host_flags_acquire(HostFlags **host, HostFlags **dev, int *slot) {
  std::lock_guard<std::mutex> lk(g_flag_mu);
  if (!g_flag_host) {
    DCORA_HIP(hipHostMalloc((void **)&g_flag_host, kFlagStride * kFlagSlots, hipHostMallocMapped | hipHostMallocPortable));
    std::memset((void *)g_flag_host, 0, kFlagStride * kFlagSlots);
    DCORA_HIP(hipHostGetDevicePointer((void **)&g_flag_dev, (void *)g_flag_host, 0));
    for (int i = kFlagSlots - 1; i >= 0; --i) g_flag_free.push_back(i);
  }
  if (g_flag_free.empty()) {  // more live problems than slots: a page of its own
    DCORA_HIP(hipHostMalloc((void **)host, sizeof(HostFlags), hipHostMallocMapped | hipHostMallocPortable));
    std::memset((void *)*host, 0, sizeof(HostFlags));
    DCORA_HIP(hipHostGetDevicePointer((void **)dev, (void *)*host, 0));
    *slot = -1;
    return DCORA_OK;
  }
  *slot = g_flag_free.back();
  g_flag_free.pop_back();
  *host = (HostFlags *)((char *)g_flag_host + kFlagStride * (size_t)*slot);
  *dev = (HostFlags *)((char *)g_flag_dev + kFlagStride * (size_t)*slot);
  std::memset((void *)*host, 0, sizeof(HostFlags));
  return DCORA_OK;
}
void host_flags_release(HostFlags *host, int slot) {
  if (!host) return;
  if (slot < 0) {
    (void)hipHostFree((void *)host);
    return;
  }
  std::lock_guard<std::mutex> lk(g_flag_mu);
  g_flag_free.push_back(slot);
}

DeviceProblem::~DeviceProblem() {
  if (st) (void)hipStreamSynchronize(st);  // nothing of this problem is in flight when its flag words are recycled
  for (hipEvent_t e : run_events) (void)hipEventDestroy(e);
  host_flags_release(hf, hf_slot);
  if (own_stream && st) stream_release(device, st);
}

int DeviceProblem::upload(const double *h, double *d, size_t n) {
  DCORA_HIP(hipMemcpyAsync(d, h, n * sizeof(double), hipMemcpyHostToDevice, st));
  return DCORA_OK;
}
int DeviceProblem::download(const double *d, double *h, size_t n) {
  DCORA_HIP(hipMemcpyAsync(h, d, n * sizeof(double), hipMemcpyDeviceToHost, st));
  DCORA_HIP(hipStreamSynchronize(st));
  return DCORA_OK;
}

int DeviceProblem::init(const dcora_dims &dims, const HostCsr &Qh, const double *Gh, double reg, int device_,
                        hipStream_t shared) {
  if (dims.r < 1 || dims.r > 16 || (dims.d != 2 && dims.d != 3) || dims.n < 0 || dims.l < 0 || dims.b < 0) {
    set_last_error("bad dims (need 1 <= r <= 16, d in {2,3})");
    return DCORA_ERR_BAD_ARG;
  }
  const bool init_timing = env::init_timing();
  const auto ti0 = std::chrono::steady_clock::now();
  auto lap = [&](const char *what) {
    if (init_timing)
      std::fprintf(stderr, "[init] %-12s %.3f ms\n", what,
                   std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - ti0).count());
  };
  m = make_mani(dims);
  if (Qh.n != m.k) {
    set_last_error("Q dimension does not match (d+1) n + l + b");
    return DCORA_ERR_BAD_ARG;
  }
  device = device_;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0) {
    set_last_error("no HIP device available: libdcora_hip has no CPU fallback");
    return DCORA_ERR_NO_DEVICE;
  }
  DCORA_HIP(hipSetDevice(device));
  if (shared) {
    st = shared;
    own_stream = false;
  } else {
    const int rcs = stream_acquire(device, &st);
    if (rcs) return rcs;
    own_stream = true;
  }
  lap("stream");
  int rc = Q.upload(Qh);
  if (rc) return rc;
  lap("Q upload");
  // block form of Q for graphs large enough to be bandwidth-bound (the scalar-CSR kernel exposes more parallelism
  // and wins while the launch is latency-bound)
  const bool want_bsr = m.n >= 8192;
  if (m.se && m.r <= 8 && m.n > 0 && want_bsr && !env::generic_solver()) {
    rc = Qb.upload(bsr_from_csr(Qh, m.d + 1));
    if (rc) return rc;
    has_bsr = true;
  }
  const size_t N = (size_t)nelem();
  const size_t NS = (size_t)m.n * m.d * m.d + m.l + 1;
  // (the fused Hessian kernel stages the first matrix tile with clamped, unconditional loads: it needs nnz > 0)
  fused = fused_supported(m) && Qh.nnz() > 0 && !env::generic_solver();
  group = group_supported(m) && !env::generic_solver();
  // One allocation for the whole solver workspace, zeroed by one memset: the reference re-creates its problem on
  // every Agent::updateX (ref src/Agent.cpp:1252), so creation must cost a fraction of a solve -- thirty hipMalloc
  // and ten synchronous hipMemset calls took 4-8 ms.
  {
    struct Slot {
      DevBuf<double> *b;
      size_t n;
    };
    std::vector<Slot> slots;
    for (DevBuf<double> *b : {&G, &X0, &X1, &EG0, &EG1, &RG0, &RG1, &delta, &eta, &Heta, &res, &z, &Hd, &W, &Zt})
      slots.push_back({b, N});
    slots.push_back({&delta2, N});
    if (fused) {
      slots.push_back({&res2, N});
      slots.push_back({&Zpart, (size_t)fused_nsplit(m) * N});  // slice 0 doubles as Z of the sparse preconditioner
    }
    slots.push_back({&S0, NS});
    slots.push_back({&S1, NS});
    // 2 doubles per slot; the Q-apply may add up to kMaxPartials / 2 long-row blocks to its kMaxPartials row blocks
    for (DevBuf<double> *b : {&pB, &pC, &p1, &p2, &p3}) slots.push_back({b, (size_t)4 * kMaxPartials});
    slots.push_back({&pA, 2 * (size_t)kBsrMaxGrid});  // the block-CSR Q-apply runs up to kBsrMaxGrid workgroups
    slots.push_back({&scal, 64});
    auto pad = [](size_t n) { return (n + 31) & ~(size_t)31; };  // 256-byte aligned slices
    size_t total = 0;
    for (const Slot &sl : slots) total += pad(sl.n);
    DCORA_HIP(arena.alloc(total));
    DCORA_HIP(hipMemsetAsync(arena.p, 0, total * sizeof(double), st));
    size_t off = 0;
    for (const Slot &sl : slots) {
      sl.b->borrow(arena.p + off, sl.n);
      off += pad(sl.n);
    }
  }
  DCORA_HIP(ctl.alloc(1));
  {
    const int rc2 = host_flags_acquire(&hf, &hf_dev, &hf_slot);
    if (rc2) return rc2;
  }
  DCORA_HIP(hipStreamSynchronize(st));  // the zeroed workspace is visible to every stream before the first upload
  lap("workspace");
  if (Gh) {
    rc = set_G_host(Gh);
    if (rc) return rc;
  }
  if (reg >= 0) {
    rc = build_preconditioner(Qh, reg);
    if (rc) return rc;
  }
  lap("precond");
  // the one-launch tCG run: dense inverse, the whole residual in one LDS image, n / 2 workgroups co-resident
  if (fused && has_precond && !sparse_precond && !has_bsr && Q.n_long == 0 && env::solver_tcg() >= 0) {
    hipDeviceProp_t prop;
    DCORA_HIP(hipGetDeviceProperties(&prop, device));
    tcg_run_ok = tcg_run_supported(m, ldm, prop.multiProcessorCount, tcg_run_max_rows_nnz(m, Qh.rp.data()));
    if (tcg_run_ok) {
      DCORA_HIP(tcg_sync.alloc((size_t)tcg_run_sync_words()));
      DCORA_HIP(hipMemsetAsync(tcg_sync.p, 0, sizeof(unsigned) * (size_t)tcg_run_sync_words(), st));
    }
  }
  return DCORA_OK;
}

int DeviceProblem::set_G_host(const double *Gh) {
  DCORA_HIP(hipSetDevice(device));
  if (Gh) {
    DCORA_HIP(hipMemcpy(G.p, Gh, (size_t)nelem() * sizeof(double), hipMemcpyHostToDevice));
    has_G = true;
  } else {
    DCORA_HIP(hipMemset(G.p, 0, (size_t)nelem() * sizeof(double)));
    has_G = false;
  }
  return DCORA_OK;
}

// The dense inverses of SEVERAL matrices (the agents of a session) built in one batch of launches and put into the cache,
// so that the problems created next attach to them: matrices of equal size whose inverse is not cached yet and that would
// take the dense form.  Best effort: whatever is not built here is built by the problem that needs it.
int precond_prebuild_dense(const std::vector<const HostCsr *> &Qs, double reg, int block, int device) {
  if (env::precond_mode() == 2 || env::factor_on_host() || env::precond_cache_mb() <= 0) return DCORA_OK;
  std::map<int, std::vector<size_t>> by_k;
  std::vector<PrecondKey> keys(Qs.size());
  for (size_t i = 0; i < Qs.size(); ++i) {
    const int k = Qs[i]->n;
    const bool want_sparse = env::precond_mode() ? false : (k > kDensePrecondMaxK);
    if (want_sparse || k < 1) continue;
    keys[i] = make_precond_key(*Qs[i], reg, block, device, false);
    PrecondEntry ent;
    if (precond_cache_find(keys[i], &ent, false)) continue;
    bool dup = false;  // (two agents with the same matrix: one build)
    for (size_t j : by_k[k]) dup = dup || keys[j] == keys[i];
    if (!dup) by_k[k].push_back(i);
  }
  DCORA_HIP(hipSetDevice(device));
  for (auto &grp : by_k) {
    const std::vector<size_t> &idx = grp.second;
    if (idx.size() < 2) continue;
    const auto t0 = std::chrono::steady_clock::now();
    const int k = grp.first, ldm = ((k + 127) / 128) * 128;
    std::vector<HostCsr> Ms(idx.size());
    std::vector<const HostCsr *> As;
    std::vector<std::shared_ptr<DevBuf<double>>> bufs;
    std::vector<double *> outs;
    std::vector<long> nnzL(idx.size(), 0);
    // host work per matrix (the shifted copy; nnz(L) of the SPARSE factor from the pattern, what SURVEY 8(d) prices the
    // preconditioner by) on threads of its own, beside the device's batch
    std::vector<std::thread> th;
    for (size_t q = 0; q < idx.size(); ++q) Ms[q] = csr_shift_diag(*Qs[idx[q]], reg);
    for (size_t q = 0; q < idx.size(); ++q)
      th.emplace_back([&, q] {
        CholSymbolic sym;
        chol_symbolic(Ms[q], block, &sym);
        long nz = 0;
        for (const CholPiece &pc : sym.pieces) nz += (long)pc.c * (pc.c + 1) / 2 + (long)pc.m * pc.c;
        nnzL[q] = nz;
      });
    struct Join {
      std::vector<std::thread> &t;
      ~Join() {
        for (std::thread &x : t)
          if (x.joinable()) x.join();
      }
    } join_guard{th};
    for (size_t q = 0; q < idx.size(); ++q) {
      auto buf = std::make_shared<DevBuf<double>>();
      DCORA_HIP(buf->alloc((size_t)k * ldm + 16));
      bufs.push_back(buf);
      outs.push_back(buf->p);
    }
    for (const HostCsr &M : Ms) As.push_back(&M);
    std::vector<char> pd;
    const int rc = device_dense_spd_inverse_batch(As, device, outs, ldm, &pd);
    if (rc) return rc;
    for (std::thread &x : th) x.join();
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    for (size_t q = 0; q < idx.size(); ++q) {
      if (!pd[q]) continue;  // not positive definite somewhere in the batch: the problems build (and report) one by one
      PrecondEntry ent;
      ent.ldm = ldm;
      ent.dense = bufs[q];
      ent.bytes = bufs[q]->n * sizeof(double);
      ent.build_ms = ms / (double)idx.size();
      ent.nnzL = nnzL[q];
      ent.prebuilt = true;
      precond_cache_insert(keys[idx[q]], ent);
    }
  }
  return DCORA_OK;
}

// (Q + reg I)^-1 as a dense symmetric matrix in HBM: host sparse Cholesky once per Q (Q is reused across all
// RBCD iterations and staircase levels), k independent solves on host threads, one upload.
int DeviceProblem::build_preconditioner(const HostCsr &Qh, double reg) {
  const auto t0 = std::chrono::steady_clock::now();
  const int k = m.k;
  // host threads of the set-up: what the process may really use (host_cpus_available), shared between the problems
  // that are being set up at the same time (the agents of a session are created side by side)
  static std::atomic<int> builders{0};
  struct Builder {
    std::atomic<int> &c;
    int active;
    explicit Builder(std::atomic<int> &c_) : c(c_), active(c_.fetch_add(1) + 1) {}
    ~Builder() { c.fetch_sub(1); }
  } builder(builders);
  const int nthreads = std::max(2, host_cpus_available() / std::max(1, builder.active));
  // Large blocks: partitioned sparse inverse replayed level by level (sparse_precond.h).  Small blocks: the dense
  // inverse streams faster than 2 * depth + 2 dependent launches.  DCORA_PRECOND=dense|sparse overrides.
  const bool want_sparse = env::precond_mode() ? env::precond_mode() == 2 : (k > kDensePrecondMaxK);
  const int block = m.se ? m.d + 1 : 1;
  if (want_sparse && m.r > 16) {
    set_last_error("sparse preconditioner supports r <= 16");
    return DCORA_ERR_UNSUPPORTED;
  }
  if (!want_sparse && (size_t)k > 60000) {
    set_last_error("dense preconditioner limited to k <= 60000");
    return DCORA_ERR_UNSUPPORTED;
  }
  // the inverse depends on Q + reg I only (not on r, not on G): built once per distinct matrix, shared afterwards
  const PrecondKey key = make_precond_key(Qh, reg, block, device, want_sparse);
  PrecondEntry ent;
  const bool found = precond_cache_find(key, &ent);
  precond_cache_hit = found && !ent.prebuilt;  // (an image built ahead for THIS problem is not a hit)
  DCORA_HIP(hipSetDevice(device));
  if (!found) {
    if (env::init_timing())
      fprintf(stderr, "[precond] key + cache look-up %.1f ms\n",
              std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    HostCsr M = csr_shift_diag(Qh, reg);
    if (want_sparse) {
      PartInvHost P;
      // the stored weights are formed on the device, or stream there while the host's threads form them
      DeviceWeightSink sink(device);
      P.sink = &sink;
      const int brc = build_partitioned_inverse_auto(M, block, nthreads, device, &P);
      if (brc && brc != DCORA_ERR_NOT_PD) return brc;
      const bool ok = brc == DCORA_OK;
      if (!ok) {
        set_last_error("preconditioner: Q + reg I is not positive definite");
        return DCORA_ERR_NOT_PD;
      }
      auto img = std::make_shared<SpImage>();
      const auto tu = std::chrono::steady_clock::now();
      const int rc = img->upload(P, P.sink ? &sink : nullptr);
      if (rc) return rc;
      if (env::init_timing())
        fprintf(stderr, "[precond] image upload %.1f ms (since start %.1f ms)\n",
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tu).count(),
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
      if (env::init_timing())
        fprintf(stderr, "[precond] host image: %.1f MB wave records, %.1f MB indices, %.1f MB weights\n",
                P.mwaves.size() * sizeof(MWave) / 1e6, P.idxs.size() * 4 / 1e6, P.vals.size() * 8 / 1e6);
      ent.sparse = img;
      ent.nnzL = P.nnzL;
      ent.bytes = img->device_bytes();
    } else {
      // rows padded so every 16-byte column-pair load of a 128-column chunk is in bounds
      ent.ldm = ((k + 127) / 128) * 128;
      auto buf = std::make_shared<DevBuf<double>>();
      DCORA_HIP(buf->alloc((size_t)k * ent.ldm + 16));
      const bool host_factor = env::factor_on_host();
      if (host_factor) {  // A/B measurements: sparse Cholesky and k solves on host threads, one upload
        SparseChol chol;
        if (!chol.factor(M, block)) {
          set_last_error("preconditioner: Q + reg I is not positive definite");
          return DCORA_ERR_NOT_PD;
        }
        ent.nnzL = chol.nnzL();
        std::vector<double> inv((size_t)k * ent.ldm, 0.0);
        chol.dense_inverse(inv.data(), (size_t)ent.ldm, nthreads);
        DCORA_HIP(hipMemcpy(buf->p, inv.data(), (size_t)k * ent.ldm * sizeof(double), hipMemcpyHostToDevice));
      } else {  // dense LL^T, L^-1 and L^-T L^-1 on the device (device_chol.h)
        bool pd = false;
        const int rc = device_dense_spd_inverse(M, device, buf->p, ent.ldm, &pd);
        if (rc) return rc;
        if (!pd) {
          set_last_error("preconditioner: Q + reg I is not positive definite");
          return DCORA_ERR_NOT_PD;
        }
        // nnz(L) of the SPARSE factor of this matrix (what a triangular solve would stream: the quantity SURVEY 8(d)
        // prices the preconditioner by), from the pattern alone -- the dense factor that was formed has k (k + 1) / 2
        CholSymbolic sym;
        chol_symbolic(M, block, &sym);
        long nz = 0;
        for (const CholPiece &pc : sym.pieces) nz += (long)pc.c * (pc.c + 1) / 2 + (long)pc.m * pc.c;
        ent.nnzL = nz;
      }
      ent.dense = buf;
      ent.bytes = buf->n * sizeof(double);
    }
    ent.build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (env::init_timing()) fprintf(stderr, "[precond] built after %.1f ms\n", ent.build_ms);
    precond_cache_insert(key, ent);
  }
  if (env::init_timing())
    fprintf(stderr, "[precond] cached after %.1f ms\n",
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
  precond_nnzL = ent.nnzL;
  if (want_sparse) {
    const int rc = sp.attach(ent.sparse, m.r);
    if (rc) return rc;
    sparse_precond = true;
  } else {
    ldm = ent.ldm;
    Minv_shared = ent.dense;
    Minv.borrow(ent.dense->p, ent.dense->n);
  }
  has_precond = true;
  precond_setup_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  if (env::init_timing()) fprintf(stderr, "[precond] attached after %.1f ms\n", precond_setup_ms);
  return DCORA_OK;
}

int DeviceProblem::enq_qapply(Buf2 X, int selX, const double *Gp, Buf2 Y, int selY, double *partials, Gate g) {
  if (has_bsr) {
    launch_spmm_bsr(st, m.r, m.d, Qb.view(), X, selX, Gp, Y, selY, partials, g);
    return spmm_bsr_grid(m.n);
  }
  launch_spmm(st, m.r, Q.view(), X, selX, Gp, Y, selY, partials, g);
  return spmm_grid(m.k, m.r) + Q.n_long;
}
void DeviceProblem::enqueue_egrad(const double *X, double *EG, double *partials) {
  enq_qapply(buf1(X), 0, has_G ? G.p : nullptr, buf1(EG), 0, partials, Gate{});
}

int DeviceProblem::enq_rgrad(Buf2 X, Buf2 EG, Buf2 RG, Buf2 S, int sel, double *partials, Gate g,
                             double *posenorm) {
  if (group) return launch_g_rgrad(st, m, X, EG, RG, S, sel, partials, posenorm, g);
  launch_rgrad(st, m, X, EG, RG, S, sel, partials, g);
  return pose_grid(m);
}
int DeviceProblem::enq_retract(Buf2 X, const double *V, double alpha, Buf2 out, int selOut, Buf2 grad,
                               const double *HV, double *partials, Gate g) {
  if (group) return launch_g_retract(st, m, X, V, alpha, out, selOut, grad, HV, partials, g);
  launch_retract(st, m, X, V, alpha, out, selOut, grad, HV, partials, g);
  return pose_grid(m);
}

void DeviceProblem::enq_minv(Buf2 R, double *Z, const double *p2, int np2, Gate g) {
  if (sparse_precond)
    sp.apply(st, m.r, R, Z, g);
  else
    launch_dense_apply(st, m.r, m.k, ldm, Minv.p, R, Z, p2, np2, g);
}
void DeviceProblem::enqueue_precond(const double *X, const double *V, double *out) {
  enq_minv(buf1(V), Zt.p, nullptr, 0, Gate{});
  launch_tangent(st, m, buf1(X), Zt.p, out, nullptr, nullptr, nullptr, 0, nullptr, nullptr, 0, 0, 0);
}

// ---------------------------------------------------------------------------------------------------------
// host-pointer API
// ---------------------------------------------------------------------------------------------------------
int DeviceProblem::cost(const double *Xh, double *f) {
  DCORA_HIP(hipSetDevice(device));
  int rc = upload(Xh, X0.p, nelem());
  if (rc) return rc;
  enqueue_egrad(X0.p, EG0.p, pA.p);
  launch_sum_partials(st, pA.p, npA(), 2, 2, scal.p);
  double s[2];
  rc = download(scal.p, s, 2);
  if (rc) return rc;
  *f = 0.5 * s[0] + s[1];
  return DCORA_OK;
}
int DeviceProblem::eucgrad(const double *Xh, double *out) {
  DCORA_HIP(hipSetDevice(device));
  int rc = upload(Xh, X0.p, nelem());
  if (rc) return rc;
  enqueue_egrad(X0.p, EG0.p, nullptr);
  return download(EG0.p, out, nelem());
}
int DeviceProblem::riegrad(const double *Xh, double *out, double *norm) {
  DCORA_HIP(hipSetDevice(device));
  int rc = upload(Xh, X0.p, nelem());
  if (rc) return rc;
  enqueue_egrad(X0.p, EG0.p, nullptr);
  launch_rgrad(st, m, buf1(X0.p), buf1(EG0.p), buf1(RG0.p), buf1(S0.p), 0, pB.p, Gate{});
  launch_sum_partials(st, pB.p, npPose(), 1, 1, scal.p);
  double s;
  rc = download(scal.p, &s, 1);
  if (rc) return rc;
  if (norm) *norm = std::sqrt(s);
  if (out) return download(RG0.p, out, nelem());
  return DCORA_OK;
}
int DeviceProblem::hessvec(const double *Xh, const double *Vh, double *out) {
  DCORA_HIP(hipSetDevice(device));
  int rc = upload(Xh, X0.p, nelem());
  if (rc) return rc;
  rc = upload(Vh, delta.p, nelem());
  if (rc) return rc;
  enqueue_egrad(X0.p, EG0.p, nullptr);
  launch_rgrad(st, m, buf1(X0.p), buf1(EG0.p), buf1(RG0.p), buf1(S0.p), 0, pB.p, Gate{});
  launch_spmm(st, m.r, Q.view(), buf1(delta.p), 0, nullptr, buf1(W.p), 0, nullptr, Gate{});
  launch_hessfix(st, m, buf1(X0.p), buf1(S0.p), delta.p, W.p, Hd.p, p1.p, Gate{});
  return download(Hd.p, out, nelem());
}
bool DeviceProblem::hess_one_launch() const {
  if (spmm_dir_fix_grid(m, m.k) <= 0 || spmm_dir_fix_grid(m, m.k) + Q.n_long > 4 * kMaxPartials) return false;
  for (int jl : Q.long_host) {
    const bool euclid = m.se ? (jl % (m.d + 1) == m.d) : (jl >= m.d * m.n + m.l);
    if (!euclid) return false;
  }
  return true;
}
int DeviceProblem::hessvec_solver_form(const double *Xh, const double *Vh, double *out, double *dots) {
  if (!hess_one_launch()) {
    set_last_error("hessvec_solver_form: the one-launch form does not apply to this problem");
    return DCORA_ERR_UNSUPPORTED;
  }
  DCORA_HIP(hipSetDevice(device));
  int rc = upload(Xh, X0.p, nelem());
  if (rc) return rc;
  std::vector<double> neg(Vh, Vh + nelem());
  for (double &v : neg) v = -v;
  rc = upload(neg.data(), z.p, nelem());  // iteration 0 of the loop: delta = -z
  if (rc) return rc;
  rc = upload(Vh, delta2.p, nelem());
  if (rc) return rc;
  enqueue_egrad(X0.p, EG0.p, nullptr);
  launch_rgrad(st, m, buf1(X0.p), buf1(EG0.p), buf1(RG0.p), buf1(S0.p), 0, pB.p, Gate{});
  DCORA_HIP(hipMemsetAsync(p3.p, 0, sizeof(double), st));
  const int np = launch_spmm_dir_fix(st, m, Q.view(), buf1(X0.p), buf1(S0.p), z.p, delta2.p, delta.p, Hd.p, p3.p, 1, p1.p,
                                     ctl.p, 0, 0);
  std::vector<double> part((size_t)np);
  rc = download(p1.p, part.data(), part.size());
  if (rc) return rc;
  dots[0] = 0;
  for (double v : part) dots[0] += v;
  rc = download(Hd.p, out, nelem());
  if (rc) return rc;
  launch_spmm(st, m.r, Q.view(), buf1(delta.p), 0, nullptr, buf1(W.p), 0, nullptr, Gate{});
  launch_hessfix(st, m, buf1(X0.p), buf1(S0.p), delta.p, W.p, Hd.p, p1.p, Gate{});
  part.assign((size_t)npPose(), 0.0);
  rc = download(p1.p, part.data(), part.size());
  if (rc) return rc;
  dots[1] = 0;
  for (double v : part) dots[1] += v;
  return DCORA_OK;
}
int DeviceProblem::precondition(const double *Xh, const double *Vh, double *out) {
  if (!has_precond) {
    set_last_error("problem has no preconditioner");
    return DCORA_ERR_NO_PRECONDITIONER;
  }
  DCORA_HIP(hipSetDevice(device));
  int rc = upload(Xh, X0.p, nelem());
  if (rc) return rc;
  rc = upload(Vh, res.p, nelem());
  if (rc) return rc;
  enqueue_precond(X0.p, res.p, z.p);
  return download(z.p, out, nelem());
}
int DeviceProblem::retract(const double *Xh, const double *Vh, double *out) {
  DCORA_HIP(hipSetDevice(device));
  int rc = upload(Xh, X0.p, nelem());
  if (rc) return rc;
  rc = upload(Vh, eta.p, nelem());
  if (rc) return rc;
  launch_retract(st, m, buf1(X0.p), eta.p, 1.0, buf1(X1.p), 0, buf1(RG0.p), nullptr, nullptr, Gate{});
  return download(X1.p, out, nelem());
}
int DeviceProblem::tangent_project(const double *Xh, const double *Vh, double *out) {
  DCORA_HIP(hipSetDevice(device));
  int rc = upload(Xh, X0.p, nelem());
  if (rc) return rc;
  rc = upload(Vh, eta.p, nelem());
  if (rc) return rc;
  launch_tangent(st, m, buf1(X0.p), eta.p, z.p, nullptr, nullptr, nullptr, 0, nullptr, nullptr, 0, 0, 0);
  return download(z.p, out, nelem());
}

int DeviceProblem::eval_dev(const double *Xd, double *f, double *gradnorm) {
  enqueue_egrad(Xd, EG1.p, pA.p);
  launch_rgrad(st, m, buf1(Xd), buf1(EG1.p), Buf2{{nullptr, nullptr}}, Buf2{{nullptr, nullptr}}, 0, pB.p, Gate{});
  launch_sum_partials(st, pA.p, npA(), 2, 2, scal.p);
  launch_sum_partials(st, pB.p, npPose(), 1, 1, scal.p + 2);
  double s[3];
  int rc = download(scal.p, s, 3);
  if (rc) return rc;
  if (f) *f = 0.5 * s[0] + s[1];
  if (gradnorm) *gradnorm = std::sqrt(s[2]);
  return DCORA_OK;
}

// ---------------------------------------------------------------------------------------------------------
// QuadraticOptimizer::optimize
// ---------------------------------------------------------------------------------------------------------
int DeviceProblem::optimize(const dcora_ropt_params &prm, const double *X0h, double *Xout, dcora_ropt_result *res_out) {
  DCORA_HIP(hipSetDevice(device));
  int rc = upload(X0h, X0.p, nelem());
  if (rc) return rc;
  Buf2 Xres{{nullptr, nullptr}};
  const SolverCtl *c = nullptr;
  rc = optimize_dev(prm, &Xres, &c);
  if (rc) return rc;
  dcora_ropt_result r{};
  rc = fetch_result(&r);  // synchronises; also resolves which buffer holds the accepted iterate
  if (rc) return rc;
  if (res_out) *res_out = r;
  return download(c ? (cur_after_fetch_ ? X1.p : X0.p) : Xres.p[0], Xout, nelem());
}

int DeviceProblem::optimize_dev(const dcora_ropt_params &prm, Buf2 *Xres, const SolverCtl **ctl_out) {
  *ctl_out = nullptr;
  pending_ = false;
  if (prm.method == 0 && fused) {
    if (!has_precond) {
      set_last_error("RTR requires the preconditioner (ref src/QuadraticProblem.cpp:78-82)");
      return DCORA_ERR_NO_PRECONDITIONER;
    }
    const int rc = rtr_dev_fused(prm);
    if (rc) return rc;
    *Xres = Xb();
    *ctl_out = ctl.p;
    pending_ = true;
    return DCORA_OK;
  }
  double *p = nullptr;
  const int rc = (prm.method == 0) ? rtr_dev(prm, &last_res_, &p) : rgd_dev(prm, &last_res_, &p);
  *Xres = Buf2{{p, p}};
  return rc;
}

int DeviceProblem::fetch_result(dcora_ropt_result *res_out) {
  if (pending_) {
    SolverCtl h;
    DCORA_HIP(hipMemcpyAsync(&h, ctl.p, sizeof h, hipMemcpyDeviceToHost, st));
    DCORA_HIP(hipStreamSynchronize(st));
    DCORA_HIP(hipGetLastError());
    const double now = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch())
                           .count();
    cur_after_fetch_ = h.cur & 1;
    last_res_.success = 1;
    last_res_.fInit = h.fInit;
    last_res_.gradNormInit = h.gradNormInit;
    last_res_.fOpt = h.f1;
    last_res_.gradNormOpt = h.ngf;
    last_res_.elapsedMs = now - t0_ms_;
    last_res_.tCGStatus = h.tcg_status;
    last_res_.outer_iterations = h.outer_it;
    last_res_.inner_iterations = h.inner_total;
    last_res_.accepted_steps = h.accepted;
    pending_ = false;
  }
  if (res_out) *res_out = last_res_;
  return DCORA_OK;
}

// one preconditioned Riemannian gradient step (ref src/QuadraticOptimizer.cpp:123-150)
int DeviceProblem::rgd_dev(const dcora_ropt_params &prm, dcora_ropt_result *res_out, double **Xres) {
  const auto t0 = std::chrono::steady_clock::now();
  double f0 = 0, g0 = 0;
  int rc = eval_dev(X0.p, &f0, &g0);
  if (rc) return rc;
  enqueue_egrad(X0.p, EG0.p, nullptr);
  launch_rgrad(st, m, buf1(X0.p), buf1(EG0.p), buf1(RG0.p), buf1(S0.p), 0, pB.p, Gate{});
  const double *dir = RG0.p;
  if (prm.RGD_use_preconditioner) {
    if (!has_precond) {
      set_last_error("RGD with preconditioning requested but the problem has none");
      return DCORA_ERR_NO_PRECONDITIONER;
    }
    enqueue_precond(X0.p, RG0.p, z.p);
    dir = z.p;
  }
  launch_retract(st, m, buf1(X0.p), dir, -prm.RGD_stepsize, buf1(X1.p), 0, buf1(RG0.p), nullptr, nullptr, Gate{});
  double f1 = 0, g1 = 0;
  rc = eval_dev(X1.p, &f1, &g1);
  if (rc) return rc;
  if (res_out) {
    res_out->success = 1;
    res_out->fInit = f0;
    res_out->gradNormInit = g0;
    res_out->fOpt = f1;
    res_out->gradNormOpt = g1;
    res_out->elapsedMs = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    res_out->tCGStatus = 4;
    res_out->outer_iterations = 1;
    res_out->inner_iterations = 0;
    res_out->accepted_steps = 1;
  }
  *Xres = X1.p;
  return DCORA_OK;
}

namespace {
// spin on host-mapped words; never blocks inside the HIP runtime
template <class Pred>
bool spin_until(Pred p, double timeout_s) {
  const auto t0 = std::chrono::steady_clock::now();
  unsigned it = 0;
  while (!p()) {
    if ((++it & 1023u) == 0) {
      if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) return false;
    }
#if defined(__x86_64__)
    __builtin_ia32_pause();
#endif
  }
  return true;
}
}  // namespace

// RTRNewton with preconditioned Steihaug-Toint tCG, fully device-resident: the host enqueues the kernel
// sequence, stays at most kLookahead tCG iterations ahead of the GPU and learns about terminations from
// host-mapped flags.  (ref src/QuadraticOptimizer.cpp:52-108, 234-280; ROPTLIB semantics: SURVEY.md 3.4)
int DeviceProblem::rtr_dev(const dcora_ropt_params &prm, dcora_ropt_result *res_out, double **Xres) {
  if (!has_precond) {
    set_last_error("RTR requires the preconditioner (ref src/QuadraticProblem.cpp:78-82)");
    return DCORA_ERR_NO_PRECONDITIONER;
  }
  constexpr int kLookahead = 2;
  const auto t0 = std::chrono::steady_clock::now();
  const bool single = (prm.RTR_iterations == 1);
  SolverCtl h;
  std::memset(&h, 0, sizeof h);
  h.tol = prm.gradnorm_tol;
  h.Delta = prm.RTR_initial_radius;
  h.maxDelta = single ? prm.RTR_initial_radius : 5 * prm.RTR_initial_radius;  // :240-241, :259-260
  h.max_outer = single ? 12 : prm.RTR_iterations;                              // :254-273 (<= 11 retries)
  h.stop_on_accept = single ? 1 : 0;
  h.max_inner = prm.RTR_tCG_iterations;
  h.outer_done_stamp = INT_MAX;
  h.tcg_done_stamp = INT_MAX;
  h.tcg_status = 4;
  DCORA_HIP(hipMemcpyAsync(ctl.p, &h, sizeof h, hipMemcpyHostToDevice, st));
  DCORA_HIP(hipStreamSynchronize(st));  // h lives on this stack frame
  hf->last_seq_done = 0;
  hf->tcg_done_seq = 0;
  hf->outer_done_seq = 0;
  hf->go_seq = 0;
  hf->reject_seq = 0;
  SolverCtl *c = ctl.p;
  const long N = nelem();
  const CsrDev Qv = Q.view();
  const double *Gp = has_G ? G.p : nullptr;
  const int nA = npA(), nP = npPose(), nV = npVec();
  int seq = 0;
  auto timed_out = [&]() {
    set_last_error("rtr_dev: device did not make progress (spin timeout)");
    return DCORA_ERR_HIP;
  };

  // f(x0), grad(x0)
  // (enq_qapply, not the CSR kernel directly: the number of partial slots npA() counts follows the Q-apply kernel that
  // runs -- block-CSR for large pose graphs)
  enq_qapply(Xb(), 0, Gp, EGb(), 0, pA.p, Gate{c, ++seq, 0});
  launch_rgrad(st, m, Xb(), EGb(), RGb(), Sb(), 0, pB.p, Gate{c, ++seq, 0});
  launch_rtr_init(st, pA.p, nA, pB.p, nP, c, hf_dev, ++seq);
  int last_pace_seq = seq;

  std::vector<int> upd2_seq((size_t)std::max(1, h.max_inner));
  const SpFold sfg = sparse_precond ? sp.fold_generic() : SpFold();
  // H d in one launch (k_spmm_dir_fix) when every long row is a Euclidean column and the partial slots fit
  const bool hess_fused = hess_one_launch();
  for (int outer = 0; outer < h.max_outer; ++outer) {
    // wait for the previous decision (rtr_init / rtr_decide) before committing to another outer iteration
    if (!spin_until([&] { return hf->last_seq_done >= last_pace_seq || hf->outer_done_seq != 0; }, 20.0))
      return timed_out();
    if (hf->outer_done_seq != 0) break;
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 5.0) break;  // TimeBound :252

    // ---- tCG ----
    launch_tcg_begin(st, N, RGb(), eta.p, Heta.p, res.p, c, ++seq);
    const int tcg_first_seq = seq;
    enq_minv(buf1(res.p), Zt.p, nullptr, 0, Gate{c, ++seq, 1});
    launch_tangent(st, m, Xb(), Zt.p, z.p, res.p, p3.p, nullptr, 0, c, hf_dev, ++seq, 1, 0);
    // the direction update (k_tcg_init, then k_tcg_update2 of the previous iteration) rides in the Hessian SpMM of
    // the next one (k_spmm_dir); the direction ping-pongs between two buffers
    double *const db[2] = {delta.p, delta2.p};
    for (int j = 0; j < h.max_inner; ++j) {
      if (j >= kLookahead) {
        const int need = upd2_seq[j - kLookahead];
        if (!spin_until(
                [&] {
                  return hf->last_seq_done >= need || hf->tcg_done_seq >= tcg_first_seq || hf->outer_done_seq != 0;
                },
                20.0))
          return timed_out();
      }
      if (hf->tcg_done_seq >= tcg_first_seq) break;
      double *const dcur = db[j & 1];
      int nP1 = nP;
      if (hess_fused) {
        nP1 = launch_spmm_dir_fix(st, m, Qv, Xb(), Sb(), z.p, db[(j + 1) & 1], dcur, Hd.p, p3.p, nP, p1.p, c, ++seq, j);
      } else {
        launch_spmm_dir(st, m.r, Qv, z.p, db[(j + 1) & 1], dcur, W.p, p3.p, nP, c, ++seq, j);
        launch_hessfix(st, m, Xb(), Sb(), dcur, W.p, Hd.p, p1.p, Gate{c, ++seq, 2});
      }
      // sparse preconditioner: its two permutations (and the hub correction) ride in the kernels either side of the
      // level replay -- two launches fewer per tCG iteration
      launch_tcg_update1(st, N, dcur, Hd.p, eta.p, Heta.p, res.p, p1.p, nP1, p2.p, c, hf_dev, ++seq, j, m.r, sfg);
      if (sfg.y) {
        // as in rtr_dev_fused: only the replay's first launch is enqueued before update1's verdict is known
        const int seqB = seq;
        bool timed = false;
        const std::function<bool()> verdict = [&]() {
          if (!spin_until(
                  [&] {
                    return hf->go_seq >= seqB || hf->tcg_done_seq >= tcg_first_seq || hf->outer_done_seq != 0;
                  },
                  20.0)) {
            timed = true;
            return false;
          }
          return !(hf->tcg_done_seq >= tcg_first_seq || hf->outer_done_seq != 0);
        };
        sp.apply(st, m.r, buf1(res.p), Zt.p, Gate{c, ++seq, 2}, true, &verdict);
        if (timed) return timed_out();
        if (hf->tcg_done_seq >= tcg_first_seq || hf->outer_done_seq != 0) break;
      } else
        enq_minv(buf1(res.p), Zt.p, p2.p, nV, Gate{c, ++seq, 2});
      launch_tangent(st, m, Xb(), Zt.p, z.p, res.p, p3.p, p2.p, nV, c, hf_dev, ++seq, 2, j, sfg);
      // the inner loop exhausted: the bookkeeping of the last direction update (status TR_MAXITER) has no next SpMM
      if (j == h.max_inner - 1) launch_tcg_update2(st, N, z.p, dcur, p3.p, nP, c, hf_dev, ++seq, j);
      upd2_seq[j] = seq;
    }
    // ---- trial point, model ratio, acceptance ----
    launch_retract(st, m, Xb(), eta.p, 1.0, Xb(), 1, RGb(), Heta.p, pC.p, Gate{c, ++seq, 1});
    enq_qapply(Xb(), 1, Gp, EGb(), 1, pA.p, Gate{c, ++seq, 1});
    launch_rgrad(st, m, Xb(), EGb(), RGb(), Sb(), 1, pB.p, Gate{c, ++seq, 1});
    launch_rtr_decide(st, pA.p, nA, pB.p, nP, pC.p, nP, c, hf_dev, ++seq);
    last_pace_seq = seq;
  }
  DCORA_HIP(hipMemcpyAsync(&h, ctl.p, sizeof h, hipMemcpyDeviceToHost, st));
  DCORA_HIP(hipStreamSynchronize(st));
  DCORA_HIP(hipGetLastError());
  *Xres = (h.cur & 1) ? X1.p : X0.p;
  if (res_out) {
    res_out->success = 1;
    res_out->fInit = h.fInit;
    res_out->gradNormInit = h.gradNormInit;
    res_out->fOpt = h.f1;
    res_out->gradNormOpt = h.ngf;
    res_out->elapsedMs = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    res_out->tCGStatus = h.tcg_status;
    res_out->outer_iterations = h.outer_it;
    res_out->inner_iterations = h.inner_total;
    res_out->accepted_steps = h.accepted;
  }
  return DCORA_OK;
}

// Same algorithm with the fused kernels of solver_fused.hip: per tCG iteration
//   A (direction update + Q-apply + Riemannian Hessian correction + <d,Hd>)
//   B (step length + vector updates + |r|^2 + dense preconditioner slices)
//   C (stopping rule + slice sum + tangent projection + <z,r>)
// B + C as one launch (k_fused_pc) where that form wins; DCORA_SOLVER_BC = pc / split forces one or the other
// sum of the HIP-event times of the k_tcg_run launches recorded since the last read (gated no-op launches included)
int DeviceProblem::profile_tcg_read(double *launches, double *total_us) {
  DCORA_HIP(hipSetDevice(device));
  DCORA_HIP(hipStreamSynchronize(st));
  double us = 0;
  for (size_t i = 0; i + 1 < run_events_used; i += 2) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, run_events[i], run_events[i + 1]) == hipSuccess) us += 1e3 * ms;
  }
  *launches = (double)(run_events_used / 2);
  *total_us = us;
  run_events_used = 0;
  return DCORA_OK;
}

bool DeviceProblem::use_pc() const {
  const int forced = env::solver_bc();
  if (sparse_precond || !has_precond) return false;
  if (forced < 0) return false;
  if (!forced && !fused_pc_preferred(m, ldm)) return false;
  // the one-launch form needs more dynamic LDS than the default limit; a device that refuses the attribute keeps B + C
  return fused_pc_ready(m, ldm);
}

int DeviceProblem::rtr_dev_fused(const dcora_ropt_params &prm) {
  constexpr int kLookahead = 2;
  const auto t0 = std::chrono::steady_clock::now();
  t0_ms_ = std::chrono::duration<double, std::milli>(t0.time_since_epoch()).count();
  const bool single = (prm.RTR_iterations == 1);
  const int max_outer = single ? 12 : prm.RTR_iterations;  // ref :254-273 (<= 11 retries)
  const int max_inner = prm.RTR_tCG_iterations;
  if (seq_ > 1500000000) {  // keep the monotonic sequence far from INT_MAX
    DCORA_HIP(hipStreamSynchronize(st));
    hf->last_seq_done = 0;
    hf->tcg_done_seq = 0;
    hf->outer_done_seq = 0;
    hf->go_seq = 0;
    hf->reject_seq = 0;
    seq_ = 0;
  }
  int &seq = seq_;
  SolverCtl *c = ctl.p;
  CtlInit ci;  // written by k_rtr_init, the third kernel of the solve: the first two run with a null gate
  ci.enable = 1;
  ci.tol = prm.gradnorm_tol;
  ci.Delta = prm.RTR_initial_radius;
  ci.maxDelta = single ? prm.RTR_initial_radius : 5 * prm.RTR_initial_radius;
  ci.max_outer = max_outer;
  ci.stop_on_accept = single ? 1 : 0;
  ci.max_inner = max_inner;
  const int solve_first = seq + 1;
  const CsrDev Qv = Q.view();
  const BsrDev Qbv = has_bsr ? Qb.view() : BsrDev{};
  const double *Gp = has_G ? G.p : nullptr;
  const int nA = npA(), nPB = fused_pose_blocks(m);
  const int nPG = sparse_precond ? fused_update_grid(m) : fused_precond_grid(m);
  const double *Mi = sparse_precond ? nullptr : Minv.p;  // null: B only updates, the sparse levels follow
  // without hubs the sparse preconditioner's two permutations ride in B (scatter of the residual) and C (gather of z)
  const bool folded = sparse_precond && sp.foldable();
  const SpFold sf = folded ? sp.fold() : SpFold{};
  const int nsl = sparse_precond ? 1 : -1;
  // With the sparse preconditioner the host enqueues the replay behind the step-length kernel's verdict (below); the
  // same pacing on the dense one-launch form measured slower (1846 -> 1729-1857 it/s on the headline: its trace holds 69
  // gated no-op launches among 38 800, and waiting for a verdict there only opens gaps) and is not kept.
  // dense preconditioner: B and C are ONE launch (k_fused_pc); DCORA_SOLVER_BC=split keeps the three-launch form
  const bool pc = use_pc();
  // dense one-launch form: the whole tCG run of an RTR iteration as ONE launch (k_tcg_run) where the grid is co-resident
  // and no other solve shares the device's launch path; a run that gives up (tcg_abort_seq) switches the form off for
  // this problem and the iteration is repeated on the launches
  bool run_form = pc && tcg_run_ok && !concurrent_solves;
  unsigned *sync_p = run_form ? tcg_sync.p : nullptr;
  const int nsync = run_form ? tcg_run_sync_words() : 0;
  const int nZ = pc ? fused_pc_blocks(m) : nPB;  // <z, r> partial slots A sums in its prologue
  double *dbuf[2] = {delta.p, delta2.p};
  double *rbuf[2] = {res.p, res2.p};
  auto timed_out = [&]() {
    set_last_error("rtr_dev_fused: device did not make progress (spin timeout)");
    return DCORA_ERR_HIP;
  };
  auto outer_done = [&]() { return hf->outer_done_seq >= solve_first; };
  // f(x0), grad(x0) from buffer 0 (null gate: the control block of this solve does not exist yet)
  // cost + gradient of an evaluation in one launch where the kernel exists (small CSR blocks), else Q-apply + rgrad
  const bool gf = !has_bsr && Q.n_long == 0;
  const bool gfb = has_bsr;       // large blocks: the same in one launch on the block structure (k_spmm_bsrq<.., GRAD>)
  const Buf2 kNoBuf{{nullptr, nullptr}};  // EG of the fused evaluation: nobody reads it, so it is not written
  const int nAe = gf ? nPB : nA;  // {<XQ,X>, <X,G>} partial slots of an evaluation
  int nG;
  ++seq;
  if (gf) {
    nG = launch_fused_grad(st, m, Qv, Xb(), Gp, EGb(), RGb(), Sb(), 0, pA.p, pB.p, nullptr, Gate{});
  } else if (gfb) {
    nG = launch_fused_grad_bsr(st, m.r, m.d, Qbv, Xb(), Gp, kNoBuf, RGb(), Sb(), 0, pA.p, pB.p, nullptr, Gate{});
  } else {
    enq_qapply(Xb(), 0, Gp, EGb(), 0, pA.p, Gate{});
    ++seq;
    nG = enq_rgrad(Xb(), EGb(), RGb(), Sb(), 0, pB.p, Gate{});
  }
  launch_rtr_init(st, pA.p, nAe, pB.p, nG, c, hf_dev, ++seq, ci, sync_p, nsync);
  int last_pace_seq = seq;
  std::vector<int> fin_seq((size_t)std::max(1, max_inner));
  // The first kernel of an RTR iteration (z0 = P grad: B in "first" mode, or B + C in one launch) needs nothing the host
  // has to decide: it is gated by the RTR loop's own stamp and picks the accepted iterate's buffers from the control
  // block when it starts.  So it is enqueued BEHIND k_rtr_init / k_rtr_decide at once, before the host has seen their
  // verdict: the GPU runs it while the host reads the flag and enqueues the tCG iteration behind it, instead of idling
  // for that round trip at every iteration (a loop that had ended leaves one gated no-op).  Returns its seq, < 0 when
  // the launch failed.
  auto enqueue_first = [&]() -> int {
    if (run_form) {
      hipEvent_t e1 = nullptr;
      if (profile_tcg_runs) {
        while (run_events.size() < run_events_used + 2) {
          hipEvent_t e = nullptr;
          if (hipEventCreate(&e) != hipSuccess) return -1;
          run_events.push_back(e);
        }
        (void)hipEventRecord(run_events[run_events_used], st);
        e1 = run_events[run_events_used + 1];
        run_events_used += 2;
      }
      if (launch_tcg_run(st, m, ldm, Mi, Qv, RGb(), Xb(), Sb(), dbuf[0], dbuf[1], Hd.p, eta.p, Heta.p, z.p, p1.p, p3.p,
                         pC.p, tcg_sync.p, c, hf_dev, ++seq) < 0)
        return -1;
      if (e1) (void)hipEventRecord(e1, st);
      return seq;
    }
    if (pc) {
      if (launch_fused_pc(st, m, ldm, Mi, RGb(), Xb(), nullptr, nullptr, eta.p, Heta.p, nullptr, rbuf[0], z.p, nullptr,
                          0, p3.p, c, hf_dev, ++seq, 0, 1) < 0)
        return -1;
    } else {
      launch_fused_precond(st, m, ldm, Mi, RGb(), nullptr, nullptr, eta.p, Heta.p, nullptr, rbuf[0], Zpart.p,
                           nullptr, 0, p2.p, c, hf_dev, ++seq, 0, 1, sf);
    }
    return seq;
  };
  auto time_is_up = [&]() {  // TimeBound :252
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 5.0;
  };
  int next_first_seq = time_is_up() ? 0 : enqueue_first();
  auto run_gave_up = [&]() { return run_form && hf->tcg_abort_seq >= solve_first; };
  // The host runs ahead of the device by design, but it must not LEAVE while a one-launch run may still give up: the
  // recovery below is the only place that repeats the iteration, and what the caller enqueues next builds on its
  // result.  So with the run form the loop takes one more turn after the last iteration (`tail`), which only waits
  // until that run has ended (its first workgroup raises last_seq_done when the run finishes; a run that finishes has
  // not given up: run_grid_step) or has given up -- then the turn repeats the iteration on the launches like any other.
  // (Round 5: without it a last iteration whose run gave up was lost without a trace -- a flaky bit difference between
  // ranks sharing one GPU, where other ranks' waiting kernels keep the grid from being co-resident.)
  int last_run_seq = 0;
  for (int outer = 0; outer < max_outer || (run_form && last_run_seq > 0); ++outer) {
    const bool tail = outer >= max_outer;
    if (!tail && next_first_seq < 0) {
      DCORA_HIP(hipStreamSynchronize(st));
      set_last_error("k_fused_pc could not be launched on this device");
      return DCORA_ERR_HIP;
    }
    const int need_seq = tail ? last_run_seq : last_pace_seq;
    if (!spin_until([&] { return hf->last_seq_done >= need_seq || outer_done() || run_gave_up(); }, 20.0))
      return timed_out();
    if (tail && !run_gave_up()) break;
    if (run_gave_up()) {
      // The one-launch run of iteration `outer` (or the one before: its evaluation and decision were queued behind it)
      // found its grid not co-resident and left; everything queued behind it was a no-op (outer_done_stamp).  Re-arm
      // the control block, drop the form for this problem and take the iteration again on the launches: the run wrote
      // nothing but scratch (eta, H eta, the trial point and the control block are written when a run ENDS).
      DCORA_HIP(hipStreamSynchronize(st));
      const int armed = INT_MAX;
      DCORA_HIP(hipMemcpy(&c->outer_done_stamp, &armed, sizeof(int), hipMemcpyHostToDevice));
      const bool before = hf->last_seq_done < last_pace_seq;  // the decision of iteration outer - 1 never ran
      hf->tcg_abort_seq = 0;
      tcg_run_ok = false;
      run_form = false;
      if (before) --outer;
      next_first_seq = enqueue_first();
      if (before) {
        // the evaluation + decision of the repeated iteration follow below as usual
      }
      if (next_first_seq < 0) continue;
    }
    if (outer_done()) break;
    if (next_first_seq == 0) break;  // the time bound had passed when this iteration's first kernel was due
    // z0 = P(grad): B in "first" mode streams the preconditioner over grad, C projects and forms <z0, r0>
    const int tcg_first_seq = next_first_seq;
    last_run_seq = run_form ? tcg_first_seq : 0;
    if (!pc) {
      // A rejected step (k_rtr_decide said so in reject_seq) left the iterate and its gradient where they were: z0 is
      // the one of the iteration before, whose unprojected form the finish kernel kept in W (free on this path) --
      // one application of the sparse preconditioner less per rejection (0.25 per RBCD iteration on the 100k lattice).
      const bool reuse_z0 = sparse_precond && outer > 0 && hf->reject_seq == last_pace_seq;
      if (reuse_z0) {
        launch_fused_finish(st, m, Xb(), W.p, rbuf[0], z.p, nullptr, 0, p3.p, c, hf_dev, ++seq, 0, 1, 1, SpFold{});
      } else {
        if (sparse_precond) sp.apply(st, m.r, buf1(rbuf[0]), Zpart.p, Gate{c, ++seq, 1}, folded);
        launch_fused_finish(st, m, Xb(), Zpart.p, rbuf[0], z.p, nullptr, 0, p3.p, c, hf_dev, ++seq, 0, 1, nsl, sf,
                            sparse_precond ? W.p : nullptr);
      }
    }
    for (int j = 0; j < max_inner && !run_form; ++j) {
      if (j >= kLookahead) {
        const int need = fin_seq[j - kLookahead];
        if (!spin_until(
                [&] { return hf->last_seq_done >= need || hf->tcg_done_seq >= tcg_first_seq || outer_done(); }, 20.0))
          return timed_out();
      }
      if (hf->tcg_done_seq >= tcg_first_seq) break;
      const int par = j & 1;
      const int nP1 = launch_fused_hess(st, m, Qv, z.p, dbuf[par ^ 1], dbuf[par], Xb(), Sb(), Hd.p, p3.p, nZ, p1.p,
                                        c, ++seq, j, has_bsr ? &Qbv : nullptr);
      if (pc) {
        launch_fused_pc(st, m, ldm, Mi, RGb(), Xb(), dbuf[par], Hd.p, eta.p, Heta.p, rbuf[par], rbuf[par ^ 1], z.p,
                        p1.p, nP1, p3.p, c, hf_dev, ++seq, j, 0, pC.p);
      } else {
        launch_fused_precond(st, m, ldm, Mi, RGb(), dbuf[par], Hd.p, eta.p, Heta.p, rbuf[par], rbuf[par ^ 1],
                             Zpart.p, p1.p, nP1, p2.p, c, hf_dev, ++seq, j, 0, sf);
        if (sparse_precond) {
          // B decides whether this tCG run goes on (boundary, negative curvature); the replay's launches behind a B
          // that stopped are no-ops of ~4 us each, and runs of one or two iterations are the rule on the large
          // blocks.  So only the replay's FIRST launch is enqueued on speculation: the host reads B's verdict while
          // the GPU runs it, and enqueues the rest (and C) only behind a B that went on.
          const int seqB = seq;
          bool timed = false;
          const std::function<bool()> verdict = [&]() {
            if (!spin_until(
                    [&] {
                      return hf->go_seq >= seqB || hf->tcg_done_seq >= tcg_first_seq || outer_done();
                    },
                    20.0)) {
              timed = true;
              return false;
            }
            return !(hf->tcg_done_seq >= tcg_first_seq || outer_done());
          };
          sp.apply(st, m.r, buf1(rbuf[par ^ 1]), Zpart.p, Gate{c, ++seq, 2}, folded, &verdict);
          if (timed) return timed_out();
          if (hf->tcg_done_seq >= tcg_first_seq || outer_done()) break;
        }
        launch_fused_finish(st, m, Xb(), Zpart.p, rbuf[par ^ 1], z.p, p2.p, nPG, p3.p, c, hf_dev, ++seq, j, 0, nsl,
                            sf);
      }
      fin_seq[j] = seq;
    }
    // (one-launch dense form: the kernel that ended the tCG run has retracted already, one partial pair per workgroup)
    const int nR = pc ? nZ : enq_retract(Xb(), eta.p, 1.0, Xb(), 1, RGb(), Heta.p, pC.p, Gate{c, ++seq, 1});
    if (gf) {
      nG = launch_fused_grad(st, m, Qv, Xb(), Gp, EGb(), RGb(), Sb(), 1, pA.p, pB.p, nullptr, Gate{c, ++seq, 1});
    } else if (gfb) {
      nG = launch_fused_grad_bsr(st, m.r, m.d, Qbv, Xb(), Gp, kNoBuf, RGb(), Sb(), 1, pA.p, pB.p, nullptr,
                                 Gate{c, ++seq, 1});
    } else {
      enq_qapply(Xb(), 1, Gp, EGb(), 1, pA.p, Gate{c, ++seq, 1});
      nG = enq_rgrad(Xb(), EGb(), RGb(), Sb(), 1, pB.p, Gate{c, ++seq, 1});
    }
    launch_rtr_decide(st, pA.p, nAe, pB.p, nG, pC.p, nR, c, hf_dev, ++seq, sync_p, nsync);
    last_pace_seq = seq;
    if (outer + 1 < max_outer) next_first_seq = time_is_up() ? 0 : enqueue_first();
  }
  return DCORA_OK;
}

// ref src/QuadraticProblem.cpp:138-234 (rare path, host-paced: one scalar read-back per trial step)
int DeviceProblem::escape_saddle(const double *Xopt, double theta, const double *v, double gtol, double pgtol,
                                 bool second_order, double *Xout, int *success) {
  if (!has_precond) {
    set_last_error("escapeSaddle needs the preconditioner");
    return DCORA_ERR_NO_PRECONDITIONER;
  }
  DCORA_HIP(hipSetDevice(device));
  const int r = m.r, k = m.k;
  std::vector<double> Xp((size_t)r * k, 0.0), Xd((size_t)r * k, 0.0);
  for (int j = 0; j < k; ++j) {
    for (int t = 0; t < r - 1; ++t) Xp[(size_t)j * r + t] = Xopt[(size_t)j * (r - 1) + t];
    Xd[(size_t)j * r + (r - 1)] = v[j];
  }
  int rc = upload(Xp.data(), X0.p, nelem());
  if (rc) return rc;
  rc = upload(Xd.data(), eta.p, nelem());
  if (rc) return rc;
  DCORA_HIP(hipStreamSynchronize(st));
  double FX = 0;
  rc = eval_dev(X0.p, &FX, nullptr);
  if (rc) return rc;
  const double alpha_min = 1e-6;
  // ref :165-169: SE-Sync's second-order step when asked for, else Algorithm 7 of the DC2-PGO report
  double alpha = second_order ? std::max(16 * alpha_min, 100 * gtol / std::fabs(theta)) : 1.0;
  std::vector<double> alphas, fvals;
  *success = 0;
  while (alpha >= alpha_min) {
    launch_retract(st, m, buf1(X0.p), eta.p, alpha, buf1(X1.p), 0, buf1(RG0.p), nullptr, nullptr, Gate{});
    enqueue_egrad(X1.p, EG1.p, pA.p);
    launch_rgrad(st, m, buf1(X1.p), buf1(EG1.p), buf1(RG1.p), buf1(S1.p), 0, pB.p, Gate{});
    enqueue_precond(X1.p, RG1.p, z.p);
    launch_dot(st, nelem(), z.p, z.p, p3.p);
    launch_sum_partials(st, pA.p, npA(), 2, 2, scal.p);
    launch_sum_partials(st, pB.p, npPose(), 1, 1, scal.p + 2);
    launch_sum_partials(st, p3.p, npVec(), 1, 1, scal.p + 3);
    double s[4];
    rc = download(scal.p, s, 4);
    if (rc) return rc;
    const double FXt = 0.5 * s[0] + s[1], gn = std::sqrt(s[2]), pgn = std::sqrt(s[3]);
    alphas.push_back(alpha);
    fvals.push_back(FXt);
    if (FXt < FX && gn > gtol && pgn > pgtol) {
      *success = 1;
      return download(X1.p, Xout, nelem());
    }
    alpha /= 2;
  }
  const size_t idx = std::min_element(fvals.begin(), fvals.end()) - fvals.begin();
  if (fvals[idx] < FX) {
    launch_retract(st, m, buf1(X0.p), eta.p, alphas[idx], buf1(X1.p), 0, buf1(RG0.p), nullptr, nullptr, Gate{});
    *success = 1;
    return download(X1.p, Xout, nelem());
  }
  return DCORA_OK;
}

int DeviceProblem::time_qapply(int reps, double *avg_ms, double *bytes) {
  DCORA_HIP(hipSetDevice(device));
  hipEvent_t e0, e1;
  DCORA_HIP(hipEventCreate(&e0));
  DCORA_HIP(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) enqueue_egrad(X0.p, EG0.p, nullptr);
  DCORA_HIP(hipEventRecord(e0, st));
  for (int i = 0; i < reps; ++i) enqueue_egrad(X0.p, EG0.p, nullptr);
  DCORA_HIP(hipEventRecord(e1, st));
  DCORA_HIP(hipEventSynchronize(e1));
  float ms = 0;
  DCORA_HIP(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  *avg_ms = (double)ms / reps;
  // nnz (8+4) + (k+1) 4 + r k 8 (read X) + r k 8 (write Y) [+ r k 8 for G]
  *bytes = 12.0 * Q.nnz + 4.0 * (m.k + 1) + 16.0 * m.r * m.k + (has_G ? 8.0 * m.r * m.k : 0.0);
  return DCORA_OK;
}

}  // namespace dcora

namespace dcora {
// times the dense-preconditioner kernel of the solver (k_fused_precond in its start-of-tCG form: residual = grad,
// no step) with HIP events on the problem's stream
int DeviceProblem::time_precond(int reps, double *avg_ms, double *bytes) {
  if (!has_precond || (!fused && !sparse_precond)) {
    set_last_error("time_precond: needs a preconditioner and the fused solver path or the sparse preconditioner");
    return DCORA_ERR_UNSUPPORTED;
  }
  DCORA_HIP(hipSetDevice(device));
  launch_ctl_init(st, ctl.p, 1e-2, 100, 500, 3, 0, 50);
  hipEvent_t e0, e1;
  DCORA_HIP(hipEventCreate(&e0));
  DCORA_HIP(hipEventCreate(&e1));
  // The dense kernel is timed in its in-loop form (step length from the <d, H d> partials, vector updates, |r|^2),
  // not in the shorter form that opens a tCG run; the partials are set so that the step is a no-op.
  constexpr bool step_form = true;
  const int nPB = fused_pose_blocks(m);
  if (step_form && !sparse_precond) {
    std::vector<double> ones(std::max((size_t)nPB, (size_t)nelem()), 1.0);
    DCORA_HIP(hipMemcpyAsync(p1.p, ones.data(), sizeof(double) * nPB, hipMemcpyHostToDevice, st));
    // a non-zero residual, so that the one-launch form does not leave through its stopping rule
    DCORA_HIP(hipMemcpyAsync(res.p, ones.data(), sizeof(double) * nelem(), hipMemcpyHostToDevice, st));
    DCORA_HIP(hipMemsetAsync(Hd.p, 0, sizeof(double) * nelem(), st));
    DCORA_HIP(hipStreamSynchronize(st));
  }
  const bool bc_split = !use_pc();
  auto run = [&]() {
    if (sparse_precond)
      sp.apply(st, m.r, buf1(RG0.p), Zt.p, Gate{});
    else if (!bc_split)  // the one-launch B + C of the dense path, in its in-loop form
      launch_fused_pc(st, m, ldm, Minv.p, RGb(), Xb(), delta.p, Hd.p, eta.p, Heta.p, res.p, res2.p, z.p, p1.p,
                      nPB, p3.p, ctl.p, hf_dev, 1, 1, 0);
    else if (step_form)
      launch_fused_precond(st, m, ldm, Minv.p, RGb(), delta.p, Hd.p, eta.p, Heta.p, res.p, res2.p, Zpart.p, p1.p, nPB,
                           p2.p, ctl.p, hf_dev, 1, 1, 0);
    else
      launch_fused_precond(st, m, ldm, Minv.p, RGb(), nullptr, nullptr, eta.p, Heta.p, nullptr, res.p, Zpart.p,
                           nullptr, 0, p2.p, ctl.p, hf_dev, 1, 0, 1);
  };
  for (int i = 0; i < 3; ++i) run();
  DCORA_HIP(hipEventRecord(e0, st));
  for (int i = 0; i < reps; ++i) run();
  DCORA_HIP(hipEventRecord(e1, st));
  DCORA_HIP(hipEventSynchronize(e1));
  float ms = 0;
  DCORA_HIP(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  *avg_ms = (double)ms / reps;
  // algorithmic bytes: the k x k inverse once, the residual in, the split-K slices out
  // algorithmic bytes, dense form: the k x k inverse once; split form: + the residual in and the split-K slices out;
  // one-launch form: + r_old, H delta in and eta, H eta, r, z through once (7 r k doubles)
  *bytes = sparse_precond ? sp.bytes_per_apply(m.r)
           : !bc_split    ? 8.0 * m.k * (double)m.k + 7.0 * 8.0 * m.r * m.k
                          : 8.0 * m.k * (double)m.k + 8.0 * m.r * m.k + 8.0 * m.r * m.k * fused_nsplit(m);
  return DCORA_OK;
}
}  // namespace dcora
