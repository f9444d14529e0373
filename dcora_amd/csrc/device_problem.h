// QuadraticProblem + QuadraticOptimizer on the device: owns Q, G, the preconditioner and the solver workspace
// in HBM, and drives the RTR / tCG kernel sequence on one HIP stream
// (replaces src/QuadraticProblem.cpp and src/QuadraticOptimizer.cpp of the reference).
#pragma once
#include <hip/hip_runtime.h>

#include <functional>
#include <memory>
#include <string>
#include <vector>

#include "../../include/dcora_hip.h"
#include "host_sparse.h"
#include "kernels.h"
#include "sparse_precond.h"

namespace dcora {

void set_last_error(const std::string &s);
int hip_fail(hipError_t e, const char *what, const char *file, int line);
#define DCORA_HIP(call)                                                        \
  do {                                                                         \
    hipError_t e_ = (call);                                                    \
    if (e_ != hipSuccess) return ::dcora::hip_fail(e_, #call, __FILE__, __LINE__); \
  } while (0)

template <class T>
struct DevBuf {
  T *p = nullptr;
  size_t n = 0;
  DevBuf() = default;
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(const DevBuf &) = delete;
  DevBuf(DevBuf &&o) noexcept : p(o.p), n(o.n), borrowed(o.borrowed) {
    o.p = nullptr;
    o.n = 0;
    o.borrowed = false;
  }
  DevBuf &operator=(DevBuf &&o) noexcept {
    if (this != &o) {
      release();
      p = o.p;
      n = o.n;
      borrowed = o.borrowed;
      o.p = nullptr;
      o.n = 0;
      o.borrowed = false;
    }
    return *this;
  }
  ~DevBuf() { release(); }
  bool borrowed = false;  // p belongs to somebody else (a cached image kept alive by a shared_ptr next to this)
  void release() {
    if (p && !borrowed) (void)hipFree(p);
    p = nullptr;
    n = 0;
    borrowed = false;
  }
  void borrow(T *q, size_t count) {
    release();
    p = q;
    n = count;
    borrowed = true;
  }
  hipError_t alloc(size_t count) {
    release();
    n = count;
    if (count == 0) return hipSuccess;
    return hipMalloc((void **)&p, count * sizeof(T));
  }
};

struct DevBsr {
  int nbrows = 0, nblocks = 0;
  DevBuf<int> bp, bc;
  DevBuf<double> bv;
  int upload(const HostBsr &B);
  BsrDev view() const {
    BsrDev v;
    v.nbrows = nbrows;
    v.nblocks = nblocks;
    v.bp = bp.p;
    v.bc = bc.p;
    v.bv = bv.p;
    return v;
  }
};

struct DevCsr {
  int nrows = 0, ncols = 0, nnz = 0, n_long = 0;
  DevBuf<int> rp, ci, long_rows, long_cnt;
  std::vector<int> long_host;  // the long rows' indices, host copy
  DevBuf<double> v, long_part;
  int upload(const HostCsr &A);
  int upload(int nrows, int ncols, const int *rp, const int *ci, const double *v);
  CsrDev view() const {
    CsrDev c;
    c.nrows = nrows;
    c.nnz = nnz;
    c.rp = rp.p;
    c.ci = ci.p;
    c.v = v.p;
    c.n_long = n_long;
    c.long_rows = long_rows.p;
    c.long_part = long_part.p;
    c.long_cnt = long_cnt.p;
    return c;
  }
};

// device image of the partitioned inverse (sparse_precond.h) and its level-by-level replay
// above this the preconditioner is the partitioned sparse inverse.  Measured on MI355X in round 4 (RBCD iterations/s of
// sphere2500 split into 5 / 4 / 3 / 2 agents, tools/bench_crossover.py), dense against sparse:
//   k = 2000: 1849 / 1297 (r = 5), 821 / 560 (r = 3), 1466 / 1489 (r = 7)  -- the one-launch form k_fused_pc exists there
//   k = 2500: 1018 / 1112, 487 / 536, 1025 / 1154;  k = 3336: 654 / 784;  k = 5000: 209 / 355
// (rounds 1-3, with the replay and the ordering of the time: 8000)
constexpr int kDensePrecondMaxK = 2200;

// immutable device image of a partitioned inverse: shared between the problems that precondition with the same
// Q + reg I (the staircase levels, problems re-created per update) through the cache of precond_cache.h
struct SpImage {
  int k = 0, npieces = 0;
  long nnzL = 0, nmwaves_total = 0;
  double weights_per_apply = 0, rows_total = 0;
  std::vector<SpLevel> levels;
  DevBuf<double> vals;
  DevBuf<int> idxs, perm, out_off;
  DevBuf<MWave> mwaves;  // the wave records of the schedule (sparse_precond.h)
  // hubs (PartInvHub): Schur complement data
  int nhub = 0;
  long hub_nnz = 0;
  DevBuf<int> hub_idx;
  DevBuf<double> hub_U, hub_Sinv;
  // original unknown -> position in image 0 of the replay vector / position of its final value (-1 on a hub):
  // lets the caller's kernels write the right-hand side into y and read the result from it (SpFold, kernels.h)
  DevBuf<int> in_pos, out_pos;
  // weights from P.vals, or -- when the build streamed them (P.sink) -- the sink's device buffer, which is taken over
  int upload(const PartInvHost &P, struct DeviceWeightSink *streamed = nullptr);
  size_t device_bytes() const;
};

// WeightSink of the product: the stored weights go to the device in chunks of pinned host memory while the host's
// threads form the next chunk (three buffers in flight, recycled process-wide)
struct DeviceWeightSink : WeightSink {
  explicit DeviceWeightSink(int device_) : device(device_) {}
  ~DeviceWeightSink() override;
  bool fill_on_device(const std::vector<partinv::Fill> &fills, long long total, const std::vector<MirrorRange> &mirrors,
                      int nthreads) override;
  bool wants_device_sources() const override;
  bool begin(long long total) override;
  long long chunk_cap() const override { return kChunk; }
  double *acquire(long long n) override;
  bool commit(long long off, long long n) override;
  bool end() override;
  static constexpr long long kChunk = 8ll << 20;  // doubles per chunk: 64 MB
  static constexpr int kBuffers = 3;
  int device;
  DevBuf<double> vals;
  hipStream_t st = nullptr;
  double *pin[kBuffers] = {nullptr, nullptr, nullptr};
  hipEvent_t ev[kBuffers] = {nullptr, nullptr, nullptr};
  bool busy[kBuffers] = {false, false, false};
  int cur = -1;
};

class SparsePrecond {
 public:
  std::shared_ptr<const SpImage> im;  // the stored inverse (read-only)
  int rcap = 0;
  DevBuf<double> y, hub_w, hub_x2;    // this problem's replay vector (two ping-pong images) and hub scratch
  double weights_per_apply = 0;
  bool foldable() const { return im && im->nhub == 0 && im->in_pos.p != nullptr; }
  SpFold fold() const {
    SpFold f;
    if (foldable()) {
      f.y = y.p;
      f.in_pos = im->in_pos.p;
      f.out_pos = im->out_pos.p;
    }
    return f;
  }
  // the same for the generic-layout kernels (k_tcg_update1 / k_tangent), which also apply the hub correction
  SpFold fold_generic() const;
  int attach(std::shared_ptr<const SpImage> image, int rcap);
  int launches() const { return (int)im->levels.size() + 2; }  // permutation in, the levels, permutation out
  // Z = R A^-1 for r <= rcap right-hand sides (r x k column-major); R is picked by ctl->cur when g.ctl is set.
  // levels_only: the right-hand side is already in y and the result is read from y (fold())
  // after_first: called once the first launch of the replay is enqueued; returning false ends the enqueue there (the
  // caller has learnt meanwhile that the gate is closed and the remaining launches would be no-ops)
  void apply(hipStream_t st, int r, Buf2 R, double *Z, Gate g, bool levels_only = false,
             const std::function<bool()> *after_first = nullptr) const;
  double bytes_per_apply(int r) const;
};

int stream_acquire(int device, hipStream_t *out);  // recycled non-blocking streams (creation costs milliseconds)
void stream_release(int device, hipStream_t st);
int host_flags_acquire(HostFlags **host, HostFlags **dev, int *slot);
void host_flags_release(HostFlags *host, int slot);

// dense inverses of several matrices in one batch of launches, left in the preconditioner cache (device_problem.hip)
int precond_prebuild_dense(const std::vector<const HostCsr *> &Qs, double reg, int block, int device);

class DeviceProblem {
 public:
  ManiDesc m{};
  int device = 0;
  hipStream_t st = nullptr;
  bool own_stream = false;
  DevCsr Q;
  DevBsr Qb;          // block form of Q (SE layout): the Q-apply fast path
  bool has_bsr = false;
  // Y = X Q (+G) with optional {<XQ,X>, <X,G>} partials; returns the number of partial slots written
  int enq_qapply(Buf2 X, int selX, const double *Gp, Buf2 Y, int selY, double *partials, Gate g);
  DevBuf<double> G;     // r x k (always allocated; zero when the problem has no linear term)
  bool has_G = false;
  DevBuf<double> Minv;  // k x ldm dense inverse of Q + reg I (small blocks)
  int ldm = 0;
  SparsePrecond sp;     // partitioned sparse inverse (large blocks)
  std::shared_ptr<const DevBuf<double>> Minv_shared;
  bool precond_cache_hit = false;  // owner of the dense inverse Minv.p points to (cache)
  bool sparse_precond = false;
  bool has_precond = false;
  double precond_setup_ms = 0;
  long precond_nnzL = 0;

  // solver workspace
  DevBuf<double> X0, X1, EG0, EG1, RG0, RG1, S0, S1;
  DevBuf<double> delta, eta, Heta, res, z, Hd, W, Zt;
  DevBuf<double> delta2, res2, Zpart;  // fused path: ping-pong direction / residual, split-K slices
  // the one-launch tCG run (k_tcg_run): eligible sizes on this device; switched off for good after a run that gave up
  // (grid not co-resident) and while several solves run at the same time (the coloured mode: concurrent_solves)
  bool tcg_run_ok = false;
  bool concurrent_solves = false;
  DevBuf<unsigned> tcg_sync;
  // measurement (bench.py's roofline): HIP events on the solver's stream around every k_tcg_run launch while switched on
  bool profile_tcg_runs = false;
  std::vector<hipEvent_t> run_events;  // pairs (start, stop)
  size_t run_events_used = 0;
  int profile_tcg_read(double *launches, double *total_us);
  bool fused = false;                  // SE layout, r <= 8: three-launch tCG iteration (solver_fused.hip)
  bool group = false;                  // SE layout, r <= 8: 8-lanes-per-pose rgrad / retract kernels (any n)
  DevBuf<double> pA, pB, pC, p1, p2, p3, scal;
  DevBuf<SolverCtl> ctl;
  DevBuf<double> arena;         // one allocation behind every workspace buffer above (they borrow slices)
  int hf_slot = -1;             // slot of hf in the process-wide host-mapped page (-1: a page of its own)
  HostFlags *hf = nullptr;      // host-mapped
  HostFlags *hf_dev = nullptr;  // device view of the same words
  std::vector<double> stage;    // pageable staging for host <-> device copies

  long nelem() const { return (long)m.r * m.k; }
  Buf2 Xb() const { return Buf2{{X0.p, X1.p}}; }
  Buf2 EGb() const { return Buf2{{EG0.p, EG1.p}}; }
  Buf2 RGb() const { return Buf2{{RG0.p, RG1.p}}; }
  Buf2 Sb() const { return Buf2{{S0.p, S1.p}}; }

  ~DeviceProblem();
  // Q as host CSR; G host (may be null); reg < 0 => no preconditioner; stream may be shared (null => own)
  int init(const dcora_dims &dims, const HostCsr &Qh, const double *Gh, double reg, int device_, hipStream_t shared);
  int set_G_host(const double *Gh);
  int build_preconditioner(const HostCsr &Qh, double reg);

  // ---- device-level building blocks (all enqueue on st, no sync) ----
  // EG = X Q + G, partials pA (npA slots of 2)
  int npA() const { return has_bsr ? spmm_bsr_grid(m.n) : spmm_grid(m.k, m.r) + Q.n_long; }
  int npPose() const { return pose_grid(m); }
  // rgrad / retract with the kernel flavour of this problem; return the number of partial slots written
  int enq_rgrad(Buf2 X, Buf2 EG, Buf2 RG, Buf2 S, int sel, double *partials, Gate g, double *posenorm = nullptr);
  int enq_retract(Buf2 X, const double *V, double alpha, Buf2 out, int selOut, Buf2 grad, const double *HV,
                  double *partials, Gate g);
  int npVec() const { return vec_grid(nelem()); }
  void enqueue_egrad(const double *X, double *EG, double *partials);
  void enqueue_precond(const double *X, const double *V, double *out);  // out = Proj_X(V Minv)
  // Z = R (Q + reg I)^-1 with whichever preconditioner form this problem holds (p2 / np2: the dense kernel's
  // early-out on the residual rule)
  void enq_minv(Buf2 R, double *Z, const double *p2, int np2, Gate g);

  // ---- host-pointer API (upload, run, download, sync) ----
  int cost(const double *Xh, double *f);
  int eucgrad(const double *Xh, double *out);
  int riegrad(const double *Xh, double *out, double *norm);
  int hessvec(const double *Xh, const double *Vh, double *out);
  // the Hessian-vector product as the generic solver loop forms it (k_spmm_dir_fix: one launch); dots = {<V, H V>
  // from that kernel's partial sums, the same from k_hessfix's}.  Tests of the kernel against hessvec().
  int hessvec_solver_form(const double *Xh, const double *Vh, double *out, double *dots);
  bool hess_one_launch() const;  // every long row is a Euclidean column and the partial slots fit
  int precondition(const double *Xh, const double *Vh, double *out);
  int retract(const double *Xh, const double *Vh, double *out);
  int tangent_project(const double *Xh, const double *Vh, double *out);
  int optimize(const dcora_ropt_params &prm, const double *X0h, double *Xout, dcora_ropt_result *res);
  int escape_saddle(const double *Xopt, double theta, const double *v, double gtol, double pgtol, bool second_order,
                    double *Xout, int *success);

  // ---- device-resident solve: X0.p holds the start point; on return *Xres points at the result buffer ----
  // The fused path returns without synchronising: the result lives in Xres->p[(*ctl_out)->cur] (device-side
  // pick), statistics are fetched later with fetch_result().  Other paths return ctl_out = nullptr and a
  // resolved pointer in Xres->p[0].
  int optimize_dev(const dcora_ropt_params &prm, Buf2 *Xres, const SolverCtl **ctl_out);
  int fetch_result(dcora_ropt_result *res);
  int result_index() const { return cur_after_fetch_; }  // synchronises when a fused solve is still in flight
  // scalars of an arbitrary point on device: f and |rgrad| (synchronises)
  int eval_dev(const double *Xd, double *f, double *gradnorm);

  int time_qapply(int reps, double *avg_ms, double *bytes);
  int time_precond(int reps, double *avg_ms, double *bytes);
  bool use_pc() const;  // dense preconditioner, step + product + projection in one launch (k_fused_pc)

 private:
  int rtr_dev(const dcora_ropt_params &prm, dcora_ropt_result *res, double **Xres);
  int rtr_dev_fused(const dcora_ropt_params &prm);
  int rgd_dev(const dcora_ropt_params &prm, dcora_ropt_result *res, double **Xres);
  int seq_ = 0;            // launch sequence number, monotonic across solves (HostFlags words are never reset)
  bool pending_ = false;   // a fused solve has been enqueued and its statistics not yet fetched
  double t0_ms_ = 0;
  dcora_ropt_result last_res_{};
  int cur_after_fetch_ = 0;
  int upload(const double *h, double *d, size_t n);
  int download(const double *d, double *h, size_t n);
};

}  // namespace dcora
