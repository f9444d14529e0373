// Device-side building blocks of the DCORA hot path on gfx950 (MI355X).
// Launch wrappers only; every wrapper enqueues on `st` and never synchronises.
//
// Data layout in HBM:
//   * lifted variables / tangent vectors: r x k column-major, tight (ld = r), so the
//     (d+1) columns of one pose are one contiguous run of r(d+1) doubles;
//   * Q, S, coupling blocks: CSR (int32 rowptr/colidx, fp64 values), rows = output columns;
//   * dense preconditioner (Q + reg I)^-1: k rows with leading dimension ldm (multiple of 16);
//   * reductions: per-block partial sums in fixed slots, summed in a fixed order by the consumer
//     kernel's prologue => bitwise reproducible, no floating-point atomics.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <climits>
#include <vector>

namespace dcora {

constexpr int kMaxPartials = 1024;  // upper bound on per-kernel partial-sum slots
constexpr int kBsrMaxGrid = 4096;   // workgroups (= partial slots) of the block-CSR Q-apply
constexpr int kBlock = 256;

struct ManiDesc {
  int r, d, n, l, b, k, se;
  __host__ __device__ int rot_col(int i) const { return se ? i * (d + 1) : i * d; }
  __host__ __device__ int sphere_col(int i) const { return d * n + i; }
  __host__ __device__ int num_euc() const { return se ? n : n + b; }
  __host__ __device__ int euc_col(int e) const { return se ? e * (d + 1) + d : d * n + l + e; }
};
// layout: 0 = SE ordering when l = b = 0 (DCORA_LAYOUT_AUTO), 1 = SE, 2 = RA whatever l and b are (a range-aided
// graph without ranges or landmarks keeps the RA ordering: ref src/Graph.cpp:68-75)
inline ManiDesc make_mani(int r, int d, int n, int l, int b, int layout = 0) {
  ManiDesc m;
  m.r = r; m.d = d; m.n = n; m.l = l; m.b = b;
  m.se = (layout == 2) ? 0 : ((l == 0 && b == 0) ? 1 : 0);
  m.k = (d + 1) * n + l + b;
  return m;
}
template <class Dims>
inline ManiDesc make_mani(const Dims &dims) {
  return make_mani(dims.r, dims.d, dims.n, dims.l, dims.b, dims.layout);
}

constexpr int kLongSplit = 8;  // workgroups per long row; the last one to arrive adds the slices in order
constexpr int kLongRow = 512;  // rows with more entries (a landmark ranged from every pose) get a block of their own
struct CsrDev {
  int nrows = 0;
  int nnz = 0;
  const int *rp = nullptr;
  const int *ci = nullptr;
  const double *v = nullptr;
  int n_long = 0;                  // rows with more than kLongRow entries ...
  const int *long_rows = nullptr;  // ... and their indices
  double *long_part = nullptr;     // n_long x kLongSplit x 16 partial sums of the workgroups sharing a long row
  int *long_cnt = nullptr;         // n_long arrival counters (zero between launches)
};

// block-CSR view of a pose-graph matrix: (d+1) x (d+1) dense blocks, row-major inside a block
struct BsrDev {
  int nbrows = 0;
  int nblocks = 0;
  const int *bp = nullptr;   // block row pointers (nbrows + 1)
  const int *bc = nullptr;   // block column indices
  const double *bv = nullptr;  // (d+1)^2 values per block
};

// two-buffer handle: the RTR solver keeps the accepted iterate and the trial point in a pair of buffers and
// flips an index in device memory on acceptance, so kernels choose their operand on the device.
struct Buf2 {
  double *p[2];
};
inline Buf2 buf1(double *x) { return Buf2{{x, x}}; }
inline Buf2 buf1(const double *x) { return Buf2{{const_cast<double *>(x), const_cast<double *>(x)}}; }

// Device-resident control block of one RTR solve.  Scalars never leave HBM during the solve; the host
// only paces the launch queue by polling HostFlags (host-mapped) words.
struct SolverCtl {
  // trust-region state
  double f1, ngf, Delta, maxDelta, tol, f2, rho, fInit, gradNormInit;
  int cur;             // index of the accepted iterate's buffers (0/1)
  int outer_it, max_outer, accepted, last_accepted, stop_on_accept;
  int outer_done_stamp;  // kernels with seq > stamp are no-ops
  // tCG state, double-buffered by iteration parity
  double z_r[2], d_Pd[2], e_Pe[2], e_Pd[2];
  double alpha, e_Pe_n, norm_r0;
  int tcg_done_stamp, tcg_status, tcg_iters, inner_total, max_inner;
};
constexpr int kMaxAgents = 64;
// results of one evaluation pass, in host-mapped memory
struct EvalOut {
  double cost2, gradnorm;
  double block_norms[kMaxAgents];
  int next;
  volatile int seq;
};
struct HostFlags {
  volatile int last_seq_done;   // seq of the latest pacing kernel whose block 0 finished
  volatile int tcg_done_seq;    // seq at which the current/last tCG terminated
  volatile int outer_done_seq;  // seq at which the RTR loop terminated (0 = running)
  volatile int go_seq;          // seq of the latest fused B (or B+C) kernel whose boundary test let the tCG run go on
  volatile int reject_seq;      // seq of the latest k_rtr_decide that rejected its step (the iterate did not move)
  volatile int tcg_abort_seq;   // seq of a one-launch tCG run (k_tcg_run) that gave up: its grid was not co-resident
};

// Solver-aware launches carry (ctl, seq, gate): gate 0 = always run, 1 = skip once the RTR loop is done,
// 2 = skip once the RTR loop or the current tCG is done.  sel: 0 = accepted iterate, 1 = trial point.
struct Gate {
  const SolverCtl *ctl = nullptr;
  int seq = 0;
  int gate = 0;
};

// ---- SpMM: Y = X * A (+ G); optional partial dots {sum (X*A) o X, sum X o G}, 2 per block ---------------
int spmm_grid(int nrows, int r);
// W = (-z + beta d_old) Q with the direction written to d_new and the tCG scalar recurrence of iteration `iter`
// (k_tcg_init for iter 0, k_tcg_update2 of iteration iter - 1 otherwise) folded in
// k_spmm_dir with k_hessfix folded in (one launch per tCG iteration of the generic layout); returns the number of
// <delta, Hd> partial slots written to p1.  spmm_dir_fix_grid: 0 when r is too large for whole items per workgroup.
int spmm_dir_fix_grid(const ManiDesc &m, int nrows);
int launch_spmm_dir_fix(hipStream_t st, const ManiDesc &m, const CsrDev &A, Buf2 X, Buf2 Sblk, const double *z,
                        const double *d_old, double *d_new, double *Hd, const double *p3, int np3, double *p1,
                        SolverCtl *ctl, int seq, int iter);
void launch_spmm_dir(hipStream_t st, int r, const CsrDev &A, const double *z, const double *d_old, double *d_new,
                     double *W, const double *p3, int np3, SolverCtl *ctl, int seq, int iter);
inline int spmm_slots(const CsrDev &A, int r) { return spmm_grid(A.nrows, r) + A.n_long; }  // partial slots written
void launch_spmm(hipStream_t st, int r, const CsrDev &A, Buf2 X, int selX, const double *G, Buf2 Y, int selY,
                 double *partials, Gate g);

// BSR flavour for the SE layout (8 lanes per pose, one gather of the neighbour's r x (d+1) block per matrix block);
// returns the number of partial slots written (2 doubles each)
int spmm_bsr_grid(int nbrows);
void launch_spmm_bsr(hipStream_t st, int r, int d, const BsrDev &A, Buf2 X, int selX, const double *G, Buf2 Y,
                     int selY, double *partials, Gate g);

// ---- per-pose kernels -------------------------------------------------------------------------------
int pose_grid(const ManiDesc &m);
// RG = Proj_X(EG); Sblk_i = sym(Y_i^T EG_i) (spheres: y^T eg); partial |RG|^2 (1 per block)
void launch_rgrad(hipStream_t st, const ManiDesc &m, Buf2 X, Buf2 EG, Buf2 RG, Buf2 Sblk, int sel,
                  double *partials, Gate g);
// The sparse preconditioner's two permutations folded into the kernels around it: B scatters the new residual into
// image 0 of the replay vector (in_pos: original unknown -> position), C reads z from where the replay leaves it
// (out_pos: original unknown -> final position).  y == nullptr: not folded.
struct SpFold {
  double *y = nullptr;
  const int *in_pos = nullptr, *out_pos = nullptr;
  // hubs (sparse_precond.h, PartInvHub; the generic-layout kernels only): in_pos / out_pos are -1 on a hub unknown,
  // z = y1 - U x2, z(hub) = x2, with x2 = Sinv (r(hub) - U^T r1) (h x r values) formed inside the replay's launches
  // (hub_slice / hub_x2, sparse_precond.hip) -- the consumer reads it as it stands
  int h = 0;
  const int *hub_idx = nullptr;
  const double *hub_U = nullptr, *hub_x2 = nullptr;
};
// out = Proj_X(V); partial <out, R> when R != null.  When p2 != null the prologue evaluates the tCG
// residual stopping rule |r| <= |r0| min(|r0|^theta, kappa) from the np2 partials of |r|^2.
void launch_tangent(hipStream_t st, const ManiDesc &m, Buf2 X, const double *V, double *out, const double *R,
                    double *partials, const double *p2, int np2, SolverCtl *ctl, HostFlags *hf, int seq,
                    int gate, int iter, SpFold sf = SpFold());
// HV = Proj_X(W - V S); partial <V, HV>
void launch_hessfix(hipStream_t st, const ManiDesc &m, Buf2 X, Buf2 Sblk, const double *V, const double *W,
                    double *HV, double *partials, Gate g);
// out = Retr_X(alpha V); when grad/HV given: partial {<V,grad>, <V,HV>} (2 per block); out buffer = trial
void launch_retract(hipStream_t st, const ManiDesc &m, Buf2 X, const double *V, double alpha, Buf2 out,
                    int selOut, Buf2 grad, const double *HV, double *partials, Gate g);
// out = metric projection (polar / normalise) of c0 A + c1 B + c2 C   (B, C may be null)
void launch_polar(hipStream_t st, const ManiDesc &m, double c0, const double *A, double c1, const double *B,
                  double c2, const double *C, double *out);

// RBCD++ Nesterov bookkeeping on a pose range (modes: see kernels.hip k_nesterov)
void launch_nesterov(hipStream_t st, const ManiDesc &m, int mode, int restart, int skip_lo, int skip_hi, double alpha,
                     double gamma, double *X, double *V, double *Y, double *XPrev, double *Yloc,
                     const double *Xloc);

// ---- dense preconditioner apply: Z = R * Minv (Minv symmetric, k x k, leading dimension ldm) -------------
void launch_dense_apply(hipStream_t st, int r, int k, int ldm, const double *Minv, Buf2 R, double *Z,
                        const double *p2, int np2, Gate g);

// ---- tCG / RTR scalar+vector kernels --------------------------------------------------------------------
int vec_grid(long nelem);
// start-of-solve values of the control block, written by k_rtr_init itself when enable != 0 (the two kernels before
// it then run with a null Gate: they read the start point from buffer 0 and no stamps)
struct CtlInit {
  int enable = 0;
  double tol = 0, Delta = 0, maxDelta = 0;
  int max_outer = 0, stop_on_accept = 0, max_inner = 0;
};
// tcg_sync (optional): the grid-step counters of the one-launch tCG run queued behind this kernel, zeroed here
void launch_rtr_init(hipStream_t st, const double *pA, int npA, const double *pB, int npB, SolverCtl *ctl,
                     HostFlags *hf, int seq, CtlInit ci = CtlInit(), unsigned *tcg_sync = nullptr, int nsync = 0);
void launch_tcg_begin(hipStream_t st, long nelem, Buf2 grad, double *eta, double *Heta, double *res,
                      SolverCtl *ctl, int seq);
void launch_tcg_init(hipStream_t st, long nelem, const double *z, const double *p3, int np3, double *delta,
                     SolverCtl *ctl, int seq);
void launch_tcg_update1(hipStream_t st, long nelem, const double *delta, const double *Hd, double *eta,
                        double *Heta, double *res, const double *p1, int np1, double *p2, SolverCtl *ctl,
                        HostFlags *hf, int seq, int iter, int r = 1, SpFold sf = SpFold());
void launch_tcg_update2(hipStream_t st, long nelem, const double *z, double *delta, const double *p3, int np3,
                        SolverCtl *ctl, HostFlags *hf, int seq, int iter);
void launch_rtr_decide(hipStream_t st, const double *pA, int npA, const double *pB, int npB, const double *pC,
                       int npC, SolverCtl *ctl, HostFlags *hf, int seq, unsigned *tcg_sync = nullptr, int nsync = 0);

// ---- plain vector helpers ------------------------------------------------------------------------------
void launch_axpby(hipStream_t st, long nelem, double a, const double *x, double b, const double *y, double *out);
// out[c] = sum_i partials[i*stride + c], i < np, c < count
void launch_sum_partials(hipStream_t st, const double *partials, int np, int stride, int count, double *out);
void launch_dot(hipStream_t st, long nelem, const double *x, const double *y, double *partials);
void launch_gather_cols(hipStream_t st, int r, int ncols, const int *src_col, const double *X, double *out);
void launch_scatter_cols(hipStream_t st, int r, int ncols, const int *dst_col, const double *in, double *X);
// out[2a] = |A[:, cs[a]:cs[a+1]]|^2, out[2a+1] = <A, B> over the same columns (B may be null)
void launch_block_dots(hipStream_t st, int r, int nagents, const int *col_start, const double *A, const double *B,
                       double *out);

// ---- certification ------------------------------------------------------------------------------------
// Lambda blocks: L_i = sym( (XQ)_i^T X_i ) per Stiefel block (d x d), spheres: x^T (xq)
void launch_lambda_blocks(hipStream_t st, const ManiDesc &m, const double *X, const double *XQ, double *Lblk);
// Lanczos helpers: partial h = V^T w over nv basis vectors (nv per block), w -= V h, y -= shift x
void launch_lanczos_proj(hipStream_t st, int n, int nv, const double *V, const double *w, double *partials);
void launch_lanczos_sub(hipStream_t st, int n, int nv, const double *V, const double *h, double *w);
// sync-free Lanczos step (cert.hip): subtraction with the partial sums in its prologue, coefficient store, next vector
constexpr int kLanczosFuseParts = 256;
void launch_lanczos_sub_sum(hipStream_t st, int n, int nv, const double *V, const double *partials, int npart,
                            double *hout, double *w, double *dotpart);
void launch_lanczos_keep(hipStream_t st, int nv, const double *h, double *hout);
// a new direction whose norm is below this fraction of |S v_j| is rounding noise of the orthogonalisation, not a
// direction: the Krylov space is exhausted (an invariant subspace) and the basis continues with a fresh vector
constexpr double kLanczosDead = 1e-10;
void launch_lanczos_next(hipStream_t st, int n, const double *partials, int npart, double *beta_out, int *flag,
                         const double *w, double *vnext, const double *hrow, int nv);
void launch_scale_shift(hipStream_t st, int n, double shift, const double *x, double *y);
void launch_scale(hipStream_t st, int n, const double *alpha_dev_inv_sqrt /*device: w /= sqrt(*p)*/,
                  const double *w, double *out);

// ---- fused SE-layout solver kernels (solver_fused.hip): 8 lanes per pose, three launches per tCG iteration ----
bool fused_supported(const ManiDesc &m);
bool group_supported(const ManiDesc &m);    // 8-lanes-per-pose rgrad / retract / Nesterov kernels usable
int fused_pose_blocks(const ManiDesc &m);   // partial slots written by hess / finish
int fused_nsplit(const ManiDesc &m);        // row slices of the dense preconditioner product
int fused_precond_grid(const ManiDesc &m);  // partial slots written by precond
// Minv == nullptr: the kernel does the step / vector updates only (the sparse preconditioner follows as its own
// launches); it then writes fused_update_grid(m) partial slots, and finish is called with nsplit = 1
int fused_update_grid(const ManiDesc &m);
// returns the number of <d, H d> partials written (p1)
int launch_fused_hess(hipStream_t st, const ManiDesc &m, const CsrDev &Q, const double *z, const double *d_old,
                       double *d_new, Buf2 X, Buf2 S, double *Hd, const double *p3, int np3, double *p1,
                       SolverCtl *ctl, int seq, int iter,
                      const BsrDev *Ab = nullptr /* block-CSR copy of Q: 8-lanes-per-pose kernel */);
void launch_fused_precond(hipStream_t st, const ManiDesc &m, int ldm, const double *Minv, Buf2 grad,
                          const double *delta, const double *Hd, double *eta, double *Heta, const double *res_old,
                          double *res_new, double *Zpart, const double *p1, int np1, double *p2, SolverCtl *ctl,
                          HostFlags *hf, int seq, int iter, int first, SpFold sf = SpFold());
// B + C in one launch (dense preconditioner): returns the number of <z, r> partial slots written to p3
int fused_pc_blocks(const ManiDesc &m);
// ONE launch per tCG run (k_tcg_run, solver_fused.hip): the dense one-launch B + C form's sizes with n / 2 workgroups
// co-resident (cus = CUs of the device) and an even number of poses per k_fused_hess workgroup; max_rows_nnz: the most
// CSR entries any workgroup's 2 (d+1) matrix rows hold.  tcg_run_sync_words: unsigned words of its grid-step counters.
bool tcg_run_supported(const ManiDesc &m, int ldm, int cus, int max_rows_nnz);
int tcg_run_sync_words();
extern std::atomic<int> g_tcg_run_fault;  // test hook: that many of the next runs lose workgroup 0 before the first step
extern std::atomic<int> g_tcg_run_fault_skip;  // ... after this many launches of the run kernel that pass untouched
int tcg_run_max_rows_nnz(const ManiDesc &m, const int *rowptr);  // host CSR row pointers -> the figure above
// returns the number of workgroups (= <z, r> and pC partial pairs), < 0 when the launch is refused
int launch_tcg_run(hipStream_t st, const ManiDesc &m, int ldm, const double *Minv, const CsrDev &Q, Buf2 grad, Buf2 X,
                   Buf2 S, double *d0, double *d1, double *Hd, double *eta, double *Heta, double *z, double *p1r,
                   double *p3, double *pC, unsigned *sync, SolverCtl *ctl, HostFlags *hf, int seq);
bool fused_pc_preferred(const ManiDesc &m, int ldm);  // sizes at which it beats B + C
bool fused_pc_ready(const ManiDesc &m, int ldm);      // the current device grants the kernel its dynamic LDS
int launch_fused_pc(hipStream_t st, const ManiDesc &m, int ldm, const double *Minv, Buf2 grad, Buf2 X,
                    const double *delta, const double *Hd, double *eta, double *Heta, const double *res_old,
                    double *res_new, double *z, const double *p1, int np1, double *p3, SolverCtl *ctl, HostFlags *hf,
                    int seq, int iter, int first, double *pC = nullptr /* set: the launch that ends a tCG run retracts */);
void launch_fused_finish(hipStream_t st, const ManiDesc &m, Buf2 X, const double *Zpart, const double *res,
                         double *z, const double *p2, int np2, double *p3, SolverCtl *ctl, HostFlags *hf, int seq,
                         int iter, int first, int nsplit = -1 /* -1: fused_nsplit(m) */, SpFold sf = SpFold(),
                         double *zraw = nullptr /* the unprojected P r, kept for a rejected step */);
// group-style (8 lanes per pose) rgrad / retract / Nesterov; return the number of partial slots written
int launch_fused_grad_bsr(hipStream_t st, int r, int d, const BsrDev &A, Buf2 X, const double *G, Buf2 EG, Buf2 RG, Buf2 S,
                          int sel, double *pA, double *pB, double *posenorm, Gate g,
                          const int *agent_start = nullptr, int agents = 0, int *wg_per_agent = nullptr);
int launch_fused_grad(hipStream_t st, const ManiDesc &m, const CsrDev &Q, Buf2 X, const double *G, Buf2 EG, Buf2 RG,
                      Buf2 Sblk, int sel, double *pA, double *pB, double *posenorm, Gate g);
int launch_g_rgrad(hipStream_t st, const ManiDesc &m, Buf2 X, Buf2 EG, Buf2 RG, Buf2 Sblk, int sel, double *partials,
                   double *posenorm, Gate g);
void launch_ctl_init(hipStream_t st, SolverCtl *c, double tol, double Delta, double maxDelta, int max_outer,
                     int stop_on_accept, int max_inner);
void launch_eval_finish(hipStream_t st, int R, const int *pose_start, const double *posenorm, const double *pA,
                        int npA, EvalOut *out_dev, int seq, double *split_scratch = nullptr, int nposes = 0,
                        const double *agent_partials = nullptr, int wg_per_agent = 0);
int eval_split_doubles();
int launch_g_retract(hipStream_t st, const ManiDesc &m, Buf2 X, const double *V, double alpha, Buf2 out, int selOut,
                     Buf2 grad, const double *HV, double *partials, Gate g);
void launch_g_nesterov(hipStream_t st, const ManiDesc &m, int mode, int restart, int skip_lo, int skip_hi,
                       double alpha, double gamma, double *X, double *V, double *Y, double *XPrev, double *Yloc,
                       Buf2 Xloc, const SolverCtl *ctl, double *inner_Yloc = nullptr);

#if defined(__HIPCC__)
// ---- wave-level sums on DPP row operations (device code only) ----
// Moves inside the 16-lane rows (xor 1, xor 2, mirror of 8, mirror of 16) instead of __shfl_xor / ds_bpermute, which
// goes through the LDS crossbar; the four row sums are then read as scalars.  Fixed order => reproducible.
template <int CTRL>
__device__ __forceinline__ double dpp_move(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane_f64(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}
// sum over the 64 lanes, same value in every lane
__device__ __forceinline__ double wave_sum_dpp(double v) {
  v += dpp_move<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_move<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_move<0x141>(v);  // row_half_mirror
  v += dpp_move<0x140>(v);  // row_mirror
  return (readlane_f64(v, 0) + readlane_f64(v, 16)) + (readlane_f64(v, 32) + readlane_f64(v, 48));
}
#endif

}  // namespace dcora
