// extern "C" surface of libdcora_hip.so (declared in include/dcora_hip.h).
#include <cstdio>
#include <cstring>
#include <new>

#include "../../include/dcora_hip.h"
#include "cert.h"
#include "device_chol.h"
#include "device_problem.h"
#include "host_graph.h"
#include "ra_rbcd.h"
#include "rbcd.h"
#include "precond_cache.h"
#include "exchange.h"
#include "host_robust.h"
#include "robust.h"
#include "round.h"

namespace dcora {
const std::string &get_last_error();
}
using namespace dcora;

struct dcora_problem_s {
  DeviceProblem p;
};
struct dcora_csr_s {
  HostCsr m;
};
struct dcora_dataset_s {
  HostDataset ds;
};
struct dcora_rbcd_s {
  RbcdSession s;
  std::vector<int> sel_trace;
};
struct dcora_exchange_s {
  Exchange e;
};

namespace {
HostCsr view_csr(int n, const int *rp, const int *ci, const double *v) {
  HostCsr A;
  A.n = n;
  A.ncols = n;
  A.rp.assign(rp, rp + n + 1);
  A.ci.assign(ci, ci + rp[n]);
  A.v.assign(v, v + rp[n]);
  return A;
}
std::vector<PoseMeas> view_meas(int d, int m, const int *ids, const double *vals) {
  const int stride = d * d + d + 3;
  std::vector<PoseMeas> out(m);
  for (int k = 0; k < m; ++k) {
    PoseMeas &e = out[k];
    e.r1 = ids[4 * k];
    e.p1 = ids[4 * k + 1];
    e.r2 = ids[4 * k + 2];
    e.p2 = ids[4 * k + 3];
    const double *q = vals + (size_t)k * stride;
    for (int i = 0; i < d * d; ++i) e.R[i] = q[i];
    for (int i = 0; i < d; ++i) e.t[i] = q[d * d + i];
    e.kappa = q[d * d + d];
    e.tau = q[d * d + d + 1];
    e.weight = q[d * d + d + 2];
  }
  return out;
}
int bad(const char *msg) {
  set_last_error(msg);
  return DCORA_ERR_BAD_ARG;
}
// DCORA_LAYOUT_SE names a pose graph: it cannot hold unit spheres or landmarks
bool layout_ok(const dcora_dims *dims) {
  return dims->layout >= DCORA_LAYOUT_AUTO && dims->layout <= DCORA_LAYOUT_RA &&
         !(dims->layout == DCORA_LAYOUT_SE && (dims->l != 0 || dims->b != 0));
}
}  // namespace

#define DCORA_TRY try {
#define DCORA_CATCH                                   \
  }                                                   \
  catch (const std::bad_alloc &) {                    \
    set_last_error("host allocation failed");         \
    return DCORA_ERR_HIP;                             \
  }                                                   \
  catch (const std::exception &e) {                   \
    set_last_error(std::string("exception: ") + e.what()); \
    return DCORA_ERR_HIP;                             \
  }

extern "C" {

const char *dcora_status_string(int s) {
  switch (s) {
    case DCORA_OK: return "ok";
    case DCORA_ERR_BAD_ARG: return "bad argument";
    case DCORA_ERR_NO_DEVICE: return "no HIP device (no CPU fallback)";
    case DCORA_ERR_HIP: return "HIP runtime error";
    case DCORA_ERR_NOT_PD: return "matrix not positive definite";
    case DCORA_ERR_NO_CONVERGENCE: return "eigensolver did not converge";
    case DCORA_ERR_NO_PRECONDITIONER: return "preconditioner missing";
    case DCORA_ERR_IO: return "I/O error";
    case DCORA_ERR_UNSUPPORTED: return "unsupported configuration";
  }
  return "unknown";
}
const char *dcora_last_error(void) { return get_last_error().c_str(); }
int dcora_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

void dcora_ropt_params_default(dcora_ropt_params *p) {
  p->method = 0;
  p->verbose = 0;
  p->gradnorm_tol = 1e-2;
  p->RGD_stepsize = 1e-3;
  p->RGD_use_preconditioner = 1;
  p->RTR_iterations = 3;
  p->RTR_tCG_iterations = 50;
  p->RTR_initial_radius = 100;
}

// ---- problem ----------------------------------------------------------------------------------------------
int dcora_problem_create(const dcora_dims *dims, const int *rowptr, const int *colidx, const double *vals,
                         const double *G, double precond_reg, int device, dcora_problem_t *out) {
  if (!dims || !rowptr || !colidx || !vals || !out) return bad("null argument");
  if (!layout_ok(dims)) return bad("dims: layout SE needs l = b = 0");
  DCORA_TRY
  const int k = (dims->d + 1) * dims->n + dims->l + dims->b;
  dcora_problem_s *h = new dcora_problem_s;
  const int rc = h->p.init(*dims, view_csr(k, rowptr, colidx, vals), G, precond_reg, device, nullptr);
  if (rc) {
    delete h;
    return rc;
  }
  *out = h;
  return DCORA_OK;
  DCORA_CATCH
}
int dcora_problem_destroy(dcora_problem_t p) {
  delete p;
  return DCORA_OK;
}
int dcora_problem_set_linear_term(dcora_problem_t p, const double *G) { return p ? p->p.set_G_host(G) : bad("null"); }
int dcora_problem_cost(dcora_problem_t p, const double *X, double *f) { return p ? p->p.cost(X, f) : bad("null"); }
int dcora_problem_eucgrad(dcora_problem_t p, const double *X, double *out) {
  return p ? p->p.eucgrad(X, out) : bad("null");
}
int dcora_problem_riegrad(dcora_problem_t p, const double *X, double *out, double *norm) {
  return p ? p->p.riegrad(X, out, norm) : bad("null");
}
int dcora_problem_hessvec(dcora_problem_t p, const double *X, const double *V, double *out) {
  return p ? p->p.hessvec(X, V, out) : bad("null");
}
int dcora_debug_hessvec_solver_form(dcora_problem_t p, const double *X, const double *V, double *out, double *dots) {
  return p ? p->p.hessvec_solver_form(X, V, out, dots) : bad("null");
}
int dcora_problem_precondition(dcora_problem_t p, const double *X, const double *V, double *out) {
  return p ? p->p.precondition(X, V, out) : bad("null");
}
int dcora_problem_retract(dcora_problem_t p, const double *X, const double *V, double *out) {
  return p ? p->p.retract(X, V, out) : bad("null");
}
int dcora_problem_tangent_project(dcora_problem_t p, const double *X, const double *V, double *out) {
  return p ? p->p.tangent_project(X, V, out) : bad("null");
}
int dcora_problem_escape_saddle(dcora_problem_t p, const double *Xopt, double theta, const double *v, double gtol,
                                double pgtol, int is_second_order, double *Xout, int *success) {
  if (!p) return bad("null");
  DCORA_TRY
  return p->p.escape_saddle(Xopt, theta, v, gtol, pgtol, is_second_order != 0, Xout, success);
  DCORA_CATCH
}
int dcora_manifold_project(const dcora_dims *dims, const double *M, double *out, int device) {
  if (!dims || !M || !out) return bad("null argument");
  if (!layout_ok(dims)) return bad("dims: layout SE needs l = b = 0");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    set_last_error("no HIP device available: libdcora_hip has no CPU fallback");
    return DCORA_ERR_NO_DEVICE;
  }
  DCORA_HIP(hipSetDevice(device));
  const ManiDesc m = make_mani(*dims);
  const size_t N = (size_t)m.r * m.k;
  DevBuf<double> a, b;
  DCORA_HIP(a.alloc(N));
  DCORA_HIP(b.alloc(N));
  DCORA_HIP(hipMemcpy(a.p, M, N * sizeof(double), hipMemcpyHostToDevice));
  launch_polar(nullptr, m, 1.0, a.p, 0.0, nullptr, 0.0, nullptr, b.p);
  DCORA_HIP(hipDeviceSynchronize());
  DCORA_HIP(hipMemcpy(out, b.p, N * sizeof(double), hipMemcpyDeviceToHost));
  return DCORA_OK;
}
int dcora_optimizer_optimize(dcora_problem_t p, const dcora_ropt_params *params, const double *X0, double *Xout,
                             dcora_ropt_result *result) {
  if (!p || !params || !X0 || !Xout) return bad("null argument");
  DCORA_TRY
  return p->p.optimize(*params, X0, Xout, result);
  DCORA_CATCH
}
int dcora_problem_time_qapply(dcora_problem_t p, int reps, double *avg_ms, double *bytes) {
  return p ? p->p.time_qapply(reps, avg_ms, bytes) : bad("null");
}

// the Q-apply of `count` problems in turn on one stream: with distinct (Q, X, Y) sets whose bytes add up to more than
// the 256 MiB Infinity Cache every launch streams from HBM
int dcora_problem_time_qapply_rotating(const dcora_problem_t *ps, int count, int reps, double *avg_ms) {
  if (!ps || count < 1 || !avg_ms) return bad("null");
  for (int i = 0; i < count; ++i)
    if (!ps[i]) return bad("null problem");
  DCORA_TRY
  DeviceProblem &P0 = ps[0]->p;
  DCORA_HIP(hipSetDevice(P0.device));
  std::vector<hipStream_t> keep(count);
  for (int i = 0; i < count; ++i) {
    DCORA_HIP(hipStreamSynchronize(ps[i]->p.st));
    keep[i] = ps[i]->p.st;
    ps[i]->p.st = P0.st;
  }
  hipEvent_t e0, e1;
  DCORA_HIP(hipEventCreate(&e0));
  DCORA_HIP(hipEventCreate(&e1));
  for (int i = 0; i < count; ++i) ps[i]->p.enqueue_egrad(ps[i]->p.X0.p, ps[i]->p.EG0.p, nullptr);
  DCORA_HIP(hipEventRecord(e0, P0.st));
  for (int i = 0; i < reps; ++i) {
    DeviceProblem &P = ps[i % count]->p;
    P.enqueue_egrad(P.X0.p, P.EG0.p, nullptr);
  }
  DCORA_HIP(hipEventRecord(e1, P0.st));
  DCORA_HIP(hipEventSynchronize(e1));
  float ms = 0;
  DCORA_HIP(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  for (int i = 0; i < count; ++i) ps[i]->p.st = keep[i];
  *avg_ms = (double)ms / reps;
  return DCORA_OK;
  DCORA_CATCH
}
int dcora_problem_qapply_info(dcora_problem_t p, double *info) {
  if (!p || !info) return bad("null");
  const DeviceProblem &P = p->p;
  info[0] = P.has_bsr ? 1 : 0;
  info[1] = (double)P.Q.nnz;
  info[2] = P.has_bsr ? (double)P.Qb.nblocks : 0.0;
  // bytes of the matrix in the form the kernel reads
  info[3] = P.has_bsr ? P.Qb.nblocks * (8.0 * (P.m.d + 1) * (P.m.d + 1) + 4.0) + 4.0 * (P.m.n + 1)
                      : 12.0 * P.Q.nnz + 4.0 * (P.m.k + 1);
  return DCORA_OK;
}

// how the dense tCG iteration runs on this problem: info[0] = 0 three launches or the sparse preconditioner, 1 = A + (B and C
// in one launch), 2 = the whole tCG run in ONE launch (k_tcg_run; falls back to 1 for good after a run that gave up)
int dcora_problem_solver_info(dcora_problem_t p, double *info) {
  if (!p || !info) return bad("null");
  const DeviceProblem &P = p->p;
  info[0] = !P.use_pc() ? 0 : (P.tcg_run_ok ? 2 : 1);
  return DCORA_OK;
}
int dcora_debug_tcg_run_fault(int runs) {
  g_tcg_run_fault_skip.store(0);
  g_tcg_run_fault.store(runs < 0 ? 0 : runs);
  return DCORA_OK;
}
int dcora_debug_tcg_run_fault_at(int skip, int runs) {
  g_tcg_run_fault_skip.store(skip < 0 ? 0 : skip);
  g_tcg_run_fault.store(runs < 0 ? 0 : runs);
  return DCORA_OK;
}
int dcora_problem_time_precond(dcora_problem_t p, int reps, double *avg_ms, double *bytes) {
  return p ? p->p.time_precond(reps, avg_ms, bytes) : bad("null");
}
int dcora_problem_precond_info(dcora_problem_t p, double *info) {
  if (!p || !info) return bad("null");
  const DeviceProblem &P = p->p;
  info[0] = !P.has_precond ? 0 : (P.sparse_precond ? 2 : 1);
  info[1] = !P.has_precond ? 0 : (P.sparse_precond ? P.sp.launches() : 1);
  info[2] = (double)P.precond_nnzL;
  info[3] = P.precond_setup_ms;
  info[4] = !P.has_precond ? 0 : (P.sparse_precond ? P.sp.weights_per_apply : (double)P.m.k * P.m.k);
  return DCORA_OK;
}

// debug / test hook (not part of the public header): a digest of the stored weights of a problem's sparse
// preconditioner image, read back from the device: {count, sum, sum of |w|, sum of w (i mod 97 + 1)} -- the last one
// moves when a weight lands in another place.  Compares the ways the weights can be formed (device fill, streamed host
// fill, one-piece upload).
extern "C" int dcora_debug_sparse_weights_digest(dcora_problem_t p, double *out4) {
  if (!p || !out4) return bad("null");
  const DeviceProblem &P = p->p;
  if (!P.sparse_precond || !P.sp.im) return bad("the problem has no sparse preconditioner");
  const DevBuf<double> &v = P.sp.im->vals;
  std::vector<double> h(v.n);
  DCORA_HIP(hipSetDevice(P.device));
  DCORA_HIP(hipMemcpy(h.data(), v.p, v.n * sizeof(double), hipMemcpyDeviceToHost));
  long double s0 = 0, s1 = 0, s2 = 0;
  for (size_t i = 0; i < h.size(); ++i) {
    s0 += h[i];
    s1 += std::fabs(h[i]);
    s2 += h[i] * (double)(i % 97 + 1);
  }
  out4[0] = (double)h.size();
  out4[1] = (double)s0;
  out4[2] = (double)s1;
  out4[3] = (double)s2;
  return DCORA_OK;
}

// A CERTIFIED lower bound of the smallest eigenvalue of a matrix the PSD test has accepted.  Lanczos (full
// re-orthogonalisation) on M^-1, M = S + eta I, through the sparse Cholesky factor the test computes (host), gives a
// Ritz value theta <= theta_max(M^-1) -- so 1 / theta - eta errs UPWARD and is only an estimate.  The candidate bound
// 1 / (theta + |beta_m s_m|) - eta (Ritz residual added) is then VERIFIED the way the certificate itself is
// (ref src/DCORA_utils.cpp:1737-1747): S - lambda I must have a Cholesky factorisation.  What is returned is the
// verified shift; when the verification fails the only certified figure, -eta, is returned.
namespace {
// largest eigenvalue of the symmetric tridiagonal (a, b) of order m by bisection on the Sturm count
double tridiag_largest(const std::vector<double> &a, const std::vector<double> &b, int m) {
  double lo = a[0], hi = a[0];
  for (int i = 0; i < m; ++i) {
    const double rad = (i > 0 ? std::fabs(b[i - 1]) : 0.0) + (i + 1 < m ? std::fabs(b[i]) : 0.0);
    lo = std::min(lo, a[i] - rad);
    hi = std::max(hi, a[i] + rad);
  }
  auto count_below = [&](double x) {  // eigenvalues < x
    int c = 0;
    double d = a[0] - x;
    if (d < 0) ++c;
    for (int i = 1; i < m; ++i) {
      if (d == 0) d = 1e-300;
      d = a[i] - x - b[i - 1] * b[i - 1] / d;
      if (d < 0) ++c;
    }
    return c;
  };
  for (int it = 0; it < 200 && hi - lo > 1e-15 * std::max(std::fabs(hi), std::fabs(lo)); ++it) {
    const double mid = 0.5 * (lo + hi);
    if (count_below(mid) >= m) hi = mid; else lo = mid;
  }
  return 0.5 * (lo + hi);
}
}  // namespace
int dcora_cert_lambda_min_certified(int k, const int *rp, const int *ci, const double *v, double eta, int block,
                                    int max_iterations, double *lambda_min, int *iterations) {
  if (!rp || !ci || !v || !lambda_min) return bad("null argument");
  DCORA_TRY
  const HostCsr S = view_csr(k, rp, ci, v);
  SparseChol chol;
  if (!chol.factor(csr_shift_diag(S, eta), block)) {
    set_last_error("lambda_min_certified: S + eta I is not positive definite (the certificate was not accepted)");
    return DCORA_ERR_NOT_PD;
  }
  const int mmax = std::max(2, std::min(max_iterations, k));
  std::vector<std::vector<double>> V;
  std::vector<double> alpha, beta, w((size_t)k);
  std::vector<double> q((size_t)k);
  uint64_t sdd = 0x9E3779B97F4A7C15ull;
  double n2 = 0;
  for (int i = 0; i < k; ++i) {
    sdd = sdd * 6364136223846793005ull + 1442695040888963407ull;
    q[i] = (double)(sdd >> 11) / 9007199254740992.0 - 0.5;
    n2 += q[i] * q[i];
  }
  for (double &t : q) t /= std::sqrt(n2);
  double lam = 0, prev = 1e300, th_last = 0, bn_last = 0;
  int j = 0;
  for (; j < mmax; ++j) {
    V.push_back(q);
    chol.solve_vec(q.data(), w.data());  // w = M^-1 q
    double a = 0;
    for (int i = 0; i < k; ++i) a += q[i] * w[i];
    alpha.push_back(a);
    for (int pass = 0; pass < 2; ++pass)  // full re-orthogonalisation, twice
      for (const std::vector<double> &u : V) {
        double c = 0;
        for (int i = 0; i < k; ++i) c += u[i] * w[i];
        for (int i = 0; i < k; ++i) w[i] -= c * u[i];
      }
    double bn = 0;
    for (int i = 0; i < k; ++i) bn += w[i] * w[i];
    bn = std::sqrt(bn);
    const int m = (int)alpha.size();
    const bool check = (m % 5 == 0) || bn < 1e-14 * std::fabs(a) || j + 1 == mmax;
    if (check) {
      const double th = tridiag_largest(alpha, beta, m);
      th_last = th;
      bn_last = bn;
      lam = 1.0 / th - eta;
      if (std::fabs(prev - lam) <= 1e-3 * std::fabs(lam) + 1e-15 || bn < 1e-14 * std::fabs(a)) {
        ++j;
        break;
      }
      prev = lam;
    }
    beta.push_back(bn);
    for (int i = 0; i < k; ++i) q[i] = w[i] / bn;
  }
  // Ritz residual |beta_m s_m|: s = eigenvector of the tridiagonal for th_last, by its three-term recurrence
  double resid = 0;
  {
    const int m = (int)alpha.size();
    std::vector<double> sv((size_t)m, 0.0);
    sv[0] = 1.0;
    double nrm = 1.0;
    for (int i = 0; i + 1 < m; ++i) {
      const double b = (i < (int)beta.size() && beta[i] != 0.0) ? beta[i] : 1e-300;
      double t = (th_last - alpha[i]) * sv[i];
      if (i > 0) t -= beta[i - 1] * sv[i - 1];
      sv[i + 1] = t / b;
      nrm += sv[i + 1] * sv[i + 1];
      if (nrm > 1e200) {  // rescale
        for (int u = 0; u <= i + 1; ++u) sv[u] *= 1e-100;
        nrm *= 1e-200;
      }
    }
    resid = std::fabs(bn_last * sv[m - 1]) / std::sqrt(nrm);
  }
  const double cand = 1.0 / (th_last + resid) - eta;
  // verification: S - shift I factors  <=>  lambda_min(S) > shift (up to the rounding of the factorisation)
  double shift = cand - 1e-3 * std::fabs(cand) - 1e-13;
  if (!(shift > -eta)) shift = -eta;
  double bound = -eta;
  if (shift > -eta) {
    SparseChol verify;
    if (verify.factor(csr_shift_diag(S, -shift), block)) bound = shift;
  }
  (void)lam;
  *lambda_min = bound;
  if (iterations) *iterations = j;
  return DCORA_OK;
  DCORA_CATCH
}

int dcora_cert_suboptimality_gap(const dcora_dims *dims, const double *X, double lambda_lower_bound, double *gap,
                                 double *n_eff) {
  if (!dims || !X || !gap) return bad("null argument");
  if (!layout_ok(dims)) return bad("dims: layout SE needs l = b = 0");
  const ManiDesc m = make_mani(*dims);
  const int r = m.r;
  double rot = 0;
  for (int i = 0; i < m.n; ++i)
    for (int c = 0; c < m.d; ++c)
      for (int t = 0; t < r; ++t) {
        const double x = X[(size_t)(m.rot_col(i) + c) * r + t];
        rot += x * x;
      }
  for (int i = 0; i < m.l; ++i)
    for (int t = 0; t < r; ++t) {
      const double x = X[(size_t)m.sphere_col(i) * r + t];
      rot += x * x;
    }
  const int ne = m.num_euc();
  std::vector<double> mean((size_t)r, 0.0);
  for (int e = 0; e < ne; ++e)
    for (int t = 0; t < r; ++t) mean[t] += X[(size_t)m.euc_col(e) * r + t];
  for (int t = 0; t < r; ++t) mean[t] /= std::max(ne, 1);
  double euc = 0;
  for (int e = 0; e < ne; ++e)
    for (int t = 0; t < r; ++t) {
      const double x = X[(size_t)m.euc_col(e) * r + t] - mean[t];
      euc += x * x;
    }
  const double tr = rot + euc;
  if (n_eff) *n_eff = tr;
  *gap = 0.5 * std::max(0.0, -lambda_lower_bound) * tr;
  return DCORA_OK;
}

int dcora_precond_cache_info(double *info4) {
  if (!info4) return bad("null");
  precond_cache_stats(info4);
  return DCORA_OK;
}
int dcora_precond_cache_clear(void) {
  precond_cache_clear();
  return DCORA_OK;
}

// ---- CSR handles --------------------------------------------------------------------------------------------
int dcora_csr_info(dcora_csr_t m, int *n, int *nnz) {
  if (!m) return bad("null");
  *n = m->m.n;
  *nnz = m->m.nnz();
  return DCORA_OK;
}
int dcora_csr_copy(dcora_csr_t m, int *rp, int *ci, double *v) {
  if (!m) return bad("null");
  std::copy(m->m.rp.begin(), m->m.rp.end(), rp);
  std::copy(m->m.ci.begin(), m->m.ci.end(), ci);
  std::copy(m->m.v.begin(), m->m.v.end(), v);
  return DCORA_OK;
}
int dcora_csr_destroy(dcora_csr_t m) {
  delete m;
  return DCORA_OK;
}

// ---- certification --------------------------------------------------------------------------------------------
int dcora_cert_dual_matrix(const dcora_dims *dims, const double *X, const int *rp, const int *ci, const double *v,
                           int device, dcora_csr_t *S) {
  if (!dims || !X || !rp || !S) return bad("null argument");
  if (!layout_ok(dims)) return bad("dims: layout SE needs l = b = 0");
  DCORA_TRY
  const int k = (dims->d + 1) * dims->n + dims->l + dims->b;
  dcora_csr_s *h = new dcora_csr_s;
  const int rc = device_dual_certificate(*dims, X, view_csr(k, rp, ci, v), device, &h->m);
  if (rc) {
    delete h;
    return rc;
  }
  *S = h;
  return DCORA_OK;
  DCORA_CATCH
}
int dcora_cert_is_psd(int k, const int *rp, const int *ci, const double *v, int block, int *is_psd) {
  DCORA_TRY
  bool psd = false;
  const int rc = host_is_psd(view_csr(k, rp, ci, v), block, &psd);
  *is_psd = psd ? 1 : 0;
  return rc;
  DCORA_CATCH
}
int dcora_cert_prepare(const dcora_dims *dims, const int *rp, const int *ci, int block, int device) {
  if (!dims || !rp || !ci) return bad("null");
  if (!layout_ok(dims)) return bad("dims: layout SE needs l = b = 0");
  const ManiDesc m = make_mani(*dims);
  const int n = m.k;
  if (n <= 0) return bad("empty pattern");
  // the pattern dcora_cert_dual_matrix will hand to the PSD test: Q's entries, the d x d rotation blocks and the
  // unit-sphere diagonal of Lambda (present in Q's pattern unless an entry of Q is structurally zero), every diagonal
  std::vector<int> I, J;
  I.reserve((size_t)rp[n] + (size_t)m.n * m.d * m.d + m.l);
  J.reserve(I.capacity());
  for (int i = 0; i < n; ++i)
    for (int p = rp[i]; p < rp[i + 1]; ++p) {
      I.push_back(i);
      J.push_back(ci[p]);
    }
  for (int i = 0; i < m.n; ++i) {
    const int c = m.rot_col(i);
    for (int a = 0; a < m.d; ++a)
      for (int b = 0; b < m.d; ++b) {
        I.push_back(c + a);
        J.push_back(c + b);
      }
  }
  for (int i = 0; i < m.l; ++i) {
    I.push_back(m.sphere_col(i));
    J.push_back(m.sphere_col(i));
  }
  const std::vector<double> V(I.size(), 1.0);
  const HostCsr A = csr_shift_diag(csr_from_coo(n, n, I, J, V), 1.0);
  return device_chol_prepare(A, block < 1 ? 1 : block, device);
}
int dcora_cert_is_psd_device(int k, const int *rp, const int *ci, const double *v, int block, int device, int *is_psd,
                             double *info8) {
  DCORA_TRY
  bool pd = false;
  const int rc = device_chol_is_pd(view_csr(k, rp, ci, v), block, device, &pd, info8);
  *is_psd = pd ? 1 : 0;
  return rc;
  DCORA_CATCH
}
int dcora_chol_host_selftest(int k, const int *rp, const int *ci, const double *v, int block, int *is_pd,
                             double *resid, double *info4) {
  DCORA_TRY
  if (k > 4096) {
    set_last_error("dcora_chol_host_selftest: dense check, k <= 4096");
    return DCORA_ERR_BAD_ARG;
  }
  const HostCsr A = view_csr(k, rp, ci, v);
  CholSymbolic S;
  chol_symbolic(A, block, &S);
  std::vector<double> F;
  const bool ok = chol_numeric_host(S, v, &F);
  *is_pd = ok ? 1 : 0;
  if (info4) {
    info4[0] = (double)S.pieces.size();
    info4[1] = (double)S.nlev;
    info4[2] = (double)S.arena;
    info4[3] = S.flops;
  }
  if (resid) *resid = 0;
  if (!ok || !resid) return DCORA_OK;
  std::vector<double> L((size_t)k * k, 0.0), M((size_t)k * k, 0.0);
  for (const CholPiece &P : S.pieces) {
    const long long f = (long long)P.c + P.m;
    for (int j = 0; j < P.c; ++j) {
      for (int i = j; i < P.c; ++i) L[(size_t)(P.c0 + i) * k + P.c0 + j] = F[(size_t)(P.off + i * f + j)];
      for (int a = 0; a < P.m; ++a)
        L[(size_t)S.rows[(size_t)P.rows_off + a] * k + P.c0 + j] = F[(size_t)(P.off + (P.c + a) * f + j)];
    }
  }
  for (int io = 0; io < k; ++io)
    for (int p = rp[io]; p < rp[io + 1]; ++p) M[(size_t)S.iperm[io] * k + S.iperm[ci[p]]] = v[p];
  double worst = 0;
  for (int i = 0; i < k; ++i)
    for (int j = 0; j <= i; ++j) {
      double s = 0;
      for (int l = 0; l <= j; ++l) s += L[(size_t)i * k + l] * L[(size_t)j * k + l];
      worst = std::max(worst, std::fabs(s - M[(size_t)i * k + j]));
    }
  *resid = worst;
  return DCORA_OK;
  DCORA_CATCH
}
int dcora_chol_cache_clear(void) {
  chol_cache_clear();
  return DCORA_OK;
}
int dcora_cert_min_eig(int k, const int *rp, const int *ci, const double *v, int max_iterations, double tol, int ncv,
                       unsigned long long seed, int device, double *lambda_min, double *vec, long *num_matvecs) {
  DCORA_TRY
  LanczosResult e;
  const int rc = device_min_eig(view_csr(k, rp, ci, v), max_iterations, tol, ncv, seed, device, &e);
  if (lambda_min) *lambda_min = e.lambda;
  if (vec && (int)e.v.size() == k) std::copy(e.v.begin(), e.v.end(), vec);
  if (num_matvecs) *num_matvecs = e.matvecs;
  return rc;
  DCORA_CATCH
}
int dcora_cert_fast_verification(int k, const int *rp, const int *ci, const double *v, double eta, int block,
                                 int device, int *is_psd, double *theta, double *x, double *lambda_min) {
  DCORA_TRY
  bool psd = false;
  std::vector<double> vec;
  double th = 0, lm = 0;
  long mv = 0;
  const int rc = device_fast_verification(view_csr(k, rp, ci, v), eta, block, device, &psd, &th, &vec, &lm, &mv);
  *is_psd = psd ? 1 : 0;
  if (!psd) {
    if (theta) *theta = th;
    if (lambda_min) *lambda_min = lm;
    if (x && (int)vec.size() == k) std::copy(vec.begin(), vec.end(), x);
  }
  return rc;
  DCORA_CATCH
}

// ---- data feed ------------------------------------------------------------------------------------------------
int dcora_dataset_load_g2o(const char *path, dcora_dataset_t *out) {
  DCORA_TRY
  dcora_dataset_s *h = new dcora_dataset_s;
  std::string err;
  if (!load_g2o(path, h->ds, err)) {
    delete h;
    set_last_error(err);
    return DCORA_ERR_IO;
  }
  *out = h;
  return DCORA_OK;
  DCORA_CATCH
}
int dcora_dataset_create(int d, int n, int m, const int *ids, const double *vals, dcora_dataset_t *out) {
  if ((d != 2 && d != 3) || n < 1 || m < 0 || !out) return bad("bad dataset shape");
  DCORA_TRY
  dcora_dataset_s *h = new dcora_dataset_s;
  h->ds.d = d;
  h->ds.n = n;
  h->ds.meas = view_meas(d, m, ids, vals);
  *out = h;
  return DCORA_OK;
  DCORA_CATCH
}
int dcora_dataset_info(dcora_dataset_t ds, int *d, int *n, int *m) {
  if (!ds) return bad("null");
  *d = ds->ds.d;
  *n = ds->ds.n;
  *m = (int)ds->ds.meas.size();
  return DCORA_OK;
}
int dcora_dataset_copy(dcora_dataset_t h, int *ids, double *vals) {
  if (!h) return bad("null");
  const int d = h->ds.d, stride = d * d + d + 3;
  for (size_t k = 0; k < h->ds.meas.size(); ++k) {
    const PoseMeas &e = h->ds.meas[k];
    ids[4 * k] = e.r1;
    ids[4 * k + 1] = e.p1;
    ids[4 * k + 2] = e.r2;
    ids[4 * k + 3] = e.p2;
    double *q = vals + k * stride;
    for (int i = 0; i < d * d; ++i) q[i] = e.R[i];
    for (int i = 0; i < d; ++i) q[d * d + i] = e.t[i];
    q[d * d + d] = e.kappa;
    q[d * d + d + 1] = e.tau;
    q[d * d + d + 2] = e.weight;
  }
  return DCORA_OK;
}
int dcora_dataset_destroy(dcora_dataset_t ds) {
  delete ds;
  return DCORA_OK;
}
int dcora_dataset_chordal_init(dcora_dataset_t ds, double *T) {
  if (!ds || !T) return bad("null argument");
  DCORA_TRY
  std::vector<double> out;
  if (!chordal_initialization(ds->ds, out)) {
    set_last_error("chordal initialisation: reduced Laplacian not positive definite (disconnected graph?)");
    return DCORA_ERR_NOT_PD;
  }
  std::copy(out.begin(), out.end(), T);
  return DCORA_OK;
  DCORA_CATCH
}
int dcora_dataset_chordal_init_device(dcora_dataset_t ds, int device, double *T) {
  if (!ds || !T) return bad("null argument");
  DCORA_TRY
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    set_last_error("no HIP device available: libdcora_hip has no CPU fallback");
    return DCORA_ERR_NO_DEVICE;
  }
  std::vector<double> out;
  if (!chordal_initialization(ds->ds, out, device_spd_solver(device))) {
    set_last_error("chordal initialisation: reduced Laplacian not positive definite (disconnected graph?)");
    return DCORA_ERR_NOT_PD;
  }
  std::copy(out.begin(), out.end(), T);
  return DCORA_OK;
  DCORA_CATCH
}
int dcora_graph_build_Q_pgo(int d, int n, int agent_id, int m, const int *ids, const double *vals, dcora_csr_t *Q) {
  DCORA_TRY
  dcora_csr_s *h = new dcora_csr_s;
  h->m = build_Q_pgo(d, n, agent_id, view_meas(d, m, ids, vals));
  *Q = h;
  return DCORA_OK;
  DCORA_CATCH
}

// ---- range-aided SLAM data feed (centralised) -----------------------------------------------------------------------
struct dcora_radataset_s {
  HostRADataset ds;
};
int dcora_radataset_load_pyfg(const char *path, dcora_radataset_t *out) {
  if (!path || !out) return bad("null argument");
  DCORA_TRY
  dcora_radataset_s *h = new dcora_radataset_s;
  std::string err;
  if (!load_pyfg(path, h->ds, err)) {
    delete h;
    set_last_error(err);
    return DCORA_ERR_IO;
  }
  *out = h;
  return DCORA_OK;
  DCORA_CATCH
}
// Graph::setMeasurements(const RelativeMeasurements &) of a range-aided graph (ref src/Graph.cpp:374-470): the three
// kinds of measurements as arrays, states numbered as the Graph numbers them (poses 0 .. n - 1, unit spheres 0 .. l - 1 --
// one per range measurement --, landmarks 0 .. b - 1); every state is owned by robot 0 (the centralised agent)
int dcora_radataset_create(int d, int n, int l, int b, int m_pp, const int *pp_ids, const double *pp_vals, int m_pl,
                           const int *pl_ids, const double *pl_vals, int m_rg, const int *rg_ids, const double *rg_vals,
                           const double *gt, dcora_radataset_t *out) {
  if (!out || (m_pp > 0 && (!pp_ids || !pp_vals)) || (m_pl > 0 && (!pl_ids || !pl_vals)) ||
      (m_rg > 0 && (!rg_ids || !rg_vals)))
    return bad("null argument");
  if ((d != 2 && d != 3) || n < 0 || l < 0 || b < 0 || m_pp < 0 || m_pl < 0 || m_rg < 0) return bad("bad dimensions");
  DCORA_TRY
  auto h = std::make_unique<dcora_radataset_s>();
  HostRADataset &ds = h->ds;
  ds.d = d;
  ds.n = n;
  ds.l = l;
  ds.b = b;
  const int w = d * d + d + 3;
  for (int i = 0; i < m_pp; ++i) {
    PoseMeas m;
    m.p1 = pp_ids[2 * i];
    m.p2 = pp_ids[2 * i + 1];
    if (m.p1 < 0 || m.p1 >= n || m.p2 < 0 || m.p2 >= n) return bad("pose-pose measurement: pose out of range");
    const double *v = pp_vals + (size_t)i * w;
    for (int c = 0; c < d; ++c)
      for (int a = 0; a < d; ++a) m.R[c * d + a] = v[c * d + a];
    for (int a = 0; a < d; ++a) m.t[a] = v[d * d + a];
    m.kappa = v[d * d + d];
    m.tau = v[d * d + d + 1];
    m.weight = v[d * d + d + 2];
    ds.pose_pose.push_back(m);
  }
  for (int i = 0; i < m_pl; ++i) {
    PoseLandmarkMeasH m;
    m.i = pl_ids[2 * i];
    m.j = pl_ids[2 * i + 1];
    if (m.i < 0 || m.i >= n || m.j < 0 || m.j >= b) return bad("pose-landmark measurement: state out of range");
    const double *v = pl_vals + (size_t)i * (d + 2);
    for (int a = 0; a < d; ++a) m.t[a] = v[a];
    m.tau = v[d];
    m.weight = v[d + 1];
    ds.pose_landmark.push_back(m);
  }
  for (int i = 0; i < m_rg; ++i) {
    RangeMeasH m;
    m.type1 = rg_ids[5 * i];
    m.i = rg_ids[5 * i + 1];
    m.type2 = rg_ids[5 * i + 2];
    m.j = rg_ids[5 * i + 3];
    m.l = rg_ids[5 * i + 4];
    const bool ok1 = m.type1 == 0 ? (m.i >= 0 && m.i < n) : (m.type1 == 1 && m.i >= 0 && m.i < b);
    const bool ok2 = m.type2 == 0 ? (m.j >= 0 && m.j < n) : (m.type2 == 1 && m.j >= 0 && m.j < b);
    if (!ok1 || !ok2 || m.l < 0 || m.l >= l) return bad("range measurement: state out of range");
    m.range = rg_vals[3 * i];
    m.precision = rg_vals[3 * i + 1];
    m.weight = rg_vals[3 * i + 2];
    ds.ranges.push_back(m);
  }
  ds.gt.assign((size_t)d * ds.k(), 0.0);
  if (gt) std::copy(gt, gt + (size_t)d * ds.k(), ds.gt.begin());
  ds.pose_robot.assign((size_t)n, 0);
  ds.sphere_robot.assign((size_t)l, 0);
  ds.landmark_robot.assign((size_t)b, 0);
  *out = h.release();
  return DCORA_OK;
  DCORA_CATCH
}
// the measurements of a dataset in the arrays dcora_radataset_create takes (sizes: dcora_radataset_info); any may be NULL
int dcora_radataset_copy(dcora_radataset_t h, int *pp_ids, double *pp_vals, int *pl_ids, double *pl_vals, int *rg_ids,
                         double *rg_vals) {
  if (!h) return bad("null");
  const HostRADataset &ds = h->ds;
  const int d = ds.d, w = d * d + d + 3;
  for (size_t i = 0; i < ds.pose_pose.size(); ++i) {
    const PoseMeas &m = ds.pose_pose[i];
    if (pp_ids) {
      pp_ids[2 * i] = m.p1;
      pp_ids[2 * i + 1] = m.p2;
    }
    if (pp_vals) {
      double *v = pp_vals + i * w;
      for (int c = 0; c < d * d; ++c) v[c] = m.R[c];
      for (int a = 0; a < d; ++a) v[d * d + a] = m.t[a];
      v[d * d + d] = m.kappa;
      v[d * d + d + 1] = m.tau;
      v[d * d + d + 2] = m.weight;
    }
  }
  for (size_t i = 0; i < ds.pose_landmark.size(); ++i) {
    const PoseLandmarkMeasH &m = ds.pose_landmark[i];
    if (pl_ids) {
      pl_ids[2 * i] = m.i;
      pl_ids[2 * i + 1] = m.j;
    }
    if (pl_vals) {
      double *v = pl_vals + i * (d + 2);
      for (int a = 0; a < d; ++a) v[a] = m.t[a];
      v[d] = m.tau;
      v[d + 1] = m.weight;
    }
  }
  for (size_t i = 0; i < ds.ranges.size(); ++i) {
    const RangeMeasH &m = ds.ranges[i];
    if (rg_ids) {
      rg_ids[5 * i] = m.type1;
      rg_ids[5 * i + 1] = m.i;
      rg_ids[5 * i + 2] = m.type2;
      rg_ids[5 * i + 3] = m.j;
      rg_ids[5 * i + 4] = m.l;
    }
    if (rg_vals) {
      rg_vals[3 * i] = m.range;
      rg_vals[3 * i + 1] = m.precision;
      rg_vals[3 * i + 2] = m.weight;
    }
  }
  return DCORA_OK;
}
int dcora_radataset_info(dcora_radataset_t h, int *info) {
  if (!h || !info) return bad("null");
  info[0] = h->ds.d;
  info[1] = h->ds.n;
  info[2] = h->ds.l;
  info[3] = h->ds.b;
  info[4] = (int)h->ds.pose_pose.size();
  info[5] = (int)h->ds.pose_landmark.size();
  info[6] = (int)h->ds.ranges.size();
  return DCORA_OK;
}
int dcora_radataset_ground_truth(dcora_radataset_t h, double *gt) {
  if (!h || !gt) return bad("null");
  std::copy(h->ds.gt.begin(), h->ds.gt.end(), gt);
  return DCORA_OK;
}
int dcora_radataset_build_Q(dcora_radataset_t h, dcora_csr_t *Q) {
  if (!h || !Q) return bad("null");
  DCORA_TRY
  dcora_csr_s *c = new dcora_csr_s;
  c->m = build_Q_ra(h->ds);
  *Q = c;
  return DCORA_OK;
  DCORA_CATCH
}
int dcora_radataset_odometry_init(dcora_radataset_t h, unsigned long long seed, double *X0) {
  if (!h || !X0) return bad("null");
  DCORA_TRY
  std::vector<double> x;
  ra_odometry_initialization(h->ds, seed, x);
  std::copy(x.begin(), x.end(), X0);
  return DCORA_OK;
  DCORA_CATCH
}
int dcora_radataset_ownership(dcora_radataset_t h, int *pose_robot, int *sphere_robot, int *landmark_robot) {
  if (!h) return bad("null");
  if (pose_robot) std::copy(h->ds.pose_robot.begin(), h->ds.pose_robot.end(), pose_robot);
  if (sphere_robot) std::copy(h->ds.sphere_robot.begin(), h->ds.sphere_robot.end(), sphere_robot);
  if (landmark_robot) std::copy(h->ds.landmark_robot.begin(), h->ds.landmark_robot.end(), landmark_robot);
  return DCORA_OK;
}
int dcora_radataset_agent_columns(dcora_radataset_t h, int robot, int *dims3, int *own, int *k_a) {
  if (!h || !dims3 || !k_a) return bad("null");
  DCORA_TRY
  std::vector<int> o;
  ra_agent_columns(h->ds, robot, dims3, o);
  *k_a = (int)o.size();
  if (own) std::copy(o.begin(), o.end(), own);
  return DCORA_OK;
  DCORA_CATCH
}
int dcora_graph_extract_agent_blocks(int k, const int *rp, const int *ci, const double *v, int k_a, const int *own,
                                     dcora_csr_t *Qaa, dcora_csr_t *C) {
  if (!rp || !ci || !v || !own || !Qaa || !C) return bad("null argument");
  for (int a = 0; a < k_a; ++a)
    if (own[a] < 0 || own[a] >= k) return bad("extract_agent_blocks: column index out of range");
  DCORA_TRY
  dcora_csr_s *q = new dcora_csr_s, *c = new dcora_csr_s;
  extract_agent_blocks(view_csr(k, rp, ci, v), std::vector<int>(own, own + k_a), &q->m, &c->m);
  *Qaa = q;
  *C = c;
  return DCORA_OK;
  DCORA_CATCH
}
int dcora_radataset_destroy(dcora_radataset_t h) {
  delete h;
  return DCORA_OK;
}
int dcora_graph_precond_regularization(int k, const int *rp, const int *ci, const double *v, int device, double *reg) {
  if (!rp || !ci || !v || !reg) return bad("null argument");
  DCORA_TRY
  return device_precond_regularization(view_csr(k, rp, ci, v), device, reg);
  DCORA_CATCH
}

// ---- RBCD session -----------------------------------------------------------------------------------------------
void dcora_rbcd_options_default(dcora_rbcd_options *o) {
  o->num_robots = 5;
  o->r = 5;
  o->acceleration = 1;
  o->restart_interval = 30;
  dcora_ropt_params_default(&o->local);
  o->rank = 0;
  o->world_size = 1;
  o->device = 0;
  o->stream = nullptr;
}
int dcora_rbcd_create(dcora_dataset_t ds, const dcora_rbcd_options *opt, dcora_rbcd_t *out) {
  if (!ds || !opt || !out) return bad("null argument");
  DCORA_TRY
  dcora_rbcd_s *h = new dcora_rbcd_s;
  const int rc = h->s.init(ds->ds, *opt);
  if (rc) {
    delete h;
    return rc;
  }
  *out = h;
  return DCORA_OK;
  DCORA_CATCH
}
int dcora_rbcd_destroy(dcora_rbcd_t s) {
  delete s;
  return DCORA_OK;
}
int dcora_rbcd_set_X(dcora_rbcd_t s, const double *X) { return s ? s->s.set_X(X) : bad("null"); }
int dcora_rbcd_get_X(dcora_rbcd_t s, double *X) { return s ? s->s.get_X(X) : bad("null"); }
int dcora_rbcd_iterate(dcora_rbcd_t s, int selected, double *cost2, double *gradnorm, double *block_norms,
                       int *next_selected) {
  if (!s) return bad("null");
  DCORA_TRY
  return s->s.iterate(selected, cost2, gradnorm, block_norms, next_selected);
  DCORA_CATCH
}
int dcora_rbcd_run(dcora_rbcd_t s, int max_iters, double rgrad_tol, int *iters_done, double *cost2_trace,
                   double *gradnorm_trace, int *selected_trace) {
  if (!s) return bad("null");
  DCORA_TRY
  int selected = 0, it = 0;
  for (; it < max_iters; ++it) {
    double c2 = 0, gn = 0;
    int nxt = selected;
    const int rc = s->s.iterate(selected, &c2, &gn, nullptr, &nxt);
    if (rc) return rc;
    if (cost2_trace) cost2_trace[it] = c2;
    if (gradnorm_trace) gradnorm_trace[it] = gn;
    if (selected_trace) selected_trace[it] = selected;
    if (gn < rgrad_tol) {
      ++it;
      break;
    }
    selected = nxt;
  }
  if (iters_done) *iters_done = it;
  return DCORA_OK;
  DCORA_CATCH
}
int dcora_rbcd_iterate_set(dcora_rbcd_t s, const int *set, int count, int allow_adjacent) {
  if (!s) return bad("null");
  DCORA_TRY
  return s->s.iterate_set(set, count, allow_adjacent);
  DCORA_CATCH
}
int dcora_rbcd_set_acceleration(dcora_rbcd_t s, int acceleration) {
  if (!s) return bad("null");
  return s->s.set_acceleration(acceleration != 0);
}
int dcora_rbcd_agent_colours(dcora_rbcd_t s, int *colours, int *ncolours) {
  if (!s || !colours) return bad("null");
  return s->s.agent_colours(colours, ncolours);
}
int dcora_rbcd_evaluate(dcora_rbcd_t s, double *cost2, double *gradnorm, double *block_norms, int *next_selected) {
  if (!s) return bad("null");
  DCORA_TRY
  return s->s.evaluate_central(cost2, gradnorm, block_norms, next_selected);
  DCORA_CATCH
}
int dcora_rbcd_agent_iterate(dcora_rbcd_t s, int agent, int do_optimization) {
  if (!s) return bad("null");
  DCORA_TRY
  return s->s.agent_iterate(agent, do_optimization != 0);
  DCORA_CATCH
}
int dcora_rbcd_agent_update_neighbor(dcora_rbcd_t s, int agent, int neighbor, int count, const int *frames,
                                     const double *poses, int auxiliary) {
  if (!s || (count > 0 && (!frames || !poses))) return bad("null");
  DCORA_TRY
  return s->s.agent_update_neighbor(agent, neighbor, count, frames, poses, auxiliary != 0);
  DCORA_CATCH
}
int dcora_rbcd_agent_last_skipped(dcora_rbcd_t s, int agent, int *skipped) {
  if (!s || !skipped || agent < 0 || agent >= s->s.R) return bad("bad agent");
  *skipped = s->s.agents[agent].last_skipped ? 1 : 0;
  return DCORA_OK;
}
int dcora_rbcd_agent_get_X(dcora_rbcd_t s, int agent, double *X) {
  return (s && X) ? s->s.agent_get_X(agent, X) : bad("null");
}
int dcora_rbcd_agent_set_X(dcora_rbcd_t s, int agent, const double *X) {
  return (s && X) ? s->s.agent_set_X(agent, X) : bad("null");
}
int dcora_rbcd_agent_info(dcora_rbcd_t s, int agent, int *num_poses, int *first_pose, int *iteration_number) {
  if (!s || agent < 0 || agent >= s->s.R) return bad("bad agent");
  const AgentDev &a = s->s.agents[agent];
  if (num_poses) *num_poses = a.n;
  if (first_pose) *first_pose = a.col0 / (s->s.d + 1);
  if (iteration_number)
    *iteration_number = (int)s->s.agent_it.size() == s->s.R ? s->s.agent_it[agent] : s->s.iteration;
  return DCORA_OK;
}
int dcora_rbcd_last_result(dcora_rbcd_t s, dcora_ropt_result *res) {
  if (!s || !res) return bad("null");
  return s->s.last_result(res);
}
int dcora_rbcd_X_device_ptr(dcora_rbcd_t s, double **X_dev) {
  if (!s) return bad("null");
  *X_dev = s->s.Xg.p;
  return DCORA_OK;
}
int dcora_rbcd_public_count(dcora_rbcd_t s, int agent, int *count) {
  if (!s || agent < 0 || agent >= s->s.R) return bad("bad agent");
  *count = (int)s->s.agents[agent].public_poses.size();
  return DCORA_OK;
}
int dcora_rbcd_public_indices(dcora_rbcd_t s, int agent, int *idx) {
  if (!s || agent < 0 || agent >= s->s.R) return bad("bad agent");
  const auto &v = s->s.agents[agent].public_poses;
  std::copy(v.begin(), v.end(), idx);
  return DCORA_OK;
}
int dcora_rbcd_pack_public_dev(dcora_rbcd_t s, int agent, double *packed_dev) {
  if (!s || agent < 0 || agent >= s->s.R) return bad("bad agent");
  return s->s.pack_public(agent, packed_dev);
}
int dcora_rbcd_unpack_public_dev(dcora_rbcd_t s, int agent, const double *packed_dev) {
  if (!s || agent < 0 || agent >= s->s.R) return bad("bad agent");
  return s->s.unpack_public(agent, packed_dev);
}
int dcora_rbcd_phase_nonselected(dcora_rbcd_t s, int selected) { return s ? s->s.phase_nonselected(selected) : bad("null"); }
int dcora_rbcd_phase_selected(dcora_rbcd_t s, int selected) {
  if (!s) return bad("null");
  DCORA_TRY
  return s->s.phase_selected(selected);
  DCORA_CATCH
}
int dcora_rbcd_phase_evaluate_dev(dcora_rbcd_t s, double *out_dev) { return s ? s->s.phase_evaluate_dev(out_dev) : bad("null"); }
// measurement hook of bench.py's roofline: HIP events around the one-launch tCG runs of the session's agents
int dcora_rbcd_profile_tcg_runs(dcora_rbcd_t s, int enable) {
  if (!s) return bad("null");
  for (auto &a : s->s.agents)
    if (a.prob) a.prob->profile_tcg_runs = enable != 0;
  return DCORA_OK;
}
int dcora_rbcd_profile_tcg_read(dcora_rbcd_t s, double *out2) {
  if (!s || !out2) return bad("null");
  DCORA_TRY
  out2[0] = out2[1] = 0;
  for (auto &a : s->s.agents) {
    if (!a.prob) continue;
    double n = 0, us = 0;
    const int rc = a.prob->profile_tcg_read(&n, &us);
    if (rc) return rc;
    out2[0] += n;
    out2[1] += us;
  }
  return DCORA_OK;
  DCORA_CATCH
}
int dcora_rbcd_synchronize(dcora_rbcd_t s) {
  if (!s) return bad("null");
  DCORA_HIP(hipStreamSynchronize(s->s.st));
  return DCORA_OK;
}

// ---- neighbour exchange between ranks ------------------------------------------------------------------------------
int dcora_exchange_create(dcora_rbcd_t s, const char *job_name, dcora_exchange_t *out) {
  if (!s || !job_name || !out) return bad("null argument");
  DCORA_TRY
  dcora_exchange_s *h = new dcora_exchange_s;
  const int rc = h->e.init(&s->s, job_name);
  if (rc) {
    delete h;
    return rc;
  }
  *out = h;
  return DCORA_OK;
  DCORA_CATCH
}
int dcora_exchange_destroy(dcora_exchange_t ex) {
  delete ex;
  return DCORA_OK;
}
int dcora_exchange_info(dcora_exchange_t ex, double *info) {
  if (!ex || !info) return bad("null");
  const Exchange &e = ex->e;
  info[0] = e.mode;
  info[1] = e.num_peers();
  info[2] = (double)e.posts;
  info[3] = (double)e.waits;
  info[4] = e.bytes_posted;
  info[5] = e.post_s;
  info[6] = e.wait_s;
  info[7] = e.eval_wait_s;
  info[8] = e.halo_is_finegrained() ? 1 : 0;
  info[9] = e.waits_on_device() ? 1 : 0;
  return DCORA_OK;
}
int dcora_debug_exchange_probe_fault(int rounds) {
  g_probe_fault_rounds.store(rounds);
  return DCORA_OK;
}
int dcora_exchange_link_report(dcora_exchange_t ex, double *out4) {
  if (!ex || !out4) return bad("null");
  const Exchange &e = ex->e;
  out4[0] = e.link_rounds;
  out4[1] = e.link_gave_up_device_wait;
  out4[2] = e.link_gave_up_ipc;
  out4[3] = e.link_last_us;
  return DCORA_OK;
}
int dcora_exchange_post(dcora_exchange_t ex, const int *agents, int count) {
  if (!ex || (!agents && count > 0)) return bad("null");
  DCORA_TRY
  return ex->e.post(agents, count);
  DCORA_CATCH
}
int dcora_exchange_wait(dcora_exchange_t ex, const int *agents, int count) {
  if (!ex || (!agents && count > 0)) return bad("null");
  DCORA_TRY
  return ex->e.wait(agents, count);
  DCORA_CATCH
}
int dcora_exchange_evaluate(dcora_exchange_t ex, double *cost2, double *gradnorm, double *block_norms,
                            int *next_selected) {
  if (!ex) return bad("null");
  DCORA_TRY
  return ex->e.evaluate(cost2, gradnorm, block_norms, next_selected);
  DCORA_CATCH
}
int dcora_exchange_rbcd_iterate(dcora_exchange_t ex, int selected, double *cost2, double *gradnorm,
                                double *block_norms, int *next_selected) {
  if (!ex) return bad("null");
  DCORA_TRY
  return ex->e.rbcd_iterate(selected, cost2, gradnorm, block_norms, next_selected);
  DCORA_CATCH
}
int dcora_exchange_rbcd_tick(dcora_exchange_t ex, const int *set, int count, int allow_adjacent) {
  if (!ex || !set) return bad("null");
  DCORA_TRY
  return ex->e.rbcd_tick(set, count, allow_adjacent);
  DCORA_CATCH
}
int dcora_exchange_set_X(dcora_exchange_t ex, const double *X) {
  if (!ex || !X) return bad("null");
  DCORA_TRY
  return ex->e.set_X(X);
  DCORA_CATCH
}
int dcora_exchange_gather_X(dcora_exchange_t ex, double *X) {
  if (!ex || !X) return bad("null");
  DCORA_TRY
  return ex->e.gather_X(X);
  DCORA_CATCH
}
int dcora_exchange_host_selftest(const char *job_name, int rank, int world_size, int num_agents, int rounds,
                                 double *checksum) {
  if (!job_name) return bad("null");
  DCORA_TRY
  Exchange e;
  return e.host_selftest(job_name, rank, world_size, num_agents, rounds, checksum);
  DCORA_CATCH
}
int dcora_exchange_certify(dcora_exchange_t ex, int k, const int *rowptr, const int *colidx, const double *vals,
                           double eta, int *certified, double *theta, double *lambda_min, double *v, long long *matvecs,
                           int *distributed) {
  if (!ex) return bad("null");
  DCORA_TRY
  if (rowptr && colidx && vals) {
    const HostCsr Q = view_csr(k, rowptr, colidx, vals);
    return ex->e.certify(&Q, eta, certified, theta, lambda_min, v, matvecs, distributed);
  }
  return ex->e.certify(nullptr, eta, certified, theta, lambda_min, v, matvecs, distributed);
  DCORA_CATCH
}
int dcora_exchange_all_ready(dcora_exchange_t ex, int ready, int *all_ready) {
  if (!ex || !all_ready) return bad("null");
  DCORA_TRY
  double notready = ready ? 0.0 : 1.0;
  const int rc = ex->e.allreduce_sum(&notready, 1);
  if (rc) return rc;
  *all_ready = notready == 0.0 ? 1 : 0;
  return DCORA_OK;
  DCORA_CATCH
}
int dcora_debug_exchange_leave_stale(const char *job_name, int world_size, int num_agents) {
  if (!job_name) return bad("null");
  DCORA_TRY
  Exchange e;
  return e.debug_leave_stale(job_name, world_size, num_agents);
  DCORA_CATCH
}
int dcora_exchange_barrier(dcora_exchange_t ex) {
  if (!ex) return bad("null");
  return ex->e.barrier();
}

// ---- RBCD session, range-aided SLAM ----------------------------------------------------------------------------
struct dcora_ra_rbcd_s {
  RaRbcdSession s;
};
int dcora_ra_rbcd_create(dcora_radataset_t ds, const dcora_rbcd_options *opt, dcora_ra_rbcd_t *out) {
  if (!ds || !opt || !out) return bad("null argument");
  DCORA_TRY
  dcora_ra_rbcd_s *h = new dcora_ra_rbcd_s;
  const int rc = h->s.init(ds->ds, *opt);
  if (rc) {
    delete h;
    return rc;
  }
  *out = h;
  return DCORA_OK;
  DCORA_CATCH
}
int dcora_exchange_create_ra(dcora_ra_rbcd_t s, const char *job_name, dcora_exchange_t *out) {
  if (!s || !job_name || !out) return bad("null argument");
  DCORA_TRY
  dcora_exchange_s *h = new dcora_exchange_s;
  const int rc = h->e.init(&s->s, job_name);
  if (rc) {
    delete h;
    return rc;
  }
  *out = h;
  return DCORA_OK;
  DCORA_CATCH
}
int dcora_ra_rbcd_destroy(dcora_ra_rbcd_t s) {
  delete s;
  return DCORA_OK;
}
int dcora_ra_rbcd_info(dcora_ra_rbcd_t s, int *num_agents, int *robots) {
  if (!s || !num_agents) return bad("null");
  *num_agents = s->s.R;
  if (robots)
    for (int i = 0; i < s->s.R; ++i) robots[i] = s->s.agents[i].robot;
  return DCORA_OK;
}
int dcora_ra_rbcd_set_X(dcora_ra_rbcd_t s, const double *X) { return (s && X) ? s->s.set_X(X) : bad("null"); }
int dcora_ra_rbcd_get_X(dcora_ra_rbcd_t s, double *X) { return (s && X) ? s->s.get_X(X) : bad("null"); }
int dcora_ra_rbcd_iterate(dcora_ra_rbcd_t s, int selected, double *cost2, double *gradnorm, double *block_norms,
                          int *next_selected) {
  if (!s) return bad("null");
  DCORA_TRY
  return s->s.iterate(selected, cost2, gradnorm, block_norms, next_selected);
  DCORA_CATCH
}
int dcora_ra_rbcd_evaluate(dcora_ra_rbcd_t s, double *cost2, double *gradnorm, double *block_norms,
                           int *next_selected) {
  if (!s) return bad("null");
  DCORA_TRY
  return s->s.evaluate(cost2, gradnorm, block_norms, next_selected);
  DCORA_CATCH
}
int dcora_ra_rbcd_run(dcora_ra_rbcd_t s, int max_iters, double rgrad_tol, int *iters_done, double *cost2_trace,
                      double *gradnorm_trace, int *selected_trace) {
  if (!s) return bad("null");
  DCORA_TRY
  int selected = 0, it = 0;
  for (; it < max_iters; ++it) {
    double c2 = 0, gn = 0;
    int nxt = selected;
    const int rc = s->s.iterate(selected, &c2, &gn, nullptr, &nxt);
    if (rc) return rc;
    if (cost2_trace) cost2_trace[it] = c2;
    if (gradnorm_trace) gradnorm_trace[it] = gn;
    if (selected_trace) selected_trace[it] = selected;
    if (gn < rgrad_tol) {
      ++it;
      break;
    }
    selected = nxt;
  }
  if (iters_done) *iters_done = it;
  return DCORA_OK;
  DCORA_CATCH
}
int dcora_ra_rbcd_last_result(dcora_ra_rbcd_t s, dcora_ropt_result *res) {
  if (!s || !res) return bad("null");
  return s->s.last_result(res);
}

// ---- robust estimation ---------------------------------------------------------------------------------------
void dcora_robust_params_default(dcora_robust_params *p) {
  p->cost_type = DCORA_ROBUST_L2;
  p->GNCMaxNumIters = 20;
  p->GNCBarc = 5.0;
  p->GNCMuStep = 1.4;
  p->GNCInitMu = 1e-4;
  p->HuberThreshold = 3;
  p->TLSThreshold = 10;
}
int dcora_robust_weights(const dcora_robust_params *p, int num_updates, int n, const double *r, double *w) {
  if (!p || !r || !w) return bad("null argument");
  if (p->cost_type < DCORA_ROBUST_L2 || p->cost_type > DCORA_ROBUST_GNC_TLS) return bad("unknown robust cost type");
  RobustCost c(*p);
  for (int i = 0; i < num_updates; ++i) c.update();
  for (int i = 0; i < n; ++i) w[i] = c.weight(r[i]);
  return DCORA_OK;
}
int dcora_chi2inv(double quantile, int dof, double *out) {
  if (!out || !(quantile > 0) || !(quantile < 1) || dof < 1) return bad("chi2inv: need 0 < quantile < 1, dof >= 1");
  *out = chi2inv(quantile, dof);
  return DCORA_OK;
}
int dcora_robust_error_threshold_at_quantile(double quantile, int dimension, double *out) {
  if (!out) return bad("null argument");
  if (!error_threshold_at_quantile(quantile, dimension, out))
    return bad("quantile function currently only supports 3D problems and quantile > 0");
  return DCORA_OK;
}
int dcora_robust_single_rotation_averaging(int d, int n, const double *R, const double *kappa, double thr,
                                           double *Ropt, int *inlier) {
  if (!R || !Ropt || !inlier || n < 1 || (d != 2 && d != 3)) return bad("bad argument");
  DCORA_TRY
  std::vector<int> in;
  robust_single_rotation_averaging(d, n, R, kappa, thr, Ropt, in);
  std::fill(inlier, inlier + n, 0);
  for (int i : in) inlier[i] = 1;
  return DCORA_OK;
  DCORA_CATCH
}
int dcora_robust_single_pose_averaging(int d, int n, const double *R, const double *t, const double *kappa,
                                       const double *tau, double thr, double *Ropt, double *topt, int *inlier) {
  if (!R || !t || !Ropt || !topt || !inlier || n < 1 || (d != 2 && d != 3)) return bad("bad argument");
  DCORA_TRY
  std::vector<int> in;
  robust_single_pose_averaging(d, n, R, t, kappa, tau, thr, Ropt, topt, in);
  std::fill(inlier, inlier + n, 0);
  for (int i : in) inlier[i] = 1;
  return DCORA_OK;
  DCORA_CATCH
}
int dcora_agent_neighbor_transforms(int d, int m, const int *incoming, const double *meas_R, const double *meas_t,
                                    const double *nbr_pose, const double *my_pose, double *T_out) {
  if (!incoming || !meas_R || !meas_t || !nbr_pose || !my_pose || !T_out || m < 0 || (d != 2 && d != 3))
    return bad("bad argument");
  const int ps = d * (d + 1);
  for (int i = 0; i < m; ++i)
    neighbor_transform(d, incoming[i] != 0, meas_R + (size_t)i * d * d, meas_t + (size_t)i * d,
                       nbr_pose + (size_t)i * ps, my_pose + (size_t)i * ps, T_out + (size_t)i * ps);
  return DCORA_OK;
}
int dcora_agent_robust_neighbor_transform(int d, int m, const double *candidates, int two_stage, int min_inliers,
                                          double *T_world_robot, int *num_inliers, int *ok) {
  if (!candidates || !T_world_robot || !ok || m < 0 || (d != 2 && d != 3)) return bad("bad argument");
  DCORA_TRY
  *ok = robust_neighbor_transform(d, m, candidates, two_stage != 0, min_inliers, T_world_robot, num_inliers) ? 1 : 0;
  return DCORA_OK;
  DCORA_CATCH
}
int dcora_log_trajectory(const char *path, int d, int n, const double *T) {
  if (!path || !T || n < 0 || (d != 2 && d != 3)) return bad("bad argument");
  FILE *f = std::fopen(path, "w");
  if (!f) {
    set_last_error(std::string("cannot log trajectory to ") + path);
    return DCORA_ERR_IO;
  }
  std::fprintf(f, "# pose_index x y z qx qy qz qw\n");
  const int dh = d + 1;
  for (int i = 0; i < n; ++i) {
    double R[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}}, t[3] = {0, 0, 0};
    const double *Ti = T + (size_t)i * dh * d;
    for (int c = 0; c < d; ++c)
      for (int a = 0; a < d; ++a) R[a][c] = Ti[a + c * d];
    for (int a = 0; a < d; ++a) t[a] = Ti[a + d * d];
    // rotation matrix -> quaternion with the branch rule of Eigen::Quaterniond(Matrix3d) the reference relies on
    double q[4];  // x y z w
    double tr = R[0][0] + R[1][1] + R[2][2];
    if (tr > 0) {
      double s = std::sqrt(tr + 1.0);
      q[3] = 0.5 * s;
      s = 0.5 / s;
      q[0] = (R[2][1] - R[1][2]) * s;
      q[1] = (R[0][2] - R[2][0]) * s;
      q[2] = (R[1][0] - R[0][1]) * s;
    } else {
      int a = 0;
      if (R[1][1] > R[0][0]) a = 1;
      if (R[2][2] > R[a][a]) a = 2;
      const int b = (a + 1) % 3, c = (b + 1) % 3;
      double s = std::sqrt(R[a][a] - R[b][b] - R[c][c] + 1.0);
      q[a] = 0.5 * s;
      s = 0.5 / s;
      q[3] = (R[c][b] - R[b][c]) * s;
      q[b] = (R[b][a] + R[a][b]) * s;
      q[c] = (R[c][a] + R[a][c]) * s;
    }
    std::fprintf(f, "%d %.9f %.9f %.9f %.9f %.9f %.9f %.9f\n", i, t[0], t[1], t[2], q[0], q[1], q[2], q[3]);
  }
  std::fclose(f);
  return DCORA_OK;
}
int dcora_fixed_stiefel_variable(int r, int d, double *Y) {
  if (!Y || d < 1 || r < d) return bad("fixed_stiefel_variable: need r >= d >= 1");
  fixed_stiefel_variable(r, d, Y);
  return DCORA_OK;
}
int dcora_agent_initialize_in_global_frame(const dcora_dims *dims, const double *T_world_robot,
                                           const double *T_local, const double *YLift, double *X) {
  if (!dims || !T_world_robot || !T_local || !YLift || !X) return bad("null argument");
  if ((dims->d != 2 && dims->d != 3) || dims->r < dims->d || dims->n < 1) return bad("bad dims");
  if (!layout_ok(dims)) return bad("dims: layout SE needs l = b = 0");
  initialize_in_global_frame(dims->r, dims->d, dims->n, dims->l, dims->b, make_mani(*dims).se != 0, T_world_robot, T_local,
                             YLift, X);
  return DCORA_OK;
}
int dcora_measurement_errors(dcora_dataset_t ds, int r, const double *X, double *out, int device) {
  if (!ds || !X || !out) return bad("null argument");
  if (r < ds->ds.d || r > 16) return bad("measurement_errors: need d <= r <= 16");
  DCORA_TRY
  return measurement_errors(ds->ds, r, X, out, device);
  DCORA_CATCH
}
int dcora_solve_pgo(dcora_dataset_t ds, const dcora_ropt_params *params, const double *T0, double *Tout,
                    dcora_ropt_result *result, int device) {
  if (!ds || !params || !Tout) return bad("null argument");
  DCORA_TRY
  return solve_pgo(ds->ds, *params, T0, Tout, device, result);
  DCORA_CATCH
}
int dcora_solve_robust_pgo(dcora_dataset_t ds, const dcora_ropt_params *params, const dcora_robust_params *robust,
                           const int *fixed_weight, const double *T0, double *Tout, double *weights_out, int device) {
  if (!ds || !params || !robust || !Tout) return bad("null argument");
  DCORA_TRY
  return solve_robust_pgo(ds->ds, *params, *robust, fixed_weight, T0, Tout, weights_out, device);
  DCORA_CATCH
}

// ---- rounding ----------------------------------------------------------------------------------------------
int dcora_round_align_trajectory(const dcora_dims *dims, const double *X, const double *anchor, int global_alignment,
                                 double *trajectory, double *unit_spheres, double *landmarks, int device) {
  if (!dims || !X || !trajectory) return bad("null argument");
  if (!layout_ok(dims)) return bad("dims: layout SE needs l = b = 0");
  DCORA_TRY
  return round_align(*dims, X, anchor, global_alignment, trajectory, unit_spheres, landmarks, device);
  DCORA_CATCH
}
int dcora_round_project_solution_raslam(const dcora_dims *dims, const double *X, double *out, int device) {
  if (!dims || !X || !out) return bad("null argument");
  if (!layout_ok(dims)) return bad("dims: layout SE needs l = b = 0");
  DCORA_TRY
  return round_project_solution(*dims, X, out, device);
  DCORA_CATCH
}

}  // extern "C"

// debug / test hook (not part of the public header): runs one Nesterov bookkeeping kernel on host arrays.
// flavour 0 = thread-per-pose kernel, 1 = 8-lanes-per-pose kernel.  Arrays are r x (d+1) n, updated in place.
extern "C" int dcora_debug_nesterov(int flavour, int r, int d, int n, int mode, int restart, int skip_lo, int skip_hi,
                                    double alpha, double gamma, double *X, double *V, double *Y, double *XPrev,
                                    double *Yloc, const double *Xloc) {
  const ManiDesc m = make_mani(r, d, n, 0, 0);
  const size_t N = (size_t)r * m.k, B = N * sizeof(double);
  DevBuf<double> dX, dV, dY, dP, dYl, dXl;
  for (DevBuf<double> *b : {&dX, &dV, &dY, &dP, &dYl, &dXl}) DCORA_HIP(b->alloc(N));
  DCORA_HIP(hipMemcpy(dX.p, X, B, hipMemcpyHostToDevice));
  DCORA_HIP(hipMemcpy(dV.p, V, B, hipMemcpyHostToDevice));
  DCORA_HIP(hipMemcpy(dY.p, Y, B, hipMemcpyHostToDevice));
  DCORA_HIP(hipMemcpy(dP.p, XPrev, B, hipMemcpyHostToDevice));
  DCORA_HIP(hipMemcpy(dYl.p, Yloc, B, hipMemcpyHostToDevice));
  DCORA_HIP(hipMemcpy(dXl.p, Xloc, B, hipMemcpyHostToDevice));
  if (flavour)
    launch_g_nesterov(nullptr, m, mode, restart, skip_lo, skip_hi, alpha, gamma, dX.p, dV.p, dY.p, dP.p, dYl.p,
                      buf1(dXl.p), nullptr);
  else
    launch_nesterov(nullptr, m, mode, restart, skip_lo, skip_hi, alpha, gamma, dX.p, dV.p, dY.p, dP.p, dYl.p, dXl.p);
  DCORA_HIP(hipDeviceSynchronize());
  DCORA_HIP(hipMemcpy(X, dX.p, B, hipMemcpyDeviceToHost));
  DCORA_HIP(hipMemcpy(V, dV.p, B, hipMemcpyDeviceToHost));
  DCORA_HIP(hipMemcpy(Y, dY.p, B, hipMemcpyDeviceToHost));
  DCORA_HIP(hipMemcpy(XPrev, dP.p, B, hipMemcpyDeviceToHost));
  DCORA_HIP(hipMemcpy(Yloc, dYl.p, B, hipMemcpyDeviceToHost));
  return DCORA_OK;
}

// debug / test hook (not part of the public header): builds the partitioned inverse of an SPD matrix on the host
// and replays its schedule on the host against a plain sparse Cholesky solve -- checks the builder without a GPU.
// info = {levels (forward + backward), pieces, nnz(L), stored weights per apply}
extern "C" int dcora_debug_partinv_selftest(int n, const int *rp, const int *ci, const double *v, int block, int r,
                                            double *max_rel_err, double *info) {
  HostCsr A;
  A.n = A.ncols = n;
  A.rp.assign(rp, rp + n + 1);
  A.ci.assign(ci, ci + rp[n]);
  A.v.assign(v, v + rp[n]);
  PartInvHost P;
  if (!build_partitioned_inverse(A, block, 4, &P)) {
    set_last_error("matrix is not positive definite");
    return DCORA_ERR_NOT_PD;
  }
  SparseChol chol;
  if (!chol.factor(A, block)) return DCORA_ERR_NOT_PD;
  std::vector<double> R((size_t)n * r), Z((size_t)n * r), col((size_t)n), sol((size_t)n);
  unsigned long long s = 88172645463325252ull;
  for (double &x : R) {
    s ^= s << 13;
    s ^= s >> 7;
    s ^= s << 17;
    x = (double)(s >> 11) / 9007199254740992.0 - 0.5;
  }
  partitioned_inverse_apply_host(P, r, R.data(), Z.data());
  double err = 0, ref = 0;
  for (int t = 0; t < r; ++t) {
    for (int i = 0; i < n; ++i) col[i] = R[(size_t)i * r + t];
    chol.solve_vec(col.data(), sol.data());
    for (int i = 0; i < n; ++i) {
      err = std::max(err, std::fabs(sol[i] - Z[(size_t)i * r + t]));
      ref = std::max(ref, std::fabs(sol[i]));
    }
  }
  *max_rel_err = err / std::max(ref, 1e-300);
  info[0] = (double)P.levels.size();
  info[1] = (double)P.npieces;
  info[2] = (double)P.nnzL;
  info[3] = P.weights_read_per_apply;
  return DCORA_OK;
}

// debug / test hook (not part of the public header): the stored weights of the same matrix once written to host memory
// in one piece and once STREAMED through a WeightSink in chunks of at most `cap` doubles (what the product does towards
// the device, DeviceWeightSink) -- runs without a GPU.  out = {weights, chunks, weights that differ, largest chunk}
extern "C" int dcora_debug_partinv_stream_check(int n, const int *rp, const int *ci, const double *v, int block,
                                                long long cap, double *out) {
  HostCsr A;
  A.n = A.ncols = n;
  A.rp.assign(rp, rp + n + 1);
  A.ci.assign(ci, ci + rp[n]);
  A.v.assign(v, v + rp[n]);
  PartInvHost P, Q;
  if (!build_partitioned_inverse(A, block, 4, &P)) return DCORA_ERR_NOT_PD;
  struct HostSink : WeightSink {
    std::vector<double> all, buf;
    long long cap = 0, chunks = 0, largest = 0, expect = 0;
    bool ordered = true, ended = false;
    bool begin(long long total) override {
      all.assign((size_t)total, -12345.0);  // a weight the sink never receives keeps this value
      buf.resize((size_t)cap);
      return true;
    }
    long long chunk_cap() const override { return cap; }
    double *acquire(long long m) override {
      if (m > cap) return nullptr;
      std::fill(buf.begin(), buf.end(), 777.0);  // the builder must write (or zero) everything it commits
      return buf.data();
    }
    bool commit(long long off, long long m) override {
      ordered = ordered && off == expect;
      expect = off + m;
      ++chunks;
      largest = std::max(largest, m);
      std::copy(buf.begin(), buf.begin() + m, all.begin() + off);
      return true;
    }
    bool end() override {
      ended = true;
      return true;
    }
  } sink;
  sink.cap = cap;
  Q.sink = &sink;
  if (!build_partitioned_inverse(A, block, 3, &Q)) {
    set_last_error("the streamed build failed (a fill larger than the chunk?)");
    return DCORA_ERR_BAD_ARG;
  }
  long long differ = 0;
  if ((long long)P.vals.size() != Q.nvals || !Q.vals.empty() || !sink.ordered || !sink.ended ||
      sink.expect != Q.nvals)
    differ = -1;
  else
    for (size_t i = 0; i < P.vals.size(); ++i) differ += std::memcmp(&P.vals[i], &sink.all[i], sizeof(double)) != 0;
  out[0] = (double)Q.nvals;
  out[1] = (double)sink.chunks;
  out[2] = (double)differ;
  out[3] = (double)sink.largest;
  return DCORA_OK;
}

// measurement hook (bench.py, SURVEY 8(d): "verify with a stream-triad on the box"): a[i] = b[i] + s c[i] over three
// arrays of n doubles, `reps` launches back to back between two HIP events; GB/s counts 24 n bytes per launch
namespace {
__global__ __launch_bounds__(256) void k_stream_triad(size_t n2, const double2 *__restrict__ b,
                                                      const double2 *__restrict__ c, double2 *__restrict__ a,
                                                      double s) {
  // four 16-byte loads per array in flight per lane
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + 3 * stride < n2; i += 4 * stride) {
    double2 x[4], y[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      x[u] = b[i + u * stride];
      y[u] = c[i + u * stride];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      a[i + u * stride] = make_double2(x[u].x + s * y[u].x, x[u].y + s * y[u].y);
  }
  for (; i < n2; i += stride) {
    const double2 x = b[i], y = c[i];
    a[i] = make_double2(x.x + s * y.x, x.y + s * y.y);
  }
}
}  // namespace
extern "C" int dcora_debug_stream_triad(int device, size_t n, int reps, double *gbps) {
  DCORA_TRY
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    set_last_error("no HIP device available: libdcora_hip has no CPU fallback");
    return DCORA_ERR_NO_DEVICE;
  }
  DCORA_HIP(hipSetDevice(device));
  n &= ~(size_t)1;
  DevBuf<double> A, B, Cc;
  DCORA_HIP(A.alloc(n));
  DCORA_HIP(B.alloc(n));
  DCORA_HIP(Cc.alloc(n));
  DCORA_HIP(hipMemset(B.p, 0, n * sizeof(double)));
  DCORA_HIP(hipMemset(Cc.p, 0, n * sizeof(double)));
  hipEvent_t e0, e1;
  DCORA_HIP(hipEventCreate(&e0));
  DCORA_HIP(hipEventCreate(&e1));
  const int grid = 1024;  // 1024 / 2048 / 8192 workgroups: 5.01 / 4.85 / 4.48 TB/s
  for (int w = 0; w < 3; ++w)
    hipLaunchKernelGGL(k_stream_triad, dim3(grid), dim3(256), 0, nullptr, n / 2, (const double2 *)B.p,
                       (const double2 *)Cc.p, (double2 *)A.p, 0.5);
  DCORA_HIP(hipEventRecord(e0, nullptr));
  for (int w = 0; w < reps; ++w)
    hipLaunchKernelGGL(k_stream_triad, dim3(grid), dim3(256), 0, nullptr, n / 2, (const double2 *)B.p,
                       (const double2 *)Cc.p, (double2 *)A.p, 0.5);
  DCORA_HIP(hipEventRecord(e1, nullptr));
  DCORA_HIP(hipEventSynchronize(e1));
  float ms = 0;
  DCORA_HIP(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  *gbps = 24.0 * (double)n * reps / (ms * 1e-3) / 1e9;
  return DCORA_OK;
  DCORA_CATCH
}

// measurement hook: cost of a grid-wide barrier among `blocks` co-resident workgroups (monotonic arrival counter in
// device memory, agent-scope release / acquire, one polling lane per workgroup with s_sleep) -- the price a persistent
// one-launch tCG iteration would pay per dependency.  Every wait is bounded: a workgroup that does
// not see its peers within `spin_limit` polls gives up and raises *timeouts, the kernel always drains.
namespace {
__global__ __launch_bounds__(256) void k_grid_barrier_probe(unsigned *counter, int iters, int spin_limit,
                                                            int *timeouts, double *sink, int mode) {
  // mode 0: one arrival counter polled by every workgroup.  mode 1: two levels -- groups of 32 workgroups (words 64
  // apart) arrive on their group's counter, the last of a group arrives on the top counter, the last of all bumps one
  // release word per group; a workgroup polls only its group's release word.
  extern __shared__ double s_dyn[];
  const unsigned nb = gridDim.x;
  constexpr unsigned GS = 32;
  const unsigned ng = (nb + GS - 1) / GS, g = blockIdx.x / GS;
  const unsigned gsize = min(GS, nb - g * GS);
  unsigned *grp = counter + 64 * (1 + g), *rel = counter + 64 * (1 + ng + g);
  unsigned target = 0, epoch = 0;
  double acc = 0;
  bool dead = false;
  for (int it = 0; it < iters && !dead; ++it) {
    acc += s_dyn[threadIdx.x & 7];  // keep the LDS allocation alive
    __syncthreads();
    if (threadIdx.x == 0) {
      int spins = 0;
      if (mode == 0) {
        target += nb;
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
          if (++spins > spin_limit) break;
          __builtin_amdgcn_s_sleep(1);
        }
      } else {
        ++epoch;
        const unsigned a = __hip_atomic_fetch_add(grp, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (a + 1 == epoch * gsize) {  // last of the group
          const unsigned t = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
          if (t + 1 == epoch * ng)     // last of all: release every group
            for (unsigned q = 0; q < ng; ++q)
              __hip_atomic_store(counter + 64 * (1 + ng + q), epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
        while (__hip_atomic_load(rel, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < epoch) {
          if (++spins > spin_limit) break;
          __builtin_amdgcn_s_sleep(1);
        }
      }
      if (spins > spin_limit) atomicAdd(timeouts, 1);
      s_dyn[8] = (spins > spin_limit) ? 1.0 : 0.0;
    }
    __syncthreads();
    dead = s_dyn[8] != 0.0;
  }
  if (acc == 12345.678) sink[0] = acc;
}
}  // namespace
extern "C" int dcora_debug_grid_barrier(int device, int blocks, int iters, int lds_bytes, int mode,
                                        double *us_per_barrier, int *timeouts_out) {
  DCORA_TRY
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    set_last_error("no HIP device available: libdcora_hip has no CPU fallback");
    return DCORA_ERR_NO_DEVICE;
  }
  DCORA_HIP(hipSetDevice(device));
  hipDeviceProp_t prop;
  DCORA_HIP(hipGetDeviceProperties(&prop, device));
  if (blocks < 1 || blocks > prop.multiProcessorCount || lds_bytes < 128 || lds_bytes > 160 * 1024) {
    set_last_error("grid barrier probe: one workgroup per compute unit at most, 128 B .. 160 KB of LDS");
    return DCORA_ERR_BAD_ARG;
  }
  DevBuf<unsigned> counter;
  DevBuf<int> touts;
  DevBuf<double> sink;
  const size_t nwords = 64 * (1 + 2 * ((size_t)blocks / 32 + 1));
  DCORA_HIP(counter.alloc(nwords));
  DCORA_HIP(touts.alloc(1));
  DCORA_HIP(sink.alloc(1));
  DCORA_HIP(hipFuncSetAttribute((const void *)k_grid_barrier_probe, hipFuncAttributeMaxDynamicSharedMemorySize,
                                lds_bytes));
  hipEvent_t e0, e1;
  DCORA_HIP(hipEventCreate(&e0));
  DCORA_HIP(hipEventCreate(&e1));
  float ms[2] = {0, 0};
  const int its[2] = {1, iters};
  for (int pass = 0; pass < 2; ++pass) {
    DCORA_HIP(hipMemset(counter.p, 0, nwords * sizeof(unsigned)));
    DCORA_HIP(hipMemset(touts.p, 0, sizeof(int)));
    DCORA_HIP(hipEventRecord(e0, nullptr));
    hipLaunchKernelGGL(k_grid_barrier_probe, dim3(blocks), dim3(256), lds_bytes, nullptr, counter.p, its[pass], 200000,
                       touts.p, sink.p, mode);
    DCORA_HIP(hipEventRecord(e1, nullptr));
    DCORA_HIP(hipEventSynchronize(e1));
    DCORA_HIP(hipEventElapsedTime(&ms[pass], e0, e1));
  }
  int t = 0;
  DCORA_HIP(hipMemcpy(&t, touts.p, sizeof(int), hipMemcpyDeviceToHost));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  *timeouts_out = t;
  *us_per_barrier = 1e3 * (ms[1] - ms[0]) / std::max(1, iters - 1);
  return DCORA_OK;
  DCORA_CATCH
}

