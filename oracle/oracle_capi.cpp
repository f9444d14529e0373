// ORACLE -- test infrastructure only (see oracle.hpp).
// Flat C entry points so tests / bench.py (cpu_baseline leg) can drive the CPU
// restatement through ctypes.  Matrices are column-major double, CSR is
// row-major int32 -- the same layouts as include/dcora_hip.h.
#include <cstdio>
#include <cstdlib>

#include "oracle.hpp"

using namespace orc;

namespace {
CSR view_csr(int n, const int *rp, const int *ci, const double *v) {
  CSR A;
  A.n = n;
  A.rp.assign(rp, rp + n + 1);
  A.ci.assign(ci, ci + rp[n]);
  A.v.assign(v, v + rp[n]);
  return A;
}
Mat view_mat(int r, int c, const double *p) {
  Mat M(r, c);
  if (p) std::copy(p, p + (size_t)r * c, M.a.begin());
  return M;
}
std::vector<Meas> view_meas(int d, int m, const int *ids, const double *vals) {
  const int stride = d * d + d + 3;
  std::vector<Meas> out(m);
  for (int k = 0; k < m; ++k) {
    Meas &e = out[k];
    e.r1 = ids[4 * k + 0];
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
struct ProblemBox {
  Dims D;
  CSR Q;
  Mat G;
  bool hasG = false;
  Chol precon;
  bool hasP = false;
  Problem P;
  void bind() {
    P.D = D;
    P.Q = &Q;
    P.G = hasG ? &G : nullptr;
    P.precon = hasP ? &precon : nullptr;
  }
};
}  // namespace

extern "C" {

// ---- datasets ----------------------------------------------------------------
void *orc_g2o_load(const char *path) {
  try {
    return new Dataset(read_g2o(path));
  } catch (const std::exception &e) {
    std::fprintf(stderr, "orc_g2o_load: %s\n", e.what());
    return nullptr;
  }
}
void *orc_ds_create(int d, int n, int m, const int *ids, const double *vals) {
  Dataset *ds = new Dataset;
  ds->d = d;
  ds->n = n;
  ds->meas = view_meas(d, m, ids, vals);
  return ds;
}
void orc_ds_info(void *h, int *d, int *n, int *m) {
  Dataset *ds = (Dataset *)h;
  *d = ds->d;
  *n = ds->n;
  *m = (int)ds->meas.size();
}
void orc_ds_copy(void *h, int *ids, double *vals) {
  Dataset *ds = (Dataset *)h;
  const int d = ds->d, stride = d * d + d + 3;
  for (size_t k = 0; k < ds->meas.size(); ++k) {
    const Meas &e = ds->meas[k];
    ids[4 * k + 0] = e.r1;
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
}
void orc_ds_free(void *h) { delete (Dataset *)h; }

// ---- CSR handles ---------------------------------------------------------------
void orc_csr_info(void *h, int *n, int *nnz) {
  CSR *A = (CSR *)h;
  *n = A->n;
  *nnz = A->nnz();
}
void orc_csr_copy(void *h, int *rp, int *ci, double *v) {
  CSR *A = (CSR *)h;
  std::copy(A->rp.begin(), A->rp.end(), rp);
  std::copy(A->ci.begin(), A->ci.end(), ci);
  std::copy(A->v.begin(), A->v.end(), v);
}
void orc_csr_free(void *h) { delete (CSR *)h; }

void *orc_build_Q_pgo(int d, int n, int id, int m, const int *ids, const double *vals) {
  return new CSR(build_Q_pgo(d, n, id, view_meas(d, m, ids, vals)));
}
int orc_build_G_pgo(int r, int d, int n, int id, int m, const int *ids, const double *vals, int nnbr,
                    const int *keys, const double *poses, double *Gout) {
  PoseDict dict;
  const int dh = d + 1;
  for (int i = 0; i < nnbr; ++i)
    dict[{keys[2 * i], keys[2 * i + 1]}] =
        std::vector<double>(poses + (size_t)i * r * dh, poses + (size_t)(i + 1) * r * dh);
  Mat G;
  if (!build_G_pgo(r, d, n, id, view_meas(d, m, ids, vals), dict, G)) return 0;
  std::copy(G.a.begin(), G.a.end(), Gout);
  return 1;
}

// ---- problem -------------------------------------------------------------------
void *orc_problem_create(int r, int d, int n, int l, int b, int kdim, const int *rp, const int *ci,
                         const double *v, const double *G, double reg) {
  ProblemBox *B = new ProblemBox;
  B->D = Dims{r, d, n, l, b};
  B->Q = view_csr(kdim, rp, ci, v);
  if (G) {
    B->G = view_mat(r, kdim, G);
    B->hasG = true;
  }
  if (reg >= 0) {
    CSR M = csr_add_diag(B->Q, reg);
    B->hasP = B->precon.factor(M, (l == 0 && b == 0) ? d + 1 : 1);
  }
  B->bind();
  return B;
}
void orc_problem_free(void *h) { delete (ProblemBox *)h; }
long orc_problem_nnzL(void *h) { return ((ProblemBox *)h)->precon.nnzL(); }
double orc_f(void *h, const double *X) {
  ProblemBox *B = (ProblemBox *)h;
  return B->P.f(view_mat(B->D.r, B->D.k(), X));
}
void orc_egrad(void *h, const double *X, double *out) {
  ProblemBox *B = (ProblemBox *)h;
  Mat EG;
  B->P.egrad(view_mat(B->D.r, B->D.k(), X), EG);
  std::copy(EG.a.begin(), EG.a.end(), out);
}
double orc_rgrad(void *h, const double *X, double *out) {
  ProblemBox *B = (ProblemBox *)h;
  Mat RG;
  B->P.rgrad(view_mat(B->D.r, B->D.k(), X), RG);
  if (out) std::copy(RG.a.begin(), RG.a.end(), out);
  return norm(RG);
}
void orc_hess(void *h, const double *X, const double *V, double *out) {
  ProblemBox *B = (ProblemBox *)h;
  Mat Xm = view_mat(B->D.r, B->D.k(), X), EG, HV;
  B->P.egrad(Xm, EG);
  B->P.hess(Xm, EG, view_mat(B->D.r, B->D.k(), V), HV);
  std::copy(HV.a.begin(), HV.a.end(), out);
}
void orc_precondition(void *h, const double *X, const double *V, double *out) {
  ProblemBox *B = (ProblemBox *)h;
  Mat Z;
  B->P.precondition(view_mat(B->D.r, B->D.k(), X), view_mat(B->D.r, B->D.k(), V), Z);
  std::copy(Z.a.begin(), Z.a.end(), out);
}
void orc_precon_solve(void *h, const double *V, double *out) {
  ProblemBox *B = (ProblemBox *)h;
  Mat Z;
  B->precon.solve_rows(view_mat(B->D.r, B->D.k(), V), Z);
  std::copy(Z.a.begin(), Z.a.end(), out);
}
void orc_tangent_project(int r, int d, int n, int l, int b, const double *X, const double *V, double *out) {
  Dims D{r, d, n, l, b};
  Mat O;
  tangent_project(D, view_mat(r, D.k(), X), view_mat(r, D.k(), V), O);
  std::copy(O.a.begin(), O.a.end(), out);
}
void orc_retract(int r, int d, int n, int l, int b, const double *X, const double *V, double *out) {
  Dims D{r, d, n, l, b};
  Mat O;
  retract(D, view_mat(r, D.k(), X), view_mat(r, D.k(), V), O);
  std::copy(O.a.begin(), O.a.end(), out);
}
void orc_project_to_manifold(int r, int d, int n, int l, int b, const double *M, double *out) {
  Dims D{r, d, n, l, b};
  Mat O;
  project_to_manifold(D, view_mat(r, D.k(), M), O);
  std::copy(O.a.begin(), O.a.end(), out);
}
void orc_project_to_rotation_group(int d, const double *M, double *out) { project_to_rotation_group(d, M, out); }

// params: [method, gradnorm_tol, RGD_stepsize, RGD_use_precond, RTR_iterations, RTR_tCG_iterations, RTR_initial_radius]
// result: [success,fInit,gradNormInit,fOpt,gradNormOpt,elapsedMs,tcg_status,outer,inner,accepted]
void orc_optimize(void *h, const double *params, const double *X0, double *Xout, double *result) {
  ProblemBox *B = (ProblemBox *)h;
  ROptParams p;
  p.method = (int)params[0];
  p.gradnorm_tol = params[1];
  p.RGD_stepsize = params[2];
  p.RGD_use_preconditioner = (int)params[3];
  p.RTR_iterations = (int)params[4];
  p.RTR_tCG_iterations = (int)params[5];
  p.RTR_initial_radius = params[6];
  ROptResult res;
  Mat Y = optimize(B->P, p, view_mat(B->D.r, B->D.k(), X0), &res);
  std::copy(Y.a.begin(), Y.a.end(), Xout);
  if (result) {
    result[0] = res.success;
    result[1] = res.fInit;
    result[2] = res.gradNormInit;
    result[3] = res.fOpt;
    result[4] = res.gradNormOpt;
    result[5] = res.elapsedMs;
    result[6] = res.tcg_status;
    result[7] = res.outer_iters;
    result[8] = res.inner_iters;
    result[9] = res.accepted;
  }
}

// ---- certification -------------------------------------------------------------
void *orc_dual_certificate(int r, int d, int n, int l, int b, const double *X, int kdim, const int *rp,
                           const int *ci, const double *v) {
  Dims D{r, d, n, l, b};
  return new CSR(dual_certificate(D, view_mat(r, kdim, X), view_csr(kdim, rp, ci, v)));
}
int orc_is_psd(int kdim, const int *rp, const int *ci, const double *v, int block) {
  return is_psd(view_csr(kdim, rp, ci, v), block) ? 1 : 0;
}
int orc_min_eig(int kdim, const int *rp, const int *ci, const double *v, int maxit, double tol, int ncv,
                unsigned long long seed, double *lambda, double *vec, long *matvecs) {
  EigResult e = min_eig_pair(view_csr(kdim, rp, ci, v), maxit, tol, ncv, seed);
  *lambda = e.lambda;
  if (vec) std::copy(e.v.begin(), e.v.end(), vec);
  if (matvecs) *matvecs = e.matvecs;
  return e.ok ? 1 : 0;
}
int orc_lanczos_lm(int kdim, const int *rp, const int *ci, const double *v, double shift, int ncv, int maxit,
                   double tol, unsigned long long seed, double *lambda, double *vec, long *matvecs) {
  EigResult e = lanczos_largest_magnitude(view_csr(kdim, rp, ci, v), shift, ncv, maxit, tol, nullptr, seed);
  *lambda = e.lambda;
  if (vec) std::copy(e.v.begin(), e.v.end(), vec);
  if (matvecs) *matvecs = e.matvecs;
  return e.ok ? 1 : 0;
}
int orc_fast_verification(int kdim, const int *rp, const int *ci, const double *v, double eta, int block,
                          double *theta, double *vec, double *lambda_min) {
  std::vector<double> x;
  double th = 0, lm = 0;
  long mv = 0;
  const bool ok = fast_verification(view_csr(kdim, rp, ci, v), eta, block, &th, &x, &lm, &mv);
  if (!ok) {
    *theta = th;
    *lambda_min = lm;
    if (vec) std::copy(x.begin(), x.end(), vec);
  }
  return ok ? 1 : 0;
}
int orc_escape_saddle(void *hnext, const double *Xopt, double theta, const double *v, double gtol, double pgtol,
                      int second_order, double *Xout) {
  ProblemBox *B = (ProblemBox *)hnext;
  const int k = B->D.k();
  Mat Xo;
  const bool ok = escape_saddle(B->P, view_mat(B->D.r - 1, k, Xopt), theta, std::vector<double>(v, v + k), gtol,
                                pgtol, Xo, second_order != 0);
  if (ok) std::copy(Xo.a.begin(), Xo.a.end(), Xout);
  return ok ? 1 : 0;
}

int orc_chordal_init(void *dsh, double *Tout) {
  Dataset *ds = (Dataset *)dsh;
  Mat T = chordal_initialization(*ds);
  if (T.a.empty()) return 0;
  std::copy(T.a.begin(), T.a.end(), Tout);
  return 1;
}

// ---- range-aided SLAM (centralised) -------------------------------------------------------------------------------
void *orc_pyfg_load(const char *path) {
  try {
    return new RADataset(read_pyfg(path));
  } catch (const std::exception &e) {
    std::fprintf(stderr, "orc_pyfg_load: %s\n", e.what());
    return nullptr;
  }
}
// info: [d, n, l, b, m_pose_pose, m_pose_landmark, m_range]
void orc_ra_info(void *h, int *info) {
  RADataset *ds = (RADataset *)h;
  info[0] = ds->d;
  info[1] = ds->n;
  info[2] = ds->l;
  info[3] = ds->b;
  info[4] = (int)ds->pose_pose.size();
  info[5] = (int)ds->pose_landmark.size();
  info[6] = (int)ds->ranges.size();
}
// pp_ids m x 2, pp_vals m x (d*d+d+3); pl_ids m x 2, pl_vals m x (d+2) (t, tau, weight);
// r_ids m x 5 (type1, i, type2, j, l), r_vals m x 3 (range, precision, weight); gt d x k column-major
void orc_ra_copy(void *h, int *pp_ids, double *pp_vals, int *pl_ids, double *pl_vals, int *r_ids, double *r_vals,
                 double *gt) {
  RADataset *ds = (RADataset *)h;
  const int d = ds->d, st = d * d + d + 3;
  for (size_t k = 0; k < ds->pose_pose.size(); ++k) {
    const Meas &e = ds->pose_pose[k];
    pp_ids[2 * k] = e.p1;
    pp_ids[2 * k + 1] = e.p2;
    double *q = pp_vals + k * st;
    for (int i = 0; i < d * d; ++i) q[i] = e.R[i];
    for (int i = 0; i < d; ++i) q[d * d + i] = e.t[i];
    q[d * d + d] = e.kappa;
    q[d * d + d + 1] = e.tau;
    q[d * d + d + 2] = e.weight;
  }
  for (size_t k = 0; k < ds->pose_landmark.size(); ++k) {
    const PoseLandmarkMeas &e = ds->pose_landmark[k];
    pl_ids[2 * k] = e.i;
    pl_ids[2 * k + 1] = e.j;
    double *q = pl_vals + k * (d + 2);
    for (int i = 0; i < d; ++i) q[i] = e.t[i];
    q[d] = e.tau;
    q[d + 1] = e.weight;
  }
  for (size_t k = 0; k < ds->ranges.size(); ++k) {
    const RangeMeas &e = ds->ranges[k];
    int *q = r_ids + 5 * k;
    q[0] = e.type1;
    q[1] = e.i;
    q[2] = e.type2;
    q[3] = e.j;
    q[4] = e.l;
    r_vals[3 * k] = e.range;
    r_vals[3 * k + 1] = e.precision;
    r_vals[3 * k + 2] = e.weight;
  }
  std::copy(ds->gt.a.begin(), ds->gt.a.end(), gt);
}
void *orc_build_Q_ra(void *h) { return new CSR(build_Q_ra(*(RADataset *)h)); }
// out: d x k, RA ordering
void orc_ra_odometry_init(void *h, unsigned long long seed, double *out) {
  const Mat X = ra_odometry_initialization(*(RADataset *)h, seed);
  std::copy(X.a.begin(), X.a.end(), out);
}
void orc_ra_free(void *h) { delete (RADataset *)h; }

// ---- RBCD driver ---------------------------------------------------------------
// opts: [num_robots, r_min, r_max, max_iters, min_eig_tol, rgrad_tol, acceleration, staircase, verbose,
//        method, gradnorm_tol, RGD_stepsize, RGD_use_precond, RTR_iterations, RTR_tCG_iterations, RTR_initial_radius]
void *orc_run_rbcd(void *dsh, const double *opts, const double *X0, int x0_rows) {
  Dataset *ds = (Dataset *)dsh;
  RBCDOptions o;
  o.num_robots = (int)opts[0];
  o.r_min = (int)opts[1];
  o.r_max = (int)opts[2];
  o.max_iters = (int)opts[3];
  o.min_eig_tol = opts[4];
  o.rgrad_tol = opts[5];
  o.acceleration = (int)opts[6];
  o.staircase = (int)opts[7];
  o.verbose = (int)opts[8];
  o.opt.method = (int)opts[9];
  o.opt.gradnorm_tol = opts[10];
  o.opt.RGD_stepsize = opts[11];
  o.opt.RGD_use_preconditioner = (int)opts[12];
  o.opt.RTR_iterations = (int)opts[13];
  o.opt.RTR_tCG_iterations = (int)opts[14];
  o.opt.RTR_initial_radius = opts[15];
  o.threads = (int)opts[16];
  Mat X0m = view_mat(x0_rows, (ds->d + 1) * ds->n, X0);
  return new RBCDTrace(run_rbcd(*ds, o, X0m));
}
// coloured simultaneous updates: opts as above (r_min = the rank; max_iters = sweeps)
void *orc_run_coloured(void *dsh, const double *opts, const double *X0, int x0_rows) {
  Dataset *ds = (Dataset *)dsh;
  RBCDOptions o;
  o.num_robots = (int)opts[0];
  o.r_min = (int)opts[1];
  o.opt.method = (int)opts[9];
  o.opt.gradnorm_tol = opts[10];
  o.opt.RGD_stepsize = opts[11];
  o.opt.RGD_use_preconditioner = (int)opts[12];
  o.opt.RTR_iterations = (int)opts[13];
  o.opt.RTR_tCG_iterations = (int)opts[14];
  o.opt.RTR_initial_radius = opts[15];
  o.threads = (int)opts[16];
  Mat X0m = view_mat(x0_rows, (ds->d + 1) * ds->n, X0);
  return new RBCDTrace(run_coloured(*ds, o, X0m, (int)opts[3]));
}
// info: [total_iters, final_rank, certified, theta, lambda_min, rbcd_seconds, cert_seconds, setup_seconds]
void orc_trace_info(void *h, double *info) {
  RBCDTrace *t = (RBCDTrace *)h;
  info[0] = t->total_iters;
  info[1] = t->final_rank;
  info[2] = t->certified;
  info[3] = t->theta;
  info[4] = t->lambda_min;
  info[5] = t->rbcd_seconds;
  info[6] = t->cert_seconds;
  info[7] = t->setup_seconds;
}
void orc_trace_copy(void *h, double *cost, double *gradnorm, int *selected, int *rank, double *Xfinal) {
  RBCDTrace *t = (RBCDTrace *)h;
  std::copy(t->cost.begin(), t->cost.end(), cost);
  std::copy(t->gradnorm.begin(), t->gradnorm.end(), gradnorm);
  std::copy(t->selected.begin(), t->selected.end(), selected);
  std::copy(t->rank.begin(), t->rank.end(), rank);
  if (Xfinal) std::copy(t->Xfinal.a.begin(), t->Xfinal.a.end(), Xfinal);
}
void orc_trace_seconds(void *h, double *seconds) {
  RBCDTrace *t = (RBCDTrace *)h;
  std::copy(t->seconds.begin(), t->seconds.end(), seconds);
}
void orc_trace_free(void *h) { delete (RBCDTrace *)h; }

// ---- robust estimation ------------------------------------------------------------------------------------------------
// prm: [type, GNCMaxNumIters, GNCBarc, GNCMuStep, GNCInitMu, HuberThreshold, TLSThreshold]
static RobustParams view_robust(const double *prm) {
  RobustParams p;
  p.type = (RobustType)(int)prm[0];
  p.GNCMaxNumIters = (int)prm[1];
  p.GNCBarc = prm[2];
  p.GNCMuStep = prm[3];
  p.GNCInitMu = prm[4];
  p.HuberThreshold = prm[5];
  p.TLSThreshold = prm[6];
  return p;
}
// weights of n residuals after `updates` calls of RobustCost::update()
void orc_robust_weights(const double *prm, int updates, int n, const double *r, double *w) {
  RobustCost c(view_robust(prm));
  for (int i = 0; i < updates; ++i) c.update();
  for (int i = 0; i < n; ++i) w[i] = c.weight(r[i]);
}
double orc_chi2inv(double q, int dof) { return chi2inv(q, dof); }
double orc_error_threshold_at_quantile(double q, int dim) { return error_threshold_at_quantile(q, dim); }
// inlier: n flags
void orc_robust_rotation_averaging(int d, int n, const double *R, const double *kappa, double thr, double *Ropt,
                                   int *inlier) {
  std::vector<int> in;
  robust_single_rotation_averaging(d, n, R, kappa, thr, Ropt, in);
  for (int i = 0; i < n; ++i) inlier[i] = 0;
  for (int i : in) inlier[i] = 1;
}
void orc_robust_pose_averaging(int d, int n, const double *R, const double *t, const double *kappa, const double *tau,
                               double thr, double *Ropt, double *topt, int *inlier) {
  std::vector<int> in;
  robust_single_pose_averaging(d, n, R, t, kappa, tau, thr, Ropt, topt, in);
  for (int i = 0; i < n; ++i) inlier[i] = 0;
  for (int i : in) inlier[i] = 1;
}
void orc_measurement_errors(void *dsh, int rows, const double *T, double *out) {
  Dataset *ds = (Dataset *)dsh;
  const Mat Tm = view_mat(rows, (ds->d + 1) * ds->n, T);
  for (size_t i = 0; i < ds->meas.size(); ++i) out[i] = measurement_error(ds->meas[i], ds->d, Tm);
}
// opt: [gradnorm_tol, RTR_iterations, RTR_tCG_iterations, RTR_initial_radius]; T0 may be null (chordal start)
static ROptParams view_opt(const double *opt) {
  ROptParams p;
  p.gradnorm_tol = opt[0];
  p.RTR_iterations = (int)opt[1];
  p.RTR_tCG_iterations = (int)opt[2];
  p.RTR_initial_radius = opt[3];
  return p;
}
void orc_solve_pgo(void *dsh, const double *opt, const double *T0, double *Tout) {
  Dataset *ds = (Dataset *)dsh;
  Mat T0m;
  if (T0) T0m = view_mat(ds->d, (ds->d + 1) * ds->n, T0);
  const Mat T = solve_pgo(*ds, view_opt(opt), T0 ? &T0m : nullptr);
  std::copy(T.a.begin(), T.a.end(), Tout);
}
// fixed: m flags (fixedWeight); weights_out: m
void orc_solve_robust_pgo(void *dsh, const double *opt, const double *rprm, const int *fixed, const double *T0,
                          double *Tout, double *weights_out) {
  Dataset *ds = (Dataset *)dsh;
  Mat T0m;
  if (T0) T0m = view_mat(ds->d, (ds->d + 1) * ds->n, T0);
  std::vector<char> fx(ds->meas.size());
  for (size_t i = 0; i < fx.size(); ++i) fx[i] = fixed[i] ? 1 : 0;
  const Mat T = solve_robust_pgo(*ds, view_opt(opt), view_robust(rprm), fx, T0 ? &T0m : nullptr);
  std::copy(T.a.begin(), T.a.end(), Tout);
  for (size_t i = 0; i < ds->meas.size(); ++i) weights_out[i] = ds->meas[i].weight;
}

// ---- rounding ------------------------------------------------------------------------------------------------------
// X r x (d+1) n (SE ordering), anchor r x (d+1); out d x (d+1) n
void orc_align_lifted_trajectory(int r, int d, int n, const double *X, const double *anchor, int global, double *out) {
  Mat T;
  align_lifted_trajectory_to_frame(view_mat(r, (d + 1) * n, X), view_mat(r, d + 1, anchor), d, n, global != 0, T);
  std::copy(T.a.begin(), T.a.end(), out);
}
// X r x k (RA ordering); out d x k
void orc_project_solution_raslam(int r, int d, int n, int l, int b, const double *X, double *out) {
  Dims dm;
  dm.r = r; dm.d = d; dm.n = n; dm.l = l; dm.b = b;
  Mat P;
  project_solution_raslam(view_mat(r, dm.k(), X), dm, P);
  std::copy(P.a.begin(), P.a.end(), out);
}
// X r x k (RA ordering); traj d x (d+1) n (SE ordering), spheres d x l, landmarks d x b
void orc_ra_states_in_local_frame(int r, int d, int n, int l, int b, const double *X, double *traj, double *spheres,
                                  double *landmarks) {
  Dims dm;
  dm.r = r; dm.d = d; dm.n = n; dm.l = l; dm.b = b;
  Mat T, S, Lm;
  ra_states_in_local_frame(view_mat(r, dm.k(), X), dm, T, S, Lm);
  std::copy(T.a.begin(), T.a.end(), traj);
  std::copy(S.a.begin(), S.a.end(), spheres);
  std::copy(Lm.a.begin(), Lm.a.end(), landmarks);
}

}  // extern "C"
