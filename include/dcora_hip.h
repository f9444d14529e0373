/*
 * dcora_hip.h -- C ABI of the MI355X (gfx950) implementation of DCORA's hot path.
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++/torch types.
 * Every entry point names the reference interface it replaces (paths relative
 * to the reference tree).  Conventions shared by all calls:
 *
 *   - dense matrices are column-major double with leading dimension = rows
 *     (Eigen::MatrixXd layout, ref include/DCORA/DCORA_types.h:34);
 *   - sparse matrices are row-major CSR, int32 indices, both triangles stored
 *     (ref include/DCORA/DCORA_types.h:36);
 *   - column ordering of the lifted variable X (r x k): SE ordering
 *     [Y1 p1 ... Yn pn] when l == b == 0, otherwise RA ordering
 *     [Y1..Yn | s1..sl | p1..pn | L1..Lb]
 *     (ref src/manifold/LiftedVariable.cpp:74-106, 257-295);
 *   - host pointers are borrowed for the duration of the call only; handles own
 *     their device memory; *_dev variants take device pointers and enqueue on
 *     the handle's HIP stream without synchronising;
 *   - every function returns a dcora_status (0 = ok); nothing throws across
 *     the ABI.  The library fails loudly (DCORA_ERR_NO_DEVICE / DCORA_ERR_HIP)
 *     when no gfx950 device is usable: there is no CPU fallback.
 */
#ifndef DCORA_HIP_H_
#define DCORA_HIP_H_

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  DCORA_OK = 0,
  DCORA_ERR_BAD_ARG = 1,
  DCORA_ERR_NO_DEVICE = 2,
  DCORA_ERR_HIP = 3,
  DCORA_ERR_NOT_PD = 4,          /* Cholesky hit a non-positive pivot */
  DCORA_ERR_NO_CONVERGENCE = 5,  /* Lanczos did not converge */
  DCORA_ERR_NO_PRECONDITIONER = 6,
  DCORA_ERR_IO = 7,
  DCORA_ERR_UNSUPPORTED = 8,
  DCORA_ERR_EXCHANGE_LINK = 9    /* dcora_exchange_create: no transport between the ranks passed its start-up check */
} dcora_status;

const char *dcora_status_string(int status);
const char *dcora_last_error(void);
/* number of usable HIP devices (0 when there is none; never initialises more than the runtime) */
int dcora_device_count(void);

/* Manifold shape (ref src/manifold/LiftedManifold.cpp:18-89) and column ordering (ref src/manifold/LiftedVariable.cpp:
 * 74-106 SE: pose i = columns [i (d+1), i (d+1) + d]; :257-295 RA: rotations | unit spheres | translations | landmarks).
 * The reference chooses the manifold by GRAPH TYPE (ref src/Graph.cpp:68-75, src/QuadraticProblem.cpp:19-34): a
 * RangeAidedSLAMGraph that happens to hold neither ranges nor landmarks still uses the RA ordering.  `layout` carries that
 * choice; brace-initialising the first five members leaves it at DCORA_LAYOUT_AUTO. */
enum {
  DCORA_LAYOUT_AUTO = 0, /* SE ordering when l = b = 0, RA ordering otherwise */
  DCORA_LAYOUT_SE = 1,   /* pose graph (l and b must be 0) */
  DCORA_LAYOUT_RA = 2    /* range-aided ordering whatever l and b are */
};
typedef struct {
  int r; /* relaxation rank */
  int d; /* 2 or 3 */
  int n; /* poses (Stiefel blocks) */
  int l; /* unit spheres (oblique columns) */
  int b; /* landmarks */
  int layout; /* DCORA_LAYOUT_* */
} dcora_dims;

/* ref include/DCORA/DCORA_types.h:152-168 ROptParameters (same defaults via dcora_ropt_params_default) */
typedef struct {
  int method; /* 0 = RTR, 1 = RGD */
  int verbose;
  double gradnorm_tol;
  double RGD_stepsize;
  int RGD_use_preconditioner;
  int RTR_iterations;
  int RTR_tCG_iterations;
  double RTR_initial_radius;
} dcora_ropt_params;
void dcora_ropt_params_default(dcora_ropt_params *p);

/* ref include/DCORA/DCORA_types.h:203-233 ROPTResult (+ solver counters) */
typedef struct {
  int success;
  double fInit, gradNormInit, fOpt, gradNormOpt, elapsedMs;
  int tCGStatus; /* 0 NEGCURVTURE, 1 EXCREGION, 2 LCON, 3 SCON, 4 MAXITER */
  int outer_iterations, inner_iterations, accepted_steps;
} dcora_ropt_result;

/* ------------------------------------------------------------------------- *
 * QuadraticProblem  (replaces src/QuadraticProblem.cpp)
 * ------------------------------------------------------------------------- */
typedef struct dcora_problem_s *dcora_problem_t;

/* QuadraticProblem(shared_ptr<Graph>) (ref src/QuadraticProblem.cpp:19-34): uploads Q (k x k CSR) and G
 * (r x k, may be NULL = zero) and, when precond_reg >= 0, builds the preconditioner (Q + reg I)^-1
 * (ref src/Graph.cpp:1901-1917; reg = 0.1 for PGO).  device = HIP device ordinal. */
int dcora_problem_create(const dcora_dims *dims, const int *rowptr, const int *colidx, const double *vals,
                         const double *G, double precond_reg, int device, dcora_problem_t *out);
int dcora_problem_destroy(dcora_problem_t p);
/* Graph::linearMatrix() changed (ref src/Graph.cpp:535-540, 685-822) */
int dcora_problem_set_linear_term(dcora_problem_t p, const double *G);
/* f(Y) (ref src/QuadraticProblem.cpp:38-44) */
int dcora_problem_cost(dcora_problem_t p, const double *X, double *f);
/* EucGrad = X Q + G (ref :53-59) */
int dcora_problem_eucgrad(dcora_problem_t p, const double *X, double *out);
/* RieGrad / RieGradNorm (ref :86-123); out may be NULL */
int dcora_problem_riegrad(dcora_problem_t p, const double *X, double *out, double *norm);
/* Riemannian Hessian-vector product: EucHessianEta (ref :61-68) followed by ROPTLIB's EucHvToHv */
int dcora_problem_hessvec(dcora_problem_t p, const double *X, const double *V, double *out);
/* Test hook: the same product as the generic-layout solver loop forms it (delta Q, EucHvToHv and the partial sums of
 * <V, H V> in ONE launch, k_spmm_dir_fix); dots[0] = <V, H V> from that kernel's partials, dots[1] = the same from the
 * two-launch form.  DCORA_ERR_UNSUPPORTED when the one-launch form does not apply (a long row on a manifold column). */
int dcora_debug_hessvec_solver_form(dcora_problem_t p, const double *X, const double *V, double *out, double *dots);
/* How the tCG iteration of the local solver runs on this problem: info[0] = 0 three launches per
 * iteration (or the sparse preconditioner), 1 = two launches (Hessian product; step + dense preconditioner + projection),
 * 2 = the whole tCG run of an RTR iteration in ONE launch (dense preconditioner, n / 2 workgroups co-resident). */
int dcora_problem_solver_info(dcora_problem_t p, double *info);
/* test hook: the next `runs` one-launch tCG runs of this process lose a workgroup before their first grid-wide step, as
 * if the grid were not co-resident: the run must give up within milliseconds and the solve continue on the launches */
int dcora_debug_tcg_run_fault(int runs);
/* the same after `skip` launches of the run kernel that pass untouched (to hit the LAST iteration of a solve) */
int dcora_debug_tcg_run_fault_at(int skip, int runs);
/* PreCondition (ref :70-84, 261-297) */
int dcora_problem_precondition(dcora_problem_t p, const double *X, const double *V, double *out);
/* Retract (ref :125-136, 236-259) */
int dcora_problem_retract(dcora_problem_t p, const double *X, const double *V, double *out);
/* projectToTangentSpace (ref :299-307; src/manifold/LiftedManifold.cpp:37-41, 104-108) */
int dcora_problem_tangent_project(dcora_problem_t p, const double *X, const double *V, double *out);
/* escapeSaddle (ref :138-234): p is the problem at rank r, Xopt is (r-1) x k.  is_second_order selects the first
 * trial step: 0 => 1 (the header's default), 1 => max(16e-6, 100 gradient_tolerance / |theta|) (ref :165-169) */
int dcora_problem_escape_saddle(dcora_problem_t p, const double *Xopt, double theta, const double *v,
                                double gradient_tolerance, double preconditioned_gradient_tolerance,
                                int is_second_order, double *Xout, int *success);
/* LiftedSEManifold::project / LiftedRAManifold::project, projectToSEMatrix / projectToRAMatrix
 * (ref src/manifold/LiftedManifold.cpp:28-35, 91-102; src/DCORA_utils.cpp:2201-2220) */
int dcora_manifold_project(const dcora_dims *dims, const double *M, double *out, int device);

/* ------------------------------------------------------------------------- *
 * QuadraticOptimizer  (replaces src/QuadraticOptimizer.cpp)
 * ------------------------------------------------------------------------- */
/* optimize(Y) (ref src/QuadraticOptimizer.cpp:28-50): RTR (ref :52-108, 234-280) or one RGD step (ref :110-180) */
int dcora_optimizer_optimize(dcora_problem_t p, const dcora_ropt_params *params, const double *X0, double *Xout,
                             dcora_ropt_result *result);

/* ------------------------------------------------------------------------- *
 * Certification  (replaces src/DCORA_utils.cpp:1713-1982)
 * ------------------------------------------------------------------------- */
typedef struct dcora_csr_s *dcora_csr_t; /* host-side CSR returned by the library */
int dcora_csr_info(dcora_csr_t m, int *n, int *nnz);
int dcora_csr_copy(dcora_csr_t m, int *rowptr, int *colidx, double *vals);
int dcora_csr_destroy(dcora_csr_t m);

/* constructDualCertificateMatrixPGO / ...RASLAM (ref :1898-1982): S = Q - Lambda(X) */
int dcora_cert_dual_matrix(const dcora_dims *dims, const double *X, const int *rowptr, const int *colidx,
                           const double *vals, int device, dcora_csr_t *S);
/* isSparseSymmetricMatrixPSD (ref :1737-1747) */
int dcora_cert_is_psd(int k, const int *rowptr, const int *colidx, const double *vals, int block, int *is_psd);
/* the same test with the numeric factorisation on the device (multifrontal LL^T over the nested-dissection pieces,
 * dcora_amd/csrc/device_chol.h; the symbolic analysis is cached on the sparsity pattern).  This is what
 * dcora_cert_fast_verification runs.  info8 (optional): symbolic ms (0 on a cache hit), numeric ms, bytes of the
 * front arena, factorisation flops, tree levels, kernel launches, log det of the matrix (when PD), ms of the pattern
 * hash + cache look-up. */
int dcora_cert_is_psd_device(int k, const int *rowptr, const int *colidx, const double *vals, int block, int device,
                             int *is_psd, double *info8);
/* validation of the symbolic analysis without a GPU: the schedule of dcora_cert_is_psd_device executed by plain host
 * loops; *resid = max |P A P^T - L L^T| (dense check, k <= 4096).  Test utility, not a product path. */
int dcora_chol_host_selftest(int k, const int *rowptr, const int *colidx, const double *vals, int block, int *is_pd,
                             double *resid, double *info4);
int dcora_chol_cache_clear(void);
/* computeMinimumEigenPair(S, max_iterations, min_eig_num_tol, num_Lanczos_vectors) (ref :1809-1896) */
int dcora_cert_min_eig(int k, const int *rowptr, const int *colidx, const double *vals, int max_iterations,
                       double min_eig_num_tol, int num_lanczos_vectors, unsigned long long seed, int device,
                       double *lambda_min, double *v, long *num_matvecs);
/* fastVerification(S, eta, &theta, &x) (ref :1713-1735) */
int dcora_cert_fast_verification(int k, const int *rowptr, const int *colidx, const double *vals, double eta,
                                 int block, int device, int *is_psd, double *theta, double *x, double *lambda_min);

/* Suboptimality bound that goes with the certificate -- an ADDITION of this library: the reference reports the
 * boolean and theta only (ref examples/MultiRobotExample.cpp:321-352).  SE-Sync form: at a first-order critical point
 * X with certificate matrix S = Q - Lambda(X), every feasible Z = X'^T X' satisfies f(X) - f(X') = -1/2 <S, Z>
 * <= -1/2 lambda tr(Z) for any lower bound lambda <= min(lambda_min(S), 0).  tr(Z) is evaluated at X itself with the
 * translation / landmark columns centred (the cost does not see a common shift of them): n_eff = |rotation and
 * sphere columns|_F^2 + sum_i |p_i - mean p|^2, so gap = -1/2 lambda n_eff is exact for the rotation part and an
 * estimate for comparison points whose translations spread more than X's.  Pass lambda = -eta after a positive
 * dcora_cert_fast_verification(S, eta) (then S + eta I >= 0), or lambda_min - eta from its lambda_min output otherwise.
 * gap is in units of f (half of the "cost" the reference driver prints); n_eff may be NULL. */
int dcora_cert_suboptimality_gap(const dcora_dims *dims, const double *X, double lambda_lower_bound, double *gap,
                                 double *n_eff);
/* The eta-test only says lambda_min(S) >= -eta, and eta n_eff can exceed the cost itself when the trajectory is large
 * (n_eff grows with the spread of the translations).  After an accepted certificate this returns a CERTIFIED lower
 * bound of lambda_min(S): Lanczos with full re-orthogonalisation on (S + eta I)^-1 (the matrix the PSD test
 * factorised; sparse Cholesky on the host) until the Ritz value changes by less than 0.1 % (at most max_iterations
 * solves) gives the candidate 1 / (theta + Ritz residual) - eta, which is lowered by 0.1 % and then VERIFIED like the
 * certificate itself: S - lambda I must have a Cholesky factorisation.  The verified shift is returned; when the
 * verification fails, -eta (what the accepted certificate guarantees by itself).  DCORA_ERR_NOT_PD when S + eta I is
 * not positive definite.  An addition of this library.  lambda_lower_bound = min(lambda_min, 0) in
 * dcora_cert_suboptimality_gap gives a certified gap. */
int dcora_cert_lambda_min_certified(int k, const int *rowptr, const int *colidx, const double *vals, double eta,
                                    int block, int max_iterations, double *lambda_min, int *iterations);

/* ------------------------------------------------------------------------- *
 * Data feed  (replaces the parts of src/Graph.cpp / src/DCORA_utils.cpp that
 * produce Q, G and the measurement list)
 * ------------------------------------------------------------------------- */
/* measurement arrays: ids m x 4 int32 (r1, p1, r2, p2); vals m x (d*d + d + 3) double
 * (R column-major, t, kappa, tau, weight)  (ref include/DCORA/Measurements.h RelativePosePoseMeasurement) */
typedef struct dcora_dataset_s *dcora_dataset_t;
/* read_g2o_file (ref src/DCORA_utils.cpp:179-375); .gz not handled here */
int dcora_dataset_load_g2o(const char *path, dcora_dataset_t *out);
int dcora_dataset_create(int d, int n, int m, const int *ids, const double *vals, dcora_dataset_t *out);
int dcora_dataset_info(dcora_dataset_t ds, int *d, int *n, int *m);
int dcora_dataset_copy(dcora_dataset_t ds, int *ids, double *vals);
int dcora_dataset_destroy(dcora_dataset_t ds);
/* chordalInitialization (ref src/DCORA_solver.cpp:218-268): T is d x (d+1) n column-major (SE ordering), the start
 * point of the reference driver's InitializationMethod::Chordal (ref examples/MultiRobotExample.cpp:150-153) */
int dcora_dataset_chordal_init(dcora_dataset_t ds, double *T);
/* the same with the two sparse SPD systems solved on the device (partitioned inverse built from the device
 * factorisation, one replay each): for graphs whose host factorisation takes minutes (the 100k lattice: 237 s on the
 * host).  Same result up to the rounding of the solves. */
int dcora_dataset_chordal_init_device(dcora_dataset_t ds, int device, double *T);
/* Graph::constructQuadraticCostTermPGO (ref src/Graph.cpp:579-683) for agent `agent_id` owning n poses */
int dcora_graph_build_Q_pgo(int d, int n, int agent_id, int m, const int *ids, const double *vals, dcora_csr_t *Q);

/* Range-aided SLAM, centralised agent (CORA flow of examples/SingleRobotExample_RASLAM.cpp:59-130):
 * read_pyfg_file + getGlobalMeasurements (ref src/DCORA_utils.cpp:437-1167, 1169-1365) and
 * Graph::constructQuadraticCostTermRASLAM (ref src/Graph.cpp:824-1188).  Column ordering of X is the RA ordering. */
typedef struct dcora_radataset_s *dcora_radataset_t;
int dcora_radataset_load_pyfg(const char *path, dcora_radataset_t *out);
/* info[7] = {d, n poses, l unit spheres, b landmarks, #pose-pose, #pose-landmark, #range measurements} */
int dcora_radataset_info(dcora_radataset_t ds, int *info);
/* Graph::setMeasurements(const RelativeMeasurements &) of a range-aided graph (ref include/DCORA/Graph.h:57-147,
 * src/Graph.cpp:374-470): the dataset from measurement arrays instead of a file.  States are numbered as the Graph
 * numbers them: poses 0 .. n - 1, unit spheres 0 .. l - 1 (one per range measurement), landmarks 0 .. b - 1, all owned
 * by robot 0 (the centralised agent).
 *   pose-pose:     pp_ids m_pp x 2 (pose i, pose j); pp_vals m_pp x (d d + d + 3): R column-major, t, kappa, tau, weight
 *   pose-landmark: pl_ids m_pl x 2 (pose i, landmark j); pl_vals m_pl x (d + 2): t, tau, weight
 *   range:         rg_ids m_rg x 5 (type1, i, type2, j, unit sphere; type 0 = pose, 1 = landmark); rg_vals m_rg x 3:
 *                  range, precision, weight
 * gt: d x k ground truth in the RA ordering, or NULL.  dcora_radataset_copy writes a dataset's measurements into the
 * same arrays (any of them may be NULL; sizes from dcora_radataset_info). */
int dcora_radataset_create(int d, int n, int l, int b, int m_pp, const int *pp_ids, const double *pp_vals, int m_pl,
                           const int *pl_ids, const double *pl_vals, int m_rg, const int *rg_ids, const double *rg_vals,
                           const double *gt, dcora_radataset_t *out);
int dcora_radataset_copy(dcora_radataset_t ds, int *pp_ids, double *pp_vals, int *pl_ids, double *pl_vals, int *rg_ids,
                         double *rg_vals);
/* ground truth of the VERTEX records in RA ordering, d x k column-major (unit spheres = normalised state1 - state2) */
int dcora_radataset_ground_truth(dcora_radataset_t ds, double *gt);
int dcora_radataset_build_Q(dcora_radataset_t ds, dcora_csr_t *Q);
/* start point of the centralised CORA driver (ref examples/SingleRobotExample_RASLAM.cpp:92-150 with
 * odometryInitialization, ref src/DCORA_solver.cpp:270-302): odometry chains anchored at their ground-truth first
 * pose, ground-truth unit spheres, landmarks uniform in (-1, 1) from a splitmix64 stream seeded with `seed`
 * (the reference draws them with Matrix::Random).  X0 is d x k in the RA ordering. */
int dcora_radataset_odometry_init(dcora_radataset_t ds, unsigned long long seed, double *X0);
/* Ownership of the merged problem's variables, as the reference splits a multi-robot pyfg file
 * (getRobotMeasurements, ref src/DCORA_utils.cpp:1370-1512; landmark symbols, ref src/Graph.cpp:584-616; unit
 * spheres belong to the SOURCE robot of their range measurement, ref :1092-1097).  Robot ids: 'A' = 0, 'B' = 1,
 * ..., 'M' = 12 is the map.  Any output may be NULL. */
int dcora_radataset_ownership(dcora_radataset_t ds, int *pose_robot, int *sphere_robot, int *landmark_robot);
/* Columns of the global RA ordering owned by `robot`, listed in that agent's own RA ordering
 * [rotations | unit spheres | translations | landmarks]; dims3 = {n_a, l_a, b_a}; own (k entries of room) may be NULL */
int dcora_radataset_agent_columns(dcora_radataset_t ds, int robot, int *dims3, int *own, int *k_a);
/* An agent's share of a global quadratic form Q (k x k): Qaa = Q[own, own] in the agent's ordering and the coupling
 * C = Q[own, rest] (k_a x k, GLOBAL column indices), so that the agent's linear term is G_a = X_global C^T -- the
 * restriction the reference assembles measurement by measurement (ref src/Graph.cpp:824-1772) */
int dcora_graph_extract_agent_blocks(int k, const int *rowptr, const int *colidx, const double *vals, int k_a,
                                     const int *own, dcora_csr_t *Qaa, dcora_csr_t *C);
int dcora_radataset_destroy(dcora_radataset_t ds);
/* Graph::computePreconditionerRegularization (ref src/Graph.cpp:1921-1960): reg = lambda_max(Q) / (1e6 - 1), lambda_max
 * by Lanczos (nev 1, ncv 6, tol 1e-3) on the device; falls back to 0.1 when the eigensolver does not converge */
int dcora_graph_precond_regularization(int k, const int *rowptr, const int *colidx, const double *vals, int device,
                                       double *reg);

/* ------------------------------------------------------------------------- *
 * RBCD session: Agents + synchronous driver on device
 * (replaces Agent::iterate/updateX/getSharedStateDicts/updateNeighborStates, ref src/Agent.cpp:113-152,
 *  535-596, 844-906, 1158-1278, and the loop body of examples/MultiRobotExample.cpp:223-307)
 * ------------------------------------------------------------------------- */
typedef struct dcora_rbcd_s *dcora_rbcd_t;
typedef struct {
  int num_robots;
  int r;
  int acceleration;     /* AgentParameters::acceleration */
  int restart_interval; /* AgentParameters::restartInterval (30) */
  dcora_ropt_params local; /* AgentParameters::localOptimizationParams */
  int rank, world_size; /* this process hosts the agents a with a / ceil(num_robots / world_size) == rank:
                           consecutive agents share a rank, so the colours of a chain of agents spread evenly */
  int device;
  void *stream; /* hipStream_t the session enqueues on (e.g. the stream the caller's RCCL calls are ordered on, so
                   pack -> collective -> unpack needs no host synchronisation); NULL: the session creates its own */
} dcora_rbcd_options;
void dcora_rbcd_options_default(dcora_rbcd_options *o);

/* partition (ref examples/MultiRobotExample.cpp:56-118), per-agent Q / preconditioner, central Q */
int dcora_rbcd_create(dcora_dataset_t ds, const dcora_rbcd_options *opt, dcora_rbcd_t *out);
int dcora_rbcd_destroy(dcora_rbcd_t s);
/* Agent::setX for every agent from the global r x (d+1)n matrix (ref examples/MultiRobotExample.cpp:209-217) */
int dcora_rbcd_set_X(dcora_rbcd_t s, const double *X);
int dcora_rbcd_get_X(dcora_rbcd_t s, double *X);
/* one pass of the loop body :223-307 with `selected` as the optimising agent: non-selected iterate(false),
 * public-pose pull, selected iterate(true), central evaluation.  Outputs: 2 f, |rgrad|, per-agent |rgrad_b|
 * (num_robots doubles, may be NULL) and the greedy next selection. */
int dcora_rbcd_iterate(dcora_rbcd_t s, int selected, double *cost2, double *gradnorm, double *block_norms,
                       int *next_selected);
/* runs up to max_iters passes, stopping when |rgrad| < rgrad_tol; fills the trace arrays (may be NULL) */
int dcora_rbcd_run(dcora_rbcd_t s, int max_iters, double rgrad_tol, int *iters_done, double *cost2_trace,
                   double *gradnorm_trace, int *selected_trace);
/* One tick in which the `count` agents of `set` run Agent::iterate(true) at the same time, each on its own HIP
 * stream, every one of them reading the neighbour states as they were when the tick began: what agents that fire
 * together see in the asynchronous mode (Agent::runOptimizationLoop, ref src/Agent.cpp:650-678).  Needs
 * acceleration off, as that mode does (:651-653).  Agents of the set that share measurements are refused unless
 * allow_adjacent != 0; for mutually non-adjacent agents (one colour of dcora_rbcd_agent_colours) the tick equals
 * updating them one after the other with dcora_rbcd_iterate.  Multi-process: every rank passes the same set and
 * solves the agents it hosts; the caller exchanges public poses afterwards. */
int dcora_rbcd_iterate_set(dcora_rbcd_t s, const int *set, int count, int allow_adjacent);
/* AgentParameters::acceleration of every agent; restarts the Nesterov sequences (V = Y = X, gamma = alpha = 0) */
int dcora_rbcd_set_acceleration(dcora_rbcd_t s, int acceleration);
/* greedy colouring of the agent graph (agents adjacent when they share a measurement): colours[num_robots] */
int dcora_rbcd_agent_colours(dcora_rbcd_t s, int *colours, int *ncolours);
/* the central evaluation of dcora_rbcd_iterate alone (ref examples/MultiRobotExample.cpp:264-305); world_size 1 */
int dcora_rbcd_evaluate(dcora_rbcd_t s, double *cost2, double *gradnorm, double *block_norms, int *next_selected);
/* Agent::iterate(doOptimization) of ONE agent (ref src/Agent.cpp:535-596), for callers that keep the reference's
 * per-agent loop (examples/MultiRobotExample.cpp:223-262).  The agents of a session advance in lockstep: one call
 * per hosted agent and round; the first call of a round advances the shared Nesterov sequences. */
int dcora_rbcd_agent_iterate(dcora_rbcd_t s, int agent, int do_optimization);
/* Agent::updateNeighborStates (ref src/Agent.cpp:844-906): `count` public poses of `neighbor` (frames local to it;
 * poses: count blocks of r x (d+1), column-major) handed to `agent`.  From the first such call on the agent optimises
 * against what it was HANDED -- its own cache of neighbour poses, the plain one or (auxiliary != 0) the one used when
 * it starts from Y -- and no longer against the session's shared mirror: stale poses are used as they are, poses it
 * does not require are ignored (Graph::requireNeighborPose), and an iterate(true) whose cache misses a required pose
 * skips the optimisation (ref src/Agent.cpp:1243-1249; dcora_rbcd_agent_last_skipped tells).  Agents that are never
 * handed anything keep reading the mirror (the session-level loop, dcora_rbcd_iterate). */
int dcora_rbcd_agent_update_neighbor(dcora_rbcd_t s, int agent, int neighbor, int count, const int *frames,
                                     const double *poses, int auxiliary);
int dcora_rbcd_agent_last_skipped(dcora_rbcd_t s, int agent, int *skipped);
/* Agent::getX / setX: the agent's own block, r x (d+1) num_poses (ref src/Agent.cpp:64-77, 98-105) */
int dcora_rbcd_agent_get_X(dcora_rbcd_t s, int agent, double *X);
int dcora_rbcd_agent_set_X(dcora_rbcd_t s, int agent, const double *X);
/* num_poses, first global pose index and Agent::iteration_number() of an agent (any of the outputs may be NULL) */
int dcora_rbcd_agent_info(dcora_rbcd_t s, int agent, int *num_poses, int *first_pose, int *iteration_number);
/* statistics of the last selected agent's local solve */
int dcora_rbcd_last_result(dcora_rbcd_t s, dcora_ropt_result *res);

/* --- multi-process pieces (one process per GPU, exchange through RCCL by the caller) --- */
/* device pointer of the session's global lifted variable (r x (d+1)n, column-major, resident in HBM) */
int dcora_rbcd_X_device_ptr(dcora_rbcd_t s, double **X_dev);
/* number of public poses of agent a, and their global pose indices (ref src/Graph.h myPublicPoseIDs) */
int dcora_rbcd_public_count(dcora_rbcd_t s, int agent, int *count);
int dcora_rbcd_public_indices(dcora_rbcd_t s, int agent, int *global_pose_idx);
/* getSharedStateDicts: gather agent's public poses into a packed r x (d+1)count device buffer */
int dcora_rbcd_pack_public_dev(dcora_rbcd_t s, int agent, double *packed_dev);
/* updateNeighborStates: scatter a packed buffer received from agent's owner into the local mirror of X */
int dcora_rbcd_unpack_public_dev(dcora_rbcd_t s, int agent, const double *packed_dev);
/* the three phases of dcora_rbcd_iterate for callers that exchange between them */
int dcora_rbcd_phase_nonselected(dcora_rbcd_t s, int selected);
int dcora_rbcd_phase_selected(dcora_rbcd_t s, int selected);
/* local part of the evaluation: for every hosted agent b, |Proj(X_b Q_bb + G_b)|^2 and <X_b, X_b Q_bb + G_b>
 * written to out_dev[2*b], out_dev[2*b+1] (device, 2*num_robots doubles, entries of non-hosted agents zero) */
int dcora_rbcd_phase_evaluate_dev(dcora_rbcd_t s, double *out_dev);
int dcora_rbcd_synchronize(dcora_rbcd_t s);
/* Measurement hook (bench.py's `roofline`): while enabled, HIP events are recorded on the solver's stream around every
 * one-launch tCG run (k_tcg_run) of the session's agents; _read waits for the stream, returns {launches, sum of their
 * event times in us} since the last read and forgets them.  Not for timed regions: an event pair costs a few us. */
int dcora_rbcd_profile_tcg_runs(dcora_rbcd_t s, int enable);
int dcora_rbcd_profile_tcg_read(dcora_rbcd_t s, double *out2);

/* ------------------------------------------------------------------------- *
 * Neighbour exchange between the ranks of one node (one process per GPU): the transport the reference leaves to its
 * host -- Agent::getSharedStateDicts on the sender, Agent::updateNeighborStates on the receiver (ref src/Agent.cpp:
 * 113-152, 844-906), moved by the driver (ref examples/MultiRobotExample.cpp:236-258).  An exchange belongs to a
 * session created with rank / world_size; every rank of the job creates one under the same job name (unique per job
 * on the node: it names a POSIX shared-memory segment that carries the bootstrap, the flag words and the evaluation
 * scalars).  Pose data moves by direct stores into the halo buffer of each rank that hosts a NEIGHBOUR of the posting
 * agent (peer memory mapped through HIP IPC, xGMI between GPUs); there is no collective on the data path.  If the IPC
 * transport is not usable on some rank, all ranks stage the packed poses in the shared host segment instead
 * (dcora_exchange_info reports which).  All calls are SPMD: every rank makes the same calls with the same agent lists.
 * ------------------------------------------------------------------------- */
typedef struct dcora_exchange_s *dcora_exchange_t;
int dcora_exchange_create(dcora_rbcd_t s, const char *job_name, dcora_exchange_t *out);
int dcora_exchange_destroy(dcora_exchange_t ex);
/* info[10] = {transport (1 = IPC peer stores, 2 = shared host segment), ranks this rank stores to, posts, waits,
 * bytes posted so far, host seconds in post, host seconds in wait, host seconds waiting for the evaluation scalars,
 * 1 when this rank's halo buffer is fine-grained device memory (remote stores never served stale from the local L2),
 * 1 when the scatter kernel itself waits for the producer's flag (default where every rank has a GPU of its own and a
 * fine-grained halo buffer; DCORA_EXCHANGE_WAIT=host: the host spins)}.
 * dcora_exchange_create ends with a LINK CHECK when there is more than one rank: every rank stores a 4 KB pattern and a
 * flag into each rank it will write to, through the transport and the form of the wait it is about to use, and every
 * reader waits for the flag (bounded: 3 s) and compares the pattern word for word.  When any rank's check fails all
 * ranks step down together -- device-side wait -> host wait -> shared host segment instead of IPC peer stores -- and
 * check again; when nothing passes, the call returns DCORA_ERR_EXCHANGE_LINK on every rank within seconds instead of
 * hanging in the first post.  dcora_exchange_link_report says what was tried. */
int dcora_exchange_info(dcora_exchange_t ex, double *info10);
/* out[4] = {rounds of the link check run, 1 if the device-side wait was given up, 1 if the IPC transport was given up,
 * microseconds of the last (passed) round} */
int dcora_exchange_link_report(dcora_exchange_t ex, double *out4);
/* getSharedStateDicts of `agents`: each hosted one is written to its neighbours' ranks and flagged (one kernel per
 * agent on the session's stream; returns without synchronising).  A slot is re-used two posts later: the call first
 * waits until every reading rank has scattered that older post (an error after 120 s, never a torn read), so a post
 * may run at most two ahead of the dcora_exchange_wait calls of its readers. */
int dcora_exchange_post(dcora_exchange_t ex, const int *agents, int count);
/* updateNeighborStates: waits until the posts of those of `agents` that neighbour an agent hosted here have
 * arrived and scatters them into the session's mirror of X (enqueued on the session's stream) */
int dcora_exchange_wait(dcora_exchange_t ex, const int *agents, int count);
/* the driver's evaluation (ref examples/MultiRobotExample.cpp:264-305) without a central copy of X: every rank
 * evaluates |Proj(X_b Q_bb + G_b)| and <X_b, X_b Q_bb + G_b> of its agents, 2 R scalars are all-gathered through
 * the shared segment; same outputs as dcora_rbcd_evaluate, identical on every rank */
int dcora_exchange_evaluate(dcora_exchange_t ex, double *cost2, double *gradnorm, double *block_norms,
                            int *next_selected);
/* dcora_rbcd_iterate across the ranks: non-selected updates, post + wait, the selected agent's solve on its rank,
 * post + wait of its new public poses, evaluation */
int dcora_exchange_rbcd_iterate(dcora_exchange_t ex, int selected, double *cost2, double *gradnorm,
                                double *block_norms, int *next_selected);
/* dcora_rbcd_iterate_set across the ranks, followed by post + wait of the updated agents */
int dcora_exchange_rbcd_tick(dcora_exchange_t ex, const int *set, int count, int allow_adjacent);
/* dcora_rbcd_set_X on every rank between two barriers; dcora_rbcd_get_X of the whole X assembled from the ranks that
 * host each block (collective; X is valid on every rank) */
int dcora_exchange_set_X(dcora_exchange_t ex, const double *X);
int dcora_exchange_gather_X(dcora_exchange_t ex, double *X);
/* host barrier over the ranks of the job (does not synchronise the device) */
int dcora_exchange_barrier(dcora_exchange_t ex);
/* Agent::shouldTerminate's team condition (ref src/Agent.cpp:1137-1153: terminate only when EVERY robot reports
 * readyToTerminate): the AND of the ranks' flags through the shared segment.  SPMD; *all_ready is the same everywhere. */
int dcora_exchange_all_ready(dcora_exchange_t ex, int ready, int *all_ready);
/* fastVerification of the current iterate across the ranks (ref src/DCORA_utils.cpp:1713-1735 on the global
 * S = Q - Lambda(X); the driver calls it where examples/MultiRobotExample.cpp:329-330 does).  SPMD.  The global
 * connection Laplacian Q (k = (d+1) n, CSR) is needed on rank 0 only (NULL elsewhere): rank 0 assembles S from the
 * gathered X and runs the PSD test (Cholesky of S + eta I on its device).  When that fails the minimum eigenpair of
 * S + eta I is computed by ALL ranks: each applies the row block of S of the agents it hosts (Q_bb, the coupling
 * blocks, the Lambda blocks of its poses) to its slice of the Lanczos vectors, the public entries travel between
 * neighbouring ranks like public poses, and every inner product is summed over the ranks through the shared segment in
 * rank order (the same bits everywhere).  *certified: 1 = S + eta I >= 0; otherwise *theta = v^T S v (the curvature
 * escapeSaddle takes), *lambda_min = the eigenvalue of S + eta I, v (NULL or k doubles) its unit eigenvector, the same
 * on every rank; *distributed: 1 when the eigenpair came from the row-block Lanczos runs, 0 when they did not
 * converge and rank 0's shift-and-invert fallback (a factorisation of the whole matrix) supplied it. */
int dcora_exchange_certify(dcora_exchange_t ex, int k, const int *rowptr, const int *colidx, const double *vals,
                           double eta, int *certified, double *theta, double *lambda_min, double *v, long long *matvecs,
                           int *distributed);
/* the host half of the protocol alone, without a device: bootstrap through the shared segment, barriers, and
 * `rounds` rounds of post -> wait -> evaluation all-gather with host stores in the device's place; every rank of the
 * job calls it, *checksum comes out identical on all of them.  For multi-process tests on machines without a GPU. */
int dcora_exchange_host_selftest(const char *job_name, int rank, int world_size, int num_agents, int rounds,
                                 double *checksum);
/* test hook: leaves under the job's name what a crashed job of the same shape would (an initialised segment whose
 * creator is gone); a job started afterwards under that name must not attach to it */
int dcora_debug_exchange_leave_stale(const char *job_name, int world_size, int num_agents);
/* test hook: the link check of exchanges created afterwards in this process reports its first `rounds` rounds as failed
 * on the last rank (every rank of the job calls it with the same value), so the step-down ladder and
 * DCORA_ERR_EXCHANGE_LINK can be exercised on healthy hardware */
int dcora_debug_exchange_probe_fault(int rounds);

/* ------------------------------------------------------------------------- *
 * RBCD session for multi-robot range-aided SLAM (replaces the Agents on a RangeAidedSLAMGraph and the loop body of
 * examples/MultiRobotExample_RASLAM.cpp; ref src/Agent.cpp:535-596, 1158-1278, src/Graph.cpp:824-1772).
 * Agents = the robots of the pyfg file that own poses, in id order; variables are owned as the reference assigns
 * them (dcora_radataset_ownership); X is the merged problem's r x k matrix in the RA ordering.  Files in which the
 * passive map agent would own a landmark or a unit sphere are refused (DCORA_ERR_UNSUPPORTED).  One process.
 * ------------------------------------------------------------------------- */
typedef struct dcora_ra_rbcd_s *dcora_ra_rbcd_t;
/* opt->num_robots is ignored (the file decides); opt->local are the agents' localOptimizationParams */
int dcora_ra_rbcd_create(dcora_radataset_t ds, const dcora_rbcd_options *opt, dcora_ra_rbcd_t *out);
/* the same exchange for the agents of a multi-robot range-aided SLAM problem (ref examples/MultiRobotExample_RASLAM.cpp;
 * ownership by robot symbol, src/DCORA_utils.cpp:1370-1512): the session was created by dcora_ra_rbcd_create with
 * rank / world_size (agent i on rank i / ceil(R / world_size)); an agent's public variables are its poses, unit spheres
 * and landmarks that other agents' measurements reach.  dcora_exchange_set_X / _rbcd_iterate / _evaluate / _gather_X /
 * _post / _wait / _all_ready work as for pose graphs; _rbcd_tick and _certify are pose-graph calls (the range-aided
 * certificate is assembled from the gathered X: dcora_cert_dual_matrix + dcora_cert_fast_verification). */
int dcora_exchange_create_ra(dcora_ra_rbcd_t s, const char *job_name, dcora_exchange_t *out);
int dcora_ra_rbcd_destroy(dcora_ra_rbcd_t s);
/* number of agents and (robots != NULL) their robot ids ('A' = 0, ...) */
int dcora_ra_rbcd_info(dcora_ra_rbcd_t s, int *num_agents, int *robots);
int dcora_ra_rbcd_set_X(dcora_ra_rbcd_t s, const double *X);
int dcora_ra_rbcd_get_X(dcora_ra_rbcd_t s, double *X);
/* as dcora_rbcd_iterate / _evaluate / _run / _last_result; `selected` indexes the agents of dcora_ra_rbcd_info */
int dcora_ra_rbcd_iterate(dcora_ra_rbcd_t s, int selected, double *cost2, double *gradnorm, double *block_norms,
                          int *next_selected);
int dcora_ra_rbcd_evaluate(dcora_ra_rbcd_t s, double *cost2, double *gradnorm, double *block_norms,
                           int *next_selected);
int dcora_ra_rbcd_run(dcora_ra_rbcd_t s, int max_iters, double rgrad_tol, int *iters_done, double *cost2_trace,
                      double *gradnorm_trace, int *selected_trace);
int dcora_ra_rbcd_last_result(dcora_ra_rbcd_t s, dcora_ropt_result *res);

/* ------------------------------------------------------------------------- *
 * Robust estimation (replaces src/DCORA_robust.cpp and the robust parts of src/DCORA_solver.cpp)
 * ------------------------------------------------------------------------- */
/* RobustCostParameters::Type (ref include/DCORA/DCORA_robust.h:28-35) */
typedef enum {
  DCORA_ROBUST_L2 = 0,
  DCORA_ROBUST_L1 = 1,
  DCORA_ROBUST_TLS = 2,
  DCORA_ROBUST_HUBER = 3,
  DCORA_ROBUST_GM = 4,
  DCORA_ROBUST_GNC_TLS = 5
} dcora_robust_type;
/* RobustCostParameters (ref :25-60); defaults: L2, 20 GNC iterations, barc 5, mu step 1.4, mu init 1e-4, Huber 3, TLS 10 */
typedef struct {
  int cost_type;
  int GNCMaxNumIters;
  double GNCBarc, GNCMuStep, GNCInitMu;
  double HuberThreshold, TLSThreshold;
} dcora_robust_params;
void dcora_robust_params_default(dcora_robust_params *p);
/* RobustCost::weight(r) for n residuals after num_updates calls of RobustCost::update() (ref src/DCORA_robust.cpp:56-136) */
int dcora_robust_weights(const dcora_robust_params *p, int num_updates, int n, const double *r, double *w);
/* chi2inv (ref src/DCORA_utils.cpp:2103-2106), RobustCost::computeErrorThresholdAtQuantile (ref src/DCORA_robust.cpp:138-148) */
int dcora_chi2inv(double quantile, int dof, double *out);
int dcora_robust_error_threshold_at_quantile(double quantile, int dimension, double *out);
/* robustSingleRotationAveraging / robustSinglePoseAveraging (ref src/DCORA_solver.cpp:76-216): R holds n rotations
 * d x d column-major back to back, t n translations; kappa / tau may be NULL (1 resp. 10000 / 100 as in the
 * reference); inlier receives n flags */
int dcora_robust_single_rotation_averaging(int d, int n, const double *R, const double *kappa, double error_threshold,
                                           double *Ropt, int *inlier);
int dcora_robust_single_pose_averaging(int d, int n, const double *R, const double *t, const double *kappa,
                                       const double *tau, double error_threshold, double *Ropt, double *topt,
                                       int *inlier);
/* computeMeasurementError of every measurement of the dataset at once (ref src/DCORA_utils.cpp:2095-2101;
 * Agent::computeMeasurementResidual, ref src/Agent.cpp:1342-1389): X is r x (d+1) n (SE ordering, r >= d, lifted or
 * not); out[i] = kappa |Y1 R - Y2|^2 + tau |p2 - p1 - Y1 t|^2, weights not applied */
/* ---- cross-robot frame alignment (ref src/Agent.cpp:460-520, 694-833); poses d x (d+1) column-major [R t] ---- */
/* Agent::computeNeighborTransform for m inter-robot loop closures: incoming[i] != 0 when this robot is the
 * measurement's second pose; nbr_pose[i] the neighbour's pose of the closure in the neighbour's (world2) frame,
 * my_pose[i] my pose of it in my (world1) frame; T_out[i] = T_world2_world1 implied by closure i */
int dcora_agent_neighbor_transforms(int d, int m, const int *incoming, const double *meas_R, const double *meas_t,
                                    const double *nbr_pose, const double *my_pose, double *T_out);
/* Agent::computeRobustNeighborTransform (two_stage = 0) / computeRobustNeighborTransformTwoStage (two_stage != 0)
 * over m candidate transforms; *ok = 0 when fewer than min_inliers (AgentParameters::robustInitMinInliers) agree */
int dcora_agent_robust_neighbor_transform(int d, int m, const double *candidates, int two_stage, int min_inliers,
                                          double *T_world_robot, int *num_inliers, int *ok);
/* Logger::logTrajectory (ref src/Logger.cpp:107-145): "# pose_index x y z qx qy qz qw", one line per pose, 9 decimals;
 * T is d x (d+1) n (SE ordering); planar poses are embedded in 3D (z = 0, rotation about z) */
int dcora_log_trajectory(const char *path, int d, int n, const double *T);
/* fixedStiefelVariable (ref src/DCORA_utils.cpp:2053-2056): the lifting matrix YLift (r x d, orthonormal columns) the
 * agents share; identical on every call and in every process */
int dcora_fixed_stiefel_variable(int r, int d, double *Y);
/* Agent::initializeInGlobalFrame: X (r x k) = YLift (r x d) * T_world_robot applied to the local estimate (d x k,
 * SE ordering when dims.l = dims.b = 0, RA ordering otherwise) */
int dcora_agent_initialize_in_global_frame(const dcora_dims *dims, const double *T_world_robot,
                                           const double *T_local, const double *YLift, double *X);
int dcora_measurement_errors(dcora_dataset_t ds, int r, const double *X, double *out, int device);
/* solvePGO (ref src/DCORA_solver.cpp:304-328): T0 (d x (d+1) n) or NULL for the chordal start; one optimize() at
 * rank d with params */
int dcora_solve_pgo(dcora_dataset_t ds, const dcora_ropt_params *params, const double *T0, double *Tout,
                    dcora_ropt_result *result, int device);
/* solveRobustPGO (ref :330-409): GNC-TLS around solvePGO; fixed_weight: one flag per measurement (NULL = none
 * fixed); the final weights are written into the dataset handle and to weights_out (may be NULL) */
int dcora_solve_robust_pgo(dcora_dataset_t ds, const dcora_ropt_params *params, const dcora_robust_params *robust,
                           const int *fixed_weight, const double *T0, double *Tout, double *weights_out, int device);

/* ------------------------------------------------------------------------- *
 * Rounding / solution recovery
 * ------------------------------------------------------------------------- */
/* alignLiftedTrajectoryToFrame (ref src/DCORA_utils.cpp:2262-2289), Agent::getTrajectoryInGlobalFrame /
 * getStatesInLocalFrame (ref src/Agent.cpp:950-1034).  X is r x k in the layout of dims (SE ordering when
 * l = b = 0, RA ordering otherwise); anchor is the lifted pose [Y0 p0], r x (d+1), or NULL for pose 0 of X.
 * global_alignment != 0: the anchor's translation is the origin; 0: pose 0 of X is the origin.
 * trajectory: d x (d+1) n in the SE ordering, every rotation block projected to SO(d);
 * unit_spheres (d x l, rotated) and landmarks (d x b, rotated and translated) may be NULL. */
int dcora_round_align_trajectory(const dcora_dims *dims, const double *X, const double *anchor, int global_alignment,
                                 double *trajectory, double *unit_spheres, double *landmarks, int device);
/* projectSolutionRASLAM (ref src/DCORA_utils.cpp:1984-2031): rank-d truncation of X (r x k), reflection test,
 * SO(d) / unit-sphere projection; out is d x k in the same column layout.  Defined up to the sign convention of
 * the SVD, i.e. up to a global orthogonal transformation that the caller's refinement / alignment removes. */
int dcora_round_project_solution_raslam(const dcora_dims *dims, const double *X, double *out, int device);

/* ------------------------------------------------------------------------- *
 * Bench / profiling hooks
 * ------------------------------------------------------------------------- */
/* times `reps` launches of the Q-apply kernel Y = X Q + G of a problem with HIP events on the handle's
 * stream; returns average milliseconds per launch and the algorithmic bytes of one launch */
int dcora_problem_time_qapply(dcora_problem_t p, int reps, double *avg_ms, double *algorithmic_bytes);
/* the same kernel over `count` problems in turn on one stream: distinct (Q, X, Y) sets, so that with more than 256 MiB
 * between two uses of a set every launch streams from HBM instead of the Infinity Cache */
int dcora_problem_time_qapply_rotating(const dcora_problem_t *problems, int count, int reps, double *avg_ms);
/* which Q-apply kernel the problem runs: info[0] = 0 k_spmm (CSR), 1 k_spmm_bsr (block-CSR, pose graphs with
 * n >= 8192); info[1] = nnz(Q); info[2] = matrix blocks (block-CSR); info[3] = bytes of the stored matrix form */
int dcora_problem_qapply_info(dcora_problem_t p, double *info4);
/* same for the preconditioner application kernel z = Proj_X(r (Q + reg I)^-1) (dense-inverse streaming part) */
int dcora_problem_time_precond(dcora_problem_t p, int reps, double *avg_ms, double *algorithmic_bytes);
/* how the preconditioner (Q + reg I)^-1 of ref src/Graph.cpp:1901-1917 is held on the device:
 * info[0] = 0 none, 1 dense inverse, 2 partitioned sparse inverse (sparse_precond.h);
 * info[1] = kernel launches per application; info[2] = nnz of the Cholesky factor; info[3] = host setup ms;
 * info[4] = doubles of stored inverse one application streams */
int dcora_problem_precond_info(dcora_problem_t p, double *info5);
/* The preconditioner image (dense inverse or partitioned sparse inverse of Q + reg I) is cached inside the library,
 * keyed on the content of the matrix (rowptr, colidx, vals, reg, device): the reference re-creates its QuadraticProblem
 * on every Agent::updateX (ref src/Agent.cpp:1252) while its Graph keeps Q and the factor (ref src/Graph.cpp:523-533,
 * 1901-1917).  A second dcora_problem_create / dcora_rbcd_create on the same matrix -- the next staircase level, a
 * problem re-created per update -- attaches to the resident image instead of factorising again.
 * info[4] = {hits, misses, entries, device bytes held}.  Budget: DCORA_PRECOND_CACHE_MB (default 8192, 0 = off). */
/* addition of this library: the analysis of a certificate's PSD test ahead of time.  The dual certificate S = Q - Lambda
 * has the sparsity pattern of Q, known before the agents start: a driver calls this on another host thread while they
 * iterate; the symbolic analysis (ordering, fronts), its device image and the arena are left in the library's cache and
 * dcora_cert_fast_verification / dcora_cert_is_psd_device of a matrix with this pattern begin with the numeric phase.
 * dims as for dcora_cert_dual_matrix (r is not used); rp / ci: the CSR pattern of Q, both triangles; block as in the PSD
 * test.  The pattern analysed is the one dcora_cert_dual_matrix produces: Q's, the blocks of Lambda, every diagonal. */
int dcora_cert_prepare(const dcora_dims *dims, const int *rp, const int *ci, int block, int device);
int dcora_precond_cache_info(double *info4);
int dcora_precond_cache_clear(void);

#ifdef __cplusplus
}
#endif
#endif /* DCORA_HIP_H_ */
