// QuadraticProblem with the reference's Matrix-level interface (ref include/DCORA/QuadraticProblem.h:101-174),
// backed by the MI355X implementation through include/dcora_hip.h.  The reference constructs it from a
// shared_ptr<Graph>; the only things it reads from the Graph are (r, d, n, l, b), quadraticMatrix(),
// linearMatrix() and the preconditioner regularisation (ref src/QuadraticProblem.cpp:19-34, 42, 58, 78-79;
// src/Graph.cpp:1901-1917), which is exactly the ProblemData below -- see INTEGRATION.md for the two-line adapter.
#pragma once
#include "DCORA_types.h"
#include "Graph.h"

namespace DCORA {

struct ProblemData {
  unsigned r = 0, d = 0, n = 0, l = 0, b = 0;
  SparseMatrix Q;        // Graph::quadraticMatrix()
  Matrix G;              // Graph::linearMatrix() (may be empty = zero)
  double precond_reg = 0.1;  // 1e-1 for PGO (ref src/Graph.cpp:1906); < 0 = no preconditioner
  int device = 0;
  int layout = DCORA_LAYOUT_AUTO;  // DCORA_LAYOUT_RA: the RA ordering also when l = b = 0 (a range-aided graph type)
};

class QuadraticProblem {
 public:
  explicit QuadraticProblem(const ProblemData &pd)
      : r_(pd.r), d_(pd.d), n_(pd.n), l_(pd.l), b_(pd.b),
        se_(pd.layout == DCORA_LAYOUT_RA ? false : (pd.l == 0 && pd.b == 0)) {
    dcora_dims dims{(int)pd.r, (int)pd.d, (int)pd.n, (int)pd.l, (int)pd.b, pd.layout};
    check_status(dcora_problem_create(&dims, pd.Q.rowptr.data(), pd.Q.colidx.data(), pd.Q.vals.data(),
                                      pd.G.rows() ? pd.G.data() : nullptr, pd.precond_reg, pd.device, &h_),
                 "QuadraticProblem");
  }
  // the reference's constructor (ref include/DCORA/QuadraticProblem.h:28, src/QuadraticProblem.cpp:19-34): (r, d, n, l, b),
  // Q and the linear term from the Graph -- the SE manifold when the graph is PGO-compatible, the RA manifold otherwise;
  // the preconditioner (Q + reg I)^-1 with the Graph's regularisation (ref src/Graph.cpp:1901-1960) is built when
  // withPreconditioner is set (the local solver and escapeSaddle need it)
  explicit QuadraticProblem(const std::shared_ptr<Graph> &graph, bool withPreconditioner = false, int device = 0)
      : r_(graph->r()), d_(graph->d()), n_(graph->n()), l_(graph->l()), b_(graph->b()),
        se_(graph->isPGOCompatible()), graph_(graph) {
    const SparseMatrix &Q = graph->quadraticMatrix();
    const Matrix G = graph->linearMatrix();
    bool has_G = false;
    for (size_t i = 0; i < G.rows() * G.cols() && !has_G; ++i) has_G = G.data()[i] != 0.0;
    dcora_dims dims{(int)r_, (int)d_, (int)n_, (int)l_, (int)b_, graph->layout()};
    check_status(dcora_problem_create(&dims, Q.rowptr.data(), Q.colidx.data(), Q.vals.data(), has_G ? G.data() : nullptr,
                                      withPreconditioner ? graph->preconditionerRegularization(device) : -1.0, device,
                                      &h_),
                 "QuadraticProblem");
  }
  ~QuadraticProblem() { dcora_problem_destroy(h_); }
  QuadraticProblem(const QuadraticProblem &) = delete;
  QuadraticProblem &operator=(const QuadraticProblem &) = delete;

  unsigned int dimension() const { return d_; }
  unsigned int relaxation_rank() const { return r_; }
  unsigned int num_poses() const { return n_; }
  unsigned int num_unit_spheres() const { return l_; }
  unsigned int num_landmarks() const { return b_; }
  unsigned int problem_dimension() const { return (d_ + 1) * n_ + l_ + b_; }
  bool useSEManifold() const { return se_; }  // by graph type, ref src/QuadraticProblem.cpp:19-34

  void setLinearTerm(const Matrix &G) { check_status(dcora_problem_set_linear_term(h_, G.data()), "setLinearTerm"); }
  double f(const Matrix &Y) const {
    double out = 0;
    check_status(dcora_problem_cost(h_, Y.data(), &out), "f");
    return out;
  }
  Matrix RieGrad(const Matrix &Y) const {
    Matrix out(Y.rows(), Y.cols());
    check_status(dcora_problem_riegrad(h_, Y.data(), out.data(), nullptr), "RieGrad");
    return out;
  }
  double RieGradNorm(const Matrix &Y) const {
    double nrm = 0;
    check_status(dcora_problem_riegrad(h_, Y.data(), nullptr, &nrm), "RieGradNorm");
    return nrm;
  }
  Matrix Retract(const Matrix &Y, const Matrix &V) const {
    Matrix out(Y.rows(), Y.cols());
    check_status(dcora_problem_retract(h_, Y.data(), V.data(), out.data()), "Retract");
    return out;
  }
  Matrix PreCondition(const Matrix &Y, const Matrix &V) const {
    Matrix out(Y.rows(), Y.cols());
    check_status(dcora_problem_precondition(h_, Y.data(), V.data(), out.data()), "PreCondition");
    return out;
  }
  // ref src/QuadraticProblem.cpp:138-234; *X is written only on success
  bool escapeSaddle(const Matrix &Xopt, double theta, const Vector &v, double gradient_tolerance,
                    double preconditioned_gradient_tolerance, Matrix *X, bool isSecondOrder = false) {
    Matrix out(r_, problem_dimension());
    int ok = 0;
    check_status(dcora_problem_escape_saddle(h_, Xopt.data(), theta, v.data(), gradient_tolerance,
                                             preconditioned_gradient_tolerance, isSecondOrder ? 1 : 0, out.data(), &ok),
                 "escapeSaddle");
    if (ok) *X = out;
    return ok != 0;
  }
  dcora_problem_t handle() const { return h_; }

 private:
  unsigned r_, d_, n_, l_, b_;
  bool se_;
  std::shared_ptr<Graph> graph_;
  dcora_problem_t h_ = nullptr;
};

}  // namespace DCORA
