// QuadraticOptimizer with the reference's interface (ref include/DCORA/QuadraticOptimizer.h:33-102): holds a
// non-owning QuadraticProblem*, optimize(Y) returns the new iterate by value, getOptResult() the statistics.
#pragma once
#include "QuadraticProblem.h"

namespace DCORA {

class QuadraticOptimizer {
 public:
  QuadraticOptimizer(QuadraticProblem *p, ROptParameters params = ROptParameters()) : problem_(p), params_(params) {}
  // ref src/QuadraticOptimizer.cpp:28-50; returns its input when the gradient is already below tolerance or every
  // trust-region step was rejected (ref :54-55, :264-266)
  Matrix optimize(const Matrix &Y) {
    Matrix out(Y.rows(), Y.cols());
    dcora_ropt_params p = params_.c();
    dcora_ropt_result r;
    check_status(dcora_optimizer_optimize(problem_->handle(), &p, Y.data(), out.data(), &r), "optimize");
    result_.success = r.success != 0;
    result_.fInit = r.fInit;
    result_.gradNormInit = r.gradNormInit;
    result_.fOpt = r.fOpt;
    result_.gradNormOpt = r.gradNormOpt;
    result_.elapsedMs = r.elapsedMs;
    result_.tCGStatus = r.tCGStatus;
    return out;
  }
  // ref include/DCORA/QuadraticOptimizer.h:52: the optimizer outlives the problems it is pointed at (one per update)
  void setProblem(QuadraticProblem *p) { problem_ = p; }
  void setVerbose(bool v) { params_.verbose = v; }
  void setAlgorithm(ROptParameters::ROptMethod alg) { params_.method = alg; }
  void setRGDStepsize(double s) { params_.RGD_stepsize = s; }
  void setRTRIterations(int iter) { params_.RTR_iterations = iter; }
  void setGradientNormTolerance(double tol) { params_.gradnorm_tol = tol; }
  void setRTRInitialRadius(double radius) { params_.RTR_initial_radius = radius; }
  void setRTRtCGIterations(int iter) { params_.RTR_tCG_iterations = iter; }
  ROPTResult getOptResult() const { return result_; }

 private:
  QuadraticProblem *problem_;
  ROptParameters params_;
  ROPTResult result_;
};

}  // namespace DCORA
