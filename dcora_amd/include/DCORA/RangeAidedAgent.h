// One robot of a (multi-robot) range-aided SLAM problem with the reference's Agent semantics (ref src/Agent.cpp:
// 113-152 getSharedStateDicts, 535-596 iterate, 844-906 updateNeighborStates, 1158-1278 acceleration + updateX;
// src/Graph.cpp:195-300 shared measurements, 824-1772 data matrices).  The agent is self-contained, as the reference's
// is: it owns the measurements that touch one of its states, its iterate in ITS RA ordering
// [rotations | unit spheres | translations | landmarks], its acceleration sequences and the caches of the neighbours'
// public states it was handed; robots talk through the three dictionaries only (no shared session, no registry).
//
// Data matrices.  The reference assembles Q_a and G_a measurement by measurement; here the agent's measurements --
// its own states plus the FOREIGN states they reach (neighbours' poses / landmarks, and the unit spheres of ranges
// whose source is a neighbour) -- are numbered as one small merged problem, the library builds that problem's Q
// (dcora_radataset_create + dcora_radataset_build_Q) and cuts it into Q_aa = Q[own, own] and C = Q[own, foreign]
// (dcora_graph_extract_agent_blocks), so that G_a = X_foreign C^T: the same restriction, from the same closed forms.
// The local solve runs on the device (QuadraticProblem / QuadraticOptimizer over dcora_problem_*).
#pragma once
#include <cmath>
#include <map>
#include <memory>
#include <set>
#include <vector>

#include "QuadraticOptimizer.h"
#include "RangeAided.h"

namespace DCORA {
namespace detail {

class RangeAidedAgentCore {
 public:
  RangeAidedAgentCore(unsigned id, unsigned d, unsigned r, unsigned numRobots, bool acceleration,
                      unsigned restartInterval, const ROptParameters &prm, int device)
      : id_(id), d_(d), r_(r), R_(numRobots), accel_(acceleration), restart_(restartInterval), prm_(prm), dev_(device) {}

  unsigned n() const { return n_; }
  unsigned l() const { return l_; }
  unsigned b() const { return b_; }
  unsigned k() const { return (d_ + 1) * n_ + l_ + b_; }
  unsigned iterations() const { return iterations_; }
  bool initialized() const { return X_.rows() != 0; }
  const Matrix &X() const { return X_; }
  const SparseMatrix &quadraticMatrix() const { return Qaa_; }
  double regularization() const { return reg_; }

  // ref src/Graph.cpp:374-470 (setMeasurements) + :195-300 (shared measurements): states, public states, data matrices
  void setMeasurements(const RelativeMeasurements &meas) {
    const auto pp = meas.GetRelativePosePoseMeasurements();
    const auto pl = meas.GetRelativePoseLandmarkMeasurements();
    const auto rg = meas.GetRangeMeasurements();
    n_ = l_ = b_ = 0;
    fpose_.clear(), flm_.clear(), fsph_.clear(), pub_pose_.clear(), pub_lm_.clear(), pub_sph_.clear();
    auto seen = [&](StateType t, size_t robot, size_t p) {
      if (robot != id_) return;
      if (t == StateType::Landmark) b_ = std::max<unsigned>(b_, (unsigned)p + 1); else n_ = std::max<unsigned>(n_, (unsigned)p + 1);
    };
    for (const auto &m : pp) {
      check_mine(m.r1, m.r2);
      seen(StateType::Pose, m.r1, m.p1), seen(StateType::Pose, m.r2, m.p2);
    }
    for (const auto &m : pl) {
      check_mine(m.r1, m.r2);
      seen(StateType::Pose, m.r1, m.p1), seen(StateType::Landmark, m.r2, m.p2);
    }
    for (const auto &m : rg) {
      check_mine(m.r1, m.r2);
      seen(m.stateType1, m.r1, m.p1), seen(m.stateType2, m.r2, m.p2);
      if (m.r1 == id_) l_ = std::max<unsigned>(l_, (unsigned)m.l + 1);  // the unit sphere belongs to the source's robot
    }
    // foreign states in order of appearance; own states that a shared measurement reaches are public
    auto pose_idx = [&](size_t robot, size_t p) -> int {
      if (robot == id_) return (int)p;
      auto it = fpose_.emplace(PoseID((unsigned)robot, (unsigned)p), (unsigned)fpose_.size()).first;
      return (int)(n_ + it->second);
    };
    auto lm_idx = [&](size_t robot, size_t p) -> int {
      if (robot == id_) return (int)p;
      auto it = flm_.emplace(LandmarkID((unsigned)robot, (unsigned)p), (unsigned)flm_.size()).first;
      return (int)(b_ + it->second);
    };
    auto sph_idx = [&](size_t robot, size_t s) -> int {
      if (robot == id_) return (int)s;
      auto it = fsph_.emplace(UnitSphereID((unsigned)robot, (unsigned)s), (unsigned)fsph_.size()).first;
      return (int)(l_ + it->second);
    };
    auto mark_public = [&](StateType t, size_t robot, size_t p) {
      if (robot != id_) return;
      if (t == StateType::Landmark) pub_lm_.insert((unsigned)p); else pub_pose_.insert((unsigned)p);
    };
    const size_t w = (size_t)d_ * d_ + d_ + 3;
    std::vector<int> ppi(2 * pp.size() + 2), pli(2 * pl.size() + 2), rgi(5 * rg.size() + 5);
    std::vector<double> ppv(w * pp.size() + 1), plv((size_t)(d_ + 2) * pl.size() + 1), rgv(3 * rg.size() + 1);
    for (size_t i = 0; i < pp.size(); ++i) {
      const auto &m = pp[i];
      ppi[2 * i] = pose_idx(m.r1, m.p1), ppi[2 * i + 1] = pose_idx(m.r2, m.p2);
      if (m.r1 != m.r2) mark_public(StateType::Pose, m.r1, m.p1), mark_public(StateType::Pose, m.r2, m.p2);
      double *v = &ppv[i * w];
      for (unsigned c = 0; c < d_; ++c)
        for (unsigned a = 0; a < d_; ++a) v[c * d_ + a] = m.R(a, c);
      for (unsigned a = 0; a < d_; ++a) v[d_ * d_ + a] = m.t[a];
      v[d_ * d_ + d_] = m.kappa, v[d_ * d_ + d_ + 1] = m.tau, v[d_ * d_ + d_ + 2] = m.weight;
    }
    for (size_t i = 0; i < pl.size(); ++i) {
      const auto &m = pl[i];
      pli[2 * i] = pose_idx(m.r1, m.p1), pli[2 * i + 1] = lm_idx(m.r2, m.p2);
      if (m.r1 != m.r2) mark_public(StateType::Pose, m.r1, m.p1), mark_public(StateType::Landmark, m.r2, m.p2);
      double *v = &plv[i * (d_ + 2)];
      for (unsigned a = 0; a < d_; ++a) v[a] = m.t[a];
      v[d_] = m.tau, v[d_ + 1] = m.weight;
    }
    for (size_t i = 0; i < rg.size(); ++i) {
      const auto &m = rg[i];
      const bool lm1 = m.stateType1 == StateType::Landmark, lm2 = m.stateType2 == StateType::Landmark;
      rgi[5 * i] = lm1 ? 1 : 0, rgi[5 * i + 1] = lm1 ? lm_idx(m.r1, m.p1) : pose_idx(m.r1, m.p1);
      rgi[5 * i + 2] = lm2 ? 1 : 0, rgi[5 * i + 3] = lm2 ? lm_idx(m.r2, m.p2) : pose_idx(m.r2, m.p2);
      rgi[5 * i + 4] = sph_idx(m.r1, m.l);
      if (m.r1 != m.r2) {
        mark_public(m.stateType1, m.r1, m.p1), mark_public(m.stateType2, m.r2, m.p2);
        if (m.r1 == id_) pub_sph_.insert((unsigned)m.l);  // ref src/Graph.cpp:233-238: the source shares its unit sphere
      }
      rgv[3 * i] = m.range, rgv[3 * i + 1] = m.precision, rgv[3 * i + 2] = m.weight;
    }
    nt_ = n_ + (unsigned)fpose_.size(), lt_ = l_ + (unsigned)fsph_.size(), bt_ = b_ + (unsigned)flm_.size();
    dcora_radataset_t ds = nullptr;
    check_status(dcora_radataset_create((int)d_, (int)nt_, (int)lt_, (int)bt_, (int)pp.size(), ppi.data(), ppv.data(),
                                        (int)pl.size(), pli.data(), plv.data(), (int)rg.size(), rgi.data(), rgv.data(),
                                        nullptr, &ds),
                 "Agent::setMeasurements");
    dcora_csr_t hq = nullptr;
    int rc = dcora_radataset_build_Q(ds, &hq);
    dcora_radataset_destroy(ds);
    check_status(rc, "Agent::setMeasurements");
    const SparseMatrix Qm = take(hq);
    std::vector<int> own;
    for (unsigned i = 0; i < n_; ++i)
      for (unsigned c = 0; c < d_; ++c) own.push_back((int)(i * d_ + c));
    for (unsigned s = 0; s < l_; ++s) own.push_back((int)(d_ * nt_ + s));
    for (unsigned i = 0; i < n_; ++i) own.push_back((int)(d_ * nt_ + lt_ + i));
    for (unsigned j = 0; j < b_; ++j) own.push_back((int)(d_ * nt_ + lt_ + nt_ + j));
    dcora_csr_t hqa = nullptr, hc = nullptr;
    check_status(dcora_graph_extract_agent_blocks(Qm.n, Qm.rowptr.data(), Qm.colidx.data(), Qm.vals.data(), (int)own.size(),
                                                  own.data(), &hqa, &hc),
                 "Agent::setMeasurements");
    Qaa_ = take(hqa);
    C_ = take(hc);
    reg_ = 0.1;
    check_status(dcora_graph_precond_regularization(Qaa_.n, Qaa_.rowptr.data(), Qaa_.colidx.data(), Qaa_.vals.data(), dev_,
                                                    &reg_),
                 "Agent::setMeasurements");
    problem_.reset();
  }

  // ref src/Agent.cpp:64-77: the iterate, and a fresh acceleration (ref :1178-1187)
  void setX(const Matrix &Xin) {
    if (Xin.rows() != r_ || Xin.cols() != k()) throw std::invalid_argument("Agent::setX: expected r x k of this agent");
    X_ = Xin;
    V_ = Y_ = XPrev_ = Xin;
    gamma_ = alpha_ = 0;
  }

  // ref src/Agent.cpp:535-596
  bool iterate(bool doOptimization) {
    if (!initialized()) return false;
    ++iterations_;
    XPrev_ = X_;
    bool success;
    if (accel_) {
      gamma_ = (1 + std::sqrt(1 + 4.0 * R_ * R_ * gamma_ * gamma_)) / (2.0 * R_);  // updateGamma
      alpha_ = 1 / (gamma_ * R_);                                                     // updateAlpha
      Y_ = project(combine(1 - alpha_, X_, alpha_, V_));                               // updateY
      success = updateX(doOptimization, true);
      V_ = project(combine3(V_, gamma_, X_, Y_));                                      // updateV
      if ((iterations_ + 1) % restart_ == 0) {                                         // restartNesterovAcceleration
        X_ = XPrev_;
        updateX(doOptimization, false);
        V_ = X_, Y_ = X_;
        gamma_ = alpha_ = 0;
      }
    } else {
      success = updateX(doOptimization, false);
    }
    return success;
  }

  // ref src/Agent.cpp:113-152: my states that a shared measurement reaches
  void sharedStates(PoseDict *poses, UnitSphereDict *spheres, LandmarkDict *landmarks) const {
    poses->clear();
    if (spheres) spheres->clear();
    if (landmarks) landmarks->clear();
    for (unsigned i : pub_pose_) (*poses)[PoseID(id_, i)] = pose_of(X_, i);
    if (spheres)
      for (unsigned s : pub_sph_) (*spheres)[UnitSphereID(id_, s)] = col_of(X_, (size_t)d_ * n_ + s);
    if (landmarks)
      for (unsigned j : pub_lm_) (*landmarks)[LandmarkID(id_, j)] = col_of(X_, (size_t)(d_ + 1) * n_ + l_ + j);
  }

  // ref src/Agent.cpp:844-906: states I require go into my cache (plain or auxiliary), others are ignored
  void updateNeighborStates(unsigned neighborID, const PoseDict &poses, bool aux, const UnitSphereDict &spheres,
                            const LandmarkDict &landmarks) {
    if (neighborID == id_) throw std::invalid_argument("updateNeighborStates: neighborID is this agent");
    for (const auto &kv : poses) {
      check_state(kv.first, neighborID, kv.second, d_ + 1);
      if (fpose_.count(kv.first)) (aux ? auxPose_ : nbrPose_)[kv.first] = kv.second;
    }
    for (const auto &kv : spheres) {
      check_state(kv.first, neighborID, kv.second, 1);
      if (fsph_.count(kv.first)) (aux ? auxSph_ : nbrSph_)[kv.first] = kv.second;
    }
    for (const auto &kv : landmarks) {
      check_state(kv.first, neighborID, kv.second, 1);
      if (flm_.count(kv.first)) (aux ? auxLm_ : nbrLm_)[kv.first] = kv.second;
    }
  }
  void clearNeighborStates() { nbrPose_.clear(), auxPose_.clear(), nbrSph_.clear(), auxSph_.clear(), nbrLm_.clear(), auxLm_.clear(); }

  // Graph::linearMatrix() against one of the two caches; false when a required state has never been handed over
  bool linearTerm(bool aux, Matrix *G) const {
    const unsigned kt = (d_ + 1) * nt_ + lt_ + bt_;
    Matrix Xm(r_, kt);
    const PoseDict &P = aux ? auxPose_ : nbrPose_;
    const UnitSphereDict &S = aux ? auxSph_ : nbrSph_;
    const LandmarkDict &L = aux ? auxLm_ : nbrLm_;
    for (const auto &kv : fpose_) {
      auto it = P.find(kv.first);
      if (it == P.end()) return false;
      const unsigned i = n_ + kv.second;
      for (unsigned a = 0; a < r_; ++a) {
        for (unsigned c = 0; c < d_; ++c) Xm(a, (size_t)i * d_ + c) = it->second(a, c);
        Xm(a, (size_t)d_ * nt_ + lt_ + i) = it->second(a, d_);
      }
    }
    for (const auto &kv : fsph_) {
      auto it = S.find(kv.first);
      if (it == S.end()) return false;
      for (unsigned a = 0; a < r_; ++a) Xm(a, (size_t)d_ * nt_ + l_ + kv.second) = it->second(a, 0);
    }
    for (const auto &kv : flm_) {
      auto it = L.find(kv.first);
      if (it == L.end()) return false;
      for (unsigned a = 0; a < r_; ++a) Xm(a, (size_t)d_ * nt_ + lt_ + nt_ + b_ + kv.second) = it->second(a, 0);
    }
    *G = Matrix(r_, k());
    for (int i = 0; i < C_.n; ++i)
      for (int p = C_.rowptr[(size_t)i]; p < C_.rowptr[(size_t)i + 1]; ++p) {
        const double v = C_.vals[(size_t)p];
        const size_t j = (size_t)C_.colidx[(size_t)p];
        for (unsigned a = 0; a < r_; ++a) (*G)(a, (size_t)i) += v * Xm(a, j);
      }
    return true;
  }
  bool hasNeighbors() const { return !fpose_.empty() || !fsph_.empty() || !flm_.empty(); }
  std::shared_ptr<QuadraticProblem> problem() {
    if (!problem_) {
      ProblemData pd;
      pd.r = r_, pd.d = d_, pd.n = n_, pd.l = l_, pd.b = b_;
      pd.Q = Qaa_;
      pd.precond_reg = reg_;
      pd.device = dev_;
      pd.layout = DCORA_LAYOUT_RA;
      problem_ = std::make_shared<QuadraticProblem>(pd);
    }
    return problem_;
  }

 private:
  static SparseMatrix take(dcora_csr_t h) {
    SparseMatrix M;
    int n = 0, nnz = 0;
    dcora_csr_info(h, &n, &nnz);
    M.n = n;
    M.rowptr.resize((size_t)n + 1), M.colidx.resize((size_t)nnz), M.vals.resize((size_t)nnz);
    dcora_csr_copy(h, M.rowptr.data(), M.colidx.data(), M.vals.data());
    dcora_csr_destroy(h);
    return M;
  }
  void check_mine(size_t r1, size_t r2) const {
    if (r1 != id_ && r2 != id_) throw std::invalid_argument("Agent::setMeasurements: a measurement between two other robots");
  }
  template <class ID>
  void check_state(const ID &sid, unsigned neighborID, const Matrix &v, unsigned cols) const {
    if (sid.robot_id != neighborID) throw std::invalid_argument("updateNeighborStates: state of another robot");
    if (v.rows() != r_ || v.cols() != cols) throw std::invalid_argument("updateNeighborStates: shape of a lifted state");
  }
  Matrix pose_of(const Matrix &X, unsigned i) const {
    Matrix P(r_, d_ + 1);
    for (unsigned a = 0; a < r_; ++a) {
      for (unsigned c = 0; c < d_; ++c) P(a, c) = X(a, (size_t)i * d_ + c);
      P(a, d_) = X(a, (size_t)d_ * n_ + l_ + i);
    }
    return P;
  }
  Matrix col_of(const Matrix &X, size_t j) const {
    Matrix v(r_, 1);
    for (unsigned a = 0; a < r_; ++a) v(a, 0) = X(a, j);
    return v;
  }
  static Matrix combine(double a, const Matrix &A, double b, const Matrix &B) {
    Matrix M(A.rows(), A.cols());
    for (size_t i = 0; i < A.rows() * A.cols(); ++i) M.data()[i] = a * A.data()[i] + b * B.data()[i];
    return M;
  }
  static Matrix combine3(const Matrix &V, double g, const Matrix &X, const Matrix &Y) {  // V + g (X - Y)
    Matrix M(V.rows(), V.cols());
    for (size_t i = 0; i < V.rows() * V.cols(); ++i) M.data()[i] = V.data()[i] + g * (X.data()[i] - Y.data()[i]);
    return M;
  }
  Matrix project(const Matrix &M) const {  // projectToRAMatrix, ref src/DCORA_utils.cpp:2211-2220
    dcora_dims dims{(int)r_, (int)d_, (int)n_, (int)l_, (int)b_, DCORA_LAYOUT_RA};
    Matrix out(M.rows(), M.cols());
    check_status(dcora_manifold_project(&dims, M.data(), out.data(), dev_), "Agent: projectToManifold");
    return out;
  }
  // ref src/Agent.cpp:1216-1278
  bool updateX(bool doOptimization, bool acceleration) {
    if (!doOptimization) {
      if (acceleration) X_ = Y_;
      return true;
    }
    std::shared_ptr<QuadraticProblem> P = problem();
    if (hasNeighbors()) {
      Matrix G;
      if (!linearTerm(acceleration, &G)) return false;  // constructDataMatrices failed: skip the optimisation
      P->setLinearTerm(G);
    }
    QuadraticOptimizer opt(P.get(), prm_);
    X_ = opt.optimize(acceleration ? Y_ : X_);
    return true;
  }

  unsigned id_, d_, r_, R_;
  bool accel_;
  unsigned restart_;
  ROptParameters prm_;
  int dev_;
  unsigned n_ = 0, l_ = 0, b_ = 0, nt_ = 0, lt_ = 0, bt_ = 0, iterations_ = 0;
  std::map<PoseID, unsigned> fpose_;
  std::map<LandmarkID, unsigned> flm_;
  std::map<UnitSphereID, unsigned> fsph_;
  std::set<unsigned> pub_pose_, pub_lm_, pub_sph_;
  SparseMatrix Qaa_, C_;
  double reg_ = 0.1;
  std::shared_ptr<QuadraticProblem> problem_;
  Matrix X_, V_, Y_, XPrev_;
  double gamma_ = 0, alpha_ = 0;
  PoseDict nbrPose_, auxPose_;
  UnitSphereDict nbrSph_, auxSph_;
  LandmarkDict nbrLm_, auxLm_;
};

}  // namespace detail
}  // namespace DCORA
