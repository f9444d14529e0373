// The reference's range-aided agent test through the facade (ref tests/testAgent.cpp:157-242,
// testAgentInitializeIterateOptimizeRA): read the noiseless range-aided fixtures, build the centralised Agent on a
// RangeAidedSLAMGraph from RelativeMeasurements, initialise it at the ground truth, and check that the states in the
// local frame equal the aligned ground truth before and after iterate() -- the ground truth of a noiseless problem is a
// fixed point of the local solver.  Also the Graph / QuadraticProblem(shared_ptr<Graph>) path of
// ref src/QuadraticProblem.cpp:19-34 on the same graph: l(), b(), linearMatrix(), cost and gradient at the optimum.
#include <cmath>
#include <cstdio>
#include <string>

#include "DCORA/Agent.h"
#include "DCORA/QuadraticOptimizer.h"

static int failures = 0;
#define EXPECT(cond)                                                                  \
  do {                                                                                \
    if (!(cond)) {                                                                    \
      std::fprintf(stderr, "%s:%d: EXPECT failed: %s\n", __FILE__, __LINE__, #cond);  \
      ++failures;                                                                     \
    }                                                                                 \
  } while (0)

// Eigen's isApprox: |a - b| <= tol min(|a|, |b|) in the Frobenius norm
static bool isApprox(const DCORA::Matrix &a, const DCORA::Matrix &b, double tol) {
  if (a.rows() != b.rows() || a.cols() != b.cols()) return false;
  double diff = 0;
  for (size_t j = 0; j < a.cols(); ++j)
    for (size_t i = 0; i < a.rows(); ++i) diff += (a(i, j) - b(i, j)) * (a(i, j) - b(i, j));
  return std::sqrt(diff) <= tol * std::min(a.norm(), b.norm());
}

int main(int argc, char **argv) {
  const double OPTIMIZATION_TOL = 1e-9;  // ref tests/testAgent.cpp:20
  for (int f = 1; f < argc; ++f) {
    const DCORA::PyFGDataset dataset = DCORA::read_pyfg_file(argv[f]);
    const DCORA::Measurements global_measurements = DCORA::getGlobalMeasurements(dataset);
    const unsigned id = DCORA::CENTRALIZED_AGENT_ID;
    const unsigned d = global_measurements.ground_truth_init->d();
    const unsigned r = d;
    const unsigned n = global_measurements.ground_truth_init->n();
    const unsigned l = global_measurements.ground_truth_init->l();
    const unsigned b = global_measurements.ground_truth_init->b();
    const DCORA::PoseArray TrajectoryGroundTruth = global_measurements.ground_truth_init->getPoseArray();
    const DCORA::PointArray UnitShereGroundTruth = global_measurements.ground_truth_init->getUnitSphereArray();
    const DCORA::PointArray LandmarkGroundTruth = global_measurements.ground_truth_init->getLandmarkArray();

    // Construct and initialize
    DCORA::AgentParameters options(d, r, {id}, DCORA::GraphType::RangeAidedSLAMGraph);
    DCORA::Agent agent(id, options);
    agent.setMeasurements(global_measurements.relative_measurements);
    agent.initialize(&TrajectoryGroundTruth, &UnitShereGroundTruth, &LandmarkGroundTruth);
    EXPECT(agent.getID() == id);
    EXPECT(agent.relaxation_rank() == r);
    EXPECT(agent.dimension() == d);
    EXPECT(agent.num_poses() == n);
    EXPECT(agent.num_unit_spheres() == l);
    EXPECT(agent.num_landmarks() == b);

    // Get aligned ground truth
    const DCORA::Pose Tw0(TrajectoryGroundTruth.pose(0));
    const DCORA::PoseArray TrajectoryGroundTruthAligned = DCORA::alignTrajectoryToFrame(TrajectoryGroundTruth, Tw0);
    const DCORA::PointArray UnitSpheresGroundTruthAligned = DCORA::alignUnitSpheresToFrame(UnitShereGroundTruth, Tw0);
    const DCORA::PointArray LandmarksGroundTruthAligned = DCORA::alignLandmarksToFrame(LandmarkGroundTruth, Tw0);

    // Check default state initialization
    DCORA::Matrix TrajectoryEstimated, UnitSphereEstimated, LandmarksEstimated;
    EXPECT(agent.getStatesInLocalFrame(&TrajectoryEstimated, &UnitSphereEstimated, &LandmarksEstimated));
    EXPECT(isApprox(TrajectoryGroundTruthAligned.getData(), TrajectoryEstimated, OPTIMIZATION_TOL));
    EXPECT(isApprox(UnitSpheresGroundTruthAligned.getData(), UnitSphereEstimated, OPTIMIZATION_TOL));
    EXPECT(isApprox(LandmarksGroundTruthAligned.getData(), LandmarksEstimated, OPTIMIZATION_TOL));

    // Check the states after one iteration, and after several (the reference's optimisation thread)
    for (int round = 0; round < 4; ++round) {
      EXPECT(agent.iterate());
      agent.getStatesInLocalFrame(&TrajectoryEstimated, &UnitSphereEstimated, &LandmarksEstimated);
      EXPECT(isApprox(TrajectoryGroundTruthAligned.getData(), TrajectoryEstimated, OPTIMIZATION_TOL));
      EXPECT(isApprox(UnitSpheresGroundTruthAligned.getData(), UnitSphereEstimated, OPTIMIZATION_TOL));
      EXPECT(isApprox(LandmarksGroundTruthAligned.getData(), LandmarksEstimated, OPTIMIZATION_TOL));
    }
    EXPECT(agent.iteration_number() == 4);

    // Graph / QuadraticProblem(shared_ptr<Graph>) on the same measurements (ref src/QuadraticProblem.cpp:19-34)
    auto graph = std::make_shared<DCORA::Graph>(id, r, d, DCORA::GraphType::RangeAidedSLAMGraph);
    graph->setMeasurements(global_measurements.relative_measurements);
    EXPECT(graph->n() == n && graph->l() == l && graph->b() == b && graph->k() == (d + 1) * n + l + b);
    EXPECT(!graph->isPGOCompatible());
    const DCORA::Matrix G = graph->linearMatrix();
    EXPECT(G.rows() == r && G.cols() == graph->k() && G.norm() == 0.0);
    EXPECT(graph->preconditionerRegularization() > 0.0);
    DCORA::QuadraticProblem problem(graph, true);
    EXPECT(!problem.useSEManifold());
    EXPECT(problem.num_unit_spheres() == l && problem.num_landmarks() == b && problem.problem_dimension() == graph->k());
    DCORA::Matrix Xgt;
    EXPECT(agent.getX(&Xgt));  // still the lifted ground truth: a fixed point
    EXPECT(std::fabs(problem.f(Xgt)) < 1e-9);
    EXPECT(problem.RieGradNorm(Xgt) < 1e-6);
    DCORA::QuadraticOptimizer optimizer(&problem);
    const DCORA::Matrix Xopt = optimizer.optimize(Xgt);
    EXPECT(isApprox(Xgt, Xopt, 1e-9));

    // a pose graph refuses range measurements; reset() empties the agent
    bool threw = false;
    try {
      DCORA::Graph pg(id, r, d);
      pg.setMeasurements(global_measurements.relative_measurements);
    } catch (const std::invalid_argument &) {
      threw = true;
    }
    EXPECT(threw);
    agent.reset();
    EXPECT(agent.num_poses() == 0 && !agent.getX(&Xgt));

    // A RangeAidedSLAMGraph that holds ONLY the pose-pose measurements of the file: the reference decides the manifold
    // by graph type (ref src/Graph.cpp:68-75), so the columns stay in the RA ordering [R_1 .. R_n | t_1 .. t_n] although
    // l = b = 0 -- against the pose graph on the same measurements (SE ordering)
    {
      DCORA::RelativeMeasurements posesOnly;
      for (const auto &m : global_measurements.relative_measurements.GetRelativePosePoseMeasurements())
        posesOnly.push_back(m);
      auto graphRA = std::make_shared<DCORA::Graph>(id, r + 1, d, DCORA::GraphType::RangeAidedSLAMGraph);
      graphRA->setMeasurements(posesOnly);
      auto graphSE = std::make_shared<DCORA::Graph>(id, r + 1, d);
      graphSE->setMeasurements(posesOnly.GetRelativePosePoseMeasurements());
      EXPECT(graphRA->l() == 0 && graphRA->b() == 0 && graphRA->n() == n && !graphRA->isPGOCompatible());
      EXPECT(graphSE->isPGOCompatible() && graphSE->n() == n);
      DCORA::QuadraticProblem pRA(graphRA, true), pSE(graphSE, true);
      EXPECT(!pRA.useSEManifold() && pSE.useSEManifold());
      // a point of the manifold in both orderings: the lifted ground truth, rotated a little pose by pose
      const unsigned rr = r + 1;
      DCORA::Matrix Xse(rr, (size_t)(d + 1) * n), Xra(rr, (size_t)(d + 1) * n);
      for (unsigned i = 0; i < n; ++i) {
        const double c = std::cos(0.1 * (i + 1)), sn = std::sin(0.1 * (i + 1));
        for (unsigned col = 0; col <= d; ++col) {
          double v[4] = {0, 0, 0, 0};
          for (unsigned a = 0; a < d; ++a) v[a] = TrajectoryGroundTruth.getData()(a, (size_t)i * (d + 1) + col);
          const double v0 = c * v[0] - sn * v[1], v1 = sn * v[0] + c * v[1];  // a rotation in the first plane
          v[0] = v0;
          v[1] = v1;
          if (col == d) v[0] += 0.05 * i;
          const size_t cra = col < d ? (size_t)i * d + col : (size_t)d * n + i;
          for (unsigned a = 0; a < rr; ++a) Xse(a, (size_t)i * (d + 1) + col) = Xra(a, cra) = a < d ? v[a] : 0.0;
        }
      }
      const double fSE = pSE.f(Xse), fRA = pRA.f(Xra);
      EXPECT(fSE > 1e-3 && std::fabs(fSE - fRA) <= 1e-12 * fSE);
      const DCORA::Matrix gSE = pSE.RieGrad(Xse), gRA = pRA.RieGrad(Xra);
      double diff = 0, nrm = 0;
      for (unsigned i = 0; i < n; ++i)
        for (unsigned col = 0; col <= d; ++col) {
          const size_t cra = col < d ? (size_t)i * d + col : (size_t)d * n + i;
          for (unsigned a = 0; a < rr; ++a) {
            const double e = gSE(a, (size_t)i * (d + 1) + col) - gRA(a, cra);
            diff += e * e;
            nrm += gSE(a, (size_t)i * (d + 1) + col) * gSE(a, (size_t)i * (d + 1) + col);
          }
        }
      EXPECT(nrm > 1e-6 && std::sqrt(diff) <= 1e-11 * std::sqrt(nrm));
      // The default RTR (3 outer iterations from Delta = 100) may reject all of its steps from this start, on either
      // problem -- with the range-aided regularisation (lambda_max / 1e6, ref src/Graph.cpp:1901-1960) on a matrix
      // that has the translation gauge in its null space it does, and the oracle's restatement of the reference does
      // the same -- and then hands back its input: never an ascent; with room to shrink the radius both descend.
      DCORA::QuadraticOptimizer oSE(&pSE), oRA(&pRA);
      const DCORA::Matrix Yse = oSE.optimize(Xse), Yra = oRA.optimize(Xra);
      EXPECT(pSE.f(Yse) <= fSE * (1.0 + 1e-12) && pRA.f(Yra) <= fRA * (1.0 + 1e-12));
      DCORA::ROptParameters longer;
      longer.RTR_iterations = 20;
      DCORA::QuadraticOptimizer oSE20(&pSE, longer), oRA20(&pRA, longer);
      const DCORA::Matrix Yse20 = oSE20.optimize(Xse), Yra20 = oRA20.optimize(Xra);
      EXPECT(pSE.f(Yse20) < 0.5 * fSE && pSE.RieGradNorm(Yse20) < pSE.RieGradNorm(Xse));
      EXPECT(pRA.f(Yra20) < 0.5 * fRA && pRA.RieGradNorm(Yra20) < pRA.RieGradNorm(Xra));
      // the centralised agent of that type on those measurements: ground truth stays a fixed point
      DCORA::Agent poseOnly(id, options);
      poseOnly.setMeasurements(posesOnly);
      const DCORA::PointArray none(d, 0);
      poseOnly.initialize(&TrajectoryGroundTruth, &none, &none);
      EXPECT(poseOnly.num_unit_spheres() == 0 && poseOnly.num_landmarks() == 0);
      EXPECT(poseOnly.iterate());
      DCORA::Matrix Tpo, Spo, Lpo;
      EXPECT(poseOnly.getStatesInLocalFrame(&Tpo, &Spo, &Lpo));
      EXPECT(isApprox(TrajectoryGroundTruthAligned.getData(), Tpo, OPTIMIZATION_TOL));
    }
    std::printf("%s: d %u n %u l %u b %u ok\n", argv[f], d, n, l, b);
  }
  if (argc < 2) {
    std::fprintf(stderr, "usage: test_ra_facade file.pyfg ...\n");
    return 2;
  }
  return failures ? 1 : 0;
}
