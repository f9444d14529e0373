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
    std::printf("%s: d %u n %u l %u b %u ok\n", argv[f], d, n, l, b);
  }
  if (argc < 2) {
    std::fprintf(stderr, "usage: test_ra_facade file.pyfg ...\n");
    return 2;
  }
  return failures ? 1 : 0;
}
