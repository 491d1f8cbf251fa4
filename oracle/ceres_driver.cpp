// TEST INFRASTRUCTURE -- the pin that is missing for the solver half: the reference's own Ceres problem
// on a pose-graph file, so that kh_spa_compute can be laid beside the real thing where Ceres is installed.
//
//   make -C oracle ceres          (builds _ref/ceres_driver only where <ceres/ceres.h> and Eigen are found; this image has neither)
//   _ref/ceres_driver graph.g2o   -> one JSON document on stdout: per-iteration trace + final poses
//
// Nothing of the reference is copied: the cost functor (PoseGraph2dErrorTerm), the angle parameterisation and NormalizeAngle
// are the reference's own solvers/ceres_utils.h, included where it lies (-I $(REF)/solvers); what this file adds is what
// solver_plugins::CeresSolver does around them, restated without ROS:
//   * options                         solvers/ceres_solver.cpp:157-186 (Configure)
//   * one 1-D parameter block per x, y, yaw of a node, yaw blocks with AngleLocalParameterization,
//     sqrt information = information.llt().matrixU()                        :363-385 (AddConstraint)
//   * the first node added is held constant                                 :227-243 (Compute)
// The file format is the library's g2o SE2 text (kh_spa_save: VERTEX_SE2 id x y yaw / FIX id / EDGE_SE2 a b dx dy dyaw + the six
// upper-triangle entries of the information matrix).
#include <ceres/ceres.h>
#include <Eigen/Dense>

#include <cstdio>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <unordered_map>
#include <vector>

#include "ceres_utils.h"

namespace
{

struct Trace : public ceres::IterationCallback
{
  std::vector<ceres::IterationSummary> rows;
  ceres::CallbackReturnType operator()(const ceres::IterationSummary & s) override
  {
    rows.push_back(s);
    return ceres::SOLVER_CONTINUE;
  }
};

}  // namespace

int main(int argc, char ** argv)
{
  if (argc < 2) {std::fprintf(stderr, "usage: ceres_driver graph.g2o [function_tolerance]\n"); return 2;}
  std::ifstream in(argv[1]);
  if (!in) {std::fprintf(stderr, "cannot open %s\n", argv[1]); return 2;}
  std::vector<int> order;                                  // ids in AddNode order
  std::unordered_map<int, Eigen::Vector3d> nodes;
  struct Edge {int a, b; Eigen::Vector3d z; Eigen::Matrix3d info;};
  std::vector<Edge> edges;
  std::string line;
  while (std::getline(in, line)) {
    std::istringstream ss(line);
    std::string tag;
    ss >> tag;
    if (tag == "VERTEX_SE2") {
      int id; double x, y, t;
      ss >> id >> x >> y >> t;
      nodes.emplace(id, Eigen::Vector3d(x, y, t));
      order.push_back(id);
    } else if (tag == "EDGE_SE2") {
      Edge e; double i11, i12, i13, i22, i23, i33;
      ss >> e.a >> e.b >> e.z(0) >> e.z(1) >> e.z(2) >> i11 >> i12 >> i13 >> i22 >> i23 >> i33;
      e.info << i11, i12, i13, i12, i22, i23, i13, i23, i33;
      edges.push_back(e);
    }
  }
  if (order.empty()) {std::fprintf(stderr, "no nodes\n"); return 2;}

  ceres::Solver::Options options;                          // ceres_solver.cpp:157-186
  options.linear_solver_type = ceres::SPARSE_NORMAL_CHOLESKY;
  options.preconditioner_type = ceres::JACOBI;
  options.trust_region_strategy_type = ceres::LEVENBERG_MARQUARDT;
  options.function_tolerance = argc > 2 ? std::atof(argv[2]) : 1e-3;
  options.gradient_tolerance = 1e-6;
  options.parameter_tolerance = 1e-3;
  options.sparse_linear_algebra_library_type = ceres::SUITE_SPARSE;
  options.max_num_consecutive_invalid_steps = 3;
  options.max_consecutive_nonmonotonic_steps = options.max_num_consecutive_invalid_steps;
  options.num_threads = 50;
  options.use_nonmonotonic_steps = true;
  options.jacobi_scaling = true;
  options.min_relative_decrease = 1e-3;
  options.initial_trust_region_radius = 1e4;
  options.max_trust_region_radius = 1e8;
  options.min_trust_region_radius = 1e-16;
  options.min_lm_diagonal = 1e-6;
  options.max_lm_diagonal = 1e32;
  options.dynamic_sparsity = true;
  Trace trace;
  options.callbacks.push_back(&trace);
  options.update_state_every_iteration = false;

  ceres::Problem::Options popt;
  popt.loss_function_ownership = ceres::Ownership::DO_NOT_TAKE_OWNERSHIP;
  ceres::Problem problem(popt);
  ceres::LocalParameterization * angle = AngleLocalParameterization::Create();
  for (const Edge & e : edges) {                            // ceres_solver.cpp:363-385
    auto a = nodes.find(e.a), b = nodes.find(e.b);
    if (a == nodes.end() || b == nodes.end() || a == b) {continue;}
    const Eigen::Matrix3d sqrt_information = e.info.llt().matrixU();
    ceres::CostFunction * cost = PoseGraph2dErrorTerm::Create(e.z(0), e.z(1), e.z(2), sqrt_information);
    problem.AddResidualBlock(cost, NULL, &a->second(0), &a->second(1), &a->second(2), &b->second(0), &b->second(1), &b->second(2));
    problem.SetParameterization(&a->second(2), angle);
    problem.SetParameterization(&b->second(2), angle);
  }
  Eigen::Vector3d & first = nodes.at(order[0]);            // ceres_solver.cpp:227-243
  if (problem.HasParameterBlock(&first(0)) && problem.HasParameterBlock(&first(1)) && problem.HasParameterBlock(&first(2))) {
    problem.SetParameterBlockConstant(&first(0));
    problem.SetParameterBlockConstant(&first(1));
    problem.SetParameterBlockConstant(&first(2));
  }
  ceres::Solver::Summary summary;
  ceres::Solve(options, &problem, &summary);

  std::printf("{\"usable\": %d, \"initial_cost\": %.17g, \"final_cost\": %.17g, \"iterations\": [", summary.IsSolutionUsable() ? 1 : 0,
    summary.initial_cost, summary.final_cost);
  for (size_t i = 0; i < trace.rows.size(); ++i) {
    const ceres::IterationSummary & r = trace.rows[i];
    std::printf("%s{\"iteration\": %d, \"cost\": %.17g, \"cost_change\": %.17g, \"radius\": %.17g, \"step_norm\": %.17g, \"valid\": %d, \"successful\": %d, "
      "\"relative_decrease\": %.17g, \"gradient_max_norm\": %.17g}", i ? ", " : "", r.iteration, r.cost, r.cost_change, r.trust_region_radius, r.step_norm,
      r.step_is_valid ? 1 : 0, r.step_is_successful ? 1 : 0, r.relative_decrease, r.gradient_max_norm);
  }
  std::printf("], \"poses\": [");
  for (size_t i = 0; i < order.size(); ++i) {
    const Eigen::Vector3d & p = nodes.at(order[i]);
    std::printf("%s[%d, %.17g, %.17g, %.17g]", i ? ", " : "", order[i], p(0), p(1), p(2));
  }
  std::printf("]}\n");
  return 0;
}
