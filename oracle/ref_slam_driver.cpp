// TEST INFRASTRUCTURE ONLY -- never linked into or called from the product library.
//
// Drop-in check of the solver plugin: the reference's OWN karto::Mapper (compiled in place from
// /root/reference, see oracle/Makefile) processes a scan queue with karto_hip::HipSpaSolver attached through
// Mapper::SetScanSolver -- i.e. the real karto::ScanSolver virtual interface (AddNode(Vertex*),
// AddConstraint(Edge*), Compute, GetCorrections, Clear; Mapper.h:954-1066; call sites Mapper.cpp:1425-1427,
// 1633-1635, 2015-2028).  A recording wrapper logs every call with its inputs and the corrections each
// Compute() returned, so that the test can replay exactly the same graphs through the CPU oracle
// (oracle/spa.py) and compare (tests/test_dropin_mapper_gpu.py).
//
// Mapper parameters = config/mapper_params_offline.yaml:31-66 (the offline_sync launch, BASELINE config[0]).
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <map>
#include <sstream>
#include <string>
#include <vector>
#include <shared_mutex>
#include <mutex>
#include <fstream>
#include <unordered_map>
#include <queue>
#include <chrono>
#include <algorithm>
#include <memory>
#include <atomic>
#include <thread>
#include <iomanip>

#define private public
#define protected public
#include "karto_sdk/Mapper.h"
#undef private
#undef protected
#include "karto_hip/karto_adaptor.hpp"

using namespace karto;

namespace
{

class RecordingSolver : public ScanSolver
{
public:
  explicit RecordingSolver(FILE * log) : m_pLog(log) {}
  virtual void Compute()
  {
    const auto t0 = std::chrono::steady_clock::now();
    m_Inner.Compute();
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    const ScanSolver::IdPoseVector & c = m_Inner.GetCorrections();
    std::fprintf(m_pLog, "X %zu %.6f\n", c.size(), ms);
    for (const auto & ip : c) {
      std::fprintf(m_pLog, "P %d %.17g %.17g %.17g\n", ip.first, ip.second.GetX(), ip.second.GetY(), ip.second.GetHeading());
    }
  }
  virtual void Configure(rclcpp_lifecycle::LifecycleNode::SharedPtr node) {m_Inner.Configure(node);}
  virtual const ScanSolver::IdPoseVector & GetCorrections() const {return m_Inner.GetCorrections();}
  virtual void Clear() {std::fprintf(m_pLog, "K\n"); m_Inner.Clear();}
  virtual void Reset() {std::fprintf(m_pLog, "R\n"); m_Inner.Reset();}
  virtual void AddNode(Vertex<LocalizedRangeScan> * pVertex)
  {
    const Pose2 p = pVertex->GetObject()->GetCorrectedPose();
    std::fprintf(m_pLog, "N %d %.17g %.17g %.17g\n", pVertex->GetObject()->GetUniqueId(), p.GetX(), p.GetY(), p.GetHeading());
    m_Inner.AddNode(pVertex);
  }
  virtual void AddConstraint(Edge<LocalizedRangeScan> * pEdge)
  {
    LinkInfo * li = (LinkInfo *)(pEdge->GetLabel());
    const Pose2 d = li->GetPoseDifference();
    const Matrix3 & c = li->GetCovariance();
    std::fprintf(m_pLog, "C %d %d %.17g %.17g %.17g", pEdge->GetSource()->GetObject()->GetUniqueId(),
      pEdge->GetTarget()->GetObject()->GetUniqueId(), d.GetX(), d.GetY(), d.GetHeading());
    for (int r = 0; r < 3; ++r) {for (int q = 0; q < 3; ++q) {std::fprintf(m_pLog, " %.17g", c(r, q));}}
    std::fprintf(m_pLog, "\n");
    m_Inner.AddConstraint(pEdge);
  }
  virtual void RemoveNode(kt_int32s id) {std::fprintf(m_pLog, "D %d\n", id); m_Inner.RemoveNode(id);}
  virtual void RemoveConstraint(kt_int32s a, kt_int32s b) {std::fprintf(m_pLog, "E %d %d\n", a, b); m_Inner.RemoveConstraint(a, b);}
private:
  FILE * m_pLog;
  karto_hip::HipSpaSolver m_Inner;
};

// config/mapper_params_offline.yaml:31-66
void configure_offline(Mapper & mapper, double loop_search_distance)
{
  mapper.setParamUseScanMatching(true);
  mapper.setParamUseScanBarycenter(true);
  mapper.setParamMinimumTravelDistance(0.5);
  mapper.setParamMinimumTravelHeading(0.5);
  mapper.setParamScanBufferSize(10);
  mapper.setParamScanBufferMaximumScanDistance(10.0);
  mapper.setParamLinkMatchMinimumResponseFine(0.1);
  mapper.setParamLinkScanMaximumDistance(1.5);
  mapper.setParamLoopSearchMaximumDistance(loop_search_distance);
  mapper.setParamDoLoopClosing(true);
  mapper.setParamLoopMatchMinimumChainSize(10);
  mapper.setParamLoopMatchMaximumVarianceCoarse(3.0);
  mapper.setParamLoopMatchMinimumResponseCoarse(0.35);
  mapper.setParamLoopMatchMinimumResponseFine(0.45);
  mapper.setParamCorrelationSearchSpaceDimension(0.5);
  mapper.setParamCorrelationSearchSpaceResolution(0.01);
  mapper.setParamCorrelationSearchSpaceSmearDeviation(0.1);
  mapper.setParamLoopSearchSpaceDimension(8.0);
  mapper.setParamLoopSearchSpaceResolution(0.05);
  mapper.setParamLoopSearchSpaceSmearDeviation(0.03);
  mapper.setParamDistanceVariancePenalty(0.5);
  mapper.setParamAngleVariancePenalty(1.0);
  mapper.setParamFineSearchAngleOffset(0.00349);
  mapper.setParamCoarseSearchAngleOffset(0.349);
  mapper.setParamCoarseAngleResolution(0.0349);
  mapper.setParamMinimumAnglePenalty(0.9);
  mapper.setParamMinimumDistancePenalty(0.5);
  mapper.setParamUseResponseExpansion(true);
}

}  // namespace

extern "C" {

// Golden data for the loop-candidate enumeration: runs the scan queue through the
// reference Mapper WITHOUT a solver (CorrectPoses is then a no-op, Mapper.cpp:2016; loop closures still add
// their edges), then dumps the graph -- reference positions GetReferencePose(useScanBarycenter) in scan
// list order, adjacency in Vertex::GetAdjacentVertices order (Mapper.h:338-361) -- and, for every scan as
// query, FindNearLinkedScans (Mapper.cpp:1795-1806) and the chains successive FindPossibleLoopClosure calls
// return (Mapper.cpp:1960-2010, driven like TryCloseLoop does, Mapper.cpp:1500-1560).
int ref_slam_enumerate(
  int n_scans, int n_beams, const double * ranges, const double * odom, double loop_search_distance,
  const char * out_path)
{
  FILE * out = std::fopen(out_path, "w");
  if (!out) {return -2;}
  int accepted = 0;
  try {
    Mapper mapper;
    configure_offline(mapper, loop_search_distance);
    for (int i = 0; i < n_scans; ++i) {
      RangeReadingsVector r(ranges + static_cast<size_t>(i) * n_beams, ranges + static_cast<size_t>(i + 1) * n_beams);
      LocalizedRangeScan * s = new LocalizedRangeScan(Name("laser0"), r);
      const Pose2 p(odom[3 * i], odom[3 * i + 1], odom[3 * i + 2]);
      s->SetOdometricPose(p);
      s->SetCorrectedPose(p);
      s->SetTime(0.1 * i);
      Matrix3 cov;
      if (mapper.Process(s, &cov)) {++accepted;} else {delete s;}
    }
    const Name sensor("laser0");
    MapperGraph * graph = mapper.m_pGraph;
    const kt_bool bary = mapper.m_pUseScanBarycenter->GetValue();
    // MapperSensorManager::GetScans is an inline of Mapper.cpp; GetAllScans returns the same scans in scan-id order
    const LocalizedRangeScanVector scans = mapper.m_pMapperSensorManager->GetAllScans();
    std::fprintf(out, "G %d %.17g %u\n", accepted, loop_search_distance, mapper.m_pLoopMatchMinimumChainSize->GetValue());
    for (LocalizedRangeScan * s : scans) {
      const Pose2 p = s->GetReferencePose(bary);
      std::fprintf(out, "S %d %.17g %.17g\n", s->GetStateId(), p.GetX(), p.GetY());
      std::vector<Vertex<LocalizedRangeScan> *> adj = graph->GetVertex(s)->GetAdjacentVertices();
      std::fprintf(out, "A %d %zu", s->GetStateId(), adj.size());
      for (auto * v : adj) {std::fprintf(out, " %d", v->GetObject()->GetStateId());}
      std::fprintf(out, "\n");
    }
    for (LocalizedRangeScan * q : scans) {
      const LocalizedRangeScanVector linked = graph->FindNearLinkedScans(q, loop_search_distance);
      std::fprintf(out, "L %d %zu", q->GetStateId(), linked.size());
      for (auto * s : linked) {std::fprintf(out, " %d", s->GetStateId());}
      std::fprintf(out, "\n");
      // FindNearChains (Mapper.cpp:1683-1793) with the mapper's link_scan_maximum_distance, and for every chain the
      // scan GetClosestScanToPose (Mapper.cpp:1563-1582) picks for the query's reference pose
      {
        const std::vector<LocalizedRangeScanVector> near = graph->FindNearChains(q);
        const Pose2 qp = q->GetReferencePose(bary);
        for (const LocalizedRangeScanVector & c : near) {
          std::fprintf(out, "N %d %d %d %d\n", q->GetStateId(), c.front()->GetStateId(), c.back()->GetStateId(),
            graph->GetClosestScanToPose(c, qp)->GetStateId());
        }
      }
      kt_int32u start = 0;
      LocalizedRangeScanVector chain = graph->FindPossibleLoopClosure(q, sensor, start);
      while (!chain.empty()) {
        std::fprintf(out, "H %d %u %zu", q->GetStateId(), start, chain.size());
        for (auto * s : chain) {std::fprintf(out, " %d", s->GetStateId());}
        std::fprintf(out, "\n");
        chain = graph->FindPossibleLoopClosure(q, sensor, start);
      }
    }
    // ComputeWeightedMean (Mapper.cpp:1914-1958) known answers: deterministic pose / covariance sets
    {
      uint64_t lcg = 88172645463325252ull;
      auto rnd = [&]() {lcg = lcg * 6364136223846793005ull + 1442695040888963407ull; return (lcg >> 11) * (1.0 / 9007199254740992.0);};
      for (int t = 0; t < 12; ++t) {
        const int k = 1 + t % 5;
        Pose2Vector means;
        std::vector<Matrix3> covs;
        std::fprintf(out, "W %d", k);
        for (int i = 0; i < k; ++i) {
          const Pose2 p(20.0 * rnd() - 10.0, 20.0 * rnd() - 10.0, 6.0 * rnd() - 3.0);
          const double a = 0.01 + rnd(), b = 0.01 + rnd(), c = 0.01 + 0.2 * rnd(), r = 0.3 * rnd() * std::sqrt(a * b);
          Matrix3 m;
          m(0, 0) = a; m(1, 1) = b; m(2, 2) = c; m(0, 1) = r; m(1, 0) = r;
          means.push_back(p); covs.push_back(m);
          std::fprintf(out, " %.17g %.17g %.17g %.17g %.17g %.17g %.17g", p.GetX(), p.GetY(), p.GetHeading(), a, b, c, r);
        }
        const Pose2 w = graph->ComputeWeightedMean(means, covs);
        std::fprintf(out, " | %.17g %.17g %.17g\n", w.GetX(), w.GetY(), w.GetHeading());
      }
    }
  } catch (const std::exception & e) {
    std::fprintf(out, "! %s\n", e.what());
    std::fclose(out);
    return -1;
  }
  std::fclose(out);
  return accepted;
}


// Runs the scan queue through karto::Mapper::Process with the GPU solver plugin attached.  `ranges` is
// n_scans x n_beams, `odom` n_scans x 3.  Writes the call log to `log_path` and the final corrected pose of
// every ACCEPTED scan still in the graph (id, x, y, heading) to `out` (capacity cap rows).  Returns the number of
// those scans, or -1 when the plugin could not be constructed (no GPU).
// Removal schedule (lifelong mode's graph edits, without its policy): after queue scan remove_at[r] has been processed,
// the node of scan id remove_id[r] is taken out the way LifelongSlamToolbox::removeFromSlamGraph does it
// (src/experimental/slam_toolbox_lifelong.cpp:330-342): Mapper::RemoveNodeFromGraph + MapperSensorManager::RemoveScan.
static int run_queue(
  int n_scans, int n_beams, const double * ranges, const double * odom, double loop_search_distance,
  const char * log_path, double * out, int cap, const int * remove_at, const int * remove_id, int n_remove)
{
  FILE * log = std::fopen(log_path, "w");
  if (!log) {return -2;}
  int accepted = 0;
  try {
    Mapper mapper;
    configure_offline(mapper, loop_search_distance);
    RecordingSolver solver(log);
    mapper.SetScanSolver(&solver);
    std::vector<LocalizedRangeScan *> kept;
    for (int i = 0; i < n_scans; ++i) {
      RangeReadingsVector r(ranges + static_cast<size_t>(i) * n_beams, ranges + static_cast<size_t>(i + 1) * n_beams);
      LocalizedRangeScan * s = new LocalizedRangeScan(Name("laser0"), r);
      const Pose2 p(odom[3 * i], odom[3 * i + 1], odom[3 * i + 2]);
      s->SetOdometricPose(p);
      s->SetCorrectedPose(p);
      s->SetTime(0.1 * i);
      Matrix3 cov;
      if (mapper.Process(s, &cov)) {
        kept.push_back(s);
        // KH_LOG_FINAL_POSES (debugging aid, both sides): the scan's pose when Process() returns -- after the weighted mean
        // of AddEdges and any loop closure, which no solver call shows
        if (std::getenv("KH_LOG_FINAL_POSES")) {
          const Pose2 f = s->GetCorrectedPose();
          std::fprintf(log, "F %d %.17g %.17g %.17g\n", s->GetUniqueId(), f.GetX(), f.GetY(), f.GetHeading());
        }
      } else {
        delete s;
      }
      for (int r2 = 0; r2 < n_remove; ++r2) {
        if (remove_at[r2] != i) {continue;}
        LocalizedRangeScan * victim = mapper.m_pMapperSensorManager->GetScan(remove_id[r2]);
        if (!victim) {std::fprintf(log, "! schedule names unknown scan %d\n", remove_id[r2]); continue;}
        Vertex<LocalizedRangeScan> * vertex = mapper.m_pGraph->GetVertex(victim);
        mapper.RemoveNodeFromGraph(vertex);
        mapper.m_pMapperSensorManager->RemoveScan(victim);
        vertex->RemoveObject();
        delete vertex;
        kept.erase(std::find(kept.begin(), kept.end(), victim));
      }
    }
    for (LocalizedRangeScan * s : kept) {
      if (accepted < cap) {
        const Pose2 p = s->GetCorrectedPose();
        out[4 * accepted] = s->GetUniqueId(); out[4 * accepted + 1] = p.GetX();
        out[4 * accepted + 2] = p.GetY(); out[4 * accepted + 3] = p.GetHeading();
      }
      ++accepted;
    }
    std::fprintf(log, "Z %d\n", accepted);
  } catch (const std::exception & e) {
    std::fprintf(log, "! %s\n", e.what());
    std::fclose(log);
    return -1;
  }
  std::fclose(log);
  return accepted;
}

int ref_slam_run(
  int n_scans, int n_beams, const double * ranges, const double * odom, double loop_search_distance,
  const char * log_path, double * out, int cap)
{
  return run_queue(n_scans, n_beams, ranges, odom, loop_search_distance, log_path, out, cap, nullptr, nullptr, 0);
}

int ref_slam_run_schedule(
  int n_scans, int n_beams, const double * ranges, const double * odom, double loop_search_distance,
  const char * log_path, double * out, int cap, const int * remove_at, const int * remove_id, int n_remove)
{
  return run_queue(n_scans, n_beams, ranges, odom, loop_search_distance, log_path, out, cap, remove_at, remove_id, n_remove);
}

}  // extern "C"
