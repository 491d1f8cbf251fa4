// karto_adaptor.hpp -- thin, header-only C++ adaptors that put the reference's class surface back
// on top of the C ABI (karto_hip.h).  Include it from a translation unit that already sees the
// slam_toolbox headers (karto_sdk/Mapper.h); nothing here needs ROS beyond what Mapper.h itself pulls.
//
//   karto_hip::HipScanMatcher   same public surface as karto::ScanMatcher
//                               (lib/karto_sdk/include/karto_sdk/Mapper.h:1322-1544): Create, MatchScan<T>,
//                               CorrelateScan; throws std::runtime_error where the reference throws
//                               (Mapper.cpp:786-796, 828); Create returns NULL like Mapper.cpp:481-493.
//   karto_hip::HipSpaSolver     a karto::ScanSolver (Mapper.h:954-1066) with the semantics of
//                               solver_plugins::CeresSolver (solvers/ceres_solver.cpp); export it with
//                               PLUGINLIB_EXPORT_CLASS(karto_hip::HipSpaSolver, karto::ScanSolver).
//
// See INTEGRATION.md for the three-line changes in slam_toolbox that select these classes.
#ifndef KARTO_HIP__KARTO_ADAPTOR_HPP_
#define KARTO_HIP__KARTO_ADAPTOR_HPP_

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

#include "karto_sdk/Mapper.h"
#include "../karto_hip.h"

namespace karto_hip
{

namespace detail
{
inline kh_scan to_scan(karto::LocalizedRangeScan * pScan, std::vector<double> & points)
{
  kh_scan s;
  const karto::PointVectorDouble & pts = pScan->GetPointReadings();        // unfiltered (Karto.h:5613-5628)
  s.n = static_cast<int32_t>(pScan->GetNumberOfRangeReadings());
  points.resize(2 * pts.size());
  for (size_t i = 0; i < pts.size(); ++i) {points[2 * i] = pts[i].GetX(); points[2 * i + 1] = pts[i].GetY();}
  s.ranges = pScan->GetRangeReadings();
  s.points_xy = points.data();
  const karto::Pose2 pose = pScan->GetSensorPose();
  s.sensor_pose[0] = pose.GetX(); s.sensor_pose[1] = pose.GetY(); s.sensor_pose[2] = pose.GetHeading();
  s.device_points_xy = nullptr;            // karto keeps its scans on the host: uploaded per call
  return s;
}

// HIP device the adaptors compute on when the caller names none: environment variable KARTO_HIP_DEVICE (one process
// per GPU: the launcher exports it next to the rank), 0 otherwise
inline int default_device()
{
  const char * e = std::getenv("KARTO_HIP_DEVICE");
  return e ? std::atoi(e) : 0;
}

template<class T>
inline T stored_parameter(karto::Mapper * pMapper, const char * name)
{
  // the values AS STORED (the public getters return sqrt() of the two variance penalties, Mapper.cpp:2410-2418)
  karto::AbstractParameter * p = pMapper->GetParameterManager()->Get(name);
  return static_cast<karto::Parameter<T> *>(p)->GetValue();
}
}  // namespace detail

class HipScanMatcher
{
public:
  virtual ~HipScanMatcher()
  {
    for (auto & kv : m_Resident) {kh_device_free(kv.second.device_points);}
    kh_matcher_destroy(m_pHandle);
  }

  // Base scans stay resident on the device between calls (default on): a processed scan serves as a base scan of dozens of
  // consecutive matches (the running-scan window slides by one per scan), so its 17 KB of point readings are uploaded when it
  // first appears and again only after its pose has moved (CorrectPoses), instead of with every MatchScan.  The copies are
  // keyed by the scan's unique id and checked against its sensor pose and reading count; at most `capacity` scans are kept
  // (least recently used go first).  0 switches the cache off (every call uploads what it reads, like the round-2 adaptor).
  void SetResidentScanCapacity(size_t capacity)
  {
    m_ResidentCapacity = capacity;
    if (capacity == 0) {
      for (auto & kv : m_Resident) {kh_device_free(kv.second.device_points);}
      m_Resident.clear();
    }
  }

  static HipScanMatcher * Create(
    karto::Mapper * pMapper, kt_double searchSize, kt_double resolution,
    kt_double smearDeviation, kt_double rangeThreshold, int device = -1)
  {
    if (device < 0) {device = detail::default_device();}
    kh_matcher * h = nullptr;
    const int rc = kh_matcher_create(searchSize, resolution, smearDeviation, rangeThreshold, device, 1, &h);
    if (rc == KH_ERR_INVALID_ARG) {return NULL;}
    if (rc != KH_OK) {throw std::runtime_error(std::string("karto_hip: ") + kh_last_error());}
    HipScanMatcher * m = new HipScanMatcher();
    m->m_pHandle = h;
    m->m_Device = device;
    m->m_pMapper = pMapper;
    m->m_Resolution = resolution; m->m_SmearDeviation = smearDeviation;
    return m;
  }

  template<class T = karto::LocalizedRangeScanVector>
  kt_double MatchScan(
    karto::LocalizedRangeScan * pScan, const T & rBaseScans, karto::Pose2 & rMean,
    karto::Matrix3 & rCovariance, kt_bool doPenalize = true, kt_bool doRefineMatch = true)
  {
    SyncParameters();
    m_pLastScan = pScan;
    std::vector<std::vector<double>> store(1);
    std::vector<kh_scan> base;
    const kh_scan query = detail::to_scan(pScan, store[0]);
    Collect(rBaseScans, base, store);
    MakeResident(rBaseScans, base);
    double mean[3], cov[9], response = 0.0;
    const int rc = kh_matcher_match(
      m_pHandle, &query, base.data(), static_cast<int32_t>(base.size()), doPenalize, doRefineMatch, mean, cov, &response);
    Check(rc);
    rMean = karto::Pose2(mean[0], mean[1], mean[2]);
    for (int r = 0; r < 3; ++r) {for (int c = 0; c < 3; ++c) {rCovariance(r, c) = cov[3 * r + c];}}
    return response;
  }

  kt_double CorrelateScan(
    karto::LocalizedRangeScan * pScan, const karto::Pose2 & rSearchCenter,
    const karto::Vector2<kt_double> & rSearchSpaceOffset, const karto::Vector2<kt_double> & rSearchSpaceResolution,
    kt_double searchAngleOffset, kt_double searchAngleResolution, kt_bool doPenalize,
    karto::Pose2 & rMean, karto::Matrix3 & rCovariance, kt_bool doingFineMatch)
  {
    SyncParameters();
    m_pLastScan = pScan;
    std::vector<double> pts;
    const kh_scan query = detail::to_scan(pScan, pts);
    const double center[3] = {rSearchCenter.GetX(), rSearchCenter.GetY(), rSearchCenter.GetHeading()};
    const double off[2] = {rSearchSpaceOffset.GetX(), rSearchSpaceOffset.GetY()};
    const double res[2] = {rSearchSpaceResolution.GetX(), rSearchSpaceResolution.GetY()};
    double mean[3], cov[9], response = 0.0;
    for (int r = 0; r < 3; ++r) {for (int c = 0; c < 3; ++c) {cov[3 * r + c] = rCovariance(r, c);}}
    const int rc = kh_matcher_correlate(
      m_pHandle, 0, &query, center, off, res, searchAngleOffset, searchAngleResolution, doPenalize,
      doingFineMatch, mean, cov, &response);
    Check(rc);
    rMean = karto::Pose2(mean[0], mean[1], mean[2]);
    for (int r = 0; r < 3; ++r) {for (int c = 0; c < 3; ++c) {rCovariance(r, c) = cov[3 * r + c];}}
    return response;
  }

  // ScanMatcher::ComputePositionalCovariance (Mapper.h:1405-1412, Mapper.cpp:874-966): on the search-space probabilities
  // the last coarse CorrelateScan left in the matcher, like the reference's m_pSearchSpaceProbs
  void ComputePositionalCovariance(
    const karto::Pose2 & rBestPose, kt_double bestResponse, const karto::Pose2 & rSearchCenter,
    const karto::Vector2<kt_double> & rSearchSpaceOffset, const karto::Vector2<kt_double> & rSearchSpaceResolution,
    kt_double searchAngleResolution, karto::Matrix3 & rCovariance)
  {
    const double best[3] = {rBestPose.GetX(), rBestPose.GetY(), rBestPose.GetHeading()};
    const double center[3] = {rSearchCenter.GetX(), rSearchCenter.GetY(), rSearchCenter.GetHeading()};
    const double off[2] = {rSearchSpaceOffset.GetX(), rSearchSpaceOffset.GetY()};
    const double res[2] = {rSearchSpaceResolution.GetX(), rSearchSpaceResolution.GetY()};
    double cov[9];
    Check(kh_matcher_positional_covariance(m_pHandle, 0, best, bestResponse, center, off, res, searchAngleResolution, cov));
    for (int r = 0; r < 3; ++r) {for (int c = 0; c < 3; ++c) {rCovariance(r, c) = cov[3 * r + c];}}
  }

  // ScanMatcher::ComputeAngularCovariance (Mapper.h:1421-1427, Mapper.cpp:977-1025) for the scan of the last
  // MatchScan / CorrelateScan call (the reference reads the lookup table that call computed); writes rCovariance(2, 2)
  void ComputeAngularCovariance(
    const karto::Pose2 & rBestPose, kt_double bestResponse, const karto::Pose2 & rSearchCenter,
    kt_double searchAngleOffset, kt_double searchAngleResolution, karto::Matrix3 & rCovariance)
  {
    if (!m_pLastScan) {throw std::runtime_error("karto_hip: ComputeAngularCovariance before any CorrelateScan");}
    std::vector<double> pts;
    const kh_scan query = detail::to_scan(m_pLastScan, pts);
    const double best[3] = {rBestPose.GetX(), rBestPose.GetY(), rBestPose.GetHeading()};
    const double center[3] = {rSearchCenter.GetX(), rSearchCenter.GetY(), rSearchCenter.GetHeading()};
    double cov[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    Check(kh_matcher_angular_covariance(m_pHandle, 0, &query, best, bestResponse, center, searchAngleOffset,
      searchAngleResolution, cov));
    rCovariance(2, 2) = cov[8];
  }

  // ScanMatcher::GetCorrelationGrid (Mapper.h:1433-1437, "for debugging"): a host-side karto::CorrelationGrid holding the
  // bytes, the region of interest and the coordinate converter of the grid in HBM; refreshed on every call, owned here
  karto::CorrelationGrid * GetCorrelationGrid()
  {
    kh_grid_info gi;
    Check(kh_matcher_grid_info(m_pHandle, 0, &gi));
    if (!m_pGridMirror) {
      m_pGridMirror.reset(karto::CorrelationGrid::CreateGrid(gi.roi_w, gi.roi_h, m_Resolution, m_SmearDeviation));
    }
    Check(kh_matcher_read_grid(m_pHandle, 0, m_pGridMirror->GetDataPointer()));
    m_pGridMirror->GetCoordinateConverter()->SetOffset(karto::Vector2<kt_double>(gi.offset_x, gi.offset_y));
    return m_pGridMirror.get();
  }

  // karto::ScanMatcher::operator()(const kt_double &) is the row functor its own CorrelateScan hands to TBB
  // (Mapper.cpp:641-694, 773); it works on private members of a search in progress and has no meaning outside one.
  // Kept so that code naming it still compiles; the rows of a search are scored by the GPU inside CorrelateScan.
  void operator()(const kt_double &) const
  {
    throw std::logic_error("karto_hip::HipScanMatcher: operator()(y) is internal to CorrelateScan");
  }

  kh_matcher * GetHandle() const {return m_pHandle;}

protected:
  HipScanMatcher() : m_pHandle(nullptr), m_pMapper(nullptr), m_pLastScan(nullptr), m_Resolution(0), m_SmearDeviation(0) {}

private:
  void SyncParameters()
  {
    // the eight parameters karto::ScanMatcher reads through friend access (Mapper.cpp:590-627, 671-682)
    kh_match_params p;
    p.coarse_search_angle_offset = detail::stored_parameter<kt_double>(m_pMapper, "CoarseSearchAngleOffset");
    p.coarse_angle_resolution = detail::stored_parameter<kt_double>(m_pMapper, "CoarseAngleResolution");
    p.fine_search_angle_offset = detail::stored_parameter<kt_double>(m_pMapper, "FineSearchAngleOffset");
    p.use_response_expansion = detail::stored_parameter<kt_bool>(m_pMapper, "UseResponseExpansion") ? 1 : 0;
    p.distance_variance_penalty = detail::stored_parameter<kt_double>(m_pMapper, "DistanceVariancePenalty");
    p.minimum_distance_penalty = detail::stored_parameter<kt_double>(m_pMapper, "MinimumDistancePenalty");
    p.angle_variance_penalty = detail::stored_parameter<kt_double>(m_pMapper, "AngleVariancePenalty");
    p.minimum_angle_penalty = detail::stored_parameter<kt_double>(m_pMapper, "MinimumAnglePenalty");
    kh_matcher_set_params(m_pHandle, &p);
  }

  static void Collect(
    const karto::LocalizedRangeScanVector & rScans, std::vector<kh_scan> & base,
    std::vector<std::vector<double>> & store)
  {
    store.reserve(store.size() + rScans.size());
    for (karto::LocalizedRangeScan * s : rScans) {
      if (s == NULL) {continue;}                       // Mapper.cpp:1039-1041
      store.emplace_back();
      base.push_back(detail::to_scan(s, store.back()));
    }
  }
  static void Collect(
    const karto::LocalizedRangeScanMap & rScans, std::vector<kh_scan> & base,
    std::vector<std::vector<double>> & store)
  {
    store.reserve(store.size() + rScans.size());
    for (const auto & kv : rScans) {
      if (kv.second == NULL) {continue;}               // Mapper.cpp:1059-1061
      store.emplace_back();
      base.push_back(detail::to_scan(kv.second, store.back()));
    }
  }
  // device copies of the base scans' readings (kh_scan::device_points_xy), see SetResidentScanCapacity
  struct ResidentScan
  {
    double pose[3] = {0, 0, 0};
    int32_t n = 0;
    void * device_points = nullptr;
    uint64_t used = 0;
  };
  void MakeResident(karto::LocalizedRangeScan * pScan, kh_scan & s)
  {
    if (m_ResidentCapacity == 0 || pScan == NULL || s.n <= 0 || pScan->GetUniqueId() < 0) {return;}
    ResidentScan & r = m_Resident[pScan->GetUniqueId()];
    const int64_t bytes = static_cast<int64_t>(sizeof(double)) * 2 * s.n;
    const bool same = r.device_points != nullptr && r.n == s.n && r.pose[0] == s.sensor_pose[0] && r.pose[1] == s.sensor_pose[1] &&
      r.pose[2] == s.sensor_pose[2];
    if (!same) {
      if (r.device_points != nullptr && r.n != s.n) {kh_device_free(r.device_points); r.device_points = nullptr;}
      if (r.device_points == nullptr && kh_device_malloc(m_Device, bytes, &r.device_points) != KH_OK) {
        m_Resident.erase(pScan->GetUniqueId());
        return;                                      // no room: this call uploads the scan itself
      }
      if (kh_device_upload(r.device_points, s.points_xy, bytes) != KH_OK) {
        kh_device_free(r.device_points);
        m_Resident.erase(pScan->GetUniqueId());
        return;
      }
      r.n = s.n;
      r.pose[0] = s.sensor_pose[0]; r.pose[1] = s.sensor_pose[1]; r.pose[2] = s.sensor_pose[2];
    }
    r.used = ++m_ResidentClock;
    s.device_points_xy = static_cast<const double *>(r.device_points);
  }
  void MakeResident(const karto::LocalizedRangeScanVector & rScans, std::vector<kh_scan> & base)
  {
    const uint64_t callStart = m_ResidentClock;        // every scan this call touches gets a stamp above it
    size_t k = 0;
    for (karto::LocalizedRangeScan * p : rScans) {
      if (p == NULL) {continue;}
      MakeResident(p, base[k++]);
    }
    Trim(callStart);
  }
  void MakeResident(const karto::LocalizedRangeScanMap & rScans, std::vector<kh_scan> & base)
  {
    const uint64_t callStart = m_ResidentClock;
    size_t k = 0;
    for (const auto & kv : rScans) {
      if (kv.second == NULL) {continue;}
      MakeResident(kv.second, base[k++]);
    }
    Trim(callStart);
  }
  // Evicts least-recently-used scans down to the capacity -- but never one the CURRENT call touched (stamp above callStart):
  // kh_matcher_match reads those device buffers right after.  A call whose base chain is larger than the capacity simply
  // leaves the cache over its capacity until the next call trims it.
  void Trim(uint64_t callStart)
  {
    if (m_Resident.size() <= m_ResidentCapacity) {return;}
    std::vector<std::pair<uint64_t, int32_t>> order;
    order.reserve(m_Resident.size());
    for (const auto & kv : m_Resident) {
      if (kv.second.used <= callStart) {order.emplace_back(kv.second.used, kv.first);}
    }
    std::sort(order.begin(), order.end());
    for (size_t i = 0; i < order.size() && m_Resident.size() > m_ResidentCapacity; ++i) {
      auto it = m_Resident.find(order[i].second);
      kh_device_free(it->second.device_points);
      m_Resident.erase(it);
    }
  }

  static void Check(int rc)
  {
    if (rc == KH_ERR_SEARCH) {throw std::runtime_error("Mapper FATAL ERROR - Unable to find best position");}
    if (rc != KH_OK) {throw std::runtime_error(std::string("karto_hip: ") + kh_last_error());}
  }

  kh_matcher * m_pHandle;
  int m_Device = 0;
  std::unordered_map<int32_t, ResidentScan> m_Resident;
  size_t m_ResidentCapacity = 4096;                  // 4096 scans x 17 KB = 70 MB
  uint64_t m_ResidentClock = 0;
  karto::Mapper * m_pMapper;
  karto::LocalizedRangeScan * m_pLastScan;
  kt_double m_Resolution, m_SmearDeviation;
  std::unique_ptr<karto::CorrelationGrid> m_pGridMirror;
};

// ---------------------------------------------------------------------------------------------
class HipSpaSolver : public karto::ScanSolver
{
public:
  // device < 0: environment variable KARTO_HIP_DEVICE, else 0 (pluginlib constructs plugins without arguments)
  explicit HipSpaSolver(int device = -1) : m_pHandle(nullptr)
  {
    if (device < 0) {device = detail::default_device();}
    if (kh_spa_create(device, &m_pHandle) != KH_OK) {throw std::runtime_error(std::string("karto_hip: ") + kh_last_error());}
  }

  // Multi-GPU (one process per GPU, every process holds the same graph): the edge linearisation is sharded over the
  // ranks of `comm` and H, g are summed with RCCL inside the library (kh_comm_create, karto_hip.h); NULL switches it off
  void SetCommunicator(kh_comm * comm)
  {
    std::lock_guard<std::mutex> lock(m_Mutex);
    kh_spa_set_comm(m_pHandle, comm);
  }
  virtual ~HipSpaSolver() {kh_spa_destroy(m_pHandle);}

  // CeresSolver::Configure (ceres_solver.cpp:27-186) reads ROS parameters.  The trust-region options it
  // hard-wires (:157-186) are the library's defaults; of the string parameters only `ceres_loss_function`
  // changes the result (:60-94) and is honoured here.  The linear-solver / preconditioner / dogleg choices
  // select among Ceres back ends and have no counterpart: the GPU solver is Levenberg-Marquardt over a sparse
  // multifrontal Cholesky, the reference's default combination (:96-155).
  virtual void Configure(rclcpp_lifecycle::LifecycleNode::SharedPtr node)
  {
#ifdef KARTO_HIP_READ_ROS_PARAMETERS   // set by the plugin target inside slam_toolbox (needs the real rclcpp)
    if (!node->has_parameter("ceres_loss_function")) {
      node->declare_parameter("ceres_loss_function", rclcpp::ParameterValue(std::string("None")));
    }
    SetLossFunction(node->get_parameter("ceres_loss_function").as_string());
#else
    (void)node;
#endif
  }

  // "None" (squared loss) | "HuberLoss" | "CauchyLoss", scale 0.7 like ceres_solver.cpp:82-94; any other
  // string keeps the squared loss, as in the reference.
  void SetLossFunction(const std::string & name)
  {
    kh_spa_options o;
    kh_spa_options_default(&o);
    if (name == "HuberLoss") {o.loss_function = KH_LOSS_HUBER;} else if (name == "CauchyLoss") {o.loss_function = KH_LOSS_CAUCHY;}
    kh_spa_set_options(m_pHandle, &o);
  }

  virtual void Compute()
  {
    std::lock_guard<std::mutex> lock(m_Mutex);
    kh_spa_summary summary;
    if (kh_spa_compute(m_pHandle, &summary) != KH_OK) {return;}        // logged-and-returned in the reference
    int32_t n = 0;
    kh_spa_get_corrections(m_pHandle, &n, nullptr, nullptr);
    std::vector<int32_t> ids(n);
    std::vector<double> poses(3 * static_cast<size_t>(n));
    kh_spa_get_corrections(m_pHandle, &n, ids.data(), poses.data());
    m_Corrections.clear();
    m_Corrections.reserve(n);
    m_Graph.clear();
    for (int32_t i = 0; i < n; ++i) {
      m_Corrections.push_back(std::make_pair(ids[i], karto::Pose2(poses[3 * i], poses[3 * i + 1], poses[3 * i + 2])));
    }
  }

  virtual const karto::ScanSolver::IdPoseVector & GetCorrections() const {return m_Corrections;}

  virtual void Clear()
  {
    m_Corrections.clear();
    kh_spa_clear(m_pHandle);
  }

  virtual void Reset()
  {
    std::lock_guard<std::mutex> lock(m_Mutex);
    m_Corrections.clear();
    m_Ids.clear();
    kh_spa_reset(m_pHandle);
  }

  virtual void AddNode(karto::Vertex<karto::LocalizedRangeScan> * pVertex)
  {
    if (!pVertex) {return;}
    const karto::Pose2 pose = pVertex->GetObject()->GetCorrectedPose();
    const double p[3] = {pose.GetX(), pose.GetY(), pose.GetHeading()};
    std::lock_guard<std::mutex> lock(m_Mutex);
    const int id = pVertex->GetObject()->GetUniqueId();
    kh_spa_add_node(m_pHandle, id, p);
    m_Ids.push_back(id);
  }

  virtual void AddConstraint(karto::Edge<karto::LocalizedRangeScan> * pEdge)
  {
    std::lock_guard<std::mutex> lock(m_Mutex);
    if (!pEdge) {return;}
    const int a = pEdge->GetSource()->GetObject()->GetUniqueId();
    const int b = pEdge->GetTarget()->GetObject()->GetUniqueId();
    karto::LinkInfo * pLinkInfo = (karto::LinkInfo *)(pEdge->GetLabel());
    const karto::Pose2 diff = pLinkInfo->GetPoseDifference();
    const karto::Matrix3 & c = pLinkInfo->GetCovariance();
    const double z[3] = {diff.GetX(), diff.GetY(), diff.GetHeading()};
    double cov[9];
    for (int r = 0; r < 3; ++r) {for (int q = 0; q < 3; ++q) {cov[3 * r + q] = c(r, q);}}
    kh_spa_add_constraint(m_pHandle, a, b, z, cov);
  }

  virtual void RemoveNode(kt_int32s id)
  {
    std::lock_guard<std::mutex> lock(m_Mutex);
    kh_spa_remove_node(m_pHandle, id);
  }

  virtual void RemoveConstraint(kt_int32s sourceId, kt_int32s targetId)
  {
    std::lock_guard<std::mutex> lock(m_Mutex);
    kh_spa_remove_constraint(m_pHandle, sourceId, targetId);
  }

  virtual void ModifyNode(const int & unique_id, Eigen::Vector3d pose)
  {
    std::lock_guard<std::mutex> lock(m_Mutex);
    const double p[3] = {pose(0), pose(1), pose(2)};
    kh_spa_modify_node(m_pHandle, unique_id, p);
  }

  virtual void GetNodeOrientation(const int & unique_id, double & pose)
  {
    std::lock_guard<std::mutex> lock(m_Mutex);
    double p[3];
    if (kh_spa_get_node(m_pHandle, unique_id, p) == KH_OK) {pose = p[2];}
  }

  virtual std::unordered_map<int, Eigen::Vector3d> * getGraph()
  {
    std::lock_guard<std::mutex> lock(m_Mutex);
    m_Graph.clear();
    for (int id : m_Ids) {
      double p[3];
      if (kh_spa_get_node(m_pHandle, id, p) == KH_OK) {m_Graph[id] = Eigen::Vector3d(p[0], p[1], p[2]);}
    }
    return &m_Graph;
  }

private:
  kh_spa * m_pHandle;
  std::mutex m_Mutex;
  karto::ScanSolver::IdPoseVector m_Corrections;
  std::vector<int> m_Ids;
  std::unordered_map<int, Eigen::Vector3d> m_Graph;
};

}  // namespace karto_hip

#endif  // KARTO_HIP__KARTO_ADAPTOR_HPP_
