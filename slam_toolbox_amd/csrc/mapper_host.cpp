// ROS-free mapper front end over the GPU hot path: what karto::Mapper::Process does around the scan matcher and the
// solver plugin, so that a scan queue can be replayed end to end (BASELINE configs 1 and 5) without the reference's
// object model.  Everything numeric goes through the library's own entry points -- kh_matcher_* (sequential and loop
// matcher), kh_spa_* (solver plugin), kh_graph_* (candidate enumeration) -- and this file keeps the control flow and the
// small exact host arithmetic (poses, transforms, running-scan buffer, link bookkeeping) in the reference's operation
// order so that the run is comparable call by call with karto::Mapper driving the same queue.
//
// Reference: lib/karto_sdk/src/Mapper.cpp  Process :2679-2748, HasMovedEnough :3110-3142, ScanManager::AddRunningScan
// :183-206, MapperGraph::AddVertex :1418-1432, AddEdges :1434-1498, TryCloseLoop :1500-1561, LinkScans :1620-1639,
// LinkNearChains :1641-1663, LinkChainToScan :1665-1681, CorrectPoses :2012-2030; Karto.h LocalizedRangeScan::Update
// :5644-5704, GetSensorAt / GetCorrectedAt :5566-5586, Transform :2946-3041, Matrix3::FromAxisAngle :2482-2511.
//
// TryCloseLoop is where the GPU changes the SHAPE of the computation without changing its result: the reference matches
// one candidate chain after another; here all chains FindPossibleLoopClosure would return for the current poses are
// enumerated in one kernel, coarse-matched in one kh_matcher_match_batch, the ones passing the coarse gate fine-matched
// in a second batch, and the results consumed in the reference's order up to the first accepted closure.  CorrectPoses
// then moves every pose, so what was computed for later chains is discarded and the enumeration resumes behind the
// accepted chain with the new poses (closures are rare, the speculation almost always commits).
//
// Scope: one laser (mounted anywhere on the robot: kh_laser::offset_*, Karto.h:5566-5586), mapping mode (no localization buffer).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <memory>
#include <string>
#include <thread>
#include <atomic>
#include <vector>

#include "../../include/karto_hip.h"

namespace kh
{
void stream_synchronize(void * hip_stream);      // comm.cpp
int decay_scores(int32_t device, const kh_scan_box * reference, int32_t n, const kh_scan_box * candidates, const double * const * resident,
  const uint64_t * const * masks, int32_t n_scan, const kh_decay_params * params, int32_t * kept, double * iou, double * area_overlap,
  double * reading_overlap, double * scores);                                                                       // lifelong.hip
void set_pending_query_hook(std::function<void()> fn);       // matcher_seq.cpp (QueryHook, matcher_private.hpp)
void run_pending_query_hook();
int graph_swap(kh_graph * g, int32_t n_scans, std::vector<double> & ref_xy, std::vector<int32_t> & adj_ptr, std::vector<int32_t> & adj_idx);   // graph.hip

void set_error(const std::string & s);
void host_parallel_for(size_t n, const std::function<void(size_t)> & fn);
void host_parallel_for_wide(size_t n, const std::function<void(size_t)> & fn);

namespace
{
// cos and sin of ONE angle, the way the reference's Release build computes them: GCC (-O1 and up) merges a cos(a) / sin(a)
// pair into one sincos(a) call, and glibc's sincos is NOT bit-identical to its cos and sin everywhere (a = 0.11462314399891493:
// cos(a) = 0.9934379567501339, sincos(a) gives 0.993437956750134).  Every place where the reference takes both of the same
// angle goes through here, so that the library does not depend on whether ITS compiler merges the pair (clang does not).
static inline void ref_sincos(double a, double * s, double * c) {::sincos(a, s, c);}
constexpr double kTolerance = 1e-06;                  // KT_TOLERANCE, Math.h:41
constexpr double kPi = 3.14159265358979323846;        // Math.h:31
constexpr double k2Pi = 6.28318530717958647692;       // Math.h:32

double normalize_angle(double angle)                  // math::NormalizeAngle, Math.h:181-202
{
  while (angle < -kPi) {
    if (angle < -k2Pi) {angle += static_cast<uint32_t>(angle / -k2Pi) * k2Pi;} else {angle += k2Pi;}
  }
  while (angle > kPi) {
    if (angle > k2Pi) {angle -= static_cast<uint32_t>(angle / k2Pi) * k2Pi;} else {angle -= k2Pi;}
  }
  return angle;
}

struct Pose {double x = 0.0, y = 0.0, h = 0.0;};
inline bool same_pose(const Pose & a, const Pose & b) {return a.x == b.x && a.y == b.y && a.h == b.h;}   // Karto.h:2180-2183

struct Mat3
{
  double m[3][3];
  void identity() {std::memset(m, 0, sizeof(m)); m[0][0] = m[1][1] = m[2][2] = 1.0;}
  // Rodrigues' rotation about a unit axis, in the operation order of Matrix3::FromAxisAngle (Karto.h:2482-2511) -- the solver log
  // is compared with the reference's at 17 digits: diagonal a_i a_i (1 - c) + c; off-diagonal (a_i a_j)(1 - c) -+ a_k s, "+" where
  // (i, j, k) is an odd permutation
  void from_axis_angle(double ax, double ay, double az, double angle)
  {
    const double axis[3] = {ax, ay, az};
    double s, c;
    ref_sincos(angle, &s, &c);
    const double versine = 1.0 - c;
    for (int i = 0; i < 3; ++i) {
      for (int j = 0; j < 3; ++j) {
        if (i == j) {m[i][i] = axis[i] * axis[i] * versine + c; continue;}
        const int lo = i < j ? i : j, hi = i < j ? j : i;
        const double symmetric = axis[lo] * axis[hi] * versine, skew = axis[3 - i - j] * s;
        m[i][j] = ((j - i + 3) % 3 == 2) ? symmetric + skew : symmetric - skew;
      }
    }
  }
  Pose mul(const Pose & p) const                                          // Matrix3 * Pose2, Karto.h:2654-2666
  {
    Pose r;
    r.x = m[0][0] * p.x + m[0][1] * p.y + m[0][2] * p.h;
    r.y = m[1][0] * p.x + m[1][1] * p.y + m[1][2] * p.h;
    r.h = m[2][0] * p.x + m[2][1] * p.y + m[2][2] * p.h;
    return r;
  }
};

// karto::Transform(rPose1, rPose2).TransformPose(src), Karto.h:2946-3024
Pose transform_pose(const Pose & p1, const Pose & p2, const Pose & src)
{
  Mat3 rot;
  Pose t;
  if (same_pose(p1, p2)) {
    rot.identity();
  } else {
    rot.from_axis_angle(0, 0, 1, p2.h - p1.h);
    if (p1.x != 0.0 || p1.y != 0.0) {
      const Pose r = rot.mul(p1);
      t.x = p2.x - r.x; t.y = p2.y - r.y;
    } else {
      t.x = p2.x; t.y = p2.y;
    }
    t.h = p2.h - p1.h;
  }
  const Pose r = rot.mul(src);
  Pose out;
  out.x = t.x + r.x; out.y = t.y + r.y;
  out.h = normalize_angle(src.h + t.h);
  return out;
}

// One processed scan: what LocalizedRangeScan holds (zero mount offset: the sensor pose is the corrected pose with its
// heading normalised, GetSensorAt / GetCorrectedAt Karto.h:5566-5586)
struct MScan
{
  int32_t id = -1;
  double time = 0.0;
  Pose odometric, corrected;
  std::vector<double> ranges;
  std::vector<double> points;        // unfiltered point readings, x0 y0 x1 y1 ...
  std::vector<double> filtered;      // point readings with the range inside [minimum range, range threshold]
  std::vector<uint64_t> filter_mask; // bit i: reading i is one of `filtered` (the node-decay kernel reads `points` in HBM through it)
  double score = 1.0;                // Vertex::GetScore (Mapper.h: vertices start at 1.0)
  double barycenter[2] = {0.0, 0.0};
  double bbox[4] = {0.0, 0.0, 0.0, 0.0};   // min x, min y, max x, max y of the sensor position and the filtered readings
  int32_t n_filtered = 0;
  // the unfiltered readings once more, in HBM: as a base scan of a match the scan is read where it lies (kh_scan::
  // device_points_xy).  Uploaded on first use and again after the pose has moved (update_scan marks it stale).
  static constexpr int kMaxDeviceSlots = 16;
  double * d_points[kMaxDeviceSlots] = {};       // one copy per distinct device of the mapper (slot 0 = the mapper's own device)
  uint16_t d_fresh = 0;                          // bit k: the copy in slot k holds the current points
  Pose sensor;                                   // GetSensorAt(corrected), refreshed by update_scan (every write of `corrected` is followed by one)
  Pose sensor_pose() const {return sensor;}
  MScan() = default;
  MScan(const MScan &) = delete;
  MScan & operator=(const MScan &) = delete;
  // d_points is a slot of the mapper's device slabs (take_slot / remove_node / kh_mapper_destroy), not owned here
};

struct Laser
{
  int32_t n = 0;
  double min_angle = 0, ang_res = 0, min_range = 0, max_range = 0, range_threshold = 0;
  Pose offset;                                   // LaserRangeFinder::GetOffsetPose: the sensor in the robot's frame
};

// LocalizedRangeScan::GetSensorAt (Karto.h:5566-5569): Transform(robot pose).TransformPose(offset pose)
Pose sensor_at(const Laser & L, const Pose & robot) {return transform_pose(Pose(), robot, L.offset);}

// LocalizedRangeScan::GetCorrectedAt (Karto.h:5576-5588): the robot pose that puts the sensor at `sensor`
Pose corrected_at(const Laser & L, const Pose & sensor)
{
  const double offset_length = std::sqrt(L.offset.x * L.offset.x + L.offset.y * L.offset.y);       // Vector2::Length
  const double offset_heading = L.offset.h;
  const double angle_offset = std::atan2(L.offset.y, L.offset.x);
  const double heading = normalize_angle(sensor.h);
  double sin_w, cos_w;
  ref_sincos(heading + angle_offset - offset_heading, &sin_w, &cos_w);
  Pose robot;                                      // Pose2::operator-: positions subtract, the heading difference is normalised
  robot.x = sensor.x - offset_length * cos_w;
  robot.y = sensor.y - offset_length * sin_w;
  robot.h = normalize_angle(sensor.h - offset_heading);
  return robot;
}

// LocalizedRangeScan::Update, Karto.h:5644-5704
void update_scan(MScan & s, const Laser & L)
{
  s.sensor = sensor_at(L, s.corrected);
  const Pose sp = s.sensor;
  s.points.resize(2 * static_cast<size_t>(L.n));
  s.d_fresh = 0;
  s.filtered.clear();
  s.filter_mask.assign((static_cast<size_t>(L.n) + 63) / 64, 0);
  double sum_x = 0.0, sum_y = 0.0;
  int32_t n_filtered = 0;
  double bb[4] = {sp.x, sp.y, sp.x, sp.y};
  for (int32_t i = 0; i < L.n; ++i) {
    const double r = s.ranges[i];
    const double angle = sp.h + L.min_angle + static_cast<uint32_t>(i) * L.ang_res;
    double sin_a, cos_a;
    ref_sincos(angle, &sin_a, &cos_a);
    const double px = sp.x + (r * cos_a);
    const double py = sp.y + (r * sin_a);
    s.points[2 * i] = px; s.points[2 * i + 1] = py;
    if (r >= L.min_range && r <= L.range_threshold) {          // math::InRange
      sum_x += px; sum_y += py; ++n_filtered;
      s.filtered.push_back(px); s.filtered.push_back(py);
      s.filter_mask[static_cast<size_t>(i) >> 6] |= 1ull << (i & 63);
      bb[0] = std::min(bb[0], px); bb[1] = std::min(bb[1], py); bb[2] = std::max(bb[2], px); bb[3] = std::max(bb[3], py);
    }
  }
  const double n_points = static_cast<double>(n_filtered);
  if (n_points != 0.0) {
    s.barycenter[0] = sum_x / n_points; s.barycenter[1] = sum_y / n_points;
  } else {
    s.barycenter[0] = sp.x; s.barycenter[1] = sp.y;
  }
  s.n_filtered = n_filtered;
  std::copy(bb, bb + 4, s.bbox);
}

}  // namespace
}  // namespace kh

using namespace kh;

struct kh_mapper
{
  kh_mapper_params p;
  Laser laser;
  int32_t device = 0, max_candidates = 64;
  kh_matcher * seq = nullptr;                            // = member 0 of seq_group
  kh_matcher * loop = nullptr;                           // = member 0 of loop_group
  // candidate batches (loop closure, near chains) are dealt over the members of these groups: one member per entry of the
  // device list the mapper was created on (kh_mapper_create_on_devices), each with its own scan copies
  kh_matcher_group * seq_group = nullptr;
  kh_matcher_group * loop_group = nullptr;
  std::vector<int32_t> member_device;                    // device of member k
  std::vector<int32_t> member_slot;                      // scan-copy slot of member k (members on one device share a slot)
  int32_t n_slots = 1;
  kh_spa * solver = nullptr;
  kh_graph * graph = nullptr;
  std::vector<std::unique_ptr<MScan>> scans;             // processed scans, index = state id = unique id; null once removed
  std::vector<int32_t> alive;                            // ids still in the scan map, ascending: the graph store's scan list
  std::vector<int32_t> compact_of;                       // id -> position in `alive`, -1 when removed
  std::vector<double> sync_xy; std::vector<int32_t> sync_ptr, sync_idx;      // sync_graph's scratch (swapped with the store's arrays)
  // KH_MAPPER_TIMING=1 (measurement aid): wall time per piece of the host code, printed by kh_mapper_destroy
  double prof_ms[12] = {0}; long prof_n[12] = {0};
  bool lifelong = false;
  kh_decay_params decay;
  std::vector<int32_t> running;
  int32_t last = -1;
  // set when Process() fails after its scan was committed to the scan list, the graph store and the solver: the
  // `id - 1 = previous scan` bookkeeping no longer matches, so every later Process() refuses instead of mis-linking
  bool failed = false;
  std::vector<std::vector<int32_t>> adj;                 // Vertex::GetAdjacentVertices order (Mapper.h:338-361)
  std::vector<std::vector<int32_t>> out_edges;           // targets of the edges whose SOURCE is the vertex (AddEdge's duplicate test)
  int64_t n_edges = 0;
  bool graph_dirty = true;
  FILE * log = nullptr;
  kh_mapper_stats stats;
  // device copies of the scans' readings: slots of 2 * laser.n doubles carved from slabs of 256 (one hipMalloc per 256
  // scans instead of one per scan), recycled when a node is removed
  std::vector<double *> d_slabs[MScan::kMaxDeviceSlots], d_free_slots[MScan::kMaxDeviceSlots];
  std::vector<int32_t> slot_device;                      // device of scan-copy slot q
};

namespace kh
{
namespace
{

kh_scan as_kh_scan(const MScan & s)
{
  kh_scan k;
  k.n = static_cast<int32_t>(s.ranges.size());
  k.ranges = s.ranges.data();
  k.points_xy = s.points.data();
  const Pose sp = s.sensor_pose();
  k.sensor_pose[0] = sp.x; k.sensor_pose[1] = sp.y; k.sensor_pose[2] = sp.h;
  k.device_points_xy = nullptr;
  return k;
}

// the scan's readings resident in scan-copy slot `slot` (= on that slot's device): the device address, or NULL when the
// copy could not be made (the match call then uploads the scan itself)
// on_stream (nullptr = none): the upload is queued on that HIP stream instead of waited for -- for a scan whose next reader is a
// kernel of the same stream (the sequential matcher's: a synchronous copy of 17 KB out of pageable memory costs 25 us of the
// caller's time, once per accepted scan and again for every scan a loop closure moved)
const double * resident_points(kh_mapper * m, MScan & s, int slot, void * on_stream = nullptr)
{
  const int64_t bytes = static_cast<int64_t>(sizeof(double)) * static_cast<int64_t>(s.points.size());
  if (bytes <= 0 || s.points.size() != 2 * static_cast<size_t>(m->laser.n)) {return nullptr;}
  if (!s.d_points[slot]) {
    if (m->d_free_slots[slot].empty()) {
      constexpr int kSlabScans = 256;
      void * p = nullptr;
      if (kh_device_malloc(m->slot_device[slot], bytes * kSlabScans, &p) == KH_OK) {
        m->d_slabs[slot].push_back(static_cast<double *>(p));
        for (int k = kSlabScans - 1; k >= 0; --k) {m->d_free_slots[slot].push_back(static_cast<double *>(p) + static_cast<size_t>(k) * s.points.size());}
      }
    }
    if (!m->d_free_slots[slot].empty()) {
      s.d_points[slot] = m->d_free_slots[slot].back(); m->d_free_slots[slot].pop_back();
      s.d_fresh &= static_cast<uint16_t>(~(1u << slot));
    }
  }
  if (s.d_points[slot] && !(s.d_fresh & (1u << slot)) &&
    (on_stream ? kh_device_upload_on(s.d_points[slot], s.points.data(), bytes, on_stream) : kh_device_upload(s.d_points[slot], s.points.data(), bytes)) == KH_OK) {
    s.d_fresh |= static_cast<uint16_t>(1u << slot);
  }
  return (s.d_points[slot] && (s.d_fresh & (1u << slot))) ? s.d_points[slot] : nullptr;
}

// the scan as a BASE scan of a match on the mapper's own device: its readings resident there
kh_scan as_base_scan(kh_mapper * m, MScan & s, void * on_stream = nullptr)
{
  kh_scan k = as_kh_scan(s);
  k.device_points_xy = resident_points(m, s, 0, on_stream);
  return k;
}

void reference_xy(const kh_mapper * m, const MScan & s, double xy[2])     // GetReferencePose(useScanBarycenter)
{
  if (m->p.use_scan_barycenter) {xy[0] = s.barycenter[0]; xy[1] = s.barycenter[1];} else {const Pose sp = s.sensor_pose(); xy[0] = sp.x; xy[1] = sp.y;}
}

// the graph store the enumeration kernels and the near-chain walks read: reference positions + adjacency of the scans
// still in the map, in id order (a removed scan is a NULL entry the reference's walks skip)
struct ProfScope
{
  kh_mapper * m; int k; std::chrono::steady_clock::time_point t0;
  ProfScope(kh_mapper * m_, int k_) : m(m_), k(k_), t0(std::chrono::steady_clock::now()) {}
  ~ProfScope() {m->prof_ms[k] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); m->prof_n[k] += 1;}
};
static const char * const kProfNames[12] = {"sync_graph", "near_linked", "decay_boxes", "decay_scores", "remove_node", "update_scan(new)", "add_to_graph",
                                           "loop_enumeration", "link_near_chains", "", "", ""};

int sync_graph(kh_mapper * m)
{
  ProfScope prof(m, 0);
  m->alive.clear();
  m->compact_of.assign(m->scans.size(), -1);
  for (size_t i = 0; i < m->scans.size(); ++i) {
    if (m->scans[i]) {m->compact_of[i] = static_cast<int32_t>(m->alive.size()); m->alive.push_back(static_cast<int32_t>(i));}
  }
  const size_t n = m->alive.size();
  // (scratch kept by the mapper and swapped with the store's arrays: no allocation, no copy -- see kh::graph_swap)
  std::vector<double> & xy = m->sync_xy;
  std::vector<int32_t> & ptr = m->sync_ptr, & idx = m->sync_idx;
  xy.resize(2 * n); ptr.resize(n + 1); idx.clear();
  ptr[0] = 0;
  for (size_t c = 0; c < n; ++c) {
    reference_xy(m, *m->scans[m->alive[c]], &xy[2 * c]);
    ptr[c + 1] = ptr[c] + static_cast<int32_t>(m->adj[m->alive[c]].size());
  }
  idx.reserve(static_cast<size_t>(ptr[n]));
  for (size_t c = 0; c < n; ++c) {
    for (int32_t w : m->adj[m->alive[c]]) {idx.push_back(m->compact_of[w]);}
  }
  int rc = kh::graph_swap(m->graph, static_cast<int32_t>(n), xy, ptr, idx);
  if (rc) {return rc;}
  // the reference bounds its candidate walks by the scan map's SIZE, in id space (Mapper.cpp:1974-1976, 1751-1756)
  const int32_t n_visit = static_cast<int32_t>(std::lower_bound(m->alive.begin(), m->alive.end(), static_cast<int32_t>(n)) - m->alive.begin());
  rc = kh_graph_set_scan_limit(m->graph, n_visit);
  if (rc == KH_OK) {m->graph_dirty = false;}
  return rc;
}

// SetSensorPose (Karto.h:5552-5557): corrected = GetCorrectedAt(pose), then Update
void set_sensor_pose(kh_mapper * m, MScan & s, const double pose[3])
{
  Pose sensor; sensor.x = pose[0]; sensor.y = pose[1]; sensor.h = pose[2];
  s.corrected = corrected_at(m->laser, sensor);
  update_scan(s, m->laser);
  if (s.id < 0) {return;}               // not in the graph yet (the match of a new scan): AddScan appends it where it stands
  if (!m->graph_dirty && s.id < static_cast<int32_t>(m->compact_of.size()) && m->compact_of[s.id] >= 0) {
    double xy[2];
    reference_xy(m, s, xy);
    if (kh_graph_set_position(m->graph, m->compact_of[s.id], xy) != KH_OK) {m->graph_dirty = true;}
  } else {
    m->graph_dirty = true;
  }
}

// MapperGraph::LinkScans (Mapper.cpp:1620-1639) incl. AddEdge's "edge already exists" test (:1585-1618)
int link_scans(kh_mapper * m, int32_t from, int32_t to, const double mean[3], const double cov[9])
{
  for (int32_t t : m->out_edges[from]) {if (t == to) {return KH_OK;}}     // not a new edge: nothing is attached
  m->out_edges[from].push_back(to);
  m->adj[from].push_back(to);
  m->adj[to].push_back(from);
  ++m->n_edges;
  // the store follows edit by edit while nothing was removed or re-posed since it was built (sync_graph rebuilds otherwise)
  if (!m->graph_dirty && kh_graph_add_edge(m->graph, m->compact_of[from], m->compact_of[to]) != KH_OK) {m->graph_dirty = true;}
  // LinkInfo(pFromScan->GetCorrectedPose(), pToScan->GetCorrectedAt(rMean), rCovariance)
  const MScan & f = *m->scans[from];
  const double pose1[3] = {f.corrected.x, f.corrected.y, f.corrected.h};
  Pose mean_sensor; mean_sensor.x = mean[0]; mean_sensor.y = mean[1]; mean_sensor.h = mean[2];
  const Pose to_robot = corrected_at(m->laser, mean_sensor);
  const double pose2[3] = {to_robot.x, to_robot.y, to_robot.h};
  double diff[3], cov_out[9];
  int rc = kh_link_info(pose1, pose2, cov, diff, cov_out);
  if (rc) {return rc;}
  if (m->log) {
    std::fprintf(m->log, "C %d %d %.17g %.17g %.17g", from, to, diff[0], diff[1], diff[2]);
    for (int k = 0; k < 9; ++k) {std::fprintf(m->log, " %.17g", cov_out[k]);}
    std::fprintf(m->log, "\n");
  }
  rc = kh_spa_add_constraint(m->solver, from, to, diff, cov_out);
  // the plugin logs and carries on when it cannot add a constraint (ceres_solver.cpp:354-361)
  return (rc == KH_OK || rc == KH_ERR_NOT_FOUND || rc == KH_ERR_INVALID_ARG) ? KH_OK : rc;
}

// MapperGraph::LinkChainToScan (Mapper.cpp:1665-1681)
int link_chain_to_scan(kh_mapper * m, const std::vector<int32_t> & chain, int32_t scan, const double mean[3], const double cov[9])
{
  double pose[2];
  reference_xy(m, *m->scans[scan], pose);
  // GetClosestScanToPose (Mapper.cpp:1563-1582)
  int32_t closest = -1;
  double best = 1.7976931348623157e308;
  for (int32_t c : chain) {
    double xy[2];
    reference_xy(m, *m->scans[c], xy);
    const double dx = pose[0] - xy[0], dy = pose[1] - xy[1];
    const double d = dx * dx + dy * dy;
    if (d < best) {best = d; closest = c;}
  }
  if (closest < 0) {return KH_OK;}
  double cxy[2];
  reference_xy(m, *m->scans[closest], cxy);
  const double dx = pose[0] - cxy[0], dy = pose[1] - cxy[1];
  const double squaredDistance = dx * dx + dy * dy;
  if (squaredDistance < m->p.link_scan_maximum_distance * m->p.link_scan_maximum_distance + kTolerance) {
    return link_scans(m, closest, scan, mean, cov);
  }
  return KH_OK;
}

// MapperGraph::CorrectPoses (Mapper.cpp:2012-2030)
int correct_poses(kh_mapper * m)
{
  const auto t0 = std::chrono::steady_clock::now();
  kh_spa_summary sum;
  const int rc = kh_spa_compute(m->solver, &sum);
  const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  m->stats.solver_ms += ms; m->stats.loop_closures += 1;
  if (rc != KH_OK && rc != KH_ERR_SOLVER && rc != KH_ERR_NOT_FOUND) {return rc;}
  int32_t n = 0;
  kh_spa_get_corrections(m->solver, &n, nullptr, nullptr);
  std::vector<int32_t> ids(static_cast<size_t>(n));
  std::vector<double> poses(3 * static_cast<size_t>(n));
  if (n) {kh_spa_get_corrections(m->solver, &n, ids.data(), poses.data());}
  if (m->log) {
    std::fprintf(m->log, "X %d %.6f\n", n, ms);
    for (int32_t k = 0; k < n; ++k) {std::fprintf(m->log, "P %d %.17g %.17g %.17g\n", ids[k], poses[3 * k], poses[3 * k + 1], poses[3 * k + 2]);}
  }
  // SetCorrectedPoseAndUpdate of every scan: N x P cos / sin in libm (the reference does the same, serially)
  const auto t1 = std::chrono::steady_clock::now();
  auto one = [&](size_t k) {
    const int32_t id = ids[k];
    if (id < 0 || id >= static_cast<int32_t>(m->scans.size()) || !m->scans[id]) {return;}
    MScan & s = *m->scans[id];
    // (a pose the solve left bit for bit where it was -- a component the closure did not touch and whose own residuals are at
    // rest -- re-projects to the same readings: nothing to do, and the device copy stays valid)
    if (std::memcmp(&s.corrected.x, &poses[3 * k], sizeof(double)) == 0 && std::memcmp(&s.corrected.y, &poses[3 * k + 1], sizeof(double)) == 0 &&
      std::memcmp(&s.corrected.h, &poses[3 * k + 2], sizeof(double)) == 0) {return;}
    s.corrected.x = poses[3 * k]; s.corrected.y = poses[3 * k + 1]; s.corrected.h = poses[3 * k + 2];
    update_scan(s, m->laser);
  };
  // thousands of scans x 1081 sincos: wider than the matcher's worker pool (32 threads suit its sub-millisecond bursts;
  // this is milliseconds of uniform work -- N x 1081 libm calls that have to stay on the host for bit-exactness,
  // Karto.h:5488-5493, 5644-5704 -- so it goes to the wide pool in chunks of 32 scans)
  if (n >= 2048) {
    const size_t chunks = (static_cast<size_t>(n) + 31) / 32;
    host_parallel_for_wide(chunks, [&](size_t c) {
      for (size_t k = 32 * c; k < std::min(static_cast<size_t>(n), 32 * c + 32); ++k) {one(k);}
    });
  } else {
    host_parallel_for(static_cast<size_t>(n), one);
  }
  m->stats.update_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count();
  m->graph_dirty = true;
  if (m->log) {std::fprintf(m->log, "K\n");}
  return kh_spa_clear(m->solver);
}

struct MatchOut {double response; double mean[3]; double cov[9];};

// n independent MatchScan calls (query i against chain i) on the members of `group` (candidate i on member i % members,
// each member on the scan copies of its own device), results in candidate order
int match_chains(kh_mapper * m, kh_matcher_group * group, const std::vector<kh_scan> & queries, const std::vector<std::vector<int32_t>> & chains,
  bool penalize, bool refine, std::vector<MatchOut> & out)
{
  const size_t n = chains.size();
  out.assign(n, MatchOut());
  if (n == 0) {return KH_OK;}
  const int32_t nm = kh_matcher_group_size(group);
  std::vector<kh_scan> base;
  std::vector<int32_t> begin(n + 1, 0);
  std::vector<const double *> table;
  for (size_t i = 0; i < n; ++i) {
    const int32_t member = static_cast<int32_t>(i % static_cast<size_t>(nm));
    for (int32_t c : chains[i]) {
      MScan & s = *m->scans[c];
      base.push_back(as_kh_scan(s));
      const size_t row = table.size();
      table.resize(row + static_cast<size_t>(nm), nullptr);
      table[row + member] = resident_points(m, s, m->member_slot[member]);       // only the member that will read it
    }
    begin[i + 1] = static_cast<int32_t>(base.size());
  }
  std::vector<double> means(3 * n), covs(9 * n), resp(n);
  std::vector<int32_t> status(n, 0);
  const auto t0 = std::chrono::steady_clock::now();
  const int rc = kh_matcher_group_match_batch(group, static_cast<int32_t>(n), queries.data(), base.data(), begin.data(), table.data(),
      penalize ? 1 : 0, refine ? 1 : 0, means.data(), covs.data(), resp.data(), status.data());
  m->stats.match_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  m->stats.matches += static_cast<int64_t>(n);
  if (rc) {return rc;}
  for (size_t i = 0; i < n; ++i) {
    if (status[i] != KH_OK) {return status[i];}             // the reference throws (Mapper.cpp:786-796, 828)
    out[i].response = resp[i];
    std::copy(means.begin() + 3 * i, means.begin() + 3 * i + 3, out[i].mean);
    std::copy(covs.begin() + 9 * i, covs.begin() + 9 * i + 9, out[i].cov);
  }
  return KH_OK;
}

// scan ids of the run [first, last] of the graph store's scan list
std::vector<int32_t> run_of(const kh_mapper * m, int32_t first, int32_t last)
{
  std::vector<int32_t> v;
  for (int32_t i = first; i <= last; ++i) {v.push_back(m->alive[i]);}
  return v;
}

// MapperGraph::TryCloseLoop (Mapper.cpp:1500-1561), speculative batches
int try_close_loop(kh_mapper * m, int32_t scan_id, bool & closed)
{
  closed = false;
  int32_t start_id = 0;                            // rStartNum, in id space
  const int32_t n_scans = static_cast<int32_t>(m->scans.size());
  while (start_id < n_scans) {
    if (m->graph_dirty) {const int rc = sync_graph(m); if (rc) {return rc;}}
    // positions in the graph store's scan list (= the scans still alive, in id order)
    const int32_t query = m->compact_of[scan_id];
    int32_t start = static_cast<int32_t>(std::lower_bound(m->alive.begin(), m->alive.end(), start_id) - m->alive.begin());
    // every chain successive FindPossibleLoopClosure calls would return from `start` on, for the CURRENT poses
    std::vector<int32_t> chain_begin(2, 0), flat(2 * static_cast<size_t>(m->max_candidates));
    int32_t n_chains = 0;
    int rc;
    {ProfScope prof(m, 7);
    rc = kh_graph_find_loop_candidates_from(m->graph, 1, &query, &start, m->p.loop_search_maximum_distance,
        m->p.loop_match_minimum_chain_size, chain_begin.data(), flat.data(), m->max_candidates, &n_chains);
    }
    if (rc) {return rc;}
    if (n_chains > m->max_candidates) {
      flat.resize(2 * static_cast<size_t>(n_chains));
      rc = kh_graph_find_loop_candidates_from(m->graph, 1, &query, &start, m->p.loop_search_maximum_distance,
          m->p.loop_match_minimum_chain_size, chain_begin.data(), flat.data(), n_chains, &n_chains);
      if (rc) {return rc;}
    }
    if (n_chains == 0) {break;}
    m->stats.loop_candidates += n_chains;
    std::vector<std::vector<int32_t>> chains(static_cast<size_t>(n_chains));
    for (int32_t c = 0; c < n_chains; ++c) {chains[c] = run_of(m, flat[2 * c], flat[2 * c + 1]);}
    MScan & scan = *m->scans[scan_id];
    // coarse: m_pLoopScanMatcher->MatchScan(pScan, candidateChain, bestPose, covariance, false, false), all chains at once
    std::vector<MatchOut> coarse;
    rc = match_chains(m, m->loop_group, std::vector<kh_scan>(chains.size(), as_kh_scan(scan)), chains, false, false, coarse);
    if (rc) {return rc;}
    std::vector<int32_t> passing;
    for (int32_t c = 0; c < n_chains; ++c) {
      if (coarse[c].response > m->p.loop_match_minimum_response_coarse &&
        coarse[c].cov[0] < m->p.loop_match_maximum_variance_coarse && coarse[c].cov[4] < m->p.loop_match_maximum_variance_coarse)
      {
        passing.push_back(c);
      }
    }
    // fine: tmpScan at the coarse pose against the same chain on the sequential matcher (doPenalize false), again one batch
    std::vector<std::unique_ptr<MScan>> tmp;
    std::vector<kh_scan> fine_queries;
    std::vector<std::vector<int32_t>> fine_chains;
    for (int32_t c : passing) {
      std::unique_ptr<MScan> t(new MScan());
      t->ranges = scan.ranges; t->corrected = scan.corrected;
      Pose best; best.x = coarse[c].mean[0]; best.y = coarse[c].mean[1]; best.h = coarse[c].mean[2];
      t->corrected = corrected_at(m->laser, best);                 // tmpScan.SetSensorPose(bestPose), Mapper.cpp:1533
      update_scan(*t, m->laser);
      fine_queries.push_back(as_kh_scan(*t));
      fine_chains.push_back(chains[c]);
      tmp.push_back(std::move(t));
    }
    std::vector<MatchOut> fine;
    rc = match_chains(m, m->seq_group, fine_queries, fine_chains, false, true, fine);
    if (rc) {return rc;}
    // consume in the reference's order up to the first accepted closure
    int32_t accepted = -1;
    for (size_t i = 0; i < passing.size(); ++i) {
      if (!(fine[i].response < m->p.loop_match_minimum_response_fine)) {accepted = static_cast<int32_t>(i); break;}
    }
    if (accepted < 0) {break;}                       // every chain was looked at with the poses it would have seen
    const int32_t c = passing[accepted];
    set_sensor_pose(m, scan, fine[accepted].mean);
    rc = link_chain_to_scan(m, chains[c], scan_id, fine[accepted].mean, fine[accepted].cov);
    if (rc) {return rc;}
    rc = correct_poses(m);
    if (rc) {return rc;}
    closed = true;
    // FindPossibleLoopClosure returned this chain at its terminating scan (rStartNum stays there): resume behind it
    start_id = m->alive[flat[2 * c + 1]] + 1;
    m->stats.speculation_discarded += n_chains - (c + 1);
  }
  return KH_OK;
}

// Mapper::RemoveNodeFromGraph (Mapper.cpp:2964-3021) + MapperSensorManager::RemoveScan (:208-218)
int remove_node(kh_mapper * m, int32_t id)
{
  ProfScope prof(m, 4);
  if (id < 0 || id >= static_cast<int32_t>(m->scans.size()) || !m->scans[id]) {
    set_error("RemoveNode: Failed to find node matching id");
    return KH_ERR_NOT_FOUND;
  }
  // 1) the edges leave the adjacent vertices, the graph and the optimizer
  const std::vector<int32_t> neighbours = m->adj[id];
  for (int32_t a : neighbours) {
    auto pos = std::find(m->adj[a].begin(), m->adj[a].end(), id);
    if (pos == m->adj[a].end()) {continue;}                    // "Failed to find any edge in adj. vertex"
    m->adj[a].erase(pos);
    int32_t source = id, target = a;
    auto out = std::find(m->out_edges[a].begin(), m->out_edges[a].end(), id);
    if (out != m->out_edges[a].end()) {source = a; target = id; m->out_edges[a].erase(out);}
    if (m->log) {std::fprintf(m->log, "E %d %d\n", source, target);}
    const int rc = kh_spa_remove_constraint(m->solver, source, target);
    if (rc != KH_OK && rc != KH_ERR_NOT_FOUND) {return rc;}
    --m->n_edges;
  }
  // 2) the vertex leaves the optimizer, 3) the graph and the scan map
  if (m->log) {std::fprintf(m->log, "D %d\n", id);}
  const int rc = kh_spa_remove_node(m->solver, id);
  if (rc != KH_OK && rc != KH_ERR_NOT_FOUND) {return rc;}
  m->adj[id].clear(); m->out_edges[id].clear();
  for (int q = 0; q < m->n_slots; ++q) {
    if (m->scans[id]->d_points[q]) {m->d_free_slots[q].push_back(m->scans[id]->d_points[q]);}
  }
  m->scans[id].reset();
  m->graph_dirty = true;
  m->stats.nodes_removed += 1;
  return KH_OK;
}

kh_scan_box box_of(const kh_mapper * m, const MScan & s)
{
  kh_scan_box b;
  std::memset(&b, 0, sizeof(b));
  b.barycenter[0] = s.barycenter[0]; b.barycenter[1] = s.barycenter[1];
  b.bbox_size[0] = s.bbox[2] - s.bbox[0]; b.bbox_size[1] = s.bbox[3] - s.bbox[1];     // BoundingBox2::GetSize
  b.unique_id = s.id; b.n_edges = static_cast<int32_t>(m->adj[s.id].size()); b.score = s.score;
  b.n_points = static_cast<int32_t>(s.filtered.size() / 2); b.points_xy = s.filtered.data();
  return b;
}

// LifelongSlamToolbox::evaluateNodeDepreciation (slam_toolbox_lifelong.cpp:149-178), lifelong_search_use_tree false
int lifelong_step(kh_mapper * m, int32_t id)
{
  const auto t0 = std::chrono::steady_clock::now();
  const MScan & s = *m->scans[id];
  const double w = s.bbox[2] - s.bbox[0], h = s.bbox[3] - s.bbox[1];
  const double radius = std::sqrt(w * w + h * h) / 2.0;
  if (m->graph_dirty) {const int rc = sync_graph(m); if (rc) {return rc;}}
  std::vector<int32_t> near(64);
  int32_t n = 0;
  int rc;
  {
  ProfScope prof(m, 1);
  rc = kh_graph_find_near_linked(m->graph, m->compact_of[id], radius, near.data(), static_cast<int32_t>(near.size()), &n);
  }
  if (rc) {return rc;}
  if (n > static_cast<int32_t>(near.size())) {
    near.resize(static_cast<size_t>(n));
    rc = kh_graph_find_near_linked(m->graph, m->compact_of[id], radius, near.data(), n, &n);
    if (rc) {return rc;}
  }
  near.resize(static_cast<size_t>(n));
  for (int32_t & c : near) {c = m->alive[c];}                     // graph store positions -> scan ids
  const kh_scan_box ref = box_of(m, s);
  std::vector<kh_scan_box> cands;
  // the candidates' readings where the matcher left them in HBM (a copy a pose update made stale is refreshed first); a scan
  // without a device copy (slot allocation failed) sends the whole call down the packed form
  std::vector<const double *> resident;
  std::vector<const uint64_t *> masks;
  bool all_resident = m->laser.n <= 4096;
  {ProfScope prof(m, 2);
  for (int32_t c : near) {
    MScan & cs = *m->scans[c];
    cands.push_back(box_of(m, cs));
    // (a candidate inside the scan buffer, and vertices 0 and 1, keep their score whatever their readings say, :204-207: the mapper
    // does not read the overlap metrics, so their readings are not fetched -- the new scan itself, always a candidate, has no copy yet)
    const bool score_kept = s.id - cs.id < m->decay.scan_buffer_size || cs.id == 0 || cs.id == 1;
    const double * d = (all_resident && !score_kept) ? resident_points(m, cs, 0) : nullptr;
    if (!d && !score_kept) {all_resident = false;}
    resident.push_back(d); masks.push_back(cs.filter_mask.data());
  }
  }
  std::vector<int32_t> kept(near.size(), 0);
  std::vector<double> scores(near.size(), 0.0);
  {ProfScope prof(m, 3);
  rc = kh::decay_scores(m->device, &ref, n, cands.data(), all_resident ? resident.data() : nullptr, all_resident ? masks.data() : nullptr,
      all_resident ? m->laser.n : 0, &m->decay, kept.data(), nullptr, nullptr, nullptr, scores.data());
  }
  if (rc) {return rc;}
  for (size_t k = 0; k < near.size(); ++k) {
    if (!kept[k]) {continue;}
    if (scores[k] < m->decay.removal_score) {
      rc = remove_node(m, near[k]);
      if (rc) {return rc;}
    } else {
      m->scans[near[k]]->score = scores[k];
    }
  }
  m->stats.lifelong_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  return KH_OK;
}

}  // namespace
}  // namespace kh

extern "C" {

void kh_mapper_params_default(kh_mapper_params * p)
{
  if (!p) {return;}
  // config/mapper_params_offline.yaml:31-66 (the offline / sync launches)
  p->use_scan_matching = 1; p->use_scan_barycenter = 1;
  p->minimum_time_interval = 3600.0;                 // Mapper.cpp:2108-2118 (not in the yaml)
  p->minimum_travel_distance = 0.5; p->minimum_travel_heading = 0.5;
  p->scan_buffer_size = 10; p->scan_buffer_maximum_scan_distance = 10.0;
  p->link_match_minimum_response_fine = 0.1; p->link_scan_maximum_distance = 1.5;
  p->loop_search_maximum_distance = 3.0; p->do_loop_closing = 1;
  p->loop_match_minimum_chain_size = 10;
  p->loop_match_maximum_variance_coarse = 3.0 * 3.0;  // the setter squares it (Mapper.cpp:2512-2515)
  p->loop_match_minimum_response_coarse = 0.35; p->loop_match_minimum_response_fine = 0.45;
  p->correlation_search_space_dimension = 0.5; p->correlation_search_space_resolution = 0.01;
  p->correlation_search_space_smear_deviation = 0.1;
  p->loop_search_space_dimension = 8.0; p->loop_search_space_resolution = 0.05; p->loop_search_space_smear_deviation = 0.03;
  p->match.coarse_search_angle_offset = 0.349; p->match.coarse_angle_resolution = 0.0349;
  p->match.fine_search_angle_offset = 0.00349; p->match.use_response_expansion = 1;
  p->match.distance_variance_penalty = 0.5 * 0.5; p->match.minimum_distance_penalty = 0.5;
  p->match.angle_variance_penalty = 1.0 * 1.0; p->match.minimum_angle_penalty = 0.9;
}

int kh_mapper_create_on_devices(const kh_mapper_params * params, const kh_laser * laser, const int32_t * devices, int32_t n_devices,
  int32_t max_candidates, kh_mapper ** out)
{
  if (!out || !params || !laser || laser->n_beams <= 0 || max_candidates < 1 || !devices || n_devices < 1) {return KH_ERR_INVALID_ARG;}
  *out = nullptr;
  std::unique_ptr<kh_mapper> m(new kh_mapper());
  m->p = *params; m->device = devices[0]; m->max_candidates = max_candidates;
  m->laser.n = laser->n_beams; m->laser.min_angle = laser->minimum_angle; m->laser.ang_res = laser->angular_resolution;
  m->laser.min_range = laser->minimum_range; m->laser.max_range = laser->maximum_range; m->laser.range_threshold = laser->range_threshold;
  m->laser.offset.x = laser->offset_x; m->laser.offset.y = laser->offset_y; m->laser.offset.h = laser->offset_heading;
  std::memset(&m->stats, 0, sizeof(m->stats));
  auto fail = [&](int rc) {kh_mapper_destroy(m.release()); return rc;};
  // scan copies: one slot per DISTINCT device (members that share a device share the copies)
  for (int32_t k = 0; k < n_devices; ++k) {
    int32_t slot = -1;
    for (size_t q = 0; q < m->slot_device.size(); ++q) {if (m->slot_device[q] == devices[k]) {slot = static_cast<int32_t>(q);}}
    if (slot < 0) {
      if (static_cast<int>(m->slot_device.size()) >= MScan::kMaxDeviceSlots) {
        kh::set_error("kh_mapper_create_on_devices: more than 16 distinct devices");
        return fail(KH_ERR_INVALID_ARG);
      }
      slot = static_cast<int32_t>(m->slot_device.size());
      m->slot_device.push_back(devices[k]);
    }
    m->member_device.push_back(devices[k]);
    m->member_slot.push_back(slot);
  }
  m->n_slots = static_cast<int32_t>(m->slot_device.size());
  // Mapper::Initialize (Mapper.cpp:2606-2631): the sequential matcher; MapperGraph's constructor: the loop matcher (:1397-1400);
  // here one of each per member, member 0 = the reference's two matchers
  int rc = kh_matcher_group_create(params->correlation_search_space_dimension, params->correlation_search_space_resolution,
      params->correlation_search_space_smear_deviation, laser->range_threshold, devices, n_devices, max_candidates, &m->seq_group);
  if (rc) {return fail(rc);}
  rc = kh_matcher_group_create(params->loop_search_space_dimension, params->loop_search_space_resolution,
      params->loop_search_space_smear_deviation, laser->range_threshold, devices, n_devices, max_candidates, &m->loop_group);
  if (rc) {return fail(rc);}
  rc = kh_matcher_group_set_params(m->seq_group, &params->match); if (rc) {return fail(rc);}
  rc = kh_matcher_group_set_params(m->loop_group, &params->match); if (rc) {return fail(rc);}
  m->seq = kh_matcher_group_member(m->seq_group, 0);
  m->loop = kh_matcher_group_member(m->loop_group, 0);
  rc = kh_spa_create(m->device, &m->solver); if (rc) {return fail(rc);}
  rc = kh_graph_create(m->device, &m->graph); if (rc) {return fail(rc);}
  *out = m.release();
  return KH_OK;
}

int kh_mapper_create(const kh_mapper_params * params, const kh_laser * laser, int32_t device, int32_t max_candidates, kh_mapper ** out)
{
  return kh_mapper_create_on_devices(params, laser, &device, 1, max_candidates, out);
}

void kh_mapper_destroy(kh_mapper * m)
{
  if (!m) {return;}
  if (std::getenv("KH_MAPPER_TIMING")) {
    std::fprintf(stderr, "[kh_mapper] host pieces (ms, calls):");
    for (int k = 0; k < 12; ++k) {if (m->prof_n[k]) {std::fprintf(stderr, "  %s %.1f (%ld)", kh::kProfNames[k], m->prof_ms[k], m->prof_n[k]);}}
    std::fprintf(stderr, "\n");
  }
  if (m->log) {std::fclose(m->log);}
  kh_matcher_group_destroy(m->seq_group); kh_matcher_group_destroy(m->loop_group);
  kh_spa_destroy(m->solver); kh_graph_destroy(m->graph);
  for (auto & slabs : m->d_slabs) {
    for (double * slab : slabs) {kh_device_free(slab);}
  }
  delete m;
}

int kh_mapper_set_log(kh_mapper * m, const char * path)
{
  if (!m) {return KH_ERR_INVALID_ARG;}
  if (m->log) {std::fclose(m->log); m->log = nullptr;}
  if (path) {
    m->log = std::fopen(path, "w");
    if (!m->log) {kh::set_error("kh_mapper_set_log: cannot open the file"); return KH_ERR_IO;}
  }
  return KH_OK;
}

kh_spa * kh_mapper_solver(kh_mapper * m) {return m ? m->solver : nullptr;}

// Mapper::Process (Mapper.cpp:2679-2748)
int kh_mapper_process(kh_mapper * m, const double * ranges, const double odometric_pose[3], double time, int32_t * accepted,
  double corrected_pose[3], double covariance[9])
{
  if (!m || !ranges || !odometric_pose || !accepted) {return KH_ERR_INVALID_ARG;}
  *accepted = 0;
  if (m->failed) {
    kh::set_error("kh_mapper_process: an earlier call failed after its scan had entered the graph; the handle is unusable");
    return KH_ERR_SOLVER;
  }
  const auto t_begin = std::chrono::steady_clock::now();
  std::unique_ptr<MScan> scan(new MScan());
  scan->ranges.assign(ranges, ranges + m->laser.n);
  scan->time = time;
  scan->odometric.x = odometric_pose[0]; scan->odometric.y = odometric_pose[1]; scan->odometric.h = odometric_pose[2];
  scan->corrected = scan->odometric;                      // the caller's SetCorrectedPose(odometric pose)
  MScan * last = m->last >= 0 ? m->scans[m->last].get() : nullptr;
  // update the scan's corrected pose based on the last correction (:2699-2703)
  if (last) {scan->corrected = transform_pose(last->odometric, last->corrected, scan->odometric);}
  // HasMovedEnough (:3110-3142)
  if (last) {
    bool moved = false;
    if (scan->time - last->time >= m->p.minimum_time_interval) {moved = true;}
    // the scanner's pose for the two odometric poses (GetSensorAt, :3123-3124)
    const Pose last_scanner = sensor_at(m->laser, last->odometric), scanner = sensor_at(m->laser, scan->odometric);
    if (!moved) {
      const double deltaHeading = normalize_angle(scanner.h - last_scanner.h);
      if (std::fabs(deltaHeading) >= m->p.minimum_travel_heading) {moved = true;}
    }
    if (!moved) {
      const double dx = last_scanner.x - scanner.x, dy = last_scanner.y - scanner.y;
      if (dx * dx + dy * dy >= m->p.minimum_travel_distance * m->p.minimum_travel_distance - kTolerance) {moved = true;}
    }
    if (!moved) {return KH_OK;}
  }
  double cov[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  if (!(m->p.use_scan_matching && last)) {ProfScope prof(m, 5); update_scan(*scan, m->laser);}
  // correct the scan against the running scans (:2713-2724)
  if (m->p.use_scan_matching && last) {
    // LocalizedRangeScan::Update of the new scan (1081 sincos) is handed to the matcher as a QueryHook: the match's rasteriser needs
    // the query's sensor pose and nothing else of it, so the readings are computed behind its launches, while the GPU works; the
    // matcher runs the hook before anything reads the readings, whatever path the call takes, and before it returns
    MScan * const sp = scan.get();
    sp->sensor = sensor_at(m->laser, sp->corrected);
    sp->points.assign(2 * static_cast<size_t>(m->laser.n), 0.0);
    const kh_scan q = as_kh_scan(*sp);
    kh::set_pending_query_hook([m, sp]() {ProfScope prof(m, 5); update_scan(*sp, m->laser);});
    std::vector<kh_scan> base;
    // (uploads of scans not yet resident go in front of the match on the matcher's own stream; the match returns after its
    // last kernel, so every other reader finds them in place)
    void * seq_stream = kh_matcher_stream(m->seq);
    for (int32_t r : m->running) {base.push_back(as_base_scan(m, *m->scans[r], seq_stream));}
    double mean[3], response = 0.0;
    const auto t0 = std::chrono::steady_clock::now();
    const int rc = kh_matcher_match(m->seq, &q, base.data(), static_cast<int32_t>(base.size()), 1, 1, mean, cov, &response);
    kh::run_pending_query_hook();             // (an argument check that refused the call before the hook was taken)
    m->stats.match_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    m->stats.matches += 1;
    if (rc) {
      // an early return of the match (invalid argument, empty query, a HIP error) may leave the uploads queued above in flight:
      // wait for them, so that no other stream reads a copy marked fresh while it is still arriving
      kh::stream_synchronize(seq_stream);
      return rc;
    }
    // KH_MAPPER_DUMP_MATCH=<scan id>:<path> (debugging aid): the inputs and the result of this sequential match as raw doubles
    // (n_base, n_beams, query pose, query ranges, per base scan: pose, ranges; mean, covariance, response)
    static const char * dump_spec = std::getenv("KH_MAPPER_DUMP_MATCH");
    if (dump_spec && std::atoi(dump_spec) == static_cast<int>(m->scans.size()) && std::strchr(dump_spec, ':')) {
      if (FILE * f = std::fopen(std::strchr(dump_spec, ':') + 1, "wb")) {
        const double hdr[2] = {static_cast<double>(base.size()), static_cast<double>(q.n)};
        std::fwrite(hdr, 8, 2, f);
        std::fwrite(q.sensor_pose, 8, 3, f); std::fwrite(q.ranges, 8, static_cast<size_t>(q.n), f);
        for (const kh_scan & b : base) {std::fwrite(b.sensor_pose, 8, 3, f); std::fwrite(b.ranges, 8, static_cast<size_t>(b.n), f);}
        std::fwrite(mean, 8, 3, f); std::fwrite(cov, 8, 9, f); std::fwrite(&response, 8, 1, f);
        std::fclose(f);
      }
    }
    set_sensor_pose(m, *scan, mean);
  }
  // AddScan: state id = unique id = position in the list (:2727)
  const int32_t id = static_cast<int32_t>(m->scans.size());
  scan->id = id;
  m->scans.push_back(std::move(scan));
  m->adj.emplace_back(); m->out_edges.emplace_back();
  // from here on the scan is part of the mapper: an error return leaves the handle marked failed
  struct CommitGuard {kh_mapper * m; bool armed; ~CommitGuard() {if (armed) {m->failed = true;}}} guard{m, true};
  if (!m->graph_dirty) {
    double xy[2];
    reference_xy(m, *m->scans[id], xy);
    m->compact_of.push_back(static_cast<int32_t>(m->alive.size()));
    m->alive.push_back(id);
    if (kh_graph_append_scan(m->graph, xy) != KH_OK) {m->graph_dirty = true;}
    // the scan map's size in id space bounds the candidate walks (see sync_graph)
    const int32_t n_alive = static_cast<int32_t>(m->alive.size());
    const int32_t n_visit = static_cast<int32_t>(std::lower_bound(m->alive.begin(), m->alive.end(), n_alive) - m->alive.begin());
    if (!m->graph_dirty && kh_graph_set_scan_limit(m->graph, n_visit) != KH_OK) {m->graph_dirty = true;}
  }
  MScan & s = *m->scans[id];
  if (m->p.use_scan_matching) {
    // AddVertex (:1418-1432): the solver node carries the corrected pose
    const double node[3] = {s.corrected.x, s.corrected.y, s.corrected.h};
    if (m->log) {std::fprintf(m->log, "N %d %.17g %.17g %.17g\n", id, node[0], node[1], node[2]);}
    int rc;
    {ProfScope prof(m, 6); rc = kh_spa_add_node(m->solver, id, node);}
    if (rc) {return rc;}
    // AddEdges (:1434-1498)
    std::vector<double> means, covs;
    const bool previous_gone = last && !m->scans[id - 1];       // AddEdges returns at once (Mapper.cpp:1444-1447)
    if (last && !previous_gone) {
      const Pose sp = s.sensor_pose();
      const double scan_pose[3] = {sp.x, sp.y, sp.h};
      rc = link_scans(m, id - 1, id, scan_pose, cov); if (rc) {return rc;}
      means.insert(means.end(), scan_pose, scan_pose + 3);
      covs.insert(covs.end(), cov, cov + 9);
      rc = link_chain_to_scan(m, m->running, id, scan_pose, cov); if (rc) {return rc;}
    }
    // LinkNearChains (:1641-1663): the near chains are independent matches of the same scan -> one batch
    if (!previous_gone) {
      if (m->graph_dirty) {rc = sync_graph(m); if (rc) {return rc;}}
      std::vector<int32_t> flat(2 * static_cast<size_t>(m->max_candidates));
      int32_t n_chains = 0;
      const int32_t query = m->compact_of[id];
      {ProfScope prof(m, 8);
      rc = kh_graph_find_near_chains(m->graph, query, m->p.link_scan_maximum_distance, flat.data(), m->max_candidates, &n_chains);
      }
      if (rc) {return rc;}
      if (n_chains > m->max_candidates) {
        flat.resize(2 * static_cast<size_t>(n_chains));
        rc = kh_graph_find_near_chains(m->graph, query, m->p.link_scan_maximum_distance, flat.data(), n_chains, &n_chains);
        if (rc) {return rc;}
      }
      std::vector<std::vector<int32_t>> chains;
      for (int32_t c = 0; c < n_chains; ++c) {
        if (flat[2 * c + 1] - flat[2 * c] + 1 < m->p.loop_match_minimum_chain_size) {continue;}
        chains.push_back(run_of(m, flat[2 * c], flat[2 * c + 1]));
      }
      std::vector<MatchOut> res;
      rc = match_chains(m, m->seq_group, std::vector<kh_scan>(chains.size(), as_kh_scan(s)), chains, false, true, res);
      if (rc) {return rc;}
      for (size_t c = 0; c < chains.size(); ++c) {
        if (res[c].response > m->p.link_match_minimum_response_fine - kTolerance) {
          means.insert(means.end(), res[c].mean, res[c].mean + 3);
          covs.insert(covs.end(), res[c].cov, res[c].cov + 9);
          rc = link_chain_to_scan(m, chains[c], id, res[c].mean, res[c].cov); if (rc) {return rc;}
        }
      }
    }
    if (!means.empty()) {
      double wm[3];
      rc = kh_weighted_mean(static_cast<int32_t>(means.size() / 3), means.data(), covs.data(), wm); if (rc) {return rc;}
      set_sensor_pose(m, s, wm);
    }
    // AddRunningScan (:183-206)
    m->running.push_back(id);
    {
      auto sq = [&]() {
        const Pose f = m->scans[m->running.front()]->sensor_pose(), b = m->scans[m->running.back()]->sensor_pose();
        const double dx = f.x - b.x, dy = f.y - b.y;
        return dx * dx + dy * dy;
      };
      double squaredDistance = sq();
      while (m->running.size() > static_cast<size_t>(m->p.scan_buffer_size) ||
        squaredDistance > m->p.scan_buffer_maximum_scan_distance * m->p.scan_buffer_maximum_scan_distance - kTolerance)
      {
        m->running.erase(m->running.begin());
        squaredDistance = sq();
      }
    }
    if (m->p.do_loop_closing) {
      bool closed = false;
      rc = kh::try_close_loop(m, id, closed);
      if (rc) {return rc;}
    }
  }
  m->last = id;
  if (m->lifelong && m->p.use_scan_matching) {
    const int rc = kh::lifelong_step(m, id);
    if (rc) {return rc;}
  }
  guard.armed = false;
  if (m->log && std::getenv("KH_LOG_FINAL_POSES")) {     // debugging aid, see oracle/ref_slam_driver.cpp
    std::fprintf(m->log, "F %d %.17g %.17g %.17g\n", id, s.corrected.x, s.corrected.y, s.corrected.h);
  }
  *accepted = 1;
  if (corrected_pose) {corrected_pose[0] = s.corrected.x; corrected_pose[1] = s.corrected.y; corrected_pose[2] = s.corrected.h;}
  if (covariance) {std::copy(cov, cov + 9, covariance);}
  m->stats.scans_processed += 1;
  m->stats.process_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
  return KH_OK;
}

int32_t kh_mapper_num_scans(const kh_mapper * m) {return m ? static_cast<int32_t>(m->scans.size()) : 0;}
int64_t kh_mapper_num_edges(const kh_mapper * m) {return m ? m->n_edges : 0;}

int kh_mapper_get_poses(const kh_mapper * m, double * corrected_poses)
{
  if (!m || !corrected_poses) {return KH_ERR_INVALID_ARG;}
  for (size_t i = 0; i < m->scans.size(); ++i) {
    if (!m->scans[i]) {corrected_poses[3 * i] = corrected_poses[3 * i + 1] = corrected_poses[3 * i + 2] = std::nan(""); continue;}   // removed
    corrected_poses[3 * i] = m->scans[i]->corrected.x; corrected_poses[3 * i + 1] = m->scans[i]->corrected.y;
    corrected_poses[3 * i + 2] = m->scans[i]->corrected.h;
  }
  return KH_OK;
}

int kh_mapper_get_scan(const kh_mapper * m, int32_t index, kh_scan * scan, kh_scan_box * box)
{
  if (!m || index < 0 || index >= static_cast<int32_t>(m->scans.size()) || !m->scans[index]) {return KH_ERR_NOT_FOUND;}
  const MScan & s = *m->scans[index];
  if (scan) {*scan = kh::as_kh_scan(s);}
  if (box) {*box = kh::box_of(m, s);}
  return KH_OK;
}

int kh_mapper_get_adjacency(const kh_mapper * m, int32_t scan_id, int32_t * adjacent, int32_t capacity, int32_t * n)
{
  if (!m || !n || scan_id < 0 || scan_id >= static_cast<int32_t>(m->scans.size()) || !m->scans[scan_id]) {return KH_ERR_NOT_FOUND;}
  const std::vector<int32_t> & a = m->adj[scan_id];
  *n = static_cast<int32_t>(a.size());
  if (adjacent) {std::copy(a.begin(), a.begin() + std::min<size_t>(a.size(), static_cast<size_t>(std::max(capacity, 0))), adjacent);}
  return KH_OK;
}

int kh_mapper_set_node_score(kh_mapper * m, int32_t scan_id, double score)
{
  if (!m || scan_id < 0 || scan_id >= static_cast<int32_t>(m->scans.size()) || !m->scans[scan_id]) {return KH_ERR_NOT_FOUND;}
  m->scans[scan_id]->score = score;
  return KH_OK;
}

int kh_mapper_remove_node(kh_mapper * m, int32_t scan_id)
{
  if (!m) {return KH_ERR_INVALID_ARG;}
  // The next Process() reads the last scan and every scan of the running window; the reference (GetLastScan /
  // GetRunningScans hold raw pointers, Mapper.cpp:2688, 2716) would be left with dangling ones, and its only caller --
  // the lifelong policy, which skips the newest scan_buffer_size scans -- never asks for this.  Refused here.
  if (scan_id == m->last || std::find(m->running.begin(), m->running.end(), scan_id) != m->running.end()) {
    kh::set_error("RemoveNode: the scan is the last scan or in the running-scan window");
    return KH_ERR_INVALID_ARG;
  }
  return kh::remove_node(m, scan_id);
}

int kh_mapper_set_lifelong(kh_mapper * m, const kh_decay_params * params)
{
  if (!m) {return KH_ERR_INVALID_ARG;}
  m->lifelong = params != nullptr;
  if (params) {m->decay = *params; m->decay.scan_buffer_size = m->p.scan_buffer_size;}
  return KH_OK;
}

int32_t kh_mapper_num_alive(const kh_mapper * m)
{
  int32_t n = 0;
  if (m) {for (const auto & s : m->scans) {n += s ? 1 : 0;}}
  return n;
}

int kh_mapper_get_alive(const kh_mapper * m, int32_t * ids)
{
  if (!m || !ids) {return KH_ERR_INVALID_ARG;}
  int32_t n = 0;
  for (size_t i = 0; i < m->scans.size(); ++i) {if (m->scans[i]) {ids[n++] = static_cast<int32_t>(i);}}
  return KH_OK;
}

int kh_mapper_get_stats(const kh_mapper * m, kh_mapper_stats * out)
{
  if (!m || !out) {return KH_ERR_INVALID_ARG;}
  *out = m->stats;
  int64_t seq[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (m->seq && kh_matcher_seq_stats(m->seq, seq) == KH_OK) {
    out->fused_matches = seq[0]; out->fused_fine_passes = seq[1]; out->fused_declined = seq[6]; out->fused_declined_reason = seq[7];
  }
  return KH_OK;
}

}  // extern "C"
