// TEST INFRASTRUCTURE ONLY -- never linked into or called from the product library.
//
// extern "C" driver over the reference's OWN karto_sdk sources (compiled in place from
// /root/reference/lib/karto_sdk/src/{Karto,Mapper}.cpp against oracle/ref_stubs; see
// oracle/Makefile).  It exists to (1) pin the C restatement in oracle/karto_oracle.c and
// (2) generate the golden fixtures under tests/golden/ (tests/golden/make_golden.py).
// Only this translation unit sees private members (#define private public below); the
// reference sources themselves are compiled unmodified.
#include <cmath>
#include <cstdint>
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

using namespace karto;

namespace
{
LaserRangeFinder * g_lrf = nullptr;
int g_scan_counter = 0;
const char * kLaserName = "laser0";
}

extern "C" {

// Registers one LaserRangeFinder (Karto.h:3874-4368) in the SensorManager singleton.
int ref_init_laser(
  double min_angle, double max_angle, double ang_res,
  double min_range, double max_range, double range_threshold)
{
  if (g_lrf != nullptr) {
    SensorManager::GetInstance()->UnregisterSensor(g_lrf);
    delete g_lrf;
    g_lrf = nullptr;
  }
  g_lrf = LaserRangeFinder::CreateLaserRangeFinder(LaserRangeFinder_Custom, Name(kLaserName));
  g_lrf->SetMinimumRange(min_range);
  g_lrf->SetMaximumRange(max_range);
  g_lrf->SetMinimumAngle(min_angle);
  g_lrf->SetMaximumAngle(max_angle);
  g_lrf->SetAngularResolution(ang_res);
  g_lrf->SetRangeThreshold(range_threshold);
  SensorManager::GetInstance()->RegisterSensor(g_lrf);
  return static_cast<int>(g_lrf->GetNumberOfRangeReadings());
}

// LaserRangeFinder::SetOffsetPose: where the sensor sits on the robot (call after ref_init_laser)
int ref_set_laser_offset(double x, double y, double heading)
{
  if (g_lrf == nullptr) {return -1;}
  g_lrf->SetOffsetPose(Pose2(x, y, heading));
  return 0;
}

void ref_set_threads(int n) {tbb::ref_thread_count() = n;}

void * ref_mapper_create() {return new Mapper();}
void ref_mapper_destroy(void * m) {delete static_cast<Mapper *>(m);}

// The eight parameters ScanMatcher reads through friend access (Mapper.cpp:590-627, 671-682).
// NOTE the reference setters square the two variance penalties (Mapper.cpp:2562-2570).
void ref_mapper_set_match_params(
  void * m, double coarse_search_angle_offset, double coarse_angle_resolution,
  double fine_search_angle_offset, int use_response_expansion,
  double distance_variance_penalty_sqrt, double minimum_distance_penalty,
  double angle_variance_penalty_sqrt, double minimum_angle_penalty)
{
  Mapper * p = static_cast<Mapper *>(m);
  p->setParamCoarseSearchAngleOffset(coarse_search_angle_offset);
  p->setParamCoarseAngleResolution(coarse_angle_resolution);
  p->setParamFineSearchAngleOffset(fine_search_angle_offset);
  p->setParamUseResponseExpansion(use_response_expansion != 0);
  p->setParamDistanceVariancePenalty(distance_variance_penalty_sqrt);
  p->setParamMinimumDistancePenalty(minimum_distance_penalty);
  p->setParamAngleVariancePenalty(angle_variance_penalty_sqrt);
  p->setParamMinimumAnglePenalty(minimum_angle_penalty);
}

void ref_mapper_get_variance_penalties(void * m, double * dist_var, double * ang_var)
{
  Mapper * p = static_cast<Mapper *>(m);
  *dist_var = p->m_pDistanceVariancePenalty->GetValue();
  *ang_var = p->m_pAngleVariancePenalty->GetValue();
}

void * ref_matcher_create(
  void * mapper, double search_size, double resolution, double smear, double range_threshold)
{
  try {
    return ScanMatcher::Create(
      static_cast<Mapper *>(mapper), search_size, resolution, smear, range_threshold);
  } catch (const std::exception & e) {
    return nullptr;
  }
}
void ref_matcher_destroy(void * h) {delete static_cast<ScanMatcher *>(h);}

void * ref_scan_create(const double * ranges, int n, const double * pose)
{
  RangeReadingsVector r(ranges, ranges + n);
  LocalizedRangeScan * s = new LocalizedRangeScan(Name(kLaserName), r);
  Pose2 p(pose[0], pose[1], pose[2]);
  s->SetOdometricPose(p);
  s->SetCorrectedPose(p);
  s->SetUniqueId(g_scan_counter++);
  return s;
}
void ref_scan_destroy(void * s) {delete static_cast<LocalizedRangeScan *>(s);}
void ref_scan_set_pose(void * s, const double * pose)
{
  static_cast<LocalizedRangeScan *>(s)->SetCorrectedPoseAndUpdate(Pose2(pose[0], pose[1], pose[2]));
}
void ref_scan_set_sensor_pose(void * s, const double * pose)
{
  static_cast<LocalizedRangeScan *>(s)->SetSensorPose(Pose2(pose[0], pose[1], pose[2]));
}
void ref_scan_get_sensor_pose(void * s, double * pose)
{
  Pose2 p = static_cast<LocalizedRangeScan *>(s)->GetSensorPose();
  pose[0] = p.GetX(); pose[1] = p.GetY(); pose[2] = p.GetHeading();
}
// unfiltered world points (Karto.h:5613-5628, default wantFiltered=false)
int ref_scan_points(void * s, double * xy, int cap)
{
  const PointVectorDouble & pts = static_cast<LocalizedRangeScan *>(s)->GetPointReadings(false);
  int n = static_cast<int>(pts.size());
  for (int i = 0; i < n && i < cap; ++i) {
    xy[2 * i] = pts[i].GetX();
    xy[2 * i + 1] = pts[i].GetY();
  }
  return n;
}

int ref_find_valid_points(void * h, void * s, const double * viewpoint, double * xy, int cap)
{
  ScanMatcher * m = static_cast<ScanMatcher *>(h);
  PointVectorDouble v = m->FindValidPoints(
    static_cast<LocalizedRangeScan *>(s), Vector2<kt_double>(viewpoint[0], viewpoint[1]));
  int n = static_cast<int>(v.size());
  for (int i = 0; i < n && i < cap; ++i) {
    xy[2 * i] = v[i].GetX();
    xy[2 * i + 1] = v[i].GetY();
  }
  return n;
}

static void fill_cov(const Matrix3 & c, double * cov)
{
  for (int r = 0; r < 3; ++r) {
    for (int q = 0; q < 3; ++q) {cov[3 * r + q] = c(r, q);}
  }
}

// Mapper.cpp:534-639.  Returns response; -1e9 on exception.
double ref_match_scan(
  void * h, void * scan, void ** base, int n_base, int do_penalize, int do_refine,
  double * mean, double * cov)
{
  ScanMatcher * m = static_cast<ScanMatcher *>(h);
  LocalizedRangeScanVector v;
  for (int i = 0; i < n_base; ++i) {v.push_back(static_cast<LocalizedRangeScan *>(base[i]));}
  Pose2 pm;
  Matrix3 c;
  try {
    double r = m->MatchScan(
      static_cast<LocalizedRangeScan *>(scan), v, pm, c, do_penalize != 0, do_refine != 0);
    mean[0] = pm.GetX(); mean[1] = pm.GetY(); mean[2] = pm.GetHeading();
    fill_cov(c, cov);
    return r;
  } catch (const std::exception & e) {
    return -1e9;
  }
}

// Re-centres the grid on the scan and rasterises the base scans, i.e. Mapper.cpp:543-574 only.
void ref_add_scans(void * h, void * scan, void ** base, int n_base)
{
  ScanMatcher * m = static_cast<ScanMatcher *>(h);
  LocalizedRangeScan * pScan = static_cast<LocalizedRangeScan *>(scan);
  Pose2 scanPose = pScan->GetSensorPose();
  Rectangle2<kt_int32s> roi = m->m_pCorrelationGrid->GetROI();
  Vector2<kt_double> offset;
  offset.SetX(scanPose.GetX() - (0.5 * (roi.GetWidth() - 1) * m->m_pCorrelationGrid->GetResolution()));
  offset.SetY(scanPose.GetY() - (0.5 * (roi.GetHeight() - 1) * m->m_pCorrelationGrid->GetResolution()));
  m->m_pCorrelationGrid->GetCoordinateConverter()->SetOffset(offset);
  LocalizedRangeScanVector v;
  for (int i = 0; i < n_base; ++i) {v.push_back(static_cast<LocalizedRangeScan *>(base[i]));}
  m->AddScans(v, scanPose.GetPosition());
}

// Mapper.cpp:712-862 on whatever grid is currently rasterised.
double ref_correlate_scan(
  void * h, void * scan, const double * center, double off_x, double off_y,
  double res_x, double res_y, double ang_off, double ang_res, int do_penalize, int fine,
  double * mean, double * cov)
{
  ScanMatcher * m = static_cast<ScanMatcher *>(h);
  Pose2 pm;
  Matrix3 c;
  for (int r = 0; r < 3; ++r) {
    for (int q = 0; q < 3; ++q) {c(r, q) = cov[3 * r + q];}
  }
  try {
    double r = m->CorrelateScan(
      static_cast<LocalizedRangeScan *>(scan), Pose2(center[0], center[1], center[2]),
      Vector2<kt_double>(off_x, off_y), Vector2<kt_double>(res_x, res_y), ang_off, ang_res,
      do_penalize != 0, pm, c, fine != 0);
    mean[0] = pm.GetX(); mean[1] = pm.GetY(); mean[2] = pm.GetHeading();
    fill_cov(c, cov);
    return r;
  } catch (const std::exception & e) {
    return -1e9;
  }
}

// geometry: {width, height, widthStep, roi.x, roi.y, roi.w, roi.h, kernelSize, dataSize}
void ref_grid_info(void * h, int * out, double * offset_scale)
{
  CorrelationGrid * g = static_cast<ScanMatcher *>(h)->m_pCorrelationGrid;
  out[0] = g->GetWidth(); out[1] = g->GetHeight(); out[2] = g->GetWidthStep();
  out[3] = g->GetROI().GetX(); out[4] = g->GetROI().GetY();
  out[5] = g->GetROI().GetWidth(); out[6] = g->GetROI().GetHeight();
  out[7] = g->m_KernelSize; out[8] = g->GetDataSize();
  offset_scale[0] = g->GetCoordinateConverter()->GetOffset().GetX();
  offset_scale[1] = g->GetCoordinateConverter()->GetOffset().GetY();
  offset_scale[2] = g->GetCoordinateConverter()->GetScale();
}
const uint8_t * ref_grid_data(void * h) {return static_cast<ScanMatcher *>(h)->m_pCorrelationGrid->GetDataPointer();}
const uint8_t * ref_kernel_data(void * h) {return static_cast<ScanMatcher *>(h)->m_pCorrelationGrid->m_pKernel;}

// lookup table left behind by the last CorrelateScan/ComputeOffsets (Karto.h:6797-6894)
int ref_lookup_angles(void * h) {return static_cast<int>(static_cast<ScanMatcher *>(h)->m_pGridLookup->m_Size);}
int ref_lookup_row(void * h, int angle_index, int32_t * out, int cap)
{
  const LookupArray * a = static_cast<ScanMatcher *>(h)->m_pGridLookup->GetLookupArray(angle_index);
  int n = static_cast<int>(a->GetSize());
  std::memcpy(out, a->GetArrayPointer(), sizeof(int32_t) * std::min(n, cap));
  return n;
}
void ref_compute_offsets(void * h, void * scan, double angle_center, double ang_off, double ang_res)
{
  static_cast<ScanMatcher *>(h)->m_pGridLookup->ComputeOffsets(
    static_cast<LocalizedRangeScan *>(scan), angle_center, ang_off, ang_res);
}
// Mapper.cpp:1172-1208 for the current lookup table
double ref_get_response(void * h, int angle_index, int grid_index)
{
  return static_cast<ScanMatcher *>(h)->GetResponse(angle_index, grid_index);
}
// ROI-shifted index of a world position (Mapper.cpp:660-662)
int ref_world_to_grid_index(void * h, double x, double y)
{
  CorrelationGrid * g = static_cast<ScanMatcher *>(h)->m_pCorrelationGrid;
  Vector2<kt_int32s> gp = g->WorldToGrid(Vector2<kt_double>(x, y));
  return g->GridIndex(gp, false);
}
// search-space-probs grid left behind by the last coarse CorrelateScan
int ref_probs(void * h, double * out, int cap)
{
  Grid<kt_double> * p = static_cast<ScanMatcher *>(h)->m_pSearchSpaceProbs;
  int w = p->GetWidth(), hh = p->GetHeight(), ws = p->GetWidthStep();
  int k = 0;
  for (int y = 0; y < hh; ++y) {
    for (int x = 0; x < w; ++x) {
      if (k < cap) {out[k] = p->GetDataPointer()[y * ws + x];}
      ++k;
    }
  }
  return k;
}

// LinkInfo::Update (Mapper.h:174-188): measurement + rotated covariance the solver consumes.
void ref_link_info(const double * pose1, const double * pose2, const double * cov, double * diff, double * cov_out)
{
  Matrix3 c;
  for (int r = 0; r < 3; ++r) {
    for (int q = 0; q < 3; ++q) {c(r, q) = cov[3 * r + q];}
  }
  LinkInfo li(Pose2(pose1[0], pose1[1], pose1[2]), Pose2(pose2[0], pose2[1], pose2[2]), c);
  Pose2 d = li.GetPoseDifference();
  diff[0] = d.GetX(); diff[1] = d.GetY(); diff[2] = d.GetHeading();
  fill_cov(li.GetCovariance(), cov_out);
}
// Matrix3::Inverse (Karto.h:2533-2577)
// Transform(pose1, pose2).TransformPose(src), Karto.h:2946-3024 (what Mapper::Process does to a new scan's odometric pose,
// Mapper.cpp:2699-2703)
void ref_transform_pose(const double * pose1, const double * pose2, const double * src, double * out)
{
  Transform t(Pose2(pose1[0], pose1[1], pose1[2]), Pose2(pose2[0], pose2[1], pose2[2]));
  const Pose2 r = t.TransformPose(Pose2(src[0], src[1], src[2]));
  out[0] = r.GetX(); out[1] = r.GetY(); out[2] = r.GetHeading();
}

void ref_matrix3_inverse(const double * a, double * out)
{
  Matrix3 c;
  for (int r = 0; r < 3; ++r) {
    for (int q = 0; q < 3; ++q) {c(r, q) = a[3 * r + q];}
  }
  fill_cov(c.Inverse(), out);
}

// OccupancyGrid::CreateFromScans (Karto.h:5947-5962) on the given scans: dims = {width, height, width step},
// offset = grid offset; cells / pass / hits are copied out when the capacity allows.  Returns the data size.
int ref_occupancy_from_scans(
  void ** scans, int n, double resolution, int * dims, double * offset, uint8_t * cells, uint32_t * pass,
  uint32_t * hits, int cap)
{
  LocalizedRangeScanVector v;
  for (int i = 0; i < n; ++i) {v.push_back(static_cast<LocalizedRangeScan *>(scans[i]));}
  OccupancyGrid * g = OccupancyGrid::CreateFromScans(v, resolution);
  if (g == NULL) {return 0;}
  dims[0] = g->GetWidth(); dims[1] = g->GetHeight(); dims[2] = g->GetWidthStep();
  offset[0] = g->GetCoordinateConverter()->GetOffset().GetX();
  offset[1] = g->GetCoordinateConverter()->GetOffset().GetY();
  const int size = g->GetDataSize();
  if (size <= cap) {
    std::memcpy(cells, g->GetDataPointer(), size);
    std::memcpy(pass, g->m_pCellPassCnt->GetDataPointer(), static_cast<size_t>(size) * 4);
    std::memcpy(hits, g->m_pCellHitsCnt->GetDataPointer(), static_cast<size_t>(size) * 4);
  }
  delete g;
  return size;
}

}  // extern "C"
