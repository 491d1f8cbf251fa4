// TEST INFRASTRUCTURE ONLY -- never linked into or called from the product library.
//
// Drop-in check of the scan matcher inside the reference's OWN karto::Mapper.  karto::ScanMatcher is a
// concrete class (INTEGRATION.md section 2 shows the two-pointer type change a maintainer makes); to run the
// unmodified Mapper.cpp against the GPU matcher without touching the reference, this translation unit
// supplies STRONG definitions of the two MatchScan instantiations Mapper.cpp uses (Mapper.cpp:1472, 1511-1535,
// 1653-1654, 2714-2717).  In the reference object they are implicit template instantiations, i.e. weak
// symbols reached through the PLT, so the linker binds every call site in Mapper.o to the definitions below.
// Each karto::ScanMatcher instance gets a karto_hip::HipScanMatcher created with the same four arguments the
// Mapper passed to ScanMatcher::Create (Mapper.cpp:1397-1400, 2613-2617).
//
// Linked only into _ref/libkarto_ref_slam_gpu.so (oracle/Makefile); _ref/libkarto_ref_slam.so keeps the
// reference matcher, and tests/test_dropin_mapper_gpu.py requires the two to produce identical runs.
#include <cmath>
#include <cstdint>
#include <cstdio>
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

namespace
{
std::mutex g_mutex;
std::unordered_map<const karto::ScanMatcher *, karto_hip::HipScanMatcher *> g_matchers;
long g_calls = 0;

karto_hip::HipScanMatcher * matcher_for(karto::ScanMatcher * self, karto::LocalizedRangeScan * pScan)
{
  std::lock_guard<std::mutex> lock(g_mutex);
  ++g_calls;
  auto it = g_matchers.find(self);
  if (it != g_matchers.end()) {return it->second;}
  karto::Mapper * m = self->m_pMapper;
  const bool sequential = (self == m->m_pSequentialScanMatcher);
  const double range_threshold = pScan->GetLaserRangeFinder()->GetRangeThreshold();
  karto_hip::HipScanMatcher * h = sequential ?
    karto_hip::HipScanMatcher::Create(m, m->m_pCorrelationSearchSpaceDimension->GetValue(),
      m->m_pCorrelationSearchSpaceResolution->GetValue(), m->m_pCorrelationSearchSpaceSmearDeviation->GetValue(),
      range_threshold) :
    karto_hip::HipScanMatcher::Create(m, m->m_pLoopSearchSpaceDimension->GetValue(),
      m->m_pLoopSearchSpaceResolution->GetValue(), m->m_pLoopSearchSpaceSmearDeviation->GetValue(), range_threshold);
  if (!h) {throw std::runtime_error("ref_gpu_matcher_shim: HipScanMatcher::Create returned NULL");}
  g_matchers[self] = h;
  return h;
}
}  // namespace

namespace karto
{
template<>
kt_double ScanMatcher::MatchScan<LocalizedRangeScanVector>(
  LocalizedRangeScan * pScan, const LocalizedRangeScanVector & rBaseScans, Pose2 & rMean, Matrix3 & rCovariance,
  kt_bool doPenalize, kt_bool doRefineMatch)
{
  return matcher_for(this, pScan)->MatchScan(pScan, rBaseScans, rMean, rCovariance, doPenalize, doRefineMatch);
}

template<>
kt_double ScanMatcher::MatchScan<LocalizedRangeScanMap>(
  LocalizedRangeScan * pScan, const LocalizedRangeScanMap & rBaseScans, Pose2 & rMean, Matrix3 & rCovariance,
  kt_bool doPenalize, kt_bool doRefineMatch)
{
  return matcher_for(this, pScan)->MatchScan(pScan, rBaseScans, rMean, rCovariance, doPenalize, doRefineMatch);
}
}  // namespace karto

extern "C" {
// number of MatchScan calls the Mapper routed through the GPU matcher (the test asserts it is not zero)
long ref_gpu_matcher_calls() {return g_calls;}
void ref_gpu_matcher_release()
{
  std::lock_guard<std::mutex> lock(g_mutex);
  for (auto & kv : g_matchers) {delete kv.second;}
  g_matchers.clear();
}
}
