#include "karto_hip/karto_adaptor.hpp"
int probe()
{
  karto_hip::HipSpaSolver * s = nullptr;
  karto::ScanSolver * base = s;          // is-a karto::ScanSolver
  (void)base;
  karto_hip::HipScanMatcher * m = karto_hip::HipScanMatcher::Create(nullptr, 0.3, 0.01, 0.03, 12.0);
  karto::Pose2 mean; karto::Matrix3 cov;
  karto::LocalizedRangeScanVector v; karto::LocalizedRangeScanMap mp;
  if (m) {m->MatchScan(nullptr, v, mean, cov); m->MatchScan(nullptr, mp, mean, cov, false, false);
    m->CorrelateScan(nullptr, mean, karto::Vector2<kt_double>(), karto::Vector2<kt_double>(), 0.1, 0.01, true, mean, cov, false);}
  return 0;
}
