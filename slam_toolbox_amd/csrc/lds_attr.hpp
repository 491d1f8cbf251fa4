// Dynamic-LDS limit of a kernel, set once per device (shared by the matcher's and the solver's kernel files).
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>

namespace kh
{

// The bookkeeping, free of HIP calls (kh_selftest_lds_attr drives it with made-up device ids): `done` holds one bit per device
// 0 .. 63; a device beyond that has no bit and is "pending" every time (its attribute is simply set again at each launch).
inline bool lds_attr_pending(const std::atomic<unsigned long long> & done, int dev)
{
  if (dev < 0 || dev > 63) {return true;}
  return (done.load(std::memory_order_acquire) & (1ull << dev)) == 0;
}
inline void lds_attr_mark(std::atomic<unsigned long long> & done, int dev)
{
  if (dev >= 0 && dev <= 63) {done.fetch_or(1ull << dev, std::memory_order_release);}
}

// hipFuncAttributeMaxDynamicSharedMemorySize once per DEVICE and kernel (the attribute is kept per device: a launch on a device
// that never set it fails for more than 64 KB).  Racing threads set the same value twice at worst; a failed call leaves the
// device pending, so the next launch tries again instead of failing for good.
inline void allow_dynamic_lds(const void * kernel, int bytes, std::atomic<unsigned long long> & done)
{
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) {return;}
  if (lds_attr_pending(done, dev)) {
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess) {lds_attr_mark(done, dev);}
    else {(void)hipGetLastError();}
  }
}

}  // namespace kh
