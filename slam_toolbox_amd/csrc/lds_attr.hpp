// Dynamic-LDS limit of a kernel, set once per device (shared by the matcher's and the solver's kernel files).
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>

namespace kh
{

// hipFuncAttributeMaxDynamicSharedMemorySize once per DEVICE and kernel (the attribute is kept per device: a launch on a device
// that never set it fails for more than 64 KB).  `done` = one bit per device; racing threads set the same value twice at worst.
inline void allow_dynamic_lds(const void * kernel, int bytes, std::atomic<unsigned long long> & done)
{
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) {return;}
  const unsigned long long bit = 1ull << (dev & 63);
  if ((done.load(std::memory_order_acquire) & bit) == 0) {
    (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    done.fetch_or(bit, std::memory_order_release);
  }
}

}  // namespace kh
