// RCCL inside the library: the collectives of the multi-GPU paths (one process per GPU).
//
//   (B) SPA linearisation  every rank linearises its block of the edges; the partial normal equations H || g are summed
//                          with ONE ncclAllReduce(sum, f64) per evaluation point; the
//                          reference has a single ceres::Solve on one host (solvers/ceres_solver.cpp:243-244)
//   (A) candidate matches  independent units, no data-path collective; the 13 result doubles per pair are collected with
//                          one fixed-size ncclAllGather
//
// librccl is bound at run time (dlopen) the first time a communicator is asked for: a single-GPU user of libkartohip
// never loads it, and a process that already holds an RCCL (e.g. the one PyTorch ships) shares it through the soname.
// xGMI is point to point; the 5 MB of H || g ride one ring all-reduce -- no call pattern of the reference is
// translated here, there is none.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <atomic>
#include "lds_attr.hpp"
#include <rccl/rccl.h>

#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>

#include "../../include/karto_hip.h"

namespace kh
{
void set_error(const std::string & s);

namespace
{
struct RcclApi
{
  void * handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  const char * (*GetErrorString)(ncclResult_t) = nullptr;
  std::string why;
};

RcclApi & rccl()
{
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    // KH_RCCL_LIBRARY names the one library to bind (a site-specific build; the tests name a missing file to walk the
    // not-found path)
    const char * forced = std::getenv("KH_RCCL_LIBRARY");
    const char * names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    if (forced && forced[0]) {
      api.handle = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
    } else {
      for (const char * n : names) {
        api.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (api.handle) {break;}
      }
    }
    if (!api.handle) {
      // dlerror() hands the message out ONCE and clears it: a second call returns NULL
      const char * e = dlerror();
      api.why = std::string("librccl not found: ") + (e ? e : "?");
      return;
    }
    auto sym = [&](const char * name) {
      void * p = dlsym(api.handle, name);
      if (!p && api.why.empty()) {api.why = std::string("librccl lacks ") + name;}
      return p;
    };
    api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
    api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
    api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
    api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(sym("ncclAllReduce"));
    api.AllGather = reinterpret_cast<decltype(api.AllGather)>(sym("ncclAllGather"));
    api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
  });
  return api;
}

int rccl_fail(const char * what, ncclResult_t r)
{
  RcclApi & api = rccl();
  set_error(std::string(what) + ": " + (api.GetErrorString ? api.GetErrorString(r) : "RCCL error"));
  return KH_ERR_HIP;
}
}  // namespace
}  // namespace kh

struct kh_comm
{
  ncclComm_t comm = nullptr;
  int32_t device = 0, rank = 0, world = 1;
};

namespace kh
{
// (library-internal: the mapper's host code is free of HIP) waits for everything queued on the stream
void stream_synchronize(void * hip_stream) {(void)hipStreamSynchronize(static_cast<hipStream_t>(hip_stream));}
}  // namespace kh

extern "C" {

int kh_comm_unique_id(uint8_t id[KH_COMM_ID_BYTES])
{
  if (!id) {return KH_ERR_INVALID_ARG;}
  static_assert(KH_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "kh_comm id size");
  kh::RcclApi & api = kh::rccl();
  if (!api.why.empty()) {kh::set_error(api.why); return KH_ERR_NO_DEVICE;}
  ncclUniqueId u;
  const ncclResult_t r = api.GetUniqueId(&u);
  if (r != ncclSuccess) {return kh::rccl_fail("ncclGetUniqueId", r);}
  std::memcpy(id, u.internal, NCCL_UNIQUE_ID_BYTES);
  return KH_OK;
}

int kh_comm_create(int32_t device, int32_t rank, int32_t world, const uint8_t id[KH_COMM_ID_BYTES], kh_comm ** out)
{
  if (!out || !id || world < 1 || rank < 0 || rank >= world) {return KH_ERR_INVALID_ARG;}
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
    kh::set_error("no usable HIP device (libkartohip has no CPU fallback)");
    return KH_ERR_NO_DEVICE;
  }
  kh::RcclApi & api = kh::rccl();
  if (!api.why.empty()) {kh::set_error(api.why); return KH_ERR_NO_DEVICE;}
  if (hipSetDevice(device) != hipSuccess) {kh::set_error("hipSetDevice failed"); return KH_ERR_HIP;}
  ncclUniqueId u;
  std::memcpy(u.internal, id, NCCL_UNIQUE_ID_BYTES);
  kh_comm * c = new kh_comm();
  c->device = device; c->rank = rank; c->world = world;
  const ncclResult_t r = api.CommInitRank(&c->comm, world, u, rank);
  if (r != ncclSuccess) {delete c; return kh::rccl_fail("ncclCommInitRank", r);}
  *out = c;
  return KH_OK;
}

void kh_comm_destroy(kh_comm * c)
{
  if (!c) {return;}
  if (c->comm) {(void)hipSetDevice(c->device); (void)kh::rccl().CommDestroy(c->comm);}
  delete c;
}

int32_t kh_comm_rank(const kh_comm * c) {return c ? c->rank : -1;}
int32_t kh_comm_world(const kh_comm * c) {return c ? c->world : 0;}
int32_t kh_comm_device(const kh_comm * c) {return c ? c->device : -1;}

int kh_comm_allreduce_sum_f64(kh_comm * c, double * device_buf, int64_t count, void * hip_stream)
{
  if (!c || !device_buf || count < 0) {return KH_ERR_INVALID_ARG;}
  if (count == 0) {return KH_OK;}
  const ncclResult_t r = kh::rccl().AllReduce(device_buf, device_buf, static_cast<size_t>(count), ncclFloat64, ncclSum, c->comm,
      static_cast<hipStream_t>(hip_stream));
  if (r != ncclSuccess) {return kh::rccl_fail("ncclAllReduce", r);}
  return KH_OK;
}

int kh_comm_allgather_f64(kh_comm * c, const double * device_send, double * device_recv, int64_t count_per_rank, void * hip_stream)
{
  if (!c || !device_send || !device_recv || count_per_rank < 0) {return KH_ERR_INVALID_ARG;}
  if (count_per_rank == 0) {return KH_OK;}
  const ncclResult_t r = kh::rccl().AllGather(device_send, device_recv, static_cast<size_t>(count_per_rank), ncclFloat64, c->comm,
      static_cast<hipStream_t>(hip_stream));
  if (r != ncclSuccess) {return kh::rccl_fail("ncclAllGather", r);}
  return KH_OK;
}

// ---- device buffers for callers that have no HIP binding of their own (the collectives above take device pointers) ----
int kh_device_malloc(int32_t device, int64_t bytes, void ** out)
{
  if (!out || bytes < 0) {return KH_ERR_INVALID_ARG;}
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
    kh::set_error("no usable HIP device (libkartohip has no CPU fallback)");
    return KH_ERR_NO_DEVICE;
  }
  // (the caller's current device is put back: the multi-device mapper calls this for every non-primary slot on the
  // application's thread, and a host that shares the thread must not find its device switched)
  int prev = -1;
  (void)hipGetDevice(&prev);
  const bool ok = hipSetDevice(device) == hipSuccess && hipMalloc(out, static_cast<size_t>(bytes > 0 ? bytes : 1)) == hipSuccess;
  if (prev >= 0 && prev != device) {(void)hipSetDevice(prev);}
  if (!ok) {kh::set_error("hipMalloc failed"); return KH_ERR_HIP;}
  return KH_OK;
}

void kh_device_free(void * p) {if (p) {(void)hipFree(p);}}

int kh_device_upload(void * device_dst, const void * host_src, int64_t bytes)
{
  if (!device_dst || !host_src || bytes < 0) {return KH_ERR_INVALID_ARG;}
  if (hipMemcpy(device_dst, host_src, static_cast<size_t>(bytes), hipMemcpyHostToDevice) != hipSuccess) {kh::set_error("hipMemcpy H2D failed"); return KH_ERR_HIP;}
  return KH_OK;
}

/* The same copy queued on `hip_stream` (what a kernel launched on that stream afterwards reads is the new content); the host
 * buffer may be reused when the call returns (pageable memory is staged by the runtime before it does). */
int kh_device_upload_on(void * device_dst, const void * host_src, int64_t bytes, void * hip_stream)
{
  if (!device_dst || !host_src || bytes < 0) {return KH_ERR_INVALID_ARG;}
  if (hipMemcpyAsync(device_dst, host_src, static_cast<size_t>(bytes), hipMemcpyHostToDevice, static_cast<hipStream_t>(hip_stream)) != hipSuccess) {
    kh::set_error("hipMemcpyAsync H2D failed"); return KH_ERR_HIP;
  }
  return KH_OK;
}

/* the per-device bookkeeping of allow_dynamic_lds (lds_attr.hpp) driven with made-up device ids: 0 when every check holds,
 * otherwise the number of the first check that failed.  Needs no device. */
int kh_selftest_lds_attr(void)
{
  std::atomic<unsigned long long> done{0};
  for (int dev = -1; dev < 70; ++dev) {if (!kh::lds_attr_pending(done, dev)) {return 1;}}           // nothing is set at the start
  kh::lds_attr_mark(done, 3);
  if (kh::lds_attr_pending(done, 3)) {return 2;}                                                    // a marked device is done ...
  for (int dev = 0; dev < 70; ++dev) {if (dev != 3 && !kh::lds_attr_pending(done, dev)) {return 3;}} // ... and no other one with it (3 + 64 included)
  kh::lds_attr_mark(done, 0); kh::lds_attr_mark(done, 63);
  if (kh::lds_attr_pending(done, 0) || kh::lds_attr_pending(done, 63) || kh::lds_attr_pending(done, 3)) {return 4;}
  if (!kh::lds_attr_pending(done, 1) || !kh::lds_attr_pending(done, 62)) {return 5;}
  kh::lds_attr_mark(done, 64); kh::lds_attr_mark(done, 67); kh::lds_attr_mark(done, -1);            // devices without a bit: never "done",
  if (!kh::lds_attr_pending(done, 64) || !kh::lds_attr_pending(done, 67) || !kh::lds_attr_pending(done, -1)) {return 6;}
  if (done.load() != ((1ull << 3) | 1ull | (1ull << 63))) {return 7;}                               // and they do not touch the others' bits
  kh::lds_attr_mark(done, 3);
  if (done.load() != ((1ull << 3) | 1ull | (1ull << 63))) {return 8;}                               // marking twice changes nothing
  return 0;
}

int kh_device_download(void * host_dst, const void * device_src, int64_t bytes)
{
  if (!host_dst || !device_src || bytes < 0) {return KH_ERR_INVALID_ARG;}
  // "waits for all queued work" is a promise about the device the buffer lives on, not the caller's current one
  hipPointerAttribute_t attr;
  int prev = -1;
  if (hipPointerGetAttributes(&attr, device_src) == hipSuccess && hipGetDevice(&prev) == hipSuccess && attr.device != prev) {
    if (hipSetDevice(attr.device) != hipSuccess) {kh::set_error("hipSetDevice failed"); return KH_ERR_HIP;}
  } else {
    prev = -1;
  }
  const bool ok = hipDeviceSynchronize() == hipSuccess &&
    hipMemcpy(host_dst, device_src, static_cast<size_t>(bytes), hipMemcpyDeviceToHost) == hipSuccess;
  if (prev >= 0) {(void)hipSetDevice(prev);}
  if (!ok) {kh::set_error("hipMemcpy D2H failed"); return KH_ERR_HIP;}
  return KH_OK;
}

}  // extern "C"
