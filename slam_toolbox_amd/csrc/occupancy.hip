// karto::OccupancyGrid::CreateFromScans on the GPU: the map that slam_toolbox
// publishes every map_update_interval and that "end-to-end map build" (BASELINE config[4]) ends with.
//
//   ComputeDimensions  Karto.h:6086-6112 (+ LocalizedRangeScan::Update's bounding box, Karto.h:5694-5700)
//   AddScan            Karto.h:6148-6189   RayTrace   Karto.h:6199-6232
//   Grid::TraceLine    Karto.h:4874-4927   Update / UpdateCell  Karto.h:6240-6274
//
// One thread per beam: the clip to the range threshold and both WorldToGrid roundings are the reference's
// IEEE operations (compiled without FMA contraction), the Bresenham walk is integer, and the pass / hit
// counters are uint32 atomics in L2 -- increments commute, so the counters, hence the cells, are bit-exact.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/karto_hip.h"

namespace kh
{
void set_error(const std::string & s);

__device__ __forceinline__ double o_round(double v) {return v >= 0.0 ? floor(v + 0.5) : ceil(v - 0.5);}   // Math.h:87-90
__device__ __forceinline__ int32_t o_to_int(double v)
{
  if (!(v > -2147483649.0 && v < 2147483648.0)) {return INT32_MIN;}
  return (int32_t)v;
}

struct OccDev
{
  int32_t width, height, ws;
  double off_x, off_y, scale;
  uint32_t * pass;
  uint32_t * hits;
  uint8_t * cells;
};

// beams: [n_beams] of (range, point x, point y, sensor x, sensor y) packed as 5 doubles
__global__ __launch_bounds__(256) void k_occ_trace(
  OccDev g, const double * __restrict__ beams, int64_t n_beams, double range_threshold, double min_range, double max_range)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_beams) {return;}
  const double r = beams[5 * i];
  double px = beams[5 * i + 1], py = beams[5 * i + 2];
  const double sx = beams[5 * i + 3], sy = beams[5 * i + 4];
  const bool valid_end = r < (range_threshold - 1e-06);                  // Karto.h:6167
  if (r <= min_range || r >= max_range || r != r) {return;}             // Karto.h:6169-6172
  if (r >= range_threshold) {                                           // Karto.h:6173-6180
    const double ratio = range_threshold / r;
    const double dx = px - sx, dy = py - sy;
    px = sx + ratio * dx; py = sy + ratio * dy;
  }
  int32_t x0 = o_to_int(o_round((sx - g.off_x) * g.scale)), y0 = o_to_int(o_round((sy - g.off_y) * g.scale));
  int32_t x1 = o_to_int(o_round((px - g.off_x) * g.scale)), y1 = o_to_int(o_round((py - g.off_y) * g.scale));
  const int32_t tx = x1, ty = y1;
  // Grid<kt_int32u>::TraceLine, Karto.h:4874-4927
  const bool steep = abs(y1 - y0) > abs(x1 - x0);
  int32_t t;
  if (steep) {t = x0; x0 = y0; y0 = t; t = x1; x1 = y1; y1 = t;}
  if (x0 > x1) {t = x0; x0 = x1; x1 = t; t = y0; y0 = y1; y1 = t;}
  const int32_t deltaX = x1 - x0, deltaY = abs(y1 - y0);
  int32_t error = 0, y = y0;
  const int32_t ystep = y0 < y1 ? 1 : -1;
  for (int32_t x = x0; x <= x1; x++) {
    const int32_t cx = steep ? y : x, cy = steep ? x : y;
    error += deltaY;
    if (2 * error >= deltaX) {y += ystep; error -= deltaX;}
    if (cx >= 0 && cx < g.width && cy >= 0 && cy < g.height) {atomicAdd(&g.pass[cx + (int64_t)cy * g.ws], 1u);}
  }
  if (valid_end && tx >= 0 && tx < g.width && ty >= 0 && ty < g.height) {    // Karto.h:6213-6229
    atomicAdd(&g.pass[tx + (int64_t)ty * g.ws], 1u);
    atomicAdd(&g.hits[tx + (int64_t)ty * g.ws], 1u);
  }
}

__global__ __launch_bounds__(256) void k_occ_update(OccDev g, uint32_t min_pass, double threshold)
{
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= (int64_t)g.ws * g.height) {return;}
  uint8_t c = 0;                                                       // GridStates_Unknown (Clear())
  const uint32_t p = g.pass[k];
  if (p > min_pass) {                                                   // Karto.h:6244-6252
    const double ratio = (double)g.hits[k] / (double)p;
    c = ratio > threshold ? 100 : 255;
  }
  g.cells[k] = c;
}

}  // namespace kh

using namespace kh;

struct kh_occupancy
{
  int32_t device = 0;
  hipStream_t stream = nullptr;
  OccDev dev;
  double * d_beams = nullptr; size_t cap_beams = 0;
  double * h_beams = nullptr; size_t cap_hbeams = 0;
  hipEvent_t ev[2] = {nullptr, nullptr};
  double trace_ms = 0.0; int64_t beams_traced = 0;
};

extern "C" {

int kh_occupancy_compute_dimensions(
  int32_t n_scans, const kh_scan * scans, double min_range, double range_threshold, double resolution,
  int32_t * width, int32_t * height, double offset[2])
{
  if (n_scans <= 0 || !scans || !width || !height || !offset || !(resolution > 0)) {return KH_ERR_INVALID_ARG;}
  // BoundingBox2 (Karto.h:2846-2903) over every scan's box = sensor position + in-range points (Karto.h:5694-5700)
  double min_x = 999999999999999999.99999, min_y = 999999999999999999.99999;
  double max_x = -999999999999999999.99999, max_y = -999999999999999999.99999;
  auto add = [&](double x, double y) {
    min_x = x < min_x ? x : min_x; min_y = y < min_y ? y : min_y;
    max_x = x > max_x ? x : max_x; max_y = y > max_y ? y : max_y;
  };
  for (int32_t s = 0; s < n_scans; ++s) {
    if (scans[s].n < 0 || (scans[s].n > 0 && (!scans[s].ranges || !scans[s].points_xy))) {return KH_ERR_INVALID_ARG;}
    add(scans[s].sensor_pose[0], scans[s].sensor_pose[1]);
    for (int32_t i = 0; i < scans[s].n; ++i) {
      const double r = scans[s].ranges[i];
      if (r >= min_range && r <= range_threshold) {add(scans[s].points_xy[2 * i], scans[s].points_xy[2 * i + 1]);}   // math::InRange
    }
  }
  const double scale = 1.0 / resolution;
  auto round_half_away = [](double v) {return v >= 0.0 ? std::floor(v + 0.5) : std::ceil(v - 0.5);};
  *width = static_cast<int32_t>(round_half_away((max_x - min_x) * scale));     // Karto.h:6106-6111
  *height = static_cast<int32_t>(round_half_away((max_y - min_y) * scale));
  offset[0] = min_x; offset[1] = min_y;
  return KH_OK;
}

int kh_occupancy_create(int32_t width, int32_t height, double offset_x, double offset_y, double resolution,
  int32_t device, kh_occupancy ** out)
{
  if (!out) {return KH_ERR_INVALID_ARG;}
  *out = nullptr;
  if (width <= 0 || height <= 0 || !(resolution > 0) || static_cast<int64_t>(width + 7) * height > (1ll << 31) - 4096) {
    set_error("OccupancyGrid: invalid dimensions");
    return KH_ERR_INVALID_ARG;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) {
    set_error("no usable HIP device (libkartohip has no CPU fallback)");
    return KH_ERR_NO_DEVICE;
  }
  kh_occupancy * g = new kh_occupancy();
  g->device = device;
  g->dev.width = width; g->dev.height = height; g->dev.ws = (width + 7) & ~7;      // Karto.h:4640
  g->dev.off_x = offset_x; g->dev.off_y = offset_y; g->dev.scale = 1.0 / resolution;
  g->dev.pass = nullptr; g->dev.hits = nullptr; g->dev.cells = nullptr;
  const size_t size = static_cast<size_t>(g->dev.ws) * height;
  if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking) != hipSuccess ||
    hipEventCreate(&g->ev[0]) != hipSuccess || hipEventCreate(&g->ev[1]) != hipSuccess ||
    hipMalloc(reinterpret_cast<void **>(&g->dev.pass), size * 4) != hipSuccess ||
    hipMalloc(reinterpret_cast<void **>(&g->dev.hits), size * 4) != hipSuccess ||
    hipMalloc(reinterpret_cast<void **>(&g->dev.cells), size) != hipSuccess ||
    hipMemset(g->dev.pass, 0, size * 4) != hipSuccess || hipMemset(g->dev.hits, 0, size * 4) != hipSuccess ||
    hipMemset(g->dev.cells, 0, size) != hipSuccess)
  {
    set_error("kh_occupancy_create: HIP allocation failed");
    kh_occupancy_destroy(g);
    return KH_ERR_HIP;
  }
  *out = g;
  return KH_OK;
}

void kh_occupancy_destroy(kh_occupancy * g)
{
  if (!g) {return;}
  (void)hipSetDevice(g->device);
  if (g->stream) {(void)hipStreamSynchronize(g->stream);}
  (void)hipFree(g->dev.pass); (void)hipFree(g->dev.hits); (void)hipFree(g->dev.cells); (void)hipFree(g->d_beams);
  if (g->h_beams) {(void)hipHostFree(g->h_beams);}
  if (g->ev[0]) {(void)hipEventDestroy(g->ev[0]);}
  if (g->ev[1]) {(void)hipEventDestroy(g->ev[1]);}
  if (g->stream) {(void)hipStreamDestroy(g->stream);}
  delete g;
}

int kh_occupancy_clear(kh_occupancy * g)
{
  if (!g) {return KH_ERR_INVALID_ARG;}
  if (hipSetDevice(g->device) != hipSuccess) {return KH_ERR_HIP;}
  const size_t size = static_cast<size_t>(g->dev.ws) * g->dev.height;
  if (hipMemsetAsync(g->dev.pass, 0, size * 4, g->stream) != hipSuccess || hipMemsetAsync(g->dev.hits, 0, size * 4, g->stream) != hipSuccess ||
    hipMemsetAsync(g->dev.cells, 0, size, g->stream) != hipSuccess || hipStreamSynchronize(g->stream) != hipSuccess) {return KH_ERR_HIP;}
  return KH_OK;
}

int kh_occupancy_add_scans(kh_occupancy * g, int32_t n_scans, const kh_scan * scans, double range_threshold,
  double min_range, double max_range)
{
  if (!g || n_scans < 0 || (n_scans > 0 && !scans)) {return KH_ERR_INVALID_ARG;}
  if (n_scans == 0) {return KH_OK;}
  if (hipSetDevice(g->device) != hipSuccess) {return KH_ERR_HIP;}
  size_t total = 0;
  for (int32_t s = 0; s < n_scans; ++s) {
    if (scans[s].n < 0 || (scans[s].n > 0 && (!scans[s].ranges || !scans[s].points_xy))) {return KH_ERR_INVALID_ARG;}
    total += static_cast<size_t>(scans[s].n);
  }
  if (total == 0) {return KH_OK;}
  if (total * 5 > g->cap_hbeams) {
    if (g->h_beams) {(void)hipStreamSynchronize(g->stream); (void)hipHostFree(g->h_beams); g->h_beams = nullptr;}
    if (g->d_beams) {(void)hipFree(g->d_beams); g->d_beams = nullptr;}
    const size_t cap = std::max(total * 5, g->cap_hbeams + g->cap_hbeams / 2);
    if (hipHostMalloc(reinterpret_cast<void **>(&g->h_beams), cap * 8, hipHostMallocDefault) != hipSuccess ||
      hipMalloc(reinterpret_cast<void **>(&g->d_beams), cap * 8) != hipSuccess)
    {
      set_error("kh_occupancy_add_scans: staging allocation failed");
      g->cap_hbeams = 0;
      return KH_ERR_HIP;
    }
    g->cap_hbeams = cap;
  }
  size_t k = 0;
  for (int32_t s = 0; s < n_scans; ++s) {
    const double sx = scans[s].sensor_pose[0], sy = scans[s].sensor_pose[1];
    for (int32_t i = 0; i < scans[s].n; ++i, ++k) {
      double * b = g->h_beams + 5 * k;
      b[0] = scans[s].ranges[i]; b[1] = scans[s].points_xy[2 * i]; b[2] = scans[s].points_xy[2 * i + 1]; b[3] = sx; b[4] = sy;
    }
  }
  if (hipMemcpyAsync(g->d_beams, g->h_beams, total * 5 * 8, hipMemcpyHostToDevice, g->stream) != hipSuccess) {return KH_ERR_HIP;}
  (void)hipEventRecord(g->ev[0], g->stream);
  hipLaunchKernelGGL(k_occ_trace, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0, g->stream, g->dev, g->d_beams,
    static_cast<int64_t>(total), range_threshold, min_range, max_range);
  (void)hipEventRecord(g->ev[1], g->stream);
  if (hipStreamSynchronize(g->stream) != hipSuccess) {
    set_error(std::string("kh_occupancy_add_scans: ") + hipGetErrorString(hipGetLastError()));
    return KH_ERR_HIP;
  }
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, g->ev[0], g->ev[1]);
  g->trace_ms += ms; g->beams_traced += static_cast<int64_t>(total);
  return KH_OK;
}

int kh_occupancy_update(kh_occupancy * g, uint32_t min_pass_through, double occupancy_threshold)
{
  if (!g) {return KH_ERR_INVALID_ARG;}
  if (hipSetDevice(g->device) != hipSuccess) {return KH_ERR_HIP;}
  const int64_t size = static_cast<int64_t>(g->dev.ws) * g->dev.height;
  hipLaunchKernelGGL(k_occ_update, dim3(static_cast<unsigned>((size + 255) / 256)), dim3(256), 0, g->stream, g->dev,
    min_pass_through, occupancy_threshold);
  if (hipStreamSynchronize(g->stream) != hipSuccess) {return KH_ERR_HIP;}
  return KH_OK;
}

int kh_occupancy_read(kh_occupancy * g, uint8_t * cells, uint32_t * pass, uint32_t * hits)
{
  if (!g) {return KH_ERR_INVALID_ARG;}
  if (hipSetDevice(g->device) != hipSuccess) {return KH_ERR_HIP;}
  const size_t size = static_cast<size_t>(g->dev.ws) * g->dev.height;
  if (cells && hipMemcpy(cells, g->dev.cells, size, hipMemcpyDeviceToHost) != hipSuccess) {return KH_ERR_HIP;}
  if (pass && hipMemcpy(pass, g->dev.pass, size * 4, hipMemcpyDeviceToHost) != hipSuccess) {return KH_ERR_HIP;}
  if (hits && hipMemcpy(hits, g->dev.hits, size * 4, hipMemcpyDeviceToHost) != hipSuccess) {return KH_ERR_HIP;}
  return KH_OK;
}

int kh_occupancy_info(kh_occupancy * g, int32_t * width, int32_t * height, int32_t * width_step, double * trace_ms, int64_t * beams)
{
  if (!g) {return KH_ERR_INVALID_ARG;}
  if (width) {*width = g->dev.width;}
  if (height) {*height = g->dev.height;}
  if (width_step) {*width_step = g->dev.ws;}
  if (trace_ms) {*trace_ms = g->trace_ms;}
  if (beams) {*beams = g->beams_traced;}
  return KH_OK;
}

}  // extern "C"
