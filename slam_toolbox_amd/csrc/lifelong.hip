// Lifelong-mapping node-decay scoring on the GPU:
// src/experimental/slam_toolbox_lifelong.cpp:199-250 (computeObjectiveScore), :253-292 (computeScore),
// :295-329 (computeScores), :373-478 (the bounding-box / reading overlap metrics).
//
// One wave per candidate vertex: the lanes count the candidate's filtered point readings that fall strictly
// inside the intersection of the two scan boxes (the only O(P) part), lane 0 evaluates the metrics and the
// objective with the reference's IEEE operations (no FMA contraction -> bit-exact).  The candidate filter of
// computeScores (IoU below lifelong_minimum_score or fewer than 2 edges -> dropped, and not counted as a
// candidate) is a first kernel so that num_candidates is known to the second.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/karto_hip.h"

namespace kh
{
void set_error(const std::string & s);

struct BoxDev {double bx, by, w, h; int32_t id, edges; double score; int64_t pt_begin; int32_t n_pts, pad;};

__device__ __forceinline__ void d_bounds(const BoxDev & a, const BoxDev & b, double & x_l, double & x_u, double & y_l, double & y_u)
{
  const double a_ux = a.bx + (a.w / 2.0), a_uy = a.by + (a.h / 2.0), a_lx = a.bx - (a.w / 2.0), a_ly = a.by - (a.h / 2.0);
  const double b_ux = b.bx + (b.w / 2.0), b_uy = b.by + (b.h / 2.0), b_lx = b.bx - (b.w / 2.0), b_ly = b.by - (b.h / 2.0);
  x_u = a_ux < b_ux ? a_ux : b_ux;      // std::min(s1, s2): returns s2 only if s2 < s1
  y_u = a_uy < b_uy ? a_uy : b_uy;
  x_l = a_lx < b_lx ? b_lx : a_lx;      // std::max
  y_l = a_ly < b_ly ? b_ly : a_ly;
  // NB std::min(a, b) = (b < a) ? b : a and std::max(a, b) = (a < b) ? b : a; for non-NaN doubles the forms above
  // pick the same value (ties are equal values)
}
__device__ __forceinline__ double d_intersect(const BoxDev & a, const BoxDev & b)
{
  double x_l, x_u, y_l, y_u;
  d_bounds(a, b, x_l, x_u, y_l, y_u);
  const double v = (y_u - y_l) * (x_u - x_l);
  return v < 0.0 ? 0.0 : v;
}
__device__ __forceinline__ double d_iou(const BoxDev & a, const BoxDev & b)
{
  const double i = d_intersect(a, b);
  const double uni = (a.w * a.h) + (b.w * b.h) - i;
  return i / uni;
}

// one wave per candidate: the filter of computeScores (IoU below lifelong_minimum_score or fewer than 2 edges -> dropped), the
// reading count, the metrics and the objective.  (computeScore hands the number of surviving candidates to
// computeObjectiveScore, which computes candidate_scale_factor from it and never uses it, :231-240: nothing here needs the
// count, so the filter does not have to be a kernel of its own.)
__global__ __launch_bounds__(256) void k_decay(BoxDev ref, const BoxDev * cands, int32_t n, const double * points, kh_decay_params p,
  double * iou_out, double * area_out, double * reading_out, double * score_out, int32_t * kept)
{
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (wave >= n) {return;}
  const BoxDev c = cands[wave];
  double x_l, x_u, y_l, y_u;
  d_bounds(ref, c, x_l, x_u, y_l, y_u);
  int inner = 0;
  const double * pts = points + 2 * c.pt_begin;
  for (int i = lane; i < c.n_pts; i += 64) {
    const double x = pts[2 * i], y = pts[2 * i + 1];
    inner += (x < x_u && x > x_l && y < y_u && y > y_l) ? 1 : 0;
  }
#pragma unroll
  for (int s = 32; s > 0; s >>= 1) {inner += __shfl_xor(inner, s);}
  if (lane != 0) {return;}
  const double iou = d_iou(ref, c);
  const int keep = !(iou < p.iou_thresh || c.edges < 2);
  const double area = d_intersect(ref, c) / (c.h * c.w);
  const double reading = (double)inner / (double)c.n_pts;
  iou_out[wave] = iou; kept[wave] = keep;
  area_out[wave] = area; reading_out[wave] = reading;
  double score = 0.0;
  if (keep) {
    const bool lynch = c.id == 0 || c.id == 1;
    if (ref.id - c.id < p.scan_buffer_size || lynch) {
      score = c.score;
    } else {
      if (iou > p.iou_match && c.edges < 3) {
        score = -1.0;
      } else {
        const double overlap = p.overlap_scale * (reading < area ? reading : area);            // std::min(area, reading)
        double csf = p.constraint_scale * (double)(c.edges - 2);
        csf = csf < 0.0 ? 0.0 : csf;                                                            // std::max(0., ...)
        csf = csf < 1.0 ? csf : 1.0;                                                            // std::min(1.0, ...)
        csf = overlap < csf ? overlap : csf;                                                    // std::min(csf, overlap)
        score = c.score * (1.0 + csf) - overlap - p.nearby_penalty;
        if (score > 1.0) {score = 1.0;}
      }
    }
  }
  score_out[wave] = score;
}

}  // namespace kh

using namespace kh;

extern "C" {

void kh_decay_params_default(kh_decay_params * p)
{
  // slam_toolbox_lifelong.cpp:60-100; scan_buffer_size from the mapper (mapper_params_lifelong.yaml)
  p->iou_thresh = 0.10; p->iou_match = 0.85; p->removal_score = 0.10; p->overlap_scale = 0.5;
  p->constraint_scale = 0.05; p->nearby_penalty = 0.001; p->candidates_scale = 0.03; p->scan_buffer_size = 10;
}

int kh_lifelong_scores(int32_t device, const kh_scan_box * reference, int32_t n, const kh_scan_box * candidates,
  const kh_decay_params * params, int32_t * kept, double * iou, double * area_overlap, double * reading_overlap,
  double * scores)
{
  if (!reference || n < 0 || (n > 0 && !candidates) || !params) {return KH_ERR_INVALID_ARG;}
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) {
    set_error("no usable HIP device (libkartohip has no CPU fallback)");
    return KH_ERR_NO_DEVICE;
  }
  if (n == 0) {return KH_OK;}
  if (hipSetDevice(device) != hipSuccess) {return KH_ERR_HIP;}
  // One call per accepted scan of a lifelong mapper, a dozen candidates each: the call is all latency.  Inputs are packed
  // straight into ONE pinned block (boxes, then the points), one asynchronous copy in, one kernel, one copy out, one wait
  // (the first version: three blocking copies in, a memset, two kernels, two blocking copies out = 110 us per call,
  // 3.7 s of the 50 000-scan replay).  Scratch is kept per calling thread and device.
  struct Scratch
  {
    int32_t device = -1;
    hipStream_t stream = nullptr;
    char * h_in = nullptr; char * d_in = nullptr; char * h_out = nullptr; char * d_out = nullptr;
    size_t cap_in = 0, cap_out = 0;
    void release()
    {
      if (h_in) {(void)hipHostFree(h_in);} if (h_out) {(void)hipHostFree(h_out);}
      if (d_in) {(void)hipFree(d_in);} if (d_out) {(void)hipFree(d_out);}
      if (stream) {(void)hipStreamDestroy(stream);}
      *this = Scratch();
    }
    ~Scratch() {}          // freed with the process: the HIP runtime may already be gone when thread-local destructors run
  };
  static thread_local Scratch scratch;
  if (scratch.device != device) {if (scratch.device >= 0) {scratch.release();} scratch.device = device;}
  auto fail = [&](const char * what) {
    set_error(std::string("kh_lifelong_scores: ") + what);
    return KH_ERR_HIP;
  };
  const size_t nn = static_cast<size_t>(n);
  size_t n_pts = 0;
  for (int32_t k = 0; k < n; ++k) {
    if (candidates[k].n_points < 0 || (candidates[k].n_points > 0 && !candidates[k].points_xy)) {return KH_ERR_INVALID_ARG;}
    n_pts += static_cast<size_t>(candidates[k].n_points);
  }
  const size_t in_bytes = nn * sizeof(BoxDev) + std::max<size_t>(n_pts, 1) * 16;
  const size_t out_bytes = nn * (4 * sizeof(double) + sizeof(int32_t));
  if (!scratch.stream && hipStreamCreateWithFlags(&scratch.stream, hipStreamNonBlocking) != hipSuccess) {return fail("stream creation failed");}
  if (in_bytes > scratch.cap_in) {
    if (scratch.h_in) {(void)hipHostFree(scratch.h_in); scratch.h_in = nullptr;}
    if (scratch.d_in) {(void)hipFree(scratch.d_in); scratch.d_in = nullptr;}
    scratch.cap_in = std::max(in_bytes, 2 * scratch.cap_in);
    if (hipHostMalloc(reinterpret_cast<void **>(&scratch.h_in), scratch.cap_in, hipHostMallocDefault) != hipSuccess ||
      hipMalloc(reinterpret_cast<void **>(&scratch.d_in), scratch.cap_in) != hipSuccess) {scratch.release(); return fail("allocation failed");}
  }
  if (out_bytes > scratch.cap_out) {
    if (scratch.h_out) {(void)hipHostFree(scratch.h_out); scratch.h_out = nullptr;}
    if (scratch.d_out) {(void)hipFree(scratch.d_out); scratch.d_out = nullptr;}
    scratch.cap_out = std::max(out_bytes, 2 * scratch.cap_out);
    if (hipHostMalloc(reinterpret_cast<void **>(&scratch.h_out), scratch.cap_out, hipHostMallocDefault) != hipSuccess ||
      hipMalloc(reinterpret_cast<void **>(&scratch.d_out), scratch.cap_out) != hipSuccess) {scratch.release(); return fail("allocation failed");}
  }
  auto box = [](const kh_scan_box & s) {
    BoxDev b; b.bx = s.barycenter[0]; b.by = s.barycenter[1]; b.w = s.bbox_size[0]; b.h = s.bbox_size[1];
    b.id = s.unique_id; b.edges = s.n_edges; b.score = s.score; b.pt_begin = 0; b.n_pts = 0; b.pad = 0;
    return b;
  };
  BoxDev * h_boxes = reinterpret_cast<BoxDev *>(scratch.h_in);
  double * h_pts = reinterpret_cast<double *>(scratch.h_in + nn * sizeof(BoxDev));
  size_t at = 0;
  for (int32_t k = 0; k < n; ++k) {
    h_boxes[k] = box(candidates[k]);
    h_boxes[k].pt_begin = static_cast<int64_t>(at);
    h_boxes[k].n_pts = candidates[k].n_points;
    std::copy(candidates[k].points_xy, candidates[k].points_xy + 2 * static_cast<size_t>(candidates[k].n_points), h_pts + 2 * at);
    at += static_cast<size_t>(candidates[k].n_points);
  }
  const BoxDev ref = box(*reference);
  hipStream_t st = scratch.stream;
  if (hipMemcpyAsync(scratch.d_in, scratch.h_in, in_bytes, hipMemcpyHostToDevice, st) != hipSuccess) {return fail("upload failed");}
  const BoxDev * d_boxes = reinterpret_cast<const BoxDev *>(scratch.d_in);
  const double * d_pts = reinterpret_cast<const double *>(scratch.d_in + nn * sizeof(BoxDev));
  double * d_iou = reinterpret_cast<double *>(scratch.d_out), * d_area = d_iou + nn, * d_read = d_iou + 2 * nn, * d_score = d_iou + 3 * nn;
  int32_t * d_kept = reinterpret_cast<int32_t *>(d_iou + 4 * nn);
  hipLaunchKernelGGL(k_decay, dim3((n + 3) / 4), dim3(256), 0, st, ref, d_boxes, n, d_pts, *params, d_iou, d_area, d_read, d_score, d_kept);
  if (hipMemcpyAsync(scratch.h_out, scratch.d_out, out_bytes, hipMemcpyDeviceToHost, st) != hipSuccess ||
    hipStreamSynchronize(st) != hipSuccess) {return fail("kernel or download failed");}
  const double * out = reinterpret_cast<const double *>(scratch.h_out);
  const int32_t * k_host = reinterpret_cast<const int32_t *>(out + 4 * nn);
  for (size_t k = 0; k < nn; ++k) {
    if (iou) {iou[k] = out[k];}
    if (area_overlap) {area_overlap[k] = out[nn + k];}
    if (reading_overlap) {reading_overlap[k] = out[2 * nn + k];}
    if (scores) {scores[k] = out[3 * nn + k];}
    if (kept) {kept[k] = k_host[k];}
  }
  return KH_OK;
}

}  // extern "C"
