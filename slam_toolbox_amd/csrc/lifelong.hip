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

// pts: the candidate's readings (device-visible address); n_pts: how many readings count (the denominator of the reading overlap);
// n_scan: 0 = every one of the n_pts readings at pts counts (the packed form of kh_lifelong_scores), otherwise pts holds the n_scan
// UNFILTERED readings of the scan and bit i of the candidate's mask says whether reading i passed the range filter
struct BoxDev {double bx, by, w, h; int32_t id, edges; double score; const double * pts; int32_t n_pts, n_scan;};

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
// ticket / target / flag (round 6): inputs and outputs live in host-coherent memory; the wave that brings the ticket counter to
// `target` -- the last one of the call -- raises `flag` <- seq behind a system-scope release, and the host reads the results without a
// download or a stream drain (the call is all latency: one per accepted scan of a lifelong mapper)
// masks / mask_words (the resident form, kh::lifelong_scores_resident): candidate k's filter bits at masks + k * mask_words
__global__ __launch_bounds__(256) void k_decay(BoxDev ref, const BoxDev * cands, int32_t n, const unsigned long long * masks, int32_t mask_words,
  kh_decay_params p, double * iou_out, double * area_out, double * reading_out, double * score_out, int32_t * kept, unsigned int * ticket,
  unsigned int target, int32_t * flag, int32_t seq)
{
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (wave >= n) {return;}
  // (the box and the lane's mask word are requested together: both live in host memory, one PCIe round trip instead of two)
  unsigned long long mword = 0;
  if (masks && lane < mask_words) {mword = masks[static_cast<size_t>(wave) * mask_words + lane];}
  const BoxDev c = cands[wave];
  double x_l, x_u, y_l, y_u;
  d_bounds(ref, c, x_l, x_u, y_l, y_u);
  int inner = 0;
  const double * pts = c.pts;
  if (c.n_scan > 0 && !pts) {
    // (a candidate whose readings the caller left out: its score does not depend on them)
  } else if (c.n_scan > 0) {
    const double2 * p2 = reinterpret_cast<const double2 *>(pts);
#pragma unroll 4
    for (int k = 0; k * 64 < c.n_scan; ++k) {
      const unsigned long long w = __shfl(mword, k);
      const int i = k * 64 + lane;
      const double2 q = p2[i < c.n_scan ? i : 0];
      const bool counts = i < c.n_scan && ((w >> lane) & 1ull);
      inner += (counts && q.x < x_u && q.x > x_l && q.y < y_u && q.y > y_l) ? 1 : 0;
    }
  } else {
    for (int i = lane; i < c.n_pts; i += 64) {
      const double x = pts[2 * i], y = pts[2 * i + 1];
      inner += (x < x_u && x > x_l && y < y_u && y > y_l) ? 1 : 0;
    }
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
  if (ticket) {
    __threadfence_system();
    if (atomicAdd(ticket, 1u) + 1u == target) {
      __threadfence_system();
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// kh_lifelong_scores, and the form the mapper calls (resident != nullptr): candidate k's UNFILTERED readings already lie in HBM at
// resident[k] (n_scan of them: the copy the matcher reads the scan from as a base scan) and masks[k] holds one bit per reading, set
// where the reading passed the range filter (LocalizedRangeScan::Update's InRange test) -- nothing but 64 + n_scan / 8 bytes per
// candidate crosses PCIe, instead of 16 bytes per filtered reading.
int decay_scores(int32_t device, const kh_scan_box * reference, int32_t n, const kh_scan_box * candidates, const double * const * resident,
  const uint64_t * const * masks, int32_t n_scan, const kh_decay_params * params, int32_t * kept, double * iou, double * area_overlap,
  double * reading_overlap, double * scores)
{
  if (!reference || n < 0 || (n > 0 && !candidates) || !params) {return KH_ERR_INVALID_ARG;}
  const int32_t mask_words = resident ? (n_scan + 63) / 64 : 0;
  if (resident && (!masks || n_scan <= 0 || mask_words > 64)) {return KH_ERR_INVALID_ARG;}
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) {
    set_error("no usable HIP device (libkartohip has no CPU fallback)");
    return KH_ERR_NO_DEVICE;
  }
  if (n == 0) {return KH_OK;}
  if (hipSetDevice(device) != hipSuccess) {return KH_ERR_HIP;}
  // One call per accepted scan of a lifelong mapper, a dozen candidates each: the call is all latency.  Inputs are packed
  // straight into ONE host-coherent block the kernel reads in place (boxes, then the masks or the points), the results come back
  // through another, and the last wave of the call raises a flag the host polls: one launch, no copy, no stream drain.  (The first
  // version: three blocking copies in, a memset, two kernels, two blocking copies out = 110 us per call, 3.7 s of the 50 000-scan
  // replay; one pinned block each way + one wait: 44 us.)  Scratch is kept per calling thread and device.
  struct Scratch
  {
    int32_t device = -1;
    hipStream_t stream = nullptr;
    char * h_in = nullptr; char * d_in = nullptr; char * h_out = nullptr; char * d_out = nullptr;
    size_t cap_in = 0, cap_out = 0;
    bool coherent = true;                  // h_in / h_out are host-coherent mappings the kernel reads and writes in place
    unsigned int * d_ticket = nullptr; unsigned int tickets = 0;
    int32_t * h_flag = nullptr; int32_t seq = 0;
    void release()
    {
      if (h_in) {(void)hipHostFree(h_in);} if (h_out) {(void)hipHostFree(h_out);}
      if (d_in) {(void)hipFree(d_in);} if (d_out) {(void)hipFree(d_out);}
      if (d_ticket) {(void)hipFree(d_ticket);} if (h_flag) {(void)hipHostFree(h_flag);}
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
    if (candidates[k].n_points < 0) {return KH_ERR_INVALID_ARG;}
    if (resident) {
      if (!masks[k]) {return KH_ERR_INVALID_ARG;}                // (resident[k] == NULL: the candidate's readings are not counted, reading overlap 0)
    } else {
      if (candidates[k].n_points > 0 && !candidates[k].points_xy) {return KH_ERR_INVALID_ARG;}
      n_pts += static_cast<size_t>(candidates[k].n_points);
    }
  }
  const size_t tail_bytes = resident ? nn * static_cast<size_t>(mask_words) * 8 : std::max<size_t>(n_pts, 1) * 16;
  const size_t in_bytes = nn * sizeof(BoxDev) + tail_bytes;
  const size_t out_bytes = nn * (4 * sizeof(double) + sizeof(int32_t));
  if (!scratch.stream && hipStreamCreateWithFlags(&scratch.stream, hipStreamNonBlocking) != hipSuccess) {return fail("stream creation failed");}
  const unsigned int host_flags = hipHostMallocPortable | hipHostMallocMapped | hipHostMallocCoherent;
  if (scratch.coherent && !scratch.h_flag) {
    if (hipHostMalloc(reinterpret_cast<void **>(&scratch.h_flag), 64, host_flags) != hipSuccess ||
      hipMalloc(reinterpret_cast<void **>(&scratch.d_ticket), sizeof(unsigned int)) != hipSuccess ||
      hipMemset(scratch.d_ticket, 0, sizeof(unsigned int)) != hipSuccess) {
      (void)hipGetLastError();
      if (scratch.h_flag) {(void)hipHostFree(scratch.h_flag); scratch.h_flag = nullptr;}
      if (scratch.d_ticket) {(void)hipFree(scratch.d_ticket); scratch.d_ticket = nullptr;}
      scratch.coherent = false;            // (a platform without host-coherent mappings keeps the copies and the stream drain)
    } else {
      scratch.h_flag[0] = 0;
    }
  }
  if (in_bytes > scratch.cap_in) {
    if (scratch.h_in) {(void)hipHostFree(scratch.h_in); scratch.h_in = nullptr;}
    if (scratch.d_in) {(void)hipFree(scratch.d_in); scratch.d_in = nullptr;}
    const size_t want = std::max(in_bytes, 2 * scratch.cap_in);
    scratch.cap_in = 0;
    if (scratch.coherent) {
      if (hipHostMalloc(reinterpret_cast<void **>(&scratch.h_in), want, host_flags) != hipSuccess) {scratch.release(); return fail("allocation failed");}
    } else if (hipHostMalloc(reinterpret_cast<void **>(&scratch.h_in), want, hipHostMallocDefault) != hipSuccess ||
      hipMalloc(reinterpret_cast<void **>(&scratch.d_in), want) != hipSuccess) {scratch.release(); return fail("allocation failed");}
    scratch.cap_in = want;
  }
  if (out_bytes > scratch.cap_out) {
    if (scratch.h_out) {(void)hipHostFree(scratch.h_out); scratch.h_out = nullptr;}
    if (scratch.d_out) {(void)hipFree(scratch.d_out); scratch.d_out = nullptr;}
    const size_t want = std::max(out_bytes, 2 * scratch.cap_out);
    scratch.cap_out = 0;
    if (scratch.coherent) {
      if (hipHostMalloc(reinterpret_cast<void **>(&scratch.h_out), want, host_flags) != hipSuccess) {scratch.release(); return fail("allocation failed");}
    } else if (hipHostMalloc(reinterpret_cast<void **>(&scratch.h_out), want, hipHostMallocDefault) != hipSuccess ||
      hipMalloc(reinterpret_cast<void **>(&scratch.d_out), want) != hipSuccess) {scratch.release(); return fail("allocation failed");}
    scratch.cap_out = want;
  }
  auto box = [](const kh_scan_box & s) {
    BoxDev b; b.bx = s.barycenter[0]; b.by = s.barycenter[1]; b.w = s.bbox_size[0]; b.h = s.bbox_size[1];
    b.id = s.unique_id; b.edges = s.n_edges; b.score = s.score; b.pts = nullptr; b.n_pts = 0; b.n_scan = 0;
    return b;
  };
  // (the kernel sees the input block at d_base: the block itself when it is host-coherent, its device copy otherwise)
  char * d_base = scratch.coherent ? scratch.h_in : scratch.d_in;
  BoxDev * h_boxes = reinterpret_cast<BoxDev *>(scratch.h_in);
  char * h_tail = scratch.h_in + nn * sizeof(BoxDev);
  const char * d_tail = d_base + nn * sizeof(BoxDev);
  size_t at = 0;
  for (int32_t k = 0; k < n; ++k) {
    h_boxes[k] = box(candidates[k]);
    h_boxes[k].n_pts = candidates[k].n_points;
    if (resident) {
      h_boxes[k].pts = resident[k];
      h_boxes[k].n_scan = n_scan;
      std::copy(masks[k], masks[k] + mask_words, reinterpret_cast<uint64_t *>(h_tail) + static_cast<size_t>(k) * mask_words);
    } else {
      h_boxes[k].pts = reinterpret_cast<const double *>(d_tail) + 2 * at;
      std::copy(candidates[k].points_xy, candidates[k].points_xy + 2 * static_cast<size_t>(candidates[k].n_points), reinterpret_cast<double *>(h_tail) + 2 * at);
      at += static_cast<size_t>(candidates[k].n_points);
    }
  }
  const BoxDev ref = box(*reference);
  hipStream_t st = scratch.stream;
  const BoxDev * d_boxes = reinterpret_cast<const BoxDev *>(d_base);
  const unsigned long long * d_masks = resident ? reinterpret_cast<const unsigned long long *>(d_tail) : nullptr;
  char * d_res = scratch.coherent ? scratch.h_out : scratch.d_out;
  double * d_iou = reinterpret_cast<double *>(d_res), * d_area = d_iou + nn, * d_read = d_iou + 2 * nn, * d_score = d_iou + 3 * nn;
  int32_t * d_kept = reinterpret_cast<int32_t *>(d_iou + 4 * nn);
  if (scratch.coherent) {
    scratch.tickets += static_cast<unsigned int>(n);
    scratch.seq = scratch.seq == 0x7fffffff ? 1 : scratch.seq + 1;
    hipLaunchKernelGGL(k_decay, dim3((n + 3) / 4), dim3(256), 0, st, ref, d_boxes, n, d_masks, mask_words, *params, d_iou, d_area, d_read, d_score, d_kept,
                       scratch.d_ticket, scratch.tickets, scratch.h_flag, scratch.seq);
    if (hipGetLastError() != hipSuccess) {scratch.tickets -= static_cast<unsigned int>(n); return fail("launch failed");}
    volatile int32_t * flag = scratch.h_flag;
    uint64_t spins = 0;
    while (*flag != scratch.seq) {
      __builtin_ia32_pause();
      if ((++spins & 0x3fff) == 0) {
        const hipError_t e = hipStreamQuery(st);
        if (e == hipSuccess) {if (*flag == scratch.seq) {break;} scratch.release(); return fail("the stream drained without the result flag");}
        if (e != hipErrorNotReady) {scratch.release(); return fail("kernel failed");}
      }
    }
  } else {
    if (hipMemcpyAsync(scratch.d_in, scratch.h_in, in_bytes, hipMemcpyHostToDevice, st) != hipSuccess) {return fail("upload failed");}
    hipLaunchKernelGGL(k_decay, dim3((n + 3) / 4), dim3(256), 0, st, ref, d_boxes, n, d_masks, mask_words, *params, d_iou, d_area, d_read, d_score, d_kept,
                       (unsigned int *)nullptr, 0u, (int32_t *)nullptr, 0);
    if (hipMemcpyAsync(scratch.h_out, scratch.d_out, out_bytes, hipMemcpyDeviceToHost, st) != hipSuccess ||
      hipStreamSynchronize(st) != hipSuccess) {return fail("kernel or download failed");}
  }
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
  return kh::decay_scores(device, reference, n, candidates, nullptr, nullptr, 0, params, kept, iou, area_overlap, reading_overlap, scores);
}

}  // extern "C"
