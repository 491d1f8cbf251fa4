// In-process multi-device candidate sharding: one scan matcher per device of a list, a
// batch of independent (query, base chain) matches dealt round robin to the members, every member driven by its own host
// thread, results written back in candidate order.
//
// Reference behaviour this has to preserve: MapperGraph::TryCloseLoop (lib/karto_sdk/src/Mapper.cpp:1500-1561) walks the
// candidate chains in order and accepts the FIRST that passes -- so the caller needs every result in candidate order, and
// the matches themselves are independent (each owns its correlation grid), which is why they shard without a collective:
// one 8-GPU node is one process's worth of devices.  The same device may be listed more than once (two members on one
// GPU): that is how a 1-GPU box exercises the N > 1 path.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <functional>
#include <string>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/karto_hip.h"

namespace kh
{
void set_pending_query_hook(std::function<void()> fn);       // matcher_seq.cpp (QueryHook, matcher_private.hpp)
void run_pending_query_hook();
void set_error(const std::string & s);
void host_parallel_for(size_t n, const std::function<void(size_t)> & fn);
int32_t matcher_max_batch(const kh_matcher * m);
}

// One persistent host thread per group member beyond the first (the calling thread drives member 0): a batch hands every
// member its share and waits.  A thread per call cost the mapper about a millisecond per closure, and an exception on a
// short-lived thread (bad_alloc from the share's vectors) called std::terminate.
class MemberWorker
{
public:
  MemberWorker() : thread_([this] {loop();}) {}
  ~MemberWorker()
  {
    {std::lock_guard<std::mutex> lk(mu_); quit_ = true;}
    cv_.notify_all();
    thread_.join();
  }
  void post(std::function<void()> task)
  {
    {std::lock_guard<std::mutex> lk(mu_); task_ = std::move(task); busy_ = true;}
    cv_.notify_all();
  }
  void wait()
  {
    std::unique_lock<std::mutex> lk(mu_);
    done_.wait(lk, [this] {return !busy_;});
  }
private:
  void loop()
  {
    for (;;) {
      std::function<void()> task;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [this] {return quit_ || static_cast<bool>(task_);});
        if (quit_) {return;}
        task.swap(task_);
      }
      task();                                     // (the task catches its own exceptions and reports them through its share)
      {std::lock_guard<std::mutex> lk(mu_); busy_ = false;}
      done_.notify_all();
    }
  }
  std::mutex mu_;
  std::condition_variable cv_, done_;
  std::function<void()> task_;
  bool busy_ = false, quit_ = false;
  std::thread thread_;
};

struct kh_matcher_group
{
  std::vector<kh_matcher *> members;
  std::vector<int32_t> devices;
  std::vector<std::unique_ptr<MemberWorker>> workers;     // workers[k - 1] drives member k
  std::mutex call_mu;                                      // one batch at a time per group (the workers hold one task each)
  int32_t max_batch = 0;
};

extern "C" {

int kh_matcher_group_create(double search_size, double resolution, double smear_deviation, double range_threshold,
  const int32_t * devices, int32_t n_devices, int32_t max_batch_per_member, kh_matcher_group ** out)
{
  if (!out || !devices || n_devices < 1 || max_batch_per_member < 1) {return KH_ERR_INVALID_ARG;}
  *out = nullptr;
  kh_matcher_group * g = new kh_matcher_group();
  g->max_batch = max_batch_per_member;
  for (int32_t k = 0; k < n_devices; ++k) {
    kh_matcher * m = nullptr;
    const int rc = kh_matcher_create(search_size, resolution, smear_deviation, range_threshold, devices[k], max_batch_per_member, &m);
    if (rc) {kh_matcher_group_destroy(g); return rc;}
    g->members.push_back(m);
    g->devices.push_back(devices[k]);
  }
  try {
    for (int32_t k = 1; k < n_devices; ++k) {g->workers.emplace_back(new MemberWorker());}
  } catch (const std::exception & e) {
    kh::set_error(std::string("kh_matcher_group_create: ") + e.what());
    kh_matcher_group_destroy(g);
    return KH_ERR_HIP;
  }
  *out = g;
  return KH_OK;
}

void kh_matcher_group_destroy(kh_matcher_group * g)
{
  if (!g) {return;}
  g->workers.clear();                            // joins the member threads
  for (kh_matcher * m : g->members) {kh_matcher_destroy(m);}
  delete g;
}

int kh_matcher_group_set_params(kh_matcher_group * g, const kh_match_params * p)
{
  if (!g || !p) {return KH_ERR_INVALID_ARG;}
  for (kh_matcher * m : g->members) {
    const int rc = kh_matcher_set_params(m, p);
    if (rc) {return rc;}
  }
  return KH_OK;
}

int32_t kh_matcher_group_size(const kh_matcher_group * g) {return g ? static_cast<int32_t>(g->members.size()) : 0;}

kh_matcher * kh_matcher_group_member(kh_matcher_group * g, int32_t index)
{
  return (g && index >= 0 && index < static_cast<int32_t>(g->members.size())) ? g->members[index] : nullptr;
}

int32_t kh_matcher_group_device(const kh_matcher_group * g, int32_t index)
{
  return (g && index >= 0 && index < static_cast<int32_t>(g->devices.size())) ? g->devices[index] : -1;
}

int kh_matcher_group_match_batch(kh_matcher_group * g, int32_t n, const kh_scan * queries, const kh_scan * base, const int32_t * base_begin,
  const double * const * base_device_points, int32_t do_penalize, int32_t do_refine, double * means, double * covs, double * responses,
  int32_t * status)
{
  if (!g || n < 0 || !queries || !base_begin || !means || !covs || !responses) {return KH_ERR_INVALID_ARG;}
  const int32_t nm = static_cast<int32_t>(g->members.size());
  struct Share
  {
    std::vector<int32_t> pairs;                  // candidate indices of this member, ascending
    int rc = KH_OK;
    std::string error;
  };
  std::vector<Share> shares(nm);
  for (int32_t i = 0; i < n; ++i) {shares[i % nm].pairs.push_back(i);}
  auto run_share = [&](int32_t k) {
    Share & sh = shares[k];
    const int32_t cap = g->max_batch;
    for (size_t at = 0; at < sh.pairs.size() && sh.rc == KH_OK; at += static_cast<size_t>(cap)) {
      const int32_t nb = static_cast<int32_t>(std::min<size_t>(cap, sh.pairs.size() - at));
      std::vector<kh_scan> q(nb), b;
      std::vector<int32_t> begin(nb + 1, 0);
      for (int32_t j = 0; j < nb; ++j) {
        const int32_t i = sh.pairs[at + j];
        q[j] = queries[i];
        for (int32_t t = base_begin[i]; t < base_begin[i + 1]; ++t) {
          kh_scan s = base[t];
          // a device copy is only good on the device it lives on: the caller's table names the copy of (scan, member)
          s.device_points_xy = base_device_points ? base_device_points[static_cast<size_t>(t) * nm + k] : (nm == 1 ? base[t].device_points_xy : nullptr);
          b.push_back(s);
        }
        begin[j + 1] = static_cast<int32_t>(b.size());
      }
      std::vector<double> mean(3 * static_cast<size_t>(nb)), cov(9 * static_cast<size_t>(nb)), resp(nb);
      std::vector<int32_t> st(nb, KH_OK);
      sh.rc = kh_matcher_match_batch(g->members[k], nb, q.data(), b.empty() ? nullptr : b.data(), begin.data(), do_penalize, do_refine,
          mean.data(), cov.data(), resp.data(), st.data());
      if (sh.rc) {sh.error = kh_last_error(); break;}          // the message is thread local: carried to the caller below
      for (int32_t j = 0; j < nb; ++j) {
        const int32_t i = sh.pairs[at + j];
        std::copy(mean.begin() + 3 * j, mean.begin() + 3 * j + 3, means + 3 * static_cast<size_t>(i));
        std::copy(cov.begin() + 9 * j, cov.begin() + 9 * j + 9, covs + 9 * static_cast<size_t>(i));
        responses[i] = resp[j];
        if (status) {status[i] = st[j];}
      }
    }
  };
  auto run = [&](int32_t k) {
    try {
      run_share(k);
    } catch (const std::exception & e) {         // e.g. bad_alloc from the share's vectors: an error code, not std::terminate
      shares[k].rc = KH_ERR_HIP;
      shares[k].error = std::string("kh_matcher_group_match_batch: ") + e.what();
    }
  };
  // member 0 works on the calling thread, the others on their persistent workers; members without work are left alone
  std::lock_guard<std::mutex> call_lock(g->call_mu);
  for (int32_t k = 1; k < nm; ++k) {
    if (!shares[k].pairs.empty()) {g->workers[k - 1]->post([&run, k] {run(k);});}
  }
  if (!shares[0].pairs.empty()) {run(0);}
  for (int32_t k = 1; k < nm; ++k) {
    if (!shares[k].pairs.empty()) {g->workers[k - 1]->wait();}
  }
  for (int32_t k = 0; k < nm; ++k) {
    if (shares[k].rc) {kh::set_error(shares[k].error); return shares[k].rc;}
  }
  return KH_OK;
}

// MapperGraph::TryCloseLoop's two matches for n candidate chains at once (Mapper.cpp:1515-1549): the coarse match on the loop
// matcher, the gate, and for the chains that pass the fine match of a temporary scan placed at the coarse pose (tmpScan,
// :1527-1534: same range readings, sensor pose = bestPose, point readings recomputed) on the sequential matcher.  The batch is
// cut into `pieces`; while the coarse matcher works on piece k + 1 a second host thread runs the fine matches of piece k on
// the fine matcher -- two handles, two streams, each used by one thread: the host halves of one stage (job tables, uploads,
// downloads, covariances) disappear behind the kernels of the other.
int kh_loop_closure_batch(kh_matcher * coarse, kh_matcher * fine, int32_t n, const kh_scan * queries, const kh_scan * base,
  const int32_t * base_begin, double min_angle, double angular_resolution, double minimum_response_coarse, double maximum_variance_coarse,
  int32_t pieces, double * coarse_means, double * coarse_covs, double * coarse_responses, int32_t * passed, double * fine_means,
  double * fine_covs, double * fine_responses)
{
  if (!coarse || !fine || n < 0 || !queries || !base_begin || !coarse_means || !coarse_covs || !coarse_responses || !passed || !fine_means ||
    !fine_covs || !fine_responses) {return KH_ERR_INVALID_ARG;}
  if (n == 0) {return KH_OK;}
  if (coarse == fine) {
    // a handle is not thread-safe (slots, staging buffers and stream are shared): the two stages run on two threads
    kh::set_error("kh_loop_closure_batch: the coarse and the fine matcher must be two handles");
    return KH_ERR_INVALID_ARG;
  }
  pieces = std::max(1, std::min(pieces, n));
  {
    const int32_t cap = std::min(kh::matcher_max_batch(coarse), kh::matcher_max_batch(fine));
    if (cap > 0 && (n + pieces - 1) / pieces > cap) {
      kh::set_error("kh_loop_closure_batch: a piece of " + std::to_string((n + pieces - 1) / pieces) + " candidates exceeds the matchers' max_batch of " +
        std::to_string(cap) + " (raise `pieces` or create the handles with a larger max_batch)");
      return KH_ERR_INVALID_ARG;
    }
  }
  std::vector<int32_t> bound(pieces + 1);
  for (int32_t k = 0; k <= pieces; ++k) {bound[k] = static_cast<int32_t>(static_cast<int64_t>(n) * k / pieces);}
  int fine_rc = KH_OK;
  std::string fine_error;
  // the gate and the fine matches of piece k (its coarse results are in)
  auto fine_piece = [&](int32_t k) {
    std::vector<int32_t> ids;
    for (int32_t i = bound[k]; i < bound[k + 1]; ++i) {
      // Mapper.cpp:1519-1521
      passed[i] = (coarse_responses[i] > minimum_response_coarse && coarse_covs[9 * i] < maximum_variance_coarse &&
        coarse_covs[9 * i + 4] < maximum_variance_coarse) ? 1 : 0;
      if (passed[i]) {ids.push_back(i);}
    }
    if (ids.empty()) {return;}
    const size_t m = ids.size();
    std::vector<std::vector<double>> points(m);
    std::vector<kh_scan> q(m), b;
    std::vector<int32_t> begin(m + 1, 0);
    for (size_t j = 0; j < m; ++j) {
      const int32_t i = ids[j];
      q[j] = queries[i];
      std::copy(coarse_means + 3 * i, coarse_means + 3 * i + 3, q[j].sensor_pose);
      points[j].resize(2 * static_cast<size_t>(q[j].n));
      q[j].points_xy = points[j].data(); q[j].device_points_xy = nullptr;
      for (int32_t t = base_begin[i]; t < base_begin[i + 1]; ++t) {b.push_back(base[t]);}
      begin[j + 1] = static_cast<int32_t>(b.size());
    }
    // LocalizedRangeScan::Update of the temporary scans (one libm sincos per reading: 15 us a scan, on the worker pool) as the
    // match's QueryHook: the fine matcher's rasteriser reads the temporary scans' POSES only, so their readings are made behind
    // its launches, while the GPU stamps the chains
    kh::set_pending_query_hook([&]() {
      kh::host_parallel_for(m, [&](size_t j) {
        kh_scan_points(q[j].ranges, q[j].n, q[j].sensor_pose, min_angle, angular_resolution, points[j].data());
      });
    });
    std::vector<double> mean(3 * m), cov(9 * m), resp(m);
    std::vector<int32_t> st(m, KH_OK);
    int rc = kh_matcher_match_batch(fine, static_cast<int32_t>(m), q.data(), b.empty() ? nullptr : b.data(), begin.data(), 0, 1,
        mean.data(), cov.data(), resp.data(), st.data());
    kh::run_pending_query_hook();          // (an argument check that refused the call before the hook was taken)
    for (size_t j = 0; j < m && rc == KH_OK; ++j) {rc = st[j];}
    if (rc) {fine_rc = rc; fine_error = kh_last_error(); return;}
    for (size_t j = 0; j < m; ++j) {
      const int32_t i = ids[j];
      std::copy(mean.begin() + 3 * j, mean.begin() + 3 * j + 3, fine_means + 3 * static_cast<size_t>(i));
      std::copy(cov.begin() + 9 * j, cov.begin() + 9 * j + 9, fine_covs + 9 * static_cast<size_t>(i));
      fine_responses[i] = resp[j];
    }
  };
  auto coarse_piece = [&](int32_t k) -> int {
    const int32_t i0 = bound[k], nk = bound[k + 1] - bound[k];
    std::vector<int32_t> st(nk, KH_OK);
    int rc = kh_matcher_match_batch(coarse, nk, queries + i0, base, base_begin + i0, 0, 0, coarse_means + 3 * static_cast<size_t>(i0),
        coarse_covs + 9 * static_cast<size_t>(i0), coarse_responses + i0, st.data());
    for (int32_t j = 0; j < nk && rc == KH_OK; ++j) {rc = st[j];}
    return rc;
  };
  if (pieces == 1) {
    // one piece: nothing to overlap -- both stages on the calling thread (round 6: a thread was created, woken through a condition
    // variable and joined per call, ~0.15 ms of the batch)
    const int rc = coarse_piece(0);
    if (rc) {return rc;}
    fine_piece(0);
    if (fine_rc) {kh::set_error(fine_error); return fine_rc;}
    return KH_OK;
  }
  std::mutex mu;
  std::condition_variable cv;
  int32_t ready = 0;                 // pieces whose coarse results are in
  bool abort = false;
  std::thread consumer([&] {
    for (int32_t k = 0; k < pieces && fine_rc == KH_OK; ++k) {
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] {return ready > k || abort;});
        if (abort) {return;}
      }
      fine_piece(k);
    }
  });
  int rc = KH_OK;
  for (int32_t k = 0; k < pieces && rc == KH_OK; ++k) {
    rc = coarse_piece(k);
    {
      std::lock_guard<std::mutex> lk(mu);
      if (rc) {abort = true;} else {ready = k + 1;}
    }
    cv.notify_all();
  }
  consumer.join();
  if (rc) {return rc;}
  if (fine_rc) {kh::set_error(fine_error); return fine_rc;}
  return KH_OK;
}

}  // extern "C"
