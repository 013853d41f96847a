#include "mad_icp.h"

#include <chrono>

#include <cstring>

#include "device.h"

namespace madicp_host {

MADicp::MADicp(double min_ball, double rho_ker, double b_ratio, int num_threads)
  : rho_ker_(rho_ker), min_ball_(min_ball), b_ratio_(b_ratio), num_threads_(num_threads) {
  X_ = Pose::identity();
  std::memset(H_adder_, 0, sizeof(H_adder_));
  std::memset(b_adder_, 0, sizeof(b_adder_));
}

void MADicp::setMoving(const ContainerType& leaf_means) {
  if (leaf_means.empty()) throw std::invalid_argument("MADicp::setMoving: no leaves");
  moving_ = leaf_means;
  moving_tree_ = nullptr;
  L_ = static_cast<int>(moving_.size());
  n_matched_ = 0;
  matched_.assign(L_, 0);
}

void MADicp::setMoving(MADtree& scan_tree) {
  moving_.clear();
  moving_tree_ = &scan_tree;
  L_ = scan_tree.numLeaves();
  n_matched_ = 0;
  matched_.assign(L_, 0);
}

void MADicp::compute(const std::vector<MADtree*>& fixed, int n_iters, bool truncated, const std::function<void()>& while_in_flight,
                     bool eager) {
  if (L_ <= 0) throw std::runtime_error("MADicp::compute: setMoving was not called");
  if (fixed.empty()) throw std::runtime_error("MADicp::compute: no fixed tree");
  if (n_iters < 1) return;
  DeviceLock lock(Device::mutex());
  madicp_ctx* ctx = Device::ctx();
  // A loop the caller cut short (Pipeline's realtime budget): the reference only resets matched_ in iteration
  // MAX_ICP_ITS - 1, so after an early break the flags are the OR of every round that ran (pipeline.cpp:167-176); and a
  // round count that changes from frame to frame must not instantiate hipGraphs inside a time-critical frame.
  // The context is shared by everything in the process that goes through Device::ctx(): what this call changes it puts
  // back to what it FOUND (another user's use_graph = 0 for profiling survives a truncated or eager frame).
  struct Restore {
    madicp_ctx* c;
    int64_t match_all = -1, use_graph = -1;  // -1: untouched
    void set(const char* key, int64_t value, int64_t& saved) {
      int64_t was = 0;
      check(madicp_ctx_get_option(c, key, &was), "madicp_ctx_get_option");
      if (was == value) return;
      check(madicp_ctx_set_option(c, key, value), "madicp_ctx_set_option");
      saved = was;
    }
    ~Restore() {
      if (match_all >= 0) madicp_ctx_set_option(c, "match_all_rounds", match_all);
      if (use_graph >= 0) madicp_ctx_set_option(c, "use_graph", use_graph);
    }
  } restore{ctx};
  if (truncated) restore.set("match_all_rounds", 1, restore.match_all);
  if (truncated || eager) restore.set("use_graph", 0, restore.use_graph);
  std::vector<int> ids;
  ids.reserve(fixed.size());
  for (MADtree* t : fixed) ids.push_back(t->deviceId());
  double X[12];
  std::memcpy(X, X_.R, sizeof(X_.R));
  std::memcpy(X + 9, X_.t, sizeof(X_.t));
  const madicp_icp_params p{min_ball_, rho_ker_, b_ratio_};
  int ticket = -1;
  const auto t0 = std::chrono::steady_clock::now();
  auto since = [](std::chrono::steady_clock::time_point from) {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - from).count();
  };
  if (moving_tree_) {
    check(madicp_stream_submit_tree(ctx, moving_tree_->deviceId(), ids.data(), static_cast<int>(ids.size()), X, &p, n_iters,
                                    &ticket),
          "madicp_stream_submit_tree");
  } else {
    check(madicp_stream_submit(ctx, moving_.front().data(), L_, ids.data(), static_cast<int>(ids.size()), X, &p, n_iters,
                               &ticket),
          "madicp_stream_submit");
  }
  phase_ms_[0] = since(t0);
  const auto t1 = std::chrono::steady_clock::now();
  if (while_in_flight) {
    try {
      while_in_flight();
    } catch (...) {  // the registration is in flight: its slot must be collected whatever the caller's work did
      int32_t nm = 0;
      madicp_stream_collect(ctx, ticket, X, H_adder_, b_adder_, matched_.data(), &nm, &visits_);
      throw;
    }
  }
  phase_ms_[1] = since(t1);
  const auto t2 = std::chrono::steady_clock::now();
  int32_t n_matched = 0;
  check(madicp_stream_collect(ctx, ticket, X, H_adder_, b_adder_, matched_.data(), &n_matched, &visits_),
        "madicp_stream_collect");
  phase_ms_[2] = since(t2);
  n_matched_ = n_matched;
  std::memcpy(X_.R, X, sizeof(X_.R));
  std::memcpy(X_.t, X + 9, sizeof(X_.t));
}

}  // namespace madicp_host
