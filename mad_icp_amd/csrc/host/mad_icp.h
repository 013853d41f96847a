// MADicp — point-to-plane Gauss-Newton registration of a scan's leaves against K keyframe trees.
// Mirrors the reference class (mad_icp/src/odometry/mad_icp.h:41-79): same constructor arguments, the
// public X_ / H_adder_ / b_adder_ that Pipeline reads (pipeline.cpp:195,223).  The per-round methods of the
// reference (resetAdders / update / updateState, mad_icp.cpp:43-117) are one call here — compute() — because
// the whole loop runs on the device without host round trips: one streamed submission (madicp_stream_submit /
// madicp_stream_collect), no device allocation, free or synchronisation per scan.
#pragma once
#include <functional>
#include <cstdint>
#include <vector>

#include "linalg.h"
#include "mad_tree.h"

namespace madicp_host {

class MADicp {
 public:
  MADicp(double min_ball, double rho_ker, double b_ratio, int num_threads);
  ~MADicp() = default;  // owns nothing on the device: the stream slots belong to the context
  MADicp(const MADicp&) = delete;
  MADicp& operator=(const MADicp&) = delete;

  // setMoving (mad_icp.cpp:53-55): the leaves of the current scan's tree, sensor frame.  With a tree that is (or is
  // about to be) resident the moving set is taken from its leaf records on the device — nothing is uploaded twice.
  void setMoving(MADtree& scan_tree);
  void setMoving(const ContainerType& leaf_means);
  void init(const Pose& moving_in_fixed) { X_ = moving_in_fixed; }  // mad_icp.cpp:57

  // n_iters rounds of {resetAdders; update(tree) for every fixed tree; updateState} on the device.
  // matched flags are those of the last round (cleared before it: pipeline.cpp:172-176).
  // truncated: the caller cut the loop short of MAX_ICP_ITS (Pipeline's realtime budget): matched flags are the OR over
  // the rounds that ran, like the reference's after an early break (pipeline.cpp:167-176)
  // `while_in_flight` (optional) runs on the calling thread between the submission of the registration and the wait for its
  // results — host work that has nothing to do with this registration (Pipeline: staging the next scan's look-ahead build)
  // `eager`: launch kernel by kernel, never through a hipGraph (a graph is instantiated per launch geometry and keyframe
  // count — milliseconds, and several of them while another stream of the context is busy: a frame with a look-ahead build
  // in flight must not pay that)
  void compute(const std::vector<MADtree*>& fixed, int n_iters, bool truncated = false,
               const std::function<void()>& while_in_flight = {}, bool eager = false);

  int numMoving() const { return L_; }
  int numMatched() const { return n_matched_; }

  Pose X_;
  double H_adder_[36];  // row-major
  double b_adder_[6];
  std::vector<uint8_t> matched_;
  uint64_t visits_ = 0;  // internal nodes visited by the last compute() (instrumentation)
  double phase_ms_[3] = {0., 0., 0.};  // last compute(): submission, the caller's work beside it, the wait for the result (instrumentation)

  double rho_ker_;  // as given by the caller (the sqrt of mad_icp.cpp:32 is taken inside the library)
  double min_ball_;
  double b_ratio_;
  int num_threads_;

 private:
  ContainerType moving_;            // host leaf means (when set from a container)
  MADtree* moving_tree_ = nullptr;  // or the resident tree whose leaves are the moving set
  int L_ = 0;
  int n_matched_ = 0;
};

}  // namespace madicp_host
