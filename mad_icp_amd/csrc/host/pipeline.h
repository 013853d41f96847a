// Pipeline — the odometry frame step.  Same constructor, same public methods and the same by-value
// compute(stamp, cloud) as the reference class (mad_icp/src/odometry/pipeline.h:45-103), so
// apps/cpp_runners/bin_runner.cpp:106-186 and the pybind module keep working unchanged.
//
// What moved: the data association + Gauss-Newton loop (pipeline.cpp:166-193) runs on the MI355X through
// libmadicp_hip.so.  Every scan's MAD-tree is uploaded once, when it is built (asynchronously, on the library's copy
// stream), and stays in HBM while the frame is in the 10-frame window or a keyframe: the scan's moving leaves are taken
// from the resident tree, applyTransform runs on the device (pipeline.cpp:224), promotion to keyframe
// (pipeline.cpp:234-253) is a pointer move, and leaving the window / eviction (pipeline.cpp:229-232,254-257) returns
// the buffers to the library's pool without a synchronisation.  What stayed on the CPU: tree construction of the
// incoming scan, deskew, the constant-velocity predictor, keyframe selection.
//
// Additive: prefetch(next_cloud) starts building the NEXT scan's tree on another thread — the build does not depend
// on the current pose (pipeline.cpp:140-141 builds in the sensor frame).  compute() itself is synchronous, so what the
// build overlaps is whatever runs until the compute() of that scan.  Up to three look-aheads are kept beside the scan being
// consumed, each matched to its scan by size and end points, so a caller that reads ahead (bin_runner / the launcher on a
// dataset) issues prefetch(i + d) before compute(i).  Averaged over a drive at 120 k points the frame takes 2.5 ms without,
// 1.67 ms with d = 1, 1.41 ms with d = 2, 1.29 ms with d = 3 (tools/lookahead_probe.py), with the reference's own trees bit
// for bit.  Individual frames are bimodal — one that waits about a build's length, then d quick ones of 0.7 ms: a build's
// critical path (2.1-2.6 ms inside the pipeline) does not get shorter with more threads, so depth d buys d builds per build
// latency until the 16 threads are busy.  With the
// device front-end on, prefetch(i + 1) before compute(i) hands over the NEXT scan; compute(i) starts its construction on the
// library's build stream (madicp_tree_build_begin) as soon as its own registration is submitted: the host side of that build
// (staging, launches, the wait for the leaf count) is hidden behind the registration and the frame becomes device-bound
// (0.90 -> 0.64 ms; the kernels of the two streams interleave, profiles/r3_o_lookahead_overlap.md); one look-ahead there.  For
// deskewed datasets the tree needs the previous poses: prefetch() then computes the pose-independent half of deskew ahead —
// the azimuth of every point and their order (deskew.h).
#pragma once
#include <array>
#include <cstddef>
#include <deque>
#include <future>
#include <memory>
#include <vector>

#include "deskew.h"
#include "linalg.h"
#include "mad_icp.h"
#include "mad_tree.h"
#include "types.h"
#include "vel_estimator.h"

namespace madicp_host {

// tools/constants.h:31-35
static constexpr int CHUNKS = 1024;
static constexpr int SMOOTHING_T = 10;
static constexpr int MAX_ICP_ITS = 15;
static constexpr int FRAME_WINDOW = 10;

struct Frame {  // tools/frame.h:37-52; the tree is owned here
  Pose frame_to_map_ = Pose::identity();
  std::unique_ptr<MADtree> tree_;
  double stamp_ = 0.;
  double weight_ = 0.;
  int frame_ = 0;
};

class Pipeline {
 public:
  Pipeline(double sensor_hz, bool deskew, double b_max, double rho_ker, double p_th, double b_min, double b_ratio,
           int num_keyframes, int num_threads, bool realtime);
  ~Pipeline();

  const Matrix4d currentPose() const { return toMatrix(frame_to_map_); }
  const std::vector<Matrix4d> trajectory() const;
  const Matrix4d keyframePose() const { return toMatrix(keyframe_to_map_); }
  bool isInitialized() const { return is_initialized_; }
  size_t currentID() const { return seq_; }
  size_t keyframeID() const { return seq_keyframe_; }
  bool isMapUpdated() { return is_map_updated_; }
  const ContainerType currentLeaves();
  const ContainerType modelLeaves();
  void compute(const double& curr_stamp, ContainerType curr_cloud_mem);

  // additive (not in the reference): start building the tree of the scan that the NEXT compute() will be given
  void prefetch(ContainerType next_cloud);
  // additive: the same two calls on a VIEW of the caller's points (n x 3 doubles, only read during the call) — what the
  // Python bindings use, so that a frame does not begin with a 3 MB allocation + copy of its by-value argument
  void computeView(const double& curr_stamp, const Vector3d* curr_cloud, size_t n);
  void prefetchView(const Vector3d* next_cloud, size_t n);

  // additive: the device front-end.  When on, compute() uploads the scan once and deskew
  // (pipeline.cpp:79-123) and MADtree::build (mad_tree.cpp:47-130) run on the MI355X; the tree never exists on the host
  // unless currentLeaves() / modelLeaves() ask for it.  Default: ON (round 5: for deskew = false; round 6: for deskewed
  // datasets too); the MAD_ICP_GPU_BUILD environment variable ("0" = the host builder, whose trees are the reference's bit
  // for bit) and this call override it.  Device-built trees have the host builder's topology, member order and leaf
  // representatives and differ from host-built ones in the last bits of their larger nodes (mad_icp_amd/csrc/hip/
  // tree_build.hip.h): without deskew poses stay within ~1e-12 m of the oracle pipeline's over full-size drives; with deskew
  // ONE frame from the oracle's state is the oracle's frame to 1e-5 (tests/test_gpu_deskew_one_step.py), and a drive stays
  // inside the envelope the reference shows against itself (the compensated cloud depends on the previous poses' last bits
  // and tree construction is chaotic in its input: tests/test_gpu_frontend_oracle.py).
  void setDeviceFrontEnd(bool on) {
    if (!on) dropDeviceLookAhead();
    device_frontend_ = on;
  }
  bool deviceFrontEnd() const { return device_frontend_; }
  // additive: one frame straight from sensor records — float32 (x, y, z, intensity ...) `stride_floats` apart, range
  // filter and optional KITTI correction as in apps/cpp_runners/bin_runner.cpp:126-166 — ingest, deskew, build and
  // registration all on the device (implies the device front-end for this frame)
  void computeRecords(const double& curr_stamp, const float* records, size_t n_records, int stride_floats, double min_range,
                      double max_range, bool kitti_correction);

  // instrumentation (not in the reference)
  double lastInliersRatio() const { return last_inliers_ratio_; }
  int lastRounds() const { return last_rounds_; }  // GN rounds the last frame ran (realtime = true can cut them short)
  // Test seam for realtime = true: with pre_ms >= 0 the wall clock of the budget rule (pipeline.cpp:160-169) is replaced by
  // a model — this frame's preprocessing took pre_ms, a GN round takes round_ms — so the round count is a deterministic
  // function that can be held to the reference's per-round check (tests/test_boundary.py); pre_ms < 0: the wall clock again.
  void setTimingForTest(double pre_ms, double round_ms) { virtual_pre_ms_ = pre_ms; virtual_round_ms_ = round_ms; }
  double lastIcpMs() const { return last_icp_ms_; }
  double lastBuildMs() const { return last_build_ms_; }
  // the last frame's registration split: submission, the look-ahead begun beside it, the wait for the result (ms)
  std::array<double, 3> lastIcpPhasesMs() const { return {icp_.phase_ms_[0], icp_.phase_ms_[1], icp_.phase_ms_[2]}; }
  size_t numKeyframes() const { return keyframes_.size(); }
  size_t lookAheadHits() const { return look_ahead_hits_; }  // frames whose tree had been built ahead (prefetch)

  static Matrix4d toMatrix(const Pose& p);

 protected:
  void initialize(const double& curr_stamp, ContainerType& curr_cloud);
  void deskew(ContainerType& curr_cloud, const Pose& T_prev, const Pose& T_now, const DeskewOrder* prep = nullptr);
  void naiveVelocity(const Pose& T_prev, const Pose& T_now, double* vel6) const;  // pipeline.cpp:82-86
  std::unique_ptr<MADtree> buildOnDevice(int cloud_id);  // deskew (if due) + build + release of the cloud
  void computeWithTree(const double& curr_stamp, std::unique_ptr<MADtree> current_tree, ContainerType* curr_cloud, double t_pre);

  MADicp icp_;
  VelEstimator vel_estimator_;
  Pose frame_to_map_;
  Pose keyframe_to_map_;
  double current_velocity_[6];
  std::deque<std::unique_ptr<Frame>> keyframes_;
  std::deque<std::unique_ptr<Frame>> frames_;
  std::vector<Pose> trajectory_;
  MADtree* current_tree_view_ = nullptr;  // the last scan's tree (owned by a Frame in frames_ / keyframes_)
  size_t current_num_leaves_ = 0;
  // what a look-ahead result is matched to its scan by: size, end points and a digest of a strided sample of the points
  struct DevKey {
    size_t n = 0;
    Vector3d first{}, last{};
    uint64_t digest = 0;
    static DevKey of(const ContainerType& c);
    bool matches(const ContainerType& c) const;
    static DevKey of(const Vector3d* c, size_t n);
    bool matches(const Vector3d* c, size_t n) const;
  };
  // look-ahead builds: up to three scans ahead of the one being consumed (kMaxLookAhead), each matched to its scan by its
  // key (a prefetch(i + 1) issued BEFORE compute(i) must not cost scan i its tree)
  struct Prefetched {
    DevKey key;
    std::future<LinearTree> tree;
  };
  static constexpr size_t kMaxLookAhead = 4;  // scan i being consumed, up to three more building
  std::deque<Prefetched> prefetched_;
  void waitPrefetched();  // every look-ahead build has finished (their trees stay available)
  // device front-end: ONE construction in flight on the library's build stream (`dev_pending_`), and the tree of the scan
  // before it, collected when the next look-ahead was begun (`dev_ready_`) — with the call order prefetch(i + 1),
  // compute(i) the tree of scan i is collected at prefetch(i + 1) and scan i + 1 is built while scan i registers
  // deskewed datasets: the tree needs the two previous poses, but the azimuth order of the scan does not — that half of
  // Pipeline::deskew (atan2 per point, the sort: most of a deskewed frame on the host) is what prefetch() computes ahead
  struct DeskewAhead {
    DevKey key;
    std::future<DeskewOrder> order;
  };
  std::deque<DeskewAhead> deskew_ahead_;
  unsigned dev_pending_ = 0;  // ticket (MADtree::beginDeviceBuild), 0: none
  ContainerType dev_next_cloud_;  // the scan prefetch() was given, staged and begun by compute() WHILE its registration is in
                                  // flight (the host side of a begin — 3 MB into pinned memory, ~60 launches — is a third of
                                  // a millisecond that would otherwise sit in front of the registration)
  bool dev_next_staged_ = false;  // dev_next_cloud_ holds a scan that has not been begun (the vector keeps its memory between scans)
  void beginStagedLookAhead();
  DevKey dev_pending_key_, dev_ready_key_;
  std::unique_ptr<MADtree> dev_ready_;
  void collectDeviceLookAhead();  // dev_pending_ -> dev_ready_
  void dropDeviceLookAhead(bool staged_too = true);  // forget both (and the scan staged for the next frame)
  double virtual_pre_ms_ = -1.0, virtual_round_ms_ = 0.0;
  int last_rounds_ = 0;
  double round_ms_estimate_ = 0.05;  // device time of one GN round, from the previous frame (realtime budget)
  bool device_frontend_ = false;
  bool deskew_, realtime_;
  int num_keyframes_, num_threads_, max_parallel_levels_;
  double sensor_hz_, b_max_, p_th_, b_min_;
  size_t seq_ = 0;
  size_t seq_keyframe_ = 0;
  bool is_initialized_ = false;
  bool is_map_updated_ = false;
  float loop_time_;
  double last_inliers_ratio_ = 0., last_icp_ms_ = 0., last_build_ms_ = 0.;
  size_t look_ahead_hits_ = 0;
};

}  // namespace madicp_host

// the reference declares these names at global scope; keep them reachable the same way
using madicp_host::ContainerType;
using madicp_host::Pipeline;
