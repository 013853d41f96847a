#include "pipeline.h"

#include "deskew.h"
#include "device.h"
#include "task_pool.h"

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <limits>
#include <utility>

namespace madicp_host {

namespace {
double now_ms() {
  using clk = std::chrono::steady_clock;
  return std::chrono::duration<double, std::milli>(clk::now().time_since_epoch()).count();
}
}  // namespace

Matrix4d Pipeline::toMatrix(const Pose& p) {
  Matrix4d M;
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) M(r, c) = p.R[3 * r + c];
    M(r, 3) = p.t[r];
  }
  M(3, 0) = M(3, 1) = M(3, 2) = 0.0;
  M(3, 3) = 1.0;
  return M;
}

Pipeline::Pipeline(double sensor_hz, bool deskew, double b_max, double rho_ker, double p_th, double b_min,
                   double b_ratio, int num_keyframes, int num_threads, bool realtime)
  : icp_(b_max, rho_ker, b_ratio, num_threads),
    vel_estimator_(sensor_hz),
    deskew_(deskew),
    realtime_(realtime),
    num_keyframes_(num_keyframes),
    num_threads_(num_threads),
    sensor_hz_(sensor_hz),
    b_max_(b_max),
    p_th_(p_th),
    b_min_(b_min) {
  frame_to_map_ = Pose::identity();
  keyframe_to_map_ = Pose::identity();
  std::memset(current_velocity_, 0, sizeof(current_velocity_));
  loop_time_ = (1. / sensor_hz_) * 1000;
  max_parallel_levels_ = static_cast<int>(std::log2(num_threads));  // pipeline.cpp:64
  TaskPool::instance().set_limit(num_threads);  // omp_set_num_threads(num_threads), pipeline.cpp:65
  // The device front-end is the DEFAULT: an unmodified caller gets deskew (where the dataset asks for it) and MAD-tree
  // construction on the MI355X.  Round 5 made it so for deskew = false (tests/test_gpu_frontend_oracle.py holds that path to
  // the oracle pipeline directly); round 6 for deskew = true too: ONE frame — deskew, build, registration — from the oracle's
  // state is the oracle's frame to 1e-5 m / 1e-5 rad (tests/test_gpu_deskew_one_step.py), and over a drive neither path is
  // bit-stable anyway (the reference is not against itself: tests/envelope.py), so nothing but 3 ms per frame spoke for the
  // host path there.  MAD_ICP_GPU_BUILD=0 keeps the host builder (the reference's trees bit for bit); setDeviceFrontEnd()
  // overrides the environment.
  device_frontend_ = true;
  if (const char* e = std::getenv("MAD_ICP_GPU_BUILD")) {
    if (e[0] == '1') device_frontend_ = true;
    if (e[0] == '0') device_frontend_ = false;
  }
}

void Pipeline::waitPrefetched() {
  for (Prefetched& p : prefetched_)
    if (p.tree.valid()) p.tree.wait();
  for (DeskewAhead& a : deskew_ahead_)
    if (a.order.valid()) a.order.wait();
}

// A look-ahead result belongs to the scan it was computed from: size, end points and a digest of a strided sample of the
// points (up to 1024 of them, every coordinate's bit pattern) — a re-filtered, jittered or padded copy with the same size
// and end points does not get another scan's tree.
static uint64_t cloud_digest(const Vector3d* c, size_t n) {
  const size_t step = std::max<size_t>(1, n / 1024);
  uint64_t h = 0x9e3779b97f4a7c15ull ^ n;
  for (size_t i = 0; i < n; i += step) {
    uint64_t w[3];
    std::memcpy(w, c[i].data(), 24);
    for (int k = 0; k < 3; ++k) {
      h ^= w[k] + 0x9e3779b97f4a7c15ull + (h << 6) + (h >> 2);
      h = (h << 13) | (h >> 51);
    }
  }
  return h;
}
Pipeline::DevKey Pipeline::DevKey::of(const Vector3d* c, size_t n) {
  DevKey k;
  k.n = n;
  if (n) {
    k.first = c[0];
    k.last = c[n - 1];
    k.digest = cloud_digest(c, n);
  }
  return k;
}
bool Pipeline::DevKey::matches(const Vector3d* c, size_t count) const {
  return n == count && n > 0 && std::memcmp(first.data(), c[0].data(), 24) == 0 &&
         std::memcmp(last.data(), c[n - 1].data(), 24) == 0 && digest == cloud_digest(c, n);
}
Pipeline::DevKey Pipeline::DevKey::of(const ContainerType& c) { return of(c.data(), c.size()); }
bool Pipeline::DevKey::matches(const ContainerType& c) const { return matches(c.data(), c.size()); }

void Pipeline::collectDeviceLookAhead() {
  if (!dev_pending_) return;
  const unsigned ticket = dev_pending_;
  dev_pending_ = 0;
  dev_ready_.reset();  // (an older collected tree whose scan never came)
  dev_ready_ = MADtree::collectDeviceBuild(ticket);  // (null: cancelled by a synchronous build of another Pipeline)
  dev_ready_key_ = dev_pending_key_;
}

void Pipeline::beginStagedLookAhead() {
  if (!dev_next_staged_) return;
  dev_next_staged_ = false;  // (the buffer keeps its memory for the next scan: see prefetchView)
  collectDeviceLookAhead();  // (the one slot: whatever was in flight is collected first)
  dev_pending_ = MADtree::beginDeviceBuild(dev_next_cloud_, b_max_, b_min_);
  if (!dev_pending_) return;
  dev_pending_key_ = DevKey::of(dev_next_cloud_);
}

void Pipeline::dropDeviceLookAhead(bool staged_too) {
  if (staged_too) dev_next_staged_ = false;
  if (dev_pending_) {
    MADtree::cancelDeviceBuild(dev_pending_);
    dev_pending_ = 0;
  }
  dev_ready_.reset();
}

Pipeline::~Pipeline() {  // Frames own their trees; trees release their HBM copies
  waitPrefetched();
  try {
    dropDeviceLookAhead();
  } catch (...) {
  }
}

const std::vector<Matrix4d> Pipeline::trajectory() const {
  std::vector<Matrix4d> out;
  out.reserve(trajectory_.size());
  for (const Pose& p : trajectory_) out.push_back(toMatrix(p));
  return out;
}

// pipeline.cpp:82-86
void Pipeline::naiveVelocity(const Pose& T_prev, const Pose& T_now, double* vel) const {
  naive_velocity(T_prev, T_now, sensor_hz_, vel);
}

// pipeline.cpp:79-123 — motion compensation in 1024 azimuth chunks, host version (deskew.h; the device front-end runs
// madicp_cloud_deskew instead).  `prep`: the pose-independent half, when prefetch() computed it ahead.
void Pipeline::deskew(ContainerType& cloud, const Pose& T_prev, const Pose& T_now, const DeskewOrder* prep) {
  deskew_cloud(cloud, T_prev, T_now, sensor_hz_, prep, nullptr);
}

// pipeline.cpp:267-284
void Pipeline::initialize(const double& curr_stamp, ContainerType& cloud) {
  auto frame = std::make_unique<Frame>();
  frame->frame_ = int(seq_);
  frame->frame_to_map_ = frame_to_map_;
  frame->stamp_ = curr_stamp;
  frame->tree_ = std::make_unique<MADtree>(std::move(cloud), b_max_, b_min_, max_parallel_levels_);
  frame->tree_->deviceId();  // first keyframe: resident from now on
  keyframes_.push_back(std::move(frame));
  trajectory_.push_back(Pose::identity());
  is_initialized_ = true;
  is_map_updated_ = true;
  seq_++;
}

// The scan as a VIEW (the bindings' entry point: pybind hands a by-value ContainerType over as a fresh 3 MB allocation + copy
// per call, and fresh pages cost ~1 us each — 0.35 ms per cloud, twice per look-ahead frame, which is what the "look-ahead
// cliff" of rounds 3-5 was: profiles/r5_u_lookahead_canary.md).  The memory is only read during the call.
void Pipeline::prefetchView(const Vector3d* next_cloud, size_t n) {
  if (!next_cloud || n == 0) return;
  if (device_frontend_) {
    // the tree is built on the GPU — a host build would only compete for the CPU — and the look-ahead is the library's:
    // collect the construction in flight (the scan about to be consumed), start this one beside the coming registration
    if (deskew_) return;  // (the tree needs the previous pose)
    // begun by the next compute(), behind its registration's submission; assign() into a buffer that lives as long as the
    // Pipeline: after the first frames no allocation, no fresh pages
    dev_next_cloud_.assign(next_cloud, next_cloud + n);
    dev_next_staged_ = true;
    if (!is_initialized_ || (!dev_pending_ && !dev_ready_)) beginStagedLookAhead();  // (nothing to hide behind yet)
    return;
  }
  prefetch(ContainerType(next_cloud, next_cloud + n));
}

void Pipeline::prefetch(ContainerType next_cloud) {
  if (next_cloud.empty()) return;
  if (device_frontend_) {
    prefetchView(next_cloud.data(), next_cloud.size());
    return;
  }
  // with deskew the tree is built from the motion-compensated cloud, which needs the pose of the frame before it — but the
  // azimuth of every point and their order do not (deskew.h): that half is started now
  if (deskew_ && is_initialized_) {
    while (deskew_ahead_.size() >= kMaxLookAhead) {
      if (deskew_ahead_.front().order.valid()) deskew_ahead_.front().order.wait();
      deskew_ahead_.pop_front();
    }
    DeskewAhead a;
    a.key = DevKey::of(next_cloud);
    a.order = std::async(std::launch::async, [cloud = std::move(next_cloud)]() { return deskew_order(cloud); });
    deskew_ahead_.push_back(std::move(a));
    return;
  }
  while (prefetched_.size() >= kMaxLookAhead) {  // the oldest one makes room (its build is waited for)
    if (prefetched_.front().tree.valid()) prefetched_.front().tree.wait();
    prefetched_.pop_front();
  }
  Prefetched p;
  p.key = DevKey::of(next_cloud);
  const double b_max = b_max_, b_min = b_min_;
  const int levels = max_parallel_levels_;
  p.tree = std::async(std::launch::async, [cloud = std::move(next_cloud), b_max, b_min, levels]() mutable {
    return build_tree(cloud.front().data(), static_cast<int64_t>(cloud.size()), b_max, b_min, levels);
  });
  prefetched_.push_back(std::move(p));
}

// the device front-end for one frame: the cloud is resident; deskew when the reference would (pipeline.cpp:138-139),
// build, hand the cloud's buffer back
std::unique_ptr<MADtree> Pipeline::buildOnDevice(int cloud_id) {
  DeviceLock lock(Device::mutex());
  madicp_ctx* ctx = Device::ctx();
  MADtree::cancelDeviceBuild(0);  // deskew and build need the builder's scratch: a look-ahead of another Pipeline gives way
  std::unique_ptr<MADtree> tree;
  try {
    if (is_initialized_ && deskew_ && trajectory_.size() > 1) {
      double vel[6];
      naiveVelocity(trajectory_[trajectory_.size() - 2], trajectory_[trajectory_.size() - 1], vel);
      check(madicp_cloud_deskew(ctx, cloud_id, vel, sensor_hz_, nullptr), "madicp_cloud_deskew");
    }
    tree = std::make_unique<MADtree>(MADtree::DeviceCloud{cloud_id}, b_max_, b_min_);
  } catch (...) {
    madicp_cloud_release(ctx, cloud_id);
    throw;
  }
  check(madicp_cloud_release(ctx, cloud_id), "madicp_cloud_release");
  return tree;
}

void Pipeline::computeRecords(const double& curr_stamp, const float* records, size_t n_records, int stride_floats,
                              double min_range, double max_range, bool kitti_correction) {
  is_map_updated_ = false;
  if (!records || n_records == 0) throw std::invalid_argument("Pipeline::computeRecords: no records");
  const double t_pre = now_ms();
  waitPrefetched();
  dropDeviceLookAhead();          // (ingest shares the builder's scratch ...
  MADtree::cancelDeviceBuild(0);  //  ... with every Pipeline of the process)
  int cloud_id = -1;
  {
    DeviceLock lock(Device::mutex());
    int64_t kept = 0;
    check(madicp_cloud_ingest_f32(Device::ctx(), records, static_cast<int64_t>(n_records), stride_floats, min_range, max_range,
                                  kitti_correction ? 1 : 0, &cloud_id, &kept),
          "madicp_cloud_ingest_f32");
  }
  computeWithTree(curr_stamp, buildOnDevice(cloud_id), nullptr, t_pre);
}

// pipeline.cpp:125-265, the device front-end's half: the scan is only READ (key of a look-ahead, or the upload), so a view does
void Pipeline::computeView(const double& curr_stamp, const Vector3d* curr_cloud, size_t n) {
  if (!curr_cloud || n == 0) throw std::invalid_argument("Pipeline::compute: empty cloud");
  if (!device_frontend_) {  // the host builder takes the points over: it needs its own copy
    compute(curr_stamp, ContainerType(curr_cloud, curr_cloud + n));
    return;
  }
  is_map_updated_ = false;
  // a tree built ahead for exactly this scan?
  std::unique_ptr<MADtree> current_tree;
  const double t_pre = now_ms();
  waitPrefetched();
  if (dev_ready_ && dev_ready_key_.matches(curr_cloud, n)) {
    current_tree = std::move(dev_ready_);  // collected when the next look-ahead was begun
  } else if (dev_pending_ && dev_pending_key_.matches(curr_cloud, n)) {
    collectDeviceLookAhead();
    current_tree = std::move(dev_ready_);
  }
  if (current_tree) ++look_ahead_hits_;
  if (!current_tree) {
    // Nothing looked ahead for THIS scan.  A construction in flight is then most likely for the NEXT one (a caller whose
    // first scan came without a prefetch stays one ahead from there on): it is collected and kept — the synchronous build
    // below needs the builder's scratch, so it has to be finished either way — not thrown away.
    collectDeviceLookAhead();
    int cloud_id = -1;
    {
      DeviceLock lock(Device::mutex());
      check(madicp_cloud_upload(Device::ctx(), curr_cloud[0].data(), static_cast<int64_t>(n), &cloud_id), "madicp_cloud_upload");
    }
    current_tree = buildOnDevice(cloud_id);
  }
  computeWithTree(curr_stamp, std::move(current_tree), nullptr, t_pre);
}

// pipeline.cpp:125-265
void Pipeline::compute(const double& curr_stamp, ContainerType curr_cloud) {
  if (curr_cloud.empty()) throw std::invalid_argument("Pipeline::compute: empty cloud");
  if (device_frontend_) {
    computeView(curr_stamp, curr_cloud.data(), curr_cloud.size());
    return;
  }
  is_map_updated_ = false;
  // a tree built ahead for exactly this scan?
  std::unique_ptr<MADtree> current_tree;
  const double t_pre = now_ms();
  if (!prefetched_.empty() && !(deskew_ && is_initialized_ && trajectory_.size() > 1)) {
    // the look-ahead built for exactly this scan, if there is one; older look-aheads are for scans that never came
    for (size_t q = 0; q < prefetched_.size(); ++q) {
      const Prefetched& p = prefetched_[q];
      if (p.key.matches(curr_cloud)) {
        for (size_t d = 0; d < q; ++d) {
          if (prefetched_.front().tree.valid()) prefetched_.front().tree.wait();
          prefetched_.pop_front();
        }
        LinearTree built = prefetched_.front().tree.get();  // (waits for the builder thread)
        prefetched_.pop_front();
        current_tree = std::make_unique<MADtree>(std::move(built));
        ++look_ahead_hits_;
        break;
      }
    }
  }
  computeWithTree(curr_stamp, std::move(current_tree), &curr_cloud, t_pre);
}

// the frame step once the scan's tree exists (or, host path, is still to be built from *cloud)
void Pipeline::computeWithTree(const double& curr_stamp, std::unique_ptr<MADtree> current_tree, ContainerType* cloud,
                               double t_pre) {
  if (!is_initialized_) {
    if (current_tree) {
      auto frame = std::make_unique<Frame>();
      frame->frame_ = int(seq_);
      frame->frame_to_map_ = frame_to_map_;
      frame->stamp_ = curr_stamp;
      frame->tree_ = std::move(current_tree);
      frame->tree_->deviceId();
      keyframes_.push_back(std::move(frame));
      trajectory_.push_back(Pose::identity());
      is_initialized_ = true;
      is_map_updated_ = true;
      seq_++;
    } else {
      initialize(curr_stamp, *cloud);
    }
    current_tree_view_ = keyframes_.back()->tree_.get();
    current_num_leaves_ = size_t(current_tree_view_->numLeaves());
    return;
  }

  if (!current_tree) {
    if (deskew_ && trajectory_.size() > 1) {
      // the azimuth order computed ahead for exactly this scan, if there is one (older ones: scans that never came)
      DeskewOrder ahead;
      bool have = false;
      for (size_t q = 0; q < deskew_ahead_.size() && !have; ++q) {
        if (!deskew_ahead_[q].key.matches(*cloud)) continue;
        for (size_t d = 0; d < q; ++d) {
          if (deskew_ahead_.front().order.valid()) deskew_ahead_.front().order.wait();
          deskew_ahead_.pop_front();
        }
        ahead = deskew_ahead_.front().order.get();
        deskew_ahead_.pop_front();
        have = true;
        ++look_ahead_hits_;
      }
      deskew(*cloud, trajectory_[trajectory_.size() - 2], trajectory_[trajectory_.size() - 1], have ? &ahead : nullptr);
    }
    current_tree = std::make_unique<MADtree>(std::move(*cloud), b_max_, b_min_, max_parallel_levels_);
  }
  // resident from now on (frame window, maybe keyframe later); the copy runs on the library's copy stream while the
  // host goes on, and the moving leaves are read from it on the device (pipeline.cpp:143-144,154)
  current_tree->deviceId();
  current_num_leaves_ = size_t(current_tree->numLeaves());
  last_build_ms_ = now_ms() - t_pre;

  // constant-velocity prediction (pipeline.cpp:146-152)
  double dx[6];
  for (int i = 0; i < 6; ++i) dx[i] = current_velocity_[i] * 1. / sensor_hz_;
  const Pose prediction = compose(frame_to_map_, motion_from_twist(dx));

  icp_.setMoving(*current_tree);
  icp_.init(prediction);

  const float preprocessing_time = virtual_pre_ms_ >= 0 ? float(virtual_pre_ms_) : float(now_ms() - t_pre);
  if (virtual_pre_ms_ >= 0) round_ms_estimate_ = virtual_round_ms_;
  const double t_icp = now_ms();
  // The reference re-checks its wall-clock budget before every round: round k runs iff preprocessing + the k rounds run
  // so far + ONE MORE round's time (its `icp_time` term: the duration of the round before, counted a second time) still
  // fit loop_time - 5 ms (pipeline.cpp:167-169).  The device loop is one submission, so the same test is turned into a
  // round count BEFORE it starts, at the per-round time the previous frame measured: with budget B and round time t,
  // rounds = 0 if B < 0, else max(1, floor(B / t)) (round 0 is tested with icp_time = 0), capped at MAX_ICP_ITS — all of
  // them unless the budget is nearly spent: a round is ~0.02 ms here.
  int rounds = MAX_ICP_ITS;
  if (realtime_) {
    const double remaining = double(loop_time_) - 5.0 - double(preprocessing_time);
    rounds = remaining < 0 ? 0
                           : int(std::min<double>(MAX_ICP_ITS, std::max(1.0, std::floor(remaining / std::max(round_ms_estimate_, 1e-3)))));
  }
  int matched_leaves = 0;
  if (rounds > 0) {
    std::vector<MADtree*> fixed;
    fixed.reserve(keyframes_.size());
    for (auto& f : keyframes_) fixed.push_back(f->tree_.get());
    // (cut short: the matched flags are the OR over the rounds that ran — the reference resets them in iteration
    // MAX_ICP_ITS - 1 only — and the launch goes kernel by kernel, no graph is instantiated for an odd round count)
    const bool looking_ahead = device_frontend_ && (dev_pending_ || dev_next_staged_);
    icp_.compute(fixed, rounds, rounds < MAX_ICP_ITS, [this]() {
      if (device_frontend_) beginStagedLookAhead();
    }, looking_ahead);
    matched_leaves = icp_.numMatched();
  }
  last_icp_ms_ = now_ms() - t_icp;
  last_rounds_ = rounds;
  if (rounds > 0) round_ms_estimate_ = last_icp_ms_ / rounds;

  frame_to_map_ = icp_.X_;
  const double inliers_ratio = double(matched_leaves) / double(current_num_leaves_);
  last_inliers_ratio_ = inliers_ratio;
  trajectory_.push_back(frame_to_map_);

  std::vector<Pose> odom_window;
  for (int i = std::max(0, int(trajectory_.size()) - SMOOTHING_T); i < int(trajectory_.size()); ++i)
    odom_window.push_back(trajectory_[i]);
  vel_estimator_.init(current_velocity_);
  vel_estimator_.setOdometry(odom_window);
  vel_estimator_.oneRound();
  std::memcpy(current_velocity_, vel_estimator_.X_, sizeof(current_velocity_));

  auto current_frame = std::make_unique<Frame>();
  current_frame->frame_ = int(seq_);
  current_frame->frame_to_map_ = frame_to_map_;
  current_frame->stamp_ = curr_stamp;
  current_frame->weight_ = det_of_inverse6(icp_.H_adder_);  // pipeline.cpp:223
  // device copy: transformed now, stream-ordered behind the registration; host copy: when somebody reads it
  current_tree->applyTransform(frame_to_map_.R, frame_to_map_.t);
  current_frame->tree_ = std::move(current_tree);
  current_tree_view_ = current_frame->tree_.get();

  frames_.push_back(std::move(current_frame));
  if (frames_.size() > size_t(FRAME_WINDOW)) frames_.pop_front();  // its HBM buffers go back to the library's pool

  if (inliers_ratio < p_th_) {  // keyframe promotion (pipeline.cpp:234-262)
    double best_weight = std::numeric_limits<double>::max();
    int new_seq = 0;
    size_t best = frames_.size();
    for (size_t i = 0; i < frames_.size(); ++i) {
      if (frames_[i]->weight_ < best_weight) {
        best_weight = frames_[i]->weight_;
        new_seq = frames_[i]->frame_;
        best = i;
      }
    }
    if (best < frames_.size()) {  // (all-NaN weights would dereference null in the reference; skip instead)
      std::unique_ptr<Frame> best_frame;
      while (!frames_.empty() && frames_.front()->frame_ <= new_seq) {
        if (frames_.front()->frame_ == new_seq) best_frame = std::move(frames_.front());
        frames_.pop_front();
      }
      // promotion is a pointer move: the tree has been resident, in the map frame, since its own frame
      keyframe_to_map_ = best_frame->frame_to_map_;
      keyframes_.push_back(std::move(best_frame));
      if (keyframes_.size() > size_t(num_keyframes_)) keyframes_.pop_front();  // eviction: buffers back to the pool
      is_map_updated_ = true;
      seq_keyframe_ = new_seq;
    }
  }
  seq_++;
}

// the reference's current_leaves_ are pointers into the current tree, so after compute() they read map-frame means
// (pipeline.cpp:290-297); here they are materialised when asked for (the visualiser), not every frame
// pipeline.cpp:290-297: `current_leaves_` is filled by the first REGISTERED frame (pipeline.cpp:143-144); initialize() (:267-283)
// leaves it empty, so the reference returns nothing after the first scan — mirrored
const ContainerType Pipeline::currentLeaves() {
  return (current_tree_view_ && trajectory_.size() > 1) ? current_tree_view_->leafMeans() : ContainerType{};
}

const ContainerType Pipeline::modelLeaves() {  // pipeline.cpp:299-308
  ContainerType leaves;
  for (auto& frame : keyframes_) {
    const ContainerType l = frame->tree_->leafMeans();
    leaves.insert(leaves.end(), l.begin(), l.end());
  }
  return leaves;
}

}  // namespace madicp_host
