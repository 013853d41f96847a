// ORACLE — TEST INFRASTRUCTURE ONLY (see mad_oracle.h / linalg.h headers).  PARITY UNPINNED.
#include "mad_oracle.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <future>
#include <limits>
#include <omp.h>
#include <sys/time.h>
#include <utility>

namespace oracle {

// ----------------------------------------------------------------------------------------------------
// tools/utils.h
// ----------------------------------------------------------------------------------------------------

// utils.h:37-52 — in-place partition; the upper cursor is a reverse iterator, so the element it
// designates is the one just before `hi`.  The swap order decides the point order inside each child
// and therefore the summation order of the child's mean/covariance.
template <typename Pred>
static IteratorType split(IteratorType begin, IteratorType end, const Pred& predicate) {
  IteratorType lo = begin;
  IteratorType hi = end;
  while (lo != hi) {
    if (predicate(*lo)) {
      ++lo;
    } else {
      std::swap(*lo, *(hi - 1));
      --hi;
    }
  }
  return hi;
}

// utils.h:54-73
static int computeMeanAndCovariance(Vec3& mean, Mat3& cov, IteratorType begin, IteratorType end) {
  mean = {{0, 0, 0}};
  for (int j = 0; j < 3; ++j)
    for (int i = 0; i < 3; ++i) cov(i, j) = 0.0;
  int k = 0;
  for (IteratorType it = begin; it != end; ++it) {
    const Vec3& v = *it;
    for (int i = 0; i < 3; ++i) mean[i] += v[i];
    for (int j = 0; j < 3; ++j)
      for (int i = 0; i < 3; ++i) cov(i, j) += v[i] * v[j];
    ++k;
  }
  const double inv_k = 1. / k;
  for (int i = 0; i < 3; ++i) mean[i] *= inv_k;
  for (int j = 0; j < 3; ++j)
    for (int i = 0; i < 3; ++i) cov(i, j) *= inv_k;
  for (int j = 0; j < 3; ++j)
    for (int i = 0; i < 3; ++i) cov(i, j) -= mean[i] * mean[j];
  const double bessel = double(k) / double(k - 1);
  for (int j = 0; j < 3; ++j)
    for (int i = 0; i < 3; ++i) cov(i, j) *= bessel;
  return k;
}

// utils.h:75-97 — R is eigenvectors_.transpose(); std::min/max keep the first argument on NaN
static int computeBoundingBox(Vec3& b_max, const Vec3& center, const Mat3& eigvecs, IteratorType begin, IteratorType end) {
  int k = 0;
  Vec3 bbox_neg = {{0, 0, 0}};
  Vec3 bbox_pos = {{0, 0, 0}};
  for (IteratorType it = begin; it != end; ++it) {
    const Vec3 v = mulT(eigvecs, *it - center);
    for (int i = 0; i < 3; ++i) {
      bbox_neg[i] = std::min<double>(bbox_neg[i], v[i]);
      bbox_pos[i] = std::max<double>(bbox_pos[i], v[i]);
    }
    ++k;
  }
  b_max = bbox_pos - bbox_neg;
  return k;
}

// ----------------------------------------------------------------------------------------------------
// tools/mad_tree.cpp
// ----------------------------------------------------------------------------------------------------

MADtree::MADtree(ContainerType* vec, IteratorType begin, IteratorType end, double b_max, double b_min, int level,
                 int max_parallel_level, MADtree* parent, MADtree* plane_predecessor) {
  build(vec, begin, end, b_max, b_min, level, max_parallel_level, parent, plane_predecessor);
}

static MADtree* makeSubtree(ContainerType* vec, IteratorType begin, IteratorType end, double b_max, double b_min,
                            int level, int max_parallel_level, MADtree* parent, MADtree* plane_predecessor) {
  return new MADtree(vec, begin, end, b_max, b_min, level, max_parallel_level, parent, plane_predecessor);
}

// mad_tree.cpp:47-130
void MADtree::build(ContainerType* vec, IteratorType begin, IteratorType end, double b_max, double b_min, int level,
                    int max_parallel_level, MADtree* parent, MADtree* plane_predecessor) {
  parent_ = parent;
  Mat3 cov;
  computeMeanAndCovariance(mean_, cov, begin, end);
  double evals[3];
  eig3_compute_direct(cov, evals, eigenvectors_);
  num_points_ = computeBoundingBox(bbox_, mean_, eigenvectors_, begin, end);

  if (bbox_[2] < b_max) {  // leaf (:64)
    if (plane_predecessor) {
      eigenvectors_.setCol(0, plane_predecessor->eigenvectors_.col(0));
    } else if (num_points_ < 3) {
      MADtree* node = this;
      while (node->parent_ && node->num_points_ < 3) node = node->parent_;
      eigenvectors_.setCol(0, node->eigenvectors_.col(0));
    }
    // :76-86 — `nearest_point` is a reference to *begin, so the winner is written through it
    Vec3& nearest_point = *begin;
    double shortest_dist = std::numeric_limits<double>::max();
    for (IteratorType it = begin; it != end; ++it) {
      const Vec3 v = *it;
      const double dist = norm(v - mean_);
      if (dist < shortest_dist) {
        nearest_point = v;
        shortest_dist = dist;
      }
    }
    mean_ = nearest_point;
    return;
  }
  if (!plane_predecessor) {
    if (bbox_[0] < b_min) plane_predecessor = this;
  }

  const Vec3 split_plane_normal = eigenvectors_.col(2);
  const Vec3 mean = mean_;
  IteratorType middle =
    split(begin, end, [&](const Vec3& p) -> bool { return dotc(p - mean, split_plane_normal) < double(0); });

  if (level >= max_parallel_level) {
    left_ = new MADtree(vec, begin, middle, b_max, b_min, level + 1, max_parallel_level, this, plane_predecessor);
    right_ = new MADtree(vec, middle, end, b_max, b_min, level + 1, max_parallel_level, this, plane_predecessor);
  } else {
    std::future<MADtree*> l = std::async(makeSubtree, vec, begin, middle, b_max, b_min, level + 1, max_parallel_level,
                                         this, plane_predecessor);
    std::future<MADtree*> r = std::async(makeSubtree, vec, middle, end, b_max, b_min, level + 1, max_parallel_level,
                                         this, plane_predecessor);
    left_ = l.get();
    right_ = r.get();
  }
}

// mad_tree.cpp:144-152
const MADtree* MADtree::bestMatchingLeafFast(const Vec3& query) const {
  const MADtree* node = this;
  while (node->left_ || node->right_) {
    const Vec3 split_plane_normal = node->eigenvectors_.col(2);
    node = (dotc(query - node->mean_, split_plane_normal) < double(0)) ? node->left_ : node->right_;
  }
  return node;
}

const MADtree* MADtree::bestMatchingLeafFastDepth(const Vec3& query, int& depth) const {
  const MADtree* node = this;
  depth = 0;
  while (node->left_ || node->right_) {
    const Vec3 split_plane_normal = node->eigenvectors_.col(2);
    node = (dotc(query - node->mean_, split_plane_normal) < double(0)) ? node->left_ : node->right_;
    ++depth;
  }
  return node;
}

// mad_tree.cpp:154-163
void MADtree::getLeafs(LeafList& out) {
  if (!left_ && !right_) {
    out.push_back(this);
    return;
  }
  if (left_) left_->getLeafs(out);
  if (right_) right_->getLeafs(out);
}

// mad_tree.cpp:165-172
void MADtree::applyTransform(const Mat3& r, const Vec3& t) {
  mean_ = mul(r, mean_) + t;
  eigenvectors_ = mul(r, eigenvectors_);
  if (left_) left_->applyTransform(r, t);
  if (right_) right_->applyTransform(r, t);
}

// ----------------------------------------------------------------------------------------------------
// odometry/mad_icp.cpp
// ----------------------------------------------------------------------------------------------------

MADicp::MADicp(double min_ball, double rho_ker, double b_ratio, int num_threads)
  : rho_ker_(std::sqrt(rho_ker)), min_ball_(min_ball), b_ratio_(b_ratio), num_threads_(num_threads) {
  X_ = Iso3::Identity();
  H_adder_.setZero();
  b_adder_.setZero();
  H_adders_ = std::vector<Mat6>(num_threads);
  b_adders_ = std::vector<Vec6>(num_threads);
  depth_adders_ = std::vector<long long>(num_threads, 0);
}

void MADicp::resetAdders() {  // mad_icp.cpp:43-51
  H_adder_.setZero();
  b_adder_.setZero();
  for (int i = 0; i < num_threads_; ++i) {
    H_adders_[i].setZero();
    b_adders_[i].setZero();
  }
}

void MADicp::setMoving(const LeafList& moving_leaves) { moving_leaves_ = moving_leaves; }

void MADicp::init(const Iso3& moving_in_fixed) { X_ = moving_in_fixed; }

// mad_icp.cpp:59-72
void MADicp::errorAndJacobian(double& e, double J[6], const MADtree& fixed, const MADtree& moving,
                              const Vec3& moving_transformed) const {
  const Vec3& fixed_point = fixed.mean_;
  const Vec3 fixed_normal = fixed.eigenvectors_.col(0);
  const Vec3& moving_point = moving.mean_;
  const Mat3& R = X_.R;

  e = dotc(moving_transformed - fixed_point, fixed_normal);
  for (int j = 0; j < 3; ++j) J[j] = dotc(fixed_normal, R.col(j));
  const Mat3 S = skew(moving_point);
  const Vec3 neg = {{-J[0], -J[1], -J[2]}};
  for (int j = 0; j < 3; ++j) J[3 + j] = dotc(neg, S.col(j));
}

// mad_icp.cpp:74-103
void MADicp::update(const MADtree* fixed_tree) {
  const int thread_id = omp_get_thread_num();
  Mat6& H = H_adders_[thread_id];
  Vec6& b = b_adders_[thread_id];
  long long depth_sum = 0;

  for (size_t idx = 0; idx < moving_leaves_.size(); ++idx) {
    MADtree* moving = moving_leaves_[idx];
    const Vec3 ml = apply(X_, moving->mean_);
    int depth;
    const MADtree* f = fixed_tree->bestMatchingLeafFastDepth(ml, depth);
    depth_sum += depth;

    const double src_ball = min_ball_ + b_ratio_ * norm(moving->mean_);
    const bool rejected = norm(ml - f->mean_) > src_ball;
    if (trace_) {
      trace_->nn[idx] = f;
      trace_->rejected[idx] = rejected ? 1 : 0;
    }
    if (rejected) continue;

    moving->matched_ = true;

    double J[6];
    double e;
    errorAndJacobian(e, J, *f, *moving, ml);

    double scale = 1.;
    const double chi = std::fabs(e);  // `abs(e)` at :93 — fabs intended
    if (chi > rho_ker_) scale = rho_ker_ / chi;
    const double w = 1. - f->bbox_[0] / min_ball_;
    scale *= w * w;

    // scale * J^T * J  ==  (scale * J^T) * J ; scale * J^T * e == (scale * J^T) * e
    double sJ[6];
    for (int i = 0; i < 6; ++i) sJ[i] = scale * J[i];
    for (int j = 0; j < 6; ++j)
      for (int i = 0; i < 6; ++i) H(i, j) += sJ[i] * J[j];
    for (int i = 0; i < 6; ++i) b[i] += sJ[i] * e;
  }
  depth_adders_[thread_id] += depth_sum;
}

// mad_icp.cpp:105-117
void MADicp::updateState() {
  for (int i = 0; i < num_threads_; ++i) {
    for (int c = 0; c < 6; ++c)
      for (int r = 0; r < 6; ++r) H_adder_(r, c) += H_adders_[i](r, c);
    for (int r = 0; r < 6; ++r) b_adder_[r] += b_adders_[i][r];
  }
  Vec6 neg_b;
  for (int r = 0; r < 6; ++r) neg_b[r] = -b_adder_[r];
  const Vec6 dx = ldlt6_solve(H_adder_, neg_b);
  Iso3 dX = Iso3::Identity();
  dX.R = expMapSO3({{dx[3], dx[4], dx[5]}});
  dX.t = {{dx[0], dx[1], dx[2]}};
  X_ = compose(X_, dX);
}

// ----------------------------------------------------------------------------------------------------
// odometry/vel_estimator.cpp
// ----------------------------------------------------------------------------------------------------

VelEstimator::VelEstimator(double sensor_hz) {
  X_.setZero();
  H_adder_.setZero();
  b_adder_.setZero();
  ts_ = 1. / sensor_hz;
}

void VelEstimator::init(const Vec6& velocity) { X_ = velocity; }

void VelEstimator::setOdometry(const std::vector<Iso3>& odometry) { odometry_ = odometry; }

// vel_estimator.cpp:45-61 (J = I * delta_t is folded into update())
void VelEstimator::errorAndJacobian(Vec6& e, const Iso3& T_now, const Iso3& T_prev, double delta_t) {
  const Iso3 T_now_to_prev = compose(inverse(T_prev), T_now);
  for (int i = 0; i < 3; ++i) e[i] = delta_t * X_[i] - T_now_to_prev.t[i];
  const Mat3& L = T_now_to_prev.R;
  double angles[3];
  angles[0] = std::atan2(-L(1, 2), L(2, 2));
  angles[1] = std::asin(L(0, 2));
  angles[2] = std::atan2(-L(0, 1), L(0, 0));
  for (int i = 0; i < 3; ++i) e[3 + i] = delta_t * X_[3 + i] - angles[i];
}

// vel_estimator.cpp:63-79
void VelEstimator::update(const Iso3& T_now, const Iso3& T_prev, double delta_t, double weight) {
  Vec6 e;
  errorAndJacobian(e, T_now, T_prev, delta_t);
  double scale = 1.;
  double chi2 = 0.0;  // squaredNorm of a contiguous 6-vector: three Packet2d partial sums, then predux
  {
    const double p0 = e[0] * e[0] + (e[2] * e[2] + e[4] * e[4]);
    const double p1 = e[1] * e[1] + (e[3] * e[3] + e[5] * e[5]);
    chi2 = p0 + p1;
  }
  const double chi = std::sqrt(chi2);
  if (chi > E_THRESHOLD_VEL) scale = E_THRESHOLD_VEL / chi;
  // scale * weight * J^T * J with J = delta_t * I : only the diagonal receives non-zero terms
  const double sw = scale * weight;
  const double swj = sw * delta_t;
  for (int i = 0; i < 6; ++i) {
    H_adder_(i, i) += swj * delta_t;
    b_adder_[i] += swj * e[i];
  }
}

// vel_estimator.cpp:81-97
void VelEstimator::oneRound() {
  H_adder_.setZero();
  b_adder_.setZero();
  const Iso3 T_now = odometry_.back();
  for (size_t i = 0; i < odometry_.size() - 1; ++i) {
    const Iso3 T_prev = odometry_[i];
    const double delta_t = (odometry_.size() - 1 - i) * ts_;
    const double weight = 1.f - double(odometry_.size() - 2 - i) / double(odometry_.size() - 1);
    update(T_now, T_prev, delta_t, weight);
  }
  Vec6 neg_b;
  for (int r = 0; r < 6; ++r) neg_b[r] = -b_adder_[r];
  const Vec6 dx = ldlt6_solve(H_adder_, neg_b);
  for (int r = 0; r < 6; ++r) X_[r] += dx[r];
}

// ----------------------------------------------------------------------------------------------------
// odometry/pipeline.cpp
// ----------------------------------------------------------------------------------------------------

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
  current_tree_ = nullptr;
  frame_to_map_ = Iso3::Identity();
  keyframe_to_map_ = Iso3::Identity();
  current_velocity_.setZero();
  seq_ = 0;
  seq_keyframe_ = 0;
  is_initialized_ = false;
  is_map_updated_ = false;
  loop_time = (1. / sensor_hz_) * 1000;
  max_parallel_levels_ = static_cast<int>(std::log2(num_threads));
  omp_set_num_threads(num_threads);
}

Pipeline::~Pipeline() {  // pipeline.cpp:68-77
  while (!frames_.empty()) {
    delete frames_.front()->tree_;
    delete frames_.front();
    frames_.pop_front();
  }
  while (!keyframes_.empty()) {
    delete keyframes_.front()->tree_;
    delete keyframes_.front();
    keyframes_.pop_front();
  }
}

// pipeline.cpp:79-123
void Pipeline::deskew(ContainerType* curr_cloud, const Iso3& T_prev, const Iso3& T_now) {
  const double ts = 1. / sensor_hz_;
  const Iso3 T_now_to_prev = compose(inverse(T_prev), T_now);
  const Vec3 w = logMapSO3(T_now_to_prev.R);
  Vec6 naive_vel;
  for (int i = 0; i < 3; ++i) {
    naive_vel[i] = T_now_to_prev.t[i] / ts;
    naive_vel[3 + i] = w[i] / ts;
  }
  using AzimuthPair = std::pair<double, Vec3>;
  std::vector<AzimuthPair> sorted(curr_cloud->size());
  for (size_t i = 0; i < sorted.size(); ++i) {
    const Vec3& point = curr_cloud->at(i);
    sorted[i] = std::make_pair(std::atan2(point[1], point[0]), point);
  }
  std::sort(sorted.begin(), sorted.end(),
            [](const AzimuthPair& first, const AzimuthPair& second) -> bool { return first.first < second.first; });

  const double resolution = 2 * M_PI / double(CHUNKS);
  const double delta = ts / double(CHUNKS - 1);
  double t = -ts;
  auto pose_at = [&](double tt) {
    Iso3 m = Iso3::Identity();
    m.R = expMapSO3({{naive_vel[3] * tt, naive_vel[4] * tt, naive_vel[5] * tt}});
    m.t = {{naive_vel[0] * tt, naive_vel[1] * tt, naive_vel[2] * tt}};
    return m;
  };
  Iso3 meas_pose_to_robot = pose_at(t);
  double angle = M_PI - resolution;
  for (int i = int(sorted.size()) - 1; i >= 0; --i) {
    if (sorted[i].first < angle) {
      angle -= resolution;
      t += delta;
      meas_pose_to_robot = pose_at(t);
    }
    (*curr_cloud)[i] = apply(meas_pose_to_robot, sorted[i].second);
  }
}

static double now_ms() {
  struct timeval tv;
  gettimeofday(&tv, nullptr);
  return double(tv.tv_sec) * 1000. + 1e-3 * double(tv.tv_usec);
}

// pipeline.cpp:125-265
Iso3 Pipeline::predict() const {  // (instrumentation: exactly compute()'s lines below, on the state as it stands)
  Vec6 dx;
  for (int i = 0; i < 6; ++i) dx[i] = current_velocity_[i] * 1. / sensor_hz_;
  Iso3 dX = Iso3::Identity();
  dX.R = expMapSO3({{dx[3], dx[4], dx[5]}});
  dX.t = {{dx[0], dx[1], dx[2]}};
  return compose(frame_to_map_, dX);
}

void Pipeline::compute(const double& curr_stamp, ContainerType curr_cloud_mem) {
  ContainerType* curr_cloud = &curr_cloud_mem;
  is_map_updated_ = false;

  if (!is_initialized_) {
    initialize(curr_stamp, curr_cloud);
    return;
  }
  const double preprocessing_start = now_ms();

  if (deskew_ && trajectory_.size() > 1)
    deskew(curr_cloud, trajectory_[trajectory_.size() - 2], trajectory_[trajectory_.size() - 1]);

  current_tree_ =
    new MADtree(curr_cloud, curr_cloud->begin(), curr_cloud->end(), b_max_, b_min_, 0, max_parallel_levels_, nullptr, nullptr);
  current_leaves_.clear();
  current_tree_->getLeafs(current_leaves_);

  Vec6 dx;
  for (int i = 0; i < 6; ++i) dx[i] = current_velocity_[i] * 1. / sensor_hz_;
  Iso3 dX = Iso3::Identity();
  dX.R = expMapSO3({{dx[3], dx[4], dx[5]}});
  dX.t = {{dx[0], dx[1], dx[2]}};
  const Iso3 prediction = compose(frame_to_map_, dX);

  icp_.setMoving(current_leaves_);
  icp_.init(prediction);
  last_guess_ = prediction;

  float icp_time = 0;
  float total_icp_time = 0;
  const float preprocessing_time = virtual_pre_ms_ >= 0 ? float(virtual_pre_ms_) : float(now_ms() - preprocessing_start);
  const double loop_start = now_ms();
  last_rounds_ = 0;

  for (size_t icp_iteration = 0; icp_iteration < size_t(MAX_ICP_ITS); ++icp_iteration) {
    const float remaining_time = loop_time - 5.0 - (preprocessing_time + total_icp_time + icp_time);
    if (realtime_ && remaining_time < 0) break;

    const double icp_start = now_ms();
    if (icp_iteration == size_t(MAX_ICP_ITS - 1)) {
      for (MADtree* l : current_leaves_) l->matched_ = false;
    }
    icp_.resetAdders();

    const int nk = int(keyframes_.size());
#pragma omp parallel for
    for (int k = 0; k < nk; ++k) {
      icp_.update(keyframes_[k]->tree_);
    }
    icp_.updateState();

    icp_time = virtual_pre_ms_ >= 0 ? float(virtual_round_ms_) : float(now_ms() - icp_start);
    total_icp_time += icp_time;
    ++last_rounds_;
  }
  last_icp_ms_ = now_ms() - loop_start;

  frame_to_map_ = icp_.X_;

  int matched_leaves = 0;
  for (MADtree* l : current_leaves_)
    if (l->matched_) matched_leaves++;
  const double inliers_ratio = double(matched_leaves) / double(current_leaves_.size());
  last_inliers_ratio_ = inliers_ratio;

  trajectory_.push_back(frame_to_map_);

  std::vector<Iso3> odom_window;
  for (int i = std::max(0, int(trajectory_.size()) - SMOOTHING_T); i < int(trajectory_.size()); ++i)
    odom_window.push_back(trajectory_[i]);

  vel_estimator_.init(current_velocity_);
  vel_estimator_.setOdometry(odom_window);
  vel_estimator_.oneRound();
  current_velocity_ = vel_estimator_.X_;

  Frame* current_frame = new Frame;
  current_frame->frame_ = int(seq_);
  current_frame->frame_to_map_ = frame_to_map_;
  current_frame->stamp_ = curr_stamp;
  current_frame->weight_ = det6(inverse6(icp_.H_adder_));
  current_tree_->applyTransform(frame_to_map_.R, frame_to_map_.t);
  current_frame->tree_ = current_tree_;
  current_frame->leaves_ = current_leaves_;

  frames_.push_back(current_frame);
  if (frames_.size() > size_t(FRAME_WINDOW)) {
    delete frames_.front()->tree_;
    delete frames_.front();
    frames_.pop_front();
  }

  if (inliers_ratio < p_th_) {
    double best_weight = std::numeric_limits<double>::max();
    int new_seq = 0;
    Frame* best_frame = nullptr;
    for (Frame* frame : frames_) {
      if (frame->weight_ < best_weight) {
        best_weight = frame->weight_;
        new_seq = frame->frame_;
        best_frame = frame;
      }
    }
    // NOTE (reference behaviour): if every weight is NaN/inf-max no frame is selected and the reference
    // dereferences a null best_frame at pipeline.cpp:260; the restatement skips the promotion instead.
    if (best_frame) {
      while (!frames_.empty() && frames_.front()->frame_ <= new_seq) {
        if (frames_.front()->frame_ < new_seq) {
          delete frames_.front()->tree_;
          delete frames_.front();
        }
        frames_.pop_front();
      }
      keyframes_.push_back(best_frame);
      if (keyframes_.size() > size_t(num_keyframes_)) {
        delete keyframes_.front()->tree_;
        delete keyframes_.front();
        keyframes_.pop_front();
      }
      is_map_updated_ = true;
      seq_keyframe_ = new_seq;
      keyframe_to_map_ = best_frame->frame_to_map_;
    }
  }
  seq_++;
}

// pipeline.cpp:267-284
void Pipeline::initialize(const double& curr_stamp, ContainerType* curr_cloud) {
  Frame* current_frame = new Frame;
  current_frame->frame_ = int(seq_);
  current_frame->frame_to_map_ = frame_to_map_;
  current_frame->stamp_ = curr_stamp;
  current_frame->tree_ =
    new MADtree(curr_cloud, curr_cloud->begin(), curr_cloud->end(), b_max_, b_min_, 0, max_parallel_levels_, nullptr, nullptr);
  current_frame->tree_->getLeafs(current_frame->leaves_);
  keyframes_.push_back(current_frame);
  trajectory_.push_back(Iso3::Identity());
  is_initialized_ = true;
  is_map_updated_ = true;
  seq_++;
}

ContainerType Pipeline::currentLeaves() const {  // pipeline.cpp:290-297
  ContainerType leaves;
  for (MADtree* leaf : current_leaves_) leaves.push_back(leaf->mean_);
  return leaves;
}

ContainerType Pipeline::modelLeaves() const {  // pipeline.cpp:299-308
  ContainerType leaves;
  for (auto frame : keyframes_)
    for (MADtree* leaf : frame->leaves_) leaves.push_back(leaf->mean_);
  return leaves;
}

// ----------------------------------------------------------------------------------------------------
// pybind/tools wrappers
// ----------------------------------------------------------------------------------------------------

void MADtreeTool::build(ContainerType vec, double b_max, double b_min, int max_parallel_level) {
  ContainerType* p = &vec;
  mad_tree_.reset(new MADtree(p, p->begin(), p->end(), b_max, b_min, 0, max_parallel_level, nullptr, nullptr));
}

const MADtree* MADtreeTool::search(const Vec3& query) const { return mad_tree_->bestMatchingLeafFast(query); }

MADicpTool::MADicpTool(int num_threads) : num_threads_(num_threads) {
  max_parallel_levels_ = static_cast<int>(std::log2(num_threads_));
  omp_set_num_threads(num_threads_);
}

void MADicpTool::setQueryCloud(ContainerType query, double b_max, double b_min) {
  ContainerType* p = &query;
  query_leaves_.clear();
  query_tree_.reset(new MADtree(p, p->begin(), p->end(), b_max, b_min, 0, max_parallel_levels_, nullptr, nullptr));
  query_tree_->getLeafs(query_leaves_);
}

void MADicpTool::setReferenceCloud(ContainerType reference, double b_max, double b_min) {
  ContainerType* p = &reference;
  ref_b_max_ = b_max;
  ref_tree_.reset(new MADtree(p, p->begin(), p->end(), ref_b_max_, b_min, 0, max_parallel_levels_, nullptr, nullptr));
}

// mad_icp_wrapper.h:54-102
Iso3 MADicpTool::compute(const Iso3& T, size_t max_icp_iterations, double rho_ker, double b_ratio) {
  mad_icp_.reset(new MADicp(ref_b_max_, rho_ker, b_ratio, 1));
  mad_icp_->setMoving(query_leaves_);
  mad_icp_->init(T);
  for (size_t icp_iteration = 0; icp_iteration < max_icp_iterations; ++icp_iteration) {
    if (icp_iteration == max_icp_iterations - 1) {
      for (MADtree* l : query_leaves_) l->matched_ = false;
    }
    mad_icp_->resetAdders();
    mad_icp_->update(ref_tree_.get());
    mad_icp_->updateState();
  }
  return mad_icp_->X_;
}

}  // namespace oracle
