// ORACLE — TEST INFRASTRUCTURE ONLY.  extern "C" surface over the CPU restatement so that tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg can drive it through ctypes.  The shipped
// product (mad_icp_amd/) never loads this library.
//
// Conventions: 3x3 rotations and 6x6 matrices cross this ABI ROW-major (numpy's natural order);
// a pose is 12 doubles = R row-major (9) followed by t (3).  Clouds are (N,3) float64 C-contiguous.
#include "mad_oracle.h"

#include <chrono>
#include <cstring>
#include <omp.h>
#include <unordered_map>

using namespace oracle;

namespace {

struct TreeHandle {
  MADtree* root = nullptr;
  LeafList leaves;                                       // DFS, left first (mad_tree.cpp:154-163)
  std::unordered_map<const MADtree*, uint32_t> ordinal;  // leaf -> DFS ordinal
  bool owns = true;  // (false: a keyframe tree borrowed from a Pipeline, orc_pipeline_keyframe_borrow)
  ~TreeHandle() { if (owns) delete root; }
};

Iso3 pose_from(const double* x) {
  Iso3 X;
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) X.R(r, c) = x[r * 3 + c];
  X.t = {{x[9], x[10], x[11]}};
  return X;
}
void pose_to(const Iso3& X, double* x) {
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) x[r * 3 + c] = X.R(r, c);
  for (int i = 0; i < 3; ++i) x[9 + i] = X.t[i];
}
ContainerType cloud_from(const double* pts, int64_t n) {
  ContainerType c(static_cast<size_t>(n));
  if (n) std::memcpy(c.data(), pts, size_t(n) * sizeof(Vec3));
  return c;
}
void index_leaves(TreeHandle* h) {
  h->leaves.clear();
  h->ordinal.clear();
  h->root->getLeafs(h->leaves);
  for (size_t i = 0; i < h->leaves.size(); ++i) h->ordinal[h->leaves[i]] = uint32_t(i);
}
int64_t count_nodes(const MADtree* n) { return n ? 1 + count_nodes(n->left_) + count_nodes(n->right_) : 0; }

struct Exporter {
  double* mean;
  double* evecs;
  double* bbox;
  int32_t* left;
  int32_t* right;
  int32_t* num_points;
  int32_t next = 0;
  int32_t walk(const MADtree* n) {  // preorder numbering
    const int32_t id = next++;
    for (int i = 0; i < 3; ++i) mean[id * 3 + i] = n->mean_[i];
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) evecs[id * 9 + r * 3 + c] = n->eigenvectors_(r, c);
    for (int i = 0; i < 3; ++i) bbox[id * 3 + i] = n->bbox_[i];
    num_points[id] = n->num_points_;
    left[id] = n->left_ ? walk(n->left_) : -1;
    right[id] = n->right_ ? walk(n->right_) : -1;
    return id;
  }
};

}  // namespace

extern "C" {

// ---- restated Eigen routines, exposed for the numpy cross-checks --------------------------------
void orc_eig3(const double* A_rowmajor, double* evals, double* evecs_rowmajor) {
  Mat3 A, V;
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) A(r, c) = A_rowmajor[r * 3 + c];
  eig3_compute_direct(A, evals, V);
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) evecs_rowmajor[r * 3 + c] = V(r, c);
}
void orc_ldlt6_solve(const double* A_rowmajor, const double* b, double* x) {
  Mat6 A;
  Vec6 rhs;
  for (int r = 0; r < 6; ++r) {
    rhs[r] = b[r];
    for (int c = 0; c < 6; ++c) A(r, c) = A_rowmajor[r * 6 + c];
  }
  const Vec6 s = ldlt6_solve(A, rhs);
  for (int r = 0; r < 6; ++r) x[r] = s[r];
}
double orc_det_inverse6(const double* A_rowmajor) {
  Mat6 A;
  for (int r = 0; r < 6; ++r)
    for (int c = 0; c < 6; ++c) A(r, c) = A_rowmajor[r * 6 + c];
  return det6(inverse6(A));
}
void orc_expmap_so3(const double* w, double* R_rowmajor) {
  const Mat3 R = expMapSO3({{w[0], w[1], w[2]}});
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) R_rowmajor[r * 3 + c] = R(r, c);
}
void orc_logmap_so3(const double* R_rowmajor, double* w) {
  Mat3 R;
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) R(r, c) = R_rowmajor[r * 3 + c];
  const Vec3 o = logMapSO3(R);
  for (int i = 0; i < 3; ++i) w[i] = o[i];
}

// ---- MADtree ----------------------------------------------------------------------------------------
void* orc_tree_build(const double* pts, int64_t n, double b_max, double b_min, int max_parallel_level) {
  if (n <= 0) return nullptr;
  ContainerType cloud = cloud_from(pts, n);
  TreeHandle* h = new TreeHandle;
  h->root = new MADtree(&cloud, cloud.begin(), cloud.end(), b_max, b_min, 0, max_parallel_level, nullptr, nullptr);
  index_leaves(h);
  return h;
}
void orc_tree_free(void* h) { delete static_cast<TreeHandle*>(h); }
int64_t orc_tree_num_nodes(void* h) { return count_nodes(static_cast<TreeHandle*>(h)->root); }
int64_t orc_tree_num_leaves(void* h) { return int64_t(static_cast<TreeHandle*>(h)->leaves.size()); }

// preorder dump; evecs row-major 3x3 per node; child ids in the same numbering, -1 = none
void orc_tree_export(void* h, double* mean, double* evecs, double* bbox, int32_t* left, int32_t* right,
                     int32_t* num_points) {
  Exporter ex{mean, evecs, bbox, left, right, num_points};
  ex.walk(static_cast<TreeHandle*>(h)->root);
}
void orc_tree_leaves(void* h, double* mean, double* normal, double* bbox0) {
  TreeHandle* t = static_cast<TreeHandle*>(h);
  for (size_t i = 0; i < t->leaves.size(); ++i) {
    const MADtree* l = t->leaves[i];
    for (int k = 0; k < 3; ++k) {
      mean[i * 3 + k] = l->mean_[k];
      normal[i * 3 + k] = l->eigenvectors_(k, 0);
    }
    bbox0[i] = l->bbox_[0];
  }
}
void orc_tree_transform(void* h, const double* R_rowmajor, const double* t) {
  Mat3 R;
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) R(r, c) = R_rowmajor[r * 3 + c];
  static_cast<TreeHandle*>(h)->root->applyTransform(R, {{t[0], t[1], t[2]}});
}
// searchCloud (mad_tree_wrapper.h:48-67): leaf ordinal, optional visited depth and distance per query
void orc_tree_search(void* h, const double* q, int64_t n, uint32_t* out_leaf, int32_t* out_depth, double* out_dist) {
  TreeHandle* t = static_cast<TreeHandle*>(h);
  for (int64_t i = 0; i < n; ++i) {
    const Vec3 query = {{q[i * 3], q[i * 3 + 1], q[i * 3 + 2]}};
    int depth;
    const MADtree* leaf = t->root->bestMatchingLeafFastDepth(query, depth);
    out_leaf[i] = t->ordinal.at(leaf);
    if (out_depth) out_depth[i] = depth;
    if (out_dist) out_dist[i] = norm(query - leaf->mean_);
  }
}

// ---- MADicp -----------------------------------------------------------------------------------------
// One call of MADicp::update (mad_icp.cpp:74-103) for one fixed tree on one thread, with the
// correspondence trace.  out_corr[i] = DFS ordinal of the NN leaf; out_rejected[i] = gate result.
// H row-major 36, b 6.  Returns the number of internal nodes visited (sum of descent depths).
int64_t orc_icp_linearize(void* moving_h, void* fixed_h, const double* X12, double min_ball, double rho_ker,
                          double b_ratio, double* out_H, double* out_b, uint32_t* out_corr, uint8_t* out_rejected,
                          uint8_t* out_matched) {
  TreeHandle* mv = static_cast<TreeHandle*>(moving_h);
  TreeHandle* fx = static_cast<TreeHandle*>(fixed_h);
  const int saved = omp_get_max_threads();
  (void)saved;
  MADicp icp(min_ball, rho_ker, b_ratio, 1);
  icp.setMoving(mv->leaves);
  icp.init(pose_from(X12));
  MADicp::Trace trace;
  trace.nn.resize(mv->leaves.size());
  trace.rejected.resize(mv->leaves.size());
  icp.trace_ = &trace;
  for (MADtree* l : mv->leaves) l->matched_ = false;
  icp.resetAdders();
  icp.update(fx->root);
  for (int r = 0; r < 6; ++r) {
    out_b[r] = icp.b_adders_[0][r];
    for (int c = 0; c < 6; ++c) out_H[r * 6 + c] = icp.H_adders_[0](r, c);
  }
  for (size_t i = 0; i < mv->leaves.size(); ++i) {
    if (out_corr) out_corr[i] = fx->ordinal.at(trace.nn[i]);
    if (out_rejected) out_rejected[i] = trace.rejected[i];
    if (out_matched) out_matched[i] = mv->leaves[i]->matched_ ? 1 : 0;
  }
  return icp.depth_adders_[0];
}

// The GN driver loop of Pipeline::compute (pipeline.cpp:154-155,166-193) / MADicpWrapper::compute
// (mad_icp_wrapper.h:72-81): n_iters rounds of {resetAdders; omp parallel for over the K fixed trees
// -> update; updateState}; matched flags cleared before the last round.  X12 is updated in place.
// out_X_iters (optional) receives the pose BEFORE each round (n_iters x 12), so a per-iteration parity
// test can inject exactly the pose the oracle linearised at.  Returns wall-clock ms of the loop.
double orc_icp_register(void* moving_h, void** fixed_hs, int K, double* X12, int n_iters, double min_ball,
                        double rho_ker, double b_ratio, int num_threads, double* out_H, double* out_b,
                        uint8_t* out_matched, double* out_X_iters, int64_t* out_depth_sum) {
  TreeHandle* mv = static_cast<TreeHandle*>(moving_h);
  omp_set_num_threads(num_threads);
  MADicp icp(min_ball, rho_ker, b_ratio, num_threads);
  icp.setMoving(mv->leaves);
  icp.init(pose_from(X12));
  const auto t0 = std::chrono::steady_clock::now();
  for (int it = 0; it < n_iters; ++it) {
    if (out_X_iters) pose_to(icp.X_, out_X_iters + size_t(it) * 12);
    if (it == n_iters - 1)
      for (MADtree* l : mv->leaves) l->matched_ = false;
    icp.resetAdders();
#pragma omp parallel for
    for (int k = 0; k < K; ++k) icp.update(static_cast<TreeHandle*>(fixed_hs[k])->root);
    icp.updateState();
  }
  const auto t1 = std::chrono::steady_clock::now();
  pose_to(icp.X_, X12);
  if (out_H)
    for (int r = 0; r < 6; ++r)
      for (int c = 0; c < 6; ++c) out_H[r * 6 + c] = icp.H_adder_(r, c);
  if (out_b)
    for (int r = 0; r < 6; ++r) out_b[r] = icp.b_adder_[r];
  if (out_matched)
    for (size_t i = 0; i < mv->leaves.size(); ++i) out_matched[i] = mv->leaves[i]->matched_ ? 1 : 0;
  if (out_depth_sum) {
    long long s = 0;
    for (long long d : icp.depth_adders_) s += d;
    *out_depth_sum = s;
  }
  return std::chrono::duration<double, std::milli>(t1 - t0).count();
}

// ---- Pipeline ---------------------------------------------------------------------------------------
void* orc_pipeline_create(double sensor_hz, int deskew, double b_max, double rho_ker, double p_th, double b_min,
                          double b_ratio, int num_keyframes, int num_threads, int realtime) {
  return new Pipeline(sensor_hz, deskew != 0, b_max, rho_ker, p_th, b_min, b_ratio, num_keyframes, num_threads,
                      realtime != 0);
}
void orc_pipeline_free(void* p) { delete static_cast<Pipeline*>(p); }
void orc_pipeline_compute(void* p, double stamp, const double* pts, int64_t n) {
  static_cast<Pipeline*>(p)->compute(stamp, cloud_from(pts, n));
}
void orc_pipeline_current_pose(void* p, double* X12) { pose_to(static_cast<Pipeline*>(p)->currentPose(), X12); }
void orc_pipeline_keyframe_pose(void* p, double* X12) { pose_to(static_cast<Pipeline*>(p)->keyframePose(), X12); }
int64_t orc_pipeline_current_id(void* p) { return int64_t(static_cast<Pipeline*>(p)->currentID()); }
int64_t orc_pipeline_keyframe_id(void* p) { return int64_t(static_cast<Pipeline*>(p)->keyframeID()); }
int orc_pipeline_is_map_updated(void* p) { return static_cast<Pipeline*>(p)->isMapUpdated() ? 1 : 0; }
int64_t orc_pipeline_num_keyframes(void* p) { return int64_t(static_cast<Pipeline*>(p)->numKeyframes()); }
void orc_pipeline_set_virtual_times(void* p, double pre_ms, double round_ms) {
  static_cast<Pipeline*>(p)->virtual_pre_ms_ = pre_ms;
  static_cast<Pipeline*>(p)->virtual_round_ms_ = round_ms;
}
int orc_pipeline_last_rounds(void* p) { return static_cast<Pipeline*>(p)->last_rounds_; }
double orc_pipeline_last_icp_ms(void* p) { return static_cast<Pipeline*>(p)->last_icp_ms_; }
double orc_pipeline_last_inliers_ratio(void* p) { return static_cast<Pipeline*>(p)->last_inliers_ratio_; }
int64_t orc_pipeline_current_leaves(void* p, double* out, int64_t cap) {
  const ContainerType l = static_cast<Pipeline*>(p)->currentLeaves();
  if (out && int64_t(l.size()) <= cap && !l.empty()) std::memcpy(out, l.data(), l.size() * sizeof(Vec3));
  return int64_t(l.size());
}
int64_t orc_pipeline_model_leaves(void* p, double* out, int64_t cap) {
  const ContainerType l = static_cast<Pipeline*>(p)->modelLeaves();
  if (out && int64_t(l.size()) <= cap && !l.empty()) std::memcpy(out, l.data(), l.size() * sizeof(Vec3));
  return int64_t(l.size());
}

// The state a ONE-STEP parity test starts a frame from (tests/test_gpu_deskew_one_step.py): the prediction the last frame's
// loop started at, and keyframe tree k as it stands in the local map (map frame), in orc_tree_export's form.
void orc_pipeline_last_guess(void* p, double* X12) { pose_to(static_cast<Pipeline*>(p)->last_guess_, X12); }
void orc_pipeline_predict(void* p, double* X12) { pose_to(static_cast<Pipeline*>(p)->predict(), X12); }
// keyframe tree k as a tree handle that does NOT own it: valid until the pipeline's next compute() (which may evict it)
void* orc_pipeline_keyframe_borrow(void* p, int64_t k) {
  const MADtree* t = static_cast<Pipeline*>(p)->keyframeTree(size_t(k));
  if (!t) return nullptr;
  TreeHandle* h = new TreeHandle;
  h->root = const_cast<MADtree*>(t);
  h->owns = false;
  index_leaves(h);
  return h;
}
int64_t orc_pipeline_keyframe_num_nodes(void* p, int64_t k) {
  const MADtree* t = static_cast<Pipeline*>(p)->keyframeTree(size_t(k));
  return t ? count_nodes(t) : 0;
}
void orc_pipeline_keyframe_export(void* p, int64_t k, double* mean, double* evecs, double* bbox, int32_t* left, int32_t* right,
                                  int32_t* num_points) {
  const MADtree* t = static_cast<Pipeline*>(p)->keyframeTree(size_t(k));
  if (!t) return;
  Exporter ex{mean, evecs, bbox, left, right, num_points};
  ex.walk(t);
}

// Pipeline::deskew on its own (pipeline.cpp:79-123) — the checker of the device front-end's madicp_cloud_deskew.
// pts: (n,3) in / out; out_vel6 (optional): naive_vel of :82-86, what the device entry point is given.
void orc_deskew(double* pts, int64_t n, const double* Tprev12, const double* Tnow12, double sensor_hz, double* out_vel6) {
  struct Open : Pipeline {
    using Pipeline::Pipeline;
    using Pipeline::deskew;
  };
  Open p(sensor_hz, true, 0.2, 0.1, 0.8, 0.1, 0.02, 4, 1, false);
  ContainerType c = cloud_from(pts, n);
  const Iso3 Tp = pose_from(Tprev12), Tn = pose_from(Tnow12);
  p.deskew(&c, Tp, Tn);
  if (n) std::memcpy(pts, c.data(), size_t(n) * sizeof(Vec3));
  if (out_vel6) {
    const double ts = 1. / sensor_hz;
    const Iso3 rel = compose(inverse(Tp), Tn);
    const Vec3 w = logMapSO3(rel.R);
    for (int i = 0; i < 3; ++i) {
      out_vel6[i] = rel.t[i] / ts;
      out_vel6[3 + i] = w[i] / ts;
    }
  }
}

int orc_num_procs() { return omp_get_num_procs(); }

}  // extern "C"
