// ORACLE — TEST INFRASTRUCTURE ONLY.  The oracle's extern "C" surface (oracle_capi.cpp, same symbols, same conventions)
// implemented by the REFERENCE'S OWN classes: this file includes the reference's headers and is linked with the
// reference's own translation units, compiled from where they lie under /root/reference against the Eigen stand-in of
// oracle/eigen_standin (see its header for what that does and does not pin).  Built by oracle/build_ref_standin.sh into
// oracle/_ref/libmad_ref_standin.so; driven by tests/test_reference_structure_pin.py through the same ctypes wrapper as
// the oracle (tests/oracle_lib.py with MADICP_ORACLE_SO pointing here), so both sides run the same script and their
// outputs are compared bit for bit.  Never part of the product.
#include <odometry/mad_icp.h>
#include <odometry/pipeline.h>
#include <tools/mad_tree.h>

#include <chrono>
#include <cstdint>
#include <cstring>
#include <unordered_map>

namespace {

struct TreeHandle {
  MADtree* root = nullptr;
  LeafList leaves;  // MADtree::getLeafs order (mad_tree.cpp:154-163)
  std::unordered_map<const MADtree*, uint32_t> ordinal;
  ~TreeHandle() { delete root; }
};

Eigen::Isometry3d pose_from(const double* x) {
  Eigen::Isometry3d X;
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) X.linear()(r, c) = x[r * 3 + c];
  X.translation() = Eigen::Vector3d(x[9], x[10], x[11]);
  return X;
}
void pose_to(const Eigen::Isometry3d& X, double* x) {
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) x[r * 3 + c] = X.linear()(r, c);
  for (int i = 0; i < 3; ++i) x[9 + i] = X.translation()(i);
}
void pose_to(const Eigen::Matrix<double, 4, 4>& M, double* x) {
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) x[r * 3 + c] = M(r, c);
  for (int i = 0; i < 3; ++i) x[9 + i] = M(i, 3);
}
ContainerType cloud_from(const double* pts, int64_t n) {
  ContainerType c(static_cast<size_t>(n));
  static_assert(sizeof(Eigen::Vector3d) == 24, "three doubles");
  if (n) std::memcpy(static_cast<void*>(c.data()), pts, size_t(n) * 24);
  return c;
}
void index_leaves(TreeHandle* h) {
  h->leaves.clear();
  h->ordinal.clear();
  h->root->getLeafs(std::back_insert_iterator<LeafList>(h->leaves));
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
  int32_t walk(const MADtree* n) {
    const int32_t id = next++;
    for (int i = 0; i < 3; ++i) mean[id * 3 + i] = n->mean_(i);
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) evecs[id * 9 + r * 3 + c] = n->eigenvectors_(r, c);
    for (int i = 0; i < 3; ++i) bbox[id * 3 + i] = n->bbox_(i);
    num_points[id] = n->num_points_;
    left[id] = n->left_ ? walk(n->left_) : -1;
    right[id] = n->right_ ? walk(n->right_) : -1;
    return id;
  }
};

struct OpenPipeline : Pipeline {  // the members the oracle's ABI reports are protected in the reference
  using Pipeline::Pipeline;
  using Pipeline::deskew;
  size_t numKeyframes() const { return keyframes_.size(); }
  double inliersRatio() const {  // pipeline.cpp:197-204, recomputed from the leaves the last compute() left behind
    if (current_leaves_.empty()) return 0.0;
    int matched = 0;
    for (MADtree* l : current_leaves_) matched += l->matched_ ? 1 : 0;
    return double(matched) / double(current_leaves_.size());
  }
};

}  // namespace

extern "C" {

const char* orc_flavour() { return "reference sources + Eigen stand-in"; }

void orc_eig3(const double* A_rowmajor, double* evals, double* evecs_rowmajor) {
  Eigen::Matrix3d A;
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) A(r, c) = A_rowmajor[r * 3 + c];
  Eigen::SelfAdjointEigenSolver<Eigen::Matrix3d> es;
  es.computeDirect(A);
  for (int r = 0; r < 3; ++r) {
    evals[r] = es.eigenvalues()(r);
    for (int c = 0; c < 3; ++c) evecs_rowmajor[r * 3 + c] = es.eigenvectors()(r, c);
  }
}
void orc_ldlt6_solve(const double* A_rowmajor, const double* b, double* x) {
  Matrix6d A;
  Vector6d rhs;
  for (int r = 0; r < 6; ++r) {
    rhs(r) = b[r];
    for (int c = 0; c < 6; ++c) A(r, c) = A_rowmajor[r * 6 + c];
  }
  const Vector6d s = A.ldlt().solve(rhs);
  for (int r = 0; r < 6; ++r) x[r] = s(r);
}
double orc_det_inverse6(const double* A_rowmajor) {
  Matrix6d A;
  for (int r = 0; r < 6; ++r)
    for (int c = 0; c < 6; ++c) A(r, c) = A_rowmajor[r * 6 + c];
  return A.inverse().determinant();
}
void orc_expmap_so3(const double* w, double* R_rowmajor) {
  const Eigen::Matrix3d R = expMapSO3(Eigen::Vector3d(w[0], w[1], w[2]));
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) R_rowmajor[r * 3 + c] = R(r, c);
}
void orc_logmap_so3(const double* R_rowmajor, double* w) {
  Eigen::Matrix3d R;
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) R(r, c) = R_rowmajor[r * 3 + c];
  const Eigen::Vector3d o = logMapSO3(R);
  for (int i = 0; i < 3; ++i) w[i] = o(i);
}

// ---- MADtree ----------------------------------------------------------------------------------------
void* orc_tree_build(const double* pts, int64_t n, double b_max, double b_min, int max_parallel_level) {
  if (n <= 0) return nullptr;
  ContainerType cloud = cloud_from(pts, n);
  TreeHandle* h = new TreeHandle;
  h->root = new MADtree(&cloud, cloud.begin(), cloud.end(), b_max, b_min, 0, max_parallel_level, nullptr, nullptr);
  index_leaves(h);
  // the reference never initialises matched_ (mad_tree.h:92); the ABI reports it, so give it the oracle's start value
  for (MADtree* l : h->leaves) l->matched_ = false;
  return h;
}
void orc_tree_free(void* h) { delete static_cast<TreeHandle*>(h); }
int64_t orc_tree_num_nodes(void* h) { return count_nodes(static_cast<TreeHandle*>(h)->root); }
int64_t orc_tree_num_leaves(void* h) { return int64_t(static_cast<TreeHandle*>(h)->leaves.size()); }
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
      mean[i * 3 + k] = l->mean_(k);
      normal[i * 3 + k] = l->eigenvectors_(k, 0);
    }
    bbox0[i] = l->bbox_(0);
  }
}
void orc_tree_transform(void* h, const double* R_rowmajor, const double* t) {
  Eigen::Matrix3d R;
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) R(r, c) = R_rowmajor[r * 3 + c];
  static_cast<TreeHandle*>(h)->root->applyTransform(R, Eigen::Vector3d(t[0], t[1], t[2]));
}
// searchCloud (mad_tree_wrapper.h:48-67).  The reference does not count visited nodes: depth is reported as -1.
void orc_tree_search(void* h, const double* q, int64_t n, uint32_t* out_leaf, int32_t* out_depth, double* out_dist) {
  TreeHandle* t = static_cast<TreeHandle*>(h);
  for (int64_t i = 0; i < n; ++i) {
    const Eigen::Vector3d query(q[i * 3], q[i * 3 + 1], q[i * 3 + 2]);
    const MADtree* leaf = t->root->bestMatchingLeafFast(query);
    out_leaf[i] = t->ordinal.at(leaf);
    if (out_depth) out_depth[i] = -1;
    if (out_dist) out_dist[i] = (query - leaf->mean_).norm();
  }
}

// ---- MADicp -----------------------------------------------------------------------------------------
// One MADicp::update (mad_icp.cpp:74-103) on one thread.  H, b and the matched flags come out of update() itself; the
// reference keeps no correspondence record, so corr / rejected are re-derived beside it with the reference's own
// bestMatchingLeafFast and the expression of mad_icp.cpp:81-82.
int64_t orc_icp_linearize(void* moving_h, void* fixed_h, const double* X12, double min_ball, double rho_ker,
                          double b_ratio, double* out_H, double* out_b, uint32_t* out_corr, uint8_t* out_rejected,
                          uint8_t* out_matched) {
  TreeHandle* mv = static_cast<TreeHandle*>(moving_h);
  TreeHandle* fx = static_cast<TreeHandle*>(fixed_h);
  MADicp icp(min_ball, rho_ker, b_ratio, 1);
  icp.setMoving(mv->leaves);
  icp.init(pose_from(X12));
  for (MADtree* l : mv->leaves) l->matched_ = false;
  icp.resetAdders();
  icp.update(fx->root);
  for (int r = 0; r < 6; ++r) {
    out_b[r] = icp.b_adders_[0](r);
    for (int c = 0; c < 6; ++c) out_H[r * 6 + c] = icp.H_adders_[0](r, c);
  }
  for (size_t i = 0; i < mv->leaves.size(); ++i) {
    const MADtree* moving = mv->leaves[i];
    const Eigen::Vector3d ml = icp.X_ * moving->mean_;
    const MADtree* f = fx->root->bestMatchingLeafFast(ml);
    const double src_ball = icp.min_ball_ + icp.b_ratio_ * moving->mean_.norm();
    if (out_corr) out_corr[i] = fx->ordinal.at(f);
    if (out_rejected) out_rejected[i] = (ml - f->mean_).norm() > src_ball ? 1 : 0;
    if (out_matched) out_matched[i] = moving->matched_ ? 1 : 0;
  }
  return -1;
}

// The driver loop exactly as Pipeline::compute writes it (pipeline.cpp:166-193), around the reference's MADicp
double orc_icp_register(void* moving_h, void** fixed_hs, int K, double* X12, int n_iters, double min_ball,
                        double rho_ker, double b_ratio, int num_threads, double* out_H, double* out_b,
                        uint8_t* out_matched, double* out_X_iters, int64_t* out_depth_sum) {
  TreeHandle* mv = static_cast<TreeHandle*>(moving_h);
  omp_set_num_threads(num_threads);
  MADicp icp(min_ball, rho_ker, b_ratio, num_threads);
  icp.setMoving(mv->leaves);
  icp.init(pose_from(X12));
  std::vector<const MADtree*> fixed(static_cast<size_t>(K));
  for (int k = 0; k < K; ++k) fixed[k] = static_cast<TreeHandle*>(fixed_hs[k])->root;
  const auto t0 = std::chrono::steady_clock::now();
  for (int it = 0; it < n_iters; ++it) {
    if (out_X_iters) pose_to(icp.X_, out_X_iters + size_t(it) * 12);
    if (it == n_iters - 1)
      for (MADtree* l : mv->leaves) l->matched_ = false;
    icp.resetAdders();
#pragma omp parallel for
    for (int k = 0; k < K; ++k) icp.update(fixed[k]);
    icp.updateState();
  }
  const auto t1 = std::chrono::steady_clock::now();
  pose_to(icp.X_, X12);
  if (out_H)
    for (int r = 0; r < 6; ++r)
      for (int c = 0; c < 6; ++c) out_H[r * 6 + c] = icp.H_adder_(r, c);
  if (out_b)
    for (int r = 0; r < 6; ++r) out_b[r] = icp.b_adder_(r);
  if (out_matched)
    for (size_t i = 0; i < mv->leaves.size(); ++i) out_matched[i] = mv->leaves[i]->matched_ ? 1 : 0;
  if (out_depth_sum) *out_depth_sum = -1;
  return std::chrono::duration<double, std::milli>(t1 - t0).count();
}

// ---- Pipeline ---------------------------------------------------------------------------------------
void* orc_pipeline_create(double sensor_hz, int deskew, double b_max, double rho_ker, double p_th, double b_min,
                          double b_ratio, int num_keyframes, int num_threads, int realtime) {
  return new OpenPipeline(sensor_hz, deskew != 0, b_max, rho_ker, p_th, b_min, b_ratio, num_keyframes, num_threads,
                          realtime != 0);
}
void orc_pipeline_free(void* p) { delete static_cast<OpenPipeline*>(p); }
void orc_pipeline_compute(void* p, double stamp, const double* pts, int64_t n) {
  static_cast<OpenPipeline*>(p)->compute(stamp, cloud_from(pts, n));
}
void orc_pipeline_current_pose(void* p, double* X12) { pose_to(static_cast<OpenPipeline*>(p)->currentPose(), X12); }
void orc_pipeline_keyframe_pose(void* p, double* X12) { pose_to(static_cast<OpenPipeline*>(p)->keyframePose(), X12); }
int64_t orc_pipeline_current_id(void* p) { return int64_t(static_cast<OpenPipeline*>(p)->currentID()); }
int64_t orc_pipeline_keyframe_id(void* p) { return int64_t(static_cast<OpenPipeline*>(p)->keyframeID()); }
int orc_pipeline_is_map_updated(void* p) { return static_cast<OpenPipeline*>(p)->isMapUpdated() ? 1 : 0; }
int64_t orc_pipeline_num_keyframes(void* p) { return int64_t(static_cast<OpenPipeline*>(p)->numKeyframes()); }
double orc_pipeline_last_icp_ms(void*) { return 0.0; }
double orc_pipeline_last_inliers_ratio(void* p) { return static_cast<OpenPipeline*>(p)->inliersRatio(); }
int64_t orc_pipeline_current_leaves(void* p, double* out, int64_t cap) {
  const ContainerType l = static_cast<OpenPipeline*>(p)->currentLeaves();
  if (out && int64_t(l.size()) <= cap && !l.empty()) std::memcpy(out, static_cast<const void*>(l.data()), l.size() * 24);
  return int64_t(l.size());
}
int64_t orc_pipeline_model_leaves(void* p, double* out, int64_t cap) {
  const ContainerType l = static_cast<OpenPipeline*>(p)->modelLeaves();
  if (out && int64_t(l.size()) <= cap && !l.empty()) std::memcpy(out, static_cast<const void*>(l.data()), l.size() * 24);
  return int64_t(l.size());
}

// Pipeline::deskew on its own (pipeline.cpp:79-123)
void orc_deskew(double* pts, int64_t n, const double* Tprev12, const double* Tnow12, double sensor_hz, double* out_vel6) {
  OpenPipeline p(sensor_hz, true, 0.2, 0.1, 0.8, 0.1, 0.02, 4, 1, false);
  ContainerType c = cloud_from(pts, n);
  const Eigen::Isometry3d Tp = pose_from(Tprev12), Tn = pose_from(Tnow12);
  p.deskew(&c, Tp, Tn);
  if (n) std::memcpy(pts, static_cast<const void*>(c.data()), size_t(n) * 24);
  if (out_vel6) {  // pipeline.cpp:82-86
    const double ts = 1. / sensor_hz;
    Vector6d naive_vel;
    Eigen::Isometry3d rel = Tp.inverse() * Tn;
    naive_vel.head(3) = rel.translation();
    naive_vel.tail(3) = logMapSO3(rel.linear());
    naive_vel = naive_vel / ts;
    for (int i = 0; i < 6; ++i) out_vel6[i] = naive_vel(i);
  }
}

int orc_num_procs() { return omp_get_num_procs(); }

}  // extern "C"
