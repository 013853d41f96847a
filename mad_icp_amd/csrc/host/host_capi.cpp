// libmadicp_host.so — C ABI over the host tree builder (include/madicp_host.h).
#include <cstring>
#include <vector>

#include "linalg.h"
#include "madicp_host.h"
#include "task_pool.h"
#include "deskew.h"
#include "tree_builder.h"

struct madicp_host_tree {
  madicp_host::LinearTree tree;
};

extern "C" {

madicp_host_tree* madicp_host_tree_build(const double* points, int64_t n, double b_max, double b_min,
                                         int max_parallel_level) {
  if (!points || n <= 0) return nullptr;
  std::vector<double> copy(points, points + 3 * n);
  madicp_host_tree* t = new madicp_host_tree;
  t->tree = madicp_host::build_tree(copy.data(), n, b_max, b_min, max_parallel_level);
  return t;
}
void madicp_host_tree_free(madicp_host_tree* t) { delete t; }
int32_t madicp_host_tree_num_nodes(const madicp_host_tree* t) { return t ? t->tree.num_nodes() : 0; }
int32_t madicp_host_tree_num_leaves(const madicp_host_tree* t) { return t ? t->tree.num_leaves() : 0; }
const madicp_node* madicp_host_tree_nodes(const madicp_host_tree* t) { return t ? t->tree.nodes.data() : nullptr; }
const int32_t* madicp_host_tree_leaf_nodes(const madicp_host_tree* t) { return t ? t->tree.leaf_nodes.data() : nullptr; }
void madicp_host_tree_leaf_means(const madicp_host_tree* t, double* out) {
  if (!t || !out) return;
  for (size_t i = 0; i < t->tree.leaf_nodes.size(); ++i)
    std::memcpy(out + 3 * i, t->tree.nodes[t->tree.leaf_nodes[i]].mean, 3 * sizeof(double));
}
void madicp_host_tree_transform(madicp_host_tree* t, const double R[9], const double tr[3]) {
  if (t) madicp_host::transform_tree(t->tree, R, tr);
}

void madicp_host_gn_update(const double H[36], const double b[6], double X[12]) {
  using namespace madicp_host;
  double nb[6], dx[6];
  for (int i = 0; i < 6; ++i) nb[i] = -b[i];
  ldlt6_solve(H, nb, dx);
  Pose cur, d;
  std::memcpy(cur.R, X, sizeof(cur.R));
  std::memcpy(cur.t, X + 9, sizeof(cur.t));
  exp_so3(dx + 3, d.R);
  d.t[0] = dx[0]; d.t[1] = dx[1]; d.t[2] = dx[2];
  const Pose out = compose(cur, d);
  std::memcpy(X, out.R, sizeof(out.R));
  std::memcpy(X + 9, out.t, sizeof(out.t));
}
double madicp_host_det_of_inverse6(const double H[36]) { return madicp_host::det_of_inverse6(H); }

double madicp_host_tree_rho2(const madicp_host_tree* t) { return t ? t->tree.rho2 : -1.0; }
void madicp_host_set_threads(int n) { madicp_host::TaskPool::instance().set_limit(n); }

int64_t madicp_host_debug_partition(double* points, int64_t n, const double mean[3], const double normal[3], int impl) {
  if (!points || n < 0 || !mean || !normal) return -1;
  return madicp_host::debug_partition(points, n, mean, normal, impl);
}

int64_t madicp_host_debug_tree_points(double* points, int64_t n, double b_max, double b_min, int max_parallel_level) {
  if (!points || n <= 0) return -1;
  return madicp_host::build_tree(points, n, b_max, b_min, max_parallel_level).num_leaves();
}

// Pipeline::deskew on its own (csrc/host/deskew.h).  route 0: the parallel azimuth order, the reference's serial sort when
// two azimuths tie; route 1: the reference's route always.  Returns 1 when the parallel order was used, 0 when the serial
// route ran, < 0 on bad arguments.
int madicp_host_debug_deskew(double* points, int64_t n, const double T_prev[12], const double T_now[12], double sensor_hz, int route,
                             double* out_velocity6) {
  if (!points || n < 0 || !T_prev || !T_now || !(sensor_hz > 0.0)) return -1;
  madicp_host::ContainerType cloud(static_cast<size_t>(n));
  if (n) std::memcpy(static_cast<void*>(cloud.data()), points, sizeof(double) * 3 * static_cast<size_t>(n));
  madicp_host::Pose a, b;
  std::memcpy(a.R, T_prev, sizeof(a.R));
  std::memcpy(a.t, T_prev + 9, sizeof(a.t));
  std::memcpy(b.R, T_now, sizeof(b.R));
  std::memcpy(b.t, T_now + 9, sizeof(b.t));
  madicp_host::DeskewOrder prep = madicp_host::deskew_order(cloud);
  if (route == 1) prep.ties = true;
  const int fast = prep.ties ? 0 : 1;
  madicp_host::deskew_cloud(cloud, a, b, sensor_hz, &prep, out_velocity6);
  if (n) std::memcpy(points, static_cast<const void*>(cloud.data()), sizeof(double) * 3 * static_cast<size_t>(n));
  return fast;
}

}  // extern "C"
