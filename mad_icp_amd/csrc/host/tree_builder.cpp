#include "tree_builder.h"

#include <cmath>
#include <cstring>
#include <future>
#include <limits>

#include "linalg.h"

namespace madicp_host {
namespace {

struct Ctx {
  double* pts;  // xyz triples
  double b_max, b_min;
  int max_parallel_level;
};

// what a leaf may need from its ancestors (reference mad_tree.cpp:64-74)
struct Inherited {
  const double* plane_normal;  // col 0 of the top-most flat ancestor (bbox0 < b_min), or null
  const double* small_normal;  // col 0 of the nearest ancestor holding >= 3 points (or the root), or null at the root
};

inline double* P(const Ctx& c, int64_t i) { return c.pts + 3 * i; }

// utils.h:54-73: one pass, mean and sample covariance (full 3x3 accumulated, symmetric by construction)
void mean_cov(const Ctx& c, int64_t b, int64_t e, double* mean, double* cov /*row-major 9*/) {
  double m[3] = {0, 0, 0};
  double s[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int64_t i = b; i < e; ++i) {
    const double* v = P(c, i);
    m[0] += v[0]; m[1] += v[1]; m[2] += v[2];
    s[0] += v[0] * v[0]; s[1] += v[0] * v[1]; s[2] += v[0] * v[2];
    s[3] += v[1] * v[0]; s[4] += v[1] * v[1]; s[5] += v[1] * v[2];
    s[6] += v[2] * v[0]; s[7] += v[2] * v[1]; s[8] += v[2] * v[2];
  }
  const int k = static_cast<int>(e - b);
  const double inv_k = 1. / k;
  for (int i = 0; i < 3; ++i) mean[i] = m[i] * inv_k;
  const double bessel = double(k) / double(k - 1);
  for (int r = 0; r < 3; ++r)
    for (int q = 0; q < 3; ++q) {
      double x = s[3 * r + q] * inv_k;
      x -= mean[r] * mean[q];
      cov[3 * r + q] = x * bessel;
    }
}

// utils.h:75-97: extents of the points in the eigen frame, 0 included; min/max keep the running value on NaN
void bbox_extents(const Ctx& c, int64_t b, int64_t e, const double* mean, const double* V /*row-major, cols = eigvecs*/,
                  double* ext) {
  double lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
  const double c0[3] = {V[0], V[3], V[6]}, c1[3] = {V[1], V[4], V[7]}, c2[3] = {V[2], V[5], V[8]};
  for (int64_t i = b; i < e; ++i) {
    const double* p = P(c, i);
    const double d[3] = {p[0] - mean[0], p[1] - mean[1], p[2] - mean[2]};
    const double v[3] = {dot3c(c0, d), dot3c(c1, d), dot3c(c2, d)};
    for (int a = 0; a < 3; ++a) {
      if (v[a] < lo[a]) lo[a] = v[a];
      if (hi[a] < v[a]) hi[a] = v[a];
    }
  }
  for (int a = 0; a < 3; ++a) ext[a] = hi[a] - lo[a];
}

// utils.h:37-52: partition; the non-matching element is swapped with the one just below the upper cursor
int64_t partition_by_plane(const Ctx& c, int64_t b, int64_t e, const double* mean, const double* normal) {
  int64_t lo = b, hi = e;
  while (lo != hi) {
    double* p = P(c, lo);
    const double d[3] = {p[0] - mean[0], p[1] - mean[1], p[2] - mean[2]};
    if (dot3c(d, normal) < 0.0) {
      ++lo;
    } else {
      double* q = P(c, hi - 1);
      for (int a = 0; a < 3; ++a) {
        const double t = p[a];
        p[a] = q[a];
        q[a] = t;
      }
      --hi;
    }
  }
  return hi;
}

void build_range(const Ctx& c, int64_t b, int64_t e, int level, Inherited inh, std::vector<madicp_node>& out) {
  const int n_pts = static_cast<int>(e - b);
  double mean[3], cov[9], w[3], V[9], ext[3];
  mean_cov(c, b, e, mean, cov);
  eig3_sym(cov, w, V);
  bbox_extents(c, b, e, mean, V, ext);

  const size_t self = out.size();
  out.emplace_back();
  madicp_node nd;
  nd.bbox0 = ext[0];

  if (ext[2] < c.b_max) {  // leaf: mad_tree.cpp:64-88
    double normal[3] = {V[0], V[3], V[6]};
    if (inh.plane_normal) {
      std::memcpy(normal, inh.plane_normal, sizeof(normal));
    } else if (n_pts < 3 && inh.small_normal) {
      std::memcpy(normal, inh.small_normal, sizeof(normal));
    }
    // surface point = the member nearest to the centroid; the reference writes the winner through a
    // reference to *begin (mad_tree.cpp:76), mirrored here so the cloud ends up in the same state
    double* first = P(c, b);
    double shortest = std::numeric_limits<double>::max();
    for (int64_t i = b; i < e; ++i) {
      const double* p = P(c, i);
      const double v[3] = {p[0], p[1], p[2]};
      const double d[3] = {v[0] - mean[0], v[1] - mean[1], v[2] - mean[2]};
      const double dist = norm3(d);
      if (dist < shortest) {
        first[0] = v[0]; first[1] = v[1]; first[2] = v[2];
        shortest = dist;
      }
    }
    std::memcpy(nd.mean, first, sizeof(nd.mean));
    std::memcpy(nd.dir, normal, sizeof(nd.dir));
    nd.right = 0;
    nd.leaf_id = 0;  // assigned after the splice
    out[self] = nd;
    return;
  }

  const double col0[3] = {V[0], V[3], V[6]};
  const double col2[3] = {V[2], V[5], V[8]};
  if (!inh.plane_normal && ext[0] < c.b_min) inh.plane_normal = col0;  // mad_tree.cpp:90-93
  if (n_pts >= 3 || !inh.small_normal) inh.small_normal = col0;         // mad_tree.cpp:68-72, walked top-down

  const int64_t mid = partition_by_plane(c, b, e, mean, col2);

  std::memcpy(nd.mean, mean, sizeof(nd.mean));
  std::memcpy(nd.dir, col2, sizeof(nd.dir));
  nd.leaf_id = -1;

  if (level >= c.max_parallel_level) {
    build_range(c, b, mid, level + 1, inh, out);
    nd.right = static_cast<int32_t>(out.size() - self);
    build_range(c, mid, e, level + 1, inh, out);
  } else {
    auto task = [&c, level, inh](int64_t tb, int64_t te) {
      std::vector<madicp_node> sub;
      sub.reserve(static_cast<size_t>(2 * (te - tb)));
      build_range(c, tb, te, level + 1, inh, sub);
      return sub;
    };
    std::future<std::vector<madicp_node>> fl = std::async(std::launch::async, task, b, mid);
    std::future<std::vector<madicp_node>> fr = std::async(std::launch::async, task, mid, e);
    const std::vector<madicp_node> l = fl.get();
    const std::vector<madicp_node> r = fr.get();
    out.insert(out.end(), l.begin(), l.end());
    nd.right = static_cast<int32_t>(out.size() - self);
    out.insert(out.end(), r.begin(), r.end());
  }
  out[self] = nd;
}

}  // namespace

LinearTree build_tree(double* points, int64_t n, double b_max, double b_min, int max_parallel_level) {
  LinearTree t;
  if (n <= 0) return t;
  Ctx c{points, b_max, b_min, max_parallel_level};
  t.nodes.reserve(static_cast<size_t>(2 * n));
  build_range(c, 0, n, 0, Inherited{nullptr, nullptr}, t.nodes);
  t.nodes.shrink_to_fit();
  // getLeafs() order == order of appearance in the preorder array (mad_tree.cpp:154-163)
  for (size_t i = 0; i < t.nodes.size(); ++i)
    if (t.nodes[i].right == 0) {
      t.nodes[i].leaf_id = static_cast<int32_t>(t.leaf_nodes.size());
      t.leaf_nodes.push_back(static_cast<int32_t>(i));
    }
  return t;
}

void transform_tree(LinearTree& tree, const double* R, const double* t) {
  for (madicp_node& nd : tree.nodes) {
    double m[3], d[3];
    matvec3(R, nd.mean, m);
    matvec3(R, nd.dir, d);
    for (int i = 0; i < 3; ++i) {
      nd.mean[i] = m[i] + t[i];
      nd.dir[i] = d[i];
    }
  }
}

}  // namespace madicp_host
