#include "tree_builder.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <chrono>
#include <cstdio>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>

#if defined(__SSE2__)
#include <emmintrin.h>
#endif

#include "linalg.h"
#include "task_pool.h"

namespace madicp_host {
namespace {

// development timeline (environment MADICP_HOST_TRACE=<file>): one line per forked node / sequential chunk of the LAST
// build: kind level points t_start_us t_end_us thread — where the parallel build's wall time goes
struct TraceEv {
  char kind;
  int level;
  int64_t n;
  double t0, t1;
  size_t tid;
};
std::mutex g_trace_mu;
std::vector<TraceEv> g_trace;
const char* g_trace_path = std::getenv("MADICP_HOST_TRACE");
std::chrono::steady_clock::time_point g_trace_t0;
inline double trace_now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - g_trace_t0).count(); }
inline void trace_add(char kind, int level, int64_t n, double t0) {
  if (!g_trace_path) return;
  const double t1 = trace_now();
  std::lock_guard<std::mutex> lk(g_trace_mu);
  g_trace.push_back(TraceEv{kind, level, n, t0, t1, std::hash<std::thread::id>()(std::this_thread::get_id()) % 1000});
}

struct Ctx {
  double* pts;  // xyz triples
  double b_max, b_min;
  int max_parallel_level;  // levels above this depth run as tasks (0: everything on the calling thread)
  // scratch, one entry per point, indexed like pts: a node only touches the entries of its own range [b, e), so the
  // sub-trees built by different threads never share one
  uint8_t* side;    // 1: the point lies on the negative side of its node's split plane (goes left)
  int32_t* reject;  // partition_from_flags: positions of the points that go right, ascending
  bool reference_loop;  // development A/B (environment MADICP_HOST_PARTITION=loop): partition with the reference's swap loop
};

// Task policy.  The reference forks two std::async threads per node above `max_parallel_level` and the parent waits
// (mad_tree.cpp:99-117): 2^level leaf tasks, badly balanced because MAD-trees split at the mean, not the median.
// Here the same argument buys 4x as many, smaller tasks (two more levels), the parent builds one child itself, a
// range below kTaskMinPoints is never forked, and the order-independent pass of a big node (the bounding box) is cut
// over idle threads.  None of this can change a bit of the result: every sum keeps its sequential order.
int env_int(const char* name, int dflt) {
  const char* e = std::getenv(name);
  return e ? std::atoi(e) : dflt;
}
// (development overrides: MADICP_HOST_EXTRA_LEVELS, MADICP_HOST_TASK_MIN)
const int kExtraTaskLevels = env_int("MADICP_HOST_EXTRA_LEVELS", 2);
const int64_t kTaskMinPoints = env_int("MADICP_HOST_TASK_MIN", 2048);  // (4096: 1.80 ms, 2048: 1.67 ms, 1024: 1.68 ms per 120 k scan)
// a range of fewer points is not cut (development override: environment MADICP_HOST_BBOX_SLICE_MIN).  Cutting the
// 120 k-point root's pass over the pool — order-independent, so legal — was measured again in round 3 with the pool fixed
// and the workers pre-woken: the root goes from 0.34 to 0.46-0.70 ms.  The slices leave the points in other cores' caches in
// shared state and the partition that follows WRITES them: every written line first has to be taken back across the fabric.
int64_t bbox_slice_min_points() {
  static const int64_t v = [] {
    const char* e = std::getenv("MADICP_HOST_BBOX_SLICE_MIN");
    return e ? std::atoll(e) : (int64_t(1) << 20);
  }();
  return v;
}

// what a leaf may need from its ancestors (reference mad_tree.cpp:64-74)
struct Inherited {
  const double* plane_normal;  // col 0 of the top-most flat ancestor (bbox0 < b_min), or null
  const double* small_normal;  // col 0 of the nearest ancestor holding >= 3 points (or the root), or null at the root
};

inline double* P(const Ctx& c, int64_t i) { return c.pts + 3 * i; }

// utils.h:54-73: one pass, mean and sample covariance (full 3x3 accumulated, symmetric by construction)
void mean_cov(const Ctx& c, int64_t b, int64_t e, double* mean, double* cov /*row-major 9*/) {
  double m[3] = {0, 0, 0};
  // the reference accumulates all nine products; v_r*v_q and v_q*v_r are the same double, so six sums give the same
  // nine values
  double s00 = 0, s01 = 0, s02 = 0, s11 = 0, s12 = 0, s22 = 0;
#if defined(__SSE2__)
  {  // the same nine running sums, two per register: every lane is its own sequential chain, as in the scalar loop
    __m128d m01 = _mm_setzero_pd(), m2_ = _mm_setzero_pd();
    __m128d a = _mm_setzero_pd(), bb = _mm_setzero_pd(), cc = _mm_setzero_pd();  // (s00,s01) (s02,s11) (s12,s22)
    for (int64_t i = b; i < e; ++i) {
      const double* v = P(c, i);
      const __m128d xy = _mm_loadu_pd(v);          // (x, y)
      const __m128d zz = _mm_set1_pd(v[2]);        // (z, z)
      const __m128d xx = _mm_unpacklo_pd(xy, xy);  // (x, x)
      const __m128d yy = _mm_unpackhi_pd(xy, xy);  // (y, y)
      m01 = _mm_add_pd(m01, xy);
      m2_ = _mm_add_sd(m2_, zz);
      a = _mm_add_pd(a, _mm_mul_pd(xx, xy));                         // x*x, x*y
      bb = _mm_add_pd(bb, _mm_mul_pd(xy, _mm_shuffle_pd(zz, xy, 2)));  // x*z, y*y
      cc = _mm_add_pd(cc, _mm_mul_pd(_mm_shuffle_pd(yy, zz, 0), zz));  // y*z, z*z
    }
    double t[2];
    _mm_storeu_pd(t, m01); m[0] = t[0]; m[1] = t[1];
    m[2] = _mm_cvtsd_f64(m2_);
    _mm_storeu_pd(t, a); s00 = t[0]; s01 = t[1];
    _mm_storeu_pd(t, bb); s02 = t[0]; s11 = t[1];
    _mm_storeu_pd(t, cc); s12 = t[0]; s22 = t[1];
  }
#else
  for (int64_t i = b; i < e; ++i) {
    const double* v = P(c, i);
    m[0] += v[0]; m[1] += v[1]; m[2] += v[2];
    s00 += v[0] * v[0]; s01 += v[0] * v[1]; s02 += v[0] * v[2];
    s11 += v[1] * v[1]; s12 += v[1] * v[2];
    s22 += v[2] * v[2];
  }
#endif
  const double s[9] = {s00, s01, s02, s01, s11, s12, s02, s12, s22};
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

// utils.h:75-97: extents of the points in the eigen frame, 0 included; min/max keep the running value on NaN.
// min and max do not depend on the order of the points, so a big range is cut into slices.
// The third projection IS the value the split predicate tests: `(p - mean).dot(eigenvectors.col(2)) < 0`
// (mad_tree.cpp:95-97) and row 2 of `R * (p - mean)` (utils.h:89) are the same three products added in the same order —
// so this pass also records every point's side of the split plane (NaN: false, like the reference's `< 0`), and the
// partition that follows for an internal node needs no floating-point arithmetic at all.
void bbox_lohi(const Ctx& c, int64_t b, int64_t e, const double* mean, const double* V, double* lo, double* hi) {
  uint8_t* side = c.side;
#if defined(__SSE2__) && !defined(MADICP_REDUX_SCALAR_ONLY)
  // two of the three projections per instruction; every lane does exactly the scalar sequence
  // (c_a0 d0 + c_a1 d1) + c_a2 d2, and min_pd/max_pd keep their SECOND operand on NaN or equality — the running value,
  // like the `if (v < lo) lo = v` of the scalar loop below
  const __m128d A0 = _mm_set_pd(V[1], V[0]), A1 = _mm_set_pd(V[4], V[3]), A2 = _mm_set_pd(V[7], V[6]);
  const double c20 = V[2], c21 = V[5], c22 = V[8];
  __m128d lo01 = _mm_set_pd(lo[1], lo[0]), hi01 = _mm_set_pd(hi[1], hi[0]);
  __m128d lo2 = _mm_set_sd(lo[2]), hi2 = _mm_set_sd(hi[2]);
  const double m0 = mean[0], m1 = mean[1], m2 = mean[2];
  for (int64_t i = b; i < e; ++i) {
    const double* p = P(c, i);
    const double d0 = p[0] - m0, d1 = p[1] - m1, d2 = p[2] - m2;
    const __m128d v01 = _mm_add_pd(_mm_add_pd(_mm_mul_pd(A0, _mm_set1_pd(d0)), _mm_mul_pd(A1, _mm_set1_pd(d1))),
                                   _mm_mul_pd(A2, _mm_set1_pd(d2)));
    const double s2 = (c20 * d0 + c21 * d1) + c22 * d2;
    side[i] = s2 < 0.0 ? 1 : 0;
    const __m128d v2 = _mm_set_sd(s2);
    lo01 = _mm_min_pd(v01, lo01);
    hi01 = _mm_max_pd(v01, hi01);
    lo2 = _mm_min_sd(v2, lo2);
    hi2 = _mm_max_sd(v2, hi2);
  }
  double t[2];
  _mm_storeu_pd(t, lo01); lo[0] = t[0]; lo[1] = t[1];
  _mm_storeu_pd(t, hi01); hi[0] = t[0]; hi[1] = t[1];
  lo[2] = _mm_cvtsd_f64(lo2);
  hi[2] = _mm_cvtsd_f64(hi2);
#else
  const double c0[3] = {V[0], V[3], V[6]}, c1[3] = {V[1], V[4], V[7]}, c2[3] = {V[2], V[5], V[8]};
  for (int64_t i = b; i < e; ++i) {
    const double* p = P(c, i);
    const double d[3] = {p[0] - mean[0], p[1] - mean[1], p[2] - mean[2]};
    const double v[3] = {dot3c(c0, d), dot3c(c1, d), dot3c(c2, d)};
    side[i] = v[2] < 0.0 ? 1 : 0;
    for (int a = 0; a < 3; ++a) {
      if (v[a] < lo[a]) lo[a] = v[a];
      if (hi[a] < v[a]) hi[a] = v[a];
    }
  }
#endif
}

void bbox_extents(const Ctx& c, int64_t b, int64_t e, const double* mean, const double* V /*row-major, cols = eigvecs*/,
                  double* ext, int slices) {
  double lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
  if (slices <= 1 || e - b < bbox_slice_min_points()) {
    bbox_lohi(c, b, e, mean, V, lo, hi);
  } else {
    struct LoHi { double lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0}; };
    std::vector<LoHi> part(static_cast<size_t>(slices));
    std::vector<TaskPool::Handle> jobs;
    TaskPool& pool = TaskPool::instance();
    const int64_t step = (e - b + slices - 1) / slices;
    for (int s = 1; s < slices; ++s) {
      const int64_t sb = b + s * step, se = std::min(e, sb + step);
      if (sb >= se) break;
      jobs.push_back(pool.submit([&c, sb, se, mean, V, &part, s] { bbox_lohi(c, sb, se, mean, V, part[s].lo, part[s].hi); }));
    }
    bbox_lohi(c, b, std::min(e, b + step), mean, V, part[0].lo, part[0].hi);
    for (const TaskPool::Handle& j : jobs) pool.wait(j);
    for (const LoHi& q : part)
      for (int a = 0; a < 3; ++a) {
        if (q.lo[a] < lo[a]) lo[a] = q.lo[a];
        if (hi[a] < q.hi[a]) hi[a] = q.hi[a];
      }
  }
  for (int a = 0; a < 3; ++a) ext[a] = hi[a] - lo[a];
}

// utils.h:37-52: partition; the non-matching element is swapped with the one just below the upper cursor
int64_t partition_by_plane(const Ctx& c, int64_t b, int64_t e, const double* mean, const double* normal) {
  // same sequence of swaps as the reference's loop, written without a data-dependent branch (the side test is a coin
  // flip to the branch predictor): both elements are rewritten every step, swapped or not
  int64_t lo = b, hi = e;
  while (lo != hi) {
    double* p = P(c, lo);
    double* q = P(c, hi - 1);
    const double p0 = p[0], p1 = p[1], p2 = p[2];
    const double q0 = q[0], q1 = q[1], q2 = q[2];
    const double d[3] = {p0 - mean[0], p1 - mean[1], p2 - mean[2]};
    const bool keep = dot3c(d, normal) < 0.0;
    p[0] = keep ? p0 : q0; p[1] = keep ? p1 : q1; p[2] = keep ? p2 : q2;
    q[0] = keep ? q0 : p0; q[1] = keep ? q1 : p1; q[2] = keep ? q2 : p2;
    lo += keep ? 1 : 0;
    hi -= keep ? 0 : 1;
  }
  return hi;
}

// The same permutation as partition_by_plane, from the side flags the bounding-box pass left behind — without the
// reference loop's dependency chain (every step of `split` needs the previous decision to know which element it looks
// at next: load -> subtract -> dot -> compare -> cursor, ~27 cycles per point, 1 ms for the root of a 120 k-point scan).
//
// What utils.h:37-52 does, in closed form.  The lower cursor walks the front of the range; whenever it meets a point
// that has to go right (a "hole"), that point is swapped to the top of the not yet examined back part and the point
// that was there is examined in its place, again and again until one of them stays.  Hence, with the holes h_1 < h_2 <
// ... (front points that go right) and the back points that go left k_1 > k_2 > ... (descending positions):
//   * hole h_r ends up holding the point of k_r;
//   * every back point that goes right ends up ONE position lower (the swap that examined it put it there);
//   * the point of hole h_r ends up at the top of "its" run: position e - 1 for r = 1, k_{r-1} - 1 after that;
//   * it ends when the cursors meet: at k_m if the front has no hole left below it, else at the hole h_{m+1} that no
//     back point is left to fill.
// One descending sweep over the back part does all of it: `carry` is what has to be written at the position the sweep
// stands on — the hole's point at the top of a run, the shifted point below it.  Bit-identical by construction and
// checked against the loop above on exhaustive small and random large flag patterns (tests/test_host_builder.py).
int64_t partition_from_flags(const Ctx& c, int64_t b, int64_t e) {
  const uint8_t* side = c.side;
  int32_t* rej = c.reject + b;
  // positions that go right, ascending (branch-free compaction: the slot is written every step, kept when it counts)
  int64_t n_rej = 0;
  for (int64_t p = b; p < e; ++p) {
    rej[n_rej] = static_cast<int32_t>(p);
    n_rej += side[p] ? 0 : 1;
  }
  if (n_rej == 0) return e;  // everything stays
  // (points as three 64-bit words: the selects below become conditional moves — the side of a point is a coin flip to
  // the branch predictor, and a mispredicted branch per point is what the sweep would otherwise cost)
  struct P3 { uint64_t x, y, z; };
  static_assert(sizeof(P3) == 24, "a point is three doubles");
  P3* pts = reinterpret_cast<P3*>(c.pts);
  P3 sink;
  int64_t r = 0;
  int64_t hole = rej[0];
  int64_t q = e - 1;
  P3 carry = pts[hole];
  for (;;) {
    if (q == hole) {  // the back part is used up: the hole is where the two parts meet
      pts[q] = carry;
      return hole;
    }
    const P3 y = pts[q];
    pts[q] = carry;
    const bool left = side[q] != 0;
    // goes left: it fills the hole, and the next hole's point opens the next run one position down; goes right: it is
    // what the next position down receives
    P3* dst = left ? &pts[hole] : &sink;
    *dst = y;
    r += left ? 1 : 0;
    const int64_t next = r < n_rej ? rej[r] : e;
    if (left && next >= q) return q;  // no hole left below q: the front walks up to q undisturbed
    hole = left ? next : hole;
    const P3 nc = pts[hole];
    carry.x = left ? nc.x : y.x;
    carry.y = left ? nc.y : y.y;
    carry.z = left ? nc.z : y.z;
    --q;
  }
}

// one node: statistics, leaf test, and — for an internal node — the partition.  Returns true for a leaf.
// `col0` / `mid` are only set for internal nodes; `col0` must outlive the children (they may inherit it).
bool make_node(const Ctx& c, int64_t b, int64_t e, Inherited& inh, madicp_node& nd, double* col0, int64_t& mid,
               int bbox_slices) {
  const int n_pts = static_cast<int>(e - b);
  double mean[3], cov[9], w[3], V[9], ext[3];
  mean_cov(c, b, e, mean, cov);
  eig3_sym(cov, w, V);
  bbox_extents(c, b, e, mean, V, ext, bbox_slices);
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
    nd.leaf_id = 0;  // assigned after the flattening
    return true;
  }

  col0[0] = V[0]; col0[1] = V[3]; col0[2] = V[6];
  const double col2[3] = {V[2], V[5], V[8]};
  if (!inh.plane_normal && ext[0] < c.b_min) inh.plane_normal = col0;  // mad_tree.cpp:90-93
  if (n_pts >= 3 || !inh.small_normal) inh.small_normal = col0;         // mad_tree.cpp:68-72, walked top-down

  mid = c.reference_loop ? partition_by_plane(c, b, e, mean, col2) : partition_from_flags(c, b, e);  // (same permutation)

  std::memcpy(nd.mean, mean, sizeof(nd.mean));
  std::memcpy(nd.dir, col2, sizeof(nd.dir));
  nd.leaf_id = -1;
  return false;
}

// a whole sub-tree on the calling thread, appended to `out` in preorder
void build_sequential(const Ctx& c, int64_t b, int64_t e, Inherited inh, NodeVec& out) {
  const size_t self = out.size();
  out.emplace_back();
  madicp_node nd;
  double col0[3];
  int64_t mid = 0;
  if (make_node(c, b, e, inh, nd, col0, mid, 1)) {
    out[self] = nd;
    return;
  }
  build_sequential(c, b, mid, inh, out);
  nd.right = static_cast<int32_t>(out.size() - self);
  build_sequential(c, mid, e, inh, out);
  out[self] = nd;
}

// The forked top of the tree: a node of it is either one record with two children, or a chunk (a sub-tree that one
// thread built, already in preorder with relative offsets — position independent).
struct Piece {
  NodeVec chunk;                   // non-empty: a finished sub-tree
  madicp_node nd;                  // else: this node ...
  std::unique_ptr<Piece> left, right;  // ... and its sub-trees
  size_t size = 0;                 // nodes in this piece
};

std::unique_ptr<Piece> build_forked(const Ctx& c, int64_t b, int64_t e, int level, Inherited inh, int bbox_slices) {
  auto piece = std::make_unique<Piece>();
  if (level >= c.max_parallel_level || e - b < kTaskMinPoints) {
    const double t0 = g_trace_path ? trace_now() : 0.0;
    piece->chunk.reserve(static_cast<size_t>(std::min<int64_t>(2 * (e - b), 1 << 16)));
    build_sequential(c, b, e, inh, piece->chunk);
    piece->size = piece->chunk.size();
    trace_add('S', level, e - b, t0);
    return piece;
  }
  double col0[3];  // lives until both children are done (they may point at it through `inh`)
  int64_t mid = 0;
  const double t_node = g_trace_path ? trace_now() : 0.0;
  const bool is_leaf = make_node(c, b, e, inh, piece->nd, col0, mid, bbox_slices);
  trace_add('N', level, e - b, t_node);
  if (is_leaf) {
    piece->chunk.push_back(piece->nd);
    piece->size = 1;
    return piece;
  }
  // the SMALLER child as a pool task, the larger one here: the points were just written by this core, a child that
  // moves to another core (another L3 on the hosts this runs on) reads them at half the speed (timeline: 347 us against
  // 155 us for the two halves of a 120 k-point root), so the cold start is given to the half with less to read.  A level
  // down there are twice as many busy threads, so the bounding-box pass gets half the slices.
  const int child_slices = std::max(1, bbox_slices / 2);
  TaskPool& pool = TaskPool::instance();
  Piece* self = piece.get();
  TaskPool::Handle j_other;
  if (mid - b <= e - mid) {
    j_other = pool.submit([&c, self, b, mid, level, inh, child_slices] {
      self->left = build_forked(c, b, mid, level + 1, inh, child_slices);
    });
    piece->right = build_forked(c, mid, e, level + 1, inh, child_slices);
  } else {
    j_other = pool.submit([&c, self, e, mid, level, inh, child_slices] {
      self->right = build_forked(c, mid, e, level + 1, inh, child_slices);
    });
    piece->left = build_forked(c, b, mid, level + 1, inh, child_slices);
  }
  pool.wait(j_other);
  piece->size = 1 + piece->left->size + piece->right->size;
  piece->nd.right = static_cast<int32_t>(1 + piece->left->size);
  return piece;
}

// preorder = this node, the left piece, the right piece.  The forked nodes are written here; the chunks (whole
// sub-trees, each with (size + 1) / 2 leaves) are only listed, with their node and leaf offsets, and copied by tasks.
struct ChunkRef {
  const Piece* piece;
  size_t node_off, leaf_off;
};
void layout(const Piece& p, size_t node_off, size_t& leaf_off, madicp_node* out, std::vector<ChunkRef>& chunks) {
  if (!p.chunk.empty()) {
    chunks.push_back(ChunkRef{&p, node_off, leaf_off});
    leaf_off += (p.chunk.size() + 1) / 2;
    return;
  }
  out[node_off] = p.nd;
  layout(*p.left, node_off + 1, leaf_off, out, chunks);
  layout(*p.right, node_off + 1 + p.left->size, leaf_off, out, chunks);
}

// |mean - origin|_2 of an internal node, 0 when it is not finite (like madicp_tree_upload's validation pass)
inline double spread_of(const madicp_node& nd, const double* o) {
  const double e0 = nd.mean[0] - o[0], e1 = nd.mean[1] - o[1], e2 = nd.mean[2] - o[2];
  const double r = std::sqrt((e0 * e0 + e1 * e1) + e2 * e2);
  return std::isfinite(r) ? r : 0.0;
}

// copy one chunk into place and number its leaves (getLeafs() order == order of appearance in preorder); returns the
// largest spread_of() among its internal nodes
double place_chunk(const ChunkRef& r, madicp_node* out, int32_t* leaf_nodes, const double* origin) {
  const NodeVec& ch = r.piece->chunk;
  std::memcpy(out + r.node_off, ch.data(), ch.size() * sizeof(madicp_node));
  size_t leaf = r.leaf_off;
  double rho = 0.0;
  for (size_t i = 0; i < ch.size(); ++i)
    if (ch[i].right == 0) {
      out[r.node_off + i].leaf_id = static_cast<int32_t>(leaf);
      leaf_nodes[leaf++] = static_cast<int32_t>(r.node_off + i);
    } else {
      rho = std::max(rho, spread_of(ch[i], origin));
    }
  return rho;
}

}  // namespace

LinearTree build_tree(double* points, int64_t n, double b_max, double b_min, int max_parallel_level) {
  LinearTree t;
  if (n <= 0) return t;
  const int hw = static_cast<int>(std::max(1u, std::thread::hardware_concurrency()));
  int levels = max_parallel_level > 0 ? max_parallel_level + kExtraTaskLevels : 0;
  while (levels > 0 && (1 << (levels - 1)) > 2 * hw) --levels;  // never far more leaf tasks than cores
  // per-point scratch of the partition (uninitialised: every node writes its own range before it reads it)
  std::unique_ptr<uint8_t[]> side(new uint8_t[static_cast<size_t>(n)]);
  std::unique_ptr<int32_t[]> reject(new int32_t[static_cast<size_t>(n)]);
  const char* how = std::getenv("MADICP_HOST_PARTITION");
  Ctx c{points, b_max, b_min, levels, side.get(), reject.get(), how && std::strcmp(how, "loop") == 0};
  if (levels == 0) {
    t.nodes.reserve(static_cast<size_t>(2 * n));
    build_sequential(c, 0, n, Inherited{nullptr, nullptr}, t.nodes);
    t.nodes.shrink_to_fit();
    // getLeafs() order == order of appearance in the preorder array (mad_tree.cpp:154-163)
    t.leaf_nodes.reserve((t.nodes.size() + 1) / 2);
    double rho = 0.0;
    for (size_t i = 0; i < t.nodes.size(); ++i)
      if (t.nodes[i].right == 0) {
        t.nodes[i].leaf_id = static_cast<int32_t>(t.leaf_nodes.size());
        t.leaf_nodes.push_back(static_cast<int32_t>(i));
      } else {
        rho = std::max(rho, spread_of(t.nodes[i], t.nodes[0].mean));
      }
    t.rho2 = rho;
    return t;
  }
  const int slices = std::min(hw, 1 << std::min(levels, 4));
  if (g_trace_path) {
    std::lock_guard<std::mutex> lk(g_trace_mu);
    g_trace.clear();
    g_trace_t0 = std::chrono::steady_clock::now();
  }
  const std::unique_ptr<Piece> top = build_forked(c, 0, n, 0, Inherited{nullptr, nullptr}, slices);
  const double t_forked = g_trace_path ? trace_now() : 0.0;
  t.nodes.resize(top->size);  // (uninitialised: every element is written below)
  t.leaf_nodes.resize((top->size + 1) / 2);
  std::vector<ChunkRef> chunks;
  size_t n_leaves = 0;
  layout(*top, 0, n_leaves, t.nodes.data(), chunks);
  trace_add('L', 0, static_cast<int64_t>(chunks.size()), t_forked);  // allocation + layout
  // a handful of copy tasks, each a contiguous run of chunks of about the same number of nodes
  TaskPool& pool = TaskPool::instance();
  const size_t n_tasks = std::min<size_t>(16, chunks.size());
  std::vector<std::function<void()>> copy_fns;
  std::function<void()> own_copy;
  madicp_node* out = t.nodes.data();
  int32_t* leaf_nodes = t.leaf_nodes.data();
  // (the root is a forked node unless the whole tree is one chunk: either way its mean is final before any copy task
  // starts — layout() wrote it, or it is element 0 of the only chunk)
  double origin[3];
  std::memcpy(origin, top->chunk.empty() ? top->nd.mean : top->chunk[0].mean, sizeof(origin));
  std::vector<double> rho_part(n_tasks + 1, 0.0);
  size_t first = 0;
  for (size_t k = 0; k < n_tasks; ++k) {
    const size_t want = top->size * (k + 1) / n_tasks;  // nodes up to which this task copies
    size_t last = first;
    while (last < chunks.size() && (k + 1 == n_tasks || chunks[last].node_off < want)) ++last;
    if (last == first) continue;
    const ChunkRef* cb = chunks.data() + first;
    const ChunkRef* ce = chunks.data() + last;
    double* rho_out = &rho_part[k];
    auto run = [cb, ce, out, leaf_nodes, rho_out, &origin] {
      const double t0 = g_trace_path ? trace_now() : 0.0;
      double rho = 0.0;
      int64_t nn = 0;
      for (const ChunkRef* r = cb; r != ce; ++r) {
        rho = std::max(rho, place_chunk(*r, out, leaf_nodes, origin));
        nn += static_cast<int64_t>(r->piece->chunk.size());
      }
      *rho_out = rho;
      trace_add('P', 0, nn, t0);
    };
    if (k + 1 == n_tasks) own_copy = run; else copy_fns.push_back(run);
    first = last;
  }
  const std::vector<TaskPool::Handle> jobs = pool.submit_batch(std::move(copy_fns));  // one lock, one wake-up
  if (own_copy) own_copy();
  for (const TaskPool::Handle& j : jobs) pool.wait(j);
  // the forked nodes (a few dozen) were written by layout()
  {
    std::vector<const Piece*> stack{top.get()};
    while (!stack.empty()) {
      const Piece* p = stack.back();
      stack.pop_back();
      if (!p->chunk.empty()) continue;
      rho_part[n_tasks] = std::max(rho_part[n_tasks], spread_of(p->nd, origin));
      stack.push_back(p->left.get());
      stack.push_back(p->right.get());
    }
  }
  t.rho2 = *std::max_element(rho_part.begin(), rho_part.end());
  if (g_trace_path) {
    trace_add('C', 0, static_cast<int64_t>(top->size), t_forked);  // layout + copy tasks
    std::lock_guard<std::mutex> lk(g_trace_mu);
    if (std::FILE* f = std::fopen(g_trace_path, "w")) {
      for (const TraceEv& ev : g_trace) std::fprintf(f, "%c %d %lld %.1f %.1f %zu\n", ev.kind, ev.level, (long long)ev.n, ev.t0, ev.t1, ev.tid);
      std::fclose(f);
    }
  }
  return t;
}

// test hook (madicp_host_debug_partition): one partition of `pts` about the plane (mean, normal), by the reference's
// loop (impl 0) or by flags + closed form (impl 1); returns the split position
int64_t debug_partition(double* pts, int64_t n, const double* mean, const double* normal, int impl) {
  std::vector<uint8_t> side(static_cast<size_t>(std::max<int64_t>(n, 1)));
  std::vector<int32_t> reject(static_cast<size_t>(std::max<int64_t>(n, 1)));
  Ctx c{pts, 0.0, 0.0, 0, side.data(), reject.data(), false};
  if (impl == 0) return partition_by_plane(c, 0, n, mean, normal);
  for (int64_t i = 0; i < n; ++i) {
    const double d[3] = {pts[3 * i] - mean[0], pts[3 * i + 1] - mean[1], pts[3 * i + 2] - mean[2]};
    side[static_cast<size_t>(i)] = dot3c(d, normal) < 0.0 ? 1 : 0;
  }
  return partition_from_flags(c, 0, n);
}

void transform_tree(LinearTree& tree, const double* R, const double* t) {
  if (tree.rho2 >= 0.0) tree.rho2 *= (1.0 + 1e-12);  // rotation invariant up to rounding
  auto run = [&tree, R, t](size_t b, size_t e) {
    for (size_t k = b; k < e; ++k) {
      madicp_node& nd = tree.nodes[k];
      double m[3], d[3];
      matvec3(R, nd.mean, m);
      matvec3(R, nd.dir, d);
      for (int i = 0; i < 3; ++i) {
        nd.mean[i] = m[i] + t[i];
        nd.dir[i] = d[i];
      }
    }
  };
  const size_t n = tree.nodes.size();
  constexpr size_t kSlice = 8192;  // nodes per task: per-node work is independent
  if (n < 2 * kSlice) {
    run(0, n);
    return;
  }
  TaskPool& pool = TaskPool::instance();
  std::vector<TaskPool::Handle> jobs;
  for (size_t b = kSlice; b < n; b += kSlice) jobs.push_back(pool.submit([run, b, n] { run(b, std::min(n, b + kSlice)); }));
  run(0, kSlice);
  for (const TaskPool::Handle& j : jobs) pool.wait(j);
}

}  // namespace madicp_host
