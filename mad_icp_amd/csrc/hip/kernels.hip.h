// Device code for the MAD-ICP hot path on gfx950 (MI355X).  Single translation unit: included by
// madicp_capi.hip only.
//
// Kernels (each cites the reference function it replaces; paths relative to the reference repo):
//   moving_prep       : per moving leaf, cache |p| for the gate        (mad_icp.cpp:81, `moving->mean_.norm()`)
//   nn_descend        : batched MADtree::bestMatchingLeafFast           (mad_tree.cpp:144-152, mad_tree_wrapper.h:48-67)
//   tree_transform    : MADtree::applyTransform                         (mad_tree.cpp:165-172)
//   icp_linearize     : MADicp::update over K trees, fused transform -> descent -> gate -> e,J -> weight ->
//                       wave/block reduction of (H,b)                   (mad_icp.cpp:59-103 under pipeline.cpp:180-183)
//   icp_solve         : adder join + LDLT + expSO3 + pose update        (mad_icp.cpp:105-117)
//   icp_reduce/update : the same split in two around the RCCL all-reduce (multi-GPU)
//   icp_finish        : matched-leaf count                              (pipeline.cpp:197-204)
//
// Numerics contract: IEEE fp64, the reference's (Eigen's) operation order, NO FMA contraction — every
// branch decision (descent side test, gate) is bit-identical to the CPU path.  Enforced by the pragma
// below and by -ffp-contract=off on the command line.
//
// Why no MFMA: nothing here is a dense contraction.  Per (leaf, tree) pair the work is a ~16-step
// dependent pointer chase (56 useful bytes of a 64 B node per step) followed by ~150 flops; the 6x6
// accumulation is a reduction over pairs.  The kernel is bound by the memory system (L2 / Infinity
// Cache gather latency and bandwidth), so the levers are the ones used here: 64 B node records so one
// visit is one cache line, DFS-preorder storage so spatially coherent queries walk contiguous memory,
// an XCD-aware unit -> workgroup map so each XCD's private L2 only ever sees its own trees, many
// queries in flight per CU, and a deterministic register -> wave shuffle -> LDS -> partial reduction.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "madicp_hip.h"

#pragma clang fp contract(off)

namespace madicp {

constexpr int kBlock = 256;          // threads per workgroup = 4 wave64
constexpr int kWaves = kBlock / 64;
constexpr int kAcc = 29;             // 21 (lower triangle of H, column by column) + 6 (b) + accepted pairs + nodes visited
constexpr int kSolveThreads = 1024;

struct TreeRef {
  const madicp_node* nodes;
  int32_t n_nodes;
  int32_t n_leaves;
};

// One registration in flight; lives in device memory, written by the host before each launch sequence
// and advanced by icp_solve.  Keeping every per-registration quantity behind this one pointer is what lets
// a single captured hipGraph serve every scan / keyframe set of the same launch geometry.
struct Job {
  const double* moving;  // (L,4): x y z |p|  (sensor frame)
  uint8_t* matched;      // (L) matched_ flags of the last round
  uint32_t* corr;        // optional (K,L) correspondence trace
  double* x_iters;       // optional (n_iters,12) pose before each round
  int32_t L;
  int32_t K;
  int32_t n_iters;
  int32_t iter;
  int32_t flags;         // kFlagNoUpdate
  int32_t n_matched;
  unsigned long long visits;  // internal nodes visited (all rounds; exact: integer-valued doubles summed)
  double X[12];          // R row-major, t
  double min_ball, rho, b_ratio;
  double H[36];          // row-major, of the last round
  double b[6];
  double n_pairs;        // accepted (leaf,tree) pairs of the last round
  TreeRef trees[MADICP_MAX_TREES];
};
constexpr int kFlagNoUpdate = 1;

// ---------------------------------------------------------------------------------------------------
// fp64 helpers with the reference's evaluation order (see oracle/linalg.h for the derivation).
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ double dotc(double a0, double a1, double a2, double b0, double b1, double b2) {
#ifdef MADICP_REDUX_SCALAR_ONLY
  return a0 * b0 + (a1 * b1 + a2 * b2);
#else
  return (a0 * b0 + a1 * b1) + a2 * b2;  // contiguous 3-vector redux: packet of two, then the tail
#endif
}
__device__ __forceinline__ double dots(double a0, double a1, double a2, double b0, double b1, double b2) {
  return a0 * b0 + (a1 * b1 + a2 * b2);  // strided row of a small lazy product
}

struct NodeV {
  double m0, m1, m2, d0, d1, d2;
  int right, leaf_id;
  double bbox0;
};
// one 64 B record = four 16 B loads from one cache line
// (pointers that were themselves loaded from memory are "generic" to the compiler; the explicit global
// address space turns flat_load into global_load)
typedef double vd2 __attribute__((ext_vector_type(2)));
typedef double vd4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) vd2* gptr_d2;
typedef const __attribute__((address_space(1))) vd4* gptr_d4;
__device__ __forceinline__ NodeV load_node(const madicp_node* __restrict__ nodes, int idx) {
  gptr_d2 p = (gptr_d2)(uintptr_t)(nodes + idx);
  const vd2 a = p[0], b = p[1], c = p[2], d = p[3];
  NodeV n;
  n.m0 = a.x; n.m1 = a.y; n.m2 = b.x;
  n.d0 = b.y; n.d1 = c.x; n.d2 = c.y;
  const long long bits = __double_as_longlong(d.x);
  n.right = static_cast<int>(bits & 0xffffffffll);
  n.leaf_id = static_cast<int>(bits >> 32);
  n.bbox0 = d.y;
  return n;
}

// greedy root->leaf descent, no backtracking (mad_tree.cpp:144-152)
__device__ __forceinline__ NodeV descend(const madicp_node* __restrict__ nodes, double q0, double q1, double q2,
                                         int& node_idx, int& depth) {
  int idx = 0, d = 0;
  NodeV n = load_node(nodes, 0);
  while (n.right != 0) {
    const double side = dotc(q0 - n.m0, q1 - n.m1, q2 - n.m2, n.d0, n.d1, n.d2);
    idx = (side < 0.0) ? idx + 1 : idx + n.right;
    n = load_node(nodes, idx);
    ++d;
  }
  node_idx = idx;
  depth = d;
  return n;
}

// ---------------------------------------------------------------------------------------------------
__global__ void moving_prep(const double* __restrict__ xyz, double* __restrict__ out, int L) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= L) return;
  const double x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
  double4 o;
  o.x = x; o.y = y; o.z = z;
  o.w = sqrt(dotc(x, y, z, x, y, z));
  reinterpret_cast<double4*>(out)[i] = o;
}

__global__ void nn_descend(const madicp_node* __restrict__ nodes, const double* __restrict__ q, long long n,
                           uint32_t* __restrict__ out_leaf, uint32_t* __restrict__ out_node,
                           double* __restrict__ out_dist, int32_t* __restrict__ out_depth) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const double q0 = q[3 * i], q1 = q[3 * i + 1], q2 = q[3 * i + 2];
    int idx, depth;
    const NodeV leaf = descend(nodes, q0, q1, q2, idx, depth);
    if (out_leaf) out_leaf[i] = static_cast<uint32_t>(leaf.leaf_id);
    if (out_node) out_node[i] = static_cast<uint32_t>(idx);
    if (out_depth) out_depth[i] = depth;
    if (out_dist) {
      const double e0 = q0 - leaf.m0, e1 = q1 - leaf.m1, e2 = q2 - leaf.m2;
      out_dist[i] = sqrt(dotc(e0, e1, e2, e0, e1, e2));
    }
  }
}

// mean <- R mean + t ; dir <- R dir   (R row-major)
__global__ void tree_transform(madicp_node* __restrict__ nodes, int n, const double* __restrict__ Rt) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double R[9], t[3];
#pragma unroll
  for (int k = 0; k < 9; ++k) R[k] = Rt[k];
#pragma unroll
  for (int k = 0; k < 3; ++k) t[k] = Rt[9 + k];
  madicp_node nd = nodes[i];
  const double m0 = nd.mean[0], m1 = nd.mean[1], m2 = nd.mean[2];
  const double d0 = nd.dir[0], d1 = nd.dir[1], d2 = nd.dir[2];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    nd.mean[r] = dots(R[3 * r], R[3 * r + 1], R[3 * r + 2], m0, m1, m2) + t[r];
    nd.dir[r] = dots(R[3 * r], R[3 * r + 1], R[3 * r + 2], d0, d1, d2);
  }
  nodes[i] = nd;
}

// ---------------------------------------------------------------------------------------------------
// icp_linearize
//
// Work decomposition.  A *unit* is (tree k, chunk c): kBlock*QPT consecutive moving leaves against one
// keyframe tree.  Units are ordered tree-major and cut into 8 contiguous ranges, one per XCD; workgroup
// b runs on XCD b % 8 (observed dispatch rule — used for speed only, never for correctness), so every
// XCD's private 4 MiB L2 only serves the nodes of its own ~K/8 trees.  Inside an XCD the workgroups
// stride over the range.  Moving leaves arrive in the DFS order of the scan's own MAD-tree, i.e. spatially
// sorted, so the 64 lanes of a wave walk the same upper path (one cache line per level for the whole
// wave) and only diverge near the leaves.
//
// grid = (8 * slots, n_scans); blockIdx.y selects the registration (scans batched in flight).
// partials: [scan][gridDim.x][kAcc]
// ---------------------------------------------------------------------------------------------------
template <int QPT>
__global__ __launch_bounds__(kBlock) void icp_linearize(Job* __restrict__ jobs, double* __restrict__ partials) {
  Job* job = jobs + blockIdx.y;
  const int L = job->L;
  const int K = job->K;
  const bool last_round = (job->iter == job->n_iters - 1);
  const double* __restrict__ moving = job->moving;
  uint8_t* __restrict__ matched = job->matched;
  uint32_t* __restrict__ corr = job->corr;

  double R[9], t[3];
#pragma unroll
  for (int k = 0; k < 9; ++k) R[k] = job->X[k];
#pragma unroll
  for (int k = 0; k < 3; ++k) t[k] = job->X[9 + k];
  const double min_ball = job->min_ball, rho = job->rho, b_ratio = job->b_ratio;

  double acc[kAcc];
#pragma unroll
  for (int v = 0; v < kAcc; ++v) acc[v] = 0.0;
  unsigned int visits = 0;

  constexpr int kChunk = kBlock * QPT;
  const int C = (L + kChunk - 1) / kChunk;
  const long long U = (long long)K * C;
  const int xcd = blockIdx.x & 7;
  const int slot = blockIdx.x >> 3;
  const int nslots = gridDim.x >> 3;
  const long long lo = (xcd * U) >> 3;
  const long long hi = ((xcd + 1) * U) >> 3;

  for (long long u = lo + slot; u < hi; u += nslots) {
    const int k = static_cast<int>(u / C);
    const int c = static_cast<int>(u - (long long)k * C);
    const madicp_node* __restrict__ nodes = job->trees[k].nodes;

    // QPT independent descents per lane, advanced together so their loads overlap
    double px[QPT], py[QPT], pz[QPT], pn[QPT], q0[QPT], q1[QPT], q2[QPT];
    int idx[QPT];
    bool live[QPT], valid[QPT];
    NodeV nd[QPT];
#pragma unroll
    for (int j = 0; j < QPT; ++j) {
      const int i = c * kChunk + j * kBlock + threadIdx.x;
      valid[j] = i < L;
      vd4 p = {0.0, 0.0, 0.0, 0.0};
      if (valid[j]) p = ((gptr_d4)(uintptr_t)moving)[i];
      px[j] = p.x; py[j] = p.y; pz[j] = p.z; pn[j] = p.w;
      // ml = X * p  (Isometry3d * Vector3d: linear()*p + translation(), mad_icp.cpp:78)
      q0[j] = t[0] + dots(R[0], R[1], R[2], p.x, p.y, p.z);
      q1[j] = t[1] + dots(R[3], R[4], R[5], p.x, p.y, p.z);
      q2[j] = t[2] + dots(R[6], R[7], R[8], p.x, p.y, p.z);
      idx[j] = 0;
    }
#pragma unroll
    for (int j = 0; j < QPT; ++j) {
      nd[j] = load_node(nodes, 0);
      live[j] = valid[j] && nd[j].right != 0;
    }
    bool any = false;
#pragma unroll
    for (int j = 0; j < QPT; ++j) any |= live[j];
    while (any) {
      any = false;
#pragma unroll
      for (int j = 0; j < QPT; ++j) {
        if (live[j]) {
          const double side = dotc(q0[j] - nd[j].m0, q1[j] - nd[j].m1, q2[j] - nd[j].m2, nd[j].d0, nd[j].d1, nd[j].d2);
          idx[j] = (side < 0.0) ? idx[j] + 1 : idx[j] + nd[j].right;
          ++visits;
        }
      }
#pragma unroll
      for (int j = 0; j < QPT; ++j) {
        if (live[j]) {
          nd[j] = load_node(nodes, idx[j]);
          live[j] = nd[j].right != 0;
          any |= live[j];
        }
      }
    }

#pragma unroll
    for (int j = 0; j < QPT; ++j) {
      if (!valid[j]) continue;
      const int i = c * kChunk + j * kBlock + threadIdx.x;
      const NodeV& f = nd[j];
      // gate (mad_icp.cpp:81-83)
      const double src_ball = min_ball + b_ratio * pn[j];
      const double g0 = q0[j] - f.m0, g1 = q1[j] - f.m1, g2 = q2[j] - f.m2;
      const bool rejected = sqrt(dotc(g0, g1, g2, g0, g1, g2)) > src_ball;
      if (corr) corr[(long long)k * L + i] = static_cast<uint32_t>(f.leaf_id) | (rejected ? 0x80000000u : 0u);
      if (rejected) continue;
      if (last_round) matched[i] = 1;  // idempotent byte store (mad_icp.cpp:85)

      // errorAndJacobian (mad_icp.cpp:59-72)
      const double e = dotc(g0, g1, g2, f.d0, f.d1, f.d2);
      double J[6];
      J[0] = dotc(f.d0, f.d1, f.d2, R[0], R[3], R[6]);
      J[1] = dotc(f.d0, f.d1, f.d2, R[1], R[4], R[7]);
      J[2] = dotc(f.d0, f.d1, f.d2, R[2], R[5], R[8]);
      // -J[0:3] * skew(p): columns of skew(p) are (0,pz,-py), (-pz,0,px), (py,-px,0)
      const double a0 = -J[0], a1 = -J[1], a2 = -J[2];
      J[3] = dotc(a0, a1, a2, 0.0, pz[j], -py[j]);
      J[4] = dotc(a0, a1, a2, -pz[j], 0.0, px[j]);
      J[5] = dotc(a0, a1, a2, py[j], -px[j], 0.0);

      // Huber x planarity weight (mad_icp.cpp:92-98; `abs` there is fabs — SURVEY fact 4)
      double scale = 1.0;
      const double chi = fabs(e);
      if (chi > rho) scale = rho / chi;
      const double w = 1.0 - f.bbox0 / min_ball;
      scale *= w * w;

      double sJ[6];
#pragma unroll
      for (int r = 0; r < 6; ++r) sJ[r] = scale * J[r];
      int v = 0;
#pragma unroll
      for (int cc = 0; cc < 6; ++cc)
#pragma unroll
        for (int r = cc; r < 6; ++r) acc[v++] += sJ[r] * J[cc];
#pragma unroll
      for (int r = 0; r < 6; ++r) acc[21 + r] += sJ[r] * e;
      acc[27] += 1.0;
    }
  }

  // deterministic reduction: lanes (xor-free shuffle-down tree) -> waves (LDS, fixed order) -> partial
  acc[28] = static_cast<double>(visits);  // integer-valued: its sums are exact in any order
  __shared__ double red[kWaves][kAcc];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
#pragma unroll
  for (int v = 0; v < kAcc; ++v) {
    double x = acc[v];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
    if (lane == 0) red[wave][v] = x;
  }
  __syncthreads();
  if (threadIdx.x < kAcc) {
    double s = red[0][threadIdx.x];
#pragma unroll
    for (int w = 1; w < kWaves; ++w) s += red[w][threadIdx.x];
    partials[((long long)blockIdx.y * gridDim.x + blockIdx.x) * kAcc + threadIdx.x] = s;
  }
}

// ---------------------------------------------------------------------------------------------------
// 6x6 LDLT (lower, diagonal pivoting) factor + solve — the algorithm of Eigen::LDLT that
// `H_adder_.ldlt().solve(-b_adder_)` runs (mad_icp.cpp:111).  A: row-major, only the lower triangle is
// read.  One lane; dynamic indexing goes through private scratch, which is irrelevant at this size.
// ---------------------------------------------------------------------------------------------------
__device__ inline void ldlt6_solve(const double* A, const double* rhs, double* x) {
  double m[6][6];
  int tr[6];
  double tmp[6];
  for (int r = 0; r < 6; ++r)
    for (int c = 0; c < 6; ++c) m[r][c] = A[r * 6 + c];
  for (int k = 0; k < 6; ++k) {
    int big = k;
    double best = fabs(m[k][k]);
    for (int i = k + 1; i < 6; ++i) {
      const double a = fabs(m[i][i]);
      if (a > best) { best = a; big = i; }
    }
    tr[k] = big;
    if (k != big) {
      for (int j = 0; j < k; ++j) { const double s = m[k][j]; m[k][j] = m[big][j]; m[big][j] = s; }
      for (int i = big + 1; i < 6; ++i) { const double s = m[i][k]; m[i][k] = m[i][big]; m[i][big] = s; }
      { const double s = m[k][k]; m[k][k] = m[big][big]; m[big][big] = s; }
      for (int i = k + 1; i < big; ++i) { const double s = m[i][k]; m[i][k] = m[big][i]; m[big][i] = s; }
    }
    if (k > 0) {
      for (int j = 0; j < k; ++j) tmp[j] = m[j][j] * m[k][j];
      double a = m[k][0] * tmp[0];
      for (int j = 1; j < k; ++j) a += m[k][j] * tmp[j];
      m[k][k] -= a;
      for (int i = k + 1; i < 6; ++i) {
        double s = m[i][0] * tmp[0];
        for (int j = 1; j < k; ++j) s += m[i][j] * tmp[j];
        m[i][k] -= s;
      }
    }
    const double akk = m[k][k];
    const bool ok = fabs(akk) > 0.0;
    if (k == 0 && !ok) {
      for (int j = 0; j < 6; ++j) tr[j] = j;
      break;
    }
    if (ok)
      for (int i = k + 1; i < 6; ++i) m[i][k] /= akk;
  }
  for (int i = 0; i < 6; ++i) x[i] = rhs[i];
  for (int k = 0; k < 6; ++k)
    if (tr[k] != k) { const double s = x[k]; x[k] = x[tr[k]]; x[tr[k]] = s; }
  for (int i = 1; i < 6; ++i) {
    double a = m[i][0] * x[0];
    for (int j = 1; j < i; ++j) a += m[i][j] * x[j];
    x[i] -= a;
  }
  const double tol = 2.2250738585072014e-308;  // numeric_limits<double>::min()
  for (int i = 0; i < 6; ++i) x[i] = (fabs(m[i][i]) > tol) ? x[i] / m[i][i] : 0.0;
  for (int i = 4; i >= 0; --i) {
    double a = m[i + 1][i] * x[i + 1];
    for (int j = i + 2; j < 6; ++j) a += m[j][i] * x[j];
    x[i] -= a;
  }
  for (int k = 5; k >= 0; --k)
    if (tr[k] != k) { const double s = x[k]; x[k] = x[tr[k]]; x[tr[k]] = s; }
}

// lie_algebra.h:39-52, R row-major
__device__ inline void exp_so3(const double* w, double* R) {
  const double th2 = dotc(w[0], w[1], w[2], w[0], w[1], w[2]);
  const double th = sqrt(th2);
  const double W[9] = {0.0, -w[2], w[1], w[2], 0.0, -w[0], -w[1], w[0], 0.0};
  if (th2 < 1e-8) {
    for (int i = 0; i < 9; ++i) R[i] = ((i % 4 == 0) ? 1.0 : 0.0) + W[i];
    return;
  }
  double Kx[9], cK[9];
  const double omc = 2.0 * sin(th / 2.0) * sin(th / 2.0);
  const double s = sin(th);
  for (int i = 0; i < 9; ++i) { Kx[i] = W[i] / th; cK[i] = omc * Kx[i]; }
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) {
      const double kk = dots(cK[3 * r], cK[3 * r + 1], cK[3 * r + 2], Kx[c], Kx[3 + c], Kx[6 + c]);
      R[3 * r + c] = (((r == c) ? 1.0 : 0.0) + s * Kx[3 * r + c]) + kk;
    }
}

// total: kAcc sums of this round (already joined over workgroups / ranks).  One lane.
__device__ inline void gn_update(Job* job, const double* total) {
  double H[36], b[6];
  int v = 0;
  for (int c = 0; c < 6; ++c)
    for (int r = c; r < 6; ++r) {
      H[r * 6 + c] = total[v];
      H[c * 6 + r] = total[v];  // mirror: see DESIGN.md "H symmetry"
      ++v;
    }
  for (int r = 0; r < 6; ++r) b[r] = total[21 + r];
  for (int i = 0; i < 36; ++i) job->H[i] = H[i];
  for (int i = 0; i < 6; ++i) job->b[i] = b[i];
  job->n_pairs = total[27];
  job->visits += static_cast<unsigned long long>(total[28]);
  const int it = job->iter;
  if (job->x_iters)
    for (int i = 0; i < 12; ++i) job->x_iters[(long long)it * 12 + i] = job->X[i];
  if (!(job->flags & kFlagNoUpdate)) {
    double nb[6], dx[6], dR[9], Rn[9], tn[3];
    for (int r = 0; r < 6; ++r) nb[r] = -b[r];
    ldlt6_solve(H, nb, dx);
    exp_so3(dx + 3, dR);
    const double* R = job->X;
    const double* t = job->X + 9;
    for (int r = 0; r < 3; ++r) {
      for (int c = 0; c < 3; ++c)
        Rn[3 * r + c] = dots(R[3 * r], R[3 * r + 1], R[3 * r + 2], dR[c], dR[3 + c], dR[6 + c]);
      tn[r] = dots(R[3 * r], R[3 * r + 1], R[3 * r + 2], dx[0], dx[1], dx[2]) + t[r];
    }
    for (int i = 0; i < 9; ++i) job->X[i] = Rn[i];
    for (int i = 0; i < 3; ++i) job->X[9 + i] = tn[i];
  }
  job->iter = it + 1;
}

// join of the per-workgroup partials in a fixed order: 32 segments summed in parallel, then in sequence
__device__ inline void join_partials(const double* __restrict__ partials, int nblocks, double* total /*LDS kAcc*/) {
  __shared__ double seg[32][kAcc];
  const int j = threadIdx.x & 31;
  const int s = threadIdx.x >> 5;
  const int seg_len = (nblocks + 31) / 32;
  if (j < kAcc) {
    double a = 0.0;
    const int b0 = s * seg_len;
    const int b1 = min(nblocks, b0 + seg_len);
    for (int b = b0; b < b1; ++b) a += partials[(long long)b * kAcc + j];
    seg[s][j] = a;
  }
  __syncthreads();
  if (threadIdx.x < kAcc) {
    double a = seg[0][threadIdx.x];
    for (int k = 1; k < 32; ++k) a += seg[k][threadIdx.x];
    total[threadIdx.x] = a;
  }
  __syncthreads();
}

// before the last round the matched_ flags are cleared (pipeline.cpp:172-176)
__device__ inline void clear_matched_if_next_is_last(Job* job) {
  if (job->iter + 1 == job->n_iters - 1)
    for (int i = threadIdx.x; i < job->L; i += blockDim.x) job->matched[i] = 0;
}

// single-GPU: join + solve + update in one launch; grid = n_scans, block = kSolveThreads
__global__ __launch_bounds__(kSolveThreads) void icp_solve(Job* __restrict__ jobs, const double* __restrict__ partials,
                                                          int nblocks) {
  __shared__ double total[kAcc];
  Job* job = jobs + blockIdx.x;
  join_partials(partials + (long long)blockIdx.x * nblocks * kAcc, nblocks, total);
  clear_matched_if_next_is_last(job);
  if (threadIdx.x == 0) gn_update(job, total);
}

// multi-GPU: join -> totals[scan][kAcc] | ncclAllReduce(sum) | update
__global__ __launch_bounds__(kSolveThreads) void icp_reduce(Job* __restrict__ jobs, const double* __restrict__ partials,
                                                           int nblocks, double* __restrict__ totals) {
  __shared__ double total[kAcc];
  Job* job = jobs + blockIdx.x;
  join_partials(partials + (long long)blockIdx.x * nblocks * kAcc, nblocks, total);
  clear_matched_if_next_is_last(job);
  if (threadIdx.x < kAcc) totals[blockIdx.x * kAcc + threadIdx.x] = total[threadIdx.x];
}
__global__ void icp_update(Job* __restrict__ jobs, const double* __restrict__ totals, int n_scans) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s < n_scans) gn_update(jobs + s, totals + s * kAcc);
}

// matched-leaf count (pipeline.cpp:197-204); grid = n_scans
__global__ __launch_bounds__(kBlock) void icp_finish(Job* __restrict__ jobs) {
  Job* job = jobs + blockIdx.x;
  __shared__ int cnt[kWaves];
  int c = 0;
  for (int i = threadIdx.x; i < job->L; i += blockDim.x) c += job->matched[i] ? 1 : 0;
  for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
  if ((threadIdx.x & 63) == 0) cnt[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    int s = 0;
    for (int w = 0; w < kWaves; ++w) s += cnt[w];
    job->n_matched = s;
  }
}

}  // namespace madicp
