// MAD-tree construction on the device.  Replaces MADtree::build / makeSubtree / getLeafs
// (reference mad_tree.cpp:47-142,154-163 with utils.h:37-97) for callers that opt in: the scan goes to HBM once, the
// linear 64-byte node array is produced there, and nothing of the tree ever crosses PCIe.
//
// Same decisions as the reference, node by node — mean / sample covariance of the node's points (utils.h:54-73), the
// closed-form eigen-decomposition (common/eig3.h, the one source the host builder uses), extents in the eigen frame with 0
// included (utils.h:75-97), leaf rule bbox(2) < b_max, normal inheritance from the top-most flat ancestor or the nearest
// ancestor with >= 3 points (mad_tree.cpp:64-74), leaf mean snapped to the member nearest the centroid, FIRST member on
// ties (:76-86), split at the centroid along the largest eigenvector (:95-97) — and the same MEMBER ORDER: every node's
// points end up in the permutation the reference's in-place `split` (utils.h:37-52) leaves, in all three regimes
// (common/split_order.h: the closed form per point, from the side flags and two rank tables — pinned against the
// reference's loop on the CPU, tests/test_split_order.py).  The order decides which member represents a leaf whenever two
// members tie in distance to the centroid (every two-point leaf does), and with it the device-built tree has the host
// builder's leaf representatives except where a decision sits within rounding of its threshold.  What is NOT the same
// bits: the reference adds a node's points one after the other in that order, a serial chain that no parallel machine
// can follow, and the eigen-solver's atan2 / cos / sin come from a different math library — centroids, split normals and
// extents differ in their last bits.  Every sum here has a FIXED shape (lane-strided partial sums, xor butterfly, chunk
// partials in chunk order), so a build is bit-reproducible run to run and independent of scheduling
// (tests/test_gpu_frontend.py states the measured rates and the pose bound).
//
// What bounds it.  The arithmetic is nothing (~150 M lane-instructions for a 120 k-point scan: microseconds of the
// chip); a node needs sums -> eigen-solve (~1300 dependent fp64 instructions, ~5 us on one wave) -> extents -> partition in
// sequence, and a child cannot start before its parent has split: the build is a chain of depth x (that sequence), ~17
// levels deep.  So the design is level-synchronous — every node of a level at once, one launch sequence per level, the
// points ping-ponging between two buffers so that a stable partition is an out-of-place scatter inside the node's own
// range — and the only question per node is how many lanes share its chain:
//   chip  (n > kChipMin = 512, first levels): the node's points are cut into 2048-point chunks, one workgroup per chunk; two
//          kernels per level (statistics + extents + left counts + the chunk's rank tables | scatter), three for the root
//          (its sums first); every workgroup recombines the per-chunk partials of its node in chunk order (loaded in
//          parallel, added in order), so nothing waits for a single combiner and the result does not depend on scheduling;
//   team  (n > kTeamMin = 512 past the chip levels): ONE workgroup of four wavefronts per node — every wavefront a
//          contiguous quarter of the node, the quarters playing the part of the chip regime's chunks, barriers where those
//          need a kernel boundary (round 4: one wavefront per such node made steps 6-8 of a scan 72 + 46 + 33 us);
//   wave  (32 < n <= kTeamMin): one wavefront per node, no barrier — strided sums, xor butterfly, wave-uniform
//          eigen-solve, ONE batch of eight points per lane that stays in registers from the sweep (extents, sides, rank
//          tables in the wavefront's LDS scratch) to the scatter;
//   quad  (n <= 32): FOUR lanes per node, 16 nodes per wavefront (most nodes of a MAD-tree hold a handful of points; a
//          wave-uniform eigen-solve per such node would spend 64 lanes on one, a single lane per node makes the sweep a
//          long serial loop): the wave regime in miniature — quad ballots, two-step xor reductions, the side flags of a
//          node in one 32-bit word and the rank tables as bit selects.
// (A launch that finishes whole sub-trees in LDS instead of a kernel per level was built and measured slower:
// profiles/r4_e_subtree_negative.md.)
// Nodes are created in scheduling order into a temporary array; ids and queue slots are handed out by ONE atomic per
// workgroup (wave regime) or per wavefront (quad regime) on counters that each own a 128-byte line — with one atomic
// per node on shared lines the allocation alone cost 60 us per level (1 600 nodes x 3 atomics x ~12 ns).  The final
// DFS-preorder position needs no bottom-up pass: the leaves partition the (permuted) point array, so with S[i] = number
// of leaves that start before point i (one exclusive scan of the leaf-start marks), a node owning points [b, e) that
// was reached by `t` left turns sits at preorder index 2 S[b] + t, its right child 2 (S[mid] - S[b]) further, and a
// leaf's getLeafs() ordinal is S[b].
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../common/eig3.h"
#include "../common/split_order.h"
#include "madicp_hip.h"

#pragma clang fp contract(off)

namespace madicp {
namespace tb {

#ifndef MADICP_TB_SMALL
#define MADICP_TB_SMALL 32
#endif
#ifndef MADICP_TB_SERIAL_SMALL
#define MADICP_TB_SERIAL_SMALL 1
#endif
constexpr bool kSerialSmall = MADICP_TB_SERIAL_SMALL != 0;  // quad-regime nodes add their points up in member order (see quad_node)
constexpr int kSmallMax = MADICP_TB_SMALL;  // quad regime: a node with at most this many points is handled by four lanes
static_assert(kSmallMax <= 32, "the quad regime keeps a node's side flags in one 32-bit word");
#ifndef MADICP_TB_CHIP_MIN
#define MADICP_TB_CHIP_MIN 512
#endif
#ifndef MADICP_TB_CHIP_LEVELS
#define MADICP_TB_CHIP_LEVELS 6
#endif
constexpr int kChipMin = MADICP_TB_CHIP_MIN;        // chip regime above this many points ...
constexpr int kChipLevels = MADICP_TB_CHIP_LEVELS;  // ... during the first levels only (afterwards the wave regime takes any size)
constexpr int kTeamMin = 512;  // team regime: past the chip levels a node of more points gets a whole workgroup; a wave-regime
                               // node is at most this: one batch of eight points per lane (64 * kWU)
constexpr int kChunk = 2048;     // points per workgroup in the chip regime (256 threads x 8)
constexpr int kMaxBig = 256;     // chip-regime nodes per level (more go to the wave regime)
constexpr int kMaxLevels = 96;   // a deeper tree is reported as an error, never walked into

enum : int { kLeaf = 1, kHasPlane = 2, kHasSmall = 4, kDone = 8, kLeafPending = 16, kHasSums = 32, kChunkSums = 64, kSumRight = 128 };

struct BNode {          // a node while the tree is being built
  double mean[3];       // internal: centroid; leaf: the member nearest to it
  double dir[3];        // internal: split normal (eigenvector 2); leaf: surface normal
  double col0[3];       // internal: own normal (eigenvector 0) — children may inherit it
  double plane_n[3];    // inherited: normal of the top-most flat ancestor (kHasPlane)
  double small_n[3];    // inherited: normal of the nearest ancestor with >= 3 points (kHasSmall)
  double bbox0;
  double sums[9];       // kHasSums: the nine sums of this node's points, accumulated by the parent's scatter sweep
  int32_t begin, end, mid, left_turns;
  int32_t flags, level;
  int32_t sum_first, sum_n;  // kChunkSums: the node's sums are the per-chunk partials [sum_first, sum_first + sum_n) that its
                             // chip-regime parent's scatter left behind (left or right half: kSumRight), added in order
  int32_t child, pad_;       // internal: temporary id of the left child (the right one is child + 1) — what the breadth-first
                             // layout of the top levels walks (top_bfs_body)
};
static_assert(sizeof(BNode) == 240, "BNode layout");

struct Counter {  // one 128-byte line per counter: atomics on different counters do not queue behind each other
  int32_t v;
  int32_t pad_[31];
};
struct HeadLine {
  int32_t v;      // temporary nodes allocated
  int32_t error;  // 1: node capacity, 2: depth
  int32_t pad_[30];
};
struct State {  // counters and results of one build, device resident
  HeadLine n_nodes;
  // one line of results, cleared as a whole before every summary pass
  int32_t n_leaves;     // after the scan
  int32_t n_top;        // internal nodes above level kTopLevels (the LDS-staged top of icp_round)
  int32_t max_level;
  int32_t n_valid;      // temporary nodes that were finished (== n_nodes.v unless something went wrong)
  unsigned long long rho_bits;  // max |mean - origin|_2 over the internal nodes, as the bits of a non-negative double
  double origin[3];             // the root's mean
  int32_t pad_[20];
  Counter q_count[kMaxLevels + 2];      // wave-regime nodes queued per level
  Counter team_count[kMaxLevels + 2];   // team-regime nodes queued per level
  Counter small_count[kMaxLevels + 2];  // quad-regime nodes queued per level
  Counter big_count[kMaxLevels + 2];    // chip-regime nodes queued per level
};

struct Params {
  const double* cloud;  // level 0 reads the caller's cloud
  double* buf[2];       // level l > 0 reads buf[l & 1]; level l writes buf[(l + 1) & 1]
  BNode* nodes;
  int32_t node_cap;
  State* st;
  int4* q[2];           // wave-regime queues, by level parity; an entry is {node id, begin, end, level}: one hop
                        // from the queue to everything the sweep needs
  int32_t* big[2];      // chip-regime lists, by level parity
  int4* small[2];       // quad-regime queues, by level parity (same entries)
  int4* team[2];        // team-regime queues, by level parity (same entries)
  uint32_t* leaf_start; // (n_points + 1): 1 where a leaf's point range starts
  uint32_t* S;          // (n_points + 1): its exclusive scan — leaves in front of a point (made when the levels are done)
  uint32_t* tile_sums;  // scan scratch: marks per 1024-point tile
  double* partLR;       // chip regime, one block of part_stride doubles per level 0 .. kChipLevels: per chunk slot 18 doubles —
                        // the nine sums of the chunk's points that go left, then of those that go right (written by the
                        // PARENT level's scatter; level 0: by tb_init into the first nine).  Per level, not by parity:
                        // a small child of a chip node reads its block when its step comes, levels later.
  long part_stride;
  double* part2;        // chip regime: per chunk slot 8 doubles (lo 3, hi 3, left count)
  int32_t* tab;         // (n_points) rank tables of the split's permutation (common/split_order.h), one slice per node — its
                        // own point range [b, e) — cut like the node: per 2048-point chunk (chip regime: written by
                        // tb_chip_stats, read by tb_chip_scatter) or per wavefront's quarter (team regime): the positions
                        // (relative to b) of the slice's points that go left, in order, from the front of the slice, of those
                        // that go right from its back.  (The wave regime keeps its tables in LDS, the quad regime in a word.)
  int32_t* top_ids;     // (kTopMax) breadth-first layout of the first kTopLevels levels (top_bfs_body, inside three level launches):
  uint32_t* top_link;   // temporary ids of the internal nodes in that order, and their link words
  int32_t* top_front;   // (2 + 2 * 1024) the layout between two of its parts: entries and first position of the next level, then
                        // the entries' ids and their left children's ids
  int32_t n_points;
  double b_max, b_min;
  int32_t first_step;   // wave / quad nodes created above this level wait in its queues: while the chip regime runs (levels
                        // 0 .. kChipLevels - 1 of a big cloud) such nodes are rare, and a launch per level that looks at
                        // two empty queues cost 3 us each.  A queue index is therefore a STEP (>= the node's level); the
                        // level proper travels in the entry (it selects the point buffer) and in the node.
};

// (selects, not P.buf[level & 1]: a dynamically indexed member sends the whole by-value Params to scratch memory)
__device__ __forceinline__ const double* level_in(const Params& P, int level) {
  return level == 0 ? P.cloud : ((level & 1) ? P.buf[1] : P.buf[0]);
}
__device__ __forceinline__ double* level_out(const Params& P, int level) { return (level & 1) ? P.buf[0] : P.buf[1]; }
__device__ __forceinline__ int4* level_q(const Params& P, int level) { return (level & 1) ? P.q[1] : P.q[0]; }
__device__ __forceinline__ int32_t* level_big(const Params& P, int level) { return (level & 1) ? P.big[1] : P.big[0]; }
__device__ __forceinline__ int4* level_small(const Params& P, int level) { return (level & 1) ? P.small[1] : P.small[0]; }
__device__ __forceinline__ int4* level_team(const Params& P, int level) { return (level & 1) ? P.team[1] : P.team[0]; }
__device__ __forceinline__ double* level_part(const Params& P, int level) { return P.partLR + (long)min(level, kChipLevels) * P.part_stride; }

// ---- per-node arithmetic, shared by the three regimes ------------------------------------------------------
// utils.h:54-73 after the sums: s = {sum x, sum y, sum z, sum xx, xy, xz, yy, yz, zz}
__device__ __forceinline__ void mean_cov_from_sums(const double* s, int k, double* mean, double* cov) {
  const double inv_k = 1. / k;
  #pragma unroll
  for (int i = 0; i < 3; ++i) mean[i] = s[i] * inv_k;
  const double S9[9] = {s[3], s[4], s[5], s[4], s[6], s[7], s[5], s[7], s[8]};
  const double bessel = double(k) / double(k - 1);
  #pragma unroll
  for (int r = 0; r < 3; ++r)
    #pragma unroll
    for (int q = 0; q < 3; ++q) {
      double x = S9[3 * r + q] * inv_k;
      x -= mean[r] * mean[q];
      cov[3 * r + q] = x * bessel;
    }
}
__device__ __forceinline__ void add_point(double* s, double x, double y, double z) {
  s[0] += x; s[1] += y; s[2] += z;
  s[3] += x * x; s[4] += x * y; s[5] += x * z;
  s[6] += y * y; s[7] += y * z;
  s[8] += z * z;
}
// coordinates of p - mean in the eigen frame (utils.h:85): v_a = col_a . d, contiguous-reduction order
__device__ __forceinline__ void eigen_coords(const double* V, const double* mean, double x, double y, double z, double* v) {
  const double d0 = x - mean[0], d1 = y - mean[1], d2 = z - mean[2];
  v[0] = madicp_host::sum3c(V[0] * d0, V[3] * d1, V[6] * d2);
  v[1] = madicp_host::sum3c(V[1] * d0, V[4] * d1, V[7] * d2);
  v[2] = madicp_host::sum3c(V[2] * d0, V[5] * d1, V[8] * d2);
}
// the split test (mad_tree.cpp:96): (p - mean) . col2 < 0  — the same products as v[2] above, commuted
__device__ __forceinline__ bool goes_left(const double* mean, const double* col2, double x, double y, double z) {
  const double d0 = x - mean[0], d1 = y - mean[1], d2 = z - mean[2];
  return madicp_host::sum3c(d0 * col2[0], d1 * col2[1], d2 * col2[2]) < 0.0;
}
__device__ __forceinline__ void minmax_update(double* lo, double* hi, const double* v) {
  #pragma unroll
  for (int a = 0; a < 3; ++a) {  // `if`, not fmin/fmax: a NaN coordinate leaves the running value alone (utils.h:87-88)
    if (v[a] < lo[a]) lo[a] = v[a];
    if (hi[a] < v[a]) hi[a] = v[a];
  }
}

// what the children of an internal node inherit (mad_tree.cpp:64-74,90-93)
// what a node read of itself when it started (one batch of independent loads) and hands to its children
struct Inherit {
  int flags, left_turns, level, sum_first, sum_n;
  double plane_n[3], small_n[3];
};
__device__ __forceinline__ Inherit load_inherit(const BNode& nd, int level) {
  Inherit h;
  h.flags = nd.flags;
  h.left_turns = nd.left_turns;
  h.level = level;
  h.sum_first = nd.sum_first;
  h.sum_n = nd.sum_n;
#pragma unroll
  for (int i = 0; i < 3; ++i) { h.plane_n[i] = nd.plane_n[i]; h.small_n[i] = nd.small_n[i]; }
  return h;
}
__device__ __forceinline__ void make_child(BNode& c, const Inherit& p, int parent_id, const double* col0, double ext0, int n_parent,
                                           double b_min, int begin, int end, bool is_left) {
  int fl = 0;
  if (p.flags & kHasPlane) {
    fl |= kHasPlane;
#pragma unroll
    for (int i = 0; i < 3; ++i) c.plane_n[i] = p.plane_n[i];
  } else if (ext0 < b_min) {
    fl |= kHasPlane;
#pragma unroll
    for (int i = 0; i < 3; ++i) c.plane_n[i] = col0[i];
  } else {
#pragma unroll
    for (int i = 0; i < 3; ++i) c.plane_n[i] = 0.0;
  }
  if (n_parent >= 3 || !(p.flags & kHasSmall)) {
#pragma unroll
    for (int i = 0; i < 3; ++i) c.small_n[i] = col0[i];
  } else {
#pragma unroll
    for (int i = 0; i < 3; ++i) c.small_n[i] = p.small_n[i];
  }
  fl |= kHasSmall;
  c.flags = fl;
  c.begin = begin;
  c.end = end;
  c.mid = 0;
  c.left_turns = p.left_turns + (is_left ? 1 : 0);
  c.level = p.level + 1;
  c.sum_first = 0;
  c.sum_n = 0;
  (void)parent_id;
  c.bbox0 = 0.0;
}

// the sums of a node whose chip-regime parent left them as per-chunk partials: added in chunk order (every lane the same
// loads, the same order: identical totals everywhere)
__device__ __forceinline__ void chunk_sums(const Params& P, const Inherit& h, double* s) {
  const double* part = level_part(P, h.level) + ((h.flags & kSumRight) ? 9 : 0);
#pragma unroll
  for (int k = 0; k < 9; ++k) s[k] = 0.0;
  for (int c = 0; c < h.sum_n; ++c) {
    const double* q = part + (long)(h.sum_first + c) * 18;
#pragma unroll
    for (int k = 0; k < 9; ++k) s[k] += q[k];
  }
}

// which queue a node of n points at `level` belongs to: 0 lane, 1 wave, 2 chip
__device__ __forceinline__ int regime_of(int n, int level) {
  if (n <= kSmallMax) return 0;
  if (n > kChipMin && level < kChipLevels) return 2;
  return 1;
}
// queue a node with one atomic of its own (root, children of chip-regime nodes: a handful per level)
__device__ __forceinline__ void enqueue_single(const Params& P, int id, int begin, int end, int level) {
  const int n = end - begin;
  State* st = P.st;
  if (level > kMaxLevels) {
    st->n_nodes.error = 2;
    return;
  }
  int kind = regime_of(n, level);
  if (kind == 2) {
    const int pos = atomicAdd(&st->big_count[level].v, 1);
    if (pos < kMaxBig) {
      level_big(P, level)[pos] = id;
      return;
    }
    kind = 1;
  }
  const int step = max(level, P.first_step);
  if (kind == 0)
    level_small(P, step)[atomicAdd(&st->small_count[step].v, 1)] = make_int4(id, begin, end, level);
  else if (n > kTeamMin)
    level_team(P, step)[atomicAdd(&st->team_count[step].v, 1)] = make_int4(id, begin, end, level);
  else
    level_q(P, step)[atomicAdd(&st->q_count[step].v, 1)] = make_int4(id, begin, end, level);
}

// the surface normal of a leaf (mad_tree.cpp:64-74)
__device__ __forceinline__ void leaf_normal(const Inherit& nd, int n, const double* V, double* out) {
  if (nd.flags & kHasPlane) {
    #pragma unroll
    for (int i = 0; i < 3; ++i) out[i] = nd.plane_n[i];
  } else if (n < 3 && (nd.flags & kHasSmall)) {
    #pragma unroll
    for (int i = 0; i < 3; ++i) out[i] = nd.small_n[i];
  } else {
    out[0] = V[0]; out[1] = V[3]; out[2] = V[6];
  }
}

// ---- wave helpers ---------------------------------------------------------------------------------------------
// The value of lane (lane ^ M) — the partner of an xor butterfly — WITHOUT the LDS crossbar: __shfl_xor of a double is two
// ds_bpermute_b32 plus an address, ~100 cycles of latency each, and a node's reductions are 15-24 such butterflies (in-kernel
// stamps: 3.3 us of tb_chip_scatter for its 18 sums, 2 us of every sweep for its 6 extents).  gfx950 has a VALU route for
// every distance: v_permlane32_swap / v_permlane16_swap for 32 / 16 (a copy of the value swapped against itself leaves the
// lower and the upper partner in the two registers), DPP row_ror:8 for 8, row_shl:4 / row_shr:4 under complementary bank masks
// for 4, quad_perm for 2 and 1.  Same partners, same order, same operands as the __shfl_xor loops they replace: the same bits.
// (MADICP_TB_BPERMUTE=1 restores those, for the A/B.)
#ifndef MADICP_TB_BPERMUTE
#define MADICP_TB_BPERMUTE 0
#endif
template <int M>
__device__ __forceinline__ double xor_fetch(double v) {
  if (MADICP_TB_BPERMUTE) return __shfl_xor(v, M, 64);
  const int lo = __double2loint(v), hi = __double2hiint(v);
  if (M == 32 || M == 16) {
    // after swapping a register against a copy of itself: [0] holds the lower partner's value on both sides, [1] the upper's
    const auto l2 = M == 32 ? __builtin_amdgcn_permlane32_swap((unsigned)lo, (unsigned)lo, false, false)
                            : __builtin_amdgcn_permlane16_swap((unsigned)lo, (unsigned)lo, false, false);
    const auto h2 = M == 32 ? __builtin_amdgcn_permlane32_swap((unsigned)hi, (unsigned)hi, false, false)
                            : __builtin_amdgcn_permlane16_swap((unsigned)hi, (unsigned)hi, false, false);
    const bool upper = (threadIdx.x & M) != 0;  // (an upper lane's partner is the lower one)
    return __hiloint2double((int)(upper ? h2[0] : h2[1]), (int)(upper ? l2[0] : l2[1]));
  }
  if (M == 8) return dpp_fetch<0x128>(v);  // row_ror:8
  if (M == 4) {
    // banks 0 and 2 of a row read four lanes up (row_shl:4), banks 1 and 3 four lanes down (row_shr:4)
    const int l1 = __builtin_amdgcn_update_dpp(lo, lo, 0x104, 0xf, 0x5, false);
    const int h1 = __builtin_amdgcn_update_dpp(hi, hi, 0x104, 0xf, 0x5, false);
    return __hiloint2double(__builtin_amdgcn_update_dpp(h1, hi, 0x114, 0xf, 0xA, false),
                            __builtin_amdgcn_update_dpp(l1, lo, 0x114, 0xf, 0xA, false));
  }
  if (M == 2) return dpp_fetch<0x4E>(v);  // quad_perm [2,3,0,1]
  return dpp_fetch<0xB1>(v);              // quad_perm [1,0,3,2]
}
__device__ __forceinline__ double wave_sum(double v) {  // xor butterfly: every lane ends with the same bits
  v += xor_fetch<32>(v); v += xor_fetch<16>(v); v += xor_fetch<8>(v);
  v += xor_fetch<4>(v); v += xor_fetch<2>(v); v += xor_fetch<1>(v);
  return v;
}
#define MADICP_TB_KEEP_STEP(M, CMP) { const double o = xor_fetch<M>(v); if (CMP) v = o; }
__device__ __forceinline__ double wave_min_keep(double v) {  // min with "keep mine unless the other is smaller"
  MADICP_TB_KEEP_STEP(32, o < v) MADICP_TB_KEEP_STEP(16, o < v) MADICP_TB_KEEP_STEP(8, o < v)
  MADICP_TB_KEEP_STEP(4, o < v) MADICP_TB_KEEP_STEP(2, o < v) MADICP_TB_KEEP_STEP(1, o < v)
  return v;
}
__device__ __forceinline__ double wave_max_keep(double v) {
  MADICP_TB_KEEP_STEP(32, v < o) MADICP_TB_KEEP_STEP(16, v < o) MADICP_TB_KEEP_STEP(8, v < o)
  MADICP_TB_KEEP_STEP(4, v < o) MADICP_TB_KEEP_STEP(2, v < o) MADICP_TB_KEEP_STEP(1, v < o)
  return v;
}
#undef MADICP_TB_KEEP_STEP
__device__ __forceinline__ int wave_sum_int(int v) {
  #pragma unroll
  for (int m = 32; m > 0; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}

// ---- init ---------------------------------------------------------------------------------------------------
// Workgroup 0 clears the State and sets the root up; the others clear the leaf-start marks (no fill commands in front of
// a build: each was a dispatch of its own).  256 threads.
__global__ __launch_bounds__(256) void tb_init(const Params P, int clear_wgs) {
  if ((int)blockIdx.x > clear_wgs) {
    // The workgroups behind the clearing ones: the ROOT's nine sums per 2048-point chunk (chip regime: what tb_chip_sums did
    // as a launch of its own — the first launches of a build are spaced by the host's enqueue time, not by their work, so a
    // launch less is ~8 us less).  Needs nothing the other workgroups write: the root is points [0, n) of the caller's cloud.
    __shared__ double s_red[4][9];
    const int chunk = (int)blockIdx.x - 1 - clear_wgs;
    const double* __restrict__ in = P.cloud;
    const int cb = chunk * kChunk, ce = min(cb + kChunk, P.n_points);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    double s[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    {  // 8 points per thread, every load issued before the first add
      double x[8], y[8], z[8];
      bool ok[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = cb + (int)threadIdx.x + 256 * u;
        ok[u] = i < ce;
        const long j = ok[u] ? i : cb;
        x[u] = in[3 * j]; y[u] = in[3 * j + 1]; z[u] = in[3 * j + 2];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (ok[u]) add_point(s, x[u], y[u], z[u]);
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) s[k] = wave_sum(s[k]);
    if (lane == 0) {
#pragma unroll
      for (int k = 0; k < 9; ++k) s_red[wv][k] = s[k];
    }
    __syncthreads();
    if (threadIdx.x < 9)
      level_part(P, 0)[(long)chunk * 18 + threadIdx.x] = ((s_red[0][threadIdx.x] + s_red[1][threadIdx.x]) + s_red[2][threadIdx.x]) + s_red[3][threadIdx.x];
    return;
  }
  if (blockIdx.x != 0) {
    const long n = (long)P.n_points + 1;
    uint32_t* __restrict__ m = P.leaf_start;  // (256-byte aligned, padded to a multiple of 4 and more: ensure_scratch)
    const long n4 = (n + 3) / 4;
    for (long i = (long)(blockIdx.x - 1) * blockDim.x + threadIdx.x; i < n4; i += (long)clear_wgs * blockDim.x)
      reinterpret_cast<uint4*>(m)[i] = make_uint4(0u, 0u, 0u, 0u);
    return;
  }
  State* st = P.st;
  {
    uint4* z = reinterpret_cast<uint4*>(st);
    for (int i = threadIdx.x; i < (int)(sizeof(State) / sizeof(uint4)); i += blockDim.x) z[i] = make_uint4(0u, 0u, 0u, 0u);
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  BNode& r = P.nodes[0];
  #pragma unroll
  for (int i = 0; i < 3; ++i) { r.mean[i] = 0; r.dir[i] = 0; r.col0[i] = 0; r.plane_n[i] = 0; r.small_n[i] = 0; }
  r.bbox0 = 0.0;
  r.begin = 0;
  r.end = P.n_points;
  r.mid = 0;
  r.left_turns = 0;
  r.flags = 0;
  r.level = 0;
  r.sum_first = 0;
  r.sum_n = 0;
  st->n_nodes.v = 1;
  enqueue_single(P, 0, 0, P.n_points, 0);
}
static_assert(sizeof(State) % sizeof(uint4) == 0, "State is cleared 16 bytes at a time");

// development instrumentation (-DMADICP_TB_STAMPS, tools/tb_stamps.py): per level, the 100 MHz wall clock at a few points
#ifdef MADICP_TB_STAMPS
// [level][workgroup < 512][wave 4][slot 16]: plain stores by lane 0 of each wave (no atomics: they perturb what they measure)
__device__ unsigned long long g_tb_stamps[24][512][4][16];
__device__ __forceinline__ void tb_stamp(int level, int k) {
  if (level < 24 && blockIdx.x < 512 && k < 16) g_tb_stamps[level][blockIdx.x][(threadIdx.x >> 6) & 3][k] = wall_clock64();
}
#define TB_STAMP_MIN(level, k) tb_stamp(level, k)
#define TB_STAMP_MAX(level, k) tb_stamp(level, k)
#else
#define TB_STAMP_MIN(level, k)
#define TB_STAMP_MAX(level, k)
#endif

typedef double vd2a __attribute__((ext_vector_type(2), aligned(8)));  // 16-byte load of two doubles at 8-byte alignment

// ---- wave regime: one wavefront per node ------------------------------------------------------------------------
// What a node hands to the allocation step that follows it (ids and queue slots come from one atomic per workgroup)
struct Split {
  bool split;
  int b, mid, e;
  double col0[3];
  double ext0;
  double sL[9], sR[9];  // sums of the points that went left / right (the children start from them)
  Inherit inh;          // what the node read of itself when it started
};

// clamped 8-deep strided access: lane's points i0, i0 + 64, ..., i0 + 448 of [b, e); all twenty-four loads are issued
// before the first use (a dependent-latency loop of one load per iteration costs ~700 cycles per 64 points)
#ifndef MADICP_TB_WU
#define MADICP_TB_WU 8
#endif
constexpr int kWU = MADICP_TB_WU;
static_assert(kTeamMin <= 64 * kWU, "a wave-regime node is one batch of kWU points per lane");
#define TB_LOAD4(in, i0, e, b, x, y, z, ok)                         \
  _Pragma("unroll") for (int u_ = 0; u_ < kWU; ++u_) {             \
    const int i_ = (i0) + 64 * u_;                                  \
    ok[u_] = i_ < (e);                                              \
    const long j_ = ok[u_] ? i_ : (b);                              \
    x[u_] = in[3 * j_]; y[u_] = in[3 * j_ + 1]; z[u_] = in[3 * j_ + 2]; \
  }

// Everything of a node except handing out the children's ids: statistics, leaf test, a leaf's representative, or the
// stable scatter of an internal node.  Whole wave, every lane the same control flow.
constexpr int kRedStride = 65;  // doubles per row of a wave's reduction scratch (64 lanes + 1: lanes reading different rows
                                // hit different banks)
__device__ __forceinline__ Split wave_node(const Params& P, const int4 ent, double* red /* LDS, 18 x kRedStride of this wave */) {
  const int lane = threadIdx.x & 63;
  const int id = ent.x, b = ent.y, e = ent.z, n = e - b, level = ent.w;
  BNode& nd = P.nodes[id];
  const double* __restrict__ in = level_in(P, level);
  Split sp;
  sp.split = false;
  sp.b = b; sp.e = e; sp.mid = b;
  sp.col0[0] = sp.col0[1] = sp.col0[2] = 0.0;
  sp.ext0 = 0.0;
  // Everything the node needs of itself in ONE batch of independent loads, and its points touched (one load per
  // 64-byte line, first 1365 points) so that they are on their way while the eigen-solve runs: a level is a chain of
  // dependent first touches of memory another kernel has just written — count -> queue entry -> node -> points ->
  // atomics — at ~2 us each, and that chain, not arithmetic or bandwidth, was most of a level's time.
  sp.inh = load_inherit(nd, level);
  double s9[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) s9[k] = nd.sums[k];
#ifndef MADICP_TB_TOUCH
#define MADICP_TB_TOUCH 0
#endif
  constexpr int kTouch = MADICP_TB_TOUCH;  // line-touches made (round 2-3: eight, worth 1.5 us of a level then; round 4: the sixteen registers they hold across the eigen-solve cost 18 spilled registers, 11 us of a build — none)
  double touch[kTouch > 0 ? kTouch : 1];
#pragma unroll
  for (int u = 0; u < kTouch; ++u) {
    const long j = min(8 * ((long)lane + 64 * u), 3 * (long)n - 1);  // doubles: one per 64-byte line
    touch[u] = in[3 * (long)b + j];
  }
  double mean[3], V[9], w[3], ext[3];
  if (lane == 0) TB_STAMP_MAX(level, 3);
  if (!(sp.inh.flags & kLeafPending) && n > 64 * kWU) {  // (never queued here: enqueue_single / team_node send such nodes to the team regime)
    if (lane == 0) P.st->n_nodes.error = 1;
    return sp;
  }
  if (sp.inh.flags & kLeafPending) {  // a chip-regime node that turned out to be a leaf: statistics are already there
#pragma unroll
    for (int i = 0; i < 3; ++i) mean[i] = nd.mean[i];
    V[0] = nd.col0[0]; V[3] = nd.col0[1]; V[6] = nd.col0[2];
  } else {
    double s[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (sp.inh.flags & kHasSums) {  // the parent's scatter sweep already added this node's points up
#pragma unroll
      for (int k = 0; k < 9; ++k) s[k] = s9[k];
    } else if (sp.inh.flags & kChunkSums) {
      chunk_sums(P, sp.inh, s);
    } else {
      for (int base = b; base < e; base += 64 * kWU) {  // (wave-uniform trip count: every lane takes part in every step)
        const int i0 = base + lane;
        double x[kWU], y[kWU], z[kWU];
        bool ok[kWU];
        TB_LOAD4(in, i0, e, b, x, y, z, ok)
#pragma unroll
        for (int u = 0; u < kWU; ++u)
          if (ok[u]) add_point(s, x[u], y[u], z[u]);
      }
#pragma unroll
      for (int k = 0; k < 9; ++k) s[k] = wave_sum(s[k]);
    }
    double cov[9];
    if (lane == 0) TB_STAMP_MAX(level, 4);
    mean_cov_from_sums(s, n, mean, cov);
    madicp_host::eig3_sym(cov, w, V);
#pragma unroll
    for (int u = 0; u < kTouch; ++u) asm volatile("" ::"v"(touch[u]));  // (the touches are consumed here, not before)
    if (lane == 0) TB_STAMP_MAX(level, 5);
    // A wave-regime node holds at most kTeamMin = 512 points: ONE batch of eight points per lane, which stay in registers
    // from the sweep that reads them to the scatter that writes them.
    // Sweep A: extents in the eigen frame, the side of every point, the children's sums, and the two rank tables of the
    // split's permutation (common/split_order.h) — the positions of the points that go left, in order, from the front of
    // the wavefront's table (LDS: the reduction scratch, free until the scatter is done), of those that go right from its
    // back.  A point's rank is the running count of the ballots before it; nothing of this needs the totals, so it runs
    // before the leaf test can be made.
    double lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
    double sL[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, sR[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned short* stab = reinterpret_cast<unsigned short*>(red);
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    double x[kWU], y[kWU], z[kWU];
    bool ok[kWU];
    int lbp[kWU];
    unsigned int lbits = 0;
    int lcount = 0;  // lefts in front of the current 64 points (wave-uniform)
    {
      const int i0 = b + lane;
      TB_LOAD4(in, i0, e, b, x, y, z, ok)
#pragma unroll
      for (int u = 0; u < kWU; ++u) {  // (wave-uniform: the ballots below need every lane)
        double v[3] = {0, 0, 0};
        if (ok[u]) {
          eigen_coords(V, mean, x[u], y[u], z[u], v);
          minmax_update(lo, hi, v);
        }
        const bool left = ok[u] && v[2] < 0.0;  // the split test of mad_tree.cpp:96 (v[2] holds its very products)
        const unsigned long long lm = __ballot(left);
        lbp[u] = lcount + __popcll(lm & lt);
        lbits |= (left ? 1u : 0u) << u;
        if (ok[u]) {
          const int p = lane + 64 * u;
          stab[left ? lbp[u] : n - 1 - (p - lbp[u])] = (unsigned short)p;
          if (left) add_point(sL, x[u], y[u], z[u]); else add_point(sR, x[u], y[u], z[u]);
        }
        lcount += __popcll(lm);
      }
    }
    if (lane == 0) TB_STAMP_MAX(level, 6);
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      lo[a] = wave_min_keep(lo[a]);
      hi[a] = wave_max_keep(hi[a]);
      ext[a] = hi[a] - lo[a];
    }
    const int nl = lcount;
    if (lane == 0) nd.bbox0 = ext[0];
    const bool leaf = (ext[2] < P.b_max) || nl == 0 || nl == n;  // (an empty side cannot be split: b_max <= 0 or NaN input)
    if (!leaf) {
      const int mid = b + nl;
      {  // The scatter: every point to the place the reference's `split` (utils.h:37-52) would have left it in.  The tables
         // were written by this wavefront's other lanes.
        wave_lds_order();
        double* __restrict__ out = level_out(P, level);
        int dst[kWU];
#pragma unroll
        for (int u = 0; u < kWU; ++u) {
          dst[u] = 0;
          if (ok[u]) {
            const madicp_host::SplitPlan sp2 = madicp_host::split_plan((lbits >> u) & 1u, lane + 64 * u, lbp[u], nl, n);
            dst[u] = sp2.idx;
            if (sp2.kind == 1) dst[u] = stab[n - 1 - sp2.idx];
            if (sp2.kind == 2) dst[u] = (int)stab[sp2.idx] - 1;
          }
        }
#pragma unroll
        for (int u = 0; u < kWU; ++u)
          if (ok[u]) {
            const long d = (long)b + dst[u];
            out[3 * d] = x[u]; out[3 * d + 1] = y[u]; out[3 * d + 2] = z[u];
          }
        wave_lds_order();  // (the table's memory is the reduction scratch of the next block)
      }
      if (!kSerialSmall || nl > kSmallMax || n - nl > kSmallMax) {  // (quad-regime children add their own points up)
         // the children's 18 sums are only needed by lane 0 (it writes the child records): through LDS — every lane
         // stores its 18 partials (column-major, conflict-free), lanes 0..17 add one column each in lane order, lane 0
         // collects — instead of 18 xor butterflies through the LDS crossbar (216 ds_bpermute: ~3 us of every node)
#pragma unroll
        for (int k = 0; k < 9; ++k) {
          red[k * kRedStride + lane] = sL[k];
          red[(9 + k) * kRedStride + lane] = sR[k];
        }
        wave_lds_order();
        double col = 0.0;
        if (lane < 18) {
          for (int j = 0; j < 64; ++j) col += red[lane * kRedStride + j];
        }
        wave_lds_order();
        if (lane < 18) red[lane] = col;  // (row 0, lanes 0..17: every row has been read by now)
        wave_lds_order();
        if (lane == 0) {
#pragma unroll
          for (int k = 0; k < 9; ++k) { sp.sL[k] = red[k]; sp.sR[k] = red[9 + k]; }
        }
        wave_lds_order();  // (the next node's stores must not overtake these reads)
      }
      if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 3; ++i) { nd.mean[i] = mean[i]; nd.dir[i] = V[3 * i + 2]; nd.col0[i] = V[3 * i]; }
        nd.mid = mid;
      }
      sp.split = true;
      sp.mid = mid;
      sp.col0[0] = V[0]; sp.col0[1] = V[3]; sp.col0[2] = V[6];
      sp.ext0 = ext[0];
      if (lane == 0) TB_STAMP_MAX(level, 7);
      return sp;
    }
  }
  // leaf: normal, and the member nearest to the centroid (first one on ties, mad_tree.cpp:76-86)
  double best = 1.7976931348623157e308;
  int besti = 0x7fffffff;
  for (int i = b + lane; i < e; i += 64) {
    const double d[3] = {in[3 * (long)i] - mean[0], in[3 * (long)i + 1] - mean[1], in[3 * (long)i + 2] - mean[2]};
    const double dist = madicp_host::norm3(d);
    if (dist < best) { best = dist; besti = i; }
  }
#pragma unroll
  for (int m = 32; m > 0; m >>= 1) {
    const double ob = __shfl_xor(best, m, 64);
    const int oi = __shfl_xor(besti, m, 64);
    if (ob < best || (ob == best && oi < besti)) { best = ob; besti = oi; }
  }
  if (besti == 0x7fffffff) besti = b;  // every distance NaN: the reference keeps *begin
  if (lane == 0) {
    double nrm[3];
    leaf_normal(sp.inh, n, V, nrm);
#pragma unroll
    for (int i = 0; i < 3; ++i) { nd.mean[i] = in[3 * (long)besti + i]; nd.dir[i] = nrm[i]; }
    nd.flags = (sp.inh.flags & ~kLeafPending) | kLeaf | kDone;
    P.leaf_start[b] = 1u;
  }
  return sp;
}

// ---- quad regime: FOUR lanes per node of at most kSmallMax points (16 nodes per wavefront) -------------------------
// A single lane per node makes the per-node chain long because one lane does everything serially (a 32-point sweep is ~12 us on a
// wave that has its SIMD to itself).  Four lanes share a node here: point i of the node belongs to lane i % 4, the
// scatter positions come from the quad's four bits of a wave ballot (like the wave regime, with a quad as the "wave"),
// sums and extents are reduced over the quad with two xor shuffles.  The eigen-solve runs on all four lanes (identical
// inputs, identical results): 16 solves per wave instead of 64, still 16 times fewer than the wave regime's one.
// Control flow is wave-uniform (`steps` = the wave's longest node, lanes past their node's end are masked out), so the
// ballots and shuffles always see whole quads.
__device__ __forceinline__ double quad_sum(double v) {
  v += xor_fetch<1>(v);
  v += xor_fetch<2>(v);
  return v;
}
// Nodes of the quad regime (at most 32 points: most nodes of a tree, nearly all its leaves) add their points up ONE AFTER
// THE OTHER in member order — the reference's own chain (utils.h:54-73) — so their centroid and covariance are the host
// builder's bit for bit: lane 0 of the quad carries the three coordinate sums, lane 1 (xx, xy, xz), lane 2 (yy, yz, zz),
// every point broadcast to the four lanes with a DPP quad permute.  (Larger nodes cannot: a chain of 500 or 100 000 dependent
// additions is the one thing a wavefront is slow at.)  -DMADICP_TB_SERIAL_SMALL=0 restores the lane-strided sums.
template <int K>
__device__ __forceinline__ double quad_bcast(double v) {  // the value of lane K of the caller's quad
  constexpr int ctrl = K | (K << 2) | (K << 4) | (K << 6);  // quad_perm:[K,K,K,K]
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), ctrl, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), ctrl, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ Split quad_node(const Params& P, const int4 ent, bool have, int steps) {
  const int lane = threadIdx.x & 63, ql = lane & 3, qshift = lane & ~3;
  const int id = have ? ent.x : 0, b = ent.y, e = have ? ent.z : ent.y, n = e - b, level = ent.w;
  BNode& nd = P.nodes[id];
  const double* __restrict__ in = level_in(P, level);
  Split sp;
  sp.split = false;
  sp.b = b; sp.e = e; sp.mid = b;
  sp.col0[0] = sp.col0[1] = sp.col0[2] = 0.0;
  sp.ext0 = 0.0;
  sp.inh = load_inherit(nd, level);
  double s[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) s[k] = nd.sums[k];
  const long last = (long)max(e - 1, b);
  auto load_pt = [&](int step, double& x, double& y, double& z) {
    const long j = min((long)b + 4 * step + ql, last);
    x = in[3 * j]; y = in[3 * j + 1]; z = in[3 * j + 2];
  };
  if (kSerialSmall) {
    const bool l0 = ql == 0, l1 = ql == 1, l2 = ql == 2;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    double px, py, pz;
    load_pt(0, px, py, pz);
    for (int st = 0; st < steps; ++st) {  // (wave-uniform trip count; the next four points are requested before these are used)
      const double x4 = px, y4 = py, z4 = pz;
      if (st + 1 < steps) load_pt(st + 1, px, py, pz);
#define TB_SERIAL_STEP(K)                                                                        \
      {                                                                                            \
        const double x = quad_bcast<K>(x4), y = quad_bcast<K>(y4), z = quad_bcast<K>(z4);          \
        const double u0 = l2 ? y : x, u1 = l1 ? x : y, u2 = l1 ? x : z;                            \
        const double w0 = l0 ? 1.0 : (l1 ? x : y), w1 = l0 ? 1.0 : (l1 ? y : z), w2 = l0 ? 1.0 : z; \
        if (have && 4 * st + K < n) { a0 += u0 * w0; a1 += u1 * w1; a2 += u2 * w2; }               \
      }
      TB_SERIAL_STEP(0) TB_SERIAL_STEP(1) TB_SERIAL_STEP(2) TB_SERIAL_STEP(3)
#undef TB_SERIAL_STEP
    }
    s[0] = quad_bcast<0>(a0); s[1] = quad_bcast<0>(a1); s[2] = quad_bcast<0>(a2);
    s[3] = quad_bcast<1>(a0); s[4] = quad_bcast<1>(a1); s[5] = quad_bcast<1>(a2);
    s[6] = quad_bcast<2>(a0); s[7] = quad_bcast<2>(a1); s[8] = quad_bcast<2>(a2);
  } else if (have && (sp.inh.flags & kChunkSums)) {  // a tiny child of a chip-regime node: its sums are chunk partials
    chunk_sums(P, sp.inh, s);
  } else if (!(sp.inh.flags & kHasSums)) {  // (nobody summed it: only the root of a tiny cloud) — wave-uniform loop
#pragma unroll
    for (int k = 0; k < 9; ++k) s[k] = 0.0;
    for (int st = 0; st < steps; ++st) {
      double x, y, z;
      load_pt(st, x, y, z);
      if (have && b + 4 * st + ql < e) add_point(s, x, y, z);
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) s[k] = quad_sum(s[k]);
  }
  double mean[3], cov[9], w[3], V[9];
  mean_cov_from_sums(s, max(n, 1), mean, cov);
  madicp_host::eig3_sym(cov, w, V);
  // Sweep A: extents, the children's sums, and the side of every point — the quad's four bits of a wave ballot per step,
  // collected into ONE word per node (bit p = the point at position p goes left): the word is all the split's permutation
  // needs (common/split_order.h, the rank tables of the larger regimes become bit selects)
  double lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
  double sL[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, sR[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned int mask = 0;
  double nx, ny, nz;
  load_pt(0, nx, ny, nz);
  for (int st = 0; st < steps; ++st) {  // (wave-uniform trip count; the next point is requested before this one is used)
    const double x = nx, y = ny, z = nz;
    if (st + 1 < steps) load_pt(st + 1, nx, ny, nz);
    const bool valid = have && b + 4 * st + ql < e;
    double v[3] = {0, 0, 0};
    if (valid) {
      eigen_coords(V, mean, x, y, z, v);
      minmax_update(lo, hi, v);
    }
    const bool left = valid && v[2] < 0.0;
    const unsigned int lm = (unsigned int)(__ballot(left) >> qshift) & 0xfu;
    mask |= lm << (4 * st);
    if (!kSerialSmall && valid) {  // (serial mode: the children, quad nodes themselves, add their own points up in order)
      if (left) add_point(sL, x, y, z); else add_point(sR, x, y, z);
    }
  }
  double ext[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    {  // (the quad's two xor steps, by DPP quad_perm: same partners and order as the shuffles they replace)
      double ol = xor_fetch<1>(lo[a]), oh = xor_fetch<1>(hi[a]);
      if (ol < lo[a]) lo[a] = ol;
      if (hi[a] < oh) hi[a] = oh;
      ol = xor_fetch<2>(lo[a]); oh = xor_fetch<2>(hi[a]);
      if (ol < lo[a]) lo[a] = ol;
      if (hi[a] < oh) hi[a] = oh;
    }
    ext[a] = hi[a] - lo[a];
  }
  const int nl = __popc(mask);
  const bool leaf = !have || (ext[2] < P.b_max) || nl == 0 || nl == n;
  // (both branches below keep whole quads together: `leaf` is the same in the four lanes of a node)
  if (!kSerialSmall) {
#pragma unroll
    for (int k = 0; k < 9; ++k) {  // outside the branch: the shuffles need every lane
      sp.sL[k] = quad_sum(sL[k]);
      sp.sR[k] = quad_sum(sR[k]);
    }
  }
  // Sweep B, ONE more pass over the points (the next four requested before these are used: the passes are chains of L1 round
  // trips).  Internal nodes: every point to the place the reference's `split` (utils.h:37-52) would have left it in.  Leaves:
  // the member nearest to the centroid — every lane over its own points, then the quad's best, smallest index on ties.
  double best = 1.7976931348623157e308;
  int besti = 0x7fffffff;
  {
    double* __restrict__ out = level_out(P, level);
    double qx, qy, qz;
    load_pt(0, qx, qy, qz);
    for (int st = 0; st < steps; ++st) {
      const double x = qx, y = qy, z = qz;
      if (st + 1 < steps) load_pt(st + 1, qx, qy, qz);
      const int pp = 4 * st + ql;
      if (have && pp < n) {
        if (!leaf) {
          const long d = (long)b + madicp_host::split_dst_small(mask, n, pp);
          out[3 * d] = x; out[3 * d + 1] = y; out[3 * d + 2] = z;
        } else {
          const double dv[3] = {x - mean[0], y - mean[1], z - mean[2]};
          const double dist = madicp_host::norm3(dv);
          if (dist < best) { best = dist; besti = b + pp; }
        }
      }
    }
  }
  {
    double ob = xor_fetch<1>(best);
    int oi = __builtin_amdgcn_mov_dpp(besti, 0xB1, 0xf, 0xf, false);  // quad_perm [1,0,3,2]
    if (ob < best || (ob == best && oi < besti)) { best = ob; besti = oi; }
    ob = xor_fetch<2>(best);
    oi = __builtin_amdgcn_mov_dpp(besti, 0x4E, 0xf, 0xf, false);      // quad_perm [2,3,0,1]
    if (ob < best || (ob == best && oi < besti)) { best = ob; besti = oi; }
  }
  if (!have) return sp;
  if (ql == 0) nd.bbox0 = ext[0];
  if (!leaf) {
    const int mid = b + nl;
    if (ql == 0) {
#pragma unroll
      for (int k = 0; k < 3; ++k) { nd.mean[k] = mean[k]; nd.dir[k] = V[3 * k + 2]; nd.col0[k] = V[3 * k]; }
      nd.mid = mid;
    }
    sp.split = true;
    sp.mid = mid;
    sp.col0[0] = V[0]; sp.col0[1] = V[3]; sp.col0[2] = V[6];
    sp.ext0 = ext[0];
    return sp;
  }
  if (besti == 0x7fffffff) besti = b;  // every distance NaN: the reference keeps *begin
  if (ql == 0) {
    double nrm[3];
    leaf_normal(sp.inh, n, V, nrm);
#pragma unroll
    for (int k = 0; k < 3; ++k) { nd.mean[k] = in[3 * (long)besti + k]; nd.dir[k] = nrm[k]; }
    nd.flags = sp.inh.flags | kLeaf | kDone;
    P.leaf_start[b] = 1u;
  }
  return sp;
}

// the two children of a split node: records, and their places in the next level's queues
__device__ __forceinline__ void emit_children(const Params& P, int id, const Split& sp, int c, int slot_small, int slot_wave, int next_step) {
  BNode& nd = P.nodes[id];
  const int level = sp.inh.level;
  const int n = sp.e - sp.b;
  nd.child = c;
  make_child(P.nodes[c], sp.inh, id, sp.col0, sp.ext0, n, P.b_min, sp.b, sp.mid, true);
  make_child(P.nodes[c + 1], sp.inh, id, sp.col0, sp.ext0, n, P.b_min, sp.mid, sp.e, false);
  const int nL = sp.mid - sp.b, nR = sp.e - sp.mid;
  // the children's sums, accumulated by this node's sweep — except for children of the quad regime in serial mode, which add
  // their own points up in member order
  if (!kSerialSmall || nL > kSmallMax) {
#pragma unroll
    for (int k = 0; k < 9; ++k) P.nodes[c].sums[k] = sp.sL[k];
    P.nodes[c].flags |= kHasSums;
  }
  if (!kSerialSmall || nR > kSmallMax) {
#pragma unroll
    for (int k = 0; k < 9; ++k) P.nodes[c + 1].sums[k] = sp.sR[k];
    P.nodes[c + 1].flags |= kHasSums;
  }
  nd.flags = sp.inh.flags | kDone;
  // (children of a wave/lane node are never chip-regime: n <= 4096, or past the chip levels)
  const int4 eL = make_int4(c, sp.b, sp.mid, level + 1), eR = make_int4(c + 1, sp.mid, sp.e, level + 1);
  if (nL <= kSmallMax) level_small(P, next_step)[slot_small++] = eL; else level_q(P, next_step)[slot_wave++] = eL;
  if (nR <= kSmallMax) level_small(P, next_step)[slot_small] = eR; else level_q(P, next_step)[slot_wave] = eR;
}

// ---- team regime: a node of more than kTeamMin points past the chip levels, ONE workgroup (four wavefronts) per node -----
// One wavefront per such node made steps 6-8 of a 120 k-point scan 72 + 46 + 33 us (nodes of 500 - 6 800 points, two
// sweeps of 512 points a batch each).  Here every wavefront owns a contiguous quarter of the node — the slices play the
// part of the chip regime's chunks: slice-local rank tables, a prefix over four counts, the same search
// (common/split_order.h) — and the workgroup synchronises with barriers where the chunks of a chip node need a kernel
// boundary.  The children compute their own sums (a sweep of at most kTeamMin points for a wave-regime child).
constexpr int kTeamWaves = 4;
__device__ __forceinline__ void nearest_update(double& best, int& besti, const double* mean, double x, double y, double z, int i) {
  const double d[3] = {x - mean[0], y - mean[1], z - mean[2]};
  const double dist = madicp_host::norm3(d);
  if (dist < best || (dist == best && i < besti)) { best = dist; besti = i; }
}
__device__ __forceinline__ void team_node(const Params& P, const int4 ent, int step) {
  __shared__ double s_sum[kTeamWaves][9];
  __shared__ double s_ext[kTeamWaves][6];
  __shared__ int s_nl[kTeamWaves];
  __shared__ int s_pref[kTeamWaves + 1];
  __shared__ double s_best[kTeamWaves];
  __shared__ int s_besti[kTeamWaves];
  State* st = P.st;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  const int id = ent.x, b = ent.y, e = ent.z, n = e - b, level = ent.w;
  BNode& nd = P.nodes[id];
  const Inherit inh = load_inherit(nd, level);
  double s9[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) s9[k] = nd.sums[k];
  const double* __restrict__ in = level_in(P, level);
  double* __restrict__ out = level_out(P, level);
  const int S = (n + kTeamWaves - 1) / kTeamWaves;  // the wavefront's slice of the node
  const int sb = min(b + wv * S, e), se = min(sb + S, e);
  // ---- the node's nine sums: handed down by the parent, or a sweep of the slices (added in slice order)
  double s[9];
  if (inh.flags & kHasSums) {
#pragma unroll
    for (int k = 0; k < 9; ++k) s[k] = s9[k];
  } else if (inh.flags & kChunkSums) {
    chunk_sums(P, inh, s);
  } else {
#pragma unroll
    for (int k = 0; k < 9; ++k) s[k] = 0.0;
    for (int base = sb; base < se; base += 64 * kWU) {
      const int i0 = base + lane;
      double x[kWU], y[kWU], z[kWU];
      bool ok[kWU];
      TB_LOAD4(in, i0, se, sb, x, y, z, ok)
#pragma unroll
      for (int u = 0; u < kWU; ++u)
        if (ok[u]) add_point(s, x[u], y[u], z[u]);
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) s[k] = wave_sum(s[k]);
    if (lane == 0) {
#pragma unroll
      for (int k = 0; k < 9; ++k) s_sum[wv][k] = s[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 9; ++k) s[k] = ((s_sum[0][k] + s_sum[1][k]) + s_sum[2][k]) + s_sum[3][k];
  }
  double mean[3], cov[9], w3[3], V[9];
  mean_cov_from_sums(s, n, mean, cov);
  madicp_host::eig3_sym(cov, w3, V);  // (every wavefront the same values: nothing to broadcast)
  // ---- sweep A over the slice: extents, sides, slice-local rank tables
  double lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
  int32_t* tab = P.tab;
  int lcount = 0;
  for (int base = sb; base < se; base += 64 * kWU) {  // (wave-uniform trip count)
    const int i0 = base + lane;
    double x[kWU], y[kWU], z[kWU];
    bool ok[kWU];
    TB_LOAD4(in, i0, se, sb, x, y, z, ok)
#pragma unroll
    for (int u = 0; u < kWU; ++u) {
      double v[3] = {0, 0, 0};
      if (ok[u]) {
        eigen_coords(V, mean, x[u], y[u], z[u], v);
        minmax_update(lo, hi, v);
      }
      const bool left = ok[u] && v[2] < 0.0;
      const unsigned long long lm = __ballot(left);
      if (ok[u]) {
        const int q = i0 + 64 * u - sb;  // position in the slice
        const int lbp = lcount + __popcll(lm & lt);
        tab[left ? (long)sb + lbp : (long)se - 1 - (q - lbp)] = q + (sb - b);
      }
      lcount += __popcll(lm);
    }
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    lo[a] = wave_min_keep(lo[a]);
    hi[a] = wave_max_keep(hi[a]);
  }
  if (lane == 0) {
#pragma unroll
    for (int a = 0; a < 3; ++a) { s_ext[wv][a] = lo[a]; s_ext[wv][3 + a] = hi[a]; }
    s_nl[wv] = lcount;
  }
  __syncthreads();  // (also: every slice's table entries are written)
  double ext[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    double L = s_ext[0][a], H = s_ext[0][3 + a];
#pragma unroll
    for (int w = 1; w < kTeamWaves; ++w) {
      if (s_ext[w][a] < L) L = s_ext[w][a];
      if (H < s_ext[w][3 + a]) H = s_ext[w][3 + a];
    }
    ext[a] = H - L;
  }
  const int nl = ((s_nl[0] + s_nl[1]) + s_nl[2]) + s_nl[3];
  if (threadIdx.x <= kTeamWaves) {  // exclusive prefix of the slices' left counts
    int r = 0;
    for (int w = 0; w < (int)threadIdx.x; ++w) r += s_nl[w];
    s_pref[threadIdx.x] = r;
  }
  const bool leaf = (ext[2] < P.b_max) || nl == 0 || nl == n;
  if (leaf) {  // (rare: more than kTeamMin points within b_max of each other)
    double best = 1.7976931348623157e308;
    int besti = 0x7fffffff;
    for (int i = sb + lane; i < se; i += 64) nearest_update(best, besti, mean, in[3 * (long)i], in[3 * (long)i + 1], in[3 * (long)i + 2], i);
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) {
      const double ob = __shfl_xor(best, m, 64);
      const int oi = __shfl_xor(besti, m, 64);
      if (ob < best || (ob == best && oi < besti)) { best = ob; besti = oi; }
    }
    if (lane == 0) { s_best[wv] = best; s_besti[wv] = besti; }
    __syncthreads();
    if (threadIdx.x == 0) {
      for (int k = 1; k < kTeamWaves; ++k)
        if (s_best[k] < best || (s_best[k] == best && s_besti[k] < besti)) { best = s_best[k]; besti = s_besti[k]; }
      if (besti == 0x7fffffff) besti = b;  // every distance NaN: the reference keeps *begin
      double nrm[3];
      leaf_normal(inh, n, V, nrm);
#pragma unroll
      for (int i = 0; i < 3; ++i) { nd.mean[i] = in[3 * (long)besti + i]; nd.dir[i] = nrm[i]; }
      nd.bbox0 = ext[0];
      nd.flags = inh.flags | kLeaf | kDone;
      P.leaf_start[b] = 1u;
    }
    return;
  }
  // ---- the node's record and its children (thread 0) while everybody scatters
  if (threadIdx.x == 0) {
    const int c = atomicAdd(&st->n_nodes.v, 2);
    if (c + 2 > P.node_cap) {
      st->n_nodes.error = 1;
    } else {
      const double col0[3] = {V[0], V[3], V[6]};
#pragma unroll
      for (int i = 0; i < 3; ++i) { nd.mean[i] = mean[i]; nd.dir[i] = V[3 * i + 2]; nd.col0[i] = V[3 * i]; }
      nd.bbox0 = ext[0];
      nd.mid = b + nl;
      nd.flags = inh.flags | kDone;
      nd.child = c;
      make_child(P.nodes[c], inh, id, col0, ext[0], n, P.b_min, b, b + nl, true);
      make_child(P.nodes[c + 1], inh, id, col0, ext[0], n, P.b_min, b + nl, e, false);
#pragma unroll
      for (int k = 0; k < 2; ++k) {  // (a handful of team nodes per level: one atomic per child)
        const int cb = k ? b + nl : b, ce = k ? e : b + nl, cn = ce - cb;
        const int4 ce4 = make_int4(c + k, cb, ce, level + 1);
        if (cn <= kSmallMax) level_small(P, step + 1)[atomicAdd(&st->small_count[step + 1].v, 1)] = ce4;
        else if (cn > kTeamMin) level_team(P, step + 1)[atomicAdd(&st->team_count[step + 1].v, 1)] = ce4;
        else level_q(P, step + 1)[atomicAdd(&st->q_count[step + 1].v, 1)] = ce4;
      }
    }
  }
  __syncthreads();  // (s_pref)
  // ---- sweep B: every point to the place the reference's split would have left it in; the slices are the "chunks" of the
  // rank search
  auto lefts_of = [&](int c) { return s_pref[c + 1] - s_pref[c]; };
  int lc2 = s_pref[wv];
  for (int base = sb; base < se; base += 64 * kWU) {
    const int i0 = base + lane;
    double x[kWU], y[kWU], z[kWU];
    bool ok[kWU];
    TB_LOAD4(in, i0, se, sb, x, y, z, ok)
    int dst[kWU];
#pragma unroll
    for (int u = 0; u < kWU; ++u) {
      double v[3] = {0, 0, 0};
      if (ok[u]) eigen_coords(V, mean, x[u], y[u], z[u], v);
      const bool left = ok[u] && v[2] < 0.0;
      const unsigned long long lm = __ballot(left);
      dst[u] = 0;
      if (ok[u]) {
        const madicp_host::SplitPlan pl = madicp_host::split_plan(left, i0 + 64 * u - b, lc2 + __popcll(lm & lt), nl, n);
        dst[u] = pl.idx;
        if (pl.kind == 1) {
          int c, local;
          madicp_host::find_right_chunk(s_pref, kTeamWaves, 0, kTeamWaves, S, n, pl.idx, lefts_of, c, local);
          dst[u] = tab[(long)b + min(n, (c + 1) * S) - 1 - local];
        } else if (pl.kind == 2) {
          int c, local;
          madicp_host::find_left_chunk(s_pref, kTeamWaves, 0, kTeamWaves, pl.idx, lefts_of, c, local);
          dst[u] = tab[(long)b + (long)c * S + local] - 1;
        }
      }
      lc2 += __popcll(lm);
    }
#pragma unroll
    for (int u = 0; u < kWU; ++u)
      if (ok[u]) {
        const long d = (long)b + dst[u];
        out[3 * d] = x[u]; out[3 * d + 1] = y[u]; out[3 * d + 2] = z[u];
      }
  }
}

// ---- breadth-first layout of the first kTopLevels levels (what layout_top does on the host for uploaded trees) ---
// ONE workgroup of 256 threads, over the TEMPORARY nodes (a node knows its children: BNode::child), so that it needs neither
// the leaf scan nor the emitted array: it runs in three parts, each as the FIRST workgroup of a level launch late enough for every node it reads
// to be finished (frontend_capi.inc.h, kTopParts: a node of level L is finished by step L + 5) — 6-15 us each beside a step's
// own nodes (a level is one dependent memory hop, the children's flags and child ids; eleven levels were 19 us as a kernel of
// their own behind the emission, 26-34 us as ONE extra workgroup of the summary kernel or of a level kernel — whatever came
// next had to wait for it — and on a side stream the fork's event cost the main stream 18 us).  Output: the temporary ids of the internal nodes of levels
// 0 .. top_levels - 1 in breadth-first order (P.top_ids) and their link words (P.top_link: positions of the children in that
// order, or "leaf" / "below the top"); everything that needs the leaf scan (DFS index, leaf ordinals) is filled in by the
// emission kernel, entry by entry.  A level has at most 1024 internal nodes: four consecutive entries per thread.
__device__ __forceinline__ void top_bfs_body(const Params& P, int lev_from, int lev_to, int top_levels, int top_max) {
  __shared__ int s_id[2][1024], s_child[2][1024];
  __shared__ int s_w[4];
  __shared__ int s_total;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int ncur = 0, base = 0;
  if (lev_from == 0) {
    const BNode& root = P.nodes[0];
    if (!(root.flags & kLeaf)) {
      ncur = 1;
      if (threadIdx.x == 0) { s_id[0][0] = 0; s_child[0][0] = root.child; }
    }
  } else {  // a later part picks up where the one before it stopped
    ncur = P.top_front[0];
    base = P.top_front[1];
    for (int t = threadIdx.x; t < ncur; t += blockDim.x) { s_id[0][t] = P.top_front[2 + t]; s_child[0][t] = P.top_front[2 + 1024 + t]; }
  }
  __syncthreads();
  int par = 0;
  for (int lev = lev_from; lev < lev_to && ncur > 0; ++lev, par ^= 1) {
    const bool deeper = lev + 1 < top_levels;
    int ch[4], lc[4], rc[4];
    bool on[4], l_leaf[4], r_leaf[4], l_in[4], r_in[4];
    int c = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int t = 4 * (int)threadIdx.x + u;
      on[u] = t < ncur && base + t < top_max - 1;
      ch[u] = on[u] ? s_child[par][t] : 0;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {  // (the eight loads of a thread are independent: one memory round trip per level)
      int lf = kLeaf, rf = kLeaf;
      lc[u] = rc[u] = 0;
      if (on[u]) {
        lf = P.nodes[ch[u]].flags; lc[u] = P.nodes[ch[u]].child;
        rf = P.nodes[ch[u] + 1].flags; rc[u] = P.nodes[ch[u] + 1].child;
      }
      l_leaf[u] = (lf & kLeaf) != 0;
      r_leaf[u] = (rf & kLeaf) != 0;
      l_in[u] = on[u] && !l_leaf[u] && deeper;
      r_in[u] = on[u] && !r_leaf[u] && deeper;
      c += (l_in[u] ? 1 : 0) + (r_in[u] ? 1 : 0);
    }
    int incl = c;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int o = __shfl_up(incl, d, 64);
      if (lane >= d) incl += o;
    }
    if (lane == 63) s_w[wv] = incl;
    __syncthreads();
    int off = 0;
    for (int k = 0; k < wv; ++k) off += s_w[k];
    if (threadIdx.x == blockDim.x - 1) s_total = off + incl;
    int excl = off + incl - c;
    const int next_base = base + ncur;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (!on[u]) continue;
      const int t = 4 * (int)threadIdx.x + u;
      const int first = next_base + excl;
      if (l_in[u]) { s_id[par ^ 1][excl] = ch[u]; s_child[par ^ 1][excl] = lc[u]; }
      if (r_in[u]) {
        const int at = excl + (l_in[u] ? 1 : 0);
        s_id[par ^ 1][at] = ch[u] + 1; s_child[par ^ 1][at] = rc[u];
      }
      P.top_ids[base + t] = s_id[par][t];
      P.top_link[base + t] = top_link_word(l_in[u] ? first : -1, r_in[u] ? first + (l_in[u] ? 1 : 0) : -1, l_leaf[u], r_leaf[u]);
      excl += (l_in[u] ? 1 : 0) + (r_in[u] ? 1 : 0);
    }
    __syncthreads();
    base = next_base;
    ncur = s_total;
    __syncthreads();
  }
  if (lev_to < top_levels) {  // hand the next level's entries to the next part
    if (threadIdx.x == 0) { P.top_front[0] = ncur; P.top_front[1] = base; }
    for (int t = threadIdx.x; t < ncur; t += blockDim.x) { P.top_front[2 + t] = s_id[par][t]; P.top_front[2 + 1024 + t] = s_child[par][t]; }
  }
}

// One step of the wave and quad regimes (`level` is the step: the queue index; a node's own level is in its entry).
// 256 threads = 4 wavefronts.
// (Round 3, measured at compile time and not kept: the eigen-solve as ONE out-of-line function shared by the wave and quad
// regimes — arguments and results in registers — does not free the registers it was hoped to: the caller's live state has
// to survive the call, 230 VGPRs + 168 bytes of scratch at two waves per SIMD, 168 + 392 bytes at three, against 256 + 96
// inlined.  The kernel's registers are the sweep's, not the solver's.)
#ifndef MADICP_TB_WPE
#define MADICP_TB_WPE 2
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(MADICP_TB_WPE, MADICP_TB_WPE))) void tb_level(const Params P, int level, int bfs_from, int bfs_to, int top_levels, int top_max) {
  // bfs_to > bfs_from: the first workgroup of this launch makes levels [bfs_from, bfs_to) of the top's breadth-first layout
  const int bfs = bfs_to > bfs_from ? 1 : 0;
  if (bfs && blockIdx.x == 0) {
    top_bfs_body(P, bfs_from, bfs_to, top_levels, top_max);
    return;
  }
  const int G = (int)gridDim.x - bfs, bid = (int)blockIdx.x - bfs;
  State* st = P.st;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int cntW = st->q_count[level].v, cntS = st->small_count[level].v, cntT = st->team_count[level].v;
  if (lane == 0) TB_STAMP_MIN(level, 0);
  if (level + 1 > kMaxLevels) {
    if ((cntW > 0 || cntS > 0 || cntT > 0) && bid == 0 && threadIdx.x == 0) st->n_nodes.error = 2;
    return;
  }
  // ---- team regime: the FIRST workgroups take the nodes of more than kTeamMin points, one node each (dispatched first:
  // they are the longest); the rest of the grid shares the other two queues as before
  const int wgT = cntT > 0 ? max(1, min(cntT, G - 1)) : 0;  // (always leaves a workgroup for the other queues)
  if (bid < wgT) {
    for (int t = bid; t < cntT; t += wgT) {  // (workgroup-uniform)
      team_node(P, level_team(P, level)[t], level);
      __syncthreads();  // (the team scratch is reused)
    }
    if (G > 1 || (cntW == 0 && cntS == 0)) return;
  }
  const int gx = G > 1 ? G - wgT : 1, bx = G > 1 ? bid - wgT : 0;
  // ---- wave regime: the four waves of a workgroup take four consecutive queue entries; ids and queue slots of the
  // children come from ONE atomic each per workgroup
  __shared__ int s_split[4], s_ns[4], s_nw[4];
  __shared__ int s_base_id, s_base_small, s_base_wave;
  __shared__ double s_red[4][18 * kRedStride];
  // The two regimes run side by side: the FIRST ceil(cntS / 64) workgroups take the quad-regime queue (first, because
  // workgroups are dispatched in order and only two fit a CU: behind the wave-regime workgroups the lanes would start
  // when those finish — measured: a level then costs the sum of the two sides instead of the longer one), the others
  // the wave-regime queue.
  const bool single = gx == 1;  // (tiny clouds: the one workgroup does both, one after the other)
  int wgS = min((cntS + 63) / 64, gx);
  if (!single && cntW > 0 && wgS >= gx) wgS = gx - 1;
  const int wgW = single ? 1 : gx - wgS;
  const int wblock = single ? 0 : bx - wgS;  // index among the wave-side workgroups
  const int4* qw = level_q(P, level);
  if (wblock >= 0 && wgW > 0)
  for (int t0 = wblock * 4; t0 < cntW; t0 += wgW * 4) {  // (workgroup-uniform trip count)
    const int t = t0 + wv;
    const bool active = t < cntW;
    int id = -1;
    Split sp;
    sp.split = false;
    if (active) {
      const int4 ent = qw[t];
      id = ent.x;
      sp = wave_node(P, ent, s_red[wv]);
    }
    const int nL = sp.split ? sp.mid - sp.b : 0, nR = sp.split ? sp.e - sp.mid : 0;
    const int my_small = sp.split ? ((nL <= kSmallMax) + (nR <= kSmallMax)) : 0;
    if (lane == 0) {
      s_split[wv] = sp.split ? 1 : 0;
      s_ns[wv] = my_small;
      s_nw[wv] = sp.split ? 2 - my_small : 0;
    }
    __syncthreads();
    if (lane == 0) TB_STAMP_MAX(level, 10);  // (the workgroup's four nodes are through: what follows is the allocation proper)
    if (threadIdx.x == 0) {
      const int tot = s_split[0] + s_split[1] + s_split[2] + s_split[3];
      const int ts = s_ns[0] + s_ns[1] + s_ns[2] + s_ns[3], tw = s_nw[0] + s_nw[1] + s_nw[2] + s_nw[3];
      // unconditional (adding 0 is harmless): three independent returning atomics in flight together, one round trip
      const int a0 = atomicAdd(&st->n_nodes.v, 2 * tot);
      const int a1 = atomicAdd(&st->small_count[level + 1].v, ts);
      const int a2 = atomicAdd(&st->q_count[level + 1].v, tw);
      s_base_id = a0;
      s_base_small = a1;
      s_base_wave = a2;
    }
    __syncthreads();
    if (lane == 0) TB_STAMP_MAX(level, 8);
    if (sp.split && lane == 0) {
      int before = 0, bs = 0, bw = 0;
      for (int k = 0; k < wv; ++k) { before += s_split[k]; bs += s_ns[k]; bw += s_nw[k]; }
      const int c = s_base_id + 2 * before;
      if (c + 2 > P.node_cap) {
        st->n_nodes.error = 1;
      } else {
        emit_children(P, id, sp, c, s_base_small + bs, s_base_wave + bw, level + 1);
      }
    }
    __syncthreads();
    if (lane == 0) TB_STAMP_MAX(level, 9);
  }
  // ---- quad regime: a wave takes 16 consecutive queue entries, four lanes each; one atomic each per WAVE
  const int4* qs = level_small(P, level);
  if (!single && (bx >= wgS || wgS <= 0)) return;
  const int n_waves = max(wgS, 1) * 4, wave = bx * 4 + wv;
  for (int t0 = wave * 16; t0 < cntS; t0 += n_waves * 16) {  // (wave-uniform trip count)
    const int t = t0 + (lane >> 2);
    const bool have = t < cntS;
    int4 ent = make_int4(0, 0, 0, level);
    if (have) ent = qs[t];
    int steps = have ? (ent.z - ent.y + 3) / 4 : 0;
    {  // the wave's longest node (steps <= kSmallMax / 4): eight ballots instead of a four-step butterfly through the LDS crossbar
      int mx = 0;
#pragma unroll
      for (int k = 1; k <= kSmallMax / 4; ++k)
        if (__ballot(steps >= k) != 0ull) mx = k;
      steps = mx;
    }
    Split sp = quad_node(P, ent, have, steps);
    const bool mine = sp.split && (lane & 3) == 0;
    const unsigned long long sm = __ballot(mine);
    const int tot = __popcll(sm);
    if (tot == 0) continue;
    int base_id = 0, base_q = 0;
    if (lane == 0) {  // (two independent atomics, one round trip)
      const int a0 = atomicAdd(&st->n_nodes.v, 2 * tot);
      const int a1 = atomicAdd(&st->small_count[level + 1].v, 2 * tot);
      base_id = a0;
      base_q = a1;
    }
    base_id = __builtin_amdgcn_readfirstlane(base_id);
    base_q = __builtin_amdgcn_readfirstlane(base_q);
    if (mine) {
      const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
      const int rank = __popcll(sm & lt);
      const int c = base_id + 2 * rank;
      if (c + 2 > P.node_cap) {
        st->n_nodes.error = 1;
      } else {
        emit_children(P, ent.x, sp, c, base_q + 2 * rank, 0, level + 1);  // children of a quad node are quad nodes
      }
    }
  }
  if (lane == 0) TB_STAMP_MAX(level, 1);
}

// ---- chip regime: one workgroup per 2048-point chunk of a big node; three kernels per level --------------------
// chunk slots of this level: exclusive prefix of the per-node chunk counts (list order); every workgroup computes it
struct ChunkMap {
  int node_slot;  // index into the level's big list
  int chunk;      // chunk of that node
  int first_slot; // slot of the node's chunk 0
  int n_chunks;   // chunks of the node
};
__device__ __forceinline__ int chip_prefix(const Params& P, int level, int* s_off /* LDS [kMaxBig + 1] */) {
  const int cnt = min(P.st->big_count[level].v, kMaxBig);
  const int32_t* big = level_big(P, level);
  // 256 threads, <= 256 entries: Hillis-Steele inclusive scan in LDS
  int v = 0;
  if ((int)threadIdx.x < cnt) {
    const BNode& nd = P.nodes[big[threadIdx.x]];
    v = (nd.end - nd.begin + kChunk - 1) / kChunk;
  }
  // inclusive scan over the 256 threads: wave scans (shuffles) + the four wave totals through LDS (two barriers)
  __shared__ int s_wtot[4];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int incl = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int o = __shfl_up(incl, d, 64);
    if (lane >= d) incl += o;
  }
  if (lane == 63) s_wtot[wv] = incl;
  __syncthreads();
  int off = 0;
  for (int k = 0; k < wv; ++k) off += s_wtot[k];
  if (threadIdx.x == 0) s_off[0] = 0;
  s_off[threadIdx.x + 1] = off + incl;
  __syncthreads();
  return cnt;
}
__device__ __forceinline__ ChunkMap chip_find(const int* s_off, int cnt, int slot) {
  int lo = 0, hi = cnt;  // largest j with s_off[j] <= slot
  while (hi - lo > 1) {
    const int m = (lo + hi) >> 1;
    if (s_off[m] <= slot) lo = m; else hi = m;
  }
  ChunkMap cm;
  cm.node_slot = lo;
  cm.first_slot = s_off[lo];
  cm.chunk = slot - s_off[lo];
  cm.n_chunks = s_off[lo + 1] - s_off[lo];
  return cm;
}


// C2: node statistics (recombined by every chunk of the node: partials loaded in parallel, added in chunk order),
// extents and left count of the chunk
__global__ __launch_bounds__(256) void tb_chip_stats(const Params P, int level) {
  __shared__ int s_off[kMaxBig + 1];
  __shared__ double s_part[256][9];
  __shared__ double s_tot[9], s_mean[3], s_V[9];
  __shared__ double s_lo[4][3], s_hi[4][3];
  __shared__ int s_cnt[8][4];  // lefts per (row of 256 points, wavefront) of the chunk
  const int cnt = chip_prefix(P, level, s_off);
  const int total = s_off[cnt];
  const double* __restrict__ in = level_in(P, level);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) { TB_STAMP_MIN(level, 8); }
  for (int slot = blockIdx.x; slot < total; slot += gridDim.x) {
    const ChunkMap cm = chip_find(s_off, cnt, slot);
    BNode& nd = P.nodes[level_big(P, level)[cm.node_slot]];
    const int n = nd.end - nd.begin;
    if (lane == 0) { TB_STAMP_MAX(level, 9); }
    // the node's nine sums: per-chunk partials of its PARENT's scatter (left or right half), or of tb_init (root)
    const int nflags = nd.flags;
    const int pfirst = (nflags & kChunkSums) ? nd.sum_first : cm.first_slot;
    const int pcount = (nflags & kChunkSums) ? nd.sum_n : cm.n_chunks;
    const double* part = level_part(P, level) + ((nflags & kSumRight) ? 9 : 0);
    if (threadIdx.x < 9) s_tot[threadIdx.x] = (nflags & kHasSums) ? nd.sums[threadIdx.x] : 0.0;
    for (int base = 0; base < pcount && !(nflags & kHasSums); base += 256) {
      const int c = base + threadIdx.x;
      if (c < pcount) {
#pragma unroll
        for (int k = 0; k < 9; ++k) s_part[threadIdx.x][k] = part[(long)(pfirst + c) * 18 + k];
      }
      __syncthreads();
      if (threadIdx.x < 9) {
        double a = s_tot[threadIdx.x];
        const int m = min(256, pcount - base);
        for (int c2 = 0; c2 < m; ++c2) a += s_part[c2][threadIdx.x];
        s_tot[threadIdx.x] = a;
      }
      __syncthreads();
    }
    // the chunk's points are requested BEFORE the eigen-solve (they do not depend on it): their first touch of memory the
    // previous kernel has just written (~2 us) passes while wave 0 solves
    const int cb = nd.begin + cm.chunk * kChunk, ce = min(cb + kChunk, nd.end);
    double x[8], y[8], z[8];
    bool ok[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = cb + (int)threadIdx.x + 256 * u;
      ok[u] = i < ce;
      const long j = ok[u] ? i : cb;
      x[u] = in[3 * j]; y[u] = in[3 * j + 1]; z[u] = in[3 * j + 2];
    }
    if (threadIdx.x < 64) {  // wave 0, every lane the same values
      double tot[9], mean[3], cov[9], w[3], V[9];
      if (lane == 0) { TB_STAMP_MAX(level, 10); }
#pragma unroll
      for (int k = 0; k < 9; ++k) tot[k] = s_tot[k];
      mean_cov_from_sums(tot, n, mean, cov);
      madicp_host::eig3_sym(cov, w, V);
      if (lane == 0) { TB_STAMP_MAX(level, 11); }
      if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < 3; ++k) s_mean[k] = mean[k];
#pragma unroll
        for (int k = 0; k < 9; ++k) s_V[k] = V[k];
        if (cm.chunk == 0) {
#pragma unroll
          for (int k = 0; k < 3; ++k) { nd.mean[k] = mean[k]; nd.dir[k] = V[3 * k + 2]; nd.col0[k] = V[3 * k]; }
        }
      }
    }
    __syncthreads();
    double mean[3], V[9];
#pragma unroll
    for (int k = 0; k < 3; ++k) mean[k] = s_mean[k];
#pragma unroll
    for (int k = 0; k < 9; ++k) V[k] = s_V[k];
    if (lane == 0) { TB_STAMP_MAX(level, 12); }
    double lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
    // The sides of the chunk's points and, from them, the chunk's slice of the rank tables of the split's permutation
    // (common/split_order.h): the positions (relative to the node's first point) of the chunk's lefts, in order, from the
    // front of tab[cb, ce), of its rights from the back.  Point 256 u + t of the chunk belongs to thread t: its rank is
    // the lefts of the rows and waves in front of it (LDS) plus those of the lower lanes of its own ballot.
    unsigned int lbits = 0;
    int inw[8];
    {
      const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        double v[3] = {0, 0, 0};
        if (ok[u]) {
          eigen_coords(V, mean, x[u], y[u], z[u], v);
          minmax_update(lo, hi, v);
        }
        const bool left = ok[u] && v[2] < 0.0;
        const unsigned long long lm = __ballot(left);
        inw[u] = __popcll(lm & lt);
        lbits |= (left ? 1u : 0u) << u;
        if (lane == 0) s_cnt[u][wv] = __popcll(lm);
      }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      lo[a] = wave_min_keep(lo[a]);
      hi[a] = wave_max_keep(hi[a]);
    }
    if (lane == 0) {
#pragma unroll
      for (int a = 0; a < 3; ++a) { s_lo[wv][a] = lo[a]; s_hi[wv][a] = hi[a]; }
    }
    if (lane == 0) { TB_STAMP_MAX(level, 13); }
    __syncthreads();
    int run = 0;
    {
      int32_t* __restrict__ tab = P.tab;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        int pre = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          if (w == wv) pre = run;
          run += s_cnt[u][w];
        }
        const int q = 256 * u + (int)threadIdx.x;  // position in the chunk
        if (cb + q < ce) {
          const int rank = pre + inw[u];
          const int pnode = cm.chunk * kChunk + q;
          tab[((lbits >> u) & 1u) ? (long)cb + rank : (long)ce - 1 - (q - rank)] = pnode;
        }
      }
    }
    if (threadIdx.x == 0) {
      double L[3], H[3];
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        L[a] = s_lo[0][a]; H[a] = s_hi[0][a];
#pragma unroll
        for (int k = 1; k < 4; ++k) {
          if (s_lo[k][a] < L[a]) L[a] = s_lo[k][a];
          if (H[a] < s_hi[k][a]) H[a] = s_hi[k][a];
        }
        P.part2[(long)slot * 8 + a] = L[a];
        P.part2[(long)slot * 8 + 3 + a] = H[a];
      }
      P.part2[(long)slot * 8 + 6] = (double)run;  // the chunk's lefts
    }
    if (lane == 0) { TB_STAMP_MAX(level, 14); }
    __syncthreads();
  }
}

// C3: leaf test, children, and the scatter of the chunk — every point to the place the reference's `split`
// (utils.h:37-52) would have left it in (common/split_order.h).  A point that needs a rank-table entry finds the chunk
// that holds it by a search over the exclusive prefix of the node's per-chunk left counts (LDS), then reads the entry that
// chunk's tb_chip_stats wrote.
constexpr int kPrefMax = 4096;  // prefix entries held in LDS; a node with more chunks is searched coarsely, then walked
__global__ __launch_bounds__(256) void tb_chip_scatter(const Params P, int level) {
  __shared__ int s_off[kMaxBig + 1];
  __shared__ double s_lo[3], s_hi[3];
  __shared__ double s_p2[256][7];
  __shared__ int s_pref[kPrefMax + 1];
  __shared__ int s_before, s_carry;
  __shared__ int s_wsum[4];
  __shared__ double s_cs[4][18];
  const int cnt = chip_prefix(P, level, s_off);
  const int total = s_off[cnt];
  const double* __restrict__ in = level_in(P, level);
  double* __restrict__ out = level_out(P, level);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) { TB_STAMP_MIN(level, 0); }
  for (int slot = blockIdx.x; slot < total; slot += gridDim.x) {
    const ChunkMap cm = chip_find(s_off, cnt, slot);
    const int id = level_big(P, level)[cm.node_slot];
    BNode& nd = P.nodes[id];
    const int b = nd.begin, e = nd.end, n = e - b;
    if (lane == 0) { TB_STAMP_MAX(level, 2); }
    // thread t owns points cb + 8 t .. cb + 8 t + 7 (consecutive: one scan gives every point its count of lefts in front).
    // Requested first: nothing below depends on them until the sides are needed, and their first touch passes meanwhile.
    const int cb = b + cm.chunk * kChunk, ce = min(cb + kChunk, e);
    const int i0 = cb + 8 * (int)threadIdx.x;
    double px[8], py[8], pz[8];
    int nvalid = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int i = i0 + k;
      const bool ok = i < ce;
      const long j = ok ? i : cb;
      px[k] = in[3 * j]; py[k] = in[3 * j + 1]; pz[k] = in[3 * j + 2];
      nvalid += ok ? 1 : 0;
    }
    int shift = 0;
    // (the ROUNDED-UP granule count must fit: 8 193 chunks at shift 1 are 4 097 granules, and s_pref[n_gran] is written below)
    while (((cm.n_chunks + (1 << shift) - 1) >> shift) > kPrefMax) ++shift;
    const int n_gran = (cm.n_chunks + (1 << shift) - 1) >> shift;
    if (threadIdx.x < 3) { s_lo[threadIdx.x] = 0.0; s_hi[threadIdx.x] = 0.0; }
    if (threadIdx.x == 3) { s_before = 0; s_carry = 0; }
    __syncthreads();
    for (int base = 0; base < cm.n_chunks; base += 256) {  // the node's chunks, 256 at a time
      const int c = base + threadIdx.x;
      const bool on = c < cm.n_chunks;
      if (on) {
#pragma unroll
        for (int k = 0; k < 7; ++k) s_p2[threadIdx.x][k] = P.part2[(long)(cm.first_slot + c) * 8 + k];
      }
      const int v = on ? (int)s_p2[threadIdx.x][6] : 0;
      int incl = v;  // exclusive prefix of the left counts: wave scans + the four wave totals + what came before
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(incl, d, 64);
        if (lane >= d) incl += o;
      }
      if (lane == 63) s_wsum[wv] = incl;
      __syncthreads();
      int off = s_carry;
      for (int k = 0; k < wv; ++k) off += s_wsum[k];
      const int excl = off + incl - v;
      if (on && (c & ((1 << shift) - 1)) == 0) s_pref[c >> shift] = excl;
      if (on && c == cm.chunk) s_before = excl;
      const int m = min(256, cm.n_chunks - base);
      if (threadIdx.x < 3) {
        const int a = threadIdx.x;
        double L = s_lo[a], H = s_hi[a];
        for (int c2 = 0; c2 < m; ++c2) {
          const double l = s_p2[c2][a], h = s_p2[c2][3 + a];
          if (l < L) L = l;
          if (H < h) H = h;
        }
        s_lo[a] = L;
        s_hi[a] = H;
      }
      __syncthreads();
      if (threadIdx.x == 0) s_carry = s_carry + s_wsum[0] + s_wsum[1] + s_wsum[2] + s_wsum[3];
      __syncthreads();
    }
    if (threadIdx.x == 0) s_pref[n_gran] = s_carry;
    __syncthreads();
    const double ext0 = s_hi[0] - s_lo[0], ext2 = s_hi[2] - s_lo[2];
    const int nl = s_carry, before = s_before;
    if (lane == 0) { TB_STAMP_MAX(level, 3); }
    const bool leaf = (ext2 < P.b_max) || nl == 0 || nl == n;
    const int mid = b + nl;
    double mean[3], col2[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { mean[k] = nd.mean[k]; col2[k] = nd.dir[k]; }
    __syncthreads();  // (everybody has read the node and the shared results before chunk 0 rewrites parts of the node)
    if (leaf) {
      if (cm.chunk == 0 && threadIdx.x == 0) {  // rare: finished by the wave regime of the next level (nearest member)
        nd.bbox0 = ext0;
        nd.flags |= kLeafPending;
        const int step = max(level + 1, P.first_step);
        level_q(P, step)[atomicAdd(&P.st->q_count[step].v, 1)] = make_int4(id, b, e, level);  // (its points stay at `level`)
      }
      continue;
    }
    unsigned int lmask = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (k < nvalid && goes_left(mean, col2, px[k], py[k], pz[k])) lmask |= 1u << k;
    const int mine = __popc(lmask);
    // exclusive scan of `mine` over the 256 threads: wave scan + wave totals
    int incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int o = __shfl_up(incl, d, 64);
      if (lane >= d) incl += o;
    }
    if (lane == 63) s_wsum[wv] = incl;
    __syncthreads();
    int wave_off = 0;
    for (int k = 0; k < wv; ++k) wave_off += s_wsum[k];
    int lrun = before + wave_off + incl - mine;  // lefts of the NODE in front of this thread's first point
    if (lane == 0) { TB_STAMP_MAX(level, 4); }
    const int32_t* __restrict__ tab = P.tab;
    const double* __restrict__ p2 = P.part2 + (long)cm.first_slot * 8;
    auto lefts_of = [p2](int c) { return (int)p2[(long)c * 8 + 6]; };
    // The plan: every point's place from the closed form; for the two kinds that need a rank table the table entry is
    // REQUESTED here (a dependent first touch of memory tb_chip_stats wrote) and used after the children's sums below,
    // which do not depend on it — the round trip passes under ~2 us of shuffles.
    int dst[8];
    long tix[8];
    int tadd[8];
    madicp_host::ChunkCache rcache, lcache;  // the last chunk found for a right / a left rank
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      dst[k] = 0;
      tix[k] = -1;
      tadd[k] = 0;
      if (k < nvalid) {
        const bool left = (lmask >> k) & 1u;
        const madicp_host::SplitPlan pl = madicp_host::split_plan(left, cm.chunk * kChunk + 8 * (int)threadIdx.x + k, lrun, nl, n);
        dst[k] = pl.idx;
        // (a thread's eight points are consecutive, so are the ranks they ask for: the chunk found for one usually holds
        // the next — common/split_order.h remembers it; eight serial binary searches in LDS were 2.3 us of this kernel)
        if (pl.kind == 1) {
          int c, local;
          madicp_host::find_right_chunk_cached(rcache, s_pref, n_gran, shift, cm.n_chunks, kChunk, n, pl.idx, lefts_of, c, local);
          tix[k] = (long)b + min(n, (c + 1) * kChunk) - 1 - local;
        } else if (pl.kind == 2) {
          int c, local;
          madicp_host::find_left_chunk_cached(lcache, s_pref, n_gran, shift, cm.n_chunks, pl.idx, lefts_of, c, local);
          tix[k] = (long)b + (long)c * kChunk + local;
          tadd[k] = -1;
        }
        lrun += left ? 1 : 0;
      }
    }
    int tval[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) tval[k] = tab[tix[k] >= 0 ? tix[k] : (long)b];
    if (lane == 0) { TB_STAMP_MAX(level, 5); }
    {  // the children's sums over this chunk (they will not have to sweep their points for them)
      double sL[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, sR[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (k < nvalid) {
          if ((lmask >> k) & 1u) add_point(sL, px[k], py[k], pz[k]); else add_point(sR, px[k], py[k], pz[k]);
        }
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        sL[k] = wave_sum(sL[k]);
        sR[k] = wave_sum(sR[k]);
      }
      if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 9; ++k) { s_cs[wv][k] = sL[k]; s_cs[wv][9 + k] = sR[k]; }
      }
      __syncthreads();
      if (threadIdx.x < 18)
        level_part(P, level + 1)[(long)slot * 18 + threadIdx.x] =
            ((s_cs[0][threadIdx.x] + s_cs[1][threadIdx.x]) + s_cs[2][threadIdx.x]) + s_cs[3][threadIdx.x];
    }
    if (lane == 0) { TB_STAMP_MAX(level, 6); }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (k < nvalid) {
        const long d = (long)b + (tix[k] >= 0 ? tval[k] + tadd[k] : dst[k]);
        out[3 * d] = px[k]; out[3 * d + 1] = py[k]; out[3 * d + 2] = pz[k];
      }
    }
    if (lane == 0) { TB_STAMP_MAX(level, 7); }
    // the node's record and its children (chunk 0, thread 0): last, so that the atomic's round trip and ~100 stores do not hold
    // up the workgroup's barriers above — only the NEXT kernel reads any of it
    if (cm.chunk == 0 && threadIdx.x == 0) {
      const double col0[3] = {nd.col0[0], nd.col0[1], nd.col0[2]};
      const int c = atomicAdd(&P.st->n_nodes.v, 2);
      if (c + 2 > P.node_cap) {
        P.st->n_nodes.error = 1;
      } else {
        const Inherit inh = load_inherit(nd, level);
        nd.child = c;
        make_child(P.nodes[c], inh, id, col0, ext0, n, P.b_min, b, mid, true);
        make_child(P.nodes[c + 1], inh, id, col0, ext0, n, P.b_min, mid, e, false);
        // their sums: this node's per-chunk partials, written above by every chunk of it
        P.nodes[c].flags |= kChunkSums;
        P.nodes[c + 1].flags |= kChunkSums | kSumRight;
        P.nodes[c].sum_first = P.nodes[c + 1].sum_first = cm.first_slot;
        P.nodes[c].sum_n = P.nodes[c + 1].sum_n = cm.n_chunks;
        enqueue_single(P, c, b, mid, level + 1);
        enqueue_single(P, c + 1, mid, e, level + 1);
      }
      nd.bbox0 = ext0;
      nd.mid = mid;
      nd.flags |= kDone;
    }
    if (lane == 0) { TB_STAMP_MAX(level, 1); }
    __syncthreads();
  }
}

// ---- exclusive scan of the leaf-start marks (1024 elements per workgroup) ----------------------------------------
constexpr int kScanTile = 1024;
__device__ __forceinline__ void scan_tiles_body(const uint32_t* __restrict__ marks, int n, uint32_t* __restrict__ tile_sums, int block) {
  __shared__ uint32_t s_w[4];
  const int base = block * kScanTile + threadIdx.x * 4;
  uint32_t v = 0;
  #pragma unroll
  for (int k = 0; k < 4; ++k)
    if (base + k < n) v += marks[base + k];
  #pragma unroll
  for (int m = 32; m > 0; m >>= 1) v += __shfl_xor(v, m, 64);
  if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) tile_sums[block] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}
__global__ __launch_bounds__(256) void tb_scan_tiles(const uint32_t* __restrict__ marks, int n, uint32_t* __restrict__ tile_sums) {
  scan_tiles_body(marks, n, tile_sums, blockIdx.x);
}
// one workgroup: exclusive scan of the tile sums in place (any count), the grand total into *out_total
__global__ __launch_bounds__(256) void tb_scan_top(uint32_t* __restrict__ tile_sums, int n_tiles, int32_t* __restrict__ out_total) {
  __shared__ uint32_t s_w[4];
  __shared__ uint32_t s_carry;
  if (threadIdx.x == 0) s_carry = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int base = 0; base < n_tiles; base += 256) {
    const int i = base + threadIdx.x;
    const uint32_t v = i < n_tiles ? tile_sums[i] : 0u;
    uint32_t incl = v;
    #pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t o = __shfl_up(incl, d, 64);
      if (lane >= d) incl += o;
    }
    if (lane == 63) s_w[wv] = incl;
    __syncthreads();
    uint32_t off = s_carry;
    for (int k = 0; k < wv; ++k) off += s_w[k];
    if (i < n_tiles) tile_sums[i] = off + incl - v;
    __syncthreads();
    if (threadIdx.x == 255) s_carry = off + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) *out_total = (int32_t)s_carry;
}
// S[i] = marks before i, for i in [0, n]; tile_offset = marks before this tile.  Returns (to every thread) the marks of the tile.
__device__ __forceinline__ uint32_t scan_apply_body(const uint32_t* __restrict__ marks, int n, uint32_t tile_offset, uint32_t* __restrict__ S,
                                                    int block) {
  __shared__ uint32_t s_w[4];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int base = block * kScanTile + threadIdx.x * 4;
  uint32_t m[4], v = 0;
  #pragma unroll
  for (int k = 0; k < 4; ++k) {
    m[k] = (base + k < n) ? marks[base + k] : 0u;
    v += m[k];
  }
  uint32_t incl = v;
  #pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t o = __shfl_up(incl, d, 64);
    if (lane >= d) incl += o;
  }
  if (lane == 63) s_w[wv] = incl;
  __syncthreads();
  uint32_t off = tile_offset;
  for (int k = 0; k < wv; ++k) off += s_w[k];
  uint32_t run = off + incl - v;
  #pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (base + k <= n) S[base + k] = run;
    run += m[k];
  }
  return s_w[0] + s_w[1] + s_w[2] + s_w[3];
}
__global__ __launch_bounds__(256) void tb_scan_apply(const uint32_t* __restrict__ marks, int n, const uint32_t* __restrict__ tile_sums,
                                                     uint32_t* __restrict__ S) {
  scan_apply_body(marks, n, tile_sums[blockIdx.x], S, blockIdx.x);
}

// ---- what the host needs before it can size the tree: root mean, rho, size of the LDS-staged top -----------------
__device__ __forceinline__ void summary_body(const Params& P, int top_levels, int block, int n_blocks) {
  State* st = P.st;
  __shared__ double s_r[4];
  __shared__ int s_tops[4], s_lvl[4], s_valid[4];
  const int n = min(st->n_nodes.v, P.node_cap);
  const double o0 = P.nodes[0].mean[0], o1 = P.nodes[0].mean[1], o2 = P.nodes[0].mean[2];
  double r = 0.0;
  int tops = 0, lvl = 0, valid = 0;
  for (int i = block * blockDim.x + threadIdx.x; i < n; i += n_blocks * blockDim.x) {
    const BNode& nd = P.nodes[i];
    lvl = max(lvl, nd.level);
    valid += (nd.flags & kDone) ? 1 : 0;
    if (nd.flags & kLeaf) continue;
    const double e0 = nd.mean[0] - o0, e1 = nd.mean[1] - o1, e2 = nd.mean[2] - o2;
    const double d = sqrt((e0 * e0 + e1 * e1) + e2 * e2);
    if (d > r && d < 1.7976931348623157e308) r = d;  // finite only (validate_nodes' rule)
    if (nd.level < top_levels) ++tops;
  }
  r = wave_max_keep(r);
  tops = wave_sum_int(tops);
  valid = wave_sum_int(valid);
#pragma unroll
  for (int m = 32; m > 0; m >>= 1) lvl = max(lvl, __shfl_xor(lvl, m, 64));
  const int wv = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { s_r[wv] = r; s_tops[wv] = tops; s_lvl[wv] = lvl; s_valid[wv] = valid; }
  __syncthreads();
  if (threadIdx.x == 0) {  // one set of atomics per workgroup (one per wave on one line was 48 us of queueing)
    double R = s_r[0];
    int T = 0, Lv = 0, Vd = 0;
    for (int k = 0; k < 4; ++k) {
      if (s_r[k] > R) R = s_r[k];
      T += s_tops[k];
      Lv = max(Lv, s_lvl[k]);
      Vd += s_valid[k];
    }
    if (R > 0.0) atomicMax(&st->rho_bits, (unsigned long long)__double_as_longlong(R));
    if (T) atomicAdd(&st->n_top, T);
    if (Lv) atomicMax(&st->max_level, Lv);
    if (Vd) atomicAdd(&st->n_valid, Vd);
    if (block == 0) {
      st->origin[0] = o0; st->origin[1] = o1; st->origin[2] = o2;
    }
  }
}
__global__ __launch_bounds__(256) void tb_summary(const Params P, int top_levels) { summary_body(P, top_levels, blockIdx.x, gridDim.x); }

// The end of a build in two launches (clouds of up to kScanDirectMax tiles).  First: the tile sums of the leaf-start marks
// by the first n_tiles workgroups, the summary by the others.  Second: the scan proper — every workgroup adds up the tile
// sums in front of its own, no third kernel for that — and the last workgroup publishes what the host is waiting for in
// its pinned block, sequence number last (system-scope release): the host polls it, no copy command, no event.
constexpr int kScanDirectMax = 4096;
constexpr int kNeedSteps = 32;
struct HostLine {
  int32_t n_nodes, error, n_leaves, n_top, max_level, n_valid;
  int32_t pending_wave, pending_quad;  // queue counts of the first step that was not launched
  unsigned long long rho_bits;
  double origin[3];
  int32_t seq, pad_;
  int32_t need[kNeedSteps];  // workgroups step s of THIS build had work for (team nodes + wave nodes / 4 + quad nodes / 64): the next
                             // build of a similar cloud sizes its launches from it (a workgroup that finds nothing still costs its dispatch)
};
__global__ __launch_bounds__(256) void tb_finish_a(const Params P, int top_levels, int n_tiles) {
  if ((int)blockIdx.x < n_tiles)
    scan_tiles_body(P.leaf_start, P.n_points, P.tile_sums, blockIdx.x);
  else
    summary_body(P, top_levels, (int)blockIdx.x - n_tiles, (int)gridDim.x - n_tiles);
}
__global__ __launch_bounds__(256) void tb_finish_b(const Params P, int n_tiles, int next_step, HostLine* __restrict__ host, int seq) {
  __shared__ uint32_t s_part[4];
  uint32_t before = 0;
  for (int i = threadIdx.x; i < (int)blockIdx.x; i += blockDim.x) before += P.tile_sums[i];
  #pragma unroll
  for (int m = 32; m > 0; m >>= 1) before += __shfl_xor(before, m, 64);
  if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = before;
  __syncthreads();
  before = s_part[0] + s_part[1] + s_part[2] + s_part[3];
  __syncthreads();
  const uint32_t own = scan_apply_body(P.leaf_start, P.n_points, before, P.S, blockIdx.x);
  if ((int)blockIdx.x == n_tiles - 1) {  // (workgroup-uniform)
    if (threadIdx.x < kNeedSteps) {
      const State* st = P.st;
      const int t = threadIdx.x;
      host->need[t] = st->team_count[t].v + (st->q_count[t].v + 3) / 4 + (st->small_count[t].v + 63) / 64;
    }
    __threadfence_system();
    __syncthreads();
  }
  if ((int)blockIdx.x == n_tiles - 1 && threadIdx.x == 0) {
    State* st = P.st;
    st->n_leaves = (int32_t)(before + own);
    host->n_nodes = st->n_nodes.v;
    host->error = st->n_nodes.error;
    host->n_leaves = (int32_t)(before + own);
    host->n_top = st->n_top;
    host->max_level = st->max_level;
    host->n_valid = st->n_valid;
    host->pending_wave = st->q_count[next_step].v + st->team_count[next_step].v;
    host->pending_quad = st->small_count[next_step].v;
    host->rho_bits = st->rho_bits;
    host->origin[0] = st->origin[0]; host->origin[1] = st->origin[1]; host->origin[2] = st->origin[2];
    __hip_atomic_store(&host->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// ---- emission: temporary nodes -> the DFS-preorder madicp_node array, and everything derived from it -------------------
// One launch writes what used to take four (tb_emit, tree_compact, tb_layout_top, tree_compact_top: 41 us of the 80 us a
// build spent behind its last level): the leaf scan S answers every question the later kernels asked the emitted array —
// a node's `right` offset is 2 (S[mid] - S[begin]), a child is a leaf iff its range holds one leaf, a leaf's ordinal is
// S[begin] — so the screening record / dense leaf record of a node is made by the thread that emits it, and the records of
// the LDS-staged top by the workgroups behind the node range, one thread per entry of the breadth-first order top_bfs_body left.
__device__ __forceinline__ madicp_node emit_node(const BNode& nd, const uint32_t* __restrict__ S, int sb) {
  madicp_node o;
#pragma unroll
  for (int k = 0; k < 3; ++k) { o.mean[k] = nd.mean[k]; o.dir[k] = nd.dir[k]; }
  o.bbox0 = nd.bbox0;
  if (nd.flags & kLeaf) {
    o.right = 0;
    o.leaf_id = sb;
  } else {
    o.right = 2 * ((int)S[nd.mid] - sb);
    o.leaf_id = -1;
  }
  return o;
}
// leaf_cap > 0: the launch was enqueued BEHIND the summary without the host in between, into a tree block sized for leaf_cap
// leaves (the previous scan's count with head-room): the node count, the size of the top and the origin are then read from
// the State on the device (n_nodes / n_top / o0-o2 as passed are the block's capacities / unused), and a tree that does not
// fit — or a build that went wrong — writes nothing: the host sees the same in its line and emits again into a block of
// the right size.  What that saves is the host round trip between the summary and the emission (10-11 us of idle device).
__global__ __launch_bounds__(256) void tb_emit(const Params P, int n_nodes, madicp_node* __restrict__ out, CNode* __restrict__ cnodes,
                                               LeafRec* __restrict__ leaves, int n_top, int* __restrict__ dfs,
                                               unsigned int* __restrict__ link, int4* __restrict__ exits, CNode* __restrict__ top,
                                               double o0, double o1, double o2, int leaf_cap) {
  const uint32_t* __restrict__ S = P.S;
  const int node_blocks = (n_nodes + 255) / 256;  // (of the capacity when leaf_cap > 0: the launch geometry)
  if (leaf_cap > 0) {
    const State* st = P.st;
    const int nl = st->n_leaves;
    if (nl < 1 || nl > leaf_cap || st->n_nodes.error != 0) return;
    n_nodes = 2 * nl - 1;
    n_top = min(min(st->n_top, n_top), n_nodes);
    o0 = st->origin[0]; o1 = st->origin[1]; o2 = st->origin[2];
  }
  if ((int)blockIdx.x >= node_blocks) {  // ---- the staged top: entry e of the breadth-first order
    const int e = ((int)blockIdx.x - node_blocks) * 256 + (int)threadIdx.x;
    if (e >= n_top) return;
    const int id = P.top_ids[e];
    if (id < 0 || id >= n_nodes) return;
    const BNode& nd = P.nodes[id];
    const int sb = (int)S[nd.begin];
    const int i = 2 * sb + nd.left_turns;
    const madicp_node o = emit_node(nd, S, sb);
    CNode c;
    make_record(o, o0, o1, o2, c);
    c.right = P.top_link[e];
    top[e] = c;
    dfs[e] = i;
    link[e] = P.top_link[e];
    exits[e] = make_int4(i + 1, o.right >> 1, sb, nd.level);
    return;
  }
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= n_nodes) return;
  const BNode& nd = P.nodes[t];
  const int sb = (int)S[nd.begin];
  const int idx = 2 * sb + nd.left_turns;
  if (idx < 0 || idx >= n_nodes) return;
  const madicp_node o = emit_node(nd, S, sb);
  out[idx] = o;
  CNode c;
  c.npack = 0ull;
  c.c = 0.f;
  c.right = 0u;  // (leaf records are never read, except the root's in a single-node tree: tree_compact)
  if (o.right != 0) {
    make_record(o, o0, o1, o2, c);
    const int sm = (int)S[nd.mid];
    c.right = (unsigned int)o.right | ((sm - sb) == 1 ? kLeftLeaf : 0u) | (((int)S[nd.end] - sm) == 1 ? kRightLeaf : 0u);
  } else {
    LeafRec lr;
#pragma unroll
    for (int a = 0; a < 3; ++a) { lr.mean[a] = o.mean[a]; lr.normal[a] = o.dir[a]; }
    lr.bbox0 = o.bbox0;
    lr.pad_ = 0.0;
    leaves[sb] = lr;
  }
  cnodes[idx] = c;
}

// ---- diagnostics: the cloud in the order the construction left it (madicp_debug_tree_build_points) --------------------
// A leaf's members stay where the split of its parent put them — range [begin, end) of the point buffer its level reads.
// One wavefront per temporary node; internal nodes have nothing to copy.
__global__ __launch_bounds__(256) void tb_debug_order(const Params P, int n_nodes, double* __restrict__ out) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= n_nodes) return;
  const BNode& nd = P.nodes[i];
  if (!(nd.flags & kLeaf)) return;
  const double* __restrict__ in = level_in(P, nd.level);
  for (long j = 3 * (long)nd.begin + lane; j < 3 * (long)nd.end; j += 64) out[j] = in[j];
}


}  // namespace tb
}  // namespace madicp
