// Device code for the MAD-ICP hot path on gfx950 (MI355X).  Single translation unit: included by
// madicp_capi.hip only.
//
// Kernels (each cites the reference function it replaces; paths relative to the reference repo):
//   moving_prep       : per moving leaf, cache |p| for the gate        (mad_icp.cpp:81, `moving->mean_.norm()`)
//   nn_descend        : batched MADtree::bestMatchingLeafFast           (mad_tree.cpp:144-152, mad_tree_wrapper.h:48-67)
//   tree_transform    : MADtree::applyTransform                         (mad_tree.cpp:165-172)
//   icp_round         : one Gauss-Newton round: [adder join + LDLT + expSO3 + pose update of the PREVIOUS round
//                       (mad_icp.cpp:105-117), redundantly per workgroup] + MADicp::update over K trees at the new pose,
//                       fused transform -> descent -> gate -> e,J -> weight -> wave/block reduction of (H,b)
//                       (mad_icp.cpp:59-103 under pipeline.cpp:180-183)
//   icp_final         : join + solve of the last round, matched-leaf count (pipeline.cpp:195-204)
//   icp_reduce        : this rank's partials -> totals, in front of the RCCL all-reduce (multi-GPU)
//
// Numerics contract: IEEE fp64, the reference's (Eigen's) operation order, NO FMA contraction — every
// branch decision (descent side test, gate) is bit-identical to the CPU path.  Enforced by the pragma
// below and by -ffp-contract=off on the command line.
//
// Why no MFMA: nothing here is a dense contraction.  Per (leaf, tree) pair the work is a ~14-step
// dependent pointer chase followed by ~150 flops; the 6x6 accumulation is a reduction over pairs.
// What bounds it (measured: DESIGN.md 3.1) is the chain of dependent steps at the 3 waves/SIMD the
// accumulators allow — the L1 address path of 64-lane gathers while walking, instruction issue and
// barriers otherwise — not HBM.  So the levers are: a 16-byte screening record per visit (one gather
// instead of four, exactness certified per decision), the top of the tree in LDS, dense 64 B leaf
// records, DFS-preorder storage so spatially coherent queries walk contiguous memory, an XCD-aware
// unit -> workgroup map so each XCD's private L2 only sees its own trees, correspondences reused across
// rounds when a margin proves them unchanged, the solve of a round fused into the next round's kernel,
// and a deterministic register -> wave shuffle -> LDS -> partial reduction.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "madicp_hip.h"

#pragma clang fp contract(off)

namespace madicp {

constexpr int kBlock = 768;          // threads per icp_round workgroup = 12 wave64 = 3 per SIMD: ONE workgroup per CU
constexpr int kWaves = kBlock / 64;
constexpr int kAcc = 30;             // 21 (lower triangle of H, column by column) + 6 (b) + accepted pairs + nodes visited
                                     // (the reference's count: cached depths included) + nodes actually walked this round;
                                     // 30 doubles = 240 B: partial rows are 16-byte aligned


// 16-byte screening record of one node, same index as the exact 64-byte madicp_node.
//
// The descent test of the reference, s = fl((q-m).n) < 0 (mad_tree.cpp:148), needs 52 bytes of fp64 per
// visit = four 16-byte loads per lane, and with 64 lanes walking 64 different nodes it is the L1/TA
// address path, not arithmetic or HBM, that bounds the kernel.  The screening record lets almost every
// visit decide with ONE 16-byte load:
//     n~  = the split normal rounded to 3 x 21-bit fixed point (|n~_i - n_i| <= 2^-20)
//     c~  = fl32( n~ . (m - o) )          o = the tree's origin (root centroid)
//     s^  = n~ . (q - o) - c~             evaluated in fp64
// For every query that can reach the node,  |s^ - s| <= E := (2^-20 + 32u)(|q-o|_1 + rho) + 2^-24 |c~|
// (u = 2^-53, rho >= |m-o|_1 for every node of the tree; derivation in DESIGN.md "Exact screening").
// If |s^| > E the sign of s^ IS the sign of the reference's fp64 s; otherwise the lane loads the exact
// record and evaluates the reference expression.  The decision is therefore bit-identical by construction
// (and the parity tests check it against the oracle), while ~99.9 % of the visits cost a quarter of the
// L1 transactions and the screening array (16 B x nodes) of a keyframe pair fits an XCD's 4 MiB L2.
struct CNode {
  unsigned long long npack;  // n~: k0 | k1 << 21 | k2 << 42, k_i two's complement 21 bit, n~_i = k_i * 2^-20
  float c;                   // c~ ; +inf marks "always take the exact path" (non-finite node)
  unsigned int right;        // bits 0..29: madicp_node::right (DFS offset of the right child); bit 30: the left
                             // child is a leaf; bit 31: the right child is a leaf
};
constexpr unsigned int kRightMask = 0x3fffffffu, kLeftLeaf = 0x40000000u, kRightLeaf = 0x80000000u;
static_assert(sizeof(CNode) == 16, "screening record is 16 bytes");

// Storage: record i belongs to node i (DFS preorder, like the exact array), so the records of a sub-tree are
// contiguous and the spatially sorted queries of a wave touch a compact address range.  A parent's record says
// which of its children are leaves, so leaf records are never read.  (A cache-line-blocked layout — 3-level
// sub-trees per 128-byte line — was measured slower: it shortens one query's chain of dependent misses but
// scatters the lines that NEIGHBOURING queries share, and that sharing is what the L1/L2 hit rate lives on.)

constexpr double kScreenDelta = 9.6e-7;   // > 2^-20 + 32u, covers the rounding of |q-o|_1 and rho too
constexpr double kScreenC = 6.0e-8;       // > 2^-24 (1 + 2^-23)

// Dense per-leaf record, indexed by the leaf's getLeafs() ordinal: everything MADicp::update reads of the matched
// leaf (mean_, eigenvectors_.col(0), bbox_(0): mad_icp.cpp:64-65,81,97).  The leaves are half of the nodes of the
// exact array and are interleaved with internal nodes there, so fetching them from it touches twice the cache lines
// and — measured — the scattered, L2-missing leaf fetch cost more than the whole descent.  The ordinal needs no
// lookup: a sub-tree with s nodes has (s+1)/2 leaves, so going right at a node adds right/2 to the running ordinal.
struct LeafRec {
  double mean[3];
  double normal[3];
  double bbox0;
  double pad_;
};
static_assert(sizeof(LeafRec) == 64, "leaf record is one 64-byte line");

// what a kernel needs to walk one tree; held by value in the Job / passed as a kernel argument so that no
// dependent pointer chase precedes the first node load
struct TreeDesc {
  const madicp_node* nodes;
  const CNode* cnodes;   // screening records, same indexing as nodes
  const LeafRec* leaves; // dense leaf records, indexed by leaf ordinal
  double origin[3];      // o: mean of node 0
  double rho;            // >= |m - o|_1 for every internal node (sqrt(3) * max |m - o|_2)
  // the hot top of the tree, staged into LDS by icp_round (see "LDS-staged top levels" below)
  const CNode* top;      // n_top records, breadth-first over the first kTopLevels levels (internal nodes only)
  const int4* top_exit;  // per top entry: node index of its left child, leaves of its left sub-tree, leaf ordinal of its
                         // left-most leaf, its level (see "LDS-staged top levels")
  const int* top_dfs;    // per top entry: its own node index (only the exact-path fallback reads it)
  int32_t n_top;
  int32_t slot;          // in a Job: the caller's index of this tree (the Job lists its trees dealt over the XCD pieces, fill_job)
};

// LDS-staged top levels.  The descent is bound by the L1 address path: a 64-lane gather of 16-byte records costs
// ~80 cycles of the CU's vector-memory pipe however hot the lines are (measured: a second, cache-warm pass over the
// same queries costs as much as the first).  LDS serves the same gather in ~15 cycles.  Each workgroup works on ONE
// tree, so it copies the first kTopLevels levels of that tree's screening records (<= 2047 x 16 B, plus a 16-byte exit
// record each = 64 KiB; icp_round runs one workgroup per CU) into LDS once and walks them there; only the last few
// levels and the leaf record come from L1/L2.  A top entry's `right` word is re-purposed (the walk of these levels is
// bound by the instructions it issues, so the word is laid out for the fewest of them):
//   bits 0..10 the top entry of the LEFT child | bits 11..21 the top entry of the RIGHT child (kTopNone: that child is
//   not in the top array: a leaf, or below the staged levels) | bit 22 left child is a leaf | bit 23 right child is a leaf
// and its `top_exit` record holds what only the LAST staged step of a walk needs: x = node index of the left child (the
// right child's is x + 2y - 1), y = leaves of the left sub-tree, z = leaf ordinal of the entry's left-most leaf (so the
// running ordinal is not carried through these levels: it is z, plus y when the walk leaves to the right), w = the
// entry's level (the depth is not counted either: it is w + 1).
constexpr int kTopLevels = 11;
constexpr int kTopMax = 2048;
constexpr int kTopLdsBytes = kTopMax * (16 + 16);
constexpr unsigned int kTopNone = 0x7ffu, kTopLeftLeaf = 1u << 22, kTopRightLeaf = 1u << 23;
__host__ __device__ __forceinline__ unsigned int top_link_word(int left_entry, int right_entry, bool l_leaf, bool r_leaf) {
  return (left_entry < 0 ? kTopNone : (unsigned int)left_entry) | ((right_entry < 0 ? kTopNone : (unsigned int)right_entry) << 11) |
         (l_leaf ? kTopLeftLeaf : 0u) | (r_leaf ? kTopRightLeaf : 0u);
}
static_assert(kTopMax - 1 <= (int)kTopNone, "top entries are addressed with 11 bits, the last value means 'none'");

// What one registration hands back, written by icp_final straight into a pinned host block (no D2H copy operation
// on any stream): the caller waits for the event behind the registration's last kernel and reads it.
struct HostResult {
  double X[12];   // final pose (R row-major, t)
  double H[36];   // H_adder_ of the last round (pipeline.cpp:223 reads it)
  double b[6];
  double n_pairs;
  unsigned long long visits;
  unsigned long long walked;
  int32_t n_matched;
  int32_t iter;
  int32_t seq;    // written LAST (system-scope release): Job::seq of the registration these results belong to
  int32_t error;  // Job::error (icp_persist: an in-launch wait ran out)
};

// Streamed registrations, results out (option "publish_side", default on).  icp_final writing its results straight into the
// caller's pinned block costs it two PCIe round trips — the stores' acknowledgement before the sequence number may go out,
// then the sequence number's own at the end of the kernel: 10 us instead of 4.9, on the stream the NEXT registration is
// waiting on.  Instead it leaves them in this device-resident outbox as tagged 16-byte granules (value halves + the
// registration's sequence number: the tag, not an ordering of stores, says that a value is there — no fence), and a
// one-workgroup kernel on a side stream (icp_publish, enqueued when the registration is submitted) waits for the tags and
// carries results and matched flags to the host while the compute stream is already running the next registration.
constexpr int kOutboxGranules = 64;
struct Outbox {
  unsigned long long g[2 * kOutboxGranules];  // granule i = words 2i, 2i + 1:  0..35 H, 36..47 X, 48..53 b, 54 n_pairs,
};                                             // 55 visits, 56 walked, 57 (n_matched | iter << 32), 58 error

// One registration in flight; lives in device memory, written by the host before each launch sequence
// and advanced by workgroup 0 of every icp_round / by icp_final.  Keeping every per-registration quantity behind this one pointer is what lets
// a single captured hipGraph serve every scan / keyframe set of the same launch geometry.
struct Job {
  const double* moving;  // (L,4): x y z |p|  (sensor frame)
  uint8_t* matched;      // (L) matched_ flags of the last round
  uint32_t* corr;        // optional (K,L) correspondence trace
  double* x_iters;       // optional (n_iters,12) pose before each round
  uint32_t* cache_leaf;  // optional (K,L): leaf node index | depth << 26 found at an earlier round (see "Correspondence reuse")
  float* cache_margin;   // optional (K,L): how far that leaf may still move before any side test on its path can flip
  int32_t L;
  int32_t K;
  int32_t n_iters;
  int32_t iter;
  int32_t flags;         // kFlagNoUpdate
  int32_t n_matched;
  unsigned long long visits;  // internal nodes visited (all rounds; exact: integer-valued doubles summed)
  unsigned long long walked;  // of those, the ones this registration really walked (the rest: correspondence reuse)
  double X[12];          // final pose (R row-major, t), written by icp_final
  double Xring[2][12];   // pose of round r lives in Xring[r & 1]; Xring[0] = initial guess (host)
  double wear_ring[2][2];  // [r & 1]: the cumulative wear coefficients (alpha, beta) of round r — see "Correspondence reuse"
  double min_ball, rho, b_ratio;
  double H[36];          // row-major, of the last round
  double b[6];
  double n_pairs;        // accepted (leaf,tree) pairs of the last round
  int32_t ranges_per_tree;  // launch geometry: every tree's moving leaves are cut into this many ranges
  int32_t stage_min_leaves; // stage a tree's top levels into LDS only for units with at least this many leaves
  int32_t lds_top;          // 1 when the launch carries kTopLdsBytes of dynamic LDS
  int32_t queue_nodes;      // DEEP launches: a round is done leaf-major when the workgroup walked FEWER than this many nodes per
                            // pass in the previous round (0: never; option "leaf_major")
  int32_t seq;              // streamed registrations: what icp_final leaves in HostResult::seq when everything is written
  uint32_t epoch;           // icp_persist: distinguishes this launch's exchange granules from every earlier launch's (host counter)
  uint32_t p2p_epoch;       // option "shard_p2p": the mailbox session's registration counter (the same on every rank): tags + slot parity
  int32_t error;            // icp_persist: non-zero when a bounded in-launch wait ran out (results invalid)
#ifdef MADICP_ABLATE
  unsigned long long* dbg;  // profiling builds: per-workgroup phase time stamps of the last launch
#endif
  HostResult* host_out;     // optional: pinned host block icp_final also writes the results to
  uint8_t* host_matched;    // optional: pinned host copy of the matched_ flags (L bytes, 16-byte aligned)
  Outbox* outbox;           // optional (instead of the two above): icp_final leaves the results here, icp_publish carries them out
  TreeDesc trees[MADICP_MAX_TREES];
};
constexpr int kFlagNoUpdate = 1;
constexpr int kFlagNoReuse = 512;  // never reuse a cached correspondence (option cache_correspondences = 0)
constexpr int kFlagNoGateReuse = 2048;  // never skip a pair on its cached gate slack (option cache_gate = 0)
constexpr int kFlagInterleave = 8192;   // a range is every RPT-th group of 64 leaves, not a contiguous stretch (option interleave_ranges)
constexpr int kFlagMatchAll = 1024;  // matched_ flags are the OR over ALL rounds (the host cleared them), not the last round's:
                                     // what the reference leaves behind when its realtime check ends the loop before
                                     // iteration MAX_ICP_ITS - 1, the only one that resets them (pipeline.cpp:167-176)

// ---------------------------------------------------------------------------------------------------
// fp64 helpers with the reference's evaluation order (see oracle/linalg.h for the derivation).
// ---------------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ double dotc(double a0, double a1, double a2, double b0, double b1, double b2) {
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
// (pointers that were themselves loaded from memory are "generic" to the compiler; the explicit global
// address space turns flat_load into global_load)
typedef double vd2 __attribute__((ext_vector_type(2)));
typedef double vd4 __attribute__((ext_vector_type(4)));
typedef unsigned int vu4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) vd2* gptr_d2;
typedef const __attribute__((address_space(1))) vd4* gptr_d4;
typedef const __attribute__((address_space(1))) vu4* gptr_u4;

// exact record: one 64 B cache line, up to four 16 B loads
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
// the reference's side test on the exact record (mad_tree.cpp:148)
__device__ __forceinline__ bool exact_goes_left(const madicp_node* __restrict__ nodes, int idx, double q0, double q1,
                                                double q2) {
  gptr_d2 p = (gptr_d2)(uintptr_t)(nodes + idx);
  const vd2 a = p[0], b = p[1], c = p[2];
  return dotc(q0 - a.x, q1 - a.y, q2 - b.x, b.y, c.x, c.y) < 0.0;
}

// the same test, returning the reference's computed fp64 value s = fl((q-m).n)
__device__ __forceinline__ double exact_side_value(const madicp_node* __restrict__ nodes, int idx, double q0, double q1,
                                                   double q2) {
  gptr_d2 p = (gptr_d2)(uintptr_t)(nodes + idx);
  const vd2 a = p[0], b = p[1], c = p[2];
  return dotc(q0 - a.x, q1 - a.y, q2 - b.x, b.y, c.x, c.y);
}

// per (query, tree) constants of the screening test
struct Screen {
  double r0, r1, r2;  // (q - o) * 2^-20
  double slack;       // kScreenDelta * (|q-o|_1 + rho)
  double xslack;      // bound on |s_computed - S_real| of the exact expression: 4.1u * sum|q_i - m_i|, rounded far up
};
__device__ __forceinline__ Screen make_screen(const TreeDesc& td, double q0, double q1, double q2) {
  const double e0 = q0 - td.origin[0], e1 = q1 - td.origin[1], e2 = q2 - td.origin[2];
  const double rho = td.rho;
  Screen s;
  s.r0 = e0 * 9.5367431640625e-07;  // 2^-20, exact scaling
  s.r1 = e1 * 9.5367431640625e-07;
  s.r2 = e2 * 9.5367431640625e-07;
  s.slack = kScreenDelta * ((fabs(e0) + fabs(e1) + fabs(e2)) + rho);
  s.xslack = 1e-14 * ((fabs(e0) + fabs(e1) + fabs(e2)) + rho) + 1e-300;
  return s;
}

// the three 21-bit two's-complement fields of a screening record's first 8 bytes (x = low word, y = high word):
// one signed bit-field extract each (the middle one after a 64-bit funnel shift)
__device__ __forceinline__ void unpack_k(unsigned int x, unsigned int y, int& k0, int& k1, int& k2) {
  k0 = __builtin_amdgcn_sbfe((int)x, 0, 21);
  k1 = __builtin_amdgcn_sbfe((int)__builtin_amdgcn_alignbit(y, x, 21), 0, 21);
  k2 = __builtin_amdgcn_sbfe((int)y, 10, 21);
}

// one screened side test: true = go left.  w = the node's 16-byte screening record
__device__ __forceinline__ bool screened_goes_left(const vu4 w, const Screen& sc, const madicp_node* __restrict__ nodes, int idx,
                                                   double q0, double q1, double q2) {
  int k0, k1, k2;
    unpack_k(w.x, w.y, k0, k1, k2);
  const double c = (double)__uint_as_float(w.z);
  const double sh = fma(sc.r0, (double)k0, fma(sc.r1, (double)k1, fma(sc.r2, (double)k2, -c)));  // (fused: see descend_multi)
  if (fabs(sh) > fma(kScreenC, fabs(c), sc.slack)) return sh < 0.0;
  return exact_goes_left(nodes, idx, q0, q1, q2);
}

// greedy root->leaf descent, no backtracking (mad_tree.cpp:144-152), screened.  Returns the leaf's index in the
// node array; leaf = its getLeafs() ordinal; depth = internal nodes visited.
__device__ __forceinline__ int descend(const TreeDesc& td, double q0, double q1, double q2, int& leaf, int& depth) {
  gptr_u4 cn = (gptr_u4)(uintptr_t)td.cnodes;
  const Screen sc = make_screen(td, q0, q1, q2);
  int idx = 0, d = 0, lb = 0;
  for (;;) {
    const vu4 w = cn[idx];
    const unsigned int roff = w.w & kRightMask;
    if (roff == 0u) break;  // single-node tree: the root itself is the leaf
    const bool left = screened_goes_left(w, sc, td.nodes, idx, q0, q1, q2);
    if (left) {
      idx += 1;
    } else {
      idx += (int)roff;
      lb += (int)(roff >> 1);  // leaves of the left sub-tree
    }
    ++d;
    if (w.w & (left ? kLeftLeaf : kRightLeaf)) break;
  }
  depth = d;
  leaf = lb;
  return idx;
}

// Correspondence reuse across Gauss-Newton rounds (exact, not approximate).
// While walking, a lane also keeps M = min over the visited levels of (|s^| - E): a proven lower bound of
// |S_real| = |(q - m).n| at every node of its path (|s^ - S_real| <= E, section "Exact screening").  Moving the
// query by d changes every S_real by at most d|n| <= d(1 + 2e-16).  So if, in a later round, the leaf has moved by
// less than M in total (sum of per-round displacement BOUNDS, each from the update itself: rotation angle x |p| +
// translation length, see solve_pose — two multiply-adds per pair instead of a second transform), every side test on
// the old path still has the sign it had — and |S_real| stays far above the fp64 rounding of the reference's own
// evaluation — hence the reference's descent would end in the same leaf after the same number of steps.  The lane
// then skips the walk and only refreshes the margin.  GN converges quadratically (leaf displacement per round at
// config 3: 3e-1, 2e-2, 1e-3, 7e-5, 6e-6, ... m), so from the 4th round on almost every pair is reused.  The cache
// is per (tree, leaf): leaf ordinal | depth << 26, and the margin as a float rounded DOWN.  Levels decided by the
// exact fallback contribute |s| minus a generous bound on its rounding error (a lane that keeps walking keeps its
// whole wave on the latency chain, so margins must not be thrown away).
// Bookkeeping without a store (round 5).  Margins and slacks used to be rewritten every round (left over = old - this
// round's displacement bound): 8 bytes written per pair and round, 96 MB per launch at BASELINE configs[4], on top of the 12
// read.  Now they are kept as THRESHOLDS on a per-pair wear that only grows:
//     A_ik(r) = |p_i| alpha(r) + beta(r) + r gamma_k
//     alpha(r) = sum over rounds j <= r of (rotation bound of update j) (1 + 1e-12) + 1.75e-11
//     beta(r)  = sum of (translation bound of update j) (1 + 1e-12) + 1e-11 |t_j|_1,     gamma_k = 1e-11 (rho_k + |o_k|_1 + 1)
// — an upper bound of everything leaf i can have moved against tree k since round 0, the rounding of the computed queries,
// distances and balls included (|q|_1 <= 1.75 |p| + |t|_1).  A walk at round r0 with margin M stores T = fl32_down(M + A(r0)),
// and the pair keeps its leaf while A(r) < T; a slack g is stored the same way.  Reused pairs are READ-ONLY; alpha and beta
// come out of the round's solve (two doubles beside the pose, Job::wear_ring).  fl32 of a threshold near 1 resolves 6e-8 m:
// margins below that are lost to rounding down (such a pair walks).
//
// Gate reuse (round 5; exact like the correspondence reuse it rides on).  The reference rejects a pair when
// sqrt(|ml - f.mean|^2) > min_ball + b_ratio |p| (mad_icp.cpp:81-83).  The ball depends on the leaf alone, and while a pair
// keeps its leaf (margin > displacement, above) the distance changes by at most the displacement.  So a pair rejected with
// slack g = distance - ball stays rejected as long as the leaf has moved by less than g since — and a rejected pair
// contributes NOTHING (mad_icp.cpp:83 `continue`), so it needs neither its leaf record (four 16-byte gathers: what a converged
// pass is bound by) nor the gate nor the Jacobian.  One threshold per pair (round 6; it was two floats): a rejected pair files
// T = -min(the leaf's threshold, fl32_down(slack + wear)) — negative marks "rejected", and |T| above the wear says BOTH that the
// pair keeps its leaf and that it is still outside its ball; an accepted pair files the leaf's threshold, positive.  What the
// single float gives up: a rejected pair whose slack wears off before its leaf's threshold walks again (and arrives at the same
// leaf) instead of being evaluated in place — the same decisions, a few more walks in rounds 1-3 (measured: DESIGN.md 3.1), for
// 4 of the 12 bytes every pair reads in every round.  At BASELINE configs[4] 79 % of the pairs are rejected at the converged pose, at the headline's
// 16 keyframes 44 % (oracle count) — whole wavefronts of a far keyframe's unit skip their gathers.  Same bits on or off: the
// pairs that are evaluated are accumulated in the same order.
constexpr unsigned int kCacheIdxMask = 0x03ffffffu;
constexpr int kCacheMaxDepth = 63;
// DEEP launches (a batch shares the chip) and their leaf-major rounds (icp_leaf_major.inc.h): ranges of at least
// kQueueMinPasses passes; at most kDeepTrees trees per workgroup and 64 passes per range (a queue entry is pass | tree | lane in
// 16 bits); kQueueCap queued walkers per wavefront before the queue is walked
constexpr int kQueueMinPasses = 2;
constexpr int kDeepTreesLog2 = 4, kDeepTrees = 1 << kDeepTreesLog2;
constexpr int kDeepPasses = 64;
constexpr int kQueueCap = 1024;
constexpr int kEvalCap = 128;  // queued evaluations per wavefront: flushed 64 at a time as soon as 64 are there, a (pass, tree) adds at most 64

// QPT independent descents per lane advanced in lock-step (their loads are issued together).  Phase 1 walks the
// LDS copy of the tree's top levels (s_top / s_exit, n_top entries), phase 2 continues in global memory.
// margin[j] (in/out): running minimum of |s^| - E over the levels walked.
template <int QPT>
__device__ __forceinline__ void descend_multi(const TreeDesc& td, const vu4* s_top, const int4* s_exit, int n_top,
                                              const double (&q0)[QPT], const double (&q1)[QPT], const double (&q2)[QPT],
                                              const bool (&valid)[QPT], int (&idx)[QPT], int (&leaf)[QPT],
                                              int (&depth)[QPT], double (&margin)[QPT]) {
  gptr_u4 cn = (gptr_u4)(uintptr_t)td.cnodes;
  Screen sc[QPT];
  bool live[QPT];
  vu4 w[QPT];
#pragma unroll
  for (int j = 0; j < QPT; ++j) {
    sc[j] = make_screen(td, q0[j], q1[j], q2[j]);
    idx[j] = 0;
    leaf[j] = 0;
    depth[j] = 0;
    live[j] = valid[j];
  }
  // one screened side test that also lowers the margin; exact fallback -> margin 0
  auto side = [&](const vu4 ww, int j, int node_index) -> bool {
    int k0, k1, k2;
    unpack_k(ww.x, ww.y, k0, k1, k2);
    const double c = (double)__uint_as_float(ww.z);
    // (fused: the screening bound E covers the rounding of the UNFUSED evaluation, a fused one errs less; the sign
    // decision is the same whenever |s^| > E, and the margin stays a lower bound of |S_real|)
    const double sh = fma(sc[j].r0, (double)k0, fma(sc[j].r1, (double)k1, fma(sc[j].r2, (double)k2, -c)));
    const double slack = fma(kScreenC, fabs(c), sc[j].slack);
    const double m = fabs(sh) - slack;
    if (m > 0.0) {
      margin[j] = fmin(margin[j], m);
      return sh < 0.0;
    }
    const double sx = exact_side_value(td.nodes, node_index, q0[j], q1[j], q2[j]);
    margin[j] = fmin(margin[j], fmax(fabs(sx) - sc[j].xslack, 0.0));  // NaN -> 0 (fmax returns the non-NaN operand)
    return sx < 0.0;
  };
  if (n_top > 0) {
    int e[QPT];
    bool intop[QPT];
#pragma unroll
    for (int j = 0; j < QPT; ++j) { e[j] = 0; intop[j] = valid[j]; }
    for (;;) {
      bool any = false;
#pragma unroll
      for (int j = 0; j < QPT; ++j)
        if (intop[j]) w[j] = s_top[e[j]];
#pragma unroll
      for (int j = 0; j < QPT; ++j) {
        if (intop[j]) {
          int k0, k1, k2;
    unpack_k(w[j].x, w[j].y, k0, k1, k2);
          const double c = (double)__uint_as_float(w[j].z);
          const double sh = fma(sc[j].r0, (double)k0, fma(sc[j].r1, (double)k1, fma(sc[j].r2, (double)k2, -c)));
          const double m = fabs(sh) - fma(kScreenC, fabs(c), sc[j].slack);
          bool left;
          if (m > 0.0) {
            margin[j] = fmin(margin[j], m);
            left = sh < 0.0;
          } else {
            const double sx = exact_side_value(td.nodes, td.top_dfs[e[j]], q0[j], q1[j], q2[j]);
            margin[j] = fmin(margin[j], fmax(fabs(sx) - sc[j].xslack, 0.0));
            left = sx < 0.0;
          }
          const unsigned int link = w[j].w;
          const unsigned int nx = (link >> (left ? 0 : 11)) & kTopNone;
          if (nx != kTopNone) {
            e[j] = (int)nx;
            any = true;
          } else {  // the walk leaves the staged levels here: node index, leaf ordinal and depth from the exit record
            const int4 ex = s_exit[e[j]];
            idx[j] = left ? ex.x : ex.x + 2 * ex.y - 1;
            leaf[j] = left ? ex.z : ex.z + ex.y;
            depth[j] = ex.w + 1;
            intop[j] = false;
            if (link & (left ? kTopLeftLeaf : kTopRightLeaf)) live[j] = false;  // arrived at a leaf
          }
        }
      }
      if (!any) break;
    }
  }
  for (;;) {
    bool any = false;
#pragma unroll
    for (int j = 0; j < QPT; ++j)
      if (live[j]) w[j] = cn[idx[j]];
#pragma unroll
    for (int j = 0; j < QPT; ++j) {
      if (live[j]) {
        const unsigned int roff = w[j].w & kRightMask;
        if (roff == 0u) {
          live[j] = false;
        } else {
          const bool left = side(w[j], j, idx[j]);
          if (left) {
            idx[j] += 1;
          } else {
            idx[j] += (int)roff;
            leaf[j] += (int)(roff >> 1);
          }
          ++depth[j];
          live[j] = !(w[j].w & (left ? kLeftLeaf : kRightLeaf));
          any |= live[j];
        }
      }
    }
    if (!any) break;
  }
}

// ---------------------------------------------------------------------------------------------------
// quantised normal + offset of one internal node (the first 12 bytes of its screening record); returns |m - o|_2
__device__ __forceinline__ double make_record(const madicp_node& nd, double o0, double o1, double o2, CNode& c) {
  const double e0 = nd.mean[0] - o0, e1 = nd.mean[1] - o1, e2 = nd.mean[2] - o2;
  const bool finite = isfinite(e0) && isfinite(e1) && isfinite(e2) && fabs(nd.dir[0]) <= 1.0000001 &&
                      fabs(nd.dir[1]) <= 1.0000001 && fabs(nd.dir[2]) <= 1.0000001;  // false on NaN
  if (!finite) {
    c.npack = 0ull;
    c.c = __uint_as_float(0x7f800000u);  // +inf: the screening test can never pass -> exact path
    return 0.0;
  }
  long long k[3];
  double nt[3];
  for (int a = 0; a < 3; ++a) {
    double v = rint(nd.dir[a] * 1048576.0);
    v = fmin(fmax(v, -1048575.0), 1048575.0);
    k[a] = (long long)v;
    nt[a] = v * 9.5367431640625e-07;
  }
  c.npack = ((unsigned long long)k[0] & 0x1fffffull) | (((unsigned long long)k[1] & 0x1fffffull) << 21) |
            (((unsigned long long)k[2] & 0x1fffffull) << 42);
  c.c = (float)((nt[0] * e0 + nt[1] * e1) + nt[2] * e2);
  return sqrt((e0 * e0 + e1 * e1) + e2 * e2);
}

// builds the screening records and the dense leaf records of a tree (after upload and after every transform); grid over
// nodes.  o = the tree's origin (mean of node 0), handed over by the host, which tracks it (no read-back, no sync).
__global__ void tree_compact(const madicp_node* __restrict__ nodes, CNode* __restrict__ cnodes, LeafRec* __restrict__ leaves,
                             int n, double o0, double o1, double o2) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const madicp_node nd = nodes[i];
  CNode c;
  c.npack = 0ull;
  c.c = 0.f;
  c.right = 0u;  // leaf records are never read, except the root's in a single-node tree ("right == 0 -> leaf")
  if (nd.right != 0) {
    make_record(nd, o0, o1, o2, c);
    c.right = (unsigned int)nd.right | (nodes[i + 1].right == 0 ? kLeftLeaf : 0u) |
              (nodes[i + nd.right].right == 0 ? kRightLeaf : 0u);
  } else {
    LeafRec lr;
    for (int a = 0; a < 3; ++a) {
      lr.mean[a] = nd.mean[a];
      lr.normal[a] = nd.dir[a];
    }
    lr.bbox0 = nd.bbox0;
    lr.pad_ = 0.0;
    leaves[nd.leaf_id] = lr;
  }
  cnodes[i] = c;
}

// the same records for the LDS-staged top array: entry e describes node top_dfs[e]; link[e] is its (static) link word
__global__ void tree_compact_top(const madicp_node* __restrict__ nodes, CNode* __restrict__ top,
                                 const int* __restrict__ top_dfs, const unsigned int* __restrict__ link, int n_top,
                                 double o0, double o1, double o2) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_top) return;
  CNode c;
  make_record(nodes[top_dfs[e]], o0, o1, o2, c);
  c.right = link[e];
  top[e] = c;
}

__global__ void moving_prep(const double* __restrict__ xyz, double* __restrict__ out, int L) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= L) return;
  const double x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
  double4 o;
  o.x = x; o.y = y; o.z = z;
  o.w = sqrt(dotc(x, y, z, x, y, z));
  reinterpret_cast<double4*>(out)[i] = o;
}

// MADicp::setMoving fed from a tree that is already resident (the current scan's tree, uploaded for the frame window):
// the moving set IS that tree's leaves in getLeafs() order (pipeline.cpp:143-144,154), so nothing crosses PCIe twice
__global__ void moving_from_leaves(const LeafRec* __restrict__ leaves, double* __restrict__ out, int L) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= L) return;
  const double x = leaves[i].mean[0], y = leaves[i].mean[1], z = leaves[i].mean[2];
  double4 o;
  o.x = x; o.y = y; o.z = z;
  o.w = sqrt(dotc(x, y, z, x, y, z));
  reinterpret_cast<double4*>(out)[i] = o;
}

__global__ void nn_descend(const TreeDesc td, const double* __restrict__ q, long long n,
                           uint32_t* __restrict__ out_leaf, uint32_t* __restrict__ out_node,
                           double* __restrict__ out_dist, int32_t* __restrict__ out_depth) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const double q0 = q[3 * i], q1 = q[3 * i + 1], q2 = q[3 * i + 2];
    int depth, leaf;
    const int idx = descend(td, q0, q1, q2, leaf, depth);
    if (out_leaf) out_leaf[i] = static_cast<uint32_t>(leaf);
    if (out_node) out_node[i] = static_cast<uint32_t>(idx);
    if (out_depth) out_depth[i] = depth;
    if (out_dist) {
      const LeafRec* lr = td.leaves + leaf;
      const double e0 = q0 - lr->mean[0], e1 = q1 - lr->mean[1], e2 = q2 - lr->mean[2];
      out_dist[i] = sqrt(dotc(e0, e1, e2, e0, e1, e2));
    }
  }
}

// The same with the tree's top levels staged in LDS (as icp_round does): each workgroup copies the first kTopLevels
// levels (<= 64 KiB) once and walks them there — ~15 cycles per level instead of an L1/L2 gather — then continues in
// global memory.  Same descent, same results (descend_multi is the routine the registration uses); pays when a
// workgroup walks many queries, i.e. for searchCloud-sized batches.  Dynamic LDS: kTopLdsBytes.
__global__ __launch_bounds__(1024) void nn_descend_top(const TreeDesc td, const double* __restrict__ q, long long n,
                                                       uint32_t* __restrict__ out_leaf, uint32_t* __restrict__ out_node,
                                                       double* __restrict__ out_dist, int32_t* __restrict__ out_depth) {
  extern __shared__ __attribute__((aligned(16))) char dyn_lds[];
  vu4* s_top = reinterpret_cast<vu4*>(dyn_lds);
  int4* s_exit = reinterpret_cast<int4*>(dyn_lds + kTopMax * sizeof(vu4));
  const int n_top = min(td.n_top, kTopMax);
  {
    gptr_u4 gt = (gptr_u4)(uintptr_t)td.top;
    gptr_u4 ge = (gptr_u4)(uintptr_t)td.top_exit;
    for (int e = threadIdx.x; e < n_top; e += blockDim.x) {
      s_top[e] = gt[e];
      reinterpret_cast<vu4*>(s_exit)[e] = ge[e];
    }
  }
  __syncthreads();
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const double a0[1] = {q[3 * i]}, a1[1] = {q[3 * i + 1]}, a2[1] = {q[3 * i + 2]};
    const bool wv[1] = {true};
    int xi[1], xl[1], xd[1];
    double xm[1] = {3.0e38};
    descend_multi<1>(td, s_top, s_exit, n_top, a0, a1, a2, wv, xi, xl, xd, xm);
    if (out_leaf) out_leaf[i] = static_cast<uint32_t>(xl[0]);
    if (out_node) out_node[i] = static_cast<uint32_t>(xi[0]);
    if (out_depth) out_depth[i] = xd[0];
    if (out_dist) {
      const LeafRec* lr = td.leaves + xl[0];
      const double e0 = a0[0] - lr->mean[0], e1 = a1[0] - lr->mean[1], e2 = a2[0] - lr->mean[2];
      out_dist[i] = sqrt(dotc(e0, e1, e2, e0, e1, e2));
    }
  }
}

// mean <- R mean + t ; dir <- R dir   (R row-major)
struct Pose12 {
  double v[12];  // R row-major, t
};
__global__ void tree_transform(madicp_node* __restrict__ nodes, int n, const Pose12 Rt) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double R[9], t[3];
#pragma unroll
  for (int k = 0; k < 9; ++k) R[k] = Rt.v[k];
#pragma unroll
  for (int k = 0; k < 3; ++k) t[k] = Rt.v[9 + k];
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
// Sum of kAcc per-lane accumulators over the 64 lanes of a wave, as a reduce-scatter: at every step a
// lane keeps one half of its slots, hands the other half to its partner and adds what it receives, so the
// slot count halves each time (16+8+4+2+1 exchanges plus one final pair-wise add instead of 6 per slot).
// All cross-lane traffic is VALU (no LDS): gfx950's v_permlane32_swap / v_permlane16_swap for the two
// widest steps, DPP row_ror:8 / row_half_mirror with bank masks for the next two, quad_perm for the last.
// With X = the slot a lower lane keeps and Y = the slot an upper lane keeps, one swap/masked-DPP turns
// (X, Y) into (X', Y') such that X' + Y' is the pair total on BOTH lanes — no selects.  (An LDS-crossbar
// version of the same tree, ds_bpermute, cost ~4 us of a 19 us kernel; this one is a few hundred cycles.)
// The tree is fixed, so the result is bit-reproducible.  Afterwards lane l holds the wave total of slot
// l>>1; even lanes store it.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ double pair_sum_swap32(double x, double y) {
  const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(x), (unsigned)__double2loint(y), false, false);
  const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(y), false, false);
  return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}
__device__ __forceinline__ double pair_sum_swap16(double x, double y) {
  const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(x), (unsigned)__double2loint(y), false, false);
  const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(y), false, false);
  return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}
// CTRL: the DPP lane permutation (an involution pairing upper with lower lanes); UPPER_BANKS: the 4-lane
// banks of a 16-lane row whose lanes are "upper" in this step
template <int CTRL, int UPPER_BANKS>
__device__ __forceinline__ double pair_sum_dpp(double x, double y) {
  const int xl = __double2loint(x), xh = __double2hiint(x), yl = __double2loint(y), yh = __double2hiint(y);
  // upper lanes: X' = partner's Y ; lower lanes keep X.   lower lanes: Y' = partner's X ; upper lanes keep Y.
  const int xl2 = __builtin_amdgcn_update_dpp(xl, yl, CTRL, 0xf, UPPER_BANKS, false);
  const int xh2 = __builtin_amdgcn_update_dpp(xh, yh, CTRL, 0xf, UPPER_BANKS, false);
  const int yl2 = __builtin_amdgcn_update_dpp(yl, xl, CTRL, 0xf, 0xf ^ UPPER_BANKS, false);
  const int yh2 = __builtin_amdgcn_update_dpp(yh, xh, CTRL, 0xf, 0xf ^ UPPER_BANKS, false);
  return __hiloint2double(xh2, xl2) + __hiloint2double(yh2, yl2);
}
template <int CTRL>
__device__ __forceinline__ double dpp_fetch(double v) {
  return __hiloint2double(__builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, false),
                          __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ void wave_reduce_scatter(const double* acc, int lane, double* out32 /*LDS, 32 slots*/) {
  double a[32];
#pragma unroll
  for (int v = 0; v < 32; ++v) a[v] = (v < kAcc) ? acc[v] : 0.0;
#pragma unroll
  for (int j = 0; j < 16; ++j) a[j] = pair_sum_swap32(a[j], a[j + 16]);   // partner lane ^ 32, upper = lane & 32
#pragma unroll
  for (int j = 0; j < 8; ++j) a[j] = pair_sum_swap16(a[j], a[j + 8]);     // partner lane ^ 16, upper = lane & 16
#pragma unroll
  for (int j = 0; j < 4; ++j) a[j] = pair_sum_dpp<0x128, 0xC>(a[j], a[j + 4]);  // row_ror:8 = lane ^ 8, upper = lane & 8
#pragma unroll
  for (int j = 0; j < 2; ++j) a[j] = pair_sum_dpp<0x141, 0xA>(a[j], a[j + 2]);  // row_half_mirror = lane ^ 7, upper = lane & 4
  {                                                                        // quad_perm [2,3,0,1] = lane ^ 2, upper = lane & 2
    const bool upper = (lane & 2) != 0;
    const double keep = upper ? a[1] : a[0];
    const double give = upper ? a[0] : a[1];
    a[0] = keep + dpp_fetch<0x4E>(give);
  }
  const double tot = a[0] + dpp_fetch<0xB1>(a[0]);                         // quad_perm [1,0,3,2] = lane ^ 1
  if ((lane & 1) == 0) out32[lane >> 1] = tot;
}

// ---------------------------------------------------------------------------------------------------
// The Gauss-Newton round as ONE kernel.
//
// A registration is  n x { linearise at X_r ; join ; solve ; X_{r+1} = X_r dX }.  Run as two kernels per round the
// solve (a single workgroup: join 2.8 us + a ~1000-instruction dependent fp64 chain on one lane 3.2 us + a
// dependent dispatch 1.9 us) was 35 % of the registration, with 255 CUs idle.  Here round r's kernel STARTS with the
// solve of round r-1, done redundantly by every workgroup: each joins the previous round's per-workgroup partials
// (one per CU: 256 x 232 B, L2-resident after the first reader of an XCD) in the same fixed order and its wave 0
// runs the same LDLT (wave-uniform, divisions spread over lanes), so all workgroups hold the bit-identical X_r without
// any inter-workgroup synchronisation; the kernel boundary is the only barrier.  While wave 0 solves, the other waves
// do what does not need the pose (copy the tree's top levels into LDS, zero their accumulators).  Workgroup 0 also
// does the bookkeeping (H, b, counters, X ring).  Partials and poses are double-buffered by round parity because a
// fast workgroup may finish round r while a slow one is still reading round r-1's.  After the last round a small
// icp_final joins/solves once more and counts the matched leaves.  Launches per registration: n + 1 instead of 2n.
//
// Work decomposition of the linearisation itself.  The moving leaves of a scan are cut into `ranges_per_tree` equal
// ranges; a *unit* is (tree k, range r); the host picks the count so there is one unit per workgroup and one workgroup
// (12 waves, 3 per SIMD — what the ~150 VGPRs of the 29 fp64 accumulators + walk state allow) per CU.  Units are
// ordered tree-major and cut into 8 contiguous pieces; workgroup b takes piece b % 8, i.e. (observed dispatch rule —
// used for speed only, never for correctness) XCD b % 8 only walks its own K/8 trees, so each private 4 MiB L2 serves
// 2 trees, not 16.  WHICH trees a piece holds is the host's choice (fill_job): the caller's list dealt round-robin, tree k
// to piece k % 8 — a caller's neighbours in the list are neighbours in space (keyframes along a trajectory), and with
// them in one piece the XCD that drew the keyframes around the scan worked while the others idled (64 keyframes, 8 scans
// in flight: 2 460 -> 2 900 registrations/s; 16 keyframes, one scan: +3 %).  Moving leaves arrive in the DFS order of the scan's own MAD-tree, i.e. spatially sorted, so the
// 64 lanes of a wave share the upper path and diverge only near the leaves.
//
// grid = (8 * slots, n_scans); blockIdx.y selects the registration (scans batched in flight).
// partials: [2][n_scans][join_rows(gridDim.x)][kAcc] (rows >= gridDim.x stay zero), then walk hints [2][n_scans][gridDim.x] ; totals (multi-GPU): [n_scans][kAcc], already all-reduced
// ---------------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------------
// 6x6 LDLT (lower, diagonal pivoting) factor + solve — the algorithm of Eigen::LDLT that
// `H_adder_.ldlt().solve(-b_adder_)` runs (mad_icp.cpp:111).  A: row-major, only the lower triangle is
// read.  Everything unrolled so the matrix lives in registers (a scratch-memory version of the same code
// cost ~15 us per GN round).  WAVE-UNIFORM: called by all 64 lanes of one wave with identical arguments;
// every lane runs the same serial algorithm, except that the independent divisions of a step (the column
// scaling by the pivot, the division by D) are spread over the lanes — lane i divides element i — and
// read back with v_readlane, so a step costs one division sequence (~15 dependent instructions) instead
// of up to five.  Each quotient is still one correctly rounded fp64 division of the same operands.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ double lane_get(double v, int lane /* compile-time constant */) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane),
                          __builtin_amdgcn_readlane(__double2loint(v), lane));
}
// num[i] /= den[i], i < N (lane-uniform operands; N <= 64)
template <int N>
__device__ __forceinline__ void div_spread(double (&num)[N], const double (&den)[N]) {
  const int lane = threadIdx.x & 63;
  double a = num[0], d = den[0];
#pragma unroll
  for (int i = 1; i < N; ++i) {
    a = (lane == i) ? num[i] : a;
    d = (lane == i) ? den[i] : d;
  }
  const double q = a / d;
#pragma unroll
  for (int i = 0; i < N; ++i) num[i] = lane_get(q, i);
}
#define MADICP_SWAP(a, b) do { const double t_ = (a); (a) = (b); (b) = t_; } while (0)
template <int K>
__device__ __forceinline__ void scale_column_k(double (&m)[6][6], double akk) {  // m[i][K] /= akk for i > K
  constexpr int N = 5 - K;
  if constexpr (N > 0) {
    double num[N], den[N];
#pragma unroll
    for (int i = 0; i < N; ++i) { num[i] = m[K + 1 + i][K]; den[i] = akk; }
    div_spread<N>(num, den);
#pragma unroll
    for (int i = 0; i < N; ++i) m[K + 1 + i][K] = num[i];
  }
}
__device__ __forceinline__ void scale_column(double (&m)[6][6], int k /* unrolled loop index */, double akk) {
  switch (k) {
    case 0: scale_column_k<0>(m, akk); break;
    case 1: scale_column_k<1>(m, akk); break;
    case 2: scale_column_k<2>(m, akk); break;
    case 3: scale_column_k<3>(m, akk); break;
    case 4: scale_column_k<4>(m, akk); break;
    default: break;
  }
}
// x = A^-1 rhs by LDLT with diagonal pivoting — Eigen::LDLT's algorithm (pivot order, transpositions, the zero-matrix and
// tiny-pivot guards), evaluated for LATENCY: this solve sits on the serial path of every round, on one wave, with the
// whole chip waiting for it.  The arithmetic is free to differ from Eigen's in the last bits — the contract on the pose
// is 1e-5 m / 1e-5 rad, its inputs (the joined H, b) already differ from the reference's by their summation order, and no
// branch of the descent or the gate depends on anything here except through the pose — so, unlike everywhere else in
// this file: fused multiply-adds are allowed, a quotient is n * (1/d) with the reciprocal refined to full precision by
// two Newton steps and the quotient by one residual step (8 dependent instructions instead of the ~12 of the correctly
// rounded expansion, and ONE reciprocal serves a whole column), and independent terms are summed pairwise.  Every
// workgroup (and icp_final) runs the same instructions on the same bits, so all of them still hold the same pose.
// -DMADICP_EXACT_SOLVE restores Eigen's operation order with correctly rounded divisions.
__device__ __forceinline__ double fast_rcp(double d) {
#pragma clang fp contract(fast)
  double y = __builtin_amdgcn_rcp(d);
  y = __builtin_fma(__builtin_fma(-d, y, 1.0), y, y);
  y = __builtin_fma(__builtin_fma(-d, y, 1.0), y, y);
  return y;
}
__device__ __forceinline__ double fast_div(double n, double d, double y /* = fast_rcp(d) */) {
#pragma clang fp contract(fast)
  const double q = n * y;
  return __builtin_fma(__builtin_fma(-d, q, n), y, q);
}
#ifndef MADICP_EXACT_SOLVE
__device__ __forceinline__ void ldlt6_solve(const double* A, const double* rhs, double* x) {
#pragma clang fp contract(fast)
  double m[6][6];
  int tr[6];
#pragma unroll
  for (int r = 0; r < 6; ++r)
#pragma unroll
    for (int c = 0; c < 6; ++c) m[r][c] = A[r * 6 + c];
  bool zero_matrix = false;
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    // the pivot search reads the NOT YET updated diagonal (left-looking factorisation): off the dependent chain
    int big = k;
    double best = fabs(m[k][k]);
#pragma unroll
    for (int i = k + 1; i < 6; ++i) {
      const double a = fabs(m[i][i]);
      if (a > best) { best = a; big = i; }
    }
    if (zero_matrix) big = k;
    // the pivot is the same in every lane: as a SCALAR it turns the symmetric row/column exchange into a uniform branch
    // around plain register moves (as lane-wise selects the 15 possible exchanges were ~400 v_cndmask per solve —
    // more issue time than the whole floating-point chain)
    big = __builtin_amdgcn_readfirstlane(big);
    tr[k] = big;
#pragma unroll
    for (int p = k + 1; p < 6; ++p) {
      if (big == p) {
#pragma unroll
        for (int j = 0; j < k; ++j) MADICP_SWAP(m[k][j], m[p][j]);
#pragma unroll
        for (int i = p + 1; i < 6; ++i) MADICP_SWAP(m[i][k], m[i][p]);
        MADICP_SWAP(m[k][k], m[p][p]);
#pragma unroll
        for (int i = k + 1; i < p; ++i) MADICP_SWAP(m[i][k], m[p][i]);
      }
    }
    if (k > 0 && !zero_matrix) {
      double tmp[6];
#pragma unroll
      for (int j = 0; j < k; ++j) tmp[j] = m[j][j] * m[k][j];
      // rows k..5 against tmp: k-term dot products, pairwise
#pragma unroll
      for (int i = k; i < 6; ++i) {
        double s0 = m[i][0] * tmp[0], s1 = 0.0;
        if (k > 1) s1 = m[i][1] * tmp[1];
        if (k > 2) s0 = __builtin_fma(m[i][2], tmp[2], s0);
        if (k > 3) s1 = __builtin_fma(m[i][3], tmp[3], s1);
        if (k > 4) s0 = __builtin_fma(m[i][4], tmp[4], s0);
        m[i][k] -= (s0 + s1);
      }
    }
    const double akk = m[k][k];
    const bool ok = fabs(akk) > 0.0;
    if (k == 0 && !ok) zero_matrix = true;  // whole diagonal is zero: identity transpositions, no scaling
    if (ok && !zero_matrix && k < 5) {
      const double y = fast_rcp(akk);
#pragma unroll
      for (int i = k + 1; i < 6; ++i) m[i][k] = fast_div(m[i][k], akk, y);
    }
  }
  double y[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) y[i] = rhs[i];
#pragma unroll
  for (int k = 0; k < 6; ++k) {
#pragma unroll
    for (int p = k + 1; p < 6; ++p)
      if (tr[k] == p) MADICP_SWAP(y[k], y[p]);
  }
#pragma unroll
  for (int i = 1; i < 6; ++i) {
    double s0 = m[i][0] * y[0], s1 = 0.0;
    if (i > 1) s1 = m[i][1] * y[1];
    if (i > 2) s0 = __builtin_fma(m[i][2], y[2], s0);
    if (i > 3) s1 = __builtin_fma(m[i][3], y[3], s1);
    if (i > 4) s0 = __builtin_fma(m[i][4], y[4], s0);
    y[i] -= (s0 + s1);
  }
  const double tol = 2.2250738585072014e-308;  // numeric_limits<double>::min()
#pragma unroll
  for (int i = 0; i < 6; ++i) {  // six independent quotients: their chains overlap in the pipeline
    const double d = m[i][i];
    y[i] = (fabs(d) > tol) ? fast_div(y[i], d, fast_rcp(d)) : 0.0;
  }
#pragma unroll
  for (int i = 4; i >= 0; --i) {
    double s0 = m[i + 1][i] * y[i + 1], s1 = 0.0;
    if (i + 2 < 6) s1 = m[i + 2][i] * y[i + 2];
    if (i + 3 < 6) s0 = __builtin_fma(m[i + 3][i], y[i + 3], s0);
    if (i + 4 < 6) s1 = __builtin_fma(m[i + 4][i], y[i + 4], s1);
    if (i + 5 < 6) s0 = __builtin_fma(m[i + 5][i], y[i + 5], s0);
    y[i] -= (s0 + s1);
  }
#pragma unroll
  for (int k = 5; k >= 0; --k) {
#pragma unroll
    for (int p = k + 1; p < 6; ++p)
      if (tr[k] == p) MADICP_SWAP(y[k], y[p]);
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) x[i] = y[i];
}

// sin(x) for |x| <= 0.5 by its Taylor series in Horner form (x^19/19! < 2e-23: below half an ulp of the result); the
// library routine (argument reduction, branches) is a longer dependent chain than the solve's whole back-substitution
__device__ __forceinline__ double sin_small(double x) {
#pragma clang fp contract(fast)
  const double z = x * x;
  double p = -8.2206352466243297e-18;           // -1/19!
  p = __builtin_fma(p, z, 2.8114572543455206e-15);   //  1/17!
  p = __builtin_fma(p, z, -7.6471637318198164e-13);  // -1/15!
  p = __builtin_fma(p, z, 1.6059043836821613e-10);   //  1/13!
  p = __builtin_fma(p, z, -2.5052108385441720e-08);  // -1/11!
  p = __builtin_fma(p, z, 2.7557319223985893e-06);   //  1/9!
  p = __builtin_fma(p, z, -1.9841269841269841e-04);  // -1/7!
  p = __builtin_fma(p, z, 8.3333333333333332e-03);   //  1/5!
  p = __builtin_fma(p, z, -1.6666666666666666e-01);  // -1/3!
  return __builtin_fma(x * z, p, x);
}

// lie_algebra.h:39-52, R row-major.  Wave-uniform.  Same formula as the reference (first-order branch below theta^2 = 1e-8,
// Rodrigues with 2 sin^2(theta/2) above), evaluated like ldlt6_solve above: for latency.
__device__ __forceinline__ void exp_so3(const double* w, double* R) {
#pragma clang fp contract(fast)
  const double th2 = dotc(w[0], w[1], w[2], w[0], w[1], w[2]);
  if (th2 < 1e-8) {
    const double W[9] = {0.0, -w[2], w[1], w[2], 0.0, -w[0], -w[1], w[0], 0.0};
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = ((i % 4 == 0) ? 1.0 : 0.0) + W[i];
    return;
  }
  const double th = sqrt(th2);
  const double ith = fast_rcp(th);
  const double k[3] = {fast_div(w[0], th, ith), fast_div(w[1], th, ith), fast_div(w[2], th, ith)};
  const double Kx[9] = {0.0, -k[2], k[1], k[2], 0.0, -k[0], -k[1], k[0], 0.0};
  double sh, s;
  if (th <= 0.5) {
    sh = sin_small(0.5 * th);
    s = sin_small(th);
  } else {
    sh = sin(0.5 * th);
    s = sin(th);
  }
  const double omc = 2.0 * sh * sh;
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const double kk = Kx[3 * r] * Kx[c] + Kx[3 * r + 1] * Kx[3 + c] + Kx[3 * r + 2] * Kx[6 + c];
      R[3 * r + c] = ((r == c) ? 1.0 : 0.0) + s * Kx[3 * r + c] + omc * kk;
    }
}
#else
__device__ __forceinline__ void ldlt6_solve(const double* A, const double* rhs, double* x) {
  double m[6][6];
  int tr[6];
#pragma unroll
  for (int r = 0; r < 6; ++r)
#pragma unroll
    for (int c = 0; c < 6; ++c) m[r][c] = A[r * 6 + c];
  bool zero_matrix = false;
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    int big = k;
    double best = fabs(m[k][k]);
#pragma unroll
    for (int i = k + 1; i < 6; ++i) {
      const double a = fabs(m[i][i]);
      if (a > best) { best = a; big = i; }
    }
    if (zero_matrix) big = k;
    tr[k] = big;
#pragma unroll
    for (int p = k + 1; p < 6; ++p) {
      if (big == p) {
#pragma unroll
        for (int j = 0; j < k; ++j) MADICP_SWAP(m[k][j], m[p][j]);
#pragma unroll
        for (int i = p + 1; i < 6; ++i) MADICP_SWAP(m[i][k], m[i][p]);
        MADICP_SWAP(m[k][k], m[p][p]);
#pragma unroll
        for (int i = k + 1; i < p; ++i) MADICP_SWAP(m[i][k], m[p][i]);
      }
    }
    if (k > 0 && !zero_matrix) {
      double tmp[6];
#pragma unroll
      for (int j = 0; j < k; ++j) tmp[j] = m[j][j] * m[k][j];
      double a = m[k][0] * tmp[0];
#pragma unroll
      for (int j = 1; j < k; ++j) a += m[k][j] * tmp[j];
      m[k][k] -= a;
#pragma unroll
      for (int i = k + 1; i < 6; ++i) {
        double s = m[i][0] * tmp[0];
#pragma unroll
        for (int j = 1; j < k; ++j) s += m[i][j] * tmp[j];
        m[i][k] -= s;
      }
    }
    const double akk = m[k][k];
    const bool ok = fabs(akk) > 0.0;
    if (k == 0 && !ok) zero_matrix = true;  // whole diagonal is zero: identity transpositions, no scaling
    if (ok && !zero_matrix) {
      scale_column(m, k, akk);
    }
  }
  double y[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) y[i] = rhs[i];
#pragma unroll
  for (int k = 0; k < 6; ++k) {
#pragma unroll
    for (int p = k + 1; p < 6; ++p)
      if (tr[k] == p) MADICP_SWAP(y[k], y[p]);
  }
#pragma unroll
  for (int i = 1; i < 6; ++i) {
    double a = m[i][0] * y[0];
#pragma unroll
    for (int j = 1; j < i; ++j) a += m[i][j] * y[j];
    y[i] -= a;
  }
  const double tol = 2.2250738585072014e-308;  // numeric_limits<double>::min()
  {
    double num[6], den[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) { num[i] = y[i]; den[i] = m[i][i]; }
    div_spread<6>(num, den);
#pragma unroll
    for (int i = 0; i < 6; ++i) y[i] = (fabs(m[i][i]) > tol) ? num[i] : 0.0;
  }
#pragma unroll
  for (int i = 4; i >= 0; --i) {
    double a = m[i + 1][i] * y[i + 1];
#pragma unroll
    for (int j = i + 2; j < 6; ++j) a += m[j][i] * y[j];
    y[i] -= a;
  }
#pragma unroll
  for (int k = 5; k >= 0; --k) {
#pragma unroll
    for (int p = k + 1; p < 6; ++p)
      if (tr[k] == p) MADICP_SWAP(y[k], y[p]);
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) x[i] = y[i];
}

// lie_algebra.h:39-52, R row-major.  Wave-uniform like ldlt6_solve.  K = skew(w)/theta has six non-zero entries
// +-w_i/theta: three divisions (spread over three lanes), the rest by negation — -(a/b) and (-a)/b are the same double.
__device__ __forceinline__ void exp_so3(const double* w, double* R) {
  const double th2 = dotc(w[0], w[1], w[2], w[0], w[1], w[2]);
  const double th = sqrt(th2);
  if (th2 < 1e-8) {
    const double W[9] = {0.0, -w[2], w[1], w[2], 0.0, -w[0], -w[1], w[0], 0.0};
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = ((i % 4 == 0) ? 1.0 : 0.0) + W[i];
    return;
  }
  double k[3] = {w[0], w[1], w[2]};
  const double den[3] = {th, th, th};
  div_spread<3>(k, den);
  const double Kx[9] = {0.0, -k[2], k[1], k[2], 0.0, -k[0], -k[1], k[0], 0.0};
  double cK[9];
  const double omc = 2.0 * sin(th / 2.0) * sin(th / 2.0);
  const double s = sin(th);
#pragma unroll
  for (int i = 0; i < 9; ++i) cK[i] = omc * Kx[i];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const double kk = dots(cK[3 * r], cK[3 * r + 1], cK[3 * r + 2], Kx[c], Kx[3 + c], Kx[6 + c]);
      R[3 * r + c] = (((r == c) ? 1.0 : 0.0) + s * Kx[3 * r + c]) + kk;
    }
}

#endif  // MADICP_EXACT_SOLVE

// LDS written by some lanes of a wave, read by other lanes of the SAME wave: the hardware keeps a wave's LDS
// operations in order; this only stops the compiler from reordering them
__device__ __forceinline__ void wave_lds_order() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

#ifndef MADICP_EXACT_SOLVE
// dx = H^-1 (-b) with the 64 LANES of the wave as the parallel resource.
//
// Run "wave-uniform" (every lane the same scalar program, ldlt6_solve above) the 6x6 solve is ~1500 instructions issued
// by one wave at one fp64 instruction per four cycles: 2.2 us of every round, measured (profiles/r2_*_phase_stamps.md),
// bound by ISSUE, not by the dependency chain — shortening the chain (FMA, reciprocal) moved it by 0.2 us, making the
// pivot exchanges scalar by nothing.  Here lane 7i+j holds entry (i,j) of the augmented system [P H P^T | -P b] and a
// Gauss-Jordan sweep updates all 42 entries with ONE instruction sequence per pivot: ~25 instructions per step, ~250
// for the solve.
//   * Pivot order.  Eigen::LDLT (mad_icp.cpp:111) picks, at step k, the largest |diagonal| among the rows not yet used
//     — and its left-looking update has not touched those entries, so the order is simply the ORIGINAL diagonal sorted
//     by decreasing magnitude.  The permutation is computed up front: lane 6i+j compares |H_jj| with |H_ii|, one ballot,
//     six popcounts give every row's rank (scalar); the permuted entries are gathered from the packed totals in LDS.
//   * The elimination is Gauss-Jordan (rows above the pivot are eliminated too: no back-substitution pass), the final
//     x_i = a_i6 / a_ii guarded like Eigen's (|D_ii| <= DBL_MIN -> 0; a pivot that is exactly zero eliminates nothing).
//     For the symmetric positive definite H of a registration with matches this is the same solution to rounding
//     (the pose contract is 1e-5; measured agreement with the oracle's LDLT ~1e-15); a singular H (no matches: all
//     zero) gives dx = 0 on both sides.
// total: the joined adders in LDS (packed lower triangle, column by column: entry (r,c), r >= c, at 6c - c(c-1)/2 + r - c;
// b at 21..26).  Whole wave, identical `total` -> identical dx in every lane.
__device__ __forceinline__ void gn_solve_lanes(const double* total /*LDS*/, double (&dx)[6], const int lane /* threadIdx.x & 63 */) {
#pragma clang fp contract(fast)
  __shared__ double s_dx[8];
  unsigned int P;  // perm[r] = original row that goes to position r, 4 bits each
  {
    const int ci = min(lane / 6, 5), cj = lane % 6;
    const double di = fabs(total[6 * ci - (ci * (ci - 1)) / 2]);
    const double dj = fabs(total[6 * cj - (cj * (cj - 1)) / 2]);
    const bool before = lane < 36 && (dj > di || (dj == di && cj < ci));  // row cj is taken before row ci
    const unsigned long long mask = __ballot(before);
    P = 0u;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int rank = __builtin_popcountll((mask >> (6 * i)) & 63ull);  // (a NaN diagonal can tie ranks: indices stay in range)
      P |= (unsigned int)i << (4 * rank);
    }
  }
  const int i = min(lane / 7, 5), j = lane % 7;
  const int pi = min((int)((P >> (4 * i)) & 7u), 5);
  const int pj = min((int)((P >> (4 * min(j, 5))) & 7u), 5);
  const int hi = max(pi, pj), lo = min(pi, pj);
  const int src = (j < 6) ? (6 * lo - (lo * (lo - 1)) / 2 + (hi - lo)) : (21 + pi);
  double a = total[src];
  if (j == 6) a = -a;
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const double d = lane_get(a, 8 * k);          // pivot a_kk: the same in every lane
    const double aik = __shfl(a, 7 * i + k, 64);  // own row, pivot column
    const double akj = __shfl(a, 7 * k + j, 64);  // pivot row, own column
    const double f = fast_div(aik, d, fast_rcp(d));
    if (fabs(d) > 0.0 && i != k) a = __builtin_fma(-f, akj, a);
  }
  const double dii = __shfl(a, 8 * i, 64);
  const double tol = 2.2250738585072014e-308;  // numeric_limits<double>::min(), Eigen's guard in LDLT::solve
  const double sol = (fabs(dii) > tol) ? fast_div(a, dii, fast_rcp(dii)) : 0.0;
  if (j == 6 && lane < 42) s_dx[pi] = sol;  // un-permute: position i holds the unknown of original row perm[i]
  wave_lds_order();
#pragma unroll
  for (int r = 0; r < 6; ++r) dx[r] = s_dx[r];
  wave_lds_order();  // (the next call's stores must not overtake these reads)
}
#endif

// `moved[2]`: upper bounds of the update's rotation angle and translation length — what the correspondence reuse needs
// to bound how far any leaf moves: |X_next p - X p| <= |R|_2 (|dR - I|_2 |p| + |dt|) <= (1+1e-6)(moved[0] |p| + moved[1])
// (|expSO3(w) - I|_2 = 2 sin(|w|/2) <= |w|, and = |w| for the first-order branch; |R|_2 <= 1 + 1e-7 after 15 updates).
// `lane`: the caller's lane index — a parameter so that a caller looping over rounds (icp_persist) can hand in a copy the
// compiler cannot hoist the solve's per-lane index arithmetic out of its loop with
__device__ __forceinline__ void solve_pose(const double* total, const double (&X)[12], bool update, double (&Xn)[12],
                                           double (&H)[36], double (&b)[6], double (&moved)[2], const int lane) {
  moved[0] = 0.0;
  moved[1] = 0.0;
  {
    int v = 0;
#pragma unroll
    for (int c = 0; c < 6; ++c)
#pragma unroll
      for (int r = c; r < 6; ++r) {
        H[r * 6 + c] = total[v];
        H[c * 6 + r] = total[v];
        ++v;
      }
  }
#pragma unroll
  for (int r = 0; r < 6; ++r) b[r] = total[21 + r];
#pragma unroll
  for (int i = 0; i < 12; ++i) Xn[i] = X[i];
  if (update) {
    double dx[6], dR[9];
#ifndef MADICP_EXACT_SOLVE
    gn_solve_lanes(total, dx, lane);
#else
    (void)lane;
    double nb[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) nb[r] = -b[r];
    ldlt6_solve(H, nb, dx);
#endif
    exp_so3(dx + 3, dR);
    moved[0] = sqrt(dotc(dx[3], dx[4], dx[5], dx[3], dx[4], dx[5])) * (1.0 + 1e-6);
    moved[1] = sqrt(dotc(dx[0], dx[1], dx[2], dx[0], dx[1], dx[2])) * (1.0 + 1e-6);
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
      for (int c = 0; c < 3; ++c) Xn[3 * r + c] = dots(X[3 * r], X[3 * r + 1], X[3 * r + 2], dR[c], dR[3 + c], dR[6 + c]);
      Xn[9 + r] = dots(X[3 * r], X[3 * r + 1], X[3 * r + 2], dx[0], dx[1], dx[2]) + X[9 + r];
    }
  }
}

// Join of the per-workgroup partials in a fixed order.  Whole workgroup (kBlock threads): lane = (row group g = 0..3,
// column pair c = 0..14) of wave w; one 16-byte load covers columns 2c, 2c+1 of a row, so one wave instruction covers
// four whole rows = 928 contiguous bytes (an earlier version used 16 strided 8-byte loads per lane: 2.7x the
// instructions for the same bytes, and it was the vector-memory pipe's time that the prologue waited for).  Row group
// G = 4w + g sums rows G, G + 48, G + 96, ... in sequence; wave 0 then adds the 48 group sums in order.  The order
// depends only on the row count, so every workgroup of a launch — and icp_reduce / icp_final — computes the
// bit-identical total.
// No predicates: a scan's partials are stored as join_rows(nblocks) >= nblocks rows, a multiple of 48 and at least
// 48 * kJoinRows; the rows no workgroup writes are kept zero by the host (adding +0.0 is exact), so every lane issues
// the same kJoinRows unconditional loads off one base address.  (Per-load predicates compiled to ~25 instructions of
// control flow each — executed by all 12 waves before anything else could be issued.)
// Split in issue (loads in flight) / stage 1 (ends with the workgroup barrier) / stage 2 (wave 0), so that the
// caller can overlap the memory round trip with other work.  Readable one double past the last row (c = 14 loads
// columns 28 and "29"): the host pads.
constexpr int kJoinGroups = 4 * kWaves;  // 48
constexpr int kJoinRows = 6;             // rows per lane held in registers (nblocks <= 288); longer ones stream
__host__ __device__ constexpr int join_rows(int nblocks) {
  return (nblocks <= kJoinGroups * kJoinRows) ? kJoinGroups * kJoinRows : (nblocks + kJoinGroups - 1) / kJoinGroups * kJoinGroups;
}
typedef double vd2u __attribute__((ext_vector_type(2), aligned(8)));
typedef const __attribute__((address_space(1))) vd2u* gptr_d2u;
typedef double JoinSeg[kJoinGroups][30];
struct JoinLoads {
  vd2u x[kJoinRows];
};
__device__ __forceinline__ void join_issue(const double* __restrict__ partials, JoinLoads& jl) {
  const unsigned lane = threadIdx.x & 63;
  const unsigned g = min((lane * 274u) >> 12, 3u);  // lane / 15, lanes 60..63 duplicate group 3 (never stored)
  const unsigned c = min(lane - 15u * g, 14u);
  const unsigned G = 4u * (threadIdx.x >> 6) + g;
  gptr_d2u p = (gptr_d2u)(uintptr_t)(partials + (G * kAcc + 2u * c));
#pragma unroll
  for (int i = 0; i < kJoinRows; ++i) jl.x[i] = *(gptr_d2u)((const __attribute__((address_space(1))) double*)p + i * (kJoinGroups * kAcc));
}
__device__ __forceinline__ void join_stage1(const double* __restrict__ partials, int rows, const JoinLoads& jl,
                                            JoinSeg& seg /*LDS*/) {
  const unsigned lane = threadIdx.x & 63;
  const unsigned g = (lane * 274u) >> 12;
  const unsigned c = lane - 15u * g;
  const unsigned G = 4u * (threadIdx.x >> 6) + g;
  double a0 = 0.0, a1 = 0.0;
#pragma unroll
  for (int i = 0; i < kJoinRows; ++i) { a0 += jl.x[i].x; a1 += jl.x[i].y; }
  if (g < 4) {
    for (int row = G + kJoinGroups * kJoinRows; row < rows; row += kJoinGroups) {  // (rows > 288 only)
      const vd2u x = *(gptr_d2u)(uintptr_t)(partials + 2 * c + (long long)row * kAcc);
      a0 += x.x;
      a1 += x.y;
    }
    seg[G][2 * c] = a0;
    seg[G][2 * c + 1] = a1;
  }
  __syncthreads();
}
// stage 2 (wave 0 only): the group sums -> total[kAcc] in LDS, visible to wave 0.  THE summation order of a round's
// adders, shared by every route (icp_round's prologue, icp_final, icp_reduce, and icp_persist, whose workgroups exchange
// their rows inside the launch): with x = row & 7 (the workgroup's XCD under the observed dispatch rule — a name for the
// group, nothing depends on the placement),
//     seg[G]  = 0 + row[G] + row[G + 48] + row[G + 96] + ...      G = 0..47          (stage 1)
//     F[x]    = seg[x] + seg[x + 8] + ... + seg[x + 40]           x = 0..7           (the fold of one XCD's rows)
//     total   = F[0] + F[1] + ... + F[7]
// — 14 dependent additions here instead of 47, and the two levels are what icp_persist's XCD leaders / consumers compute.
constexpr int kFoldGroups = 8;                          // row & 7
constexpr int kFoldChains = kJoinGroups / kFoldGroups;  // 6 interleaved chains per group
__device__ __forceinline__ void join_stage2_wave0(const JoinSeg& seg, double* total /*LDS kAcc*/) {
  if (threadIdx.x < kAcc) {
    double r[kJoinGroups];
#pragma unroll
    for (int k = 0; k < kJoinGroups; ++k) r[k] = seg[k][threadIdx.x];  // all LDS reads in flight before the first add
    double a = 0.0;
#pragma unroll
    for (int x = 0; x < kFoldGroups; ++x) {
      double f = r[x];
#pragma unroll
      for (int q = 1; q < kFoldChains; ++q) f += r[x + kFoldGroups * q];
      a = (x == 0) ? f : a + f;
    }
    total[threadIdx.x] = a;
  }
  wave_lds_order();
}
// whole workgroup in, total[] valid for WAVE 0 out (icp_reduce / icp_final); rows = join_rows(nblocks)
__device__ __forceinline__ void join_partials(const double* __restrict__ partials, int rows, double* total /*LDS kAcc*/) {
  __shared__ JoinSeg seg;
  JoinLoads jl;
  join_issue(partials, jl);
  join_stage1(partials, rows, jl, seg);
  if (threadIdx.x < 64) join_stage2_wave0(seg, total);
}

__device__ __forceinline__ double wave_uniform(double v) {  // value known to be identical in all lanes -> SGPR pair
  return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}

// development instrumentation (tools/stamps.py builds a separate library with -DMADICP_STAMPS): wave 0 of every
// workgroup of scan 0 records the 100 MHz wall clock at a few points of icp_round
#ifdef MADICP_STAMPS
__device__ unsigned long long g_stamps[16 * 256 * 16];
#define MADICP_STAMP(n)                                                                                  \
  if (threadIdx.x == 0 && blockIdx.y == 0 && round < 16 && blockIdx.x < 256)                             \
  g_stamps[(round * 256 + blockIdx.x) * 16 + (n)] = wall_clock64()
#define MADICP_STAMP_WAIT() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory")
#else
#define MADICP_STAMP(n)
#define MADICP_STAMP_WAIT()
#endif

// exchange granules (icp_persist, and icp_round's FOLD variant): see icp_persist for the protocol
typedef __attribute__((address_space(1))) unsigned long long* gptr_g64;
constexpr int kRowGranules = 2 * kAcc;  // 480 bytes per row
__host__ __device__ constexpr size_t xch_level1(int n_scans, int grid) { return (size_t)2 * n_scans * grid * kRowGranules; }
__host__ __device__ constexpr size_t xch_granules(int n_scans, int grid) {
  return xch_level1(n_scans, grid) + (size_t)2 * n_scans * kFoldGroups * kRowGranules;
}
__device__ __forceinline__ void granule_store(gptr_g64 g, unsigned tag, double v) {
  const unsigned long long t = (unsigned long long)tag << 32;
  __hip_atomic_store(g, t | (unsigned)__double2loint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(g + 1, t | (unsigned)__double2hiint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// one poll: true when both halves carry `tag`
__device__ __forceinline__ bool granule_try(gptr_g64 g, unsigned tag, double& v) {
  const unsigned long long a = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned long long b = __hip_atomic_load(g + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  v = __hiloint2double((int)(unsigned)b, (int)(unsigned)a);
  return (unsigned)(a >> 32) == tag && (unsigned)(b >> 32) == tag;
}
constexpr unsigned long long kSpinLimitTicks = 20000000ull;  // 0.2 s of the 100 MHz wall clock

// ---- keyframe sharding without a collective between rounds: peer-mapped mailboxes (option "shard_p2p") ------------------------
// The join of mad_icp.cpp:106-109 over the ranks of a sharded registration is 30 doubles per scan and round.  As an
// ncclAllReduce it is a kernel of its own behind an icp_reduce launch, strictly between two rounds (13.4 us of
// round + 4.5 + 10-25 us).  Here every rank owns a MAILBOX — device memory every peer has mapped (hipIpc) — with one row of
// tagged 16-byte granules per (slot, scan, sending rank).  The round kernel joins its own partials as on one GPU; workgroup 0
// then WRITES the rank's 30 sums into its row of every peer's mailbox (one xGMI store each way, no fence: the tag in each
// half says that the value is there), and wave 0 of every workgroup polls the other ranks' rows in its OWN mailbox (local
// memory) and adds all N rows in rank order — every rank the same order, so every rank solves from the same bits.  No launch,
// no collective and no host between two rounds.  Slots: (registration parity, round parity) — a rank can be at most one
// exchange ahead of the slowest one, and never a whole registration.  Bounded spins (Job::error = 4 -> MADICP_ERR_COMM).
constexpr int kMaxRanks = 8;
constexpr int kP2pSlots = 4;
struct PeerBox {
  unsigned long long* box[kMaxRanks];  // box[q]: rank q's mailbox as mapped in THIS process (box[rank]: the own one)
  int n_ranks, rank;
  int flags_in_box;                    // icp_final: the matched flags travel over the mailboxes too (every scan's L <= kP2pFlagLeaves)
  int scan0;                           // first mailbox scan row of this launch (a batch's second half: its first scan)
  unsigned long long spin_ticks;       // 100 MHz ticks a poll may take before the registration is declared lost
};
// (the registration counter that tags the rows — the same on every rank — is Job::p2p_epoch, not a kernel argument: a captured
// launch sequence then serves every registration of its shape, like a single-GPU one)
constexpr size_t kP2pSumWords = (size_t)kP2pSlots * MADICP_MAX_BATCH * kMaxRanks * kRowGranules;
__host__ __device__ constexpr size_t p2p_row(int slot, int scan, int q) {
  return (((size_t)slot * MADICP_MAX_BATCH + (size_t)scan) * kMaxRanks + (size_t)q) * kRowGranules;
}
// Behind the sums: the matched flags of a registration's last round (mad_icp.cpp:85; a leaf is an inlier if ANY keyframe on ANY
// rank matched it, pipeline.cpp:197-204), 32 flags per tagged 8-byte word, one row per (registration parity, scan, sending
// rank).  Two parities are enough: no rank can finish registration e + 1 before every rank has started its icp_final — i.e.
// has left registration e's behind, flags read — so a row of parity e & 1 is never rewritten (by e + 2) while it is still unread.
constexpr int kP2pFlagLeaves = 131072;  // moving sets beyond this OR their flags through the communicator, as before
constexpr int kP2pFlagWords = kP2pFlagLeaves / 32;
__host__ __device__ constexpr size_t p2p_flag_row(int parity, int scan, int q) {
  return kP2pSumWords + (((size_t)parity * MADICP_MAX_BATCH + (size_t)scan) * kMaxRanks + (size_t)q) * kP2pFlagWords;
}
constexpr size_t kP2pBoxWords = kP2pSumWords + (size_t)2 * MADICP_MAX_BATCH * kMaxRanks * kP2pFlagWords;
__device__ __forceinline__ void granule_store_sys(gptr_g64 g, unsigned tag, double v) {
  const unsigned long long t = (unsigned long long)tag << 32;
  __hip_atomic_store(g, t | (unsigned)__double2loint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __hip_atomic_store(g + 1, t | (unsigned)__double2hiint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ bool granule_try_sys(gptr_g64 g, unsigned tag, double& v) {
  const unsigned long long a = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  const unsigned long long b = __hip_atomic_load(g + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  v = __hiloint2double((int)(unsigned)b, (int)(unsigned)a);
  return (unsigned)(a >> 32) == tag && (unsigned)(b >> 32) == tag;
}
// Wave 0 of a workgroup (all 64 lanes): total[] (LDS) holds this rank's adders of round `r_done` of mailbox scan row `scan`;
// on return it holds the sum over the ranks, in rank order.  `publish`: this workgroup is the one that sends the rank's row.
// `err_word` (Job::error): once ANY workgroup of the registration has given up on a peer — in this launch or an earlier one —
// nobody waits out the bound again (a lost peer costs the registration one comm_timeout_ms, not one per workgroup and round).
__device__ __forceinline__ bool p2p_exchange(const PeerBox& pb, unsigned epoch, int scan, int r_done, bool publish,
                                             double* total /*LDS kAcc*/, int* err_word) {
  const int lane = threadIdx.x & 63;
  const unsigned tag = (epoch << 8) + (unsigned)r_done + 1u;
  const int slot = (int)((epoch & 1u) << 1) | (r_done & 1);
  const double mine = lane < kAcc ? total[lane] : 0.0;
  if (publish && lane < kAcc) {
    for (int q = 0; q < pb.n_ranks; ++q)
      if (q != pb.rank) granule_store_sys((gptr_g64)(uintptr_t)pb.box[q] + p2p_row(slot, scan, pb.rank) + 2 * lane, tag, mine);
  }
  bool ok = pb.n_ranks <= 1 || __hip_atomic_load(err_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0;
  double tot = 0.0;
  const unsigned long long t0 = wall_clock64();
  for (int q = 0; q < pb.n_ranks; ++q) {
    double v = mine;
    if (q != pb.rank && lane < kAcc && ok) {
      gptr_g64 g = (gptr_g64)(uintptr_t)pb.box[pb.rank] + p2p_row(slot, scan, q) + 2 * lane;
      for (unsigned spins = 1; !granule_try_sys(g, tag, v); ++spins) {
        if ((spins & 63u) == 0u && (wall_clock64() - t0 > pb.spin_ticks ||
                                    __hip_atomic_load(err_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
          ok = false;
          break;
        }
        __builtin_amdgcn_s_sleep(2);
      }
    }
    tot = (q == 0) ? v : tot + v;
  }
  wave_lds_order();  // (every lane has read total[])
  if (lane < kAcc) total[lane] = tot;
  wave_lds_order();
  return __all(ok);
}

// Job::error = 4 (a peer was lost), unless the registration already carries an error
__device__ __forceinline__ void p2p_mark_lost(int* err_word) {
  if (__hip_atomic_load(err_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0)
    __hip_atomic_store(err_word, 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- the matched flags over the mailboxes (icp_final, whole workgroup) ----
// 32 flag bytes (0 / 1) of the moving set <-> one 32-bit payload.  `matched` is 256-byte aligned and its capacity a multiple of
// 256 bytes, so the two 16-byte loads of a word never leave the allocation; bytes at and beyond L are not flags.
__device__ __forceinline__ unsigned p2p_pack_nibble(unsigned x) { return (x | (x >> 7) | (x >> 14) | (x >> 21)) & 0xfu; }
__device__ __forceinline__ unsigned p2p_unpack_nibble(unsigned n) { return (n & 1u) | ((n & 2u) << 7) | ((n & 4u) << 14) | ((n & 8u) << 21); }
__device__ __forceinline__ unsigned p2p_pack32(const uint8_t* __restrict__ matched, int w, int L) {
  const uint4 a = reinterpret_cast<const uint4*>(matched)[2 * w], b = reinterpret_cast<const uint4*>(matched)[2 * w + 1];
  unsigned bits = p2p_pack_nibble(a.x) | (p2p_pack_nibble(a.y) << 4) | (p2p_pack_nibble(a.z) << 8) | (p2p_pack_nibble(a.w) << 12) |
                  (p2p_pack_nibble(b.x) << 16) | (p2p_pack_nibble(b.y) << 20) | (p2p_pack_nibble(b.z) << 24) | (p2p_pack_nibble(b.w) << 28);
  const int left = L - 32 * w;  // flags in this word
  if (left < 32) bits &= (1u << left) - 1u;
  return bits;
}
// this rank's flags -> its row in every peer's mailbox (stores only: they are in flight while wave 0 exchanges the sums)
__device__ __forceinline__ void p2p_flags_publish(const PeerBox& pb, unsigned epoch, int scan, const uint8_t* __restrict__ matched, int L) {
  const unsigned long long t = (unsigned long long)((epoch << 8) + 255u) << 32;
  const int W = (L + 31) >> 5;
  for (int w = threadIdx.x; w < W; w += blockDim.x) {
    const unsigned long long word = t | p2p_pack32(matched, w, L);
    for (int q = 0; q < pb.n_ranks; ++q)
      if (q != pb.rank)
        __hip_atomic_store((gptr_g64)(uintptr_t)pb.box[q] + p2p_flag_row((int)(epoch & 1u), scan, pb.rank) + w, word, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
// the peers' rows <- the own mailbox, OR-ed into `matched` (every rank ends with the same flags: OR has no order).  False when
// a peer's row never came.  Ends with an agent-scope release by every thread: icp_publish reads `matched` from another XCD.
__device__ __forceinline__ bool p2p_flags_join(const PeerBox& pb, unsigned epoch, int scan, uint8_t* __restrict__ matched, int L, int* err_word) {
  const unsigned tag = (epoch << 8) + 255u;
  const int W = (L + 31) >> 5;
  bool ok = __hip_atomic_load(err_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0;
  const unsigned long long t0 = wall_clock64();
  for (int w = threadIdx.x; w < W && ok; w += blockDim.x) {
    const unsigned own = p2p_pack32(matched, w, L);
    unsigned bits = own;
    for (int q = 0; q < pb.n_ranks && ok; ++q) {
      if (q == pb.rank) continue;
      gptr_g64 g = (gptr_g64)(uintptr_t)pb.box[pb.rank] + p2p_flag_row((int)(epoch & 1u), scan, q) + w;
      unsigned long long v = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      for (unsigned spins = 1; (unsigned)(v >> 32) != tag; ++spins) {
        if ((spins & 63u) == 0u && (wall_clock64() - t0 > pb.spin_ticks ||
                                    __hip_atomic_load(err_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
          ok = false;
          break;
        }
        __builtin_amdgcn_s_sleep(2);
        v = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      bits |= (unsigned)v;
    }
    if (ok && bits != own) {
      if (L - 32 * w >= 32) {
        uint4 a, b;
        a.x = p2p_unpack_nibble(bits & 15u); a.y = p2p_unpack_nibble((bits >> 4) & 15u);
        a.z = p2p_unpack_nibble((bits >> 8) & 15u); a.w = p2p_unpack_nibble((bits >> 12) & 15u);
        b.x = p2p_unpack_nibble((bits >> 16) & 15u); b.y = p2p_unpack_nibble((bits >> 20) & 15u);
        b.z = p2p_unpack_nibble((bits >> 24) & 15u); b.w = p2p_unpack_nibble(bits >> 28);
        reinterpret_cast<uint4*>(matched)[2 * w] = a;
        reinterpret_cast<uint4*>(matched)[2 * w + 1] = b;
      } else {
        for (int i = 32 * w; i < L; ++i) matched[i] = (uint8_t)((bits >> (i - 32 * w)) & 1u);
      }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  return ok;
}

// TRACE: also write the per-pair correspondence trace (Job::corr) — the single-round debugging entry point only
// FOLD (option "xcd_fold", an experiment kept for its measurement — profiles/r3_j_xcd_fold.md): the XCD-hierarchical join
// WITHOUT the persistent launch.  Every workgroup publishes its row as granules; the leader of each group x = blockIdx.x & 7
// waits for the group's rows at the END of the launch, folds them in the canonical order and publishes F[x]; the next
// launch's wave 0 reads eight rows (3.8 KB) instead of every workgroup joining 256 (61 KB).
// TAIL (multi-GPU, keyframe sharding): the launch itself leaves this rank's share of the round's adders in
// `totals_out` — what a separate icp_reduce launch did — so that the all-reduce can be enqueued directly behind the round.
// Every workgroup publishes its row as tagged granules (like FOLD: the tag, not an ordering of stores, says when a row is
// there) and takes a ticket; the workgroup that draws the last one folds the scan's rows in the canonical order (stage 1 by
// all its threads, stage 2 by wave 0: the bits of icp_reduce) and resets the ticket.  `totals` (the reduced sums of the
// PREVIOUS round, read in the prologue) and `totals_out` are the two parity halves of one buffer, never the same memory.
// QUEUE (= DEEP): the variant with the leaf-major rounds compiled in (icp_leaf_major.inc.h) — launched only for the geometry
// they are for (a batch sharing the chip: pick_geometry); code of that size in the one-scan kernel costs its launch 0.45 us
// by its mere presence (measured with the first form of the queue, profiles/r5_e_ab.md)
// P2P (multi-GPU, option "shard_p2p"): the join over the ranks happens INSIDE the prologue, over the peer-mapped mailboxes
// (p2p_exchange above) — the launch sequence of a sharded registration is then the single-GPU one.
template <int QPT, bool TRACE, bool FOLD = false, bool TAIL = false, bool QUEUE = false, bool P2P = false>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(3))) void icp_round(
    const Job* __restrict__ jobs, Job* __restrict__ jobs_out, double* __restrict__ partials,
    const double* __restrict__ totals, int round, int n_iters, int K, int RPT, unsigned long long* __restrict__ xch,
    double* __restrict__ totals_out, unsigned int* __restrict__ tickets, const PeerBox pb) {
  // (n_iters, K and ranges_per_tree are the same for every scan of the launch: as kernel arguments they are known
  // one memory round trip before anything read through `job`)
  // `jobs` and `jobs_out` are the SAME array: everything this kernel reads goes through the const restrict view, the
  // few fields workgroup 0 records (H, b, counters, the other pose slot) through the other — never the same element
  // through both.  With one pointer the first store made every later read of the Job (tree descriptors, options) a
  // vector load + wait + readfirstlane instead of a scalar load: three serialized round trips at the top of the unit
  // loop alone.
  const Job* job = jobs + blockIdx.y;
  Job* jout = jobs_out + blockIdx.y;
  MADICP_STAMP(0);
  typedef const __attribute__((address_space(1))) unsigned int* gptr_u1;
  typedef const __attribute__((address_space(1))) float* gptr_f1;
  typedef const __attribute__((address_space(1))) double* gptr_d1;
  // Loads whose addresses need nothing but the launch geometry go first: the previous round's partials (joined by
  // every workgroup below) and — wave 0 only, it is the one that solves — the pose that round linearised at.
  // (Measured alternatives, all slower: every lane loading the pose, which turns it into scalar loads that the rest
  // of the scalar traffic then queues behind; requesting all of the Job's scalars in one pinned batch.)
  const int prows = join_rows(gridDim.x);                          // rows stored per scan (>= gridDim.x, zero padded)
  const long long pstride = (long long)gridDim.y * prows * kAcc;  // one parity's worth of partials
  const double* __restrict__ prev_partials = partials + ((round - 1) & 1) * pstride + (long long)blockIdx.y * prows * kAcc;
  double Xp[12], wear_prev[2] = {0.0, 0.0};
  JoinLoads jl;
  if (!FOLD && round > 0 && !totals) join_issue(prev_partials, jl);
  // FOLD: the eight folded rows of the previous round (level 2 of the exchange), four values per lane of wave 0
  double fv[4] = {0.0, 0.0, 0.0, 0.0};
  bool fold_ok = true;
  if (FOLD && round > 0 && threadIdx.x < 60) {
    const unsigned tag_prev = (job->epoch << 8) + (unsigned)round;
    gptr_g64 src = (gptr_g64)(uintptr_t)xch + xch_level1(gridDim.y, gridDim.x) +
                   ((size_t)((round - 1) & 1) * gridDim.y + blockIdx.y) * kFoldGroups * kRowGranules;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int idx = threadIdx.x + 60 * j;
      const int x = idx / kAcc, c = idx - x * kAcc;
      fold_ok &= granule_try(src + x * kRowGranules + 2 * c, tag_prev, fv[j]);
    }
  }
  MADICP_STAMP(14);
  // walk hint: how many lanes of THIS workgroup had to walk in the previous round (written at the end of that round,
  // behind the two partial buffers); decides — without a vote, i.e. without a barrier per pass — whether the tree's
  // top levels are worth staging into LDS.  Speed only: the descent gives the same result from LDS or from global.
  double* __restrict__ hints = partials + 2 * pstride;
  const long long hint_slot = (long long)blockIdx.y * gridDim.x + blockIdx.x;
  const long long hint_stride = (long long)gridDim.y * gridDim.x;
  double prev_hint = 1.0;
  if (threadIdx.x < 64) {  // wave 0 solves (every lane the same values)
    gptr_d1 xr = (gptr_d1)(uintptr_t)job->Xring[(round > 0 ? round - 1 : 0) & 1];
#pragma unroll
    for (int i = 0; i < 12; ++i) Xp[i] = xr[i];
    if (round > 0) {
      prev_hint = ((gptr_d1)(uintptr_t)hints)[((round - 1) & 1) * hint_stride + hint_slot];
      gptr_d1 wr = (gptr_d1)(uintptr_t)job->wear_ring[(round - 1) & 1];
      wear_prev[0] = wr[0];
      wear_prev[1] = wr[1];
    }
  }
  MADICP_STAMP(15);
  const int U = K * RPT;  // <= MADICP_MAX_TREES x workgroups: 32-bit arithmetic (a 64-bit division is ~100 instructions)
  const int xcd = blockIdx.x & 7;
  const int slot = blockIdx.x >> 3;
  const int nslots = gridDim.x >> 3;
  const int lo = (xcd * U) >> 3;
  const int hi = ((xcd + 1) * U) >> 3;
  const int u_first = lo + slot;
  const bool have_first = u_first < hi;
  const int k_first = have_first ? u_first / RPT : 0;
  const int r_first = have_first ? u_first - k_first * RPT : 0;

  // The descriptor of the first unit's tree goes to LDS (11 lanes x 8 bytes, stored before the barrier that ends the
  // prologue), the two staging options to registers — all read BEFORE the first barrier: after one, the compiler
  // must treat global memory as clobbered and turns every uniform read of the Job into a vector load + full
  // vmcnt wait + readfirstlane, which is what the pass used to do for every field of the descriptor it touched.
  __shared__ __attribute__((aligned(16))) TreeDesc s_td;
  static_assert(sizeof(TreeDesc) == 88, "descriptor copy below moves 11 x 8 bytes");
  long long td_word = 0;
  if (threadIdx.x < 11)
    td_word = ((const __attribute__((address_space(1))) long long*)(uintptr_t)&job->trees[k_first])[threadIdx.x];
  const int opt_lds_top = job->lds_top;
  const int opt_stage_min = job->stage_min_leaves;
  const int L = job->L;
  const int flags = job->flags;
  const unsigned p2p_epoch = P2P ? job->p2p_epoch : 0u;
  constexpr bool DEEP = QUEUE;
  const int opt_leaf_major = DEEP ? job->queue_nodes : 0;
  __shared__ unsigned short s_queue[DEEP ? kWaves : 1][DEEP ? kQueueCap : 1];  // leaf-major rounds: queued walkers per wavefront
  // ... and the pairs that keep their leaf and have to be evaluated (pass | tree | lane, leaf ordinal, threshold on file): evaluated
  // densely, 64 at a time, instead of where they stand with a fifth of the lanes (icp_leaf_major.inc.h)
  __shared__ unsigned short s_eq_id[DEEP ? kWaves : 1][DEEP ? kEvalCap : 1];
  __shared__ unsigned int s_eq_leaf[DEEP ? kWaves : 1][DEEP ? kEvalCap : 1];
  __shared__ float s_eq_t[DEEP ? kWaves : 1][DEEP ? kEvalCap : 1];
  __shared__ __attribute__((aligned(16))) TreeDesc s_tds[DEEP ? kDeepTrees : 1];  // ... and the descriptors of the workgroup's trees
  const bool last_round = (round == n_iters - 1);
  const bool mark_matched = last_round || (flags & kFlagMatchAll);
  const double* __restrict__ moving = job->moving;
  uint8_t* __restrict__ matched = job->matched;
  uint32_t* __restrict__ corr = TRACE ? job->corr : nullptr;
  const double min_ball = job->min_ball, rho = job->rho, b_ratio = job->b_ratio;
#ifndef MADICP_EXACT_SOLVE
  const double inv_min_ball = fast_rcp(min_ball);
#endif
  uint32_t* __restrict__ cache_leaf = job->cache_leaf;
  float* __restrict__ cache_margin = job->cache_margin;
  const bool reuse = cache_leaf != nullptr && round > 0 && !(flags & kFlagNoReuse);
  const bool gate_file = !(flags & kFlagNoGateReuse);  // rejected pairs are filed with min(leaf's, slack's) threshold, negative
  const bool gate_reuse = reuse && gate_file;
  // Ranges.  Contiguous: range r is leaves [r S, (r + 1) S).  Interleaved (kFlagInterleave; round 6): the scan's groups of 64
  // consecutive leaves are DEALT over the ranges — group g of range r is the scan's group g RPT + r — because a contiguous stretch
  // of the leaf order is a stretch of space, and stretches differ by a factor of four in how many of their pairs pass the gate
  // (behind the sensor every keyframe of the map answers, ahead of it only the last few: BASELINE configs[4], 7.6 k .. 30.9 k
  // accepted pairs per range over 16 sampled keyframes): the launch waited for the workgroups that drew the busy stretch.
  // The loops run over a range's VIRTUAL indices v in [r S, (r + 1) S), S a multiple of 64, and phys(r, v) is the leaf behind v
  // (contiguous: v itself); a virtual index is valid when its leaf exists.  Everything indexed by leaf — coordinates, cached
  // correspondence, matched flags, trace — goes by the leaf; which workgroup adds which pairs changes, hence the order of the
  // sums: H and b differ from the contiguous launch in their last bits, every decision is the same.
  const bool inter = (flags & kFlagInterleave) != 0;
  const int S = inter ? ((((L + 63) >> 6) + RPT - 1) / RPT) << 6 : (L + RPT - 1) / RPT;  // leaves per range
  const int Lv = inter ? RPT * S : L;
  auto phys = [&](int r_, int v) -> int {
    const int n = v - r_ * S;
    return inter ? ((((n >> 6) * RPT + r_) << 6) | (n & 63)) : v;
  };

  // The first pass's loads that do not depend on the pose (leaf coordinates, cached correspondence) are issued NOW,
  // so they are in flight while the workgroup joins and solves the previous round.  Branch-free (clamped index): a
  // divergent region would wait for its loads where it ends.  (Also touching the second pass's lines and the cached
  // leaf records here was measured: no gain.)
  vd4 pv0[QPT];
  float cmar0[QPT];
  unsigned int cword0[QPT];
  {
    const int i_end = have_first ? min(Lv, (r_first + 1) * S) : 0;
    const int i_last = max(i_end - 1, 0);
#pragma unroll
    for (int j = 0; j < QPT; ++j) {
      const int i = max(min(phys(r_first, min(r_first * S + j * kBlock + (int)threadIdx.x, i_last)), L - 1), 0);
      pv0[j] = ((gptr_d4)(uintptr_t)moving)[i];
      cmar0[j] = 0.f;
      cword0[j] = 0u;
      if (reuse) {  // (uniform)
        const long long ci = (long long)k_first * L + i;
        cmar0[j] = ((gptr_f1)(uintptr_t)cache_margin)[ci];
        cword0[j] = ((gptr_u1)(uintptr_t)cache_leaf)[ci];
      }
    }
  }

  // ---- the solve of the previous round, by every workgroup (its wave 0) ------------------------------------
  __shared__ double s_total[kAcc];
  __shared__ double s_X[15];  // X_round (12), bounds of the last update's rotation angle and translation, walk hint
  __shared__ JoinSeg s_seg;
  if (threadIdx.x < 11) reinterpret_cast<long long*>(&s_td)[threadIdx.x] = td_word;
  int desc_tree = k_first;
  MADICP_STAMP(13);
  // top-level copy: dynamic LDS, present only when the host launched with kTopLdsBytes (units big enough to pay
  // for the copy)
  extern __shared__ __attribute__((aligned(16))) char dyn_lds[];
  vu4* s_top = reinterpret_cast<vu4*>(dyn_lds);
  int4* s_exit = reinterpret_cast<int4*>(dyn_lds + kTopMax * sizeof(vu4));
  int staged_tree = -1;
  __shared__ double s_hint;
  if (round > 0 && !totals) {
    if (threadIdx.x == 0) s_hint = prev_hint;
    if (FOLD) {
      if (threadIdx.x < 60) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int idx = threadIdx.x + 60 * j;
          const int x = idx / kAcc, c = idx - x * kAcc;
          s_seg[x][c] = fv[j];
        }
      }
      if (threadIdx.x < 64 && !__all(fold_ok) && threadIdx.x == 0) jout->error = 3;  // (a leader gave up: its row never came)
      __syncthreads();
    } else {
      join_stage1(prev_partials, prows, jl, s_seg);
    }
    // The barrier above also published the first unit's descriptor and the walk hint.  While wave 0 solves (2-3 us),
    // the other eleven waves have nothing to do: if lanes of this workgroup had to walk last round they copy the
    // tree's top levels into LDS NOW instead of after the prologue (~1 us of every walking round).
    const int unit_len = have_first ? min(Lv, (r_first + 1) * S) - r_first * S : 0;
    const int n_top_first = (opt_lds_top && unit_len >= opt_stage_min) ? min(s_td.n_top, kTopMax) : 0;
    if (s_hint > 0.0 && n_top_first > 0) {  // (workgroup-uniform)
      if (threadIdx.x >= 64) {
        gptr_u4 gt = (gptr_u4)(uintptr_t)s_td.top;
        gptr_u4 ge = (gptr_u4)(uintptr_t)s_td.top_exit;
        for (int e = threadIdx.x - 64; e < n_top_first; e += kBlock - 64) {
          s_top[e] = gt[e];
          reinterpret_cast<vu4*>(s_exit)[e] = ge[e];
        }
      }
      staged_tree = k_first;  // visible to everybody after the barrier that ends the prologue
    }
  }
  if (threadIdx.x < 64) {
    double Xn[12];
    double moved[2] = {0.0, 0.0};
    if (round > 0) {
      if (totals) {
        if (threadIdx.x < kAcc) s_total[threadIdx.x] = totals[blockIdx.y * kAcc + threadIdx.x];
        wave_lds_order();
      } else if (FOLD) {
        if (threadIdx.x < kAcc) {
          double a = s_seg[0][threadIdx.x];
#pragma unroll
          for (int x = 1; x < kFoldGroups; ++x) a += s_seg[x][threadIdx.x];
          s_total[threadIdx.x] = a;
        }
        wave_lds_order();
      } else {
        join_stage2_wave0(s_seg, s_total);
      }
      if (P2P) {  // this rank's adders -> every peer's mailbox; the other ranks' rows <- the own mailbox; sum in rank order
        if (!p2p_exchange(pb, p2p_epoch, pb.scan0 + (int)blockIdx.y, round - 1, blockIdx.x == 0, s_total, &jout->error) && threadIdx.x == 0)
          p2p_mark_lost(&jout->error);
      }
      MADICP_STAMP(1);
      double H[36], b[6];
      solve_pose(s_total, Xp, !(flags & kFlagNoUpdate), Xn, H, b, moved, threadIdx.x & 63);
      if (blockIdx.x == 0 && threadIdx.x == 0) {  // bookkeeping of the finished round, once per scan
#pragma unroll
        for (int i = 0; i < 36; ++i) jout->H[i] = H[i];
#pragma unroll
        for (int i = 0; i < 6; ++i) jout->b[i] = b[i];
        jout->n_pairs = s_total[27];
        jout->visits += static_cast<unsigned long long>(s_total[28]);
        jout->walked += static_cast<unsigned long long>(s_total[29]);
#pragma unroll
        for (int i = 0; i < 12; ++i) jout->Xring[round & 1][i] = Xn[i];
        jout->iter = round;
      }
    } else {
      MADICP_STAMP(1);
#pragma unroll
      for (int i = 0; i < 12; ++i) Xn[i] = Xp[i];
    }
    // cumulative wear of the correspondence / gate thresholds up to THIS round's pose ("Bookkeeping without a store")
    double wear_a = 0.0, wear_b = 0.0;
    if (round > 0) {
      wear_a = wear_prev[0] + (moved[0] * (1.0 + 1e-12) + 1.75e-11);
      wear_b = wear_prev[1] + (moved[1] * (1.0 + 1e-12) + 1e-11 * (fabs(Xn[9]) + fabs(Xn[10]) + fabs(Xn[11])));
    }
    if (threadIdx.x == 0) {
      if (blockIdx.x == 0) {
        jout->wear_ring[round & 1][0] = wear_a;
        jout->wear_ring[round & 1][1] = wear_b;
      }
      if (blockIdx.x == 0 && job->x_iters) {
#pragma unroll
        for (int i = 0; i < 12; ++i) job->x_iters[(long long)round * 12 + i] = Xn[i];  // the pose this round linearises at
      }
#pragma unroll
      for (int i = 0; i < 12; ++i) s_X[i] = Xn[i];
      s_X[12] = wear_a;
      s_X[13] = wear_b;
      s_X[14] = prev_hint;
    }
  }
  // Pose-independent set-up goes here, BEFORE the barrier that ends the prologue: eleven of the twelve waves reach this
  // point ~2 us before wave 0 has solved, and whatever they do now they do for free.
  // the matched_ flags are cleared before the last round (pipeline.cpp:172-176); all workgroups share the work
  if (round == n_iters - 2 && !(flags & kFlagMatchAll)) {
    uint4* m16 = reinterpret_cast<uint4*>(matched);  // hipMalloc'ed: 256-byte aligned
    const int stride = gridDim.x * blockDim.x;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < (L >> 4); i += stride) m16[i] = make_uint4(0, 0, 0, 0);
    if (blockIdx.x == 0)
      for (int i = (L & ~15) + threadIdx.x; i < L; i += blockDim.x) matched[i] = 0;
  }
  double acc[kAcc];
#pragma unroll
  for (int v = 0; v < kAcc; ++v) acc[v] = 0.0;
  unsigned int visits = 0, walked_visits = 0;
  bool walked = false;
  __syncthreads();
  MADICP_STAMP(2);
  double R[9], t[3];
#pragma unroll
  for (int k = 0; k < 9; ++k) R[k] = wave_uniform(s_X[k]);
#pragma unroll
  for (int k = 0; k < 3; ++k) t[k] = wave_uniform(s_X[9 + k]);
  const double wear_alpha = wave_uniform(s_X[12]), wear_beta = wave_uniform(s_X[13]);
  const double hint_nodes = wave_uniform(s_X[14]);  // nodes this workgroup walked in the previous round
  const bool stage_hint = round == 0 || hint_nodes > 0.0;
  // passes this workgroup makes in a round (every unit the same length but the last range of a tree)
  const int wg_passes = have_first ? ((hi - u_first + nslots - 1) / nslots) * ((S + kBlock - 1) / kBlock) : 0;
  // leaf-major round (DEEP launches, icp_leaf_major.inc.h)?  Workgroup-uniform, no vote: the geometry gives every unit of the
  // workgroup the same range, the trees and passes fit a queue entry, and last round the workgroup walked few nodes per pass
  const int wg_units = have_first ? (hi - u_first + nslots - 1) / nslots : 0;
  const bool leaf_major = DEEP && reuse && round >= 1 && opt_leaf_major > 0 && RPT == nslots && wg_units >= 1 &&
                          wg_units <= kDeepTrees && (S + kBlock - 1) / kBlock <= kDeepPasses &&
                          hint_nodes < (double)opt_leaf_major * (double)wg_passes;

#define MADICP_TID threadIdx.x
#include "icp_leaf_major.inc.h"  // (ends in `else`: the tree-major body below is the other branch)
  {
#include "icp_linearize_body.inc.h"
  }
#undef MADICP_TID

  MADICP_STAMP(5);
  // deterministic reduction: lanes (halving butterfly) -> waves (LDS, fixed order) -> partial
  acc[28] = static_cast<double>(visits);  // integer-valued: its sums are exact in any order
  acc[29] = static_cast<double>(walked_visits);
  __shared__ double red[kWaves][32];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  wave_reduce_scatter(acc, lane, red[wave]);
  const int n_walked = __syncthreads_count(walked ? 1 : 0);
  const unsigned tag_now = (FOLD || TAIL) ? (job->epoch << 8) + (unsigned)round + 1u : 0u;
  if (threadIdx.x < kAcc) {
    double s = red[0][threadIdx.x];
#pragma unroll
    for (int w = 1; w < kWaves; ++w) s += red[w][threadIdx.x];
    if (FOLD || TAIL)
      granule_store((gptr_g64)(uintptr_t)xch + ((size_t)((round & 1) * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * kRowGranules +
                        2 * threadIdx.x, tag_now, s);
    else
      partials[(round & 1) * pstride + ((long long)blockIdx.y * prows + blockIdx.x) * kAcc + threadIdx.x] = s;
    // the walk hint of the next round: the nodes this workgroup walked (> 0: stage the tree's top; per pass: queue the walks)
    if (threadIdx.x == 29) hints[(round & 1) * hint_stride + hint_slot] = fmax(s, static_cast<double>(n_walked));
  }
  MADICP_STAMP(6);
  if (TAIL) {
    __shared__ int s_last;
    if (threadIdx.x == 0) {  // (this wave issued the row's granules before the ticket; their tags tell the folder when they are there)
      const unsigned t = __hip_atomic_fetch_add(&tickets[blockIdx.y], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_last = (t + 1u == gridDim.x) ? 1 : 0;
    }
    __syncthreads();
    if (s_last) {  // (workgroup-uniform) every other workgroup of this scan has published its row
      gptr_g64 src = (gptr_g64)(uintptr_t)xch + ((size_t)((round & 1) * gridDim.y + blockIdx.y) * gridDim.x) * kRowGranules;
      bool expired = false;
      const unsigned long long t_start = wall_clock64();
      // stage 1 of the canonical order: seg[G] = 0 + row[G] + row[G + 48] + ... (rows no workgroup wrote are the zero rows
      // of the padded layout: adding +0.0 is exact, skipping them gives the same bits)
      for (int idx = threadIdx.x; idx < kJoinGroups * kAcc; idx += kBlock) {
        const int G = idx / kAcc, c = idx - G * kAcc;
        // the group's first kJoinRows rows (all of them up to 288 workgroups) are requested TOGETHER — one round trip to
        // the other XCDs' rows, not one per row — and polled again only where a tag was not there yet
        double v[kJoinRows];
        bool ok[kJoinRows];
#pragma unroll
        for (int i = 0; i < kJoinRows; ++i) {
          v[i] = 0.0;
          ok[i] = G + kJoinGroups * i >= (int)gridDim.x;  // (a row nobody wrote: +0.0)
        }
        for (unsigned spins = 1;; ++spins) {
          bool all = true;
#pragma unroll
          for (int i = 0; i < kJoinRows; ++i)
            if (!ok[i]) {
              ok[i] = granule_try(src + (size_t)(G + kJoinGroups * i) * kRowGranules + 2 * c, tag_now, v[i]);
              all = all && ok[i];
            }
          if (all) break;
          if ((spins & 63u) == 0u && wall_clock64() - t_start > kSpinLimitTicks) { expired = true; break; }
          __builtin_amdgcn_s_sleep(1);
        }
        double a = 0.0;
#pragma unroll
        for (int i = 0; i < kJoinRows; ++i) a += v[i];
        for (int row = G + kJoinGroups * kJoinRows; row < (int)gridDim.x; row += kJoinGroups) {  // (more than 288 workgroups only)
          gptr_g64 g = src + (size_t)row * kRowGranules + 2 * c;
          double w = 0.0;
          for (unsigned spins = 1; !granule_try(g, tag_now, w); ++spins) {
            if ((spins & 63u) == 0u && wall_clock64() - t_start > kSpinLimitTicks) { expired = true; break; }
            __builtin_amdgcn_s_sleep(1);
          }
          a += w;
        }
        s_seg[G][c] = a;
      }
      if (__syncthreads_or(expired ? 1 : 0)) {
        if (threadIdx.x == 0) jout->error = 1;
      } else if (threadIdx.x < 64) {
        join_stage2_wave0(s_seg, s_total);
        if (threadIdx.x < kAcc) totals_out[blockIdx.y * kAcc + threadIdx.x] = s_total[threadIdx.x];
      }
      if (threadIdx.x == 0) tickets[blockIdx.y] = 0u;  // for the next launch (every workgroup of this one has drawn)
    }
  }
  if (FOLD && slot == 0) {  // (workgroup-uniform) the leader of group `xcd`: wait for the group's rows, fold, publish F[x]
    gptr_g64 src = (gptr_g64)(uintptr_t)xch + ((size_t)((round & 1) * gridDim.y + blockIdx.y) * gridDim.x + xcd) * kRowGranules;
    bool expired = false;
    const unsigned long long t_start = wall_clock64();
    for (int idx = threadIdx.x; idx < nslots * kAcc; idx += kBlock) {
      const int sl = idx / kAcc, c = idx - sl * kAcc;
      gptr_g64 g = src + (size_t)sl * kFoldGroups * kRowGranules + 2 * c;  // row xcd + 8 sl
      double v = 0.0;
      for (unsigned spins = 1; !granule_try(g, tag_now, v); ++spins) {
        if ((spins & 63u) == 0u && wall_clock64() - t_start > kSpinLimitTicks) { expired = true; break; }
        __builtin_amdgcn_s_sleep(1);
      }
      s_seg[sl][c] = v;
    }
    if (__syncthreads_or(expired ? 1 : 0)) {
      if (threadIdx.x == 0) jout->error = 1;
    } else if (threadIdx.x < kAcc) {
      double rv[kJoinGroups];
#pragma unroll
      for (int sl = 0; sl < kJoinGroups; ++sl) rv[sl] = sl < nslots ? s_seg[sl][threadIdx.x] : 0.0;
      double f = 0.0;
#pragma unroll
      for (int q = 0; q < kFoldChains; ++q) {
        double sg = 0.0;
#pragma unroll
        for (int sl = q; sl < kJoinGroups; sl += kFoldChains) sg += rv[sl];
        f = (q == 0) ? sg : f + sg;
      }
      granule_store((gptr_g64)(uintptr_t)xch + xch_level1(gridDim.y, gridDim.x) +
                        ((size_t)((round & 1) * gridDim.y + blockIdx.y) * kFoldGroups + xcd) * kRowGranules + 2 * threadIdx.x,
                    tag_now, f);
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// ALL Gauss-Newton rounds of a registration as ONE launch (single GPU, every workgroup resident: one per CU).
//
// What a kernel boundary per round costs icp_round (stamps): ~1.3 us between two dependent launches, then
// ~2.7 us until the 256 partial rows of the previous round (61 KB per reader, seven eighths of them from other XCDs) have
// arrived, with every pose-independent quantity (leaf coordinates, options, descriptors, the LDS copy of the tree's top)
// fetched again.  Here the workgroups stay resident and exchange their adders inside the launch, in two levels that
// are exactly the two levels of join_stage2_wave0's summation order:
//   level 1  every workgroup publishes its row of kAcc sums; the LEADER of its group x = blockIdx.x & 7 (the workgroup
//            with blockIdx.x == x) folds the group's rows in the fixed order  F[x] = sum_q (0 + sum_i row[x + 8(q + 6i)])
//   level 2  the leaders publish F[x]; wave 0 of EVERY workgroup reads the eight F[x] (3.8 KB instead of 61 KB), adds
//            them in order and solves — all workgroups get the bit-identical pose, as in icp_round.
// Transport (the guide's "data IS the flag" form for payloads <= 4 KB): every double travels as two 8-byte granules
// {tag, 32 data bits}, each written by ONE relaxed agent-scope atomic store (write-through, sc1) and polled with relaxed
// agent-scope atomic loads until the tag matches; tag = (Job::epoch << 8) + round + 1, so a granule left by an earlier
// round, registration or launch geometry never matches and nothing has to be cleared between launches.  No fence, no
// counter, no acquire: the only words another workgroup reads are granules.  Rows are double-buffered by round parity;
// nobody can run two rounds ahead of a reader (publishing round r + 1 needs every workgroup's row of round r, hence
// everybody past their reads of round r - 1).  Correctness never depends on which CU or XCD a workgroup runs on; the
// group = XCD coincidence only makes level 1 cheap.  Every spin is bounded by the 100 MHz wall clock: a launch whose
// workgroups are not all resident ends with Job::error set instead of hanging.
// The matched_ flags are cleared with write-through stores before the first row is published (mad_icp.cpp:85 sets
// them in the last round only, from other workgroups; pipeline.cpp:172-176).
// After the last round the leaders fold once more and icp_final (next launch) reads the eight F[x] of round n - 1.
// ---------------------------------------------------------------------------------------------------
template <int QPT>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(3))) void icp_persist(
    const Job* __restrict__ jobs, Job* __restrict__ jobs_out, unsigned long long* __restrict__ xch, int n_iters, int K_arg,
    int RPT_arg) {
  constexpr bool TRACE = false;
  typedef const __attribute__((address_space(1))) unsigned int* gptr_u1;
  typedef const __attribute__((address_space(1))) float* gptr_f1;
  typedef const __attribute__((address_space(1))) double* gptr_d1;

  __shared__ __attribute__((aligned(16))) TreeDesc s_td;
  __shared__ double s_total[kAcc];
  __shared__ double s_X[15];      // the pose of the current round + the bounds of the last update: lives here between rounds
  __shared__ JoinSeg s_rows;      // leader: the group's rows [slot][value]; everybody (wave 0): the eight F[x]
  __shared__ double red[kWaves][32];
  __shared__ unsigned long long s_cnt[2];  // visits / walked of the rounds joined so far
  __shared__ int s_abort;
  extern __shared__ __attribute__((aligned(16))) char dyn_lds[];
  vu4* s_top = reinterpret_cast<vu4*>(dyn_lds);
  int4* s_exit = reinterpret_cast<int4*>(dyn_lds + kTopMax * sizeof(vu4));

  int desc_tree, staged_tree = -1;
  bool stage_hint_next = true;
  {
    const Job* job = jobs + blockIdx.y;
    const int U = K_arg * RPT_arg;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int lo = (xcd * U) >> 3, hi = ((xcd + 1) * U) >> 3;
    const int u_first = lo + slot;
    const int k_first = u_first < hi ? u_first / RPT_arg : 0;
    if (threadIdx.x < 11)
      reinterpret_cast<long long*>(&s_td)[threadIdx.x] =
          ((const __attribute__((address_space(1))) long long*)(uintptr_t)&job->trees[k_first])[threadIdx.x];
    desc_tree = k_first;
    if (threadIdx.x < 12) s_X[threadIdx.x] = ((gptr_d1)(uintptr_t)job->Xring[0])[threadIdx.x];
    if (threadIdx.x >= 12 && threadIdx.x < 15) s_X[threadIdx.x] = 0.0;
    if (threadIdx.x < 2) s_cnt[threadIdx.x] = 0ull;
    if (threadIdx.x == 0) s_abort = 0;
    // matched_ flags: cleared here, write-through, before anything of this workgroup is published
    gptr_g64 m8 = (gptr_g64)(uintptr_t)job->matched;  // hipMalloc'ed: 256-byte aligned, padded by 16 bytes
    const int L = job->L;
    const int stride = gridDim.x * blockDim.x;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < ((L + 7) >> 3); i += stride)
      __hip_atomic_store(m8 + i, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the flag stores have left before the first row does)
  }
  __syncthreads();

  for (int round = 0; round < n_iters; ++round) {
    // Everything below is derived afresh every round from values the compiler cannot prove loop-invariant — exactly what
    // a launch of icp_round derives: left to itself the compiler hoists ~30 per-lane addresses and ~60 uniform values
    // out of the round loop and spills them (168 VGPRs + 270 bytes of scratch against 151 + 0).
    int tid = threadIdx.x, bx = blockIdx.x, by = blockIdx.y, K = K_arg, RPT = RPT_arg;
    const Job* jbase = jobs;
    asm volatile("" : "+v"(tid), "+s"(bx), "+s"(by), "+s"(K), "+s"(RPT), "+s"(jbase));
    const Job* job = jbase + by;
    const int U = K * RPT;
    const int xcd = bx & 7;
    const int slot = bx >> 3;
    const int nslots = gridDim.x >> 3;
    const int lo = (xcd * U) >> 3;
    const int hi = ((xcd + 1) * U) >> 3;
    const int u_first = lo + slot;
    const bool have_first = u_first < hi;
    const int k_first = have_first ? u_first / RPT : 0;
    const int r_first = have_first ? u_first - k_first * RPT : 0;
    const int opt_lds_top = job->lds_top;
    const int opt_stage_min = job->stage_min_leaves;
    const int L = job->L;
    const int flags = job->flags;
    const unsigned tag0 = job->epoch << 8;
    const double* __restrict__ moving = job->moving;
    uint8_t* __restrict__ matched = job->matched;
    uint32_t* __restrict__ corr = nullptr;
    const double min_ball = job->min_ball, rho = job->rho, b_ratio = job->b_ratio;
#ifndef MADICP_EXACT_SOLVE
    const double inv_min_ball = fast_rcp(min_ball);
#endif
    uint32_t* __restrict__ cache_leaf = job->cache_leaf;
    float* __restrict__ cache_margin = job->cache_margin;
    // (contiguous or dealt ranges: "Ranges" in icp_round)
    const bool inter = (flags & kFlagInterleave) != 0;
    const int S = inter ? ((((L + 63) >> 6) + RPT - 1) / RPT) << 6 : (L + RPT - 1) / RPT;
    const int Lv = inter ? RPT * S : L;
    auto phys = [&](int r_, int v) -> int {
      const int n = v - r_ * S;
      return inter ? ((((n >> 6) * RPT + r_) << 6) | (n & 63)) : v;
    };
    const bool leader = slot == 0;
    const bool last_round = (round == n_iters - 1);
    const bool mark_matched = last_round || (flags & kFlagMatchAll);
    const bool reuse = cache_leaf != nullptr && round > 0 && !(flags & kFlagNoReuse);
    const bool gate_file = !(flags & kFlagNoGateReuse);
    const bool gate_reuse = reuse && gate_file;
    const unsigned tag_prev = tag0 + (unsigned)round;  // rows of round - 1 carry (round - 1) + 1
    const unsigned tag_now = tag_prev + 1u;
    // exchange rows of this scan
    const size_t l1_stride = (size_t)gridDim.y * gridDim.x * kRowGranules;  // one parity of level 1
    const size_t l2_stride = (size_t)gridDim.y * kFoldGroups * kRowGranules;
    gptr_g64 l1 = (gptr_g64)(uintptr_t)xch + (size_t)by * gridDim.x * kRowGranules;
    gptr_g64 l2 = (gptr_g64)(uintptr_t)xch + xch_level1(gridDim.y, gridDim.x) + (size_t)by * kFoldGroups * kRowGranules;

    // the first pass's pose-independent loads (leaf coordinates, cached correspondence: L1/L2 hits from the second round
    // on): in flight during the wait below
    vd4 pv0[QPT];
    float cmar0[QPT];
    unsigned int cword0[QPT];
    {
      const int i_end0 = have_first ? min(Lv, (r_first + 1) * S) : 0;
      const int i_last0 = max(i_end0 - 1, 0);
#pragma unroll
      for (int j = 0; j < QPT; ++j) {
        const int i = max(min(phys(r_first, min(r_first * S + j * kBlock + tid, i_last0)), L - 1), 0);
        pv0[j] = ((gptr_d4)(uintptr_t)moving)[i];
        cmar0[j] = 0.f;
          cword0[j] = 0u;
        if (reuse) {  // (uniform)
          const long long ci = (long long)k_first * L + i;
          cmar0[j] = ((gptr_f1)(uintptr_t)cache_margin)[ci];
          cword0[j] = ((gptr_u1)(uintptr_t)cache_leaf)[ci];
        }
      }
    }
    MADICP_STAMP(0);
    // ---- the pose of this round: wave 0 reads the eight folded rows of round - 1, adds them, solves ---------------
    if (tid < 64 && round > 0) {
      gptr_g64 src = l2 + ((round - 1) & 1) * l2_stride;
      double v[4];
      bool ok[4];
      bool expired = false;
      const unsigned long long t_start = wall_clock64();
#pragma unroll
      for (int j = 0; j < 4; ++j) ok[j] = tid >= 60;  // 60 lanes x 4 values = 8 rows x 30
      // (Measured and dropped: polling ONE granule per row first and sweeping all 240 values only then — it takes the
      // waiting workgroups off the stragglers' memory channels, but adds a dependent round trip per hop: -9 %.)
      for (unsigned spins = 1;; ++spins) {
        bool all_ok = true;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (!ok[j]) {
            const int idx = tid + 60 * j;  // = 30 x + c
            const int x = idx / kAcc, c = idx - x * kAcc;
            ok[j] = granule_try(src + x * kRowGranules + 2 * c, tag_prev, v[j]);
          }
          all_ok &= ok[j];
        }
        if (__all(all_ok)) break;
        if ((spins & 63u) == 0u && wall_clock64() - t_start > kSpinLimitTicks) { expired = true; break; }
        __builtin_amdgcn_s_sleep(1);
      }
      expired = __any(expired);
      if (expired) {
        if (tid == 0) s_abort = 1;
      } else {
        if (tid < 60) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int idx = tid + 60 * j;
            const int x = idx / kAcc, c = idx - x * kAcc;
            s_rows[x][c] = v[j];
          }
        }
        wave_lds_order();
        MADICP_STAMP(1);
        if (tid < kAcc) {
          double a = s_rows[0][tid];
#pragma unroll
          for (int x = 1; x < kFoldGroups; ++x) a += s_rows[x][tid];
          s_total[tid] = a;
        }
        wave_lds_order();
        double Xp[12], Xn[12], H[36], b[6], moved[2];
#pragma unroll
        for (int i = 0; i < 12; ++i) Xp[i] = s_X[i];
        solve_pose(s_total, Xp, !(flags & kFlagNoUpdate), Xn, H, b, moved, tid);
        wave_lds_order();  // (every lane has read the old pose)
        if (tid == 0) {
          s_cnt[0] += static_cast<unsigned long long>(s_total[28]);
          s_cnt[1] += static_cast<unsigned long long>(s_total[29]);
#pragma unroll
          for (int i = 0; i < 12; ++i) s_X[i] = Xn[i];
          s_X[12] += moved[0] * (1.0 + 1e-12) + 1.75e-11;  // cumulative wear ("Bookkeeping without a store"): alpha, beta
          s_X[13] += moved[1] * (1.0 + 1e-12) + 1e-11 * (fabs(Xn[9]) + fabs(Xn[10]) + fabs(Xn[11]));
        }
      }
    }
    if (tid < 12 && bx == 0 && job->x_iters) {  // the pose this round linearises at (wave 0: behind its own LDS stores)
      wave_lds_order();
      job->x_iters[(long long)round * 12 + tid] = s_X[tid];
    }
    double acc[kAcc];
#pragma unroll
    for (int v = 0; v < kAcc; ++v) acc[v] = 0.0;
    unsigned int visits = 0, walked_visits = 0;
    bool walked = false;
    __syncthreads();
    MADICP_STAMP(2);
    if (s_abort) break;  // (uniform)
    double R[9], t[3];
#pragma unroll
    for (int k = 0; k < 9; ++k) R[k] = wave_uniform(s_X[k]);
#pragma unroll
    for (int k = 0; k < 3; ++k) t[k] = wave_uniform(s_X[9 + k]);
    const double wear_alpha = wave_uniform(s_X[12]), wear_beta = wave_uniform(s_X[13]);
    const bool stage_hint = stage_hint_next;

#define MADICP_TID tid
#include "icp_linearize_body.inc.h"
#undef MADICP_TID

    MADICP_STAMP(5);
    // ---- reduction: lanes -> waves -> this workgroup's row, published for the leader of its group -------------------
    acc[28] = static_cast<double>(visits);
    acc[29] = static_cast<double>(walked_visits);
    const int lane = tid & 63;
    const int wave = tid >> 6;
    wave_reduce_scatter(acc, lane, red[wave]);
    const int n_walked = __syncthreads_count(walked ? 1 : 0);
    stage_hint_next = n_walked > 0;
    if (tid < kAcc) {
      double s = red[0][tid];
#pragma unroll
      for (int w = 1; w < kWaves; ++w) s += red[w][tid];
      granule_store(l1 + (round & 1) * l1_stride + (size_t)bx * kRowGranules + 2 * tid, tag_now, s);
    }
    MADICP_STAMP(6);
    // ---- level 1: the leader folds its group's rows and publishes F[x] ---------------------------------------------
    if (leader) {  // (workgroup-uniform)
      gptr_g64 src = l1 + (round & 1) * l1_stride + (size_t)xcd * kRowGranules;
      const int n_vals = nslots * kAcc;
      bool expired = false;
      const unsigned long long t_start = wall_clock64();
      for (int idx = tid; idx < n_vals; idx += kBlock) {
        const int sl = idx / kAcc, c = idx - sl * kAcc;
        gptr_g64 g = src + (size_t)sl * kFoldGroups * kRowGranules + 2 * c;  // row x + 8 sl
        double v = 0.0;
        for (unsigned spins = 1; !granule_try(g, tag_now, v); ++spins) {
          if ((spins & 63u) == 0u && wall_clock64() - t_start > kSpinLimitTicks) { expired = true; break; }
          __builtin_amdgcn_s_sleep(1);
        }
        s_rows[sl][c] = v;
      }
      MADICP_STAMP(15);
      if (__syncthreads_or(expired ? 1 : 0)) {
        if (tid == 0) s_abort = 1;
      } else if (tid < kAcc) {
        double rv[kJoinGroups];  // (every LDS read in flight before the first addition; rows the group does not have: +0.0)
#pragma unroll
        for (int sl = 0; sl < kJoinGroups; ++sl) rv[sl] = sl < nslots ? s_rows[sl][tid] : 0.0;
        double f = 0.0;
#pragma unroll
        for (int q = 0; q < kFoldChains; ++q) {
          double sg = 0.0;
#pragma unroll
          for (int sl = q; sl < kJoinGroups; sl += kFoldChains) sg += rv[sl];
          f = (q == 0) ? sg : f + sg;
        }
        granule_store(l2 + (round & 1) * l2_stride + (size_t)xcd * kRowGranules + 2 * tid, tag_now, f);
      }
      __syncthreads();  // s_rows is reused by wave 0 at the top of the next round; s_abort published
      MADICP_STAMP(14);
      if (s_abort) break;
    }
  }
  // bookkeeping of the rounds joined in this launch (icp_final adds the last round's)
  Job* jout = jobs_out + blockIdx.y;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    jout->visits += s_cnt[0];
    jout->walked += s_cnt[1];
#pragma unroll
    for (int i = 0; i < 12; ++i) jout->Xring[(n_iters - 1) & 1][i] = s_X[i];  // the pose the last round linearised at
    jout->iter = n_iters - 1;
  }
  if (threadIdx.x == 0 && s_abort) jout->error = 1;
}

// matched-leaf count (pipeline.cpp:197-204), whole workgroup; valid once the last round's linearisation is done
__device__ __forceinline__ void count_matched(Job* job) {
  __shared__ int cnt[kBlock / 64];
  const int L = job->L;
  const uint4* m16 = reinterpret_cast<const uint4*>(job->matched);
  uint8_t* hm = job->outbox ? nullptr : job->host_matched;  // pinned host copy of the flags (posted PCIe writes, 16 bytes per lane)
  int c = 0;
  for (int i = threadIdx.x; i < (L >> 4); i += blockDim.x) {
    const uint4 v = m16[i];  // flags are 0/1 bytes: popcount counts them
    c += __popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w);
    if (hm) reinterpret_cast<uint4*>(hm)[i] = v;
  }
  for (int i = (L & ~15) + threadIdx.x; i < L; i += blockDim.x) {
    const uint8_t f = job->matched[i];
    c += f ? 1 : 0;
    if (hm) hm[i] = f;
  }
  for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
  if ((threadIdx.x & 63) == 0) cnt[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    int t = 0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += cnt[w];
    job->n_matched = t;
  }
}

// after the last round: join + solve once more -> final pose, H, b of the last round, counters, matched-leaf count
// (pipeline.cpp:195-204,223).  grid = n_scans, block = kBlock (the join order depends on it).  nblocks = workgroups per scan of icp_round.
__global__ __launch_bounds__(kBlock) void icp_final(Job* __restrict__ jobs, const double* __restrict__ partials,
                                                          const double* __restrict__ totals, int nblocks, int n_scans,
                                                          const unsigned long long* __restrict__ xch, const PeerBox pb) {
  __shared__ double s_total[kAcc];
  Job* job = jobs + blockIdx.x;
  const int n = job->n_iters;
  if (totals) {
    if (threadIdx.x < kAcc) s_total[threadIdx.x] = totals[blockIdx.x * kAcc + threadIdx.x];
    __syncthreads();
  } else if (xch) {
    // behind icp_persist: the eight folded rows F[x] of the last round, left by the group leaders (level 2 of the
    // exchange; the kernel boundary made them visible, the tags are checked all the same)
    __shared__ double s_f[kFoldGroups][kAcc];
    const unsigned tag = (job->epoch << 8) + (unsigned)n;
    if (threadIdx.x < kFoldGroups * kAcc) {
      const int x = threadIdx.x / kAcc, c = threadIdx.x - x * kAcc;
      gptr_g64 src = (gptr_g64)(uintptr_t)xch + xch_level1(n_scans, nblocks) +
                     ((size_t)((n - 1) & 1) * n_scans + blockIdx.x) * kFoldGroups * kRowGranules;
      double v;
      if (!granule_try(src + x * kRowGranules + 2 * c, tag, v)) job->error = 2;  // (a round never published: icp_persist gave up)
      s_f[x][c] = v;
    }
    __syncthreads();
    if (threadIdx.x < kAcc) {
      double a = s_f[0][threadIdx.x];
#pragma unroll
      for (int x = 1; x < kFoldGroups; ++x) a += s_f[x][threadIdx.x];
      s_total[threadIdx.x] = a;
    }
    __syncthreads();
  } else {
    // sharded over peer-mapped mailboxes (option "shard_p2p"): this rank's matched flags go out first — plain stores, in flight
    // while the sums are joined — then the last round's join over the ranks, then the peers' flags are OR-ed in
    const bool box_flags = pb.n_ranks > 1 && pb.flags_in_box;
    const unsigned p2p_epoch = job->p2p_epoch;
    if (box_flags) p2p_flags_publish(pb, p2p_epoch, pb.scan0 + (int)blockIdx.x, job->matched, job->L);
    const int prows = join_rows(nblocks);
    const long long pstride = (long long)n_scans * prows * kAcc;
    join_partials(partials + ((n - 1) & 1) * pstride + (long long)blockIdx.x * prows * kAcc, prows, s_total);
    if (pb.n_ranks > 0 && threadIdx.x < 64 &&
        !p2p_exchange(pb, p2p_epoch, pb.scan0 + (int)blockIdx.x, n - 1, true, s_total, &job->error) && threadIdx.x == 0)
      p2p_mark_lost(&job->error);
    if (box_flags) {
      if (!p2p_flags_join(pb, p2p_epoch, pb.scan0 + (int)blockIdx.x, job->matched, job->L, &job->error)) p2p_mark_lost(&job->error);
      __syncthreads();  // (count_matched below reads what other threads have just OR-ed in)
    }
  }
  count_matched(job);
  if (threadIdx.x < 64) {  // wave 0, every lane the same values (solve_pose is wave-uniform)
    double Xp[12], Xn[12], H[36], b[6];
#pragma unroll
    for (int i = 0; i < 12; ++i) Xp[i] = job->Xring[(n - 1) & 1][i];
    double moved[2];
    solve_pose(s_total, Xp, !(job->flags & kFlagNoUpdate), Xn, H, b, moved, threadIdx.x & 63);
    const unsigned long long visits = job->visits + static_cast<unsigned long long>(s_total[28]);
    const unsigned long long walked_n = job->walked + static_cast<unsigned long long>(s_total[29]);
    if (threadIdx.x == 0) {
#pragma unroll
      for (int i = 0; i < 36; ++i) job->H[i] = H[i];
#pragma unroll
      for (int i = 0; i < 6; ++i) job->b[i] = b[i];
#pragma unroll
      for (int i = 0; i < 12; ++i) job->X[i] = Xn[i];
      job->n_pairs = s_total[27];
      job->visits = visits;
      job->walked = walked_n;
      job->iter = n;
    }
    // the same results for the caller: lanes 0..35 carry H, then 12 lanes X, 6 lanes b (every lane holds all of them)
    const int l = threadIdx.x;
    double h = H[0], x = Xn[0], bb = b[0];
#pragma unroll
    for (int i = 1; i < 36; ++i) h = (l == i) ? H[i] : h;
#pragma unroll
    for (int i = 1; i < 12; ++i) x = (l - 36 == i) ? Xn[i] : x;
#pragma unroll
    for (int i = 1; i < 6; ++i) bb = (l - 48 == i) ? b[i] : bb;
    if (Outbox* ob = job->outbox) {
      // ... as tagged granules in the device-resident outbox (icp_publish carries them to the host from a side stream)
      double v = l < 36 ? h : (l < 48 ? x : bb);
      if (l == 54) v = s_total[27];
      if (l == 55) v = __longlong_as_double((long long)visits);
      if (l == 56) v = __longlong_as_double((long long)walked_n);
      if (l == 58) v = __longlong_as_double((long long)job->error);
      if (l < 59 && l != 57) granule_store((gptr_g64)(uintptr_t)ob->g + 2 * l, (unsigned)job->seq, v);
    } else if (HostResult* ho = job->host_out) {
      // ... straight to the caller's pinned block
      if (l < 36) ho->H[l] = h;
      if (l >= 36 && l < 48) ho->X[l - 36] = x;
      if (l >= 48 && l < 54) ho->b[l - 48] = bb;
      if (l == 0) {
        ho->n_pairs = s_total[27];
        ho->visits = visits;
        ho->walked = walked_n;
        ho->iter = n;
        ho->error = job->error;
      }
    }
  }
  if (job->outbox) {
    __syncthreads();  // count_matched's n_matched (thread 0) is final
    if (threadIdx.x == 0)
      granule_store((gptr_g64)(uintptr_t)job->outbox->g + 2 * 57, (unsigned)job->seq,
                    __longlong_as_double((long long)(((unsigned long long)(unsigned)n << 32) | (unsigned)job->n_matched)));
    return;
  }
  // The caller's completion signal is HostResult::seq in its pinned block (no event behind the registration: a
  // barrier packet on the queue costs more than this kernel).  Every thread orders its own host stores (flags, H, X, b)
  // before the barrier; thread 0 then writes the count and releases the sequence number at system scope.
  if (job->host_out) __threadfence_system();
  __syncthreads();  // count_matched's n_matched (thread 0) is final
  if (threadIdx.x == 0 && job->host_out) {
    job->host_out->n_matched = job->n_matched;
    __hip_atomic_store(&job->host_out->seq, job->seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// Streamed registrations: carries one registration's results from its device-resident outbox to the caller's pinned block
// (see Outbox).  ONE workgroup, enqueued on the side stream when the registration is submitted: it waits — one wavefront
// polling with s_sleep, bounded — until icp_final's tags are there, then copies H / X / b / counters and the matched flags
// (written by the round kernels, i.e. before icp_final started: in memory, but possibly shadowed by stale lines of this
// XCD's L2 — hence the acquire) and releases the sequence number at system scope.  On a time-out the caller finds
// HostResult::error = 3 behind the sequence number.
__global__ __launch_bounds__(256) void icp_publish(const Outbox* __restrict__ ob, const uint8_t* __restrict__ matched, int L, int seq,
                                                   HostResult* __restrict__ ho, uint8_t* __restrict__ hm,
                                                   unsigned long long spin_ticks) {
  __shared__ int s_bad;
  __shared__ double s_v[kOutboxGranules];
  if (threadIdx.x == 0) s_bad = 0;
  __syncthreads();
  if (threadIdx.x < 59) {
    gptr_g64 g = (gptr_g64)(uintptr_t)ob->g + 2 * threadIdx.x;
    double v = 0.0;
    const unsigned long long t0 = wall_clock64();
    bool ok = granule_try(g, (unsigned)seq, v);
    while (!ok) {
      __builtin_amdgcn_s_sleep(32);
      ok = granule_try(g, (unsigned)seq, v);
      if (!ok && wall_clock64() - t0 > spin_ticks) break;  // the registration never finished (>= 10 s; the caller's time-outs if longer)
    }
    if (!ok) s_bad = 1;
    s_v[threadIdx.x] = v;
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  const bool bad = s_bad != 0;
  if (!bad) {
    const uint4* m16 = reinterpret_cast<const uint4*>(matched);
    for (int i = threadIdx.x; i < (L >> 4); i += blockDim.x) reinterpret_cast<uint4*>(hm)[i] = m16[i];
    for (int i = (L & ~15) + threadIdx.x; i < L; i += blockDim.x) hm[i] = matched[i];
    const int l = threadIdx.x;
    if (l < 36) ho->H[l] = s_v[l];
    if (l >= 36 && l < 48) ho->X[l - 36] = s_v[l];
    if (l >= 48 && l < 54) ho->b[l - 48] = s_v[l];
  }
  if (threadIdx.x == 0) {
    const unsigned long long packed = (unsigned long long)__double_as_longlong(s_v[57]);
    ho->n_pairs = s_v[54];
    ho->visits = (unsigned long long)__double_as_longlong(s_v[55]);
    ho->walked = (unsigned long long)__double_as_longlong(s_v[56]);
    ho->n_matched = (int32_t)(unsigned)(packed & 0xffffffffull);
    ho->iter = (int32_t)(unsigned)(packed >> 32);
    ho->error = bad ? 3 : (int32_t)__double_as_longlong(s_v[58]);
  }
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(&ho->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Results of a batch to the host (madicp_icp_publish_enqueue): one wavefront per scan, behind the batch's icp_final on the
// compute stream, copies what madicp_icp_fetch reads of the Job into the caller-visible pinned block and releases the
// sequence number at system scope — one launch instead of a copy command per scan and a stream synchronisation.
__global__ __launch_bounds__(64) void batch_publish(const Job* __restrict__ jobs, HostResult* __restrict__ out, int seq) {
  const Job* job = jobs + blockIdx.x;
  HostResult* ho = out + blockIdx.x;
  const int l = threadIdx.x;
  if (l < 36) ho->H[l] = job->H[l];
  if (l >= 36 && l < 48) ho->X[l - 36] = job->X[l - 36];
  if (l >= 48 && l < 54) ho->b[l - 48] = job->b[l - 48];
  if (l == 54) {
    ho->n_pairs = job->n_pairs;
    ho->visits = job->visits;
    ho->walked = job->walked;
    ho->n_matched = job->n_matched;
    ho->iter = job->iter;
    ho->error = job->error;
  }
  __threadfence_system();
  __syncthreads();
  if (l == 0) __hip_atomic_store(&ho->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// development (option "debug_collective_us"): stands where a collective would — one wavefront that waits for `ticks` of the
// 100 MHz wall clock — so that the launch structure around an all-reduce can be timed on a box with one GPU
__global__ void debug_delay(unsigned long long ticks) {
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}

// multi-GPU: this rank's partials of the round just linearised -> totals[scan][kAcc], then ncclAllReduce(sum)
__global__ __launch_bounds__(kBlock) void icp_reduce(const double* __restrict__ partials, int nblocks, int n_scans,
                                                           int round, double* __restrict__ totals) {
  __shared__ double s_total[kAcc];
  const int prows = join_rows(nblocks);
  const long long pstride = (long long)n_scans * prows * kAcc;
  join_partials(partials + (round & 1) * pstride + (long long)blockIdx.x * prows * kAcc, prows, s_total);
  if (threadIdx.x < kAcc) totals[blockIdx.x * kAcc + threadIdx.x] = s_total[threadIdx.x];
}

}  // namespace madicp
