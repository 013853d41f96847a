// The permutation `split` (reference mad_icp/src/tools/utils.h:37-52) leaves, as a function of the side flags alone —
// ONE source for the device tree builder's three regimes (csrc/hip/tree_build.hip.h) and for the CPU check that pins it
// against the reference's loop (tests/cpp/split_order_check.cpp, tests/test_split_order.py).
//
// The reference's loop walks a lower cursor up the range; a point that has to go right is swapped with the point under
// the upper cursor, which is then examined in its place.  With n points, nL of them going left (so the split position is
// nL) and `left(p)` the side of the point that STARTS at position p, the point ends at
//
//   left,  p <  nL              p                       (a front point that stays)
//   left,  p >= nL              R[r - 1]                 r = number of lefts at positions >= p (rank from the top)
//   right, p >  nL              p - 1                    (examined after a swap, handed one position down)
//   right, p <= nL              n - 1          if r = 1  r = 1 + number of rights before p
//                               L[nL - r + 1] - 1  else
//
// where L[i] is the position of the i-th left and R[j] the position of the j-th right, both counted from the front
// (0-based).  In words: the r-th front point that goes right (a "hole") receives the r-th left from the top, the hole's own
// point goes to the top of the run the upper cursor walks next (the end of the range, then just below each left it took),
// and every back point that goes right moves down by one; a right AT position nL is the hole nobody fills.  The same
// closed form drives the host builder's serial sweep (csrc/host/tree_builder.cpp, partition_from_flags); here it is stated
// per point, so that every point can be placed independently once the two rank tables exist.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define MADICP_SO_HD __host__ __device__
#else
#define MADICP_SO_HD
#endif

namespace madicp_host {

MADICP_SO_HD inline int so_popc(uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __popc(v);
#else
  return __builtin_popcount(v);
#endif
}
MADICP_SO_HD inline uint32_t so_brev(uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __brev(v);
#else
  v = ((v >> 1) & 0x55555555u) | ((v & 0x55555555u) << 1);
  v = ((v >> 2) & 0x33333333u) | ((v & 0x33333333u) << 2);
  v = ((v >> 4) & 0x0f0f0f0fu) | ((v & 0x0f0f0f0fu) << 4);
  v = ((v >> 8) & 0x00ff00ffu) | ((v & 0x00ff00ffu) << 8);
  return (v >> 16) | (v << 16);
#endif
}

// position of the r-th set bit of m counted from bit 0 (r = 1 .. popcount(m)): five halvings, no loop over the bits
MADICP_SO_HD inline int select32(uint32_t m, int r) {
  int pos = 0, c;
  c = so_popc(m & 0xffffu); if (r > c) { r -= c; pos += 16; m >>= 16; }
  c = so_popc(m & 0xffu);   if (r > c) { r -= c; pos += 8;  m >>= 8; }
  c = so_popc(m & 0xfu);    if (r > c) { r -= c; pos += 4;  m >>= 4; }
  c = so_popc(m & 0x3u);    if (r > c) { r -= c; pos += 2;  m >>= 2; }
  c = (int)(m & 1u);        if (r > c) { pos += 1; }
  return pos;
}
// ... counted from bit 31 downwards
MADICP_SO_HD inline int select32_top(uint32_t m, int r) { return 31 - select32(so_brev(m), r); }

// What a point needs to find its place: either the place itself (kind 0) or ONE entry of a rank table.
struct SplitPlan {
  int kind;  // 0: dst = idx;  1: dst = R[idx] (position of the idx-th right);  2: dst = L[idx] - 1 (idx-th left)
  int idx;
};
// p: the point's position in the node (0-based), lefts_before: lefts at positions < p, n_left: lefts of the node, n: points
MADICP_SO_HD inline SplitPlan split_plan(bool left, int p, int lefts_before, int n_left, int n) {
  SplitPlan s;
  if (left) {
    if (p < n_left) { s.kind = 0; s.idx = p; return s; }
    s.kind = 1;
    s.idx = n_left - lefts_before - 1;  // r = n_left - lefts_before, R[r - 1]
    return s;
  }
  if (p > n_left) { s.kind = 0; s.idx = p - 1; return s; }
  const int r = p - lefts_before + 1;
  if (r == 1) { s.kind = 0; s.idx = n - 1; return s; }
  s.kind = 2;
  s.idx = n_left - r + 1;
  return s;
}

// Nodes of at most 32 points: the flags are one word (bit p = the point at p goes left) and the tables are bit selects.
MADICP_SO_HD inline int split_dst_small(uint32_t mask, int n, int p) {
  const int n_left = so_popc(mask);
  const bool left = (mask >> p) & 1u;
  const int lefts_before = so_popc(mask & ((1u << p) - 1u));
  const SplitPlan s = split_plan(left, p, lefts_before, n_left, n);
  if (s.kind == 0) return s.idx;
  if (s.kind == 1) {
    const uint32_t rights = ~mask & (n >= 32 ? 0xffffffffu : ((1u << n) - 1u));
    return select32(rights, s.idx + 1);
  }
  // the idx-th left from the front is the (n_left - idx)-th from the top
  return select32_top(mask, n_left - s.idx) - 1;
}

// Nodes cut into chunks of `chunk` points (the chip regime): the rank tables are written per chunk — chunk c lists the
// positions of ITS lefts from the front of its slice of the table and of its rights from the back — and a global rank is
// turned into (chunk, rank inside the chunk) by a search over the exclusive prefix of the chunks' left counts.  `pref` holds
// that prefix for every (1 << shift)-th chunk, pref[n_gran] = all lefts (a node with more chunks than the table has room
// for is searched coarsely, then walked chunk by chunk with `lefts_of(c)`).
template <class CountFn>
MADICP_SO_HD inline void find_left_chunk(const int* pref, int n_gran, int shift, int n_chunks, int i, CountFn lefts_of, int& chunk,
                                         int& local) {
  int lo = 0, hi = n_gran;  // largest g with pref[g] <= i
  while (hi - lo > 1) {
    const int m = (lo + hi) >> 1;
    if (pref[m] <= i) lo = m; else hi = m;
  }
  int c = lo << shift, before = pref[lo];
  if (shift > 0) {
    for (;;) {
      if (c + 1 >= n_chunks) break;
      const int k = lefts_of(c);
      if (before + k > i) break;
      before += k;
      ++c;
    }
  }
  chunk = c;
  local = i - before;
}
template <class CountFn>
MADICP_SO_HD inline void find_right_chunk(const int* pref, int n_gran, int shift, int n_chunks, int chunk_points, int n, int j,
                                          CountFn lefts_of, int& chunk, int& local) {
  int lo = 0, hi = n_gran;  // largest g with (rights before granule g) <= j
  while (hi - lo > 1) {
    const int m = (lo + hi) >> 1;
    if ((m << shift) * chunk_points - pref[m] <= j) lo = m; else hi = m;
  }
  int c = lo << shift, before = c * chunk_points - pref[lo];
  if (shift > 0) {
    for (;;) {
      if (c + 1 >= n_chunks) break;
      const int size = (c + 1) * chunk_points <= n ? chunk_points : n - c * chunk_points;
      const int k = size - lefts_of(c);
      if (before + k > j) break;
      before += k;
      ++c;
    }
  }
  chunk = c;
  local = j - before;
}

// The same two searches for a caller that asks for CONSECUTIVE ranks (a thread of the chip regime's scatter owns eight
// consecutive points, and their ranks run up or down by one): the chunk found last is remembered with its first rank and its
// count, and a rank inside those bounds needs no search.  (With a coarse table, shift > 0, nothing is remembered.)
struct ChunkCache {
  int chunk = -1, first = 0, count = 0;
};
template <class CountFn>
MADICP_SO_HD inline void find_left_chunk_cached(ChunkCache& cc, const int* pref, int n_gran, int shift, int n_chunks, int i,
                                                CountFn lefts_of, int& chunk, int& local) {
  if (cc.chunk >= 0 && i >= cc.first && i < cc.first + cc.count) {
    chunk = cc.chunk;
    local = i - cc.first;
    return;
  }
  find_left_chunk(pref, n_gran, shift, n_chunks, i, lefts_of, chunk, local);
  if (shift == 0) {
    cc.chunk = chunk;
    cc.first = i - local;
    cc.count = pref[chunk + 1] - pref[chunk];
  }
}
template <class CountFn>
MADICP_SO_HD inline void find_right_chunk_cached(ChunkCache& cc, const int* pref, int n_gran, int shift, int n_chunks, int chunk_points,
                                                 int n, int j, CountFn lefts_of, int& chunk, int& local) {
  if (cc.chunk >= 0 && j >= cc.first && j < cc.first + cc.count) {
    chunk = cc.chunk;
    local = j - cc.first;
    return;
  }
  find_right_chunk(pref, n_gran, shift, n_chunks, chunk_points, n, j, lefts_of, chunk, local);
  if (shift == 0) {
    const int size = (chunk + 1) * chunk_points <= n ? chunk_points : n - chunk * chunk_points;
    cc.chunk = chunk;
    cc.first = j - local;
    cc.count = size - (pref[chunk + 1] - pref[chunk]);
  }
}

}  // namespace madicp_host
