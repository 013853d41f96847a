// CPU check of mad_icp_amd/csrc/common/split_order.h — the per-point closed form of the permutation that the reference's
// `split` leaves (mad_icp/src/tools/utils.h:37-52) — against that loop itself, restated on indices: every left/right
// pattern of up to 16 points, random patterns of up to 32 points through the bit-select form, and random patterns of up to
// 40 000 points through the rank tables, whole-node and chunked (as the three regimes of the device builder use them).
// Test infrastructure; compiled and run by tests/test_split_order.py.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "split_order.h"

using namespace madicp_host;

// utils.h:37-52 on indices: out[d] = the position the point at d started from; returns the split position
static int reference_split(const std::vector<uint8_t>& left, std::vector<int>& out) {
  const int n = (int)left.size();
  out.resize(n);
  for (int i = 0; i < n; ++i) out[i] = i;
  int lower = 0, base = n;  // upper = reverse_iterator(end): *upper is the element at base - 1
  while (lower != base) {
    if (left[out[lower]]) {
      ++lower;
    } else {
      std::swap(out[lower], out[base - 1]);
      --base;
    }
  }
  return base;
}

static bool check_tables(const std::vector<uint8_t>& left, int chunk_points, int max_gran) {
  const int n = (int)left.size();
  std::vector<int> want;
  const int mid = reference_split(left, want);
  int n_left = 0;
  for (uint8_t f : left) n_left += f;
  if (mid != n_left) return false;
  std::vector<int> got(n, -1);
  if (chunk_points <= 0) {  // whole-node tables (wave regime): lefts from the front of the table, rights from its back
    std::vector<int> tab(n);
    int lb = 0;
    for (int p = 0; p < n; ++p) {
      if (left[p]) tab[lb] = p; else tab[n - 1 - (p - lb)] = p;
      lb += left[p];
    }
    lb = 0;
    for (int p = 0; p < n; ++p) {
      const SplitPlan s = split_plan(left[p], p, lb, n_left, n);
      const int d = s.kind == 0 ? s.idx : (s.kind == 1 ? tab[n - 1 - s.idx] : tab[s.idx] - 1);
      if (d < 0 || d >= n || got[d] != -1) return false;
      got[d] = p;
      lb += left[p];
    }
  } else {  // chunked tables (chip regime)
    const int n_chunks = (n + chunk_points - 1) / chunk_points;
    std::vector<int> tab(n), cnt(n_chunks, 0);
    for (int c = 0; c < n_chunks; ++c) {
      const int cb = c * chunk_points, ce = std::min(n, cb + chunk_points);
      int lb = 0;
      for (int p = cb; p < ce; ++p) {
        if (left[p]) tab[cb + lb] = p; else tab[ce - 1 - ((p - cb) - lb)] = p;
        lb += left[p];
      }
      cnt[c] = lb;
    }
    // the coarse table of tb_chip_scatter (tree_build.hip.h): the ROUNDED-UP granule count must fit the table — the rule
    // "(n_chunks >> shift) <= max" lets 2 * max + 1 chunks through as max + 1 granules, one entry past the LDS array
    int shift = 0;
    while (((n_chunks + (1 << shift) - 1) >> shift) > max_gran) ++shift;
    const int n_gran = (n_chunks + (1 << shift) - 1) >> shift;
    if (n_gran > max_gran) return false;
    std::vector<int> pref(max_gran + 1);  // (the kernel's array: s_pref[kPrefMax + 1])
    {
      int run = 0;
      for (int c = 0; c < n_chunks; ++c) {
        if ((c & ((1 << shift) - 1)) == 0) pref[c >> shift] = run;
        run += cnt[c];
      }
      pref[n_gran] = run;
    }
    auto lefts_of = [&](int c) { return cnt[c]; };
    int lb = 0;
    ChunkCache rcache, lcache;  // (as the chip regime's scatter uses them: one pair per thread, eight consecutive points each)
    for (int p = 0; p < n; ++p) {
      if (p % 8 == 0) rcache = lcache = ChunkCache{};
      const SplitPlan s = split_plan(left[p], p, lb, n_left, n);
      int d = s.idx;
      if (s.kind == 1) {
        int c, local, c2, local2;
        find_right_chunk(pref.data(), n_gran, shift, n_chunks, chunk_points, n, s.idx, lefts_of, c, local);
        find_right_chunk_cached(rcache, pref.data(), n_gran, shift, n_chunks, chunk_points, n, s.idx, lefts_of, c2, local2);
        if (c2 != c || local2 != local) return false;
        const int ce = std::min(n, (c + 1) * chunk_points);
        d = tab[ce - 1 - local];
      } else if (s.kind == 2) {
        int c, local, c2, local2;
        find_left_chunk(pref.data(), n_gran, shift, n_chunks, s.idx, lefts_of, c, local);
        find_left_chunk_cached(lcache, pref.data(), n_gran, shift, n_chunks, s.idx, lefts_of, c2, local2);
        if (c2 != c || local2 != local) return false;
        d = tab[c * chunk_points + local] - 1;
      }
      if (d < 0 || d >= n || got[d] != -1) return false;
      got[d] = p;
      lb += left[p];
    }
  }
  return got == want;
}

int main() {
  long checked = 0;
  // every pattern of up to 16 points: bit-select form and both table forms
  for (int n = 0; n <= 16; ++n)
    for (uint32_t m = 0; m < (1u << n); ++m) {
      std::vector<uint8_t> left(n);
      for (int p = 0; p < n; ++p) left[p] = (m >> p) & 1u;
      std::vector<int> want, got(n, -1);
      reference_split(left, want);
      for (int p = 0; p < n; ++p) {
        const int d = split_dst_small(m, n, p);
        if (d < 0 || d >= n || got[d] != -1) { std::printf("small: bad destination n=%d m=%x p=%d\n", n, m, p); return 1; }
        got[d] = p;
      }
      if (got != want) { std::printf("small: n=%d m=%x differs\n", n, m); return 1; }
      if (!check_tables(left, 0, 0) || !check_tables(left, 4, 2) || !check_tables(left, 3, 64)) {
        std::printf("tables: n=%d m=%x differs\n", n, m);
        return 1;
      }
      ++checked;
    }
  std::mt19937_64 rng(7);
  for (int it = 0; it < 200000; ++it) {  // up to 32 points, all densities
    const int n = 1 + (int)(rng() % 32);
    const uint32_t dens = (uint32_t)(rng() % 5);
    uint32_t m = (uint32_t)rng();
    if (dens == 0) m &= (uint32_t)rng();
    if (dens == 1) m |= (uint32_t)rng();
    if (n < 32) m &= (1u << n) - 1u;
    std::vector<uint8_t> left(n);
    for (int p = 0; p < n; ++p) left[p] = (m >> p) & 1u;
    std::vector<int> want, got(n, -1);
    reference_split(left, want);
    for (int p = 0; p < n; ++p) got[split_dst_small(m, n, p)] = p;
    if (got != want) { std::printf("small random: n=%d m=%x differs\n", n, m); return 1; }
    ++checked;
  }
  for (int it = 0; it < 400; ++it) {  // big nodes through the tables
    const int n = 1 + (int)(rng() % 40000);
    const double pr = (rng() % 1000) / 999.0;
    std::vector<uint8_t> left(n);
    for (int p = 0; p < n; ++p) left[p] = ((rng() % 100000) / 100000.0) < pr ? 1 : 0;
    if (it % 7 == 0)  // long runs
      for (int p = 0; p < n; ++p) left[p] = ((p / (1 + it)) & 1) ? 1 : 0;
    const int chunk = it % 3 == 0 ? 2048 : 1 + (int)(rng() % 300);
    const int gran = it % 2 ? 4096 : 1 + (int)(rng() % 9);
    if (!check_tables(left, 0, 0) || !check_tables(left, chunk, gran)) {
      std::printf("tables random: n=%d chunk=%d gran=%d differs\n", n, chunk, gran);
      return 1;
    }
    ++checked;
  }
  for (int gran = 1; gran <= 6; ++gran)  // chunk counts around every multiple of the table size: 2 g + 1 chunks at shift 1 are g + 1 granules
    for (int chunks = 1; chunks <= 8 * gran + 3; ++chunks)
      for (int tail = 1; tail <= 5; tail += 2) {
        const int n = (chunks - 1) * 5 + tail;
        std::vector<uint8_t> left(n);
        for (int p = 0; p < n; ++p) left[p] = (rng() % 3) ? 1 : 0;
        if (!check_tables(left, 5, gran)) {
          std::printf("coarse table: chunks=%d gran=%d n=%d differs\n", chunks, gran, n);
          return 1;
        }
        ++checked;
      }
  std::printf("split order ok: %ld patterns\n", checked);
  return 0;
}
