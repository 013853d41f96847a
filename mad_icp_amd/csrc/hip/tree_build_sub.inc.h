// Part of tree_build.hip.h (namespace madicp::tb): the two regimes below the chip levels.
//
//   block    (n > kSubMax past the chip levels): ONE workgroup of eight wavefronts per node and level, points streamed from
//            one global buffer to the other — the chip regime's three kernels in one, because a single workgroup can
//            synchronise with a barrier where the chunks of a chip node need a kernel boundary.  Every wavefront owns a
//            contiguous slice of the node (an eighth), the slices play the part of the chip regime's chunks: slice-local rank
//            tables, a prefix over eight counts, the same search (common/split_order.h).
//   sub-tree (n <= kSubMax = 4096, at whatever level such a node appears): ONE workgroup takes the node and finishes the
//            WHOLE sub-tree below it without leaving the kernel.  The points live in LDS (96 KB) and are permuted in place —
//            a lane keeps the (at most eight) points it owns in registers from the sweep that reads them to the scatter that
//            writes them, so no second buffer is needed — the per-level node lists live in LDS, node ids come from a range
//            reserved with one global atomic per sub-tree, and a level costs two or three workgroup barriers instead of a
//            kernel boundary (a level of the former per-level kernel was 16-70 us: a chain of first touches of memory
//            another kernel had just written).  Inside, a node is handled by a TEAM of 2 / 4 / 8 wavefronts (more than 512
//            points), by one wavefront (33 .. 512) or by four lanes (up to 32) — at most eight points per lane either way.
//
// Both compute a node's sums from the node's own points (a chip-regime parent still hands its per-chunk partials down):
// the children's sums no longer travel through the parent's scatter.
#pragma once

constexpr int kSubThreads = 512;   // eight wavefronts (two per SIMD: the eigen-solve needs its 255 registers)
constexpr int kSubWaves = kSubThreads / 64;
constexpr int kSubQuadMax = kSubMax;          // quad-regime nodes of one level (singletons: one per point)
constexpr int kSubWaveMax = kSubMax / 33 + 4; // wave-regime nodes of one level (33 .. 512 points each)
constexpr int kSubTeamMax = 8;                // nodes of more than 512 points on one level (at most 7)
// dynamic LDS of tb_subtree
constexpr size_t kSubOffTab = sizeof(double) * 3 * kSubMax;                      // uint16 tab[kSubMax]
constexpr size_t kSubOffQuad = kSubOffTab + sizeof(unsigned short) * kSubMax;    // uint32 quad[2][kSubQuadMax]
constexpr size_t kSubOffWave = kSubOffQuad + sizeof(unsigned int) * 2 * kSubQuadMax;  // int2 wave[2][kSubWaveMax]
constexpr size_t kSubOffTeam = kSubOffWave + sizeof(int2) * 2 * kSubWaveMax;     // int2 team[2][kSubTeamMax]
constexpr size_t kSubLdsBytes = kSubOffTeam + sizeof(int2) * 2 * kSubTeamMax;
static_assert(kSubLdsBytes <= 150 * 1024, "tb_subtree: dynamic LDS + a few KB of static LDS must fit the CU's 160 KB");

// list entries.  Quad regime: one word — node id relative to the sub-tree's reserved range (13 bits, 8191 = the sub-tree's
// root, whose id was handed out by its parent), first point (12 bits), points - 1 (5 bits).  Wave / team: {id, b | e << 16}.
constexpr unsigned kSubRootRel = 8191u;
__device__ __forceinline__ unsigned sub_quad_pack(unsigned rel, int b, int n) { return rel | ((unsigned)b << 13) | ((unsigned)(n - 1) << 25); }
__device__ __forceinline__ int2 sub_ent_pack(unsigned rel, int b, int e) { return make_int2((int)rel, b | (e << 16)); }

__device__ __forceinline__ int team_width(int n) { return n > 2048 ? 8 : (n > 1024 ? 4 : 2); }

struct SubCtx {  // what every task of a sub-tree needs (registers: all uniform)
  double* pts;             // LDS, 3 * n0 doubles
  unsigned short* tab;     // LDS
  int b0;                  // the root's first point (absolute)
  int root_id, base;       // the root's node id; first id of the reserved range
  int level;               // level of the nodes being processed
};
__device__ __forceinline__ int sub_node_id(const SubCtx& c, unsigned rel) { return rel == kSubRootRel ? c.root_id : c.base + (int)rel; }

// nearest member to `mean` among a lane's points: (distance, index) with "first member wins" on ties (mad_tree.cpp:76-86)
__device__ __forceinline__ void nearest_update(double& best, int& besti, const double* mean, double x, double y, double z, int i) {
  const double d[3] = {x - mean[0], y - mean[1], z - mean[2]};
  const double dist = madicp_host::norm3(d);
  if (dist < best || (dist == best && i < besti)) { best = dist; besti = i; }
}
__device__ __forceinline__ void nearest_wave(double& best, int& besti) {
#pragma unroll
  for (int m = 32; m > 0; m >>= 1) {
    const double ob = __shfl_xor(best, m, 64);
    const int oi = __shfl_xor(besti, m, 64);
    if (ob < best || (ob == best && oi < besti)) { best = ob; besti = oi; }
  }
}

// the record of a finished leaf
__device__ __forceinline__ void write_leaf(const Params& P, BNode& nd, const Inherit& inh, int n, const double* V, const double* rep,
                                           double bbox0, int b_abs) {
  double nrm[3];
  leaf_normal(inh, n, V, nrm);
#pragma unroll
  for (int i = 0; i < 3; ++i) { nd.mean[i] = rep[i]; nd.dir[i] = nrm[i]; }
  nd.bbox0 = bbox0;
  nd.flags = (inh.flags & ~kLeafPending) | kLeaf | kDone;
  P.leaf_start[b_abs] = 1u;
}
// the record of a node that split, and its two children's records (ids c, c + 1)
__device__ __forceinline__ void write_split(const Params& P, BNode& nd, int id, const Inherit& inh, const double* mean, const double* V,
                                            double ext0, int b_abs, int mid_abs, int e_abs, int c) {
  const double col0[3] = {V[0], V[3], V[6]};
#pragma unroll
  for (int i = 0; i < 3; ++i) { nd.mean[i] = mean[i]; nd.dir[i] = V[3 * i + 2]; nd.col0[i] = V[3 * i]; }
  nd.bbox0 = ext0;
  nd.mid = mid_abs;
  nd.flags = inh.flags | kDone;
  make_child(P.nodes[c], inh, id, col0, ext0, e_abs - b_abs, P.b_min, b_abs, mid_abs, true);
  make_child(P.nodes[c + 1], inh, id, col0, ext0, e_abs - b_abs, P.b_min, mid_abs, e_abs, false);
}

// ---- block regime --------------------------------------------------------------------------------------------------------
// One workgroup per queue entry {id, b, e, level of the node}: a node of more than kSubMax points, or a chip-regime node
// that turned out to be a leaf (kLeafPending: only its representative is missing).
__global__ __launch_bounds__(kSubThreads) __attribute__((amdgpu_waves_per_eu(2, 2))) void tb_block_level(const Params P, int step) {
  State* st = P.st;
  const int cnt = st->q_count[step].v;
  if (step + 1 > kMaxLevels) {
    if (cnt > 0 && blockIdx.x == 0 && threadIdx.x == 0) st->n_nodes.error = 2;
    return;
  }
  __shared__ double s_sum[kSubWaves][9];
  __shared__ double s_ext[kSubWaves][6];
  __shared__ int s_nl[kSubWaves + 1];
  __shared__ double s_best[kSubWaves];
  __shared__ int s_besti[kSubWaves];
  __shared__ int s_child;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  for (int t = blockIdx.x; t < cnt; t += gridDim.x) {  // (workgroup-uniform)
    const int4 ent = level_q(P, step)[t];
    const int id = ent.x, b = ent.y, e = ent.z, n = e - b, level = ent.w;
    BNode& nd = P.nodes[id];
    const Inherit inh = load_inherit(nd, level);
    const double* __restrict__ in = level_in(P, level);
    double* __restrict__ out = level_out(P, level);
    // the wavefront's slice of the node
    const int S = (n + kSubWaves - 1) / kSubWaves;
    const int sb = min(b + wv * S, e), se = min(sb + S, e);
    double mean[3], V[9];
    if (inh.flags & kLeafPending) {  // statistics are there: the member nearest to the centroid is what is missing
#pragma unroll
      for (int i = 0; i < 3; ++i) mean[i] = nd.mean[i];
      V[0] = nd.col0[0]; V[3] = nd.col0[1]; V[6] = nd.col0[2];
      double best = 1.7976931348623157e308;
      int besti = 0x7fffffff;
      for (int i = sb + lane; i < se; i += 64) nearest_update(best, besti, mean, in[3 * (long)i], in[3 * (long)i + 1], in[3 * (long)i + 2], i);
      nearest_wave(best, besti);
      if (lane == 0) { s_best[wv] = best; s_besti[wv] = besti; }
      __syncthreads();
      if (threadIdx.x == 0) {
        for (int k = 1; k < kSubWaves; ++k)
          if (s_best[k] < best || (s_best[k] == best && s_besti[k] < besti)) { best = s_best[k]; besti = s_besti[k]; }
        if (besti == 0x7fffffff) besti = b;  // every distance NaN: the reference keeps *begin
        const double rep[3] = {in[3 * (long)besti], in[3 * (long)besti + 1], in[3 * (long)besti + 2]};
        write_leaf(P, nd, inh, n, V, rep, nd.bbox0, b);
      }
      for (long j = 3 * (long)b + threadIdx.x; j < 3 * (long)e; j += kSubThreads) P.order[j] = in[j];
      __syncthreads();
      continue;
    }
    // ---- the node's nine sums: the chip-regime parent's per-chunk partials, or a sweep of the slices
    double s[9];
    if (inh.flags & kChunkSums) {
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
      for (int k = 0; k < 9; ++k) {  // slices in order: the same bits in every wavefront
        double a = s_sum[0][k];
#pragma unroll
        for (int w = 1; w < kSubWaves; ++w) a += s_sum[w][k];
        s[k] = a;
      }
    }
    double cov[9], w3[3];
    mean_cov_from_sums(s, n, mean, cov);
    madicp_host::eig3_sym(cov, w3, V);
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
      for (int w = 1; w < kSubWaves; ++w) {
        if (s_ext[w][a] < L) L = s_ext[w][a];
        if (H < s_ext[w][3 + a]) H = s_ext[w][3 + a];
      }
      ext[a] = H - L;
    }
    int nl = 0;
#pragma unroll
    for (int w = 0; w < kSubWaves; ++w) nl += s_nl[w];
    const bool leaf = (ext[2] < P.b_max) || nl == 0 || nl == n;
    if (leaf) {  // (rare: more than kSubMax points within b_max of each other)
      double best = 1.7976931348623157e308;
      int besti = 0x7fffffff;
      for (int i = sb + lane; i < se; i += 64) nearest_update(best, besti, mean, in[3 * (long)i], in[3 * (long)i + 1], in[3 * (long)i + 2], i);
      nearest_wave(best, besti);
      if (lane == 0) { s_best[wv] = best; s_besti[wv] = besti; }
      __syncthreads();
      if (threadIdx.x == 0) {
        for (int k = 1; k < kSubWaves; ++k)
          if (s_best[k] < best || (s_best[k] == best && s_besti[k] < besti)) { best = s_best[k]; besti = s_besti[k]; }
        if (besti == 0x7fffffff) besti = b;
        const double rep[3] = {in[3 * (long)besti], in[3 * (long)besti + 1], in[3 * (long)besti + 2]};
        write_leaf(P, nd, inh, n, V, rep, ext[0], b);
      }
      for (long j = 3 * (long)b + threadIdx.x; j < 3 * (long)e; j += kSubThreads) P.order[j] = in[j];
      __syncthreads();
      continue;
    }
    // ---- children (thread 0), while everybody scatters
    if (threadIdx.x == 0) {
      const int c = atomicAdd(&st->n_nodes.v, 2);
      if (c + 2 > P.node_cap) {
        st->n_nodes.error = 1;
      } else {
        write_split(P, nd, id, inh, mean, V, ext[0], b, b + nl, e, c);
        enqueue_single(P, c, b, b + nl, level + 1);
        enqueue_single(P, c + 1, b + nl, e, level + 1);
      }
    }
    // ---- sweep B: every point to the place the reference's split would have left it in (common/split_order.h); the
    // slices are the "chunks" of the rank search
    __shared__ int s_pref[kSubWaves + 1];
    if (threadIdx.x <= kSubWaves) {  // exclusive prefix of the slices' left counts
      int r = 0;
      for (int w = 0; w < (int)threadIdx.x; ++w) r += s_nl[w];
      s_pref[threadIdx.x] = r;
    }
    __syncthreads();
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
            madicp_host::find_right_chunk(s_pref, kSubWaves, 0, kSubWaves, S, n, pl.idx, lefts_of, c, local);
            dst[u] = tab[(long)b + min(n, (c + 1) * S) - 1 - local];
          } else if (pl.kind == 2) {
            int c, local;
            madicp_host::find_left_chunk(s_pref, kSubWaves, 0, kSubWaves, pl.idx, lefts_of, c, local);
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
    __syncthreads();  // (the shared scratch is reused by the next entry)
  }
}

// ---- sub-tree regime -------------------------------------------------------------------------------------------------------
// per-level bookkeeping in (static) LDS
struct SubShared {
  int cnt[2][3];     // entries of the level lists [parity][0 quad, 1 wave, 2 team]
  int next_rel;      // node ids handed out so far (relative to the reserved range)
  int base;          // first id of the reserved range
  int error;
  double sum[kSubWaves][9];
  double ext[kSubWaves][6];
  int nl[kSubWaves];
  double best[kSubWaves];
  int besti[kSubWaves];
  int pref[kSubWaves][kSubWaves + 1];  // a wavefront's copy of its team's prefix of slice counts
};

// append a child of `n` points at relative position b to the next level's lists (one lane)
__device__ __forceinline__ void sub_append(SubShared& sh, unsigned int* quad, int2* wavl, int2* team, int par_next, unsigned rel, int b, int n) {
  if (n <= kSmallMax) {
    quad[par_next * kSubQuadMax + atomicAdd(&sh.cnt[par_next][0], 1)] = sub_quad_pack(rel, b, n);
  } else if (n <= 512) {
    wavl[par_next * kSubWaveMax + atomicAdd(&sh.cnt[par_next][1], 1)] = sub_ent_pack(rel, b, b + n);
  } else {
    team[par_next * kSubTeamMax + atomicAdd(&sh.cnt[par_next][2], 1)] = sub_ent_pack(rel, b, b + n);
  }
}

// One wavefront (tw == 1) or a team of tw wavefronts (slice `sl` of the node each): statistics, leaf test, in-place scatter.
// Every wavefront of the workgroup executes the barriers of the team path (TEAM) whether it has a slice or not (`mine`).
template <bool TEAM>
__device__ __forceinline__ void sub_wave_node(const Params& P, const SubCtx& c, SubShared& sh, unsigned int* quad, int2* wavl, int2* team,
                                              int par_next, bool mine, unsigned rel, int rb, int re, int tw, int sl, int team_first_wave) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  const int n = re - rb;
  const int id = mine ? sub_node_id(c, rel) : c.root_id;
  BNode& nd = P.nodes[id];
  Inherit inh = load_inherit(nd, c.level);
  // the slice (the whole node for a lone wavefront): at most 512 points, eight per lane, kept in registers to the scatter
  const int S = TEAM ? (n + tw - 1) / tw : n;
  const int sb = mine ? min(rb + sl * S, re) : 0, se = mine ? min(sb + S, re) : 0;
  double x[8], y[8], z[8];
  bool ok[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int i = sb + lane + 64 * u;
    ok[u] = i < se;
    const int j = ok[u] ? i : 0;
    x[u] = c.pts[3 * j]; y[u] = c.pts[3 * j + 1]; z[u] = c.pts[3 * j + 2];
  }
  double s[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int u = 0; u < 8; ++u)
    if (ok[u]) add_point(s, x[u], y[u], z[u]);
#pragma unroll
  for (int k = 0; k < 9; ++k) s[k] = wave_sum(s[k]);
  if (TEAM) {
    if (lane == 0) {
#pragma unroll
      for (int k = 0; k < 9; ++k) sh.sum[wv][k] = s[k];
    }
    __syncthreads();
    if (mine) {
#pragma unroll
      for (int k = 0; k < 9; ++k) {  // the team's slices in order
        double a = sh.sum[team_first_wave][k];
        for (int w = 1; w < tw; ++w) a += sh.sum[team_first_wave + w][k];
        s[k] = a;
      }
    }
  }
  double mean[3], cov[9], w3[3], V[9];
  mean_cov_from_sums(s, max(n, 1), mean, cov);
  madicp_host::eig3_sym(cov, w3, V);
  // sides, extents, rank tables (slice-local for a team: positions relative to the node, ranks relative to the slice)
  double lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
  unsigned lbits = 0;
  int lbp[8];
  int lcount = 0;
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    double v[3] = {0, 0, 0};
    if (ok[u]) {
      eigen_coords(V, mean, x[u], y[u], z[u], v);
      minmax_update(lo, hi, v);
    }
    const bool left = ok[u] && v[2] < 0.0;
    const unsigned long long lm = __ballot(left);
    lbp[u] = lcount + __popcll(lm & lt);
    lbits |= (left ? 1u : 0u) << u;
    if (ok[u]) {
      const int q = lane + 64 * u;  // position in the slice
      c.tab[left ? sb + lbp[u] : se - 1 - (q - lbp[u])] = (unsigned short)(q + (sb - rb));
    }
    lcount += __popcll(lm);
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    lo[a] = wave_min_keep(lo[a]);
    hi[a] = wave_max_keep(hi[a]);
  }
  double ext[3];
  int nl = lcount, lb_slice = 0;
  if (TEAM) {
    if (lane == 0) {
#pragma unroll
      for (int a = 0; a < 3; ++a) { sh.ext[wv][a] = lo[a]; sh.ext[wv][3 + a] = hi[a]; }
      sh.nl[wv] = lcount;
    }
    __syncthreads();  // (also: every slice's table entries are written, every slice's points are in registers)
    if (mine) {
      int run = 0;
      for (int w = 0; w < tw; ++w) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          const double l = sh.ext[team_first_wave + w][a], h = sh.ext[team_first_wave + w][3 + a];
          if (w == 0 || l < lo[a]) lo[a] = l;
          if (w == 0 || hi[a] < h) hi[a] = h;
        }
        if (w == sl) lb_slice = run;
        if (lane == 0) sh.pref[wv][w] = run;
        run += sh.nl[team_first_wave + w];
      }
      if (lane == 0) sh.pref[wv][tw] = run;
      nl = run;
    }
    wave_lds_order();
  } else {
    wave_lds_order();  // (the tables are read by other lanes of this wavefront)
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) ext[a] = hi[a] - lo[a];
  const bool leaf = (ext[2] < P.b_max) || nl == 0 || nl == n;
  if (mine && leaf) {  // the member nearest to the centroid, first one on ties
    double best = 1.7976931348623157e308;
    int besti = 0x7fffffff;
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (ok[u]) nearest_update(best, besti, mean, x[u], y[u], z[u], sb + lane + 64 * u);
    nearest_wave(best, besti);
    if (TEAM && lane == 0) { sh.best[wv] = best; sh.besti[wv] = besti; }
    if (!TEAM && lane == 0) {
      if (besti == 0x7fffffff) besti = rb;
      const double rep[3] = {c.pts[3 * besti], c.pts[3 * besti + 1], c.pts[3 * besti + 2]};
      write_leaf(P, nd, inh, n, V, rep, ext[0], c.b0 + rb);
    }
  }
  int dst[8];
  if (mine && !leaf) {
    // Scatter plan from the tables; the points go to their places after the barrier (team) / at once (lone wavefront: all
    // its reads are behind it)
    const int* pref = sh.pref[wv];
    auto lefts_of = [&](int k) { return pref[k + 1] - pref[k]; };
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      dst[u] = 0;
      if (ok[u]) {
        const bool left = (lbits >> u) & 1u;
        const madicp_host::SplitPlan pl = madicp_host::split_plan(left, lane + 64 * u + (sb - rb), lb_slice + lbp[u], nl, n);
        dst[u] = pl.idx;
        if (TEAM) {
          if (pl.kind == 1) {
            int k, local;
            madicp_host::find_right_chunk(pref, tw, 0, tw, S, n, pl.idx, lefts_of, k, local);
            dst[u] = c.tab[rb + min(n, (k + 1) * S) - 1 - local];
          } else if (pl.kind == 2) {
            int k, local;
            madicp_host::find_left_chunk(pref, tw, 0, tw, pl.idx, lefts_of, k, local);
            dst[u] = (int)c.tab[rb + k * S + local] - 1;
          }
        } else {
          if (pl.kind == 1) dst[u] = c.tab[re - 1 - pl.idx];
          if (pl.kind == 2) dst[u] = (int)c.tab[rb + pl.idx] - 1;
        }
      }
    }
    if (!TEAM) wave_lds_order();
  }
  if (TEAM) __syncthreads();  // every team lane has read its table entries; the leaf candidates are in LDS
  if (mine && !leaf) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (ok[u]) {
        const int d = rb + dst[u];
        c.pts[3 * d] = x[u]; c.pts[3 * d + 1] = y[u]; c.pts[3 * d + 2] = z[u];
      }
  }
  if (mine && (!TEAM || sl == 0) && lane == 0) {
    if (leaf) {
      if (TEAM) {
        double best = sh.best[team_first_wave];
        int besti = sh.besti[team_first_wave];
        for (int w = 1; w < tw; ++w)
          if (sh.best[team_first_wave + w] < best || (sh.best[team_first_wave + w] == best && sh.besti[team_first_wave + w] < besti)) {
            best = sh.best[team_first_wave + w];
            besti = sh.besti[team_first_wave + w];
          }
        if (besti == 0x7fffffff) besti = rb;
        // (the representative is read from LDS: a leaf's points are never moved)
        const double rep[3] = {c.pts[3 * besti], c.pts[3 * besti + 1], c.pts[3 * besti + 2]};
        write_leaf(P, nd, inh, n, V, rep, ext[0], c.b0 + rb);
      }
    } else {
      const int k = atomicAdd(&sh.next_rel, 2);
      const int cid = c.base + k;
      if (cid + 2 > P.node_cap || k + 2 > (int)kSubRootRel) {
        sh.error = 1;
      } else {
        write_split(P, nd, id, inh, mean, V, ext[0], c.b0 + rb, c.b0 + rb + nl, c.b0 + re, cid);
        sub_append(sh, quad, wavl, team, par_next, (unsigned)k, rb, nl);
        sub_append(sh, quad, wavl, team, par_next, (unsigned)k + 1u, rb + nl, n - nl);
      }
    }
  }
}

// sixteen nodes of at most 32 points per wavefront, four lanes each
__device__ __forceinline__ void sub_quad_nodes(const Params& P, const SubCtx& c, SubShared& sh, unsigned int* quad, int par_next,
                                               bool have, unsigned word) {
  const int lane = threadIdx.x & 63, ql = lane & 3, qshift = lane & ~3;
  const unsigned rel = word & 8191u;
  const int rb = (int)((word >> 13) & 4095u), n = have ? (int)(word >> 25) + 1 : 0;
  const int id = have ? sub_node_id(c, rel) : c.root_id;
  BNode& nd = P.nodes[id];
  Inherit inh = load_inherit(nd, c.level);
  double x[8], y[8], z[8];
  bool ok[8];
#pragma unroll
  for (int st = 0; st < 8; ++st) {
    const int i = 4 * st + ql;
    ok[st] = i < n;
    const int j = ok[st] ? rb + i : 0;
    x[st] = c.pts[3 * j]; y[st] = c.pts[3 * j + 1]; z[st] = c.pts[3 * j + 2];
  }
  double s[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int st = 0; st < 8; ++st)
    if (ok[st]) add_point(s, x[st], y[st], z[st]);
#pragma unroll
  for (int k = 0; k < 9; ++k) s[k] = quad_sum(s[k]);
  double mean[3], cov[9], w3[3], V[9];
  mean_cov_from_sums(s, max(n, 1), mean, cov);
  madicp_host::eig3_sym(cov, w3, V);
  double lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
  unsigned int mask = 0;
#pragma unroll
  for (int st = 0; st < 8; ++st) {
    double v[3] = {0, 0, 0};
    if (ok[st]) {
      eigen_coords(V, mean, x[st], y[st], z[st], v);
      minmax_update(lo, hi, v);
    }
    const bool left = ok[st] && v[2] < 0.0;
    const unsigned int lm = (unsigned int)(__ballot(left) >> qshift) & 0xfu;
    mask |= lm << (4 * st);
  }
  double ext[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
#pragma unroll
    for (int m = 1; m <= 2; m <<= 1) {
      const double ol = __shfl_xor(lo[a], m, 64), oh = __shfl_xor(hi[a], m, 64);
      if (ol < lo[a]) lo[a] = ol;
      if (hi[a] < oh) hi[a] = oh;
    }
    ext[a] = hi[a] - lo[a];
  }
  const int nl = __popc(mask);
  const bool leaf = !have || (ext[2] < P.b_max) || nl == 0 || nl == n;
  // leaves: the member nearest to the centroid (every lane over its own points, then the quad's best)
  double best = 1.7976931348623157e308;
  int besti = 0x7fffffff;
#pragma unroll
  for (int st = 0; st < 8; ++st)
    if (have && leaf && ok[st]) nearest_update(best, besti, mean, x[st], y[st], z[st], rb + 4 * st + ql);
#pragma unroll
  for (int m = 1; m <= 2; m <<= 1) {
    const double ob = __shfl_xor(best, m, 64);
    const int oi = __shfl_xor(besti, m, 64);
    if (ob < best || (ob == best && oi < besti)) { best = ob; besti = oi; }
  }
  // internal nodes: in-place scatter (every read of the node's points is behind this wavefront)
  if (have && !leaf) {
#pragma unroll
    for (int st = 0; st < 8; ++st)
      if (ok[st]) {
        const int d = rb + madicp_host::split_dst_small(mask, n, 4 * st + ql);
        c.pts[3 * d] = x[st]; c.pts[3 * d + 1] = y[st]; c.pts[3 * d + 2] = z[st];
      }
  }
  // ids and list slots for the children: one LDS atomic each per wavefront
  const bool splits = have && !leaf && ql == 0;
  const unsigned long long sm = __ballot(splits);
  const int tot = __popcll(sm);
  int base_rel = 0, base_slot = 0;
  if (tot > 0) {
    if (lane == 0) {
      base_rel = atomicAdd(&sh.next_rel, 2 * tot);
      base_slot = atomicAdd(&sh.cnt[par_next][0], 2 * tot);
    }
    base_rel = __shfl(base_rel, 0, 64);
    base_slot = __shfl(base_slot, 0, 64);
  }
  if (have && ql == 0) {
    if (leaf) {
      if (besti == 0x7fffffff) besti = rb;
      // (the representative comes from LDS: a leaf's points are never moved)
      const double rep[3] = {c.pts[3 * besti], c.pts[3 * besti + 1], c.pts[3 * besti + 2]};
      write_leaf(P, nd, inh, n, V, rep, ext[0], c.b0 + rb);
    } else {
      const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
      const int rank = __popcll(sm & lt);
      const int k = base_rel + 2 * rank;
      const int cid = c.base + k;
      if (cid + 2 > P.node_cap || k + 2 > (int)kSubRootRel) {
        sh.error = 1;
      } else {
        write_split(P, nd, id, inh, mean, V, ext[0], c.b0 + rb, c.b0 + rb + nl, c.b0 + rb + n, cid);
        quad[par_next * kSubQuadMax + base_slot + 2 * rank] = sub_quad_pack((unsigned)k, rb, nl);            // (children of a
        quad[par_next * kSubQuadMax + base_slot + 2 * rank + 1] = sub_quad_pack((unsigned)k + 1u, rb + nl, n - nl);  //  quad node are quad nodes)
      }
    }
  }
}

__global__ __launch_bounds__(kSubThreads) __attribute__((amdgpu_waves_per_eu(2, 2))) void tb_subtree(const Params P, int first) {
  State* st = P.st;
  const int t = first + (int)blockIdx.x;
  if (t >= st->sub_count.v) return;
  extern __shared__ __attribute__((aligned(16))) char sub_lds[];
  __shared__ SubShared sh;
  double* pts = reinterpret_cast<double*>(sub_lds);
  unsigned short* tab = reinterpret_cast<unsigned short*>(sub_lds + kSubOffTab);
  unsigned int* quad = reinterpret_cast<unsigned int*>(sub_lds + kSubOffQuad);
  int2* wavl = reinterpret_cast<int2*>(sub_lds + kSubOffWave);
  int2* team = reinterpret_cast<int2*>(sub_lds + kSubOffTeam);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int4 root = P.sub[t];
  const int n0 = root.z - root.y;
  SubCtx c;
  c.pts = pts;
  c.tab = tab;
  c.b0 = root.y;
  c.root_id = root.x;
  c.level = root.w;
  if (threadIdx.x == 0) {
    // every node of the sub-tree below the root gets its id from ONE reservation (a tree over n0 points has at most 2 n0 - 2
    // nodes below its root); what is not used is marked invalid at the end
    sh.base = atomicAdd(&st->n_nodes.v, 2 * n0);
    sh.next_rel = 0;
    sh.error = (sh.base + 2 * n0 > P.node_cap) ? 1 : 0;
#pragma unroll
    for (int p = 0; p < 2; ++p)
      for (int r = 0; r < 3; ++r) sh.cnt[p][r] = 0;
    sub_append(sh, quad, wavl, team, 0, kSubRootRel, 0, n0);
  }
  {
    const double* __restrict__ in = level_in(P, root.w) + 3 * (long)root.y;
    for (int j = threadIdx.x; j < 3 * n0; j += kSubThreads) pts[j] = in[j];
  }
  __syncthreads();
  c.base = sh.base;
  bool failed = sh.error != 0;
  for (int par = 0; !failed; par ^= 1, ++c.level) {
    const int nQ = sh.cnt[par][0], nW = sh.cnt[par][1], nT = sh.cnt[par][2];
    if (nQ + nW + nT == 0) break;  // (workgroup-uniform)
    if (c.level + 1 > kMaxLevels) {
      if (threadIdx.x == 0) st->n_nodes.error = 2;
      break;
    }
    __syncthreads();  // everybody has read the counts
    if (threadIdx.x == 0) sh.cnt[par][0] = sh.cnt[par][1] = sh.cnt[par][2] = 0;  // (appended to again two levels down)
    const int pn = par ^ 1;
    // ---- nodes of more than 512 points: teams of 2 / 4 / 8 wavefronts, as many nodes side by side as fit the workgroup
    for (int t0 = 0; t0 < nT;) {
      int w_off = 0, t1 = t0;
      bool mine = false;
      unsigned rel = kSubRootRel;
      int rb = 0, re = 0, tw = 2, sl = 0, first_wave = 0;
      for (; t1 < nT; ++t1) {
        const int2 ent = team[par * kSubTeamMax + t1];
        const int eb = ent.y & 0xffff, ee = ent.y >> 16;
        const int w = team_width(ee - eb);
        if (w_off + w > kSubWaves) break;
        if (wv >= w_off && wv < w_off + w) {
          mine = true;
          rel = (unsigned)ent.x;
          rb = eb; re = ee; tw = w; sl = wv - w_off; first_wave = w_off;
        }
        w_off += w;
      }
      sub_wave_node<true>(P, c, sh, quad, wavl, team, pn, mine, rel, rb, re, tw, sl, first_wave);
      __syncthreads();  // (the team scratch is reused by the next round)
      t0 = t1;
    }
    // ---- 33 .. 512 points: one wavefront per node, no barrier
    for (int k = wv; k < nW; k += kSubWaves) {
      const int2 ent = wavl[par * kSubWaveMax + k];
      sub_wave_node<false>(P, c, sh, quad, wavl, team, pn, true, (unsigned)ent.x, ent.y & 0xffff, ent.y >> 16, 1, 0, wv);
    }
    // ---- up to 32 points: four lanes per node
    for (int k0 = wv * 16; k0 < nQ; k0 += kSubWaves * 16) {
      const int k = k0 + (lane >> 2);
      const bool have = k < nQ;
      const unsigned word = have ? quad[par * kSubQuadMax + k] : 0u;
      sub_quad_nodes(P, c, sh, quad, pn, have, word);
    }
    __syncthreads();  // the next level's lists are complete
    failed = sh.error != 0;
  }
  if (threadIdx.x == 0 && sh.error) st->n_nodes.error = 1;
  __syncthreads();
  // the points in the order the construction left them (diagnostics: madicp_debug_tree_build_points), and the ids of the
  // reservation that no node took marked invalid
  {
    double* __restrict__ order = P.order + 3 * (long)root.y;
    for (int j = threadIdx.x; j < 3 * n0; j += kSubThreads) order[j] = pts[j];
    const int used = sh.next_rel;
    for (int k = used + (int)threadIdx.x; k < 2 * n0; k += kSubThreads)
      if (c.base + k < P.node_cap) P.nodes[c.base + k].flags = 0;
  }
}
