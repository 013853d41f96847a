// Part of madicp_capi.hip (included at its end): the device front-end — scans resident in HBM (madicp_cloud_*), ingest
// and deskew, MAD-tree construction on the device (row f-1).  Everything runs on the context's copy
// stream: like a tree upload it feeds the registrations, so the front-end of scan i+1 overlaps the registration of scan i,
// and the compute stream is only made to wait (by event) where it reads the result.  The one exception is the look-ahead
// build (madicp_tree_build_begin / _end): it runs on a stream of its own, so that a registration's feed — which IS on the
// copy stream — never queues behind half a millisecond of level kernels.
//
// One host synchronisation per build: after the last level the host reads 1 KB of counters (leaf count, top size,
// rho, error flags) because the launch geometry and the buffer sizes of everything downstream need the leaf count.

namespace {

struct DevCloud {
  double* xyz = nullptr;  // (n,3)
  int64_t n = 0;
  hipEvent_t ready = nullptr;  // recorded on the copy stream behind whatever produced xyz
};

// grow-only scratch of the tree builder / deskew / ingest
struct FrontScratch {
  char* block = nullptr;
  size_t cap = 0;
  int64_t n_cap = 0;  // points the block was laid out for
  tb::Params P{};
  uint32_t* S = nullptr;         // (n + 1) scan result
  uint32_t* tile_sums = nullptr; // (n / 1024 + 2)
  // deskew
  double* key[2] = {nullptr, nullptr};
  uint32_t* idx[2] = {nullptr, nullptr};
  int32_t* g = nullptr;
  int32_t* tile_min = nullptr;
  double* table = nullptr;       // thresholds | poses
  void* sort_tmp = nullptr;
  size_t sort_tmp_bytes = 0;
  tb::State* h_state = nullptr;  // pinned: copy of the device State (diagnostics, big clouds)
  tb::HostLine* h_line = nullptr; // pinned: what a build publishes to the host (tb_finish_b)
  int build_seq = 0;
  bool state_stale = false;      // h_state is older than the device State
  double* h_table = nullptr;     // pinned, same layout as `table`
  hipEvent_t h_table_read = nullptr;
  // a construction between its two halves (tree_build_begin_on / tree_build_end_on)
  struct InFlight {
    bool active = false;
    bool lookahead = false;  // begun by madicp_tree_build_begin: owns `cloud`
    hipStream_t s = nullptr;
    tb::Params P{};
    int64_t n = 0;
    bool chip = false;
    int chip_grid = 0, level_grid = 0, n_tiles = 0, levels_done = 0;
    int quiet_from = -1;   // levels past this one are expected to be empty (FrontScratch::last_depth + 1)
    bool use_need = false; // size the steps' launches from the previous build's HostLine::need
    int seq = 0;           // what tb_finish_b will publish (direct scan path)
    DevCloud cloud;        // look-ahead only
    // the emission enqueued behind the summary into a block sized from the previous scan's leaf count (tree_build_begin_on)
    bool pre = false;
    int pre_leaf_cap = 0;
    int pre_steps = 0;     // levels_done when it was enqueued: extra levels afterwards make it worthless
    DevTree pre_tree;
  } fly;
  int last_need[tb::kNeedSteps] = {};  // workgroups every step of the previous build had work for (HostLine::need); -1: unknown
  bool have_need = false;
  int last_leaves = 0;     // leaves of the previous build on this scratch (the next scan of the same sensor: within a few per cent)
  int last_depth = -1;     // deepest level of the previous build on this scratch (consecutive scans: the same +- 1)
  int64_t last_n = 0;      // ... and that build's point count (the hint is for consecutive scans of one sensor, not for any cloud)
};

constexpr int kDeskewTableMax = 1040;  // > CHUNKS + a few: thresholds fall below -pi after ~1024 steps

}  // namespace

// per context, so that contexts driven from different host threads share nothing
struct madicp_ctx::Front {
  std::unordered_map<int, DevCloud> clouds;
  FrontScratch scratch;
};

namespace {

madicp_ctx::Front& front_of(madicp_ctx* ctx) {
  if (!ctx->front) ctx->front = new madicp_ctx::Front();
  return *ctx->front;
}

size_t sort_temp_bytes(int64_t n);  // (defined below, needs rocPRIM)

// the builder's scratch has one owner at a time: between madicp_tree_build_begin and _end it is the look-ahead
int busy_with_lookahead(madicp_ctx* ctx) {
  if (ctx->front && ctx->front->scratch.fly.active)
    return fail(MADICP_ERR_CAPACITY, "a look-ahead tree build is in flight on this context: madicp_tree_build_end (or _cancel) first");
  return MADICP_OK;
}

int ensure_scratch(madicp_ctx* ctx, int64_t n, FrontScratch** out) {
  FrontScratch& fs = front_of(ctx).scratch;
  *out = &fs;
  if (!fs.h_state) {
    HIP_TRY(hipHostMalloc(&fs.h_state, sizeof(tb::State), hipHostMallocDefault));
    HIP_TRY(hipHostMalloc(&fs.h_line, sizeof(tb::HostLine), hipHostMallocDefault));
    std::memset(fs.h_line, 0, sizeof(tb::HostLine));
    HIP_TRY(hipHostMalloc(&fs.h_table, sizeof(double) * kDeskewTableMax * 13, hipHostMallocDefault));
    HIP_TRY(hipEventCreateWithFlags(&fs.h_table_read, hipEventDisableTiming));
  }
  if (n <= fs.n_cap) return MADICP_OK;
  if (fs.block) {
    HIP_TRY(hipStreamSynchronize(ctx->copy));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (ctx->build) HIP_TRY(hipStreamSynchronize(ctx->build));
    HIP_TRY(hipFree(fs.block));
    fs.block = nullptr;
    fs.cap = 0;
    fs.n_cap = 0;
  }
  const int64_t nc = n + n / 8 + 1024;  // head-room: consecutive scans differ by a few per cent
  const size_t slots = (size_t)nc / tb::kChunk + tb::kMaxBig + 8;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    const size_t o = off;
    off = align_up(off + bytes);
    return o;
  };
  const size_t o_state = take(sizeof(tb::State));
  const size_t o_buf0 = take(sizeof(double) * 3 * (size_t)nc);
  const size_t o_buf1 = take(sizeof(double) * 3 * (size_t)nc);
  const size_t o_nodes = take(sizeof(tb::BNode) * 2 * (size_t)nc);
  const size_t o_q0 = take(sizeof(int4) * ((size_t)nc / tb::kSmallMax + 64));  // wave-regime nodes hold > kSmallMax points
  const size_t o_q1 = take(sizeof(int4) * ((size_t)nc / tb::kSmallMax + 64));
  const size_t o_team0 = take(sizeof(int4) * ((size_t)nc / tb::kTeamMin + 64));
  const size_t o_team1 = take(sizeof(int4) * ((size_t)nc / tb::kTeamMin + 64));
  const size_t o_big0 = take(sizeof(int32_t) * tb::kMaxBig);
  const size_t o_big1 = take(sizeof(int32_t) * tb::kMaxBig);
  const size_t o_small0 = take(sizeof(int4) * (size_t)nc);
  const size_t o_small1 = take(sizeof(int4) * (size_t)nc);
  const size_t o_leaf = take(sizeof(uint32_t) * ((size_t)nc + 8));
  const size_t o_S = take(sizeof(uint32_t) * ((size_t)nc + 8));
  const size_t o_tiles = take(sizeof(uint32_t) * ((size_t)nc / tb::kScanTile + 8));
  const size_t o_p1 = take(sizeof(double) * 18 * slots * (tb::kChipLevels + 1));
  const size_t o_p2 = take(sizeof(double) * 8 * slots);
  const size_t o_tab = take(sizeof(int32_t) * ((size_t)nc + 8));
  const size_t o_topids = take(sizeof(int32_t) * kTopMax);
  const size_t o_toplink = take(sizeof(uint32_t) * kTopMax);
  const size_t o_topfront = take(sizeof(int32_t) * (2 + 2 * 1024));
  const size_t o_key0 = take(sizeof(double) * (size_t)nc);
  const size_t o_key1 = take(sizeof(double) * (size_t)nc);
  const size_t o_idx0 = take(sizeof(uint32_t) * (size_t)nc);
  const size_t o_idx1 = take(sizeof(uint32_t) * (size_t)nc);
  const size_t o_g = take(sizeof(int32_t) * (size_t)nc);
  const size_t o_tmin = take(sizeof(int32_t) * ((size_t)nc / tb::kScanTile + 8));
  const size_t o_table = take(sizeof(double) * kDeskewTableMax * 13);
  const size_t sort_bytes = sort_temp_bytes(nc);
  const size_t o_sort = take(sort_bytes);
  HIP_TRY(hipMalloc(&fs.block, off));
  fs.cap = off;
  fs.n_cap = nc;
  char* b = fs.block;
  fs.P = tb::Params{};
  fs.P.st = reinterpret_cast<tb::State*>(b + o_state);
  fs.P.buf[0] = reinterpret_cast<double*>(b + o_buf0);
  fs.P.buf[1] = reinterpret_cast<double*>(b + o_buf1);
  fs.P.nodes = reinterpret_cast<tb::BNode*>(b + o_nodes);
  fs.P.node_cap = static_cast<int32_t>(std::min<int64_t>(2 * nc, 0x7ffffff0));
  fs.P.q[0] = reinterpret_cast<int4*>(b + o_q0);
  fs.P.q[1] = reinterpret_cast<int4*>(b + o_q1);
  fs.P.team[0] = reinterpret_cast<int4*>(b + o_team0);
  fs.P.team[1] = reinterpret_cast<int4*>(b + o_team1);
  fs.P.big[0] = reinterpret_cast<int32_t*>(b + o_big0);
  fs.P.big[1] = reinterpret_cast<int32_t*>(b + o_big1);
  fs.P.small[0] = reinterpret_cast<int4*>(b + o_small0);
  fs.P.small[1] = reinterpret_cast<int4*>(b + o_small1);
  fs.P.leaf_start = reinterpret_cast<uint32_t*>(b + o_leaf);
  fs.P.partLR = reinterpret_cast<double*>(b + o_p1);
  fs.P.part_stride = (long)(18 * slots);
  fs.P.part2 = reinterpret_cast<double*>(b + o_p2);
  fs.P.tab = reinterpret_cast<int32_t*>(b + o_tab);
  fs.P.top_ids = reinterpret_cast<int32_t*>(b + o_topids);
  fs.P.top_link = reinterpret_cast<uint32_t*>(b + o_toplink);
  fs.P.top_front = reinterpret_cast<int32_t*>(b + o_topfront);
  fs.S = reinterpret_cast<uint32_t*>(b + o_S);
  fs.tile_sums = reinterpret_cast<uint32_t*>(b + o_tiles);
  fs.P.S = fs.S;
  fs.P.tile_sums = fs.tile_sums;
  fs.key[0] = reinterpret_cast<double*>(b + o_key0);
  fs.key[1] = reinterpret_cast<double*>(b + o_key1);
  fs.idx[0] = reinterpret_cast<uint32_t*>(b + o_idx0);
  fs.idx[1] = reinterpret_cast<uint32_t*>(b + o_idx1);
  fs.g = reinterpret_cast<int32_t*>(b + o_g);
  fs.tile_min = reinterpret_cast<int32_t*>(b + o_tmin);
  fs.table = reinterpret_cast<double*>(b + o_table);
  fs.sort_tmp = b + o_sort;
  fs.sort_tmp_bytes = sort_bytes;
  return MADICP_OK;
}

void front_destroy(madicp_ctx* ctx) {  // called by madicp_ctx_destroy (streams already drained)
  if (!ctx->front) return;
  for (auto& c : ctx->front->clouds)
    if (c.second.ready) hipEventDestroy(c.second.ready);  // (the device buffers belong to the pool)
  FrontScratch& fs = ctx->front->scratch;
  if (fs.fly.pre) {  // (a construction that was begun and never ended: its pre-sized tree block goes back with the pool)
    pool_free(ctx, fs.fly.pre_tree.block, nullptr);
    fs.fly.pre = false;
  }
  if (fs.block) hipFree(fs.block);
  if (fs.h_state) hipHostFree(fs.h_state);
  if (fs.h_line) hipHostFree(fs.h_line);
  if (fs.h_table) hipHostFree(fs.h_table);
  if (fs.h_table_read) hipEventDestroy(fs.h_table_read);
  delete ctx->front;
  ctx->front = nullptr;
}

DevCloud* find_cloud(madicp_ctx* ctx, int id) {
  if (!ctx->front) return nullptr;
  auto it = ctx->front->clouds.find(id);
  return it == ctx->front->clouds.end() ? nullptr : &it->second;
}

int new_cloud(madicp_ctx* ctx, int64_t n, DevCloud* out) {
  void* p = nullptr;
  RC_TRY(pool_alloc(ctx, sizeof(double) * 3 * (size_t)std::max<int64_t>(n, 1), ctx->copy, &p));
  out->xyz = static_cast<double*>(p);
  out->n = n;
  HIP_TRY(hipEventCreateWithFlags(&out->ready, hipEventDisableTiming));
  return MADICP_OK;
}

// a cloud that was allocated but never registered (an error on the way): buffer back to the pool, event destroyed
int drop_cloud(madicp_ctx* ctx, DevCloud& c, int rc) {
  EventRef after;
  if (fence_event(ctx, &after) != MADICP_OK) after = nullptr;
  pool_free(ctx, c.xyz, after);
  if (c.ready) hipEventDestroy(c.ready);
  c = DevCloud{};
  return rc;
}
#define CLOUD_TRY(expr)                                                                                          \
  do {                                                                                                           \
    hipError_t e_ = (expr);                                                                                      \
    if (e_ != hipSuccess) return drop_cloud(ctx, c, fail(MADICP_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_))); \
  } while (0)

int scan_marks(hipStream_t s, FrontScratch& fs, const uint32_t* marks, int64_t n, int32_t* d_total) {
  const int tiles = static_cast<int>((n + 1 + tb::kScanTile - 1) / tb::kScanTile);
  hipLaunchKernelGGL(tb::tb_scan_tiles, dim3(tiles), dim3(256), 0, s, marks, (int)n, fs.tile_sums);
  hipLaunchKernelGGL(tb::tb_scan_top, dim3(1), dim3(256), 0, s, fs.tile_sums, tiles, d_total);
  hipLaunchKernelGGL(tb::tb_scan_apply, dim3(tiles), dim3(256), 0, s, marks, (int)n, (const uint32_t*)fs.tile_sums, fs.S);
  HIP_TRY(hipGetLastError());
  return MADICP_OK;
}

// dst[i] = (float)src[i] for a block of values; false as soon as a block holds a value that is not exactly a float (NaN included:
// it compares unequal to itself; -0.0 and +-inf pass and survive the round trip).  Four doubles per instruction where the host
// has AVX2 (the plain loop, two per instruction with SSE2, was slower than the copy it replaces: 95 us against 58 for a scan).
bool block_to_f32_exact_plain(const double* src, float* dst, size_t count) {
  int bad = 0;
  for (size_t i = 0; i < count; ++i) {
    const float f = static_cast<float>(src[i]);
    dst[i] = f;
    bad |= (static_cast<double>(f) != src[i]);
  }
  return bad == 0;
}
#if !defined(__HIP_DEVICE_COMPILE__) && defined(__x86_64__)
}  // namespace
#include <immintrin.h>
namespace {
__attribute__((target("avx2"))) bool block_to_f32_exact_avx2(const double* src, float* dst, size_t count) {
  __m256d bad = _mm256_setzero_pd();
  size_t i = 0;
  for (; i + 8 <= count; i += 8) {
    const __m256d x0 = _mm256_loadu_pd(src + i), x1 = _mm256_loadu_pd(src + i + 4);
    const __m128 f0 = _mm256_cvtpd_ps(x0), f1 = _mm256_cvtpd_ps(x1);
    _mm_storeu_ps(dst + i, f0);
    _mm_storeu_ps(dst + i + 4, f1);
    bad = _mm256_or_pd(bad, _mm256_or_pd(_mm256_cmp_pd(_mm256_cvtps_pd(f0), x0, _CMP_NEQ_UQ), _mm256_cmp_pd(_mm256_cvtps_pd(f1), x1, _CMP_NEQ_UQ)));
  }
  return _mm256_movemask_pd(bad) == 0 && block_to_f32_exact_plain(src + i, dst + i, count - i);
}
__attribute__((target("avx512f"))) bool block_to_f32_exact_avx512(const double* src, float* dst, size_t count) {
  __mmask8 bad = 0;
  size_t i = 0;
  for (; i + 16 <= count; i += 16) {
    const __m512d x0 = _mm512_loadu_pd(src + i), x1 = _mm512_loadu_pd(src + i + 8);
    const __m256 f0 = _mm512_cvtpd_ps(x0), f1 = _mm512_cvtpd_ps(x1);
    _mm256_storeu_ps(dst + i, f0);
    _mm256_storeu_ps(dst + i + 8, f1);
    bad |= _mm512_cmp_pd_mask(_mm512_cvtps_pd(f0), x0, _CMP_NEQ_UQ) | _mm512_cmp_pd_mask(_mm512_cvtps_pd(f1), x1, _CMP_NEQ_UQ);
  }
  return bad == 0 && block_to_f32_exact_plain(src + i, dst + i, count - i);
}
int host_simd_level() {  // 2: AVX-512F, 1: AVX2, 0: neither
  static const int level = __builtin_cpu_supports("avx512f") ? 2 : (__builtin_cpu_supports("avx2") ? 1 : 0);
  return level;
}
#else
bool block_to_f32_exact_avx2(const double* src, float* dst, size_t count) { return block_to_f32_exact_plain(src, dst, count); }
bool block_to_f32_exact_avx512(const double* src, float* dst, size_t count) { return block_to_f32_exact_plain(src, dst, count); }
int host_simd_level() { return 0; }
#endif
bool block_to_f32_exact(const double* src, float* dst, size_t count) {
  constexpr size_t kBlock = 8192;  // (a cloud of genuine doubles is found out in its first block)
  const int simd = host_simd_level();
  for (size_t b = 0; b < count; b += kBlock) {
    const size_t len = std::min(kBlock, count - b);
    const bool ok = simd == 2 ? block_to_f32_exact_avx512(src + b, dst + b, len)
                              : (simd == 1 ? block_to_f32_exact_avx2(src + b, dst + b, len) : block_to_f32_exact_plain(src + b, dst + b, len));
    if (!ok) return false;
  }
  return true;
}

// The caller's cloud (pageable host memory) -> pinned staging -> `d_xyz` on stream `s`, staged and sent in pieces: the copy
// engine moves piece k while the host stages piece k + 1.  A cloud whose coordinates are ALL exactly floats — what a sensor
// driver, a KITTI .bin or a PointCloud2 delivers, converted to double by the caller — crosses PCIe as floats (half the
// bytes, half the transfer time in front of the builder's first kernel) and is widened on the device: the same doubles, bit
// for bit.  The first value that is not a float (synthetic double-precision noise: the first point) sends the rest the plain way.
int stage_and_send_cloud(madicp_ctx* ctx, const double* xyz, int64_t n, double* d_xyz, hipStream_t s) {
  const size_t n3 = 3 * (size_t)n;
  const size_t bytes = sizeof(double) * n3;
  const int hb = ctx->h_tree_next;
  ctx->h_tree_next ^= 1;
  HIP_TRY(hipEventSynchronize(ctx->h_tree_ev[hb]));
  if (ctx->h_tree_cap[hb] < bytes) {
    if (ctx->h_tree[hb]) HIP_TRY(hipHostFree(ctx->h_tree[hb]));
    ctx->h_tree[hb] = nullptr;
    ctx->h_tree_cap[hb] = 0;
    const size_t cap = bytes + bytes / 4;
    HIP_TRY(hipHostMalloc(&ctx->h_tree[hb], cap, hipHostMallocDefault));
    ctx->h_tree_cap[hb] = cap;
  }
  char* stage = ctx->h_tree[hb];
  size_t done3 = 0;  // values already on their way as floats
  void* d_f32 = nullptr;
  if (ctx->upload_f32 && n3 >= 4096) {
    float* hf = reinterpret_cast<float*>(stage);
    const size_t piece3 = std::max<size_t>((n3 / 2 + 3) / 4 * 4, (size_t)64 << 10);  // two pieces (half the bytes: half the pieces)
    for (size_t off = 0; off < n3; off += piece3) {
      const size_t len = std::min(piece3, n3 - off);
      if (!block_to_f32_exact(xyz + off, hf + off, len)) break;
      if (!d_f32) {  // (one device block per staging block: the event that frees the staging block is behind the widening too)
        if (ctx->d_f32_cap[hb] < sizeof(float) * n3) {
          if (ctx->d_f32[hb]) HIP_TRY(hipFree(ctx->d_f32[hb]));
          ctx->d_f32[hb] = nullptr;
          ctx->d_f32_cap[hb] = 0;
          const size_t cap = sizeof(float) * (n3 + n3 / 4);
          HIP_TRY(hipMalloc(&ctx->d_f32[hb], cap));
          ctx->d_f32_cap[hb] = cap;
        }
        d_f32 = ctx->d_f32[hb];
      }
      HIP_TRY(hipMemcpyAsync(static_cast<float*>(d_f32) + off, hf + off, sizeof(float) * len, hipMemcpyHostToDevice, s));
      done3 = off + len;
    }
    if (done3 > 0) {
      hipLaunchKernelGGL(fe::cloud_widen_f32, dim3((unsigned)((done3 / 4 + 256) / 256)), dim3(256), 0, s, (const float*)d_f32, d_xyz, (long)done3);
      HIP_TRY(hipGetLastError());
    }
    if (done3 == n3) {
      HIP_TRY(hipEventRecord(ctx->h_tree_ev[hb], s));
      return MADICP_OK;
    }
  }
  {  // (the rest — everything, for a cloud of genuine doubles — as doubles, behind the floats' region of the staging block)
    const size_t off0 = sizeof(double) * done3;
    const size_t piece = std::max<size_t>(align_up(bytes / 4), 256 << 10);
    for (size_t off = off0; off < bytes; off += piece) {
      const size_t len = std::min(piece, bytes - off);
      std::memcpy(stage + off, reinterpret_cast<const char*>(xyz) + off, len);
      HIP_TRY(hipMemcpyAsync(reinterpret_cast<char*>(d_xyz) + off, stage + off, len, hipMemcpyHostToDevice, s));
    }
  }
  HIP_TRY(hipEventRecord(ctx->h_tree_ev[hb], s));
  return MADICP_OK;
}

}  // namespace

extern "C" {

int madicp_cloud_upload(madicp_ctx* ctx, const double* xyz, int64_t n, int* out_cloud_id) {
  if (!ctx || !xyz || !out_cloud_id) return fail(MADICP_ERR_INVALID, "null argument");
  if (n < 1 || n > 0x3fffffff) return fail(MADICP_ERR_INVALID, "a cloud holds 1 .. 2^30 points");
  HIP_TRY(hipSetDevice(ctx->device));
  DevCloud c;
  RC_TRY(new_cloud(ctx, n, &c));
  // pinned staging shared with the tree uploads (two buffers, alternating)
  {
    const int rc = stage_and_send_cloud(ctx, xyz, n, c.xyz, ctx->copy);
    if (rc != MADICP_OK) return drop_cloud(ctx, c, rc);
  }
  CLOUD_TRY(hipEventRecord(c.ready, ctx->copy));
  const int id = ctx->next_id++;
  front_of(ctx).clouds[id] = c;
  *out_cloud_id = id;
  return MADICP_OK;
}

int madicp_cloud_release(madicp_ctx* ctx, int cloud_id) {
  if (!ctx) return fail(MADICP_ERR_INVALID, "ctx is null");
  DevCloud* c = find_cloud(ctx, cloud_id);
  if (!c) return fail(MADICP_ERR_INVALID, "unknown cloud id");
  HIP_TRY(hipSetDevice(ctx->device));
  EventRef after;
  RC_TRY(fence_event(ctx, &after));
  pool_free(ctx, c->xyz, after);
  if (c->ready) hipEventDestroy(c->ready);
  ctx->front->clouds.erase(cloud_id);
  return MADICP_OK;
}

int madicp_cloud_size(madicp_ctx* ctx, int cloud_id, int64_t* out_n) {
  if (!ctx || !out_n) return fail(MADICP_ERR_INVALID, "null argument");
  DevCloud* c = find_cloud(ctx, cloud_id);
  if (!c) return fail(MADICP_ERR_INVALID, "unknown cloud id");
  *out_n = c->n;
  return MADICP_OK;
}

int madicp_cloud_download(madicp_ctx* ctx, int cloud_id, double* out_xyz, int64_t n) {
  if (!ctx || !out_xyz) return fail(MADICP_ERR_INVALID, "null argument");
  DevCloud* c = find_cloud(ctx, cloud_id);
  if (!c) return fail(MADICP_ERR_INVALID, "unknown cloud id");
  if (n != c->n) return fail(MADICP_ERR_INVALID, "n mismatch");
  HIP_TRY(hipSetDevice(ctx->device));
  HIP_TRY(hipMemcpyAsync(out_xyz, c->xyz, sizeof(double) * 3 * (size_t)n, hipMemcpyDeviceToHost, ctx->copy));
  HIP_TRY(hipStreamSynchronize(ctx->copy));
  return MADICP_OK;
}

int madicp_cloud_ingest_f32(madicp_ctx* ctx, const float* records, int64_t n_records, int stride_floats, double min_range,
                            double max_range, int kitti_correction, int* out_cloud_id, int64_t* out_n) {
  if (!ctx || !records || !out_cloud_id || !out_n) return fail(MADICP_ERR_INVALID, "null argument");
  if (n_records < 1 || n_records > 0x3fffffff) return fail(MADICP_ERR_INVALID, "1 .. 2^30 records");
  if (stride_floats < 3) return fail(MADICP_ERR_INVALID, "a record holds at least x, y, z");
  RC_TRY(busy_with_lookahead(ctx));
  HIP_TRY(hipSetDevice(ctx->device));
  FrontScratch* fs = nullptr;
  // the raw records go through the pinned staging into buf[0] of the scratch (a KITTI record is 16 bytes, a point of the
  // scratch 24: wider records ask for a scratch laid out for proportionally more points)
  const size_t bytes = sizeof(float) * (size_t)stride_floats * (size_t)n_records;
  const int64_t n_layout = std::max<int64_t>(n_records, (int64_t)((bytes + 23) / 24));
  if (n_layout > 0x3fffffff) return fail(MADICP_ERR_INVALID, "records too large");
  RC_TRY(ensure_scratch(ctx, n_layout, &fs));
  const int hb = ctx->h_tree_next;
  ctx->h_tree_next ^= 1;
  HIP_TRY(hipEventSynchronize(ctx->h_tree_ev[hb]));
  if (ctx->h_tree_cap[hb] < bytes) {
    if (ctx->h_tree[hb]) HIP_TRY(hipHostFree(ctx->h_tree[hb]));
    ctx->h_tree[hb] = nullptr;
    ctx->h_tree_cap[hb] = 0;
    const size_t cap = bytes + bytes / 4;
    HIP_TRY(hipHostMalloc(&ctx->h_tree[hb], cap, hipHostMallocDefault));
    ctx->h_tree_cap[hb] = cap;
  }
  std::memcpy(ctx->h_tree[hb], records, bytes);
  float* d_rec = reinterpret_cast<float*>(fs->P.buf[0]);
  HIP_TRY(hipMemcpyAsync(d_rec, ctx->h_tree[hb], bytes, hipMemcpyHostToDevice, ctx->copy));
  HIP_TRY(hipEventRecord(ctx->h_tree_ev[hb], ctx->copy));
  uint32_t* keep = fs->P.leaf_start;
  const int blocks = static_cast<int>(std::min<int64_t>((n_records + 255) / 256, (int64_t)ctx->n_cus * 8));
  hipLaunchKernelGGL(fe::ingest_mark, dim3(blocks), dim3(256), 0, ctx->copy, (const float*)d_rec, (long)n_records, stride_floats,
                     min_range, max_range, keep);
  RC_TRY(scan_marks(ctx->copy, *fs, keep, n_records, &fs->P.st->n_leaves));
  HIP_TRY(hipMemcpyAsync(&fs->h_state->n_leaves, &fs->P.st->n_leaves, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->copy));
  HIP_TRY(hipStreamSynchronize(ctx->copy));  // the size of the result decides the allocation
  const int64_t kept = fs->h_state->n_leaves;
  if (kept < 1) return fail(MADICP_ERR_INVALID, "no point survives the range filter");
  DevCloud c;
  RC_TRY(new_cloud(ctx, kept, &c));
  // VERTICAL_ANGLE_OFFSET = (0.205 * M_PI) / 180.0 (bin_runner.cpp:55); libm sin / cos like Eigen::AngleAxisd
  const double angle = (0.205 * M_PI) / 180.0;
  hipLaunchKernelGGL(fe::ingest_scatter, dim3(blocks), dim3(256), 0, ctx->copy, (const float*)d_rec, (long)n_records, stride_floats,
                     (const uint32_t*)keep, (const uint32_t*)fs->S, kitti_correction ? 1 : 0, std::sin(angle), std::cos(angle), c.xyz);
  CLOUD_TRY(hipGetLastError());
  CLOUD_TRY(hipEventRecord(c.ready, ctx->copy));
  const int id = ctx->next_id++;
  front_of(ctx).clouds[id] = c;
  *out_cloud_id = id;
  *out_n = kept;
  return MADICP_OK;
}

}  // extern "C"

// ---- deskew --------------------------------------------------------------------------------------------------------
#include <rocprim/rocprim.hpp>

namespace {
size_t sort_temp_bytes(int64_t n) {
  size_t bytes = 0;
  double* k = nullptr;
  uint32_t* v = nullptr;
  (void)rocprim::radix_sort_pairs(nullptr, bytes, k, k, v, v, (size_t)n, 0, 64, (hipStream_t) nullptr);
  return bytes + 256;
}
}  // namespace

extern "C" {

int madicp_cloud_deskew(madicp_ctx* ctx, int cloud_id, const double velocity[6], double sensor_hz, int32_t* out_chunks) {
  if (!ctx || !velocity) return fail(MADICP_ERR_INVALID, "null argument");
  DevCloud* c = find_cloud(ctx, cloud_id);
  if (!c) return fail(MADICP_ERR_INVALID, "unknown cloud id");
  if (!(sensor_hz > 0.0)) return fail(MADICP_ERR_INVALID, "sensor_hz must be positive");
  RC_TRY(busy_with_lookahead(ctx));
  HIP_TRY(hipSetDevice(ctx->device));
  FrontScratch* fs = nullptr;
  RC_TRY(ensure_scratch(ctx, c->n, &fs));
  const int64_t n = c->n;
  // The reference's running threshold and time (pipeline.cpp:99-106,109-117), tabulated with its own arithmetic
  // (repeated subtraction / addition), and the pose of every chunk by the host's expSO3 (libm, like the reference).
  constexpr int CHUNKS = 1024;  // tools/constants.h:31
  const double ts = 1. / sensor_hz;
  const double resolution = 2 * M_PI / double(CHUNKS);
  const double delta = ts / double(CHUNKS - 1);
  HIP_TRY(hipEventSynchronize(fs->h_table_read));
  double* thr = fs->h_table;
  double* poses = fs->h_table + kDeskewTableMax;
  double angle = M_PI - resolution;
  double t = -ts;
  int n_thr = 0;
  for (int k = 0; k < kDeskewTableMax; ++k) {
    double dx[6];
    for (int i = 0; i < 6; ++i) dx[i] = velocity[i] * t;
    double* Pk = poses + 12 * k;
    {  // lie_algebra.h:39-52, with the reference's first-order branch
      const double* w = dx + 3;
      const double th2 = dotc(w[0], w[1], w[2], w[0], w[1], w[2]);
      const double th = std::sqrt(th2);
      const double W[9] = {0.0, -w[2], w[1], w[2], 0.0, -w[0], -w[1], w[0], 0.0};
      if (th2 < 1e-8) {
        for (int i = 0; i < 9; ++i) Pk[i] = ((i % 4 == 0) ? 1.0 : 0.0) + W[i];
      } else {
        double K[9], cK[9];
        const double omc = 2.0 * std::sin(th / 2.0) * std::sin(th / 2.0);
        const double s = std::sin(th);
        for (int i = 0; i < 9; ++i) {
          K[i] = W[i] / th;
          cK[i] = omc * K[i];
        }
        for (int r = 0; r < 3; ++r)
          for (int q = 0; q < 3; ++q) {
            const double kk = cK[3 * r] * K[q] + (cK[3 * r + 1] * K[3 + q] + cK[3 * r + 2] * K[6 + q]);
            Pk[3 * r + q] = (((r == q) ? 1.0 : 0.0) + s * K[3 * r + q]) + kk;
          }
      }
      Pk[9] = dx[0]; Pk[10] = dx[1]; Pk[11] = dx[2];
    }
    thr[k] = angle;
    if (angle >= -M_PI - resolution) n_thr = k + 1;  // thresholds at or below -pi can never be undercut again
    angle -= resolution;
    t += delta;
  }
  const int n_poses = std::min(n_thr + 1, kDeskewTableMax);
  HIP_TRY(hipMemcpyAsync(fs->table, fs->h_table, sizeof(double) * kDeskewTableMax * 13, hipMemcpyHostToDevice, ctx->copy));
  HIP_TRY(hipEventRecord(fs->h_table_read, ctx->copy));
  const int blocks = static_cast<int>(std::min<int64_t>((n + 255) / 256, (int64_t)ctx->n_cus * 8));
  hipLaunchKernelGGL(fe::deskew_keys, dim3(blocks), dim3(256), 0, ctx->copy, (const double*)c->xyz, (long)n, fs->key[0], fs->idx[0]);
  HIP_TRY(hipGetLastError());
  size_t tmp_bytes = fs->sort_tmp_bytes;
  HIP_TRY(rocprim::radix_sort_pairs(fs->sort_tmp, tmp_bytes, fs->key[0], fs->key[1], fs->idx[0], fs->idx[1], (size_t)n, 0, 64,
                                    ctx->copy));
  hipLaunchKernelGGL(fe::deskew_targets, dim3(blocks), dim3(256), 0, ctx->copy, (const double*)fs->key[1], (long)n,
                     (const double*)fs->table, n_thr, fs->g);
  const int tiles = static_cast<int>((n + tb::kScanTile - 1) / tb::kScanTile);
  hipLaunchKernelGGL(fe::pmin_tiles, dim3(tiles), dim3(256), 0, ctx->copy, (const int32_t*)fs->g, (long)n, fs->tile_min);
  hipLaunchKernelGGL(fe::pmin_top, dim3(1), dim3(256), 0, ctx->copy, fs->tile_min, tiles);
  // the compensated cloud replaces the input: written to a fresh buffer, the old one goes back to the pool
  void* fresh = nullptr;
  RC_TRY(pool_alloc(ctx, sizeof(double) * 3 * (size_t)n, ctx->copy, &fresh));
  int32_t* d_chunks = out_chunks ? reinterpret_cast<int32_t*>(fs->P.small[0]) : nullptr;
  hipLaunchKernelGGL(fe::deskew_apply, dim3(tiles), dim3(256), 0, ctx->copy, (const double*)c->xyz, (const uint32_t*)fs->idx[1], (long)n,
                     (const int32_t*)fs->g, (const int32_t*)fs->tile_min, (const double*)(fs->table + kDeskewTableMax), n_poses,
                     static_cast<double*>(fresh), d_chunks);
  EventRef after;
  {  // a failure from here on must not leak the fresh buffer: the cloud keeps its old points
    const hipError_t le = hipGetLastError();
    const int frc = le == hipSuccess ? fence_event(ctx, &after) : MADICP_OK;
    if (le != hipSuccess || frc != MADICP_OK) {
      hipStreamSynchronize(ctx->copy);
      pool_free(ctx, fresh, nullptr);
      return le != hipSuccess ? fail(MADICP_ERR_DEVICE, std::string("deskew: ") + hipGetErrorString(le)) : frc;
    }
  }
  pool_free(ctx, c->xyz, after);
  c->xyz = static_cast<double*>(fresh);
  HIP_TRY(hipEventRecord(c->ready, ctx->copy));
  if (out_chunks) {  // debugging / parity aid: the time chunk of every point, in walk order (largest azimuth first)
    HIP_TRY(hipMemcpyAsync(out_chunks, d_chunks, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, ctx->copy));
    HIP_TRY(hipStreamSynchronize(ctx->copy));
  }
  return MADICP_OK;
}

}  // extern "C"

// ---- MAD-tree construction on the device ------------------------------------------------------------------------------
namespace {

// the level kernels of steps [from, to) of the construction in flight, on its stream
// The breadth-first layout of the tree's LDS-staged top is made in three parts, each as the first workgroup of a level launch.
// Part [a, b) reads the nodes of levels a .. b, so they must be FINISHED, and a node of level L is finished by step L + 5 at
// the latest, not L: a small node born during the chip levels waits for step kChipLevels, and its descendants follow one
// step per level from there (a child of the root's small child, level 2, is step 7).  Hence step b + 6.  (Looking at the
// nodes themselves does not work: the array is not cleared between builds, an unwritten slot looks like last build's node.)
struct TopPart { int from, to, step; };
#ifndef MADICP_TB_TOP_EARLY  // (development: -DMADICP_TB_TOP_EARLY=5 runs the parts at step b + 1 — the schedule that looks right and is
#define MADICP_TB_TOP_EARLY 0  //  a race: the 200-frame drive of tests/test_gpu_frontend.py ended 1.8e-2 m off the host path with it)
#endif
constexpr int kTopLag = tb::kChipLevels - MADICP_TB_TOP_EARLY;
constexpr TopPart kTopParts[3] = {{0, 6, 6 + kTopLag}, {6, 9, 9 + kTopLag}, {9, kTopLevels, kTopLevels + kTopLag}};
static_assert(kTopLevels + kTopLag < 20, "the last part needs a step among the twenty that are always launched");

int tb_run_levels(FrontScratch& fs, int from, int to) {
  FrontScratch::InFlight& f = fs.fly;
  const tb::Params& P = f.P;
  hipStream_t s = f.s;
  for (int level = from; level < to; ++level) {
    if (f.chip && level < tb::kChipLevels) {
      hipLaunchKernelGGL(tb::tb_chip_stats, dim3(f.chip_grid), dim3(256), 0, s, P, level);
      hipLaunchKernelGGL(tb::tb_chip_scatter, dim3(f.chip_grid), dim3(256), 0, s, P, level);
    }
    if (level < P.first_step) continue;  // (wave / quad nodes born up here wait for step first_step: tree_build.hip.h)
    // a level has at most 2^level nodes: the early levels get a handful of workgroups, not the full grid (hundreds of
    // workgroups that only look at an empty queue still cost their dispatch)
    const int64_t nodes_max = level < 30 ? std::min<int64_t>((int64_t)1 << level, f.n) : f.n;
    // (+ one workgroup per team-regime node: at most n / kTeamMin of them, and none on a level that holds fewer nodes)
    const int64_t team_max = std::min<int64_t>(nodes_max, f.n / tb::kTeamMin);
    int grid = static_cast<int>(std::max<int64_t>(1, std::min<int64_t>(f.level_grid, (nodes_max + 3) / 4 + (nodes_max + 63) / 64 + 1) + team_max));
    // levels the previous build did not reach are launched all the same (this tree may be deeper) but with a small grid — the
    // queues are walked with a stride, so any grid is correct, and 1 600 workgroups that find an empty queue cost 4.6 us
    if (f.quiet_from >= 0 && level > f.quiet_from) grid = std::min(grid, 96);
    // ... and a step the previous scan of this sensor had little work for gets a launch of that size + half (the queues are
    // walked with a stride: a larger crop is slower for one build, never wrong)
    if (f.use_need && level < tb::kNeedSteps) grid = std::min(grid, fs.last_need[level] + fs.last_need[level] / 2 + 24);
    // the breadth-first layout of the tree's LDS-staged top: the FIRST workgroup of three level launches (kTopParts above)
    int bfs_from = 0, bfs_to = 0;
    for (const TopPart& tp : kTopParts)
      if (level == tp.step) { bfs_from = tp.from; bfs_to = tp.to; }
    hipLaunchKernelGGL(tb::tb_level, dim3(grid + (bfs_to > bfs_from ? 1 : 0)), dim3(256), 0, s, P, level, bfs_from, bfs_to, kTopLevels, kTopMax);
  }
  return MADICP_OK;
}

// What the host needs before it can size the tree — leaf count (scan of the leaf starts), root mean, rho, size of the
// LDS-staged top, error flags, whether a queue is still waiting — arrives in fs.h_line.  Two halves: the kernels that
// produce it (direct scan path only; huge clouds do everything in the second half) ...
int tb_summary_enqueue(FrontScratch& fs, int next_step, bool again) {
  FrontScratch::InFlight& f = fs.fly;
  if (again)  // (a fresh State is all zero; a second summary must not add to the first)
    HIP_TRY(hipMemsetAsync(&f.P.st->n_leaves, 0, offsetof(tb::State, q_count) - offsetof(tb::State, n_leaves), f.s));  // the results line
  if (f.n_tiles > tb::kScanDirectMax) return MADICP_OK;
  f.seq = ++fs.build_seq;
  hipLaunchKernelGGL(tb::tb_finish_a, dim3(f.n_tiles + 64), dim3(256), 0, f.s, f.P, kTopLevels, f.n_tiles);
  hipLaunchKernelGGL(tb::tb_finish_b, dim3(f.n_tiles), dim3(256), 0, f.s, f.P, f.n_tiles, next_step, fs.h_line, f.seq);
  HIP_TRY(hipGetLastError());
  return MADICP_OK;
}
// ... and the wait for it
int tb_summary_wait(madicp_ctx* ctx, FrontScratch& fs, int next_step) {
  FrontScratch::InFlight& f = fs.fly;
  tb::HostLine& hl = *fs.h_line;
  if (f.n_tiles <= tb::kScanDirectMax) {
    const int seq = f.seq;
    const unsigned check_mask = ctx->wait_mode == 0 ? 0x3ffu : 0xfu;  // (option "wait_mode": spin / yield / sleep)
    for (unsigned spins = 1; __atomic_load_n(&hl.seq, __ATOMIC_ACQUIRE) != seq; ++spins) {
      if ((spins & check_mask) == 0) {  // every few tens of microseconds: is the stream still alive?
        const hipError_t q = hipStreamQuery(f.s);
        if (q == hipSuccess) {
          if (__atomic_load_n(&hl.seq, __ATOMIC_ACQUIRE) == seq) break;
          return fail(MADICP_ERR_DEVICE, "tree build: finished without publishing its summary");
        }
        if (q != hipErrorNotReady) return fail(MADICP_ERR_DEVICE, std::string("tree build: ") + hipGetErrorString(q));
      }
      wait_pause(ctx);
    }
    return MADICP_OK;
  }
  // huge clouds: three-kernel scan, summary, the State copied back
  RC_TRY(scan_marks(f.s, fs, f.P.leaf_start, f.n, &f.P.st->n_leaves));
  hipLaunchKernelGGL(tb::tb_summary, dim3(64), dim3(256), 0, f.s, f.P, kTopLevels);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(fs.h_state, f.P.st, sizeof(tb::State), hipMemcpyDeviceToHost, f.s));
  HIP_TRY(hipStreamSynchronize(f.s));
  const tb::State& h = *fs.h_state;
  hl.n_nodes = h.n_nodes.v; hl.error = h.n_nodes.error; hl.n_leaves = h.n_leaves; hl.n_top = h.n_top;
  hl.max_level = h.max_level; hl.n_valid = h.n_valid; hl.rho_bits = h.rho_bits;
  hl.pending_wave = h.q_count[next_step].v + h.team_count[next_step].v; hl.pending_quad = h.small_count[next_step].v;
  for (int i = 0; i < 3; ++i) hl.origin[i] = h.origin[i];
  return MADICP_OK;
}

// a built tree's block: [nodes | top exit | top dfs | top link | screening records | leaf records | top records], laid out
// for CAPACITIES (the exact counts when the host knows them, the previous scan's with head-room when it does not yet)
size_t layout_built_tree(DevTree& t, char* blk, size_t node_cap, size_t leaf_cap, size_t top_cap) {
  const size_t off_nodes = 0;
  const size_t off_exit = align_up(off_nodes + sizeof(madicp_node) * node_cap);
  const size_t off_dfs = align_up(off_exit + sizeof(int4) * top_cap);
  const size_t off_link = align_up(off_dfs + sizeof(int) * top_cap);
  const size_t off_cnodes = align_up(off_link + sizeof(unsigned int) * top_cap);
  const size_t off_leaves = align_up(off_cnodes + sizeof(CNode) * node_cap);
  const size_t off_top = align_up(off_leaves + sizeof(LeafRec) * leaf_cap);
  const size_t total = align_up(off_top + sizeof(CNode) * std::max<size_t>(top_cap, 1));
  if (blk) {
    t.block = blk;
    t.nodes = reinterpret_cast<madicp_node*>(blk + off_nodes);
    t.top_exit = top_cap ? reinterpret_cast<int4*>(blk + off_exit) : nullptr;
    t.top_dfs = top_cap ? reinterpret_cast<int*>(blk + off_dfs) : nullptr;
    t.top_link = top_cap ? reinterpret_cast<unsigned int*>(blk + off_link) : nullptr;
    t.cnodes = reinterpret_cast<CNode*>(blk + off_cnodes);
    t.leaves = reinterpret_cast<LeafRec*>(blk + off_leaves);
    t.top = top_cap ? reinterpret_cast<CNode*>(blk + off_top) : nullptr;
  }
  return total;
}

// first half of a construction: everything up to the summary of step 20 is enqueued on `s`; nothing is waited for
int tree_build_begin_on(madicp_ctx* ctx, FrontScratch& fs, const double* d_xyz, int64_t n, double b_max, double b_min, hipStream_t s) {
  FrontScratch::InFlight& f = fs.fly;
  f.s = s;
  f.n = n;
  f.P = fs.P;
  f.P.cloud = d_xyz;
  f.P.n_points = static_cast<int32_t>(n);
  f.P.b_max = b_max;
  f.P.b_min = b_min;
  f.chip = n > tb::kChipMin;
  f.P.first_step = f.chip ? tb::kChipLevels : 0;
  // (State and leaf-start marks are cleared by tb_init itself)
  {  // one workgroup for the State and the root, some to clear the leaf-start marks, and (chip regime) one per chunk of the root for its sums
    const int clear_wgs = static_cast<int>(std::min<int64_t>((n + 4096) / 4096, 512));
    const int sum_wgs = f.chip ? static_cast<int>((n + tb::kChunk - 1) / tb::kChunk) : 0;
    hipLaunchKernelGGL(tb::tb_init, dim3(1 + clear_wgs + sum_wgs), dim3(256), 0, s, f.P, clear_wgs);
  }
  f.chip_grid = static_cast<int>(std::min<int64_t>(n / tb::kChunk + tb::kMaxBig, (int64_t)ctx->n_cus * 4));
  // one wave per wave-regime node (at most n / 33 of them on a level) and four lanes per small node (most levels hold far
  // fewer than the n of them this bound allows for: the queues are walked with a stride)
  f.level_grid = static_cast<int>(std::max<int64_t>(1, std::min<int64_t>((int64_t)ctx->n_cus * 8, n / (4 * (tb::kSmallMax + 1)) + n / 256 + 1)));
  f.n_tiles = static_cast<int>((n + 1 + tb::kScanTile - 1) / tb::kScanTile);
  // a 120 k-point scan at b_max = 0.2 is 17 levels deep; deeper trees (dense maps, b_max -> 0) take the loop in the second half
  f.levels_done = 20;
  // (a STEP lags the level it finishes by up to kChipLevels - 1: a small node born while the chip regime runs waits for step
  // first_step, and its descendants follow one step per level — so the steps up to last_depth + kChipLevels can hold real
  // queues and keep their grid; and a build on another cloud size than the last one forgets the hint)
  const bool similar = fs.last_n > 0 && n >= fs.last_n - fs.last_n / 8 && n <= fs.last_n + fs.last_n / 8;
  // the previous scan of this sensor was last_depth levels deep: one spare step behind it instead of the fixed twenty (a step
  // that finds its queues empty is still a launch: ~5 us) — never fewer than the breadth-first layout of the top needs
  // (kTopParts), and a deeper tree takes the loop in the second half as before
  if (similar && fs.last_depth >= 0) f.levels_done = std::min(20, std::max(fs.last_depth + 2, kTopLevels + kTopLag + 1));
  f.quiet_from = (fs.last_depth >= 0 && similar) ? fs.last_depth + tb::kChipLevels : -1;
  f.use_need = similar && fs.have_need;
  RC_TRY(tb_run_levels(fs, 0, f.levels_done));
  HIP_TRY(hipGetLastError());
  RC_TRY(tb_summary_enqueue(fs, f.levels_done, false));
  // The emission, enqueued NOW — behind the summary, without the host's 10-11 us in between — into a block sized from the
  // previous scan's leaf count: consecutive scans of one sensor differ by a few per cent.  tb_emit then takes the counts from
  // the State on the device; a tree that does not fit writes nothing and tree_build_end_on emits again (the old sequence).
  f.pre = false;
  if (similar && fs.last_leaves > 0 && f.n_tiles <= tb::kScanDirectMax) {
    const int leaf_cap = static_cast<int>(std::min<int64_t>(n, (int64_t)fs.last_leaves + fs.last_leaves / 8 + 256));
    const size_t node_cap = 2 * (size_t)leaf_cap - 1, top_cap = (size_t)kTopMax - 1;
    DevTree t;
    void* blk = nullptr;
    RC_TRY(pool_alloc(ctx, layout_built_tree(t, nullptr, node_cap, (size_t)leaf_cap, top_cap), s, &blk));
    layout_built_tree(t, static_cast<char*>(blk), node_cap, (size_t)leaf_cap, top_cap);
    hipLaunchKernelGGL(tb::tb_emit, dim3(((int)node_cap + 255) / 256 + ((int)top_cap + 255) / 256), dim3(256), 0, s, f.P, (int)node_cap, t.nodes,
                       t.cnodes, t.leaves, (int)top_cap, t.top_dfs, t.top_link, t.top_exit, t.top, 0.0, 0.0, 0.0, leaf_cap);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
      pool_free(ctx, blk, nullptr);
      return fail(MADICP_ERR_DEVICE, std::string("tb_emit: ") + hipGetErrorString(e));
    }
    f.pre = true;
    f.pre_leaf_cap = leaf_cap;
    f.pre_steps = f.levels_done;
    f.pre_tree = t;
  }
  f.active = true;
  return MADICP_OK;
}

// the pre-sized emission of a construction that will not use it (too small, deeper tree, error, cancel): block back to the pool
void drop_pre_tree(madicp_ctx* ctx, FrontScratch::InFlight& f) {
  if (!f.pre) return;
  EventRef after;
  if (fence_event(ctx, &after) != MADICP_OK) after = nullptr;
  pool_free(ctx, f.pre_tree.block, after);
  f.pre = false;
  f.pre_tree = DevTree{};
}

// second half: the host learns the leaf count (deeper trees: more levels first), the tree is sized, emitted, compacted
int tree_build_end_on(madicp_ctx* ctx, FrontScratch& fs, int* out_tree_id, int32_t* out_n_leaves) {
  FrontScratch::InFlight& f = fs.fly;
  hipStream_t s = f.s;
  const tb::Params& P = f.P;
  tb::HostLine& hl = *fs.h_line;
  f.active = false;  // (whatever happens below, the scratch is free again: every error path leaves the stream drained or dead)
  auto pending = [&]() { return hl.pending_wave > 0 || hl.pending_quad > 0; };
  {
    int rc = tb_summary_wait(ctx, fs, f.levels_done);
    while (rc == MADICP_OK && hl.error == 0 && f.levels_done < tb::kMaxLevels && pending()) {
      const int to = std::min(f.levels_done + 8, tb::kMaxLevels);
      rc = tb_run_levels(fs, f.levels_done, to);
      f.levels_done = to;
      if (rc == MADICP_OK) rc = tb_summary_enqueue(fs, f.levels_done, true);
      if (rc == MADICP_OK) rc = tb_summary_wait(ctx, fs, f.levels_done);
    }
    if (rc != MADICP_OK) {
      drop_pre_tree(ctx, f);
      return rc;
    }
  }
  fs.state_stale = true;
  fs.last_depth = hl.error == 0 ? hl.max_level : -1;
  fs.last_n = f.n;
  fs.last_leaves = hl.error == 0 ? hl.n_leaves : 0;
  fs.have_need = hl.error == 0 && f.n_tiles <= tb::kScanDirectMax;
  if (fs.have_need) std::memcpy(fs.last_need, hl.need, sizeof(fs.last_need));
  const bool pre_ok = f.pre && f.pre_steps == f.levels_done && hl.error == 0 && !pending() && hl.n_leaves >= 1 &&
                      hl.n_leaves <= f.pre_leaf_cap;
  if (!pre_ok) drop_pre_tree(ctx, f);
  if (hl.error == 1) return fail(MADICP_ERR_DEVICE, "tree build: node capacity exceeded");
  if (hl.error == 2 || pending()) return fail(MADICP_ERR_INVALID, "tree build: tree deeper than the supported 96 levels");
  const int32_t n_leaves = hl.n_leaves, n_nodes = 2 * hl.n_leaves - 1;
  if (n_leaves < 1 || hl.n_nodes != n_nodes || hl.n_valid != n_nodes) {
    drop_pre_tree(ctx, f);
    return fail(MADICP_ERR_DEVICE, "tree build: inconsistent node count (" + std::to_string(hl.n_nodes) + " nodes, " +
                                       std::to_string(hl.n_valid) + " finished, " + std::to_string(n_leaves) + " leaves)");
  }
  struct { int32_t n_top; unsigned long long rho_bits; double origin[3]; } st{hl.n_top, hl.rho_bits, {hl.origin[0], hl.origin[1], hl.origin[2]}};
  DevTree t;
  if (pre_ok) {  // already emitted (tree_build_begin_on), into a block with head-room
    t = f.pre_tree;
    f.pre = false;
    f.pre_tree = DevTree{};
  }
  t.n_nodes = n_nodes;
  t.n_leaves = n_leaves;
  t.n_top = std::min<int32_t>(st.n_top, kTopMax - 1);
  double rho;
  std::memcpy(&rho, &st.rho_bits, sizeof(rho));
  t.rho2 = rho;
  const size_t nt = (size_t)t.n_top;
  if (pre_ok) {
    if (!nt) t.top_exit = nullptr, t.top_dfs = nullptr, t.top_link = nullptr, t.top = nullptr;
    set_desc(t, st.origin);
  } else {
    void* blk = nullptr;
    RC_TRY(pool_alloc(ctx, layout_built_tree(t, nullptr, (size_t)n_nodes, (size_t)n_leaves, nt), s, &blk));
    layout_built_tree(t, static_cast<char*>(blk), (size_t)n_nodes, (size_t)n_leaves, nt);
    set_desc(t, st.origin);
    // ONE launch: the DFS-preorder node array, the screening / dense leaf records and the staged top (tree_build.hip.h)
    hipLaunchKernelGGL(tb::tb_emit, dim3((n_nodes + 255) / 256 + (t.n_top + 255) / 256), dim3(256), 0, s, P, n_nodes, t.nodes, t.cnodes, t.leaves,
                       t.n_top, t.top_dfs, t.top_link, t.top_exit, t.top, st.origin[0], st.origin[1], st.origin[2], 0);
  }
  hipError_t e = hipGetLastError();
  int rc = MADICP_OK;
  if (e == hipSuccess && rc == MADICP_OK) e = hipEventCreateWithFlags(&t.ready, hipEventDisableTiming);
  if (e == hipSuccess && rc == MADICP_OK) e = hipEventRecord(t.ready, s);
  // a tree from the build stream: the copy stream feeds registrations from trees it believes it produced itself
  // (moving_from_leaves, transforms' staging), so it is put behind the event once, here
  if (e == hipSuccess && rc == MADICP_OK && s != ctx->copy) e = hipStreamWaitEvent(ctx->copy, t.ready, 0);
  if (e != hipSuccess || rc != MADICP_OK) {
    hipStreamSynchronize(s);
    release_tree(ctx, t, nullptr);
    return rc != MADICP_OK ? rc : fail(MADICP_ERR_DEVICE, std::string("tree build: ") + hipGetErrorString(e));
  }
  const int id = ctx->next_id++;
  ctx->trees[id] = t;
  *out_tree_id = id;
  if (out_n_leaves) *out_n_leaves = n_leaves;
  return MADICP_OK;
}

}  // namespace

extern "C" {

int madicp_tree_build(madicp_ctx* ctx, int cloud_id, double b_max, double b_min, int* out_tree_id, int32_t* out_n_leaves) {
  if (!ctx || !out_tree_id) return fail(MADICP_ERR_INVALID, "null argument");
  DevCloud* c = find_cloud(ctx, cloud_id);
  if (!c) return fail(MADICP_ERR_INVALID, "unknown cloud id");
  RC_TRY(busy_with_lookahead(ctx));
  HIP_TRY(hipSetDevice(ctx->device));
  FrontScratch* fs = nullptr;
  RC_TRY(ensure_scratch(ctx, c->n, &fs));
  fs->fly.lookahead = false;
  RC_TRY(tree_build_begin_on(ctx, *fs, c->xyz, c->n, b_max, b_min, ctx->copy));
  return tree_build_end_on(ctx, *fs, out_tree_id, out_n_leaves);
}

int madicp_tree_build_begin(madicp_ctx* ctx, const double* xyz, int64_t n, double b_max, double b_min) {
  if (!ctx || !xyz) return fail(MADICP_ERR_INVALID, "null argument");
  if (n < 1 || n > 0x3fffffff) return fail(MADICP_ERR_INVALID, "a cloud holds 1 .. 2^30 points");
  RC_TRY(busy_with_lookahead(ctx));
  HIP_TRY(hipSetDevice(ctx->device));
  {
    // The look-ahead pays when construction and registration run BACK TO BACK on the device (the runtime's default hardware
    // queue map: 0.78 -> 0.65 ms per frame) and not when they run side by side and slow each other down — which is what
    // GPU_MAX_HW_QUEUES > 4 gives (0.82 -> 0.82: DESIGN.md 9.4, bench.py's lookahead_in_a_plain_process).  A multi-rank process
    // wants the many queues for its collectives: said once, so that the combination is a decision and not an accident.
    static bool warned = false;
    const char* q = std::getenv("GPU_MAX_HW_QUEUES");
    if (!warned && q && std::atoi(q) > 4) {
      warned = true;
      std::fprintf(stderr, "madicp: a look-ahead tree build was begun with GPU_MAX_HW_QUEUES=%s: measured on MI355X, the look-ahead "
                           "shortens a frame under the runtime's default queue map (0.78 -> 0.65 ms) and does not with more than four "
                           "hardware queues (0.82 -> 0.82 ms); without prefetch() the frame is the same either way\n", q);
    }
  }
  if (!ctx->build) {
    // MADICP_BUILD_CUS=<n> (experiment): the look-ahead construction only gets the first n CUs, so that its level
    // kernels cannot spread over the CUs a registration round wants all of — whatever hardware queues the runtime maps the two
    // streams to
    // (a measurement aid: only in the measurement build, -DMADICP_MEASURE)
#ifdef MADICP_MEASURE
    const char* e = std::getenv("MADICP_BUILD_CUS");
    const int n_build = e ? std::atoi(e) : 0;
#else
    const int n_build = 0;
#endif
    if (n_build > 0 && n_build < ctx->n_cus) {
      const int words = (ctx->n_cus + 31) / 32;
      std::vector<uint32_t> mask((size_t)words, 0u);
      for (int cu = 0; cu < n_build; ++cu) mask[(size_t)cu / 32] |= 1u << (cu % 32);
      if (hipExtStreamCreateWithCUMask(&ctx->build, (uint32_t)words, mask.data()) != hipSuccess) {
        (void)hipGetLastError();
        ctx->build = nullptr;
      }
    }
    if (!ctx->build) HIP_TRY(hipStreamCreateWithFlags(&ctx->build, hipStreamNonBlocking));
  }
  FrontScratch* fs = nullptr;
  RC_TRY(ensure_scratch(ctx, n, &fs));
  hipStream_t s = ctx->build;
  // the scratch's previous users ran on the copy stream (ingest, deskew, a synchronous build's tail): behind them
  {
    EventRef ev;
    RC_TRY(fence_event(ctx, &ev));
    HIP_TRY(hipStreamWaitEvent(s, ev->ev_copy, 0));
  }
  // the scan: staged through the pinned buffers the uploads share, copied on the BUILD stream
  DevCloud c;
  {
    void* p = nullptr;
    RC_TRY(pool_alloc(ctx, sizeof(double) * 3 * (size_t)n, s, &p));
    c.xyz = static_cast<double*>(p);
    c.n = n;
  }
  {
    const int rc = stage_and_send_cloud(ctx, xyz, n, c.xyz, s);
    if (rc != MADICP_OK) return drop_cloud(ctx, c, rc);
  }
  // Option "build_after_registration" (experiment, default off): the construction's KERNELS wait for whatever the compute stream
  // holds right now — the registration this look-ahead was begun beside.  Built to test the hypothesis that the look-ahead
  // cliff (0.67 ms per frame in one process configuration, 1.5 in the others) is the two kernel sets fighting for the CUs;
  // measured: it is not — with the construction strictly behind the registration the other configurations stay at 1.5-2.1 ms
  // (profiles/r5_lookahead_matrix.md).
  if (ctx->build_after_registration) {
    if (!ctx->ev_build_gate) CLOUD_TRY(hipEventCreateWithFlags(&ctx->ev_build_gate, hipEventDisableTiming));
    CLOUD_TRY(hipEventRecord(ctx->ev_build_gate, ctx->stream));
    CLOUD_TRY(hipStreamWaitEvent(s, ctx->ev_build_gate, 0));
  }
  fs->fly.lookahead = true;
  const int rc = tree_build_begin_on(ctx, *fs, c.xyz, n, b_max, b_min, s);
  if (rc != MADICP_OK) {
    hipStreamSynchronize(s);
    return drop_cloud(ctx, c, rc);
  }
  fs->fly.cloud = c;
  return MADICP_OK;
}

int madicp_tree_build_end(madicp_ctx* ctx, int* out_tree_id, int32_t* out_n_leaves) {
  if (!ctx || !out_tree_id) return fail(MADICP_ERR_INVALID, "null argument");
  if (!ctx->front || !ctx->front->scratch.fly.active || !ctx->front->scratch.fly.lookahead)
    return fail(MADICP_ERR_INVALID, "no look-ahead tree build in flight");
  HIP_TRY(hipSetDevice(ctx->device));
  FrontScratch& fs = ctx->front->scratch;
  const int rc = tree_build_end_on(ctx, fs, out_tree_id, out_n_leaves);
  // the level kernels — the only readers of the scan — are behind the summary the host has just seen (or the stream is drained)
  DevCloud c = fs.fly.cloud;
  fs.fly.cloud = DevCloud{};
  if (rc != MADICP_OK) hipStreamSynchronize(fs.fly.s);
  return drop_cloud(ctx, c, rc);
}

int madicp_tree_build_cancel(madicp_ctx* ctx) {
  if (!ctx) return fail(MADICP_ERR_INVALID, "ctx is null");
  if (!ctx->front || !ctx->front->scratch.fly.active || !ctx->front->scratch.fly.lookahead) return MADICP_OK;
  HIP_TRY(hipSetDevice(ctx->device));
  FrontScratch& fs = ctx->front->scratch;
  const hipError_t e = hipStreamSynchronize(fs.fly.s);
  fs.fly.active = false;
  fs.state_stale = true;
  drop_pre_tree(ctx, fs.fly);
  DevCloud c = fs.fly.cloud;
  fs.fly.cloud = DevCloud{};
  drop_cloud(ctx, c, MADICP_OK);
  if (e != hipSuccess) return fail(MADICP_ERR_DEVICE, std::string("tree build: ") + hipGetErrorString(e));
  return MADICP_OK;
}

int madicp_tree_info(madicp_ctx* ctx, int tree_id, int32_t* out_n_nodes, int32_t* out_n_leaves) {
  if (!ctx) return fail(MADICP_ERR_INVALID, "ctx is null");
  auto it = ctx->trees.find(tree_id);
  if (it == ctx->trees.end()) return fail(MADICP_ERR_INVALID, "unknown tree id");
  if (out_n_nodes) *out_n_nodes = it->second.n_nodes;
  if (out_n_leaves) *out_n_leaves = it->second.n_leaves;
  return MADICP_OK;
}

// Diagnostics (tests): the points of the last synchronous build on this context in the order the construction left them —
// every leaf's members in the order the splits above it produced (reference: the caller's container after MADtree::build,
// mad_tree.cpp:95-97 with utils.h:37-52, except that the reference also overwrites a leaf's first member with its
// representative, mad_tree.cpp:76-84).  Valid until the next build, ingest or deskew on the context.
#ifdef MADICP_MEASURE
int madicp_debug_tree_build_points(madicp_ctx* ctx, double* out_xyz, int64_t n) {
  if (!ctx || !out_xyz) return fail(MADICP_ERR_INVALID, "null argument");
  if (!ctx->front || !ctx->front->scratch.block || !ctx->front->scratch.h_line) return fail(MADICP_ERR_INVALID, "no build yet");
  RC_TRY(busy_with_lookahead(ctx));
  FrontScratch& fs = ctx->front->scratch;
  const tb::Params& P = fs.fly.P;
  if (n != P.n_points) return fail(MADICP_ERR_INVALID, "n is not the size of the last build's cloud");
  HIP_TRY(hipSetDevice(ctx->device));
  void* d_out = nullptr;
  RC_TRY(pool_alloc(ctx, sizeof(double) * 3 * (size_t)n, ctx->copy, &d_out));
  const int n_nodes = fs.h_line->n_nodes;
  hipLaunchKernelGGL(tb::tb_debug_order, dim3((n_nodes + 3) / 4), dim3(256), 0, ctx->copy, P, n_nodes, static_cast<double*>(d_out));
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipMemcpyAsync(out_xyz, d_out, sizeof(double) * 3 * (size_t)n, hipMemcpyDeviceToHost, ctx->copy);
  const hipError_t e2 = hipStreamSynchronize(ctx->copy);
  pool_free(ctx, d_out, nullptr);
  if (e != hipSuccess || e2 != hipSuccess)
    return fail(MADICP_ERR_DEVICE, std::string("tree build points: ") + hipGetErrorString(e != hipSuccess ? e : e2));
  return MADICP_OK;
}
#endif  // MADICP_MEASURE

// per-level node counts of the last build on this context (diagnostics for tests / tools): out[0] = levels reached,
// out[1] = lane-regime sub-trees, then 2 x 64 ints: wave-regime and chip-regime nodes per level
int madicp_tree_build_stats(madicp_ctx* ctx, int32_t out[130]) {
  if (!ctx || !out) return fail(MADICP_ERR_INVALID, "null argument");
  if (!ctx->front || !ctx->front->scratch.h_state || !ctx->front->scratch.block) return fail(MADICP_ERR_INVALID, "no build yet");
  RC_TRY(busy_with_lookahead(ctx));
  FrontScratch& fs = ctx->front->scratch;
  if (fs.state_stale) {  // (a build publishes one line to the host; the per-level counters are fetched when asked for)
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipMemcpyAsync(fs.h_state, fs.P.st, sizeof(tb::State), hipMemcpyDeviceToHost, ctx->copy));
    HIP_TRY(hipStreamSynchronize(ctx->copy));
    fs.state_stale = false;
  }
  const tb::State& st = *fs.h_state;
  out[0] = st.max_level;
  int lanes = 0;
  for (int i = 0; i <= tb::kMaxLevels; ++i) lanes += st.small_count[i].v;
  out[1] = lanes;
  for (int i = 0; i < 64; ++i) {
    out[2 + i] = st.q_count[i].v;
    out[66 + i] = st.big_count[i].v;
  }
#ifdef MADICP_TB_CHECK
  out[129] = st.n_nodes.pad_[0];  // development: mismatches counted by the in-kernel self-check
#endif
  return MADICP_OK;
}

}  // extern "C"

#ifdef MADICP_TB_STAMPS
// development only (tools/tb_stamps.py): reset / fetch the per-wave wall-clock stamps of tb_level
extern "C" int madicp_debug_tb_stamps(madicp_ctx* ctx, unsigned long long* out, int reset) {
  if (!ctx) return MADICP_ERR_INVALID;
  hipStreamSynchronize(ctx->copy);
  const size_t bytes = sizeof(unsigned long long) * 24 * 512 * 4 * 16;
  if (reset) {
    std::vector<unsigned long long> init(24 * 512 * 4 * 16, 0ull);
    return hipMemcpyToSymbol(HIP_SYMBOL(madicp::tb::g_tb_stamps), init.data(), bytes) == hipSuccess ? 0 : MADICP_ERR_DEVICE;
  }
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(madicp::tb::g_tb_stamps), bytes) == hipSuccess ? 0 : MADICP_ERR_DEVICE;
}
#endif
