// libmadicp_hip.so — implementation of the C ABI declared in include/madicp_hip.h.
// Context / buffer management, launch sequencing (eager or captured hipGraph), optional RCCL all-reduce.
#include "kernels.hip.h"

#include <hip/hip_ext.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <tuple>
#include <unordered_map>
#include <vector>

using namespace madicp;

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}

#define HIP_TRY(expr)                                                                             \
  do {                                                                                            \
    hipError_t e_ = (expr);                                                                       \
    if (e_ != hipSuccess)                                                                         \
      return fail(MADICP_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));         \
  } while (0)

#define NCCL_TRY(expr)                                                                            \
  do {                                                                                            \
    ncclResult_t r_ = (expr);                                                                     \
    if (r_ != ncclSuccess)                                                                        \
      return fail(MADICP_ERR_COMM, std::string(#expr) + ": " + ncclGetErrorString(r_));          \
  } while (0)

struct DevTree {
  madicp_node* nodes = nullptr;
  CNode* cnodes = nullptr;    // 16-byte screening records, same indexing
  LeafRec* leaves = nullptr;  // dense 64-byte leaf records, by leaf ordinal
  CNode* top = nullptr;       // top levels, breadth first (staged into LDS by icp_round)
  int2* top_exit = nullptr;
  int* top_dfs = nullptr;
  unsigned int* top_link = nullptr;
  int32_t n_top = 0;
  TreeMeta* meta = nullptr;   // device-side scratch tree_compact writes origin / radius into
  TreeDesc desc{};            // what the kernels get by value
  int32_t n_nodes = 0, n_leaves = 0;
};
struct DevMoving {
  double* xyzn = nullptr;  // (L,4)
  uint8_t* matched = nullptr;
  int32_t L = 0;
  // correspondence cache of the registration in flight for this scan: (K,L) each, grown on demand
  uint32_t* cache_leaf = nullptr;
  float* cache_margin = nullptr;
  int32_t cache_K = 0;
};

struct GraphKey {  // everything a captured launch sequence bakes in
  int grid, batch, iters, qpt, comm, lds, K, rpt, trace;
  bool operator<(const GraphKey& o) const {
    return std::tie(grid, batch, iters, qpt, comm, lds, K, rpt, trace) <
           std::tie(o.grid, o.batch, o.iters, o.qpt, o.comm, o.lds, o.K, o.rpt, o.trace);
  }
};
struct Geometry {
  int grid;             // workgroups per scan (multiple of 8)
  int ranges_per_tree;  // units per tree
  int qpt;              // leaves a lane walks at once: 1, 2 or 4
  int lds_bytes;        // dynamic LDS of the launch: kTopLdsBytes when units are big enough to stage a tree's top, else 0
};

}  // namespace

struct madicp_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  int n_cus = 256;

  std::unordered_map<int, DevTree> trees;
  std::unordered_map<int, DevMoving> movings;
  int next_id = 1;

  // registration state
  Job* d_jobs = nullptr;       // [MADICP_MAX_BATCH]
  // pinned staging ring: the host may run kStageSlots-1 batches ahead of the device
  static constexpr int kStageSlots = 4;
  Job* h_stage[kStageSlots] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t stage_ev[kStageSlots] = {nullptr, nullptr, nullptr, nullptr};
  int stage_next = 0;
  Job* h_fetch = nullptr;      // pinned read-back block
  double* d_partials = nullptr;
  size_t partials_cap = 0;     // doubles
  int partials_grid = -1, partials_batch = -1;  // geometry the zero padding rows of d_partials are valid for
  double* d_totals = nullptr;  // [MADICP_MAX_BATCH][kAcc]
  double* d_scratch = nullptr; // 12 doubles (R,t for tree_transform)
  int last_batch = 0;
  std::vector<int> last_moving;

  // options
  int blocks_per_cu = 1;  // icp_round workgroups (768 threads) per CU
  int use_graph = 1;
  int qpt_override = 0;
  int cache_corr = 1;  // reuse correspondences across GN rounds when provably unchanged
  int stage_min_leaves = 1024;  // LDS staging threshold (leaves per unit); 0 = always, huge = never (measured break-even ~1000)

  std::map<GraphKey, hipGraphExec_t> graphs;

  hipEvent_t ev_t0 = nullptr, ev_t1 = nullptr;  // madicp_icp_time_linearize

  // multi-GPU
  ncclComm_t comm = nullptr;
  int n_ranks = 1, rank = 0;
};

namespace {

int ensure_partials(madicp_ctx* ctx, size_t doubles) {
  if (doubles <= ctx->partials_cap) return MADICP_OK;
  if (ctx->d_partials) HIP_TRY(hipFree(ctx->d_partials));
  ctx->d_partials = nullptr;
  ctx->partials_cap = 0;
  HIP_TRY(hipMalloc(&ctx->d_partials, doubles * sizeof(double)));
  ctx->partials_cap = doubles;
  ctx->partials_grid = ctx->partials_batch = -1;
  // cached graphs hold the old pointer
  for (auto& g : ctx->graphs) hipGraphExecDestroy(g.second);
  ctx->graphs.clear();
  return MADICP_OK;
}

// launch geometry: 8 XCDs x slots workgroups per scan, about blocks_per_cu * n_cus in total over the batch, and
// one (tree, range) unit per workgroup so that every workgroup gets the same number of leaves
Geometry pick_geometry(const madicp_ctx* ctx, int max_L, int K, int batch) {
  Geometry g;
  // One 768-thread workgroup per CU (3 waves per SIMD is what the kernel's registers allow), one (tree, range) unit
  // per workgroup; a batch shares the CUs between its scans.  One leaf per lane and pass by default (two leaves per
  // lane share their loads but were not measured faster).
  g.qpt = ctx->qpt_override == 2 ? 2 : 1;
  long long grid = std::max<long long>(8, (long long)ctx->blocks_per_cu * ctx->n_cus / std::max(1, batch));
  const long long max_useful = (long long)K * ((max_L + 63) / 64);  // never below one wave of leaves per unit
  grid = std::max<long long>(8, std::min(grid, max_useful) / 8 * 8);
  g.grid = static_cast<int>(grid);
  g.ranges_per_tree = static_cast<int>(std::max<long long>(1, grid / K));
  const int per_range = (max_L + g.ranges_per_tree - 1) / g.ranges_per_tree;
  g.lds_bytes = per_range >= ctx->stage_min_leaves ? kTopLdsBytes : 0;
  return g;
}

struct Launch {  // one registration's launch shape
  int grid, batch, iters, qpt, lds, K, rpt, trace;
};

void launch_round(madicp_ctx* ctx, const Launch& l, int round, const double* totals) {
  dim3 g(l.grid, l.batch), b(kBlock);
  void (*kern)(const Job*, Job*, double*, const double*, int, int, int, int) =
      l.trace ? (l.qpt == 2 ? icp_round<2, true> : icp_round<1, true>) : (l.qpt == 2 ? icp_round<2, false> : icp_round<1, false>);
  hipLaunchKernelGGL(kern, g, b, l.lds, ctx->stream, (const Job*)ctx->d_jobs, ctx->d_jobs, ctx->d_partials, totals, round,
                     l.iters, l.K, l.rpt);
}

// the launch sequence of one (batched) registration; valid both eagerly and under stream capture
int enqueue_rounds(madicp_ctx* ctx, const Launch& l) {
  const int grid = l.grid, batch = l.batch, iters = l.iters;
  for (int it = 0; it < iters; ++it) {
    launch_round(ctx, l, it, (ctx->comm && it > 0) ? ctx->d_totals : nullptr);
    if (ctx->comm) {
      hipLaunchKernelGGL(icp_reduce, dim3(batch), dim3(kBlock), 0, ctx->stream, ctx->d_partials, grid, batch, it,
                         ctx->d_totals);
      NCCL_TRY(ncclAllReduce(ctx->d_totals, ctx->d_totals, (size_t)batch * kAcc, ncclDouble, ncclSum, ctx->comm,
                             ctx->stream));
    }
  }
  if (ctx->comm) {
    // a leaf is an inlier if ANY keyframe on ANY rank matched it (mad_icp.cpp:85, pipeline.cpp:197-204)
    for (int s = 0; s < batch; ++s) {
      const DevMoving& mv = ctx->movings.at(ctx->last_moving[s]);
      NCCL_TRY(ncclAllReduce(mv.matched, mv.matched, (size_t)mv.L, ncclUint8, ncclMax, ctx->comm, ctx->stream));
    }
  }
  // (icp_reduce / icp_final join with kBlock threads, like icp_round: same summation order with and without ranks)
  hipLaunchKernelGGL(icp_final, dim3(batch), dim3(kBlock), 0, ctx->stream, ctx->d_jobs, ctx->d_partials,
                     ctx->comm ? ctx->d_totals : nullptr, grid, batch);
  HIP_TRY(hipGetLastError());
  return MADICP_OK;
}

int run_rounds(madicp_ctx* ctx, const Launch& l) {
  // graphs: (conservatively) only without a communicator
  const bool graph_ok = ctx->use_graph && !ctx->comm;
  if (!graph_ok) return enqueue_rounds(ctx, l);
  const GraphKey key{l.grid, l.batch, l.iters, l.qpt, ctx->comm ? 1 : 0, l.lds, l.K, l.rpt, l.trace};
  auto it = ctx->graphs.find(key);
  if (it == ctx->graphs.end()) {
    hipGraph_t graph = nullptr;
    HIP_TRY(hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
    const int rc = enqueue_rounds(ctx, l);
    hipError_t e = hipStreamEndCapture(ctx->stream, &graph);
    if (rc != MADICP_OK) return rc;
    if (e != hipSuccess) return fail(MADICP_ERR_DEVICE, std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
    hipGraphExec_t exec = nullptr;
    HIP_TRY(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    hipGraphDestroy(graph);
    it = ctx->graphs.emplace(key, exec).first;
  }
  HIP_TRY(hipGraphLaunch(it->second, ctx->stream));
  return MADICP_OK;
}

struct RegArgs {
  int n_scans;
  const int* moving_ids;
  const int* tree_ids;
  int K;
  const double* X0;
  const madicp_icp_params* params;
  int n_iters;
  int flags;
  uint32_t* d_corr;     // single-scan debug trace (device) or null
  double* d_x_iters;    // single-scan (device) or null
  int time_launches = 0;    // > 0: measurement mode, see madicp_icp_time_linearize
  double* out_avg_us = nullptr;
};

int enqueue_registration(madicp_ctx* ctx, const RegArgs& a) {
  if (!ctx || !a.moving_ids || !a.tree_ids || !a.X0 || !a.params) return fail(MADICP_ERR_INVALID, "null argument");
  if (a.n_scans < 1 || a.n_scans > MADICP_MAX_BATCH) return fail(MADICP_ERR_CAPACITY, "n_scans out of range");
  if (a.K < 1) return fail(MADICP_ERR_INVALID, "K must be >= 1");
  if (a.K > MADICP_MAX_TREES) return fail(MADICP_ERR_CAPACITY, "K exceeds MADICP_MAX_TREES");
  if (a.n_iters < 1) return fail(MADICP_ERR_INVALID, "n_iters must be >= 1");
  HIP_TRY(hipSetDevice(ctx->device));
  // next slot of the pinned staging ring; wait until the H2D copy that last used it has executed
  const int slot = ctx->stage_next;
  ctx->stage_next = (slot + 1) % madicp_ctx::kStageSlots;
  HIP_TRY(hipEventSynchronize(ctx->stage_ev[slot]));
  Job* h_jobs = ctx->h_stage[slot];

  int max_L = 0;
  ctx->last_moving.assign(a.moving_ids, a.moving_ids + a.n_scans);
  for (int s = 0; s < a.n_scans; ++s) {
    auto mit = ctx->movings.find(a.moving_ids[s]);
    if (mit == ctx->movings.end()) return fail(MADICP_ERR_INVALID, "unknown moving id");
    DevMoving& mv = mit->second;
    if (a.n_iters > 1 && !a.time_launches && mv.cache_K < a.K) {  // (re)size this scan's correspondence cache
      HIP_TRY(hipStreamSynchronize(ctx->stream));
      hipFree(mv.cache_leaf);
      hipFree(mv.cache_margin);
      mv.cache_leaf = nullptr;
      mv.cache_margin = nullptr;
      mv.cache_K = 0;
      HIP_TRY(hipMalloc(&mv.cache_leaf, sizeof(uint32_t) * (size_t)a.K * mv.L));
      HIP_TRY(hipMalloc(&mv.cache_margin, sizeof(float) * (size_t)a.K * mv.L));
      mv.cache_K = a.K;
    }
    Job& j = h_jobs[s];
    std::memset(&j, 0, offsetof(Job, trees));
    j.moving = mv.xyzn;
    j.matched = mv.matched;
    j.cache_leaf = (a.n_iters > 1 && !a.time_launches) ? mv.cache_leaf : nullptr;
    j.cache_margin = j.cache_leaf ? mv.cache_margin : nullptr;
    j.corr = (s == 0) ? a.d_corr : nullptr;
    j.x_iters = (s == 0) ? a.d_x_iters : nullptr;
    j.L = mv.L;
    j.K = a.K;
    j.n_iters = a.n_iters;
    j.iter = 0;
    j.flags = a.flags | (ctx->cache_corr ? 0 : kFlagNoReuse);
    std::memcpy(j.X, a.X0 + 12 * s, 12 * sizeof(double));
    std::memcpy(j.Xring[0], a.X0 + 12 * s, 12 * sizeof(double));
    std::memcpy(j.Xring[1], a.X0 + 12 * s, 12 * sizeof(double));
    j.min_ball = a.params->min_ball;
    j.rho = std::sqrt(a.params->rho_ker);  // MADicp ctor, mad_icp.cpp:32
    j.b_ratio = a.params->b_ratio;
    for (int k = 0; k < a.K; ++k) {
      auto tit = ctx->trees.find(a.tree_ids[k]);
      if (tit == ctx->trees.end()) return fail(MADICP_ERR_INVALID, "unknown tree id");
      j.trees[k] = tit->second.desc;
    }
    max_L = std::max(max_L, mv.L);
    // flags are cleared on the device before the last round; with a single round that is "now"
    if (a.n_iters == 1) HIP_TRY(hipMemsetAsync(mv.matched, 0, (size_t)mv.L, ctx->stream));
  }
  const Geometry geo = pick_geometry(ctx, max_L, a.K, a.n_scans);
  const int grid = geo.grid;
  const Launch launch{grid, a.n_scans, a.n_iters, geo.qpt, geo.lds_bytes, a.K, geo.ranges_per_tree, a.d_corr ? 1 : 0};
  for (int s = 0; s < a.n_scans; ++s) {
    h_jobs[s].ranges_per_tree = geo.ranges_per_tree;
    h_jobs[s].stage_min_leaves = ctx->stage_min_leaves;
    h_jobs[s].lds_top = geo.lds_bytes ? 1 : 0;
  }
  // two round parities of partials — join_rows(grid) rows per scan, the rows beyond `grid` are zero and stay zero —
  // then two parities of per-workgroup walk hints, + one padding row (the join's 16-byte loads read one double past)
  const size_t prows = (size_t)madicp::join_rows(grid);
  const size_t partial_doubles = (size_t)2 * a.n_scans * prows * kAcc + (size_t)2 * a.n_scans * grid + kAcc;
  const int rc0 = ensure_partials(ctx, partial_doubles);
  if (rc0 != MADICP_OK) return rc0;
  if (ctx->partials_grid != grid || ctx->partials_batch != a.n_scans) {
    // a row that is padding in this geometry may have been a real row in the previous one
    HIP_TRY(hipMemsetAsync(ctx->d_partials, 0, partial_doubles * sizeof(double), ctx->stream));
    ctx->partials_grid = grid;
    ctx->partials_batch = a.n_scans;
  }
  const size_t job_bytes = offsetof(Job, trees) + sizeof(TreeDesc) * (size_t)std::max(1, a.K);
  for (int s = 0; s < a.n_scans; ++s)
    HIP_TRY(hipMemcpyAsync(ctx->d_jobs + s, h_jobs + s, job_bytes, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(hipEventRecord(ctx->stage_ev[slot], ctx->stream));
  ctx->last_batch = a.n_scans;
  if (a.time_launches > 0) {
    // n back-to-back launches of the dominant kernel inside ONE captured graph, bracketed by two events: the
    // per-launch time is defined exactly like a profiler trace of the registration graph defines it
    if (!ctx->ev_t0) {
      HIP_TRY(hipEventCreate(&ctx->ev_t0));
      HIP_TRY(hipEventCreate(&ctx->ev_t1));
    }
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    HIP_TRY(hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < a.time_launches; ++i) launch_round(ctx, launch, 0, nullptr);
    HIP_TRY(hipStreamEndCapture(ctx->stream, &graph));
    HIP_TRY(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    HIP_TRY(hipGraphLaunch(exec, ctx->stream));  // warm-up replay
    HIP_TRY(hipEventRecord(ctx->ev_t0, ctx->stream));
    HIP_TRY(hipGraphLaunch(exec, ctx->stream));
    HIP_TRY(hipEventRecord(ctx->ev_t1, ctx->stream));
    // fold the last launch's partials (round parity 0) into job->visits / H / b: icp_final with n_iters = 1 semantics
    hipLaunchKernelGGL(icp_final, dim3(a.n_scans), dim3(kBlock), 0, ctx->stream, ctx->d_jobs, ctx->d_partials,
                       (const double*)nullptr, grid, a.n_scans);
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, ctx->ev_t0, ctx->ev_t1));
    hipGraphExecDestroy(exec);
    hipGraphDestroy(graph);
    if (a.out_avg_us) *a.out_avg_us = 1e3 * ms / a.time_launches;
    return MADICP_OK;
  }
  return run_rounds(ctx, launch);
}

}  // namespace

extern "C" {

const char* madicp_last_error(void) { return g_err.c_str(); }
int madicp_abi_version(void) { return 1; }

int madicp_ctx_create(int device_id, void* stream, madicp_ctx** out) {
  if (!out) return fail(MADICP_ERR_INVALID, "out is null");
  *out = nullptr;
  int n_dev = 0;
  HIP_TRY(hipGetDeviceCount(&n_dev));
  if (device_id < 0 || device_id >= n_dev) return fail(MADICP_ERR_DEVICE, "no such HIP device");
  HIP_TRY(hipSetDevice(device_id));
  madicp_ctx* ctx = new madicp_ctx;
  ctx->device = device_id;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device_id) == hipSuccess) ctx->n_cus = prop.multiProcessorCount;
  if (stream) {
    ctx->stream = static_cast<hipStream_t>(stream);
  } else {
    hipError_t e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
      delete ctx;
      return fail(MADICP_ERR_DEVICE, std::string("hipStreamCreate: ") + hipGetErrorString(e));
    }
    ctx->own_stream = true;
  }
  hipError_t e = hipMalloc(&ctx->d_jobs, sizeof(Job) * MADICP_MAX_BATCH);
  for (int i = 0; i < madicp_ctx::kStageSlots && e == hipSuccess; ++i) {
    e = hipHostMalloc(&ctx->h_stage[i], sizeof(Job) * MADICP_MAX_BATCH, hipHostMallocDefault);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->stage_ev[i], hipEventDisableTiming);
  }
  if (e == hipSuccess) e = hipHostMalloc(&ctx->h_fetch, sizeof(Job) * MADICP_MAX_BATCH, hipHostMallocDefault);
  if (e == hipSuccess) e = hipMalloc(&ctx->d_totals, sizeof(double) * kAcc * MADICP_MAX_BATCH);
  if (e == hipSuccess) e = hipMalloc(&ctx->d_scratch, sizeof(double) * 12);
  if (e != hipSuccess) {
    madicp_ctx_destroy(ctx);
    return fail(MADICP_ERR_DEVICE, std::string("context allocation: ") + hipGetErrorString(e));
  }
  *out = ctx;
  return MADICP_OK;
}

int madicp_ctx_destroy(madicp_ctx* ctx) {
  if (!ctx) return MADICP_OK;
  hipSetDevice(ctx->device);
  if (ctx->stream) hipStreamSynchronize(ctx->stream);
  if (ctx->comm) ncclCommDestroy(ctx->comm);
  for (auto& g : ctx->graphs) hipGraphExecDestroy(g.second);
  for (auto& t : ctx->trees) {
    hipFree(t.second.nodes);
    hipFree(t.second.cnodes);
    hipFree(t.second.leaves);
    hipFree(t.second.top);
    hipFree(t.second.top_exit);
    hipFree(t.second.top_dfs);
    hipFree(t.second.top_link);
    hipFree(t.second.meta);
  }
  for (auto& m : ctx->movings) {
    hipFree(m.second.xyzn);
    hipFree(m.second.matched);
    hipFree(m.second.cache_leaf);
    hipFree(m.second.cache_margin);
  }
  if (ctx->ev_t0) hipEventDestroy(ctx->ev_t0);
  if (ctx->ev_t1) hipEventDestroy(ctx->ev_t1);
  if (ctx->d_jobs) hipFree(ctx->d_jobs);
  for (int i = 0; i < madicp_ctx::kStageSlots; ++i) {
    if (ctx->h_stage[i]) hipHostFree(ctx->h_stage[i]);
    if (ctx->stage_ev[i]) hipEventDestroy(ctx->stage_ev[i]);
  }
  if (ctx->h_fetch) hipHostFree(ctx->h_fetch);
  if (ctx->d_partials) hipFree(ctx->d_partials);
  if (ctx->d_totals) hipFree(ctx->d_totals);
  if (ctx->d_scratch) hipFree(ctx->d_scratch);
  if (ctx->own_stream && ctx->stream) hipStreamDestroy(ctx->stream);
  delete ctx;
  return MADICP_OK;
}

int madicp_ctx_synchronize(madicp_ctx* ctx) {
  if (!ctx) return fail(MADICP_ERR_INVALID, "ctx is null");
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return MADICP_OK;
}

int madicp_ctx_set_option(madicp_ctx* ctx, const char* key, int64_t value) {
  if (!ctx || !key) return fail(MADICP_ERR_INVALID, "null argument");
  const std::string k(key);
  if (k == "grid_blocks_per_cu") {
    if (value < 1 || value > 4) return fail(MADICP_ERR_INVALID, "grid_blocks_per_cu must be in 1..4");
    ctx->blocks_per_cu = (int)value;
  } else if (k == "use_graph") {
    ctx->use_graph = value ? 1 : 0;
  } else if (k == "cache_correspondences") {
    ctx->cache_corr = value ? 1 : 0;
  } else if (k == "lds_stage_min_leaves") {
    if (value < 0) return fail(MADICP_ERR_INVALID, "lds_stage_min_leaves must be >= 0");
    ctx->stage_min_leaves = (int)std::min<int64_t>(value, 1 << 30);
  } else if (k == "queries_per_lane") {
    if (value != 0 && value != 1 && value != 2) return fail(MADICP_ERR_INVALID, "queries_per_lane must be 0 (default), 1 or 2");
    ctx->qpt_override = (int)value;
  } else {
    return fail(MADICP_ERR_INVALID, "unknown option: " + k);
  }
  return MADICP_OK;
}

// ---- trees ------------------------------------------------------------------------------------------
namespace {
// (re)build the 16-byte screening records and the tree's origin / radius on the device
int compact_tree(madicp_ctx* ctx, DevTree& t) {
  HIP_TRY(hipMemsetAsync(&t.meta->rho2_bits, 0, sizeof(unsigned long long), ctx->stream));
  hipLaunchKernelGGL(tree_compact, dim3((t.n_nodes + 255) / 256), dim3(256), 0, ctx->stream, t.meta, t.cnodes, t.leaves, t.n_nodes);
  if (t.n_top > 0)
    hipLaunchKernelGGL(tree_compact_top, dim3((t.n_top + 255) / 256), dim3(256), 0, ctx->stream, t.nodes, t.top, t.top_dfs,
                       t.top_link, t.n_top);
  HIP_TRY(hipGetLastError());
  TreeMeta hm;
  HIP_TRY(hipMemcpyAsync(&hm, t.meta, sizeof(hm), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  t.desc.nodes = t.nodes;
  t.desc.cnodes = t.cnodes;
  t.desc.leaves = t.leaves;
  t.desc.top = t.top;
  t.desc.top_exit = t.top_exit;
  t.desc.top_dfs = t.top_dfs;
  t.desc.n_top = t.n_top;
  std::memcpy(t.desc.origin, hm.origin, sizeof(hm.origin));
  double rho2;
  std::memcpy(&rho2, &hm.rho2_bits, sizeof(double));
  t.desc.rho = 1.7320508075688774 * rho2 * (1.0 + 1e-12);  // |v|_1 <= sqrt(3) |v|_2, rounded up
  return MADICP_OK;
}

// Breadth-first layout of the internal nodes of the first kTopLevels levels (structure only; depends on the tree's
// topology, so it is made once at upload).  See "LDS-staged top levels" in kernels.hip.h for the link word.
void layout_top(const madicp_node* nodes, std::vector<int>& dfs, std::vector<unsigned int>& link, std::vector<int2>& exits) {
  dfs.clear();
  link.clear();
  exits.clear();
  if (nodes[0].right == 0) return;
  std::vector<int> level{0};
  dfs.push_back(0);
  for (size_t e = 0; e < dfs.size(); ++e) {
    const int i = dfs[e];
    const int l = i + 1, r = i + nodes[i].right;
    const bool l_leaf = nodes[l].right == 0, r_leaf = nodes[r].right == 0;
    const bool deeper = level[e] + 1 < kTopLevels;
    const bool l_in = !l_leaf && deeper && dfs.size() + 1 <= (size_t)kTopMax - 1;
    const bool r_in = !r_leaf && deeper && dfs.size() + (l_in ? 2 : 1) <= (size_t)kTopMax - 1;
    unsigned int w = (unsigned int)dfs.size() & kTopFirst;
    if (l_in) { dfs.push_back(l); level.push_back(level[e] + 1); w |= kTopLeftIn; }
    if (r_in) { dfs.push_back(r); level.push_back(level[e] + 1); w |= kTopRightIn; }
    if (l_leaf) w |= kTopLeftLeaf;
    if (r_leaf) w |= kTopRightLeaf;
    link.push_back(w);
    exits.push_back(make_int2(l, r));
  }
}

void free_tree(DevTree& t) {
  hipFree(t.nodes);
  hipFree(t.cnodes);
  hipFree(t.leaves);
  hipFree(t.top);
  hipFree(t.top_exit);
  hipFree(t.top_dfs);
  hipFree(t.top_link);
  hipFree(t.meta);
  t = DevTree{};
}
}  // namespace

int madicp_tree_upload(madicp_ctx* ctx, const madicp_node* nodes, int32_t n_nodes, int32_t n_leaves, int* out_tree_id) {
  if (!ctx || !nodes || !out_tree_id) return fail(MADICP_ERR_INVALID, "null argument");
  if (n_nodes < 1 || n_leaves < 1 || n_nodes != 2 * n_leaves - 1)
    return fail(MADICP_ERR_INVALID, "a MAD-tree has n_nodes == 2*n_leaves-1 >= 1");
  HIP_TRY(hipSetDevice(ctx->device));
  DevTree t;
  t.n_nodes = n_nodes;
  t.n_leaves = n_leaves;
  hipError_t e = hipMalloc(&t.nodes, sizeof(madicp_node) * (size_t)n_nodes);
  if (e == hipSuccess) e = hipMalloc(&t.cnodes, sizeof(CNode) * (size_t)n_nodes);
  if (e == hipSuccess) e = hipMalloc(&t.leaves, sizeof(LeafRec) * (size_t)n_leaves);
  if (e == hipSuccess) e = hipMalloc(&t.meta, sizeof(TreeMeta));
  TreeMeta hm{};
  hm.nodes = t.nodes;
  hm.cnodes = t.cnodes;
  hm.n_nodes = n_nodes;
  hm.n_leaves = n_leaves;
  if (e == hipSuccess) e = hipMemcpyAsync(t.meta, &hm, sizeof(hm), hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess)
    e = hipMemcpyAsync(t.nodes, nodes, sizeof(madicp_node) * (size_t)n_nodes, hipMemcpyHostToDevice, ctx->stream);
  std::vector<int> top_dfs;
  std::vector<unsigned int> top_link;
  std::vector<int2> top_exit;
  layout_top(nodes, top_dfs, top_link, top_exit);
  t.n_top = static_cast<int32_t>(top_dfs.size());
  if (t.n_top > 0) {
    if (e == hipSuccess) e = hipMalloc(&t.top, sizeof(CNode) * top_dfs.size());
    if (e == hipSuccess) e = hipMalloc(&t.top_exit, sizeof(int2) * top_dfs.size());
    if (e == hipSuccess) e = hipMalloc(&t.top_dfs, sizeof(int) * top_dfs.size());
    if (e == hipSuccess) e = hipMalloc(&t.top_link, sizeof(unsigned int) * top_dfs.size());
    if (e == hipSuccess) e = hipMemcpyAsync(t.top_exit, top_exit.data(), sizeof(int2) * top_dfs.size(), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(t.top_dfs, top_dfs.data(), sizeof(int) * top_dfs.size(), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(t.top_link, top_link.data(), sizeof(unsigned int) * top_dfs.size(), hipMemcpyHostToDevice, ctx->stream);
  }
  int rc = MADICP_OK;
  if (e == hipSuccess) rc = compact_tree(ctx, t);
  if (e == hipSuccess && rc == MADICP_OK) e = hipStreamSynchronize(ctx->stream);  // hm / nodes are caller memory
  if (e != hipSuccess || rc != MADICP_OK) {
    free_tree(t);
    return rc != MADICP_OK ? rc : fail(MADICP_ERR_DEVICE, std::string("tree upload: ") + hipGetErrorString(e));
  }
  const int id = ctx->next_id++;
  ctx->trees[id] = t;
  *out_tree_id = id;
  return MADICP_OK;
}

int madicp_tree_release(madicp_ctx* ctx, int tree_id) {
  if (!ctx) return fail(MADICP_ERR_INVALID, "ctx is null");
  auto it = ctx->trees.find(tree_id);
  if (it == ctx->trees.end()) return fail(MADICP_ERR_INVALID, "unknown tree id");
  HIP_TRY(hipSetDevice(ctx->device));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  free_tree(it->second);
  ctx->trees.erase(it);
  return MADICP_OK;
}

int madicp_tree_download(madicp_ctx* ctx, int tree_id, madicp_node* out_nodes, int32_t n_nodes) {
  if (!ctx || !out_nodes) return fail(MADICP_ERR_INVALID, "null argument");
  auto it = ctx->trees.find(tree_id);
  if (it == ctx->trees.end()) return fail(MADICP_ERR_INVALID, "unknown tree id");
  if (n_nodes != it->second.n_nodes) return fail(MADICP_ERR_INVALID, "n_nodes mismatch");
  HIP_TRY(hipSetDevice(ctx->device));
  HIP_TRY(hipMemcpyAsync(out_nodes, it->second.nodes, sizeof(madicp_node) * (size_t)n_nodes, hipMemcpyDeviceToHost,
                         ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return MADICP_OK;
}

int madicp_tree_transform(madicp_ctx* ctx, int tree_id, const double R[9], const double t[3]) {
  if (!ctx || !R || !t) return fail(MADICP_ERR_INVALID, "null argument");
  auto it = ctx->trees.find(tree_id);
  if (it == ctx->trees.end()) return fail(MADICP_ERR_INVALID, "unknown tree id");
  HIP_TRY(hipSetDevice(ctx->device));
  double Rt[12];
  std::memcpy(Rt, R, 9 * sizeof(double));
  std::memcpy(Rt + 9, t, 3 * sizeof(double));
  HIP_TRY(hipMemcpyAsync(ctx->d_scratch, Rt, sizeof(Rt), hipMemcpyHostToDevice, ctx->stream));
  const int n = it->second.n_nodes;
  hipLaunchKernelGGL(tree_transform, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, it->second.nodes, n,
                     ctx->d_scratch);
  HIP_TRY(hipGetLastError());
  const int rc = compact_tree(ctx, it->second);  // screening records follow the nodes
  if (rc != MADICP_OK) return rc;
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return MADICP_OK;
}

int madicp_nn_search_device_enqueue(madicp_ctx* ctx, int tree_id, const double* d_queries, int64_t n,
                                    uint32_t* d_out_leaf_id, uint32_t* d_out_node, double* d_out_dist,
                                    int32_t* d_out_depth) {
  if (!ctx) return fail(MADICP_ERR_INVALID, "ctx is null");
  auto it = ctx->trees.find(tree_id);
  if (it == ctx->trees.end()) return fail(MADICP_ERR_INVALID, "unknown tree id");
  if (n <= 0) return MADICP_OK;
  if (!d_queries) return fail(MADICP_ERR_INVALID, "queries is null");
  HIP_TRY(hipSetDevice(ctx->device));
  const long long blocks = std::min<long long>((n + 255) / 256, (long long)ctx->n_cus * 8);
  hipLaunchKernelGGL(nn_descend, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, it->second.desc, d_queries,
                     (long long)n, d_out_leaf_id, d_out_node, d_out_dist, d_out_depth);
  HIP_TRY(hipGetLastError());
  return MADICP_OK;
}

int madicp_nn_search(madicp_ctx* ctx, int tree_id, const double* queries, int64_t n, uint32_t* out_leaf_id,
                     uint32_t* out_node, double* out_dist, int32_t* out_depth) {
  if (!ctx) return fail(MADICP_ERR_INVALID, "ctx is null");
  if (ctx->trees.find(tree_id) == ctx->trees.end()) return fail(MADICP_ERR_INVALID, "unknown tree id");
  if (n <= 0) return MADICP_OK;
  if (!queries) return fail(MADICP_ERR_INVALID, "queries is null");
  HIP_TRY(hipSetDevice(ctx->device));
  double* d_q = nullptr;
  uint32_t *d_leaf = nullptr, *d_node = nullptr;
  double* d_dist = nullptr;
  int32_t* d_depth = nullptr;
  int rc = MADICP_OK;
  auto cleanup = [&]() {
    hipFree(d_q); hipFree(d_leaf); hipFree(d_node); hipFree(d_dist); hipFree(d_depth);
  };
#define NN_TRY(expr)                                                                                      \
  do {                                                                                                    \
    hipError_t e_ = (expr);                                                                               \
    if (e_ != hipSuccess) {                                                                               \
      cleanup();                                                                                          \
      return fail(MADICP_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));                 \
    }                                                                                                     \
  } while (0)
  NN_TRY(hipMalloc(&d_q, sizeof(double) * 3 * (size_t)n));
  if (out_leaf_id) NN_TRY(hipMalloc(&d_leaf, sizeof(uint32_t) * (size_t)n));
  if (out_node) NN_TRY(hipMalloc(&d_node, sizeof(uint32_t) * (size_t)n));
  if (out_dist) NN_TRY(hipMalloc(&d_dist, sizeof(double) * (size_t)n));
  if (out_depth) NN_TRY(hipMalloc(&d_depth, sizeof(int32_t) * (size_t)n));
  NN_TRY(hipMemcpyAsync(d_q, queries, sizeof(double) * 3 * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
  rc = madicp_nn_search_device_enqueue(ctx, tree_id, d_q, n, d_leaf, d_node, d_dist, d_depth);
  if (rc != MADICP_OK) { cleanup(); return rc; }
  if (out_leaf_id) NN_TRY(hipMemcpyAsync(out_leaf_id, d_leaf, sizeof(uint32_t) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
  if (out_node) NN_TRY(hipMemcpyAsync(out_node, d_node, sizeof(uint32_t) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
  if (out_dist) NN_TRY(hipMemcpyAsync(out_dist, d_dist, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
  if (out_depth) NN_TRY(hipMemcpyAsync(out_depth, d_depth, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
  NN_TRY(hipStreamSynchronize(ctx->stream));
#undef NN_TRY
  cleanup();
  return MADICP_OK;
}

// ---- moving side ------------------------------------------------------------------------------------
int madicp_moving_upload(madicp_ctx* ctx, const double* leaf_means, int32_t L, int* out_moving_id) {
  if (!ctx || !leaf_means || !out_moving_id) return fail(MADICP_ERR_INVALID, "null argument");
  if (L < 1) return fail(MADICP_ERR_INVALID, "L must be >= 1");
  HIP_TRY(hipSetDevice(ctx->device));
  DevMoving m;
  m.L = L;
  double* d_xyz = nullptr;
  HIP_TRY(hipMalloc(&d_xyz, sizeof(double) * 3 * (size_t)L));
  hipError_t e = hipMalloc(&m.xyzn, sizeof(double) * 4 * (size_t)L);
  if (e == hipSuccess) e = hipMalloc(&m.matched, (size_t)L);
  if (e == hipSuccess) e = hipMemsetAsync(m.matched, 0, (size_t)L, ctx->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(d_xyz, leaf_means, sizeof(double) * 3 * (size_t)L, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(moving_prep, dim3((L + 255) / 256), dim3(256), 0, ctx->stream, d_xyz, m.xyzn, L);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  hipFree(d_xyz);
  if (e != hipSuccess) {
    hipFree(m.xyzn);
    hipFree(m.matched);
    return fail(MADICP_ERR_DEVICE, std::string("moving upload: ") + hipGetErrorString(e));
  }
  const int id = ctx->next_id++;
  ctx->movings[id] = m;
  *out_moving_id = id;
  return MADICP_OK;
}

int madicp_moving_release(madicp_ctx* ctx, int moving_id) {
  if (!ctx) return fail(MADICP_ERR_INVALID, "ctx is null");
  auto it = ctx->movings.find(moving_id);
  if (it == ctx->movings.end()) return fail(MADICP_ERR_INVALID, "unknown moving id");
  HIP_TRY(hipSetDevice(ctx->device));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  hipFree(it->second.xyzn);
  hipFree(it->second.matched);
  hipFree(it->second.cache_leaf);
  hipFree(it->second.cache_margin);
  ctx->movings.erase(it);
  return MADICP_OK;
}

// ---- registration -----------------------------------------------------------------------------------
int madicp_icp_register_batch_enqueue(madicp_ctx* ctx, int n_scans, const int* moving_ids, const int* tree_ids,
                                      int K, const double* X0, const madicp_icp_params* params, int n_iters) {
  RegArgs a{n_scans, moving_ids, tree_ids, K, X0, params, n_iters, 0, nullptr, nullptr};
  return enqueue_registration(ctx, a);
}

int madicp_icp_fetch(madicp_ctx* ctx, int n_scans, double* out_X, double* out_H, double* out_b, int32_t* out_n_matched,
                     uint64_t* out_visits) {
  if (!ctx) return fail(MADICP_ERR_INVALID, "ctx is null");
  if (n_scans < 1 || n_scans > ctx->last_batch) return fail(MADICP_ERR_INVALID, "n_scans exceeds the last batch");
  HIP_TRY(hipSetDevice(ctx->device));
  for (int s = 0; s < n_scans; ++s)
    HIP_TRY(hipMemcpyAsync(ctx->h_fetch + s, ctx->d_jobs + s, offsetof(Job, trees), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  for (int s = 0; s < n_scans; ++s) {
    const Job& j = ctx->h_fetch[s];
    if (out_X) std::memcpy(out_X + 12 * s, j.X, 12 * sizeof(double));
    if (out_H) std::memcpy(out_H + 36 * s, j.H, 36 * sizeof(double));
    if (out_b) std::memcpy(out_b + 6 * s, j.b, 6 * sizeof(double));
    if (out_n_matched) out_n_matched[s] = j.n_matched;
    if (out_visits) out_visits[s] = j.visits;
  }
  return MADICP_OK;
}

int madicp_icp_fetch_matched(madicp_ctx* ctx, int scan, uint8_t* out_matched, int32_t L) {
  if (!ctx || !out_matched) return fail(MADICP_ERR_INVALID, "null argument");
  if (scan < 0 || scan >= ctx->last_batch) return fail(MADICP_ERR_INVALID, "scan index out of range");
  auto it = ctx->movings.find(ctx->last_moving[scan]);
  if (it == ctx->movings.end()) return fail(MADICP_ERR_INVALID, "moving buffer was released");
  if (L != it->second.L) return fail(MADICP_ERR_INVALID, "L mismatch");
  HIP_TRY(hipSetDevice(ctx->device));
  HIP_TRY(hipMemcpyAsync(out_matched, it->second.matched, (size_t)L, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return MADICP_OK;
}

int madicp_icp_register_batch(madicp_ctx* ctx, int n_scans, const int* moving_ids, const int* tree_ids, int K,
                              double* X, const madicp_icp_params* params, int n_iters, double* out_H, double* out_b,
                              int32_t* out_n_matched, uint64_t* out_visits) {
  const int rc = madicp_icp_register_batch_enqueue(ctx, n_scans, moving_ids, tree_ids, K, X, params, n_iters);
  if (rc != MADICP_OK) return rc;
  return madicp_icp_fetch(ctx, n_scans, X, out_H, out_b, out_n_matched, out_visits);
}

int madicp_icp_register(madicp_ctx* ctx, int moving_id, const int* tree_ids, int K, double X[12],
                        const madicp_icp_params* params, int n_iters, double out_H[36], double out_b[6],
                        uint8_t* out_matched, double* out_X_iters, uint64_t* out_visits) {
  if (!ctx || !X) return fail(MADICP_ERR_INVALID, "null argument");
  HIP_TRY(hipSetDevice(ctx->device));
  double* d_xi = nullptr;
  if (out_X_iters && n_iters > 0) HIP_TRY(hipMalloc(&d_xi, sizeof(double) * 12 * (size_t)n_iters));
  RegArgs a{1, &moving_id, tree_ids, K, X, params, n_iters, 0, nullptr, d_xi};
  int rc = enqueue_registration(ctx, a);
  if (rc == MADICP_OK) rc = madicp_icp_fetch(ctx, 1, X, out_H, out_b, nullptr, out_visits);
  if (rc == MADICP_OK && out_matched) rc = madicp_icp_fetch_matched(ctx, 0, out_matched, ctx->movings.at(moving_id).L);
  if (rc == MADICP_OK && d_xi) {
    hipError_t e = hipMemcpy(out_X_iters, d_xi, sizeof(double) * 12 * (size_t)n_iters, hipMemcpyDeviceToHost);
    if (e != hipSuccess) rc = fail(MADICP_ERR_DEVICE, std::string("x_iters copy: ") + hipGetErrorString(e));
  }
  if (d_xi) {
    hipStreamSynchronize(ctx->stream);
    hipFree(d_xi);
  }
  return rc;
}

int madicp_icp_linearize(madicp_ctx* ctx, int moving_id, const int* tree_ids, int K, const double X[12],
                         const madicp_icp_params* params, double out_H[36], double out_b[6], uint32_t* out_corr,
                         uint8_t* out_matched, uint64_t* out_visits) {
  if (!ctx || !X) return fail(MADICP_ERR_INVALID, "null argument");
  auto mit = ctx->movings.find(moving_id);
  if (mit == ctx->movings.end()) return fail(MADICP_ERR_INVALID, "unknown moving id");
  const int L = mit->second.L;
  HIP_TRY(hipSetDevice(ctx->device));
  uint32_t* d_corr = nullptr;
  if (out_corr && K > 0) HIP_TRY(hipMalloc(&d_corr, sizeof(uint32_t) * (size_t)K * L));
  RegArgs a{1, &moving_id, tree_ids, K, X, params, 1, kFlagNoUpdate, d_corr, nullptr};
  const int saved_graph = ctx->use_graph;
  ctx->use_graph = 0;  // pointers in the job differ per call; nothing to gain from a graph for one round
  int rc = enqueue_registration(ctx, a);
  ctx->use_graph = saved_graph;
  if (rc == MADICP_OK) rc = madicp_icp_fetch(ctx, 1, nullptr, out_H, out_b, nullptr, out_visits);
  if (rc == MADICP_OK && out_matched) rc = madicp_icp_fetch_matched(ctx, 0, out_matched, L);
  if (rc == MADICP_OK && d_corr) {
    hipError_t e = hipMemcpy(out_corr, d_corr, sizeof(uint32_t) * (size_t)K * L, hipMemcpyDeviceToHost);
    if (e != hipSuccess) rc = fail(MADICP_ERR_DEVICE, std::string("corr copy: ") + hipGetErrorString(e));
  }
  if (d_corr) {
    hipStreamSynchronize(ctx->stream);
    hipFree(d_corr);
  }
  return rc;
}

int madicp_icp_time_linearize(madicp_ctx* ctx, int n_scans, const int* moving_ids, const int* tree_ids, int K,
                              const double* X0, const madicp_icp_params* params, int n_launches, double* out_avg_us,
                              uint64_t* out_visits_per_launch) {
  if (n_launches < 1) return fail(MADICP_ERR_INVALID, "n_launches must be >= 1");
  // a one-round registration without pose update, its round-0 kernel launched n_launches times
  RegArgs a{n_scans, moving_ids, tree_ids, K, X0, params, 1, kFlagNoUpdate, nullptr, nullptr, n_launches, out_avg_us};
  int rc = enqueue_registration(ctx, a);
  if (rc == MADICP_OK && out_visits_per_launch) rc = madicp_icp_fetch(ctx, n_scans, nullptr, nullptr, nullptr, nullptr, out_visits_per_launch);
  return rc;
}

int madicp_icp_time_registration(madicp_ctx* ctx, int n_scans, const int* moving_ids, const int* tree_ids, int K,
                                 const double* X0, const madicp_icp_params* params, int n_iters, int reps,
                                 double* out_linearize_avg_us, double* out_solve_avg_us, uint64_t* out_visits_per_launch) {
  if (!ctx || reps < 1 || n_iters < 1) return fail(MADICP_ERR_INVALID, "bad argument");
  if (ctx->comm) return fail(MADICP_ERR_INVALID, "not available with a communicator");
  HIP_TRY(hipSetDevice(ctx->device));
  if (!ctx->ev_t0) {
    HIP_TRY(hipEventCreate(&ctx->ev_t0));
    HIP_TRY(hipEventCreate(&ctx->ev_t1));
  }
  RegArgs a{n_scans, moving_ids, tree_ids, K, X0, params, n_iters, 0, nullptr, nullptr};
  int rc = enqueue_registration(ctx, a);  // warm-up; also instantiates the graph
  if (rc != MADICP_OK) return rc;
  HIP_TRY(hipEventRecord(ctx->ev_t0, ctx->stream));
  for (int r = 0; r < reps; ++r) {
    rc = enqueue_registration(ctx, a);
    if (rc != MADICP_OK) return rc;
  }
  HIP_TRY(hipEventRecord(ctx->ev_t1, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  float ms_reg = 0.f;
  HIP_TRY(hipEventElapsedTime(&ms_reg, ctx->ev_t0, ctx->ev_t1));
  if (out_visits_per_launch) {
    rc = madicp_icp_fetch(ctx, n_scans, nullptr, nullptr, nullptr, nullptr, out_visits_per_launch);
    if (rc != MADICP_OK) return rc;
    for (int s = 0; s < n_scans; ++s) out_visits_per_launch[s] /= (uint64_t)n_iters;
  }
  // icp_final alone, replayed `reps` times as a graph: what is left of a registration is its n_iters icp_round launches
  const Geometry geo = pick_geometry(ctx, [&] { int m = 0; for (int s = 0; s < n_scans; ++s) m = std::max(m, ctx->movings.at(moving_ids[s]).L); return m; }(), K, n_scans);
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  HIP_TRY(hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
  hipLaunchKernelGGL(icp_final, dim3(n_scans), dim3(kBlock), 0, ctx->stream, ctx->d_jobs, ctx->d_partials,
                     (const double*)nullptr, geo.grid, n_scans);
  HIP_TRY(hipStreamEndCapture(ctx->stream, &graph));
  HIP_TRY(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
  {
    hipGraphLaunch(exec, ctx->stream);
    hipEventRecord(ctx->ev_t0, ctx->stream);
    for (int r = 0; r < reps; ++r) hipGraphLaunch(exec, ctx->stream);
    hipEventRecord(ctx->ev_t1, ctx->stream);
    hipError_t e = hipStreamSynchronize(ctx->stream);
    float ms_final = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms_final, ctx->ev_t0, ctx->ev_t1);
    if (e != hipSuccess) rc = fail(MADICP_ERR_DEVICE, std::string("final timing: ") + hipGetErrorString(e));
    if (rc == MADICP_OK) {
      const double per_reg = 1e3 * ms_reg / reps, per_final = 1e3 * ms_final / reps;
      if (out_solve_avg_us) *out_solve_avg_us = per_final;
      if (out_linearize_avg_us) *out_linearize_avg_us = (per_reg - per_final) / n_iters;
    }
  }
  hipGraphExecDestroy(exec);
  hipGraphDestroy(graph);
  return rc;
}


// ---- multi-GPU --------------------------------------------------------------------------------------
int madicp_comm_unique_id(uint8_t out_id[128]) {
  if (!out_id) return fail(MADICP_ERR_INVALID, "out_id is null");
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  ncclUniqueId id;
  NCCL_TRY(ncclGetUniqueId(&id));
  std::memcpy(out_id, &id, 128);
  return MADICP_OK;
}

int madicp_comm_init(madicp_ctx* ctx, const uint8_t unique_id[128], int n_ranks, int rank) {
  if (!ctx || !unique_id) return fail(MADICP_ERR_INVALID, "null argument");
  if (n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail(MADICP_ERR_INVALID, "bad rank / n_ranks");
  if (ctx->comm) return fail(MADICP_ERR_INVALID, "communicator already initialised");
  HIP_TRY(hipSetDevice(ctx->device));
  ncclUniqueId id;
  std::memcpy(&id, unique_id, 128);
  NCCL_TRY(ncclCommInitRank(&ctx->comm, n_ranks, id, rank));
  ctx->n_ranks = n_ranks;
  ctx->rank = rank;
  return MADICP_OK;
}

int madicp_comm_destroy(madicp_ctx* ctx) {
  if (!ctx) return fail(MADICP_ERR_INVALID, "ctx is null");
  if (ctx->comm) {
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    NCCL_TRY(ncclCommDestroy(ctx->comm));
    ctx->comm = nullptr;
    ctx->n_ranks = 1;
    ctx->rank = 0;
  }
  return MADICP_OK;
}

}  // extern "C"

#ifdef MADICP_STAMPS
// development only (tools/stamps.py): copies the icp_round wall-clock stamps of scan 0, [16 rounds][256 wg][16]
extern "C" int madicp_debug_stamps(madicp_ctx* ctx, unsigned long long* out) {
  if (!ctx || !out) return MADICP_ERR_INVALID;
  hipStreamSynchronize(ctx->stream);
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(madicp::g_stamps), sizeof(unsigned long long) * 16 * 256 * 16) == hipSuccess
             ? 0 : MADICP_ERR_DEVICE;
}
#endif
