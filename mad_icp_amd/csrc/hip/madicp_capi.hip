// libmadicp_hip.so — implementation of the C ABI declared in include/madicp_hip.h.
// Context / buffer management, launch sequencing (eager or captured hipGraph), optional RCCL all-reduce.
//
// Streams.  `stream` (compute) carries every registration; `copy` carries what feeds them — tree uploads + their
// screening-record builds, the moving leaves of the NEXT scan, its Job — so those overlap the registration in flight.
// Hand-over is by events only; nothing on the registration path allocates, frees or synchronises the device:
// device memory comes from a small in-context pool (a released buffer is reused once the event recorded at its
// release has passed), pinned staging is grow-only, results are written by the last kernel of a registration straight
// into a pinned host block (Job::host_out) that the caller reads after one event wait.
#include "madicp_hip_measure.h"
#include "kernels.hip.h"
#include "frontend.hip.h"  // + tree_build.hip.h: device front-end

#include <hip/hip_ext.h>
#include <rccl/rccl.h>

#include <sched.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <thread>
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

#define RC_TRY(expr)              \
  do {                            \
    const int rc_ = (expr);       \
    if (rc_ != MADICP_OK) return rc_; \
  } while (0)

#define NCCL_TRY(expr)                                                                            \
  do {                                                                                            \
    ncclResult_t r_ = (expr);                                                                     \
    if (r_ != ncclSuccess)                                                                        \
      return fail(MADICP_ERR_COMM, std::string(#expr) + ": " + ncclGetErrorString(r_));          \
  } while (0)

// a pair of events (one per stream) shared by the buffers released together; destroyed with the last of them
struct EventHolder {
  hipEvent_t ev = nullptr;       // behind everything enqueued on the compute stream
  hipEvent_t ev_copy = nullptr;  // behind everything enqueued on the copy stream
  hipEvent_t ev_build = nullptr; // behind everything enqueued on the build stream (only while a look-ahead build has used it)
  ~EventHolder() {
    if (ev) hipEventDestroy(ev);
    if (ev_copy) hipEventDestroy(ev_copy);
    if (ev_build) hipEventDestroy(ev_build);
  }
};
using EventRef = std::shared_ptr<EventHolder>;

struct PoolEntry {
  void* ptr;
  EventRef after;  // work that may still touch the buffer (null: none)
};

struct DevTree {
  char* block = nullptr;       // one device allocation: [nodes | top_exit | top_dfs | top_link] (uploaded) [cnodes | leaves | top]
  madicp_node* nodes = nullptr;
  CNode* cnodes = nullptr;    // 16-byte screening records, same indexing
  LeafRec* leaves = nullptr;  // dense 64-byte leaf records, by leaf ordinal
  CNode* top = nullptr;       // top levels, breadth first (staged into LDS by icp_round)
  int4* top_exit = nullptr;
  int* top_dfs = nullptr;
  unsigned int* top_link = nullptr;
  int32_t n_top = 0;
  TreeDesc desc{};            // what the kernels get by value
  int32_t n_nodes = 0, n_leaves = 0;
  double rho2 = 0.0;          // max |m - o|_2 over the internal nodes (host computed; rotation invariant)
  hipEvent_t ready = nullptr; // recorded on the copy stream behind the upload + record build
  bool compute_waited = false;  // the compute stream is already ordered behind `ready`
};
struct UseEvent {
  hipEvent_t ev = nullptr;
  ~UseEvent() {
    if (ev) hipEventDestroy(ev);
  }
};

struct DevMoving {
  double* xyzn = nullptr;  // (L,4)
  uint8_t* matched = nullptr;
  int32_t L = 0;
  int32_t cap_L = 0;
  // correspondence cache of the registration in flight for this scan: (K,L) each, grow-only
  uint32_t* cache_leaf = nullptr;
  float* cache_margin = nullptr;
  size_t cache_cap = 0;  // elements
  // pinned staging of the moving set, already in the device layout (x, y, z, |p|): ONE copy-engine transfer feeds the
  // registration, no kernel shares the compute units with the registration in flight
  double* h_in = nullptr;
  size_t h_in_cap = 0;        // doubles
  hipEvent_t h_in_read = nullptr;  // the transfer that reads h_in has run
  hipEvent_t ready = nullptr;      // xyzn valid (recorded on whichever stream prepared it)
  bool on_copy = false;            // `ready` was recorded on the copy stream and the compute stream has not waited yet
  // behind the last registration that read this set (compute stream; ONE event per batch, shared by its sets):
  // madicp_moving_update_async puts the copy stream behind it
  std::shared_ptr<UseEvent> last_use;
  unsigned long long copy_seq = 0;  // `ready` is the copy_seq-th event recorded on the copy stream for a moving set
};

struct GraphKey {  // everything a captured launch sequence bakes in
  int grid, batch, iters, qpt, comm, lds, K, rpt, trace, slot, persist;  // persist: 0 per-round launches, 1 icp_persist, 2 xcd_fold
  bool operator<(const GraphKey& o) const {
    return std::tie(grid, batch, iters, qpt, comm, lds, K, rpt, trace, slot, persist) <
           std::tie(o.grid, o.batch, o.iters, o.qpt, o.comm, o.lds, o.K, o.rpt, o.trace, o.slot, o.persist);
  }
};
struct Geometry {
  int grid;             // workgroups per scan (multiple of 8)
  int ranges_per_tree;  // units per tree
  int qpt;              // leaves a lane walks at once: 1, 2 or 4
  int lds_bytes;        // dynamic LDS of the launch: kTopLdsBytes when units are big enough to stage a tree's top, else 0
  int queue;            // 1: units are long enough for queued walks (the QUEUE instantiation of icp_round is launched)
};

// one streamed registration in flight (madicp_stream_submit .. madicp_stream_collect)
struct StreamSlot {
  int moving_id = -1;       // a DevMoving owned by this slot
  Job* d_job = nullptr;     // this slot's Job on the device (its graphs bake the pointer)
  Job* h_job = nullptr;     // pinned staging of it
  Outbox* d_outbox = nullptr;    // device: icp_final leaves the results here when icp_publish carries them out (option "publish_side")
  bool side = false;             // this ticket's results come through the outbox / the side stream
  HostResult* h_out = nullptr;   // pinned: icp_final writes the results here
  uint8_t* h_matched = nullptr;  // pinned: and the matched_ flags here
  size_t h_matched_cap = 0;
  hipEvent_t ev_up = nullptr;    // copy stream: Job + moving leaves are on the device
  hipEvent_t ev_done = nullptr;  // compute stream: the registration has finished
  bool pending = false;
  bool by_seq = false;      // completion is published through h_out->seq (no event was recorded)
  int ticket = -1;
  int L = 0;
};

}  // namespace

struct madicp_ctx {
  int device = 0;
  hipStream_t stream = nullptr;  // compute
  hipStream_t copy = nullptr;    // feeds: uploads, record builds, next scan's leaves
  hipStream_t build = nullptr;   // look-ahead tree construction (madicp_tree_build_begin); created on first use
  bool own_stream = false;
  int n_cus = 256;

  std::unordered_map<int, DevTree> trees;
  std::unordered_map<int, DevMoving> movings;
  int next_id = 1;

  // device memory pool
  std::multimap<size_t, PoolEntry> pool;            // free buffers by capacity
  std::unordered_map<void*, size_t> alloc_bytes;    // capacity of every buffer handed out
  size_t pool_bytes = 0;

  // registration state
  Job* d_jobs = nullptr;       // [MADICP_MAX_BATCH]
  // pinned staging ring: the host may run kStageSlots-1 batches ahead of the device
  static constexpr int kStageSlots = 4;
  Job* h_stage[kStageSlots] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t stage_ev[kStageSlots] = {nullptr, nullptr, nullptr, nullptr};
  int stage_next = 0;
  Job* h_fetch = nullptr;      // pinned read-back block
  // results of whole batches published to the host by a kernel behind the batch (madicp_icp_publish_enqueue / _collect): a ring,
  // so that the host can collect batch i after it has enqueued batch i + 1
  static constexpr int kPubSlots = 4;
  madicp::HostResult* h_pub[kPubSlots] = {nullptr, nullptr, nullptr, nullptr};
  int pub_n[kPubSlots] = {0, 0, 0, 0};
  int pub_ticket[kPubSlots] = {-1, -1, -1, -1};
  int pub_next = 0;
  unsigned long long copy_seq = 0;  // moving sets made ready on the copy stream so far (DevMoving::copy_seq)
  double* d_partials = nullptr;
  size_t partials_cap = 0;     // doubles
  long long partials_key = -1;  // the launch shape(s) the zero padding rows of d_partials are valid for
  double* d_totals = nullptr;  // [2 round parities][MADICP_MAX_BATCH][kAcc]: this rank's adders of a sharded round, reduced in place
  unsigned int* d_tickets = nullptr;  // [MADICP_MAX_BATCH]: arrival counters of icp_round's TAIL variant (zero between launches)
  hipStream_t stream2 = nullptr;      // second half of a sharded batch (option "shard_split"); created on first use
  hipStream_t pub = nullptr;          // streamed registrations: icp_publish carries results to the host beside the next registration
  int publish_side = 1;               // option "publish_side"
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  unsigned long long* d_xch = nullptr;  // icp_persist's exchange granules (kernels.hip.h), sized for every admissible geometry
  uint32_t epoch = 0;          // one per enqueued registration: Job::epoch
  int last_batch = 0;
  std::vector<int> last_moving;

  // pinned staging for tree uploads (two buffers, alternating)
  char* h_tree[2] = {nullptr, nullptr};
  size_t h_tree_cap[2] = {0, 0};
  hipEvent_t h_tree_ev[2] = {nullptr, nullptr};
  int h_tree_next = 0;

  // streamed registrations
  static constexpr int kStreamSlots = 4;
  StreamSlot slots[kStreamSlots];
  int next_ticket = 0;

  // options
  int blocks_per_cu = 1;  // icp_round workgroups (768 threads) per CU
  int deal_trees = 2;     // a Job lists the caller's trees dealt over the eight XCD pieces, rows of eight in alternating direction (fill_job)
  int units_per_wg = 1;   // when a scan has more trees than workgroups: cut every tree's leaves into enough ranges for at least
                          // this many (tree, range) units per workgroup (see pick_geometry)
  int use_graph = 1;
  int comm_graph = 0;     // capture the RCCL calls too (off: rounds are launched eagerly with a communicator)
  int qpt_override = 0;
  int cache_corr = 1;  // reuse correspondences across GN rounds when provably unchanged
  int deep_min_leaves = 512;  // option "deep_min_leaves": with 24 keyframes or more and two scans in flight (48 or more and one), a launch with
                         // more trees than workgroups per XCD piece is DEEP (one range x all the piece's trees per workgroup) when a range
                         // holds at least this many leaves (pick_geometry)
  int interleave = 2;    // option "interleave_ranges": a range is every RPT-th group of 64 leaves instead of a contiguous stretch of the
                         // scan (kernels.hip.h, "Ranges"): 0 never, 1 DEEP launches (a batch shares the chip), 2 every launch
  int cache_gate = 1;   // option "cache_gate": a pair that keeps its leaf and was rejected with more slack than it has moved since is
                        // not evaluated again (kernels.hip.h, "Gate reuse")
  int queue_walks = 8192; // option "leaf_major": a DEEP launch (a batch shares the chip) runs a round leaf-major — moving leaf once per
                        // pass for all the workgroup's trees, walkers queued and walked densely — when the workgroup walked fewer
                        // than this many nodes per pass in the previous round (0: never; icp_leaf_major.inc.h)
  int nn_lds_top = 0;  // option "nn_lds_top": nn_search batches of >= 16 k queries walk the tree's top levels from LDS
                       // (nn_descend_top).  Off: measured SLOWER for one 120 k-query launch (8.7 vs 6.6 us against a
                       // 20 k-leaf tree, 10.7 vs 9.2 us against a 120 k-leaf tree) — staging 48 KiB per workgroup costs
                       // more than the ~11 LDS levels save in a kernel this short
  int eager_when_busy = 1; // a registration queued behind another is launched kernel by kernel, not as a graph (run_rounds)
  int seq_completion = 1;  // streamed registrations publish completion through HostResult::seq instead of an event
  int host_feed_wait = 1;  // ... and the host, not the stream, waits for their feed while another one is in flight
  int match_all = 0;       // option "match_all_rounds": the matched flags a registration returns are the OR over all its rounds
  int persistent = 0;      // all rounds of a registration as ONE launch (icp_persist) where the geometry admits it
  int xcd_fold = 0;        // per-round launches whose group leaders fold their XCD's rows at the end of the launch (experiment)
  int debug_collective_us = 0;  // development: a delay kernel of this length behind every collective (tools/shard_probe.py)
  int shard_tail = 0;      // sharded rounds leave the rank's adders themselves (icp_round's TAIL variant) instead of an icp_reduce
                           // launch.  Off: built, bit-identical, measured SLOWER (profiles/r4_c_shard_probe.md: the 256 tickets
                           // on one address and the cross-XCD read of the rows cost ~8 us at the end of every round; the
                           // separate icp_reduce launch costs 4.5 us and no gap)
  int build_after_registration = 0;  // option (experiment, default off): a look-ahead construction's kernels wait for the registration in
                                     // flight (frontend_capi.inc.h; measured: does not remove the look-ahead cliff, profiles/r5_lookahead_matrix.md)
  hipEvent_t ev_build_gate = nullptr;
  int shard_p2p = 0;       // sharded rounds join over peer-mapped mailboxes inside the round kernel (madicp_p2p_attach) instead
                           // of icp_reduce + a collective between two rounds
  unsigned long long* p2p_box = nullptr;                  // this rank's mailbox (kP2pBoxWords; fine-grained device memory)
  unsigned long long* p2p_peer[madicp::kMaxRanks] = {};   // every rank's mailbox as mapped here ([rank] = p2p_box)
  bool p2p_opened[madicp::kMaxRanks] = {};                // ... opened through hipIpcOpenMemHandle (to be closed)
  bool p2p_attached = false;
  bool p2p_fresh = false;    // the own mailbox has been zeroed and exported since the last attach (madicp_p2p_export)
  bool p2p_fine = false;     // ... and it is fine-grained device memory (peers' stores are visible to a running kernel)
  bool p2p_broken = false;   // a registration of this mailbox session lost a peer: the ranks' counters may disagree from here on
  void* d_f32[2] = {nullptr, nullptr};  // device landing blocks of float uploads, one per pinned staging block (h_tree)
  size_t d_f32_cap[2] = {0, 0};
  int upload_f32 = 1;        // option "upload_f32": a cloud of float-exact coordinates crosses PCIe as floats (frontend_capi.inc.h)
  int p2p_allow_coarse = 0;  // option "p2p_allow_coarse": accept a coarse-grained mailbox (ranks that share ONE device only)
  unsigned int p2p_epoch = 0;                             // sharded registrations so far (the same count on every rank)
  int shard_split = 1;     // a sharded batch of >= 4 scans runs as two halves on two streams: one half's all-reduce under the
                           // other half's round (profiles/r4_c_shard_probe.md: -14 % per registration at 8 scans with a 15 us
                           // collective; a loss without one, and with halves of one scan)
  int stage_min_leaves = 1024;  // LDS staging threshold (leaves per unit); 0 = always, huge = never (measured break-even ~1000)

  std::map<GraphKey, hipGraphExec_t> graphs;

  hipEvent_t ev_t0 = nullptr, ev_t1 = nullptr;  // measurement entry points

  // multi-GPU: RCCL communicator, or a host-staged transport supplied by the caller (madicp_comm_init_host)
  ncclComm_t comm = nullptr;
  madicp_host_allreduce_fn host_ar = nullptr;
  void* host_ar_user = nullptr;
  char* h_comm = nullptr;       // pinned staging of the host transport (grow-only)
  size_t h_comm_cap = 0;
  int n_ranks = 1, rank = 0;
  int comm_timeout_ms = 60000;  // bounded host wait behind a registration's collectives
  bool sharded() const { return comm != nullptr || host_ar != nullptr; }

  // how the host waits for a sequence number the device publishes (stream_collect, tree_build)
  int wait_mode = 0;        // 0 spin, 1 sched_yield, 2 sleep ~50 us
  int wait_timeout_ms = 0;  // 0: unbounded

  // device front-end (frontend_capi.inc.h): resident clouds + builder scratch, created on first use
  struct Front;
  Front* front = nullptr;
};

namespace {

void front_destroy(madicp_ctx* ctx);  // frontend_capi.inc.h

constexpr size_t kAlign = 256;
size_t align_up(size_t v) { return (v + kAlign - 1) / kAlign * kAlign; }

// ---- device memory pool -----------------------------------------------------------------------------
// alloc: a pooled buffer of capacity in [bytes, 2*bytes] if there is one (ordering `user` behind the work recorded at
// its release), else hipMalloc.  Trees of consecutive scans have nearly the same size, so the steady state of an
// odometry run allocates nothing.
int pool_alloc(madicp_ctx* ctx, size_t bytes, hipStream_t user, void** out) {
  bytes = std::max<size_t>(align_up(bytes), kAlign);
  auto it = ctx->pool.lower_bound(bytes);
  if (it != ctx->pool.end() && it->first <= 2 * bytes) {
    if (it->second.after) {
      if (it->second.after->ev) HIP_TRY(hipStreamWaitEvent(user, it->second.after->ev, 0));
      if (it->second.after->ev_copy) HIP_TRY(hipStreamWaitEvent(user, it->second.after->ev_copy, 0));
      if (it->second.after->ev_build) HIP_TRY(hipStreamWaitEvent(user, it->second.after->ev_build, 0));
    }
    *out = it->second.ptr;
    ctx->pool_bytes -= it->first;
    ctx->pool.erase(it);
    return MADICP_OK;
  }
  void* p = nullptr;
  hipError_t e = hipMalloc(&p, bytes);
  if (e != hipSuccess) {  // out of memory: give the pool back and retry once
    (void)hipGetLastError();
    hipDeviceSynchronize();
    for (auto& pe : ctx->pool) {
      ctx->alloc_bytes.erase(pe.second.ptr);
      hipFree(pe.second.ptr);
    }
    ctx->pool.clear();
    ctx->pool_bytes = 0;
    e = hipMalloc(&p, bytes);
  }
  if (e != hipSuccess) return fail(MADICP_ERR_DEVICE, std::string("hipMalloc: ") + hipGetErrorString(e));
  ctx->alloc_bytes[p] = bytes;
  *out = p;
  return MADICP_OK;
}

// events behind everything enqueued so far on the two streams (neither stream is made to wait for the other)
int fence_event(madicp_ctx* ctx, EventRef* out) {
  auto h = std::make_shared<EventHolder>();
  HIP_TRY(hipEventCreateWithFlags(&h->ev, hipEventDisableTiming));
  HIP_TRY(hipEventCreateWithFlags(&h->ev_copy, hipEventDisableTiming));
  HIP_TRY(hipEventRecord(h->ev, ctx->stream));
  HIP_TRY(hipEventRecord(h->ev_copy, ctx->copy));
  if (ctx->build) {
    HIP_TRY(hipEventCreateWithFlags(&h->ev_build, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(h->ev_build, ctx->build));
  }
  *out = h;
  return MADICP_OK;
}

void pool_free(madicp_ctx* ctx, void* p, const EventRef& after) {
  if (!p) return;
  auto it = ctx->alloc_bytes.find(p);
  if (it == ctx->alloc_bytes.end()) return;
  const size_t cap = it->second;
  constexpr size_t kPoolMax = size_t(1) << 30;  // keep at most 1 GiB parked
  if (ctx->pool_bytes + cap > kPoolMax) {
    if (after && after->ev) hipEventSynchronize(after->ev);
    if (after && after->ev_copy) hipEventSynchronize(after->ev_copy);
    if (after && after->ev_build) hipEventSynchronize(after->ev_build);
    ctx->alloc_bytes.erase(it);
    hipFree(p);
    return;
  }
  ctx->pool.emplace(cap, PoolEntry{p, after});
  ctx->pool_bytes += cap;
}

int ensure_partials(madicp_ctx* ctx, size_t doubles) {
  if (doubles <= ctx->partials_cap) return MADICP_OK;
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  if (ctx->d_partials) HIP_TRY(hipFree(ctx->d_partials));
  ctx->d_partials = nullptr;
  ctx->partials_cap = 0;
  doubles = std::max(doubles, (size_t)2 * 288 * kAcc * 8);  // room for the common geometries: grows once
  HIP_TRY(hipMalloc(&ctx->d_partials, doubles * sizeof(double)));
  ctx->partials_cap = doubles;
  ctx->partials_key = -1;
  // cached graphs hold the old pointer
  for (auto& g : ctx->graphs) hipGraphExecDestroy(g.second);
  ctx->graphs.clear();
  return MADICP_OK;
}

// one poll of a host wait loop, by option "wait_mode"
inline void wait_pause(const madicp_ctx* ctx) {
  if (ctx->wait_mode == 1)
    sched_yield();
  else if (ctx->wait_mode == 2)
    std::this_thread::sleep_for(std::chrono::microseconds(50));
  else
    __builtin_ia32_pause();
}

// A collective that will not complete: the communicator is aborted (which releases the kernels parked on the stream), the
// graphs that captured its calls are dropped, the context falls back to one rank.  Always MADICP_ERR_COMM.
int comm_abort(madicp_ctx* ctx, const std::string& why) {
  for (auto& g : ctx->graphs) hipGraphExecDestroy(g.second);
  ctx->graphs.clear();
  if (ctx->comm) ncclCommAbort(ctx->comm);
  ctx->comm = nullptr;
  ctx->n_ranks = 1;
  ctx->rank = 0;
  return fail(MADICP_ERR_COMM, why + "; communicator aborted");
}

// Host wait for everything enqueued on `s`.  Without a communicator this is hipStreamSynchronize.  With one, a peer
// that never joins a collective would park this rank's stream for ever: poll instead, ask RCCL for asynchronous errors,
// and after "comm_timeout_ms" abort the communicator — the hang becomes MADICP_ERR_COMM.
int bounded_sync(madicp_ctx* ctx, hipStream_t s) {
  if (!ctx->comm) {
    HIP_TRY(hipStreamSynchronize(s));
    return MADICP_OK;
  }
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned spins = 0;; ++spins) {
    const hipError_t q = hipStreamQuery(s);
    if (q == hipSuccess) return MADICP_OK;
    if (q != hipErrorNotReady) return fail(MADICP_ERR_DEVICE, std::string("stream: ") + hipGetErrorString(q));
    if ((spins & 63) == 63) {
      ncclResult_t ar = ncclSuccess;
      const bool bad = ncclCommGetAsyncError(ctx->comm, &ar) == ncclSuccess && ar != ncclSuccess && ar != ncclInProgress;
      const long long ms = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count();
      if (bad || ms > ctx->comm_timeout_ms)
        return comm_abort(ctx, bad ? std::string("RCCL asynchronous error: ") + ncclGetErrorString(ar)
                                   : "a collective did not complete within " + std::to_string(ctx->comm_timeout_ms) +
                                         " ms (a rank did not join?)");
    }
    if (spins < 256)
      __builtin_ia32_pause();
    else
      std::this_thread::sleep_for(std::chrono::microseconds(20));
  }
}

// element-wise reduction of a DEVICE buffer over the ranks, stream-ordered on the compute stream: RCCL, or the caller's
// host transport (copy out, wait, callback, copy back)
int all_reduce(madicp_ctx* ctx, void* d_buf, size_t count, int kind, hipStream_t s) {
  if (ctx->comm) {
    if (kind == MADICP_REDUCE_SUM_F64)
      NCCL_TRY(ncclAllReduce(d_buf, d_buf, count, ncclDouble, ncclSum, ctx->comm, s));
    else
      NCCL_TRY(ncclAllReduce(d_buf, d_buf, count, ncclUint8, ncclMax, ctx->comm, s));
    if (ctx->debug_collective_us > 0)  // (development: a one-rank all-reduce launches nothing — stand in for its latency)
      hipLaunchKernelGGL(debug_delay, dim3(1), dim3(64), 0, s, (unsigned long long)ctx->debug_collective_us * 100ull);
    return MADICP_OK;
  }
  const size_t bytes = count * (kind == MADICP_REDUCE_SUM_F64 ? sizeof(double) : 1);
  if (ctx->h_comm_cap < bytes) {
    if (ctx->h_comm) HIP_TRY(hipHostFree(ctx->h_comm));
    ctx->h_comm = nullptr;
    ctx->h_comm_cap = 0;
    const size_t cap = std::max<size_t>(bytes + bytes / 4, 4096);
    HIP_TRY(hipHostMalloc(&ctx->h_comm, cap, hipHostMallocDefault));
    ctx->h_comm_cap = cap;
  }
  HIP_TRY(hipMemcpyAsync(ctx->h_comm, d_buf, bytes, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  const int rc = ctx->host_ar(ctx->host_ar_user, ctx->h_comm, (int64_t)count, kind);
  if (rc != 0) return fail(MADICP_ERR_COMM, "host all-reduce callback failed with code " + std::to_string(rc));
  HIP_TRY(hipMemcpyAsync(d_buf, ctx->h_comm, bytes, hipMemcpyHostToDevice, s));
  HIP_TRY(hipStreamSynchronize(s));  // (the staging block is reused by the next call, possibly for another stream's part)
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
  const long long max_useful = (long long)std::max(1, K) * ((max_L + 63) / 64);  // never below one wave of leaves per unit
  grid = std::max<long long>(8, std::min(grid, max_useful) / 8 * 8);
  g.grid = static_cast<int>(grid);
  g.ranges_per_tree = static_cast<int>(std::max<long long>(1, grid / std::max(1, K)));
  if (ctx->units_per_wg > 1 && K > 0) {
    // More trees than workgroups (a batch shares the chip): with one range per tree a workgroup owns whole trees, and trees
    // differ in cost (how many of the scan's leaves still have to walk them, how many match) — the launch waits for the
    // workgroup with the expensive ones.  Finer ranges give every workgroup of an XCD piece a slice of ALL the piece's trees.
    const long long want = ((long long)ctx->units_per_wg * grid + K - 1) / K;
    const long long cap = std::max<long long>(1, max_L / 256);  // (a range of fewer than 256 leaves is not worth a descriptor)
    g.ranges_per_tree = static_cast<int>(std::max<long long>(g.ranges_per_tree, std::min(want, cap)));
  }
  // DEEP launches: more trees than workgroups per XCD piece (a batch shares the chip).  Every workgroup then gets ONE range
  // of the scan and ALL the trees of its piece — ranges_per_tree = workgroups per piece, so that its units u_first, u_first +
  // nslots, ... are the same range of consecutive trees — which is what the leaf-major rounds need (icp_leaf_major.inc.h)
  const int nslots = g.grid / 8;
  // (how many leaves a range must hold for that: two passes of a workgroup — below, the leaf-major queue has nothing to compact —
  // unless the piece holds three trees or more: then one unit per workgroup means the launch waits for the workgroups that drew
  // the newest keyframes, and one range of ALL the piece's trees per workgroup pays from 512 leaves on; measured,
  // profiles/r6_deep_threshold.md: 24-64 keyframes x 1-2 scans in flight + 3 .. + 40 %, 16 keyframes - 9 %)
  // (one scan in flight against 24-47 keyframes stays as it was: - 2 .. - 4 % that way at 32 keyframes, + 3 % at 24)
  const bool small_ranges_pay = K >= 24 && (batch >= 2 || K >= 48);
  const int deep_min = small_ranges_pay ? ctx->deep_min_leaves : std::max(ctx->deep_min_leaves, madicp::kQueueMinPasses * madicp::kBlock);
  if (K >= 8 && ctx->queue_walks > 0 && g.qpt == 1 && g.ranges_per_tree < nslots &&
      max_L / nslots >= deep_min && (K + 7) / 8 + 1 <= madicp::kDeepTrees)
    g.ranges_per_tree = nslots;
  const int per_range = (max_L + g.ranges_per_tree - 1) / g.ranges_per_tree;
  g.lds_bytes = (K > 0 && per_range >= ctx->stage_min_leaves) ? kTopLdsBytes : 0;
  g.queue = (K >= 8 && ctx->queue_walks > 0 && g.qpt == 1 && g.ranges_per_tree == nslots &&
             per_range >= deep_min) ? 1 : 0;
  return g;
}

struct Launch {  // one registration's launch shape
  int grid, batch, iters, qpt, lds, K, rpt, trace, queue;
};

constexpr size_t kXchRowsMax = 1024 + 8 * MADICP_MAX_BATCH;  // level-1 rows + level-2 rows of the largest admissible launch

// May this registration run as ONE launch?  icp_persist needs every workgroup resident at once (one 768-thread workgroup
// per CU is all a CU holds), no collective between rounds, and tags of 8 bits of round.
bool use_persist(const madicp_ctx* ctx, const Launch& l) {
  return ctx->persistent && !ctx->sharded() && !l.trace && l.iters >= 2 && l.iters <= 250 && ctx->blocks_per_cu == 1 &&
         (long long)l.grid * l.batch <= ctx->n_cus && (l.grid >> 3) <= kJoinGroups && l.K >= 1 &&
         madicp::xch_granules(l.batch, l.grid) <= kXchRowsMax * 2 * madicp::kRowGranules;
}

bool use_fold(const madicp_ctx* ctx, const Launch& l);

// Where a launch sequence runs and which scratch it owns.  A registration is one part on the compute stream with the
// context's buffers; a sharded batch split in two halves (option "shard_split") is two parts with disjoint regions of the
// same buffers, the second one on the context's second stream.
struct Part {
  hipStream_t s = nullptr;
  Job* jobs = nullptr;
  double* partials = nullptr;
  unsigned long long* xch = nullptr;
  double* totals[2] = {nullptr, nullptr};  // round parity: written by round r (or icp_reduce), all-reduced, read by round r + 1
  unsigned int* tickets = nullptr;
  int scan0 = 0;  // first scan of the part within the registration's batch (its rows of the peer mailboxes)
};
Part whole_part(madicp_ctx* ctx, Job* d_jobs) {
  Part p;
  p.s = ctx->stream;
  p.jobs = d_jobs;
  p.partials = ctx->d_partials;
  p.xch = ctx->d_xch;
  p.totals[0] = ctx->d_totals;
  p.totals[1] = ctx->d_totals + (size_t)MADICP_MAX_BATCH * kAcc;
  p.tickets = ctx->d_tickets;
  return p;
}

// sharded rounds that leave the rank's adders themselves: the TAIL variant of icp_round publishes rows as exchange granules
bool use_tail(const madicp_ctx* ctx, const Launch& l) {
  return ctx->sharded() && ctx->shard_tail && !l.trace && l.qpt == 1 && l.iters <= 250 &&
         madicp::xch_level1(l.batch, l.grid) <= kXchRowsMax * 2 * madicp::kRowGranules;
}

// sharded rounds that join over the peer-mapped mailboxes inside the round kernel (option "shard_p2p", madicp_p2p_attach)
bool use_p2p(const madicp_ctx* ctx, const Launch& l) {
  return ctx->sharded() && ctx->shard_p2p && ctx->p2p_attached && !l.trace && l.qpt == 1 && l.iters <= 250;
}
// ... and whose matched flags travel over the mailboxes too (icp_final): the whole registration is then free of collectives —
// the single-GPU launch sequence, capturable, results out through the side stream.  Every scan's leaves must fit a flag row
// (every rank holds the same moving sets, so every rank decides the same way).
bool p2p_flags_fit(const madicp_ctx* ctx, const int* moving_ids, int n) {
  for (int s = 0; s < n; ++s) {
    auto it = ctx->movings.find(moving_ids[s]);
    if (it == ctx->movings.end() || it->second.L > madicp::kP2pFlagLeaves) return false;
  }
  return true;
}
madicp::PeerBox peer_box(const madicp_ctx* ctx, int scan0, bool on, bool flags_in_box = false) {
  madicp::PeerBox pb{};
  if (!on) return pb;  // (n_ranks = 0: not a mailbox launch)
  for (int q = 0; q < madicp::kMaxRanks; ++q) pb.box[q] = ctx->p2p_peer[q];
  pb.n_ranks = ctx->n_ranks;
  pb.rank = ctx->rank;
  pb.flags_in_box = flags_in_box ? 1 : 0;
  pb.scan0 = scan0;
  pb.spin_ticks = (unsigned long long)std::max(1, ctx->comm_timeout_ms) * 100000ull;  // 100 MHz ticks
  return pb;
}

// The registration counter of the mailbox session: every rank submits the same sequence of sharded registrations, so every
// rank's counter names the same registration — it tags the rows and picks their slot.  Taken once per registration (all scans of
// a batch), after every check that can refuse the submission and right before the Jobs are uploaded.
int next_p2p_epoch(madicp_ctx* ctx, const Launch& l, unsigned* out) {
  *out = 0;
  if (!use_p2p(ctx, l)) return MADICP_OK;
  if (ctx->p2p_broken)
    return fail(MADICP_ERR_COMM, "this mailbox session lost a peer in an earlier registration: the ranks' registration counters may "
                                 "disagree — madicp_p2p_export + madicp_p2p_attach again on every rank (or switch shard_p2p off)");
  *out = ++ctx->p2p_epoch;
  return MADICP_OK;
}

void launch_round(madicp_ctx* ctx, const Launch& l, const Part& p, int round, const double* totals) {
  dim3 g(l.grid, l.batch), b(kBlock);
  const madicp::PeerBox none{};
  if (use_p2p(ctx, l)) {
    const madicp::PeerBox pb = peer_box(ctx, p.scan0, true);
    if (l.queue)
      hipLaunchKernelGGL((icp_round<1, false, false, false, true, true>), g, b, l.lds, p.s, (const Job*)p.jobs, p.jobs, p.partials,
                         (const double*)nullptr, round, l.iters, l.K, l.rpt, p.xch, (double*)nullptr, (unsigned int*)nullptr, pb);
    else
      hipLaunchKernelGGL((icp_round<1, false, false, false, false, true>), g, b, l.lds, p.s, (const Job*)p.jobs, p.jobs, p.partials,
                         (const double*)nullptr, round, l.iters, l.K, l.rpt, p.xch, (double*)nullptr, (unsigned int*)nullptr, pb);
    return;
  }
  if (use_tail(ctx, l)) {
    hipLaunchKernelGGL((icp_round<1, false, false, true>), g, b, l.lds, p.s, (const Job*)p.jobs, p.jobs, p.partials, totals, round,
                       l.iters, l.K, l.rpt, p.xch, p.totals[round & 1], p.tickets, none);
    return;
  }
  void (*kern)(const Job*, Job*, double*, const double*, int, int, int, int, unsigned long long*, double*, unsigned int*,
               const madicp::PeerBox) =
      l.trace ? (l.qpt == 2 ? icp_round<2, true> : icp_round<1, true>) : (l.qpt == 2 ? icp_round<2, false> : icp_round<1, false>);
  if (use_fold(ctx, l)) kern = icp_round<1, false, true>;
  else if (l.queue && !l.trace && l.qpt == 1) kern = icp_round<1, false, false, false, true>;  // units of many passes: queued walks
  hipLaunchKernelGGL(kern, g, b, l.lds, p.s, (const Job*)p.jobs, p.jobs, p.partials, totals, round, l.iters, l.K, l.rpt, p.xch,
                     (double*)nullptr, (unsigned int*)nullptr, none);
}

// (experiment, option "xcd_fold") per-round launches with the XCD-hierarchical join: same admission rules as icp_persist
// except residency — the leaders only wait for workgroups of their own launch, which all run to completion
bool use_fold(const madicp_ctx* ctx, const Launch& l) {
  return ctx->xcd_fold && !ctx->persistent && !ctx->sharded() && !l.trace && l.qpt == 1 && l.iters >= 2 && l.iters <= 250 &&
         (l.grid >> 3) <= kJoinGroups && l.K >= 1 &&
         madicp::xch_granules(l.batch, l.grid) <= kXchRowsMax * 2 * madicp::kRowGranules;
}

// one round of a part, and what a sharded round needs behind it: this rank's share of the adders (a rank that owns no tree
// contributes zeros) -> one all-reduce of [H(21) b(6) n v w] per scan over xGMI: the serial sum of mad_icp.cpp:106-109
int enqueue_round(madicp_ctx* ctx, const Launch& l, const Part& p, int it) {
  if (use_p2p(ctx, l)) {  // the join over the ranks is inside the round kernel's prologue: the single-GPU launch sequence
    launch_round(ctx, l, p, it, nullptr);
    return MADICP_OK;
  }
  launch_round(ctx, l, p, it, (ctx->sharded() && it > 0) ? p.totals[(it - 1) & 1] : nullptr);
  if (ctx->sharded()) {
    if (!use_tail(ctx, l))
      hipLaunchKernelGGL(icp_reduce, dim3(l.batch), dim3(kBlock), 0, p.s, p.partials, l.grid, l.batch, it, p.totals[it & 1]);
    RC_TRY(all_reduce(ctx, p.totals[it & 1], (size_t)l.batch * kAcc, MADICP_REDUCE_SUM_F64, p.s));
  }
  return MADICP_OK;
}
// what closes a part: the matched flags OR-ed over the ranks, then icp_final
int enqueue_close(madicp_ctx* ctx, const Launch& l, const Part& p, const int* moving_ids) {
  const bool p2p = use_p2p(ctx, l);
  const bool box_flags = p2p && p2p_flags_fit(ctx, moving_ids, l.batch);  // (icp_final ORs them over the mailboxes itself)
  if (ctx->sharded() && !box_flags) {
    // a leaf is an inlier if ANY keyframe on ANY rank matched it (mad_icp.cpp:85, pipeline.cpp:197-204): the scans' flag
    // arrays as ONE grouped RCCL operation (one launch for the batch, not one per scan)
    if (ctx->comm && l.batch > 1) NCCL_TRY(ncclGroupStart());
    int rc = MADICP_OK;
    for (int s = 0; s < l.batch && rc == MADICP_OK; ++s) {
      const DevMoving& mv = ctx->movings.at(moving_ids[s]);
      rc = all_reduce(ctx, mv.matched, (size_t)mv.L, MADICP_REDUCE_MAX_U8, p.s);
    }
    if (ctx->comm && l.batch > 1) NCCL_TRY(ncclGroupEnd());
    if (rc != MADICP_OK) return rc;
  }
  // (icp_reduce / icp_final join with kBlock threads, like icp_round: same summation order with and without ranks)
  hipLaunchKernelGGL(icp_final, dim3(l.batch), dim3(kBlock), 0, p.s, p.jobs, p.partials,
                     (ctx->sharded() && !p2p) ? (const double*)p.totals[(l.iters - 1) & 1] : (const double*)nullptr, l.grid, l.batch,
                     use_fold(ctx, l) ? (const unsigned long long*)p.xch : (const unsigned long long*)nullptr,
                     peer_box(ctx, p.scan0, p2p, box_flags));
  HIP_TRY(hipGetLastError());
  return MADICP_OK;
}

// the launch sequence of one (batched) registration; valid both eagerly and under stream capture
int enqueue_rounds(madicp_ctx* ctx, const Launch& l, Job* d_jobs, const std::vector<int>& moving_ids) {
  const int grid = l.grid, batch = l.batch, iters = l.iters;
  if (use_persist(ctx, l)) {
    dim3 g(grid, batch), b(kBlock);
    void (*kern)(const Job*, Job*, unsigned long long*, int, int, int) = l.qpt == 2 ? icp_persist<2> : icp_persist<1>;
    hipLaunchKernelGGL(kern, g, b, l.lds, ctx->stream, (const Job*)d_jobs, d_jobs, ctx->d_xch, iters, l.K, l.rpt);
    hipLaunchKernelGGL(icp_final, dim3(batch), dim3(kBlock), 0, ctx->stream, d_jobs, ctx->d_partials, (const double*)nullptr,
                       grid, batch, (const unsigned long long*)ctx->d_xch, madicp::PeerBox{});
    HIP_TRY(hipGetLastError());
    return MADICP_OK;
  }
  const Part p = whole_part(ctx, d_jobs);
  for (int it = 0; it < iters; ++it) RC_TRY(enqueue_round(ctx, l, p, it));
  return enqueue_close(ctx, l, p, moving_ids.data());
}

// A sharded batch as TWO halves on two streams (option "shard_split"): round r of half B runs while half A's all-reduce
// of round r is in flight, and the other way round — the collective (10-25 us across xGMI for 240 bytes per scan) hides
// under the other half's round kernel instead of standing between two rounds of the same scans.  The collectives are
// enqueued in ONE order on every rank — A(0), B(0), A(1), B(1), ... — so RCCL's per-communicator ordering never
// deadlocks; each half has its own jobs, exchange rows, totals and tickets.  `l` / `p`: the two halves.
int enqueue_rounds_split(madicp_ctx* ctx, const Launch l[2], const Part p[2], const int* moving_ids) {
  HIP_TRY(hipEventRecord(ctx->ev_fork, ctx->stream));  // (jobs, moving sets, trees: everything the halves read is behind this)
  HIP_TRY(hipStreamWaitEvent(ctx->stream2, ctx->ev_fork, 0));
  int rc = MADICP_OK;
  const int iters = l[0].iters;
  for (int it = 0; it < iters && rc == MADICP_OK; ++it)
    for (int h = 0; h < 2 && rc == MADICP_OK; ++h) rc = enqueue_round(ctx, l[h], p[h], it);
  for (int h = 0; h < 2 && rc == MADICP_OK; ++h) rc = enqueue_close(ctx, l[h], p[h], moving_ids + (h ? l[0].batch : 0));
  // the compute stream carries on behind the second half whatever happened (the caller's fetch / next submission)
  hipEventRecord(ctx->ev_join, ctx->stream2);
  hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0);
  return rc;
}

// slot: which device Job array the sequence works on (-1: ctx->d_jobs; >= 0: that stream slot's) — part of the graph key
// queued_behind: the stream is known to be busy with an earlier registration.  A graph launch costs the QUEUE ~8 us more
// than the same kernels launched one by one (markers around the graph: 234 vs 229 us per streamed registration) but costs
// the HOST less, so a registration that would start at once goes as a graph (its first kernel starts sooner: 300 vs 307 us
// submit-to-result) and one that has to wait for its predecessor anyway goes kernel by kernel.
int run_rounds(madicp_ctx* ctx, const Launch& l, Job* d_jobs, int slot, const std::vector<int>& moving_ids, bool queued_behind) {
  // with a communicator the RCCL calls are captured only on request (option "comm_graph"): it could not be
  // exercised on more than one rank where this was developed
  // (a host-staged transport makes a host round trip per round: never capturable)
  // over the peer mailboxes with the flags in them (p2p_free) nothing of the registration is a collective or a host step: it
  // is captured like a single-GPU one (the tags come from Job::p2p_epoch, not from a kernel argument)
  const bool p2p = use_p2p(ctx, l);
  const bool p2p_free = p2p && p2p_flags_fit(ctx, moving_ids.data(), l.batch);
  const bool graph_ok = ctx->use_graph && (p2p_free || (!ctx->host_ar && (!ctx->comm || ctx->comm_graph) && !p2p)) &&
                        !(queued_behind && ctx->eager_when_busy && (!ctx->comm || p2p_free));
  if (!graph_ok) return enqueue_rounds(ctx, l, d_jobs, moving_ids);
  // (with a communicator the matched-flag all-reduce bakes the moving buffer's address: key on the slot only — the
  // batch path never takes the graph route with a communicator unless every scan's buffer is stable, see below)
  const GraphKey key{l.grid, l.batch, l.iters, l.qpt, (ctx->comm ? 1 : 0) + (p2p_free ? 2 : 0), l.lds, l.K, l.rpt, l.trace, slot,
                     (use_persist(ctx, l) ? 1 : (use_fold(ctx, l) ? 2 : 0)) + 4 * l.queue};
  auto it = ctx->graphs.find(key);
  if (it == ctx->graphs.end()) {
    auto instantiate = [&](Job* jobs, const GraphKey& k) -> int {
      hipGraph_t graph = nullptr;
      HIP_TRY(hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
      const int rc = enqueue_rounds(ctx, l, jobs, moving_ids);
      hipError_t e = hipStreamEndCapture(ctx->stream, &graph);
      if (rc != MADICP_OK || e != hipSuccess) {
        if (graph) hipGraphDestroy(graph);
        return rc != MADICP_OK ? rc : fail(MADICP_ERR_DEVICE, std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
      }
      hipGraphExec_t exec = nullptr;
      e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
      hipGraphDestroy(graph);
      if (e != hipSuccess) return fail(MADICP_ERR_DEVICE, std::string("hipGraphInstantiate: ") + hipGetErrorString(e));
      ctx->graphs.emplace(k, exec);
      return MADICP_OK;
    };
    if (slot >= 0 && (!ctx->comm || p2p_free)) {
      // a streamed registration: instantiate this shape for EVERY slot now (a few ms each, once), so that the first
      // lap around the ring costs the same as every later one
      for (int s = 0; s < madicp_ctx::kStreamSlots; ++s) {
        GraphKey k = key;
        k.slot = s;
        if (!ctx->graphs.count(k)) RC_TRY(instantiate(ctx->slots[s].d_job, k));
      }
    } else {
      RC_TRY(instantiate(d_jobs, key));
    }
    it = ctx->graphs.find(key);
  }
  HIP_TRY(hipGraphLaunch(it->second, ctx->stream));
  return MADICP_OK;
}

// the compute stream must see a tree's upload / record build (copy stream) before it reads the tree
int wait_tree(madicp_ctx* ctx, DevTree& t) {
  if (!t.compute_waited) {
    HIP_TRY(hipStreamWaitEvent(ctx->stream, t.ready, 0));
    t.compute_waited = true;
  }
  return MADICP_OK;
}
int wait_moving(madicp_ctx* ctx, DevMoving& m) {
  if (m.on_copy) {
    HIP_TRY(hipStreamWaitEvent(ctx->stream, m.ready, 0));
    m.on_copy = false;
  }
  return MADICP_OK;
}

// grow-only device buffers of a moving set
int reserve_moving(madicp_ctx* ctx, DevMoving& m, int L, hipStream_t user) {
  if (L <= m.cap_L) return MADICP_OK;
  EventRef after;
  if (m.xyzn) RC_TRY(fence_event(ctx, &after));
  pool_free(ctx, m.xyzn, after);
  pool_free(ctx, m.matched, after);
  pool_free(ctx, m.cache_leaf, after);
  pool_free(ctx, m.cache_margin, after);
  m.xyzn = nullptr; m.matched = nullptr; m.cache_leaf = nullptr; m.cache_margin = nullptr;
  m.cache_cap = 0;
  m.cap_L = 0;
  if (ctx->sharded()) {  // a captured sequence with collectives bakes the matched-flag buffer's address
    for (auto& g : ctx->graphs) hipGraphExecDestroy(g.second);
    ctx->graphs.clear();
  }
  const int cap = (L + L / 8 + 255) / 256 * 256;  // head-room: consecutive scans differ by a few per cent
  void* p = nullptr;
  RC_TRY(pool_alloc(ctx, sizeof(double) * 4 * (size_t)cap, user, &p));
  m.xyzn = static_cast<double*>(p);
  RC_TRY(pool_alloc(ctx, (size_t)cap + 16, user, &p));
  m.matched = static_cast<uint8_t*>(p);
  m.cap_L = cap;
  return MADICP_OK;
}
int reserve_cache(madicp_ctx* ctx, DevMoving& m, int K) {
  const size_t need = (size_t)K * (size_t)m.cap_L;
  if (need <= m.cache_cap) return MADICP_OK;
  EventRef after;
  if (m.cache_leaf) RC_TRY(fence_event(ctx, &after));
  pool_free(ctx, m.cache_leaf, after);
  pool_free(ctx, m.cache_margin, after);
  m.cache_leaf = nullptr; m.cache_margin = nullptr;
  m.cache_cap = 0;
  void* p = nullptr;
  RC_TRY(pool_alloc(ctx, sizeof(uint32_t) * need, ctx->stream, &p));
  m.cache_leaf = static_cast<uint32_t*>(p);
  RC_TRY(pool_alloc(ctx, sizeof(float) * need, ctx->stream, &p));  // one threshold per pair (kernels.hip.h, "Gate reuse")
  m.cache_margin = static_cast<float*>(p);
  m.cache_cap = need;
  return MADICP_OK;
}
int reserve_pinned_in(DevMoving& m, int L) {
  const size_t need = 4 * (size_t)L;
  if (need <= m.h_in_cap) return MADICP_OK;
  if (m.h_in) {
    if (m.h_in_read) HIP_TRY(hipEventSynchronize(m.h_in_read));
    HIP_TRY(hipHostFree(m.h_in));
    m.h_in = nullptr;
    m.h_in_cap = 0;
  }
  const size_t cap = need + need / 8 + 768;
  HIP_TRY(hipHostMalloc(&m.h_in, cap * sizeof(double), hipHostMallocDefault));
  m.h_in_cap = cap;
  if (!m.h_in_read) HIP_TRY(hipEventCreateWithFlags(&m.h_in_read, hipEventDisableTiming));
  return MADICP_OK;
}

// leaf means (host, pageable) -> pinned staging in the device layout (x, y, z, |p|) -> one asynchronous copy on `s`.
// |p| = sqrt(x.x) is the reference's `moving->mean_.norm()` (mad_icp.cpp:81), evaluated here on the host
// with the same IEEE operations the device kernel uses (no contraction: -ffp-contract=off covers this file's host
// code too; sqrt is correctly rounded on both sides) — the copy the host makes anyway, 8 bytes wider per leaf, instead
// of a kernel that reads host memory over PCIe next to the registration in flight (measured: that kernel's waves
// kept a registration workgroup off its CU and stretched a round by ~9 us).
int load_moving(madicp_ctx* ctx, DevMoving& m, const double* leaf_means, int L, hipStream_t s) {
  RC_TRY(reserve_moving(ctx, m, L, s));
  RC_TRY(reserve_pinned_in(m, L));
  HIP_TRY(hipEventSynchronize(m.h_in_read));  // (never recorded: returns at once)
  for (int i = 0; i < L; ++i) {
    const double x = leaf_means[3 * i], y = leaf_means[3 * i + 1], z = leaf_means[3 * i + 2];
    double* o = m.h_in + 4 * (size_t)i;
    o[0] = x;
    o[1] = y;
    o[2] = z;
    o[3] = std::sqrt(dotc(x, y, z, x, y, z));  // the kernel-side expression (moving_from_leaves), evaluation order included
  }
  m.L = L;
  HIP_TRY(hipMemcpyAsync(m.xyzn, m.h_in, sizeof(double) * 4 * (size_t)L, hipMemcpyHostToDevice, s));
  HIP_TRY(hipEventRecord(m.h_in_read, s));
  if (!m.ready) HIP_TRY(hipEventCreateWithFlags(&m.ready, hipEventDisableTiming));
  HIP_TRY(hipEventRecord(m.ready, s));
  m.on_copy = (s != ctx->stream);
  if (m.on_copy) m.copy_seq = ++ctx->copy_seq;
  return MADICP_OK;
}

void free_moving(madicp_ctx* ctx, DevMoving& m, const EventRef& after) {
  pool_free(ctx, m.xyzn, after);
  pool_free(ctx, m.matched, after);
  pool_free(ctx, m.cache_leaf, after);
  pool_free(ctx, m.cache_margin, after);
  if (m.h_in) {
    if (m.h_in_read) hipEventSynchronize(m.h_in_read);
    hipHostFree(m.h_in);
  }
  if (m.h_in_read) hipEventDestroy(m.h_in_read);
  if (m.ready) hipEventDestroy(m.ready);
  m = DevMoving{};
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

// A start pose that is not a rigid motion breaks the displacement bound of the correspondence reuse (it assumes
// |R|_2 <= 1 + 1e-7): such a registration re-walks every round, like the reference.
bool is_rigid(const double* X) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double d = 0.0;
      for (int k = 0; k < 3; ++k) d += X[3 * k + i] * X[3 * k + j];
      if (!(std::fabs(d - (i == j ? 1.0 : 0.0)) <= 1e-9)) return false;
    }
  return true;
}

// everything of a Job but the launch geometry
int fill_job(madicp_ctx* ctx, Job& j, DevMoving& mv, const int* tree_ids, int K, const double* X0,
             const madicp_icp_params* params, int n_iters, int flags, bool use_cache) {
  std::memset(&j, 0, offsetof(Job, trees));
  j.moving = mv.xyzn;
  j.matched = mv.matched;
  j.cache_leaf = use_cache ? mv.cache_leaf : nullptr;
  j.cache_margin = use_cache ? mv.cache_margin : nullptr;
  j.L = mv.L;
  j.K = K;
  j.n_iters = n_iters;
  j.iter = 0;
  j.epoch = ++ctx->epoch;  // (24 bits of it reach the granule tags: a tag recurs after 16 M registrations, far beyond
                           // the life of any granule of a geometry in use)
  j.error = 0;
  j.flags = flags | ((ctx->cache_corr && is_rigid(X0)) ? 0 : kFlagNoReuse) | (ctx->match_all ? kFlagMatchAll : 0) |
            (ctx->cache_gate ? 0 : kFlagNoGateReuse);
  j.queue_nodes = ctx->queue_walks;
  std::memcpy(j.X, X0, 12 * sizeof(double));
  std::memcpy(j.Xring[0], X0, 12 * sizeof(double));
  std::memcpy(j.Xring[1], X0, 12 * sizeof(double));
  j.min_ball = params->min_ball;
  j.rho = std::sqrt(params->rho_ker);  // MADicp ctor, mad_icp.cpp:32
  j.b_ratio = params->b_ratio;
  if (K == 0) std::memset(&j.trees[0], 0, sizeof(TreeDesc));
  // The kernel cuts the Job's tree list into eight contiguous pieces, one per XCD (kernels.hip.h).  The caller's list is DEALT
  // over them — position p holds the caller's tree order[p], trees 0, 8, 16, .. first — so that neighbours in the caller's
  // list (keyframes along a trajectory: similar cost for a given scan) land on different XCDs.  Everything the kernels index
  // by tree is internal and follows the Job's positions; the one per-tree OUTPUT (the correspondence trace) goes by `slot`.
  // (deal_trees = 2: boustrophedon — rows of eight alternate their direction, so that the piece that drew the newest keyframe of
  // one row draws the oldest of the next: along a trajectory the newest keyframes are the dear ones, accepted pairs grow by a
  // quarter every four keyframes at BASELINE configs[4], and round-robin hands piece 7 a tree seven places newer than piece 0's
  // in EVERY row)
  int p = 0;
  const int cols = ctx->deal_trees ? 8 : 1;
  for (int c = 0; c < cols; ++c)
    for (int row = 0; row * cols < K; ++row) {
      const int k = row * cols + ((ctx->deal_trees == 2 && (row & 1)) ? cols - 1 - c : c);
      if (k >= K) continue;
      auto tit = ctx->trees.find(tree_ids[k]);
      if (tit == ctx->trees.end()) return fail(MADICP_ERR_INVALID, "unknown tree id");
      RC_TRY(wait_tree(ctx, tit->second));
      j.trees[p] = tit->second.desc;
      j.trees[p].slot = k;
      ++p;
    }
  return MADICP_OK;
}

int check_reg_args(madicp_ctx* ctx, const void* a, const void* b, const void* c, int K, int n_iters) {
  if (!ctx || !a || !b || !c) return fail(MADICP_ERR_INVALID, "null argument");
  // a rank that owns no keyframe tree still has to join the collectives (with zero adders): K == 0 is legal there
  if (K < (ctx->sharded() ? 0 : 1)) return fail(MADICP_ERR_INVALID, "K must be >= 1");
  if (K > MADICP_MAX_TREES) return fail(MADICP_ERR_CAPACITY, "K exceeds MADICP_MAX_TREES");
  if (n_iters < 1) return fail(MADICP_ERR_INVALID, "n_iters must be >= 1");
  return MADICP_OK;
}

// partials for this launch shape: two round parities of join_rows(grid) rows per scan — the rows beyond `grid` are
// zero and stay zero — then two parities of per-workgroup walk hints, + one padding row (the join's 16-byte loads
// read one double past)
size_t partial_doubles_of(int grid, int n_scans) {
  const size_t prows = (size_t)madicp::join_rows(grid);
  return align_up(((size_t)2 * n_scans * prows * kAcc + (size_t)2 * n_scans * grid + kAcc) * sizeof(double)) / sizeof(double);
}
// (`parts` launch shapes side by side: the halves of a split sharded batch each own a region; out_doubles[h] = its size)
int prepare_partials(madicp_ctx* ctx, const Launch* shapes, int parts, size_t* out_doubles) {
  size_t total = 0;
  long long key = parts;
  for (int h = 0; h < parts; ++h) {
    out_doubles[h] = partial_doubles_of(shapes[h].grid, shapes[h].batch);
    total += out_doubles[h];
    key = key * 1000003ll + shapes[h].grid * 64ll + shapes[h].batch;
  }
  RC_TRY(ensure_partials(ctx, total));
  if (ctx->partials_key != key) {
    // a row that is padding in this geometry may have been a real row in the previous one
    HIP_TRY(hipMemsetAsync(ctx->d_partials, 0, total * sizeof(double), ctx->stream));
    ctx->partials_key = key;
  }
  return MADICP_OK;
}

// behind the registration that has just been enqueued: the moving sets it reads may be rewritten from here on
// (madicp_moving_update_async puts the copy stream behind this event)
void mark_moving_used(madicp_ctx* ctx, const std::vector<int>& ids) {
  auto ue = std::make_shared<UseEvent>();
  if (hipEventCreateWithFlags(&ue->ev, hipEventDisableTiming) != hipSuccess) {
    ue->ev = nullptr;
    return;
  }
  if (hipEventRecord(ue->ev, ctx->stream) != hipSuccess) return;
  for (int id : ids) {
    auto it = ctx->movings.find(id);
    if (it != ctx->movings.end()) it->second.last_use = ue;
  }
}

int enqueue_registration(madicp_ctx* ctx, const RegArgs& a) {
  RC_TRY(check_reg_args(ctx, a.moving_ids, a.X0, a.params, a.K, a.n_iters));
  if (a.K > 0 && !a.tree_ids) return fail(MADICP_ERR_INVALID, "null argument");
  if (a.n_scans < 1 || a.n_scans > MADICP_MAX_BATCH) return fail(MADICP_ERR_CAPACITY, "n_scans out of range");
  HIP_TRY(hipSetDevice(ctx->device));
  // next slot of the pinned staging ring; wait until the H2D copy that last used it has executed
  const int slot = ctx->stage_next;
  ctx->stage_next = (slot + 1) % madicp_ctx::kStageSlots;
  HIP_TRY(hipEventSynchronize(ctx->stage_ev[slot]));
  Job* h_jobs = ctx->h_stage[slot];

  int max_L = 0;
  ctx->last_moving.assign(a.moving_ids, a.moving_ids + a.n_scans);
  const bool use_cache = a.n_iters > 1 && !a.time_launches;
  {
    // the batch's sets that were made ready on the copy stream: that stream runs in order, so ONE wait — for the set whose
    // event was recorded last — covers them all (a wait per set was a barrier packet each on the compute stream)
    DevMoving* latest = nullptr;
    for (int s = 0; s < a.n_scans; ++s) {
      auto mit = ctx->movings.find(a.moving_ids[s]);
      if (mit == ctx->movings.end()) return fail(MADICP_ERR_INVALID, "unknown moving id");
      DevMoving& mv = mit->second;
      if (mv.on_copy && (!latest || mv.copy_seq > latest->copy_seq)) latest = &mv;
    }
    if (latest) {
      HIP_TRY(hipStreamWaitEvent(ctx->stream, latest->ready, 0));
      for (int s = 0; s < a.n_scans; ++s) ctx->movings.at(a.moving_ids[s]).on_copy = false;
    }
  }
  for (int s = 0; s < a.n_scans; ++s) {
    auto mit = ctx->movings.find(a.moving_ids[s]);
    if (mit == ctx->movings.end()) return fail(MADICP_ERR_INVALID, "unknown moving id");
    DevMoving& mv = mit->second;
    if (use_cache) RC_TRY(reserve_cache(ctx, mv, std::max(1, a.K)));
    Job& j = h_jobs[s];
    RC_TRY(fill_job(ctx, j, mv, a.tree_ids, a.K, a.X0 + 12 * s, a.params, a.n_iters, a.flags, use_cache));
    j.corr = (s == 0) ? a.d_corr : nullptr;
    j.x_iters = (s == 0) ? a.d_x_iters : nullptr;
    max_L = std::max(max_L, mv.L);
    // flags are cleared on the device before the last round; with a single round that is "now"
    if (a.n_iters == 1 || ctx->match_all) HIP_TRY(hipMemsetAsync(mv.matched, 0, (size_t)mv.L, ctx->stream));
  }
  // a sharded batch of two or more scans goes as two halves on two streams (enqueue_rounds_split): each half is a launch
  // shape of its own — its scans share the chip among themselves, not with the other half's
  // (option value 2: split from two scans on — tests, probes; never over the peer mailboxes: there is no collective to hide)
  const bool split = ctx->sharded() && a.n_scans >= (ctx->shard_split >= 2 ? 2 : 4) && ctx->shard_split && !a.time_launches &&
                     !a.d_corr && !a.d_x_iters && !(ctx->shard_p2p && ctx->p2p_attached);
  const int n_first = split ? a.n_scans / 2 : a.n_scans;
  Launch halves[2];
  Geometry geo = pick_geometry(ctx, max_L, a.K, n_first);
  for (int h = 0; h < (split ? 2 : 1); ++h) {
    const int first = h ? n_first : 0, count = h ? a.n_scans - n_first : n_first;
    if (h) geo = pick_geometry(ctx, max_L, a.K, count);
    halves[h] = Launch{geo.grid, count, a.n_iters, geo.qpt, geo.lds_bytes, a.K, geo.ranges_per_tree, a.d_corr ? 1 : 0, geo.queue};
    for (int s = first; s < first + count; ++s) {
      h_jobs[s].ranges_per_tree = geo.ranges_per_tree;
      if (ctx->interleave == 2 || (ctx->interleave == 1 && geo.queue)) h_jobs[s].flags |= kFlagInterleave;
      h_jobs[s].stage_min_leaves = ctx->stage_min_leaves;
      h_jobs[s].lds_top = geo.lds_bytes ? 1 : 0;
    }
  }
  const Launch launch = halves[0];
  const int grid = launch.grid;
  size_t part_doubles[2] = {0, 0};
  RC_TRY(prepare_partials(ctx, halves, split ? 2 : 1, part_doubles));
  if (!a.time_launches) {
    unsigned reg_epoch = 0;
    RC_TRY(next_p2p_epoch(ctx, launch, &reg_epoch));
    for (int s = 0; s < a.n_scans; ++s) h_jobs[s].p2p_epoch = reg_epoch;
  }
  const size_t job_bytes = offsetof(Job, trees) + sizeof(TreeDesc) * (size_t)std::max(1, a.K);
  // (ONE transfer for the batch — the staged Jobs are contiguous, like the device array: a copy command per scan was 5-8 us of
  // the compute stream each, 50 us of a 2 ms batch of eight; the tail of the last Job's tree list is not sent)
  HIP_TRY(hipMemcpyAsync(ctx->d_jobs, h_jobs, sizeof(Job) * (size_t)(a.n_scans - 1) + job_bytes, hipMemcpyHostToDevice, ctx->stream));
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
    for (int i = 0; i < a.time_launches; ++i) launch_round(ctx, launch, whole_part(ctx, ctx->d_jobs), 0, nullptr);
    HIP_TRY(hipStreamEndCapture(ctx->stream, &graph));
    HIP_TRY(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    HIP_TRY(hipGraphLaunch(exec, ctx->stream));  // warm-up replay
    HIP_TRY(hipEventRecord(ctx->ev_t0, ctx->stream));
    HIP_TRY(hipGraphLaunch(exec, ctx->stream));
    HIP_TRY(hipEventRecord(ctx->ev_t1, ctx->stream));
    // fold the last launch's partials (round parity 0) into job->visits / H / b: icp_final with n_iters = 1 semantics
    hipLaunchKernelGGL(icp_final, dim3(a.n_scans), dim3(kBlock), 0, ctx->stream, ctx->d_jobs, ctx->d_partials,
                       (const double*)nullptr, grid, a.n_scans, (const unsigned long long*)nullptr, madicp::PeerBox{});
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, ctx->ev_t0, ctx->ev_t1));
    hipGraphExecDestroy(exec);
    hipGraphDestroy(graph);
    if (a.out_avg_us) *a.out_avg_us = 1e3 * ms / a.time_launches;
    return MADICP_OK;
  }
  if (split) {
    if (!ctx->stream2) {
      HIP_TRY(hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking));
      HIP_TRY(hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
      HIP_TRY(hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
    }
    Part parts[2] = {whole_part(ctx, ctx->d_jobs), whole_part(ctx, ctx->d_jobs)};
    Part& q = parts[1];
    q.s = ctx->stream2;
    q.jobs = ctx->d_jobs + n_first;
    q.partials = ctx->d_partials + part_doubles[0];
    q.xch = ctx->d_xch + madicp::xch_granules(halves[0].batch, halves[0].grid);
    q.totals[0] += (size_t)n_first * kAcc;
    q.totals[1] += (size_t)n_first * kAcc;
    q.tickets += n_first;
    // the exchange rows are only read by the TAIL variant of the round kernel (option "shard_tail"): only then do the two
    // halves need disjoint regions of them that both fit
    const bool rows_used = use_tail(ctx, halves[0]) || use_tail(ctx, halves[1]);
    if (!rows_used) {
      q.xch = ctx->d_xch;
    } else if (madicp::xch_granules(halves[0].batch, halves[0].grid) + madicp::xch_granules(halves[1].batch, halves[1].grid) >
               kXchRowsMax * 2 * madicp::kRowGranules) {
      return fail(MADICP_ERR_CAPACITY, "sharded batch too large for the exchange rows of its two halves");
    }
    const int rc_split = enqueue_rounds_split(ctx, halves, parts, ctx->last_moving.data());
    if (rc_split == MADICP_OK) mark_moving_used(ctx, ctx->last_moving);
    return rc_split;
  }
  // with a communicator a captured sequence would bake the matched-flag buffers of THESE scans: launch eagerly
  // (over the mailboxes with the flags in them there is no collective: captured like a single-GPU batch)
  const int saved = ctx->use_graph;
  if (ctx->sharded() && !(use_p2p(ctx, launch) && p2p_flags_fit(ctx, ctx->last_moving.data(), a.n_scans))) ctx->use_graph = 0;
  const bool queued_behind = ctx->use_graph && ctx->eager_when_busy && hipStreamQuery(ctx->stream) == hipErrorNotReady;
  const int rc = run_rounds(ctx, launch, ctx->d_jobs, -1, ctx->last_moving, queued_behind);
  ctx->use_graph = saved;
  if (rc == MADICP_OK) mark_moving_used(ctx, ctx->last_moving);
  return rc;
}

}  // namespace

extern "C" {

const char* madicp_last_error(void) { return g_err.c_str(); }
int madicp_abi_version(void) { return 2; }

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
  hipError_t e = hipSuccess;
  if (stream) {
    ctx->stream = static_cast<hipStream_t>(stream);
  } else {
    // MADICP_CU_MASK=i/n (also lo = 0/2, hi = 1/2): the compute stream only gets slice i of n of the device's CU mask, so that n
    // processes sharing ONE GPU can have their round kernels resident side by side — what ranks polling each other's mailboxes
    // need (tests/test_sharded_world8.py, bench.py --gpus N under MADICP_BENCH_BACKEND=gloo: BASELINE configs[3] executed with
    // eight ranks on a one-GPU box; on real hardware every rank has a GPU of its own).  The driver deals the mask's bits over
    // the XCDs, so every slice holds CUs of all eight.  Said once on stderr when it takes effect: a stray setting costs CUs.
    int cu_i = -1, cu_n = 0;
    if (const char* cm = std::getenv("MADICP_CU_MASK")) {
      if (cm[0] == 'l') { cu_i = 0; cu_n = 2; }
      else if (cm[0] == 'h') { cu_i = 1; cu_n = 2; }
      else if (std::sscanf(cm, "%d/%d", &cu_i, &cu_n) != 2 || cu_n < 1 || cu_i < 0 || cu_i >= cu_n || ctx->n_cus / cu_n < 1) {
        std::fprintf(stderr, "madicp: MADICP_CU_MASK=%s ignored (expected lo, hi or i/n with 0 <= i < n <= %d)\n", cm, ctx->n_cus);
        cu_i = -1;
      }
    }
    if (cu_i >= 0) {
      const int words = (ctx->n_cus + 31) / 32;
      std::vector<uint32_t> mask((size_t)words, 0u);
      const int first = (int)((long long)ctx->n_cus * cu_i / cu_n), last = (int)((long long)ctx->n_cus * (cu_i + 1) / cu_n);
      for (int cu = first; cu < last; ++cu) mask[(size_t)cu / 32] |= 1u << (cu % 32);
      e = hipExtStreamCreateWithCUMask(&ctx->stream, (uint32_t)words, mask.data());
      if (e != hipSuccess) {
        (void)hipGetLastError();
        std::fprintf(stderr, "madicp: MADICP_CU_MASK ignored (hipExtStreamCreateWithCUMask: %s)\n", hipGetErrorString(e));
        e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
      } else {
        std::fprintf(stderr, "madicp: MADICP_CU_MASK=%d/%d in effect: the compute stream of this context runs on CUs %d..%d of %d\n", cu_i,
                     cu_n, first, last - 1, ctx->n_cus);
      }
    } else {
      e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    }
    if (e != hipSuccess) {
      delete ctx;
      return fail(MADICP_ERR_DEVICE, std::string("hipStreamCreate: ") + hipGetErrorString(e));
    }
    ctx->own_stream = true;
  }
  e = hipStreamCreateWithFlags(&ctx->copy, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&ctx->pub, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipMalloc(&ctx->d_jobs, sizeof(Job) * MADICP_MAX_BATCH);
  for (int i = 0; i < madicp_ctx::kStageSlots && e == hipSuccess; ++i) {
    e = hipHostMalloc(&ctx->h_stage[i], sizeof(Job) * MADICP_MAX_BATCH, hipHostMallocDefault);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->stage_ev[i], hipEventDisableTiming);
  }
  for (int i = 0; i < 2 && e == hipSuccess; ++i) e = hipEventCreateWithFlags(&ctx->h_tree_ev[i], hipEventDisableTiming);
  if (e == hipSuccess) e = hipHostMalloc(&ctx->h_fetch, sizeof(Job) * MADICP_MAX_BATCH, hipHostMallocDefault);
  if (e == hipSuccess) e = hipMalloc(&ctx->d_totals, sizeof(double) * kAcc * MADICP_MAX_BATCH * 2);
  if (e == hipSuccess) e = hipMalloc(&ctx->d_tickets, sizeof(unsigned int) * MADICP_MAX_BATCH);
  if (e == hipSuccess) e = hipMemset(ctx->d_tickets, 0, sizeof(unsigned int) * MADICP_MAX_BATCH);
  if (e == hipSuccess) e = hipMalloc(&ctx->d_xch, kXchRowsMax * 2 * madicp::kRowGranules * sizeof(unsigned long long));
  if (e == hipSuccess) e = hipMemset(ctx->d_xch, 0, kXchRowsMax * 2 * madicp::kRowGranules * sizeof(unsigned long long));
  for (int i = 0; i < madicp_ctx::kStreamSlots && e == hipSuccess; ++i) {
    StreamSlot& sl = ctx->slots[i];
    e = hipMalloc(&sl.d_job, sizeof(Job));
    if (e == hipSuccess) e = hipHostMalloc(&sl.h_job, sizeof(Job), hipHostMallocDefault);
    if (e == hipSuccess) e = hipHostMalloc(&sl.h_out, sizeof(HostResult), hipHostMallocDefault);
    if (e == hipSuccess) e = hipMalloc(&sl.d_outbox, sizeof(Outbox));
    if (e == hipSuccess) e = hipMemset(sl.d_outbox, 0, sizeof(Outbox));
    if (e == hipSuccess) e = hipEventCreateWithFlags(&sl.ev_up, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&sl.ev_done, hipEventDisableTiming);
  }
  if (e != hipSuccess) {
    madicp_ctx_destroy(ctx);
    return fail(MADICP_ERR_DEVICE, std::string("context allocation: ") + hipGetErrorString(e));
  }
  *out = ctx;
  return MADICP_OK;
}

int madicp_p2p_detach(madicp_ctx* ctx);
int madicp_ctx_destroy(madicp_ctx* ctx) {
  if (!ctx) return MADICP_OK;
  hipSetDevice(ctx->device);
  if (ctx->stream) hipStreamSynchronize(ctx->stream);
  if (ctx->copy) hipStreamSynchronize(ctx->copy);
  if (ctx->build) hipStreamSynchronize(ctx->build);
  if (ctx->pub) hipStreamSynchronize(ctx->pub);
  if (ctx->p2p_attached) madicp_p2p_detach(ctx);
  if (ctx->p2p_box) hipFree(ctx->p2p_box);
  if (ctx->ev_build_gate) hipEventDestroy(ctx->ev_build_gate);
  if (ctx->comm) ncclCommDestroy(ctx->comm);
  if (ctx->h_comm) hipHostFree(ctx->h_comm);
  for (auto& g : ctx->graphs) hipGraphExecDestroy(g.second);
  front_destroy(ctx);
  for (auto& t : ctx->trees)
    if (t.second.ready) hipEventDestroy(t.second.ready);
  for (auto& m : ctx->movings) {
    if (m.second.h_in) hipHostFree(m.second.h_in);
    if (m.second.h_in_read) hipEventDestroy(m.second.h_in_read);
    if (m.second.ready) hipEventDestroy(m.second.ready);
  }
  ctx->pool.clear();  // (drops the event holders)
  for (auto& a : ctx->alloc_bytes) hipFree(a.first);  // every pooled or live device buffer
  if (ctx->ev_t0) hipEventDestroy(ctx->ev_t0);
  if (ctx->ev_t1) hipEventDestroy(ctx->ev_t1);
  if (ctx->d_jobs) hipFree(ctx->d_jobs);
  for (int i = 0; i < madicp_ctx::kStageSlots; ++i) {
    if (ctx->h_stage[i]) hipHostFree(ctx->h_stage[i]);
    if (ctx->stage_ev[i]) hipEventDestroy(ctx->stage_ev[i]);
  }
  for (int i = 0; i < 2; ++i) {
    if (ctx->h_tree[i]) hipHostFree(ctx->h_tree[i]);
    if (ctx->h_tree_ev[i]) hipEventDestroy(ctx->h_tree_ev[i]);
  }
  for (int i = 0; i < madicp_ctx::kStreamSlots; ++i) {
    StreamSlot& sl = ctx->slots[i];
    if (sl.d_job) hipFree(sl.d_job);
    if (sl.h_job) hipHostFree(sl.h_job);
    if (sl.h_out) hipHostFree(sl.h_out);
    if (sl.d_outbox) hipFree(sl.d_outbox);
    if (sl.h_matched) hipHostFree(sl.h_matched);
    if (sl.ev_up) hipEventDestroy(sl.ev_up);
    if (sl.ev_done) hipEventDestroy(sl.ev_done);
  }
  if (ctx->h_fetch) hipHostFree(ctx->h_fetch);
  for (void* p : ctx->d_f32)
    if (p) hipFree(p);
  for (auto* hp : ctx->h_pub)
    if (hp) hipHostFree(hp);
  if (ctx->d_partials) hipFree(ctx->d_partials);
  if (ctx->d_totals) hipFree(ctx->d_totals);
  if (ctx->d_tickets) hipFree(ctx->d_tickets);
  if (ctx->stream2) {
    hipStreamSynchronize(ctx->stream2);
    hipStreamDestroy(ctx->stream2);
  }
  if (ctx->pub) hipStreamDestroy(ctx->pub);
  if (ctx->ev_fork) hipEventDestroy(ctx->ev_fork);
  if (ctx->ev_join) hipEventDestroy(ctx->ev_join);
  if (ctx->d_xch) hipFree(ctx->d_xch);
  if (ctx->copy) hipStreamDestroy(ctx->copy);
  if (ctx->build) hipStreamDestroy(ctx->build);
  if (ctx->own_stream && ctx->stream) hipStreamDestroy(ctx->stream);
  delete ctx;
  return MADICP_OK;
}

int madicp_ctx_synchronize(madicp_ctx* ctx) {
  if (!ctx) return fail(MADICP_ERR_INVALID, "ctx is null");
  HIP_TRY(hipStreamSynchronize(ctx->copy));
  if (ctx->build) HIP_TRY(hipStreamSynchronize(ctx->build));
  RC_TRY(bounded_sync(ctx, ctx->stream));
  if (ctx->pub) HIP_TRY(hipStreamSynchronize(ctx->pub));  // (behind the compute stream: its kernels wait for icp_final's tags)
  return MADICP_OK;
}

int madicp_ctx_set_option(madicp_ctx* ctx, const char* key, int64_t value) {
  if (!ctx || !key) return fail(MADICP_ERR_INVALID, "null argument");
  const std::string k(key);
  if (k == "grid_blocks_per_cu") {
    if (value < 1 || value > 4) return fail(MADICP_ERR_INVALID, "grid_blocks_per_cu must be in 1..4");
    ctx->blocks_per_cu = (int)value;
  } else if (k == "publish_side") {
    ctx->publish_side = value ? 1 : 0;
  } else if (k == "deal_trees") {
    if (value < 0 || value > 2) return fail(MADICP_ERR_INVALID, "deal_trees is 0 (as listed), 1 (round-robin over the XCD pieces) or 2 (alternating rows)");
    ctx->deal_trees = (int)value;
  } else if (k == "units_per_workgroup") {
    if (value < 1 || value > 64) return fail(MADICP_ERR_INVALID, "units_per_workgroup must be in 1..64");
    ctx->units_per_wg = (int)value;
  } else if (k == "use_graph") {
    ctx->use_graph = value ? 1 : 0;
  } else if (k == "comm_graph") {
    ctx->comm_graph = value ? 1 : 0;
  } else if (k == "cache_correspondences") {
    ctx->cache_corr = value ? 1 : 0;
  } else if (k == "cache_gate") {
    ctx->cache_gate = value ? 1 : 0;
  } else if (k == "deep_min_leaves") {
    if (value < 64 || value > (1 << 24)) return fail(MADICP_ERR_INVALID, "deep_min_leaves must be in 64 .. 2^24");
    ctx->deep_min_leaves = (int)value;
  } else if (k == "interleave_ranges") {
    if (value < 0 || value > 2) return fail(MADICP_ERR_INVALID, "interleave_ranges is 0 (never), 1 (batches that share the chip) or 2 (always)");
    ctx->interleave = (int)value;
  } else if (k == "leaf_major") {
    if (value < 0 || value > (1 << 20)) return fail(MADICP_ERR_INVALID, "leaf_major must be 0 (never) or a node count per pass");
    ctx->queue_walks = (int)value;
  } else if (k == "lds_stage_min_leaves") {
    if (value < 0) return fail(MADICP_ERR_INVALID, "lds_stage_min_leaves must be >= 0");
    ctx->stage_min_leaves = (int)std::min<int64_t>(value, 1 << 30);
  } else if (k == "eager_when_busy") {
    ctx->eager_when_busy = value ? 1 : 0;
  } else if (k == "seq_completion") {
    ctx->seq_completion = value ? 1 : 0;
  } else if (k == "host_feed_wait") {
    ctx->host_feed_wait = value ? 1 : 0;
  } else if (k == "xcd_fold") {
    ctx->xcd_fold = value ? 1 : 0;
  } else if (k == "debug_collective_us") {
    ctx->debug_collective_us = (int)std::max<int64_t>(0, std::min<int64_t>(value, 1000));
  } else if (k == "shard_tail") {
    ctx->shard_tail = value ? 1 : 0;
  } else if (k == "build_after_registration") {
    ctx->build_after_registration = value ? 1 : 0;
  } else if (k == "shard_p2p") {
    ctx->shard_p2p = value ? 1 : 0;
  } else if (k == "shard_split") {
    ctx->shard_split = value < 0 ? 0 : (value > 2 ? 2 : (int)value);
  } else if (k == "match_all_rounds") {
    ctx->match_all = value ? 1 : 0;
  } else if (k == "persistent") {
    ctx->persistent = value ? 1 : 0;
  } else if (k == "wait_mode") {
    if (value < 0 || value > 2) return fail(MADICP_ERR_INVALID, "wait_mode must be 0 (spin), 1 (yield) or 2 (sleep)");
    ctx->wait_mode = (int)value;
  } else if (k == "wait_timeout_ms") {
    if (value < 0) return fail(MADICP_ERR_INVALID, "wait_timeout_ms must be >= 0");
    ctx->wait_timeout_ms = (int)std::min<int64_t>(value, 1 << 30);
  } else if (k == "comm_timeout_ms") {
    if (value < 1) return fail(MADICP_ERR_INVALID, "comm_timeout_ms must be >= 1");
    ctx->comm_timeout_ms = (int)std::min<int64_t>(value, 1 << 30);
    if (ctx->p2p_attached) {  // (captured mailbox launches carry the bound of their polls as a kernel argument)
      HIP_TRY(hipStreamSynchronize(ctx->stream));
      for (auto& g : ctx->graphs) hipGraphExecDestroy(g.second);
      ctx->graphs.clear();
    }
  } else if (k == "p2p_allow_coarse") {
    ctx->p2p_allow_coarse = value ? 1 : 0;
  } else if (k == "upload_f32") {
    ctx->upload_f32 = value ? 1 : 0;
  } else if (k == "nn_lds_top") {
    ctx->nn_lds_top = value ? 1 : 0;
  } else if (k == "queries_per_lane") {
    if (value != 0 && value != 1 && value != 2) return fail(MADICP_ERR_INVALID, "queries_per_lane must be 0 (default), 1 or 2");
    ctx->qpt_override = (int)value;
  } else {
    return fail(MADICP_ERR_INVALID, "unknown option: " + k);
  }
  return MADICP_OK;
}

int madicp_ctx_get_option(madicp_ctx* ctx, const char* key, int64_t* out_value) {
  if (!ctx || !key || !out_value) return fail(MADICP_ERR_INVALID, "null argument");
  const std::string k(key);
  int64_t v = 0;
  if (k == "grid_blocks_per_cu") v = ctx->blocks_per_cu;
  else if (k == "publish_side") v = ctx->publish_side;
  else if (k == "deal_trees") v = ctx->deal_trees;
  else if (k == "units_per_workgroup") v = ctx->units_per_wg;
  else if (k == "use_graph") v = ctx->use_graph;
  else if (k == "comm_graph") v = ctx->comm_graph;
  else if (k == "cache_correspondences") v = ctx->cache_corr;
  else if (k == "cache_gate") v = ctx->cache_gate;
  else if (k == "interleave_ranges") v = ctx->interleave;
  else if (k == "deep_min_leaves") v = ctx->deep_min_leaves;
  else if (k == "leaf_major") v = ctx->queue_walks;
  else if (k == "lds_stage_min_leaves") v = ctx->stage_min_leaves;
  else if (k == "eager_when_busy") v = ctx->eager_when_busy;
  else if (k == "seq_completion") v = ctx->seq_completion;
  else if (k == "host_feed_wait") v = ctx->host_feed_wait;
  else if (k == "xcd_fold") v = ctx->xcd_fold;
  else if (k == "debug_collective_us") v = ctx->debug_collective_us;
  else if (k == "shard_tail") v = ctx->shard_tail;
  else if (k == "build_after_registration") v = ctx->build_after_registration;
  else if (k == "shard_p2p") v = ctx->shard_p2p;
  else if (k == "shard_split") v = ctx->shard_split;
  else if (k == "match_all_rounds") v = ctx->match_all;
  else if (k == "persistent") v = ctx->persistent;
  else if (k == "wait_mode") v = ctx->wait_mode;
  else if (k == "wait_timeout_ms") v = ctx->wait_timeout_ms;
  else if (k == "comm_timeout_ms") v = ctx->comm_timeout_ms;
  else if (k == "p2p_allow_coarse") v = ctx->p2p_allow_coarse;
  else if (k == "upload_f32") v = ctx->upload_f32;
  else if (k == "p2p_fine_grained") v = (ctx->p2p_box && ctx->p2p_fine) ? 1 : 0;  // (read-only: what madicp_p2p_export obtained)
  else if (k == "nn_lds_top") v = ctx->nn_lds_top;
  else if (k == "queries_per_lane") v = ctx->qpt_override;
  else return fail(MADICP_ERR_INVALID, "unknown option: " + k);
  *out_value = v;
  return MADICP_OK;
}

// ---- trees ------------------------------------------------------------------------------------------
namespace {

// Structure check of a caller-supplied node array (it arrives through a public C ABI and is then walked by kernels
// that trust its links): DFS preorder with every internal node's children inside the node's own extent, n leaves with
// unique ordinals 0..n_leaves-1.  O(n), one pass with an explicit stack.  Also returns rho2 = max |m - o|_2 over the
// internal nodes with finite means (o = mean of node 0) — what the screening bound needs.
int validate_nodes(const madicp_node* nodes, int32_t n_nodes, int32_t n_leaves, double* out_rho2) {
  std::vector<uint8_t> seen((size_t)n_leaves, 0);
  std::vector<std::pair<int32_t, int32_t>> stack;  // [begin, end) extents still to parse
  stack.emplace_back(0, n_nodes);
  const double o0 = nodes[0].mean[0], o1 = nodes[0].mean[1], o2 = nodes[0].mean[2];
  double rho2 = 0.0;
  int32_t leaves = 0;
  while (!stack.empty()) {
    const auto [b, e] = stack.back();
    stack.pop_back();
    const madicp_node& nd = nodes[b];
    if (nd.right == 0) {
      if (e - b != 1) return fail(MADICP_ERR_INVALID, "node array: a leaf with a sub-tree behind it");
      if (nd.leaf_id < 0 || nd.leaf_id >= n_leaves || seen[nd.leaf_id])
        return fail(MADICP_ERR_INVALID, "node array: leaf_id out of range or repeated");
      seen[nd.leaf_id] = 1;
      ++leaves;
    } else {
      if (nd.right < 2 || (int64_t)b + nd.right >= e) return fail(MADICP_ERR_INVALID, "node array: child link out of range");
      const double e0 = nd.mean[0] - o0, e1 = nd.mean[1] - o1, e2 = nd.mean[2] - o2;
      const double r = std::sqrt((e0 * e0 + e1 * e1) + e2 * e2);
      if (std::isfinite(r)) rho2 = std::max(rho2, r);
      stack.emplace_back(b + nd.right, e);
      stack.emplace_back(b + 1, b + nd.right);
    }
  }
  if (leaves != n_leaves) return fail(MADICP_ERR_INVALID, "node array: leaf count mismatch");
  *out_rho2 = rho2;
  return MADICP_OK;
}

// Breadth-first layout of the internal nodes of the first kTopLevels levels (structure only; depends on the tree's
// topology, so it is made once at upload).  See "LDS-staged top levels" in kernels.hip.h for the link word.
void layout_top(const madicp_node* nodes, std::vector<int>& dfs, std::vector<unsigned int>& link, std::vector<int4>& exits) {
  dfs.clear();
  link.clear();
  exits.clear();
  if (nodes[0].right == 0) return;
  std::vector<int> level{0}, first_leaf{0};  // per entry: its level, the leaf ordinal of its left-most leaf
  dfs.push_back(0);
  for (size_t e = 0; e < dfs.size(); ++e) {
    const int i = dfs[e];
    const int l = i + 1, r = i + nodes[i].right;
    const int left_leaves = nodes[i].right >> 1;  // (a left sub-tree of s nodes holds (s + 1) / 2 leaves, s = right - 1)
    const bool l_leaf = nodes[l].right == 0, r_leaf = nodes[r].right == 0;
    const bool deeper = level[e] + 1 < kTopLevels;
    const bool l_in = !l_leaf && deeper && dfs.size() + 1 <= (size_t)kTopMax - 1;
    const bool r_in = !r_leaf && deeper && dfs.size() + (l_in ? 2 : 1) <= (size_t)kTopMax - 1;
    int le = -1, re = -1;
    if (l_in) { le = (int)dfs.size(); dfs.push_back(l); level.push_back(level[e] + 1); first_leaf.push_back(first_leaf[e]); }
    if (r_in) { re = (int)dfs.size(); dfs.push_back(r); level.push_back(level[e] + 1); first_leaf.push_back(first_leaf[e] + left_leaves); }
    link.push_back(top_link_word(le, re, l_leaf, r_leaf));
    exits.push_back(make_int4(l, left_leaves, first_leaf[e], level[e]));
  }
}

void set_desc(DevTree& t, const double origin[3]) {
  t.desc.nodes = t.nodes;
  t.desc.cnodes = t.cnodes;
  t.desc.leaves = t.leaves;
  t.desc.top = t.top;
  t.desc.top_exit = t.top_exit;
  t.desc.top_dfs = t.top_dfs;
  t.desc.n_top = t.n_top;
  std::memcpy(t.desc.origin, origin, 3 * sizeof(double));
  t.desc.rho = 1.7320508075688774 * t.rho2 * (1.0 + 1e-12);  // |v|_1 <= sqrt(3) |v|_2, rounded up
}

// (re)build the 16-byte screening records, the dense leaf records and the LDS top records on stream `s`
int compact_tree(DevTree& t, hipStream_t s) {
  const double* o = t.desc.origin;
  hipLaunchKernelGGL(tree_compact, dim3((t.n_nodes + 255) / 256), dim3(256), 0, s, (const madicp_node*)t.nodes, t.cnodes,
                     t.leaves, t.n_nodes, o[0], o[1], o[2]);
  if (t.n_top > 0)
    hipLaunchKernelGGL(tree_compact_top, dim3((t.n_top + 255) / 256), dim3(256), 0, s, (const madicp_node*)t.nodes, t.top,
                       (const int*)t.top_dfs, (const unsigned int*)t.top_link, t.n_top, o[0], o[1], o[2]);
  HIP_TRY(hipGetLastError());
  return MADICP_OK;
}

void release_tree(madicp_ctx* ctx, DevTree& t, const EventRef& after) {
  pool_free(ctx, t.block, after);
  if (t.ready) hipEventDestroy(t.ready);
  t = DevTree{};
}

}  // namespace

namespace {
int tree_upload_impl(madicp_ctx* ctx, const madicp_node* nodes, int32_t n_nodes, int32_t n_leaves, bool trusted, double rho2,
                     int* out_tree_id);
}
int madicp_tree_upload(madicp_ctx* ctx, const madicp_node* nodes, int32_t n_nodes, int32_t n_leaves, int* out_tree_id) {
  return tree_upload_impl(ctx, nodes, n_nodes, n_leaves, false, 0.0, out_tree_id);
}
int madicp_tree_upload_trusted(madicp_ctx* ctx, const madicp_node* nodes, int32_t n_nodes, int32_t n_leaves, double rho2,
                               int* out_tree_id) {
  if (!(rho2 >= 0.0) || !std::isfinite(rho2)) return fail(MADICP_ERR_INVALID, "rho2 must be finite and >= 0");
  return tree_upload_impl(ctx, nodes, n_nodes, n_leaves, true, rho2, out_tree_id);
}
namespace {
int tree_upload_impl(madicp_ctx* ctx, const madicp_node* nodes, int32_t n_nodes, int32_t n_leaves, bool trusted, double rho2,
                     int* out_tree_id) {
  if (!ctx || !nodes || !out_tree_id) return fail(MADICP_ERR_INVALID, "null argument");
  if (n_nodes < 1 || n_leaves < 1 || n_nodes != 2 * n_leaves - 1)
    return fail(MADICP_ERR_INVALID, "a MAD-tree has n_nodes == 2*n_leaves-1 >= 1");
  HIP_TRY(hipSetDevice(ctx->device));
  DevTree t;
  t.n_nodes = n_nodes;
  t.n_leaves = n_leaves;
  if (trusted)
    t.rho2 = rho2;
  else
    RC_TRY(validate_nodes(nodes, n_nodes, n_leaves, &t.rho2));
  std::vector<int> top_dfs;
  std::vector<unsigned int> top_link;
  std::vector<int4> top_exit;
  layout_top(nodes, top_dfs, top_link, top_exit);
  t.n_top = static_cast<int32_t>(top_dfs.size());
  // one device block: what is uploaded first (contiguous, one copy), what the device derives from it behind
  const size_t nt = (size_t)t.n_top;
  const size_t off_nodes = 0;
  const size_t off_exit = align_up(off_nodes + sizeof(madicp_node) * (size_t)n_nodes);
  const size_t off_dfs = align_up(off_exit + sizeof(int4) * nt);
  const size_t off_link = align_up(off_dfs + sizeof(int) * nt);
  const size_t up_bytes = align_up(off_link + sizeof(unsigned int) * nt);
  const size_t off_cnodes = up_bytes;
  const size_t off_leaves = align_up(off_cnodes + sizeof(CNode) * (size_t)n_nodes);
  const size_t off_top = align_up(off_leaves + sizeof(LeafRec) * (size_t)n_leaves);
  const size_t total = align_up(off_top + sizeof(CNode) * std::max<size_t>(nt, 1));
  // pinned staging (grow-only, two buffers so that the next upload can be staged while this one is in flight)
  const int hb = ctx->h_tree_next;
  ctx->h_tree_next ^= 1;
  HIP_TRY(hipEventSynchronize(ctx->h_tree_ev[hb]));
  if (ctx->h_tree_cap[hb] < up_bytes) {
    if (ctx->h_tree[hb]) HIP_TRY(hipHostFree(ctx->h_tree[hb]));
    ctx->h_tree[hb] = nullptr;
    ctx->h_tree_cap[hb] = 0;
    const size_t cap = up_bytes + up_bytes / 4;
    HIP_TRY(hipHostMalloc(&ctx->h_tree[hb], cap, hipHostMallocDefault));
    ctx->h_tree_cap[hb] = cap;
  }
  char* hs = ctx->h_tree[hb];
  std::memcpy(hs + off_nodes, nodes, sizeof(madicp_node) * (size_t)n_nodes);
  if (nt) {
    std::memcpy(hs + off_exit, top_exit.data(), sizeof(int4) * nt);
    std::memcpy(hs + off_dfs, top_dfs.data(), sizeof(int) * nt);
    std::memcpy(hs + off_link, top_link.data(), sizeof(unsigned int) * nt);
  }
  void* blk = nullptr;
  RC_TRY(pool_alloc(ctx, total, ctx->copy, &blk));
  t.block = static_cast<char*>(blk);
  t.nodes = reinterpret_cast<madicp_node*>(t.block + off_nodes);
  t.top_exit = nt ? reinterpret_cast<int4*>(t.block + off_exit) : nullptr;
  t.top_dfs = nt ? reinterpret_cast<int*>(t.block + off_dfs) : nullptr;
  t.top_link = nt ? reinterpret_cast<unsigned int*>(t.block + off_link) : nullptr;
  t.cnodes = reinterpret_cast<CNode*>(t.block + off_cnodes);
  t.leaves = reinterpret_cast<LeafRec*>(t.block + off_leaves);
  t.top = nt ? reinterpret_cast<CNode*>(t.block + off_top) : nullptr;
  set_desc(t, nodes[0].mean);
  hipError_t e = hipMemcpyAsync(t.block, hs, up_bytes, hipMemcpyHostToDevice, ctx->copy);
  if (e == hipSuccess) e = hipEventRecord(ctx->h_tree_ev[hb], ctx->copy);
  int rc = MADICP_OK;
  if (e == hipSuccess) rc = compact_tree(t, ctx->copy);
  if (e == hipSuccess && rc == MADICP_OK) e = hipEventCreateWithFlags(&t.ready, hipEventDisableTiming);
  if (e == hipSuccess && rc == MADICP_OK) e = hipEventRecord(t.ready, ctx->copy);
  if (e != hipSuccess || rc != MADICP_OK) {
    hipStreamSynchronize(ctx->copy);
    release_tree(ctx, t, nullptr);
    return rc != MADICP_OK ? rc : fail(MADICP_ERR_DEVICE, std::string("tree upload: ") + hipGetErrorString(e));
  }
  const int id = ctx->next_id++;
  ctx->trees[id] = t;
  *out_tree_id = id;
  return MADICP_OK;
}
}  // namespace

int madicp_tree_release(madicp_ctx* ctx, int tree_id) {
  if (!ctx) return fail(MADICP_ERR_INVALID, "ctx is null");
  auto it = ctx->trees.find(tree_id);
  if (it == ctx->trees.end()) return fail(MADICP_ERR_INVALID, "unknown tree id");
  HIP_TRY(hipSetDevice(ctx->device));
  // no synchronisation: the block goes back to the pool behind an event that covers everything enqueued so far
  EventRef after;
  RC_TRY(fence_event(ctx, &after));
  release_tree(ctx, it->second, after);
  ctx->trees.erase(it);
  return MADICP_OK;
}

int madicp_tree_download(madicp_ctx* ctx, int tree_id, madicp_node* out_nodes, int32_t n_nodes) {
  if (!ctx || !out_nodes) return fail(MADICP_ERR_INVALID, "null argument");
  auto it = ctx->trees.find(tree_id);
  if (it == ctx->trees.end()) return fail(MADICP_ERR_INVALID, "unknown tree id");
  if (n_nodes != it->second.n_nodes) return fail(MADICP_ERR_INVALID, "n_nodes mismatch");
  HIP_TRY(hipSetDevice(ctx->device));
  RC_TRY(wait_tree(ctx, it->second));
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
  DevTree& tr = it->second;
  RC_TRY(wait_tree(ctx, tr));
  // stream-ordered on the compute stream (behind every registration that used the tree so far), no host sync: the
  // pose travels as a kernel argument, the new origin is computed here with the kernel's own operation order
  Pose12 P;
  std::memcpy(P.v, R, 9 * sizeof(double));
  std::memcpy(P.v + 9, t, 3 * sizeof(double));
  hipLaunchKernelGGL(tree_transform, dim3((tr.n_nodes + 255) / 256), dim3(256), 0, ctx->stream, tr.nodes, tr.n_nodes, P);
  HIP_TRY(hipGetLastError());
  double o[3];
  const double* m = tr.desc.origin;
  for (int r = 0; r < 3; ++r) o[r] = (R[3 * r] * m[0] + (R[3 * r + 1] * m[1] + R[3 * r + 2] * m[2])) + t[r];
  tr.rho2 *= (1.0 + 1e-12);  // rotation invariant up to rounding
  set_desc(tr, o);
  return compact_tree(tr, ctx->stream);  // screening records follow the nodes
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
  RC_TRY(wait_tree(ctx, it->second));
  if (n >= 16384 && it->second.desc.n_top > 0 && ctx->nn_lds_top) {
    // searchCloud-sized batch: one 1024-thread workgroup per CU stages the tree's top levels in LDS and walks them there
    const long long blocks = std::min<long long>((n + 1023) / 1024, (long long)ctx->n_cus);
    hipLaunchKernelGGL(nn_descend_top, dim3((unsigned)blocks), dim3(1024), kTopLdsBytes, ctx->stream, it->second.desc, d_queries,
                       (long long)n, d_out_leaf_id, d_out_node, d_out_dist, d_out_depth);
  } else {
    const long long blocks = std::min<long long>((n + 255) / 256, (long long)ctx->n_cus * 8);
    hipLaunchKernelGGL(nn_descend, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, it->second.desc, d_queries,
                       (long long)n, d_out_leaf_id, d_out_node, d_out_dist, d_out_depth);
  }
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
  // one pooled device block: queries | leaf | node | dist | depth
  const size_t nq = (size_t)n;
  const size_t off_leaf = align_up(sizeof(double) * 3 * nq);
  const size_t off_node = align_up(off_leaf + sizeof(uint32_t) * nq);
  const size_t off_dist = align_up(off_node + sizeof(uint32_t) * nq);
  const size_t off_depth = align_up(off_dist + sizeof(double) * nq);
  const size_t total = align_up(off_depth + sizeof(int32_t) * nq);
  void* blk = nullptr;
  RC_TRY(pool_alloc(ctx, total, ctx->stream, &blk));
  char* d = static_cast<char*>(blk);
  int rc = MADICP_OK;
  hipError_t e = hipMemcpyAsync(d, queries, sizeof(double) * 3 * nq, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess)
    rc = madicp_nn_search_device_enqueue(ctx, tree_id, reinterpret_cast<const double*>(d), n,
                                         out_leaf_id ? reinterpret_cast<uint32_t*>(d + off_leaf) : nullptr,
                                         out_node ? reinterpret_cast<uint32_t*>(d + off_node) : nullptr,
                                         out_dist ? reinterpret_cast<double*>(d + off_dist) : nullptr,
                                         out_depth ? reinterpret_cast<int32_t*>(d + off_depth) : nullptr);
  if (rc == MADICP_OK) {
    if (e == hipSuccess && out_leaf_id) e = hipMemcpyAsync(out_leaf_id, d + off_leaf, sizeof(uint32_t) * nq, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess && out_node) e = hipMemcpyAsync(out_node, d + off_node, sizeof(uint32_t) * nq, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess && out_dist) e = hipMemcpyAsync(out_dist, d + off_dist, sizeof(double) * nq, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess && out_depth) e = hipMemcpyAsync(out_depth, d + off_depth, sizeof(int32_t) * nq, hipMemcpyDeviceToHost, ctx->stream);
  }
  hipError_t e2 = hipStreamSynchronize(ctx->stream);
  pool_free(ctx, blk, nullptr);  // (stream drained: nothing can still touch it)
  if (rc != MADICP_OK) return rc;
  if (e != hipSuccess || e2 != hipSuccess)
    return fail(MADICP_ERR_DEVICE, std::string("nn_search: ") + hipGetErrorString(e != hipSuccess ? e : e2));
  return MADICP_OK;
}

// ---- moving side ------------------------------------------------------------------------------------
int madicp_moving_upload(madicp_ctx* ctx, const double* leaf_means, int32_t L, int* out_moving_id) {
  if (!ctx || !leaf_means || !out_moving_id) return fail(MADICP_ERR_INVALID, "null argument");
  if (L < 1) return fail(MADICP_ERR_INVALID, "L must be >= 1");
  HIP_TRY(hipSetDevice(ctx->device));
  DevMoving m;
  const int rc = load_moving(ctx, m, leaf_means, L, ctx->stream);
  if (rc != MADICP_OK) {
    free_moving(ctx, m, nullptr);
    return rc;
  }
  const int id = ctx->next_id++;
  ctx->movings[id] = m;
  *out_moving_id = id;
  return MADICP_OK;
}

int madicp_moving_update(madicp_ctx* ctx, int moving_id, const double* leaf_means, int32_t L) {
  if (!ctx || !leaf_means) return fail(MADICP_ERR_INVALID, "null argument");
  if (L < 1) return fail(MADICP_ERR_INVALID, "L must be >= 1");
  auto it = ctx->movings.find(moving_id);
  if (it == ctx->movings.end()) return fail(MADICP_ERR_INVALID, "unknown moving id");
  HIP_TRY(hipSetDevice(ctx->device));
  return load_moving(ctx, it->second, leaf_means, L, ctx->stream);
}

// The same on the COPY stream: the transfer runs beside whatever the compute stream is doing — a batch that reads OTHER
// moving sets — instead of behind it; the copy stream first waits for the last registration that read THIS set, the next
// registration that reads it waits for the transfer.  With two sets of moving ids used alternately the upload of batch i + 1
// hides under batch i (bench.py's configs[4] loop; madicp_icp_publish_enqueue / _collect for the results).
int madicp_moving_update_async(madicp_ctx* ctx, int moving_id, const double* leaf_means, int32_t L) {
  if (!ctx || !leaf_means) return fail(MADICP_ERR_INVALID, "null argument");
  if (L < 1) return fail(MADICP_ERR_INVALID, "L must be >= 1");
  auto it = ctx->movings.find(moving_id);
  if (it == ctx->movings.end()) return fail(MADICP_ERR_INVALID, "unknown moving id");
  HIP_TRY(hipSetDevice(ctx->device));
  DevMoving& m = it->second;
  if (m.last_use && m.last_use->ev) HIP_TRY(hipStreamWaitEvent(ctx->copy, m.last_use->ev, 0));
  return load_moving(ctx, m, leaf_means, L, ctx->copy);
}

int madicp_moving_release(madicp_ctx* ctx, int moving_id) {
  if (!ctx) return fail(MADICP_ERR_INVALID, "ctx is null");
  auto it = ctx->movings.find(moving_id);
  if (it == ctx->movings.end()) return fail(MADICP_ERR_INVALID, "unknown moving id");
  for (const StreamSlot& sl : ctx->slots)
    if (sl.moving_id == moving_id) return fail(MADICP_ERR_INVALID, "moving id belongs to a stream slot");
  HIP_TRY(hipSetDevice(ctx->device));
  EventRef after;
  RC_TRY(fence_event(ctx, &after));
  free_moving(ctx, it->second, after);
  ctx->movings.erase(it);
  return MADICP_OK;
}

// ---- streamed registrations: new scan in -> X / H / b / matched flags out ----------------------------
namespace {
// everything a slot needs for a scan of L leaves against K trees (grow-only; a no-op in steady state)
int prepare_slot(madicp_ctx* ctx, StreamSlot& sl, int L, int K, bool pinned_in, bool use_cache) {
  if (sl.moving_id < 0) {
    sl.moving_id = ctx->next_id++;
    ctx->movings[sl.moving_id] = DevMoving{};
  }
  DevMoving& mv = ctx->movings.at(sl.moving_id);
  RC_TRY(reserve_moving(ctx, mv, L, ctx->copy));
  if (pinned_in) RC_TRY(reserve_pinned_in(mv, L));
  if (use_cache) RC_TRY(reserve_cache(ctx, mv, std::max(1, K)));
  if (sl.h_matched_cap < (size_t)L + 16) {
    if (sl.h_matched) HIP_TRY(hipHostFree(sl.h_matched));
    sl.h_matched = nullptr;
    sl.h_matched_cap = 0;
    const size_t cap = (size_t)L + (size_t)L / 8 + 256;
    HIP_TRY(hipHostMalloc(&sl.h_matched, cap, hipHostMallocDefault));
    sl.h_matched_cap = cap;
  }
  return MADICP_OK;
}

int stream_submit_impl(madicp_ctx* ctx, const double* leaf_means, int32_t L, int moving_tree_id, const int* tree_ids, int K,
                       const double* X0, const madicp_icp_params* params, int n_iters, int* out_ticket) {
  RC_TRY(check_reg_args(ctx, X0, params, out_ticket, K, n_iters));
  if (K > 0 && !tree_ids) return fail(MADICP_ERR_INVALID, "null argument");
  HIP_TRY(hipSetDevice(ctx->device));
  const int ticket = ctx->next_ticket;
  StreamSlot& sl = ctx->slots[ticket % madicp_ctx::kStreamSlots];
  if (sl.pending) return fail(MADICP_ERR_CAPACITY, "stream ring full: collect the oldest ticket first");
  if (moving_tree_id >= 0) {
    auto tit = ctx->trees.find(moving_tree_id);
    if (tit == ctx->trees.end()) return fail(MADICP_ERR_INVALID, "unknown moving tree id");
    L = tit->second.n_leaves;
  } else {
    if (!leaf_means) return fail(MADICP_ERR_INVALID, "null argument");
    if (L < 1) return fail(MADICP_ERR_INVALID, "L must be >= 1");
  }
  const bool use_cache = n_iters > 1;
  // this slot — and, while they are idle, the other slots of the ring (consecutive scans are alike: the first
  // submission sizes the whole ring, later ones find nothing to do)
  RC_TRY(prepare_slot(ctx, sl, L, K, moving_tree_id < 0, use_cache));
  for (StreamSlot& other : ctx->slots)
    if (&other != &sl && !other.pending) RC_TRY(prepare_slot(ctx, other, L, K, moving_tree_id < 0, use_cache));
  DevMoving& mv = ctx->movings.at(sl.moving_id);
  // ---- feed (copy stream): the scan's leaves, then its Job ------------------------------------------------
  if (moving_tree_id >= 0) {
    DevTree& mt = ctx->trees.find(moving_tree_id)->second;
    mv.L = L;
    if (mt.compute_waited) {  // uploaded long ago: make the copy stream see whatever the compute stream did to it since
      EventRef ev;
      RC_TRY(fence_event(ctx, &ev));
      HIP_TRY(hipStreamWaitEvent(ctx->copy, ev->ev, 0));
    }
    hipLaunchKernelGGL(moving_from_leaves, dim3((L + 255) / 256), dim3(256), 0, ctx->copy, (const LeafRec*)mt.leaves, mv.xyzn, L);
    HIP_TRY(hipGetLastError());
  } else {
    RC_TRY(load_moving(ctx, mv, leaf_means, L, ctx->copy));
    mv.on_copy = false;  // ordering is by ev_up below
  }
  Job& j = *sl.h_job;
  RC_TRY(fill_job(ctx, j, mv, tree_ids, K, X0, params, n_iters, 0, use_cache));
  const Geometry geo = pick_geometry(ctx, L, K, 1);
  j.ranges_per_tree = geo.ranges_per_tree;
  if (ctx->interleave == 2 || (ctx->interleave == 1 && geo.queue)) j.flags |= kFlagInterleave;
  j.stage_min_leaves = ctx->stage_min_leaves;
  j.lds_top = geo.lds_bytes ? 1 : 0;
  HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void**>(&j.host_out), sl.h_out, 0));
  HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void**>(&j.host_matched), sl.h_matched, 0));
  // results out: through the device-resident outbox and icp_publish on the side stream (kernels.hip.h, Outbox), unless the
  // completion is an event on the compute stream or the loop is sharded (its matched flags are reduced behind icp_final)
  // — or the rounds need the whole chip to themselves: icp_persist / the xcd_fold variant wait INSIDE a launch for workgroups
  // that must all be resident, at 3 x 168 registers per SIMD lane nothing fits beside them, and an icp_publish workgroup that got
  // its CU first (the compute stream is still waiting for the feed) would keep one of them out until their bounded waits expire
  const Launch launch{geo.grid, 1, n_iters, geo.qpt, geo.lds_bytes, K, geo.ranges_per_tree, 0, geo.queue};
  const bool p2p_free = use_p2p(ctx, launch) && L <= madicp::kP2pFlagLeaves;  // (sharded, yet no collective and no host step)
  const bool side = ctx->publish_side && ctx->seq_completion && (!ctx->sharded() || p2p_free) && !use_persist(ctx, launch) &&
                    !use_fold(ctx, launch);
  j.outbox = side ? sl.d_outbox : nullptr;
  RC_TRY(next_p2p_epoch(ctx, launch, &j.p2p_epoch));
  const size_t job_bytes = offsetof(Job, trees) + sizeof(TreeDesc) * (size_t)std::max(1, K);
  j.seq = ticket + 1;
  sl.h_out->seq = 0;
  HIP_TRY(hipMemcpyAsync(sl.d_job, sl.h_job, job_bytes, hipMemcpyHostToDevice, ctx->copy));
  HIP_TRY(hipEventRecord(sl.ev_up, ctx->copy));
  // ---- the registration (compute stream) --------------------------------------------------------------
  // The compute stream must not start before the feed has landed.  While an earlier registration is still in flight
  // the HOST waits for the feed (tens of microseconds, hidden behind that registration) and the kernels follow it with
  // no barrier packet between two registrations; with nothing in flight the wait is the stream's (lowest latency).
  bool busy = false;
  for (const StreamSlot& other : ctx->slots) busy = busy || other.pending;
  if (busy && ctx->host_feed_wait)
    HIP_TRY(hipEventSynchronize(sl.ev_up));
  else
    HIP_TRY(hipStreamWaitEvent(ctx->stream, sl.ev_up, 0));
  if (n_iters == 1 || ctx->match_all) HIP_TRY(hipMemsetAsync(mv.matched, 0, (size_t)L, ctx->stream));
  {
    size_t doubles = 0;
    RC_TRY(prepare_partials(ctx, &launch, 1, &doubles));
  }
  const std::vector<int> ids{sl.moving_id};
  RC_TRY(run_rounds(ctx, launch, sl.d_job, ticket % madicp_ctx::kStreamSlots, ids, busy));
  if (side) {
    // (its wait is bounded by the longest the CALLER is prepared to wait, never less than 10 s: a compute stream legitimately
    // stalled for longer than a fixed bound would otherwise lose a registration that later completes)
    const unsigned long long spin_ms = (unsigned long long)std::max(10000, std::max(ctx->wait_timeout_ms, ctx->comm_timeout_ms));
    hipLaunchKernelGGL(icp_publish, dim3(1), dim3(256), 0, ctx->pub, (const Outbox*)sl.d_outbox, (const uint8_t*)mv.matched, L, ticket + 1,
                       j.host_out, j.host_matched, spin_ms * 100000ull);
    HIP_TRY(hipGetLastError());
  }
  sl.side = side;
  if (!ctx->seq_completion) HIP_TRY(hipEventRecord(sl.ev_done, ctx->stream));
  sl.by_seq = ctx->seq_completion != 0;
  sl.pending = true;
  sl.ticket = ticket;
  sl.L = L;
  ctx->next_ticket = ticket + 1;
  *out_ticket = ticket;
  return MADICP_OK;
}
}  // namespace

int madicp_stream_submit(madicp_ctx* ctx, const double* leaf_means, int32_t L, const int* tree_ids, int K,
                         const double X0[12], const madicp_icp_params* params, int n_iters, int* out_ticket) {
  return stream_submit_impl(ctx, leaf_means, L, -1, tree_ids, K, X0, params, n_iters, out_ticket);
}

int madicp_stream_submit_tree(madicp_ctx* ctx, int moving_tree_id, const int* tree_ids, int K, const double X0[12],
                              const madicp_icp_params* params, int n_iters, int* out_ticket) {
  if (moving_tree_id < 0) return fail(MADICP_ERR_INVALID, "unknown moving tree id");
  return stream_submit_impl(ctx, nullptr, 0, moving_tree_id, tree_ids, K, X0, params, n_iters, out_ticket);
}

int madicp_stream_collect(madicp_ctx* ctx, int ticket, double out_X[12], double out_H[36], double out_b[6],
                          uint8_t* out_matched, int32_t* out_n_matched, uint64_t* out_visits) {
  if (!ctx) return fail(MADICP_ERR_INVALID, "ctx is null");
  if (ticket < 0) return fail(MADICP_ERR_INVALID, "unknown ticket");
  StreamSlot& sl = ctx->slots[ticket % madicp_ctx::kStreamSlots];
  if (!sl.pending || sl.ticket != ticket) return fail(MADICP_ERR_INVALID, "unknown or already collected ticket");
  if (sl.by_seq) {
    // icp_final releases HostResult::seq after everything else it writes to the pinned block (kernels.hip.h)
    // (terminal errors give the slot back: a later submission must not find it "pending" for ever)
    const int32_t want = ticket + 1;
    const int32_t* seq = &sl.h_out->seq;
    const unsigned check_mask = ctx->wait_mode == 0 ? 0x3ffu : 0xfu;  // stream health: every few tens of microseconds
    const bool bounded = ctx->wait_timeout_ms > 0 || ctx->comm;
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 1; __atomic_load_n(seq, __ATOMIC_ACQUIRE) != want; ++spins) {
      if ((spins & check_mask) == 0) {
        const hipError_t q = hipStreamQuery(sl.side ? ctx->pub : ctx->stream);  // (the stream the sequence number comes from)
        if (q == hipSuccess) {
          if (__atomic_load_n(seq, __ATOMIC_ACQUIRE) == want) break;
          sl.pending = false;
          return fail(MADICP_ERR_DEVICE, "registration finished without publishing its results");
        }
        if (q != hipErrorNotReady) {
          sl.pending = false;
          return fail(MADICP_ERR_DEVICE, std::string("registration failed: ") + hipGetErrorString(q));
        }
        if (bounded) {
          const long long ms = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count();
          if (ctx->comm && ms > ctx->comm_timeout_ms) {
            // over the limit: unless the results arrived this very moment the communicator is aborted and the ticket is
            // gone — never MADICP_OK without the results copied out, and no second wait of comm_timeout_ms
            if (__atomic_load_n(seq, __ATOMIC_ACQUIRE) == want) break;
            sl.pending = false;
            return comm_abort(ctx, "a collective did not complete within " + std::to_string(ctx->comm_timeout_ms) +
                                       " ms (a rank did not join?)");
          }
          if (ctx->wait_timeout_ms > 0 && ms > ctx->wait_timeout_ms)
            return fail(MADICP_ERR_TIMEOUT, "registration still in flight after wait_timeout_ms; collect the ticket again");
        }
      }
      wait_pause(ctx);
    }
  } else {
    const hipError_t e = hipEventSynchronize(sl.ev_done);
    if (e != hipSuccess) {
      sl.pending = false;
      return fail(MADICP_ERR_DEVICE, std::string("registration failed: ") + hipGetErrorString(e));
    }
  }
  const HostResult& r = *sl.h_out;
  if (r.error) {
    sl.pending = false;
    if (r.error == 4) {
      ctx->p2p_broken = true;
      return fail(MADICP_ERR_COMM, "sharded registration: a peer's adders never arrived in this rank's mailbox (comm_timeout_ms)");
    }
    return fail(MADICP_ERR_DEVICE, r.error == 3 ? std::string("registration never left its results in the outbox (icp_publish timed out)")
                                                : "registration aborted on the device: an in-launch wait of the persistent round kernel ran out (code " + std::to_string(r.error) + ")");
  }
  if (out_X) std::memcpy(out_X, r.X, sizeof(r.X));
  if (out_H) std::memcpy(out_H, r.H, sizeof(r.H));
  if (out_b) std::memcpy(out_b, r.b, sizeof(r.b));
  if (out_matched) std::memcpy(out_matched, sl.h_matched, (size_t)sl.L);
  if (out_n_matched) *out_n_matched = r.n_matched;
  if (out_visits) *out_visits = r.visits;
  sl.pending = false;
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
  RC_TRY(bounded_sync(ctx, ctx->stream));
  for (int s = 0; s < n_scans; ++s) {
    const Job& j = ctx->h_fetch[s];
    if (j.error == 4) {
      ctx->p2p_broken = true;
      return fail(MADICP_ERR_COMM, "sharded registration: a peer's adders never arrived in this rank's mailbox (comm_timeout_ms)");
    }
    if (j.error) return fail(MADICP_ERR_DEVICE, "registration aborted on the device: an in-launch wait of the persistent round kernel ran out (code " + std::to_string(j.error) + ")");
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
  RC_TRY(bounded_sync(ctx, ctx->stream));
  return MADICP_OK;
}

// Results of the batch that was enqueued last, carried to a pinned host block by ONE small kernel behind it (instead of a copy
// command per scan and a stream synchronisation): the host collects them by ticket whenever it likes — typically after it has
// uploaded and enqueued the NEXT batch (a ring of four blocks).
int madicp_icp_publish_enqueue(madicp_ctx* ctx, int n_scans, int* out_ticket) {
  if (!ctx || !out_ticket) return fail(MADICP_ERR_INVALID, "null argument");
  if (n_scans < 1 || n_scans > ctx->last_batch) return fail(MADICP_ERR_INVALID, "n_scans exceeds the last batch");
  HIP_TRY(hipSetDevice(ctx->device));
  const int ticket = ctx->pub_next;
  const int slot = ticket % madicp_ctx::kPubSlots;
  if (ctx->pub_ticket[slot] >= 0)  // (its kernel may not have run yet: the block it will write is not free)
    return fail(MADICP_ERR_CAPACITY, "publish ring full: collect the oldest ticket first (four batches may be outstanding)");
  ++ctx->pub_next;
  if (!ctx->h_pub[slot]) {
    HIP_TRY(hipHostMalloc(&ctx->h_pub[slot], sizeof(madicp::HostResult) * MADICP_MAX_BATCH, hipHostMallocDefault));
    std::memset(ctx->h_pub[slot], 0, sizeof(madicp::HostResult) * MADICP_MAX_BATCH);
  }
  for (int s_ = 0; s_ < n_scans; ++s_) ctx->h_pub[slot][s_].seq = 0;
  madicp::HostResult* d_out = nullptr;
  HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void**>(&d_out), ctx->h_pub[slot], 0));
  hipLaunchKernelGGL(batch_publish, dim3(n_scans), dim3(64), 0, ctx->stream, (const Job*)ctx->d_jobs, d_out, ticket + 1);
  HIP_TRY(hipGetLastError());
  ctx->pub_n[slot] = n_scans;
  ctx->pub_ticket[slot] = ticket;
  *out_ticket = ticket;
  return MADICP_OK;
}

int madicp_icp_publish_collect(madicp_ctx* ctx, int ticket, int n_scans, double* out_X, double* out_H, double* out_b,
                               int32_t* out_n_matched, uint64_t* out_visits) {
  if (!ctx) return fail(MADICP_ERR_INVALID, "ctx is null");
  if (ticket < 0) return fail(MADICP_ERR_INVALID, "unknown ticket");
  const int slot = ticket % madicp_ctx::kPubSlots;
  if (ctx->pub_ticket[slot] != ticket || n_scans < 1 || n_scans > ctx->pub_n[slot])
    return fail(MADICP_ERR_INVALID, "unknown, overwritten or already collected ticket (a ring of four batches)");
  const madicp::HostResult* h = ctx->h_pub[slot];
  const auto t0 = std::chrono::steady_clock::now();
  for (int s_ = 0; s_ < n_scans; ++s_) {
    for (unsigned spins = 1; __atomic_load_n(&h[s_].seq, __ATOMIC_ACQUIRE) != ticket + 1; ++spins) {
      if ((spins & 0x3ffu) == 0) {
        const hipError_t q = hipStreamQuery(ctx->stream);
        if (q == hipSuccess) {
          if (__atomic_load_n(&h[s_].seq, __ATOMIC_ACQUIRE) == ticket + 1) break;
          return fail(MADICP_ERR_DEVICE, "batch finished without publishing its results");
        }
        if (q != hipErrorNotReady) return fail(MADICP_ERR_DEVICE, std::string("batch failed: ") + hipGetErrorString(q));
        const long long ms = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count();
        if (ctx->comm && ms > ctx->comm_timeout_ms)
          return comm_abort(ctx, "a collective did not complete within " + std::to_string(ctx->comm_timeout_ms) + " ms (a rank did not join?)");
        if (ctx->wait_timeout_ms > 0 && ms > ctx->wait_timeout_ms)
          return fail(MADICP_ERR_TIMEOUT, "batch still in flight after wait_timeout_ms; collect the ticket again");
      }
      wait_pause(ctx);
    }
  }
  ctx->pub_ticket[slot] = -1;
  for (int s_ = 0; s_ < n_scans; ++s_) {
    const madicp::HostResult& r = h[s_];
    if (r.error == 4) {
      ctx->p2p_broken = true;
      return fail(MADICP_ERR_COMM, "sharded registration: a peer's adders never arrived in this rank's mailbox (comm_timeout_ms)");
    }
    if (r.error) return fail(MADICP_ERR_DEVICE, "registration aborted on the device (code " + std::to_string(r.error) + ")");
    if (out_X) std::memcpy(out_X + 12 * s_, r.X, sizeof(r.X));
    if (out_H) std::memcpy(out_H + 36 * s_, r.H, sizeof(r.H));
    if (out_b) std::memcpy(out_b + 6 * s_, r.b, sizeof(r.b));
    if (out_n_matched) out_n_matched[s_] = r.n_matched;
    if (out_visits) out_visits[s_] = r.visits;
  }
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
  void* d_xi = nullptr;
  if (out_X_iters && n_iters > 0) RC_TRY(pool_alloc(ctx, sizeof(double) * 12 * (size_t)n_iters, ctx->stream, &d_xi));
  RegArgs a{1, &moving_id, tree_ids, K, X, params, n_iters, 0, nullptr, static_cast<double*>(d_xi)};
  int rc = enqueue_registration(ctx, a);
  if (rc == MADICP_OK) rc = madicp_icp_fetch(ctx, 1, X, out_H, out_b, nullptr, out_visits);
  if (rc == MADICP_OK && out_matched) rc = madicp_icp_fetch_matched(ctx, 0, out_matched, ctx->movings.at(moving_id).L);
  if (rc == MADICP_OK && d_xi) {
    hipError_t e = hipMemcpyAsync(out_X_iters, d_xi, sizeof(double) * 12 * (size_t)n_iters, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) rc = fail(MADICP_ERR_DEVICE, std::string("x_iters copy: ") + hipGetErrorString(e));
  }
  if (d_xi) {
    hipStreamSynchronize(ctx->stream);
    pool_free(ctx, d_xi, nullptr);
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
  void* d_corr = nullptr;
  if (out_corr && K > 0) RC_TRY(pool_alloc(ctx, sizeof(uint32_t) * (size_t)K * L, ctx->stream, &d_corr));
  RegArgs a{1, &moving_id, tree_ids, K, X, params, 1, kFlagNoUpdate, static_cast<uint32_t*>(d_corr), nullptr};
  const int saved_graph = ctx->use_graph;
  ctx->use_graph = 0;  // pointers in the job differ per call; nothing to gain from a graph for one round
  int rc = enqueue_registration(ctx, a);
  ctx->use_graph = saved_graph;
  if (rc == MADICP_OK) rc = madicp_icp_fetch(ctx, 1, nullptr, out_H, out_b, nullptr, out_visits);
  if (rc == MADICP_OK && out_matched) rc = madicp_icp_fetch_matched(ctx, 0, out_matched, L);
  if (rc == MADICP_OK && d_corr) {
    hipError_t e = hipMemcpyAsync(out_corr, d_corr, sizeof(uint32_t) * (size_t)K * L, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) rc = fail(MADICP_ERR_DEVICE, std::string("corr copy: ") + hipGetErrorString(e));
  }
  if (d_corr) {
    hipStreamSynchronize(ctx->stream);
    pool_free(ctx, d_corr, nullptr);
  }
  return rc;
}

#ifdef MADICP_MEASURE  // ---- measurement / test aids (include/madicp_hip_measure.h): only in the measurement build (mad_icp_amd/_measure)
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
                                 double* out_linearize_avg_us, double* out_solve_avg_us, uint64_t* out_visits_per_launch,
                                 uint64_t* out_walked_per_launch) {
  if (!ctx || reps < 1 || n_iters < 1) return fail(MADICP_ERR_INVALID, "bad argument");
  if (ctx->sharded()) return fail(MADICP_ERR_INVALID, "not available with a communicator");
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
  if (out_walked_per_launch)  // (h_fetch still holds the Jobs madicp_icp_fetch just read)
    for (int s = 0; s < n_scans; ++s) out_walked_per_launch[s] = ctx->h_fetch[s].walked / (uint64_t)n_iters;
  // icp_final alone, `reps` of them back to back inside ONE graph (a graph per launch would add the idle queue between two
  // graph launches, ~10 us, to a 5 us kernel): what is left of a registration is its n_iters icp_round launches
  const Geometry geo = pick_geometry(ctx, [&] { int m = 0; for (int s = 0; s < n_scans; ++s) m = std::max(m, ctx->movings.at(moving_ids[s]).L); return m; }(), K, n_scans);
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  HIP_TRY(hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
  for (int r = 0; r < reps; ++r)
    hipLaunchKernelGGL(icp_final, dim3(n_scans), dim3(kBlock), 0, ctx->stream, ctx->d_jobs, ctx->d_partials,
                       (const double*)nullptr, geo.grid, n_scans, (const unsigned long long*)nullptr, madicp::PeerBox{});
  HIP_TRY(hipStreamEndCapture(ctx->stream, &graph));
  HIP_TRY(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
  {
    hipGraphLaunch(exec, ctx->stream);
    hipEventRecord(ctx->ev_t0, ctx->stream);
    hipGraphLaunch(exec, ctx->stream);
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

int madicp_nn_time_descend(madicp_ctx* ctx, int tree_id, const double* queries, int64_t n, int reps, double* out_avg_us,
                           uint64_t* out_depth_sum) {
  if (!ctx || !queries || !out_avg_us) return fail(MADICP_ERR_INVALID, "null argument");
  if (n < 1 || reps < 1) return fail(MADICP_ERR_INVALID, "n and reps must be >= 1");
  auto it = ctx->trees.find(tree_id);
  if (it == ctx->trees.end()) return fail(MADICP_ERR_INVALID, "unknown tree id");
  HIP_TRY(hipSetDevice(ctx->device));
  if (!ctx->ev_t0) {
    HIP_TRY(hipEventCreate(&ctx->ev_t0));
    HIP_TRY(hipEventCreate(&ctx->ev_t1));
  }
  const size_t nq = (size_t)n;
  const size_t off_leaf = align_up(sizeof(double) * 3 * nq);
  const size_t off_dist = align_up(off_leaf + sizeof(uint32_t) * nq);
  const size_t off_depth = align_up(off_dist + sizeof(double) * nq);
  const size_t total = align_up(off_depth + sizeof(int32_t) * nq);
  void* blk = nullptr;
  RC_TRY(pool_alloc(ctx, total, ctx->stream, &blk));
  char* d = static_cast<char*>(blk);
  std::vector<int32_t> depth(nq);
  int rc = MADICP_OK;
  hipError_t e = hipMemcpyAsync(d, queries, sizeof(double) * 3 * nq, hipMemcpyHostToDevice, ctx->stream);
  auto launch = [&]() {
    return madicp_nn_search_device_enqueue(ctx, tree_id, reinterpret_cast<const double*>(d), n,
                                           reinterpret_cast<uint32_t*>(d + off_leaf), nullptr,
                                           reinterpret_cast<double*>(d + off_dist), reinterpret_cast<int32_t*>(d + off_depth));
  };
  if (e == hipSuccess) rc = launch();  // warm-up
  if (e == hipSuccess && rc == MADICP_OK) e = hipEventRecord(ctx->ev_t0, ctx->stream);
  for (int r = 0; r < reps && e == hipSuccess && rc == MADICP_OK; ++r) rc = launch();
  if (e == hipSuccess && rc == MADICP_OK) e = hipEventRecord(ctx->ev_t1, ctx->stream);
  if (e == hipSuccess && rc == MADICP_OK)
    e = hipMemcpyAsync(depth.data(), d + off_depth, sizeof(int32_t) * nq, hipMemcpyDeviceToHost, ctx->stream);
  hipError_t e2 = hipStreamSynchronize(ctx->stream);
  pool_free(ctx, blk, nullptr);
  if (rc != MADICP_OK) return rc;
  if (e != hipSuccess || e2 != hipSuccess)
    return fail(MADICP_ERR_DEVICE, std::string("nn_time_descend: ") + hipGetErrorString(e != hipSuccess ? e : e2));
  float ms = 0.f;
  HIP_TRY(hipEventElapsedTime(&ms, ctx->ev_t0, ctx->ev_t1));
  *out_avg_us = 1e3 * ms / reps;
  if (out_depth_sum) {
    uint64_t sum = 0;
    for (int32_t v : depth) sum += (uint64_t)v;
    *out_depth_sum = sum;
  }
  return MADICP_OK;
}

// plain device-to-device stream copy (16 bytes per lane): the measured HBM rate of THIS box, and the known byte count
// the PMC traffic counters are calibrated on (MI355X_MICROARCH.md, HBM section)
namespace madicp_detail {
__global__ void stream_copy(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
}  // namespace madicp_detail

int madicp_debug_stream_copy(madicp_ctx* ctx, int64_t bytes, int reps, double* out_gbs) {
  if (!ctx || !out_gbs) return fail(MADICP_ERR_INVALID, "null argument");
  if (bytes < 4096 || reps < 1) return fail(MADICP_ERR_INVALID, "bytes >= 4096 and reps >= 1");
  HIP_TRY(hipSetDevice(ctx->device));
  if (!ctx->ev_t0) {
    HIP_TRY(hipEventCreate(&ctx->ev_t0));
    HIP_TRY(hipEventCreate(&ctx->ev_t1));
  }
  const size_t n16 = (size_t)bytes / 16;
  void *a = nullptr, *b = nullptr;
  HIP_TRY(hipMalloc(&a, n16 * 16));
  hipError_t e = hipMalloc(&b, n16 * 16);
  if (e == hipSuccess) e = hipMemsetAsync(a, 1, n16 * 16, ctx->stream);
  const unsigned blocks = (unsigned)std::min<size_t>((n16 + 255) / 256, (size_t)ctx->n_cus * 16);
  auto run = [&]() {
    hipLaunchKernelGGL(madicp_detail::stream_copy, dim3(blocks), dim3(256), 0, ctx->stream, (const uint4*)a, (uint4*)b, n16);
  };
  if (e == hipSuccess) {
    run();
    e = hipEventRecord(ctx->ev_t0, ctx->stream);
    for (int r = 0; r < reps; ++r) run();
    if (e == hipSuccess) e = hipEventRecord(ctx->ev_t1, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  }
  float ms = 0.f;
  if (e == hipSuccess) e = hipEventElapsedTime(&ms, ctx->ev_t0, ctx->ev_t1);
  hipFree(a);
  hipFree(b);
  if (e != hipSuccess) return fail(MADICP_ERR_DEVICE, std::string("stream copy: ") + hipGetErrorString(e));
  *out_gbs = 2.0 * (double)(n16 * 16) * reps / (ms * 1e-3) / 1e9;  // read + write
  return MADICP_OK;
}

// Random 16-byte gathers over a large region: the access pattern of icp_round's dependent loads (one 16-byte screening
// record or quarter leaf record per lane, every lane another line), with a KNOWN set of lines — gather g reads the 16 bytes at
// 16 * ((g * 0x9E3779B97F4A7C15 + seed) mod n16), which the caller can enumerate — so that rocprofv3's FETCH_SIZE can be
// calibrated in the kernel's own access pattern instead of on a wide streaming copy (MI355X_MICROARCH.md, HBM section: "other
// access widths are uncalibrated: calibrate in your own access pattern").
namespace madicp_detail {
__global__ void gather16_probe(const uint4* __restrict__ src, size_t n16, unsigned long long n_gathers, unsigned long long seed,
                               unsigned int* __restrict__ sink) {
  unsigned int acc = 0;
  for (unsigned long long g = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; g < n_gathers;
       g += (unsigned long long)gridDim.x * blockDim.x) {
    const unsigned long long h = g * 0x9E3779B97F4A7C15ull + seed;
    const uint4 v = src[h % n16];
    acc ^= v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345678u) sink[0] = acc;  // (never true for the memset pattern: keeps the loads alive)
}
}  // namespace madicp_detail

int madicp_debug_gather16(madicp_ctx* ctx, int64_t region_bytes, int64_t n_gathers, uint64_t seed, int reps, double* out_avg_us) {
  if (!ctx || !out_avg_us) return fail(MADICP_ERR_INVALID, "null argument");
  if (region_bytes < 4096 || n_gathers < 1 || reps < 1) return fail(MADICP_ERR_INVALID, "region_bytes >= 4096, n_gathers >= 1, reps >= 1");
  HIP_TRY(hipSetDevice(ctx->device));
  if (!ctx->ev_t0) {
    HIP_TRY(hipEventCreate(&ctx->ev_t0));
    HIP_TRY(hipEventCreate(&ctx->ev_t1));
  }
  const size_t n16 = (size_t)region_bytes / 16;
  void* a = nullptr;
  unsigned int* sink = nullptr;
  HIP_TRY(hipMalloc(&a, n16 * 16));
  hipError_t e = hipMalloc(&sink, 64);
  if (e == hipSuccess) e = hipMemsetAsync(a, 1, n16 * 16, ctx->stream);
  const unsigned blocks = (unsigned)std::min<size_t>(((size_t)n_gathers + 255) / 256, (size_t)ctx->n_cus * 16);
  if (e == hipSuccess) e = hipEventRecord(ctx->ev_t0, ctx->stream);
  for (int r = 0; r < reps && e == hipSuccess; ++r) {
    hipLaunchKernelGGL(madicp_detail::gather16_probe, dim3(blocks), dim3(256), 0, ctx->stream, (const uint4*)a, n16,
                       (unsigned long long)n_gathers, (unsigned long long)seed + (unsigned long long)r, sink);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipEventRecord(ctx->ev_t1, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  float ms = 0.f;
  if (e == hipSuccess) e = hipEventElapsedTime(&ms, ctx->ev_t0, ctx->ev_t1);
  hipFree(a);
  if (sink) hipFree(sink);
  if (e != hipSuccess) return fail(MADICP_ERR_DEVICE, std::string("gather probe: ") + hipGetErrorString(e));
  *out_avg_us = 1e3 * ms / reps;
  return MADICP_OK;
}

#endif  // MADICP_MEASURE

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
  if (ctx->sharded()) return fail(MADICP_ERR_INVALID, "communicator already initialised");
  HIP_TRY(hipSetDevice(ctx->device));
  ncclUniqueId id;
  std::memcpy(&id, unique_id, 128);
  NCCL_TRY(ncclCommInitRank(&ctx->comm, n_ranks, id, rank));
  ctx->n_ranks = n_ranks;
  ctx->rank = rank;
  // graphs captured without the collectives are no longer the right sequence
  for (auto& g : ctx->graphs) hipGraphExecDestroy(g.second);
  ctx->graphs.clear();
  return MADICP_OK;
}

int madicp_comm_init_host(madicp_ctx* ctx, int n_ranks, int rank, madicp_host_allreduce_fn fn, void* user) {
  if (!ctx || !fn) return fail(MADICP_ERR_INVALID, "null argument");
  if (n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail(MADICP_ERR_INVALID, "bad rank / n_ranks");
  if (ctx->sharded()) return fail(MADICP_ERR_INVALID, "communicator already initialised");
  HIP_TRY(hipSetDevice(ctx->device));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  ctx->host_ar = fn;
  ctx->host_ar_user = user;
  ctx->n_ranks = n_ranks;
  ctx->rank = rank;
  for (auto& g : ctx->graphs) hipGraphExecDestroy(g.second);  // captured without the join over ranks
  ctx->graphs.clear();
  return MADICP_OK;
}

// ---- peer-mapped mailboxes (option "shard_p2p"; kernels.hip.h, "keyframe sharding without a collective between rounds") ----
// A mailbox SESSION is export -> (the caller gathers every rank's handle: a point every rank passes) -> attach.  The export
// zeroes the own mailbox, so that no row of an earlier session can carry a tag the new one expects (the registration counter
// restarts at 0 with every attach), and it does so BEFORE the handle leaves this rank: no peer can have begun a registration of
// the new session — it needs every rank's handle first, this one's included.  An attach therefore insists on a fresh export.
int madicp_p2p_export(madicp_ctx* ctx, uint8_t out_handle[64]) {
  if (!ctx || !out_handle) return fail(MADICP_ERR_INVALID, "null argument");
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t is 64 bytes");
  HIP_TRY(hipSetDevice(ctx->device));
  if (ctx->p2p_attached) return fail(MADICP_ERR_INVALID, "madicp_p2p_detach first: peers of the running session still write this mailbox");
  const size_t bytes = madicp::kP2pBoxWords * sizeof(unsigned long long);
  hipIpcMemHandle_t h;
  if (!ctx->p2p_box) {
    // Fine-grained device memory: peers write it while kernels of this device poll it.  Coarse-grained memory promises no
    // cross-device visibility while a kernel runs — over xGMI the polls could spin until comm_timeout_ms — so it is only taken
    // on request (option "p2p_allow_coarse": ranks that share ONE device, where the same L2 / memory serves writer and reader).
    for (int attempt = 0; attempt < 2 && !ctx->p2p_box; ++attempt) {
      if (attempt == 1 && !ctx->p2p_allow_coarse)
        return fail(MADICP_ERR_DEVICE, "mailbox: no exportable fine-grained device memory on this runtime (option p2p_allow_coarse = 1 "
                                       "accepts coarse-grained memory, for ranks that share one device only)");
      void* p = nullptr;
      hipError_t e = attempt == 0 ? hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained) : hipMalloc(&p, bytes);
      if (e == hipSuccess) e = hipIpcGetMemHandle(&h, p);
      if (e == hipSuccess) {
        ctx->p2p_box = static_cast<unsigned long long*>(p);
        ctx->p2p_fine = attempt == 0;
      } else {
        (void)hipGetLastError();
        if (p) hipFree(p);
        if (attempt == 1) return fail(MADICP_ERR_DEVICE, std::string("mailbox: ") + hipGetErrorString(e));
      }
    }
  }
  if (ctx->stream) HIP_TRY(hipStreamSynchronize(ctx->stream));
  HIP_TRY(hipMemset(ctx->p2p_box, 0, bytes));
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipIpcGetMemHandle(&h, ctx->p2p_box));
  std::memcpy(out_handle, &h, 64);
  ctx->p2p_fresh = true;
  return MADICP_OK;
}

int madicp_p2p_detach(madicp_ctx* ctx) {
  if (!ctx) return fail(MADICP_ERR_INVALID, "ctx is null");
  HIP_TRY(hipSetDevice(ctx->device));
  if (ctx->stream) HIP_TRY(hipStreamSynchronize(ctx->stream));
  if (ctx->pub) HIP_TRY(hipStreamSynchronize(ctx->pub));
  for (auto& g : ctx->graphs) hipGraphExecDestroy(g.second);  // (captured launches carry the peers' mappings)
  ctx->graphs.clear();
  for (int q = 0; q < madicp::kMaxRanks; ++q) {
    if (ctx->p2p_opened[q] && ctx->p2p_peer[q]) hipIpcCloseMemHandle(ctx->p2p_peer[q]);
    ctx->p2p_opened[q] = false;
    ctx->p2p_peer[q] = nullptr;
  }
  ctx->p2p_attached = false;
  ctx->p2p_broken = false;
  return MADICP_OK;
}

int madicp_p2p_attach(madicp_ctx* ctx, const uint8_t* handles, int n_ranks, int rank) {
  if (!ctx || !handles) return fail(MADICP_ERR_INVALID, "null argument");
  if (!ctx->sharded()) return fail(MADICP_ERR_INVALID, "madicp_p2p_attach needs a communicator (madicp_comm_init / madicp_comm_init_host) first");
  if (n_ranks != ctx->n_ranks || rank != ctx->rank) return fail(MADICP_ERR_INVALID, "n_ranks / rank differ from the communicator's");
  if (n_ranks > madicp::kMaxRanks) return fail(MADICP_ERR_CAPACITY, "peer mailboxes serve at most 8 ranks (one node)");
  if (!ctx->p2p_box) return fail(MADICP_ERR_INVALID, "madicp_p2p_export first: this rank's own mailbox does not exist yet");
  if (!ctx->p2p_fresh)
    return fail(MADICP_ERR_INVALID, "madicp_p2p_export again before re-attaching: the mailbox still holds the rows of the previous "
                                    "session, whose tags the new session's registration counter would meet again");
  if (ctx->p2p_attached) RC_TRY(madicp_p2p_detach(ctx));
  HIP_TRY(hipSetDevice(ctx->device));
  for (auto& g : ctx->graphs) hipGraphExecDestroy(g.second);  // (captured without the join over the ranks)
  ctx->graphs.clear();
  for (int q = 0; q < n_ranks; ++q) {
    if (q == rank) {
      ctx->p2p_peer[q] = ctx->p2p_box;
      continue;
    }
    hipIpcMemHandle_t h;
    std::memcpy(&h, handles + 64 * (size_t)q, 64);
    void* p = nullptr;
    const hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      madicp_p2p_detach(ctx);
      return fail(MADICP_ERR_COMM, "rank " + std::to_string(q) + "'s mailbox could not be mapped: " + hipGetErrorString(e));
    }
    ctx->p2p_peer[q] = static_cast<unsigned long long*>(p);
    ctx->p2p_opened[q] = true;
  }
  ctx->p2p_epoch = 0;
  ctx->p2p_fresh = false;
  ctx->p2p_broken = false;
  ctx->p2p_attached = true;
  return MADICP_OK;
}

int madicp_comm_destroy(madicp_ctx* ctx) {
  if (!ctx) return fail(MADICP_ERR_INVALID, "ctx is null");
  if (ctx->p2p_attached) RC_TRY(madicp_p2p_detach(ctx));
  if (ctx->host_ar) {
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    ctx->host_ar = nullptr;
    ctx->host_ar_user = nullptr;
    ctx->n_ranks = 1;
    ctx->rank = 0;
  }
  if (ctx->comm) {
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    for (auto& g : ctx->graphs) hipGraphExecDestroy(g.second);
    ctx->graphs.clear();
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

#include "frontend_capi.inc.h"
