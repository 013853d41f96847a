/* madicp_hip.h — C ABI of libmadicp_hip.so: the MI355X (gfx950) implementation of MAD-ICP's
 * data-association + registration hot path.
 *
 * This is the boundary the host C++ classes (mad_icp_amd/csrc/host: MADtree, MADicp, Pipeline) bind to,
 * and what a reference maintainer would bind to from mad_icp/src.  Plain pointers
 * and sizes only; no C++/torch types; no exceptions cross it.
 *
 * Conventions
 *   - every call returns 0 on success, <0 on error; madicp_last_error() gives the message of the last
 *     failing call on the calling thread.
 *   - host buffers are owned by the caller, device buffers by the library (referred to by integer ids).
 *   - 3x3 rotations / 6x6 matrices cross the ABI ROW-major; a pose is 12 doubles: R (9, row-major), t (3).
 *     (The reference holds them as column-major Eigen objects; the host classes convert.)
 *   - clouds are (N,3) float64, C-contiguous — the memory layout of std::vector<Eigen::Vector3d>
 *     (mad_icp/src/tools/mad_tree.h:42, mad_icp/src/pybind/eigen_stl_bindings.h:73-81).
 *   - all arithmetic on the path is IEEE fp64 with the reference's operation order and no FMA contraction,
 *     so descent/gate decisions are bit-identical to the CPU path.
 *   - a context is bound to one device and owns two streams: the compute stream (the caller's, or its own) carries the
 *     registrations, an internal copy stream feeds them (tree uploads, the next scan's leaves).  Calls on one context
 *     are not re-entrant and must come from one host thread at a time.
 *   - "synchronous on return" below means: the caller's HOST buffers are no longer referenced and every output
 *     argument is filled.  Uploads, transforms and releases return as soon as the host buffer has been staged; the
 *     device work behind them is ordered by events, never by a device-wide synchronisation, and nothing on the
 *     registration path allocates or frees device memory (pooled buffers, grow-only staging).
 */
#ifndef MADICP_HIP_H
#define MADICP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MADICP_OK 0
#define MADICP_ERR_INVALID (-1)   /* bad argument / unknown id            */
#define MADICP_ERR_DEVICE (-2)    /* HIP runtime error (no device, OOM …) */
#define MADICP_ERR_COMM (-3)      /* collective error: RCCL failure, a host transport's error, or a collective that did
                                     not complete within "comm_timeout_ms" (the communicator is then aborted)      */
#define MADICP_ERR_CAPACITY (-4)  /* more trees / scans than the ABI caps */
#define MADICP_ERR_TIMEOUT (-5)   /* a bounded host wait ("wait_timeout_ms") ran out; the work is still in flight */

#define MADICP_MAX_TREES 128 /* keyframe trees one registration may reference (reference README suggests 16) */
#define MADICP_MAX_BATCH 64  /* scans in flight in one batched registration                                  */

/* One MAD-tree node, 64 bytes, nodes stored in DFS preorder (left child = this + 1).
 * Replaces the heap-allocated `struct MADtree` (mad_icp/src/tools/mad_tree.h:47-102): only the fields the
 * hot path reads are kept — mean_ (:96), the split axis eigenvectors_.col(2) for internal nodes
 * (mad_tree.cpp:147-148) or the normal eigenvectors_.col(0) for leaves (mad_icp.cpp:65), bbox_(0)
 * (mad_icp.cpp:97) and the child links. */
typedef struct madicp_node {
  double mean[3]; /* internal: centroid; leaf: the surface point nearest to it (mad_tree.cpp:76-86) */
  double dir[3];  /* internal: split-plane normal (col 2); leaf: surface normal (col 0)             */
  int32_t right;  /* internal: index of the right child minus own index (>= 2); leaf: 0            */
  int32_t leaf_id;/* leaf: ordinal in getLeafs() order (mad_tree.cpp:154-163); internal: -1         */
  double bbox0;   /* bbox_(0): extent along the normal                                              */
} madicp_node;

/* MADicp constructor arguments (mad_icp/src/odometry/mad_icp.cpp:31-32).  rho_ker is passed as the user
 * gives it; the library takes the square root exactly where the reference constructor does. */
typedef struct madicp_icp_params {
  double min_ball; /* = b_max of the fixed trees (pipeline.cpp:52, mad_icp_wrapper.h:59) */
  double rho_ker;
  double b_ratio;
} madicp_icp_params;

typedef struct madicp_ctx madicp_ctx;

const char* madicp_last_error(void);
/* ABI version: bumped on any incompatible change. */
int madicp_abi_version(void);

/* ---- context ------------------------------------------------------------------------------------ */
/* stream: a hipStream_t created by the caller (e.g. torch's current stream) or NULL to let the library
 * create its own non-blocking stream on `device_id`. */
int madicp_ctx_create(int device_id, void* stream, madicp_ctx** out);
int madicp_ctx_destroy(madicp_ctx* ctx);
int madicp_ctx_synchronize(madicp_ctx* ctx);
/* Tuning knobs (all optional): key in {"grid_blocks_per_cu" (1..4), "use_graph" (0/1), "queries_per_lane" (1,2),
 * "cache_correspondences" (0/1: reuse a correspondence in later GN rounds when it is provably unchanged),
 * "cache_gate" (0/1, default 1: a pair that keeps its leaf and was rejected by the gate of mad_icp.cpp:81-83 with more slack than it
 * has moved since is not evaluated again — its leaf record is not fetched; the same decisions either way, and the same bits
 * wherever the pairs are added in scan order (every launch but the leaf-major rounds of a batch, which add their walkers last)),
 * "leaf_major" (0: never; n > 0, default 8192: when a batch shares the chip — more keyframe trees than workgroups per XCD piece —
 * every workgroup gets one range of the scan's leaves and all the trees of its piece, and a round that follows one in which the
 * workgroup walked fewer than n nodes per pass runs LEAF-MAJOR: the moving leaf is read and transformed once per pass for all
 * those trees, and the pairs that still have to walk are queued per wavefront and walked densely.  H and b come out in another
 * summation order (~1e-16 relative), so the poses of later rounds differ by as much: the decisions — leaf, depth, gate, matched
 * flags, visit count — are the same except where a pair sits exactly on a split plane or on the gate's radius (equal on every
 * data set of the test suite, asserted there; not a structural guarantee)),
 * "publish_side" (0/1, default 1: a streamed registration leaves its results in a device-resident outbox and a one-workgroup
 * kernel on a side stream carries them, and the matched flags, to the caller's pinned block while the compute stream is already
 * running the next registration — the two PCIe round trips of that hand-over were 5 us of every registration; off: the
 * closing kernel writes them itself),
 * "deal_trees" (0/1/2, default 2: a registration lists the caller's trees dealt over the eight XCD pieces of the round kernel —
 * neighbours in the caller's list, e.g. keyframes along a trajectory, cost a given scan about the same, and with them in one
 * piece one XCD worked while seven waited; 1 = round-robin, 2 = rows of eight in alternating direction, so that the piece that
 * drew the newest (dearest) keyframe of one row draws the oldest of the next; results keep the caller's indices),
 * "interleave_ranges" (0/1/2, default 2: the ranges a scan's leaves are cut into for the workgroups are every n-th group of 64
 * leaves instead of contiguous stretches — a stretch of the leaf order is a stretch of space, and stretches differ several-fold
 * in how many of their pairs pass the gate: the launch waited for the workgroups that drew the busy stretch; 1 = only in batches
 * that share the chip, 2 = every launch.  Another summation order for H and b (~1e-16), the same decisions),
 * "deep_min_leaves" (default 512: from 24 keyframes on with two or more scans in flight — from 48 on with one — a launch with more
 * trees than workgroups per XCD piece gives every workgroup one range of the scan and all the trees of its piece as soon as a range
 * holds this many leaves; otherwise the rule stays two passes of a workgroup, 1536 leaves — measured, profiles/r6_deep_threshold.md),
 * "units_per_workgroup" (1..64, default 1: with more trees than workgroups per scan, cut the leaves into enough ranges for at
 * least this many (tree, range) units per workgroup; measured: no gain),
 * "lds_stage_min_leaves" (a workgroup copies a tree's top levels into LDS when its unit holds at least this many leaves),
 * "nn_lds_top" (0/1: madicp_nn_search batches of >= 16 k queries stage the tree's top levels in LDS; default 0, measured
 * slower for single launches), "comm_graph" (0/1: with a communicator, capture the per-round RCCL all-reduces into the registration's hipGraph
 * instead of launching the rounds eagerly), "eager_when_busy" (0/1, default 1: a registration queued behind another one is
 * launched kernel by kernel instead of as a hipGraph — measured 5 us per registration cheaper on the queue),
 * "seq_completion" (0/1, default 1: a streamed registration publishes its completion through a sequence number in the
 * pinned result block that madicp_stream_collect polls, instead of an event on the stream), "host_feed_wait" (0/1,
 * default 1: while another registration is in flight the HOST waits for a streamed scan's upload before it launches the
 * rounds, so no barrier packet sits between two registrations), "wait_mode" (how madicp_stream_collect and
 * madicp_tree_build wait for the device's sequence number: 0 = spin on the calling core (default: lowest latency), 1 =
 * sched_yield between polls, 2 = sleep ~50 us between polls (a drop-in odometry process that has other threads to run)),
 * "wait_timeout_ms" (0 = unbounded, default; otherwise such a wait returns MADICP_ERR_TIMEOUT when it runs out — the
 * ticket stays collectable), "match_all_rounds" (0/1, default 0: the matched flags of a registration are
 * the OR over ALL its rounds instead of the last round's — what the reference's Pipeline leaves behind when its realtime
 * check ends the loop before iteration MAX_ICP_ITS - 1, the only one that resets them: pipeline.cpp:167-176),
 * "persistent" (0/1, default 0: all rounds of a single-GPU registration as ONE launch; bit-identical, measured slower —
 * profiles/r3_b_persist_negative.md), "shard_split" (0/1/2, default 1: with a communicator, a batch of >= 4 scans (2: of >= 2
 * scans) runs as two halves on two streams, so that one half's per-round all-reduce is in flight under the other half's round
 * kernel; collectives are enqueued in one order on every rank — the option decides how many collectives a round makes, so it MUST
 * have the same value on every rank of the communicator, as must the batch sizes the ranks submit), "shard_tail" (0/1, default 0: the sharded round kernel leaves
 * the rank's adders itself instead of a separate icp_reduce launch; bit-identical, measured slower —
 * profiles/r4_c_shard_probe.md), "xcd_fold" (0/1, default 0: per-round launches with the XCD-hierarchical join at the
 * end of each launch; bit-identical, measured slower — profiles/r3_j_xcd_fold_negative.md), "upload_f32" (0/1, default 1: a cloud
 * handed to madicp_cloud_upload / madicp_tree_build_begin whose coordinates are ALL exactly floats — what a sensor driver, a KITTI
 * .bin or a PointCloud2 delivers — crosses PCIe as floats and is widened on the device: the same doubles bit for bit, half the
 * transfer in front of the builder's first kernel; any other cloud goes as doubles), "comm_timeout_ms" (default 60000: with a communicator, how long the host waits for a
 * registration's collectives before it aborts the communicator and returns MADICP_ERR_COMM)}. */
int madicp_ctx_set_option(madicp_ctx* ctx, const char* key, int64_t value);
/* the current value of one of the keys above (a caller that changes an option of a context it shares puts it back) */
int madicp_ctx_get_option(madicp_ctx* ctx, const char* key, int64_t* out_value);

/* ---- MAD-tree (fixed side) ---------------------------------------------------------------------- */
/* Upload a linearised tree.  Replaces keeping `MADtree*` alive in Frame::tree_ (frame.h:47).  The node array is
 * validated first (DFS-preorder structure, child links inside their parent's extent, leaf ordinals unique and in
 * range): a malformed array is MADICP_ERR_INVALID, never an out-of-bounds access.  Returns once `nodes` has been
 * staged; the copy and the build of the screening records run on the context's copy stream. */
int madicp_tree_upload(madicp_ctx* ctx, const madicp_node* nodes, int32_t n_nodes, int32_t n_leaves, int* out_tree_id);
/* The same for an array whose structure the PRODUCER guarantees — madicp_host_tree_build's output (csrc/host/
 * tree_builder.cpp), which is what the host MADtree / Pipeline classes upload once per scan — so that the O(n) host
 * validation pass (0.25 ms of a 2.9 ms drop-in frame at 45 k nodes) is not paid per frame.  rho2 = max |mean_i - mean_0|_2
 * over the internal nodes with finite means (madicp_tree_upload computes it while validating; the builder while
 * numbering the leaves).  No check beyond n_nodes == 2 n_leaves - 1: a malformed array here is undefined behaviour
 * on the device.  Arrays from anywhere else go through madicp_tree_upload. */
int madicp_tree_upload_trusted(madicp_ctx* ctx, const madicp_node* nodes, int32_t n_nodes, int32_t n_leaves, double rho2,
                               int* out_tree_id);
int madicp_tree_release(madicp_ctx* ctx, int tree_id);
int madicp_tree_download(madicp_ctx* ctx, int tree_id, madicp_node* out_nodes, int32_t n_nodes);
/* MADtree::applyTransform (mad_tree.cpp:165-172): mean <- R mean + t, dir <- R dir on every node.  Stream-ordered
 * behind every registration enqueued so far; no host synchronisation. */
int madicp_tree_transform(madicp_ctx* ctx, int tree_id, const double R[9], const double t[3]);
/* Batched MADtree::bestMatchingLeafFast (mad_tree.cpp:144-152) as used by MADtreeWrapper::search /
 * searchCloud / searchCloudDist (mad_tree_wrapper.h:42-67).  queries: host (n,3).  Any output may be NULL.
 * out_leaf_id = getLeafs() ordinal, out_node = index into the node array, out_dist = |q - leaf.mean|,
 * out_depth = internal nodes visited. */
int madicp_nn_search(madicp_ctx* ctx, int tree_id, const double* queries, int64_t n, uint32_t* out_leaf_id,
                     uint32_t* out_node, double* out_dist, int32_t* out_depth);
/* Same with queries / outputs already resident on the device (device pointers), asynchronous. */
int madicp_nn_search_device_enqueue(madicp_ctx* ctx, int tree_id, const double* d_queries, int64_t n,
                                    uint32_t* d_out_leaf_id, uint32_t* d_out_node, double* d_out_dist,
                                    int32_t* d_out_depth);

/* ---- moving side -------------------------------------------------------------------------------- */
/* MADicp::setMoving (mad_icp.cpp:53-55): the sensor-frame means of the current scan's leaves, (L,3). */
int madicp_moving_upload(madicp_ctx* ctx, const double* leaf_means, int32_t L, int* out_moving_id);
/* The next scan into the SAME buffers (grow-only: no allocation, free or synchronisation in steady state). */
int madicp_moving_update(madicp_ctx* ctx, int moving_id, const double* leaf_means, int32_t L);
/* The same on the library's COPY stream: the transfer runs beside the batch the compute stream is working on — one that reads
 * OTHER moving sets — instead of behind it.  Ordered by the library: the transfer waits for the last registration that read
 * this set, the next registration that reads it waits for the transfer.  (Two sets of moving ids used alternately: the upload
 * of batch i + 1 hides under batch i.) */
int madicp_moving_update_async(madicp_ctx* ctx, int moving_id, const double* leaf_means, int32_t L);
int madicp_moving_release(madicp_ctx* ctx, int moving_id);

/* ---- registration ------------------------------------------------------------------------------- */
/* One linearisation at a given pose, no state update: resetAdders + update() over K trees
 * (mad_icp.cpp:43-51,74-103 under pipeline.cpp:178-183).  out_H (36, row-major) and out_b (6) are the
 * summed adders; out_corr (K*L, optional) gets for tree k / moving leaf i the NN leaf ordinal, with bit 31
 * set when the gate at mad_icp.cpp:81-83 rejected the pair; out_matched (L, optional) the OR over trees of
 * the accepted pairs; out_visits (optional) the total number of internal nodes visited. */
int madicp_icp_linearize(madicp_ctx* ctx, int moving_id, const int* tree_ids, int K, const double X[12],
                         const madicp_icp_params* params, double out_H[36], double out_b[6], uint32_t* out_corr,
                         uint8_t* out_matched, uint64_t* out_visits);

/* The whole GN loop on the device, no host round trip between iterations: n_iters rounds of
 * {resetAdders; update() over the K trees; updateState()} — pipeline.cpp:166-193 and
 * mad_icp_wrapper.h:72-81.  X is MADicp::X_ (in: initial guess, out: estimate); out_H/out_b are
 * MADicp::H_adder_/b_adder_ of the LAST round (pipeline.cpp:223 reads H_adder_); out_matched (L, optional)
 * are the matched_ flags of the last round (cleared before it, pipeline.cpp:172-176);
 * out_X_iters (n_iters*12, optional) the pose BEFORE each round. */
int madicp_icp_register(madicp_ctx* ctx, int moving_id, const int* tree_ids, int K, double X[12],
                        const madicp_icp_params* params, int n_iters, double out_H[36], double out_b[6],
                        uint8_t* out_matched, double* out_X_iters, uint64_t* out_visits);

/* n_scans independent registrations against the same K trees, advanced in lock-step by the same
 * launches (BASELINE config 5: scans batched in flight).  X: n_scans*12, out_H: n_scans*36,
 * out_b: n_scans*6, out_n_matched: n_scans (count of matched leaves of the last round). */
int madicp_icp_register_batch(madicp_ctx* ctx, int n_scans, const int* moving_ids, const int* tree_ids, int K,
                              double* X, const madicp_icp_params* params, int n_iters, double* out_H,
                              double* out_b, int32_t* out_n_matched, uint64_t* out_visits);

/* Asynchronous form for throughput measurement: everything stays on the device; results are fetched
 * later with madicp_icp_fetch (which synchronises).  X0: n_scans*12 initial guesses (copied on call). */
int madicp_icp_register_batch_enqueue(madicp_ctx* ctx, int n_scans, const int* moving_ids, const int* tree_ids,
                                      int K, const double* X0, const madicp_icp_params* params, int n_iters);
int madicp_icp_fetch(madicp_ctx* ctx, int n_scans, double* out_X, double* out_H, double* out_b,
                     int32_t* out_n_matched, uint64_t* out_visits);
/* matched_ flags of scan `scan` of the last batch (L bytes). */
int madicp_icp_fetch_matched(madicp_ctx* ctx, int scan, uint8_t* out_matched, int32_t L);
/* Batches in flight, results out without a stream synchronisation: madicp_icp_publish_enqueue puts ONE small kernel behind the
 * batch that was enqueued last; it carries the batch's X / H / b / counts to a pinned host block and releases a sequence
 * number.  madicp_icp_publish_collect(ticket) waits for that number (bounded like madicp_stream_collect: "wait_mode",
 * "wait_timeout_ms", "comm_timeout_ms") and copies the results out — whenever the caller likes, typically after it has uploaded
 * (madicp_moving_update_async) and enqueued the NEXT batch.  A ring of four result blocks: a ticket must be collected before
 * the fourth batch after it is published. */
int madicp_icp_publish_enqueue(madicp_ctx* ctx, int n_scans, int* out_ticket);
int madicp_icp_publish_collect(madicp_ctx* ctx, int ticket, int n_scans, double* out_X, double* out_H, double* out_b,
                               int32_t* out_n_matched, uint64_t* out_visits);

/* ---- streamed registrations: new scan in -> X / H / b / matched flags out --------------------------- */
/* What Pipeline::compute does per frame (pipeline.cpp:154-204), as one asynchronous submission: setMoving(leaf_means),
 * init(X0), n_iters rounds against K resident trees, and the read-out of X_, H_adder_, b_adder_, the matched_ flags and
 * their count.  The leaves and the job description are fed on the copy stream (they overlap the registration in
 * flight), the results are written by the registration's last kernel straight into pinned host memory, a sequence
 * number last, and madicp_stream_collect polls that number (spinning on the calling thread until the registration has
 * finished; an event instead with option "seq_completion" = 0).  Up to 4 tickets may be outstanding; collect them in any order before
 * their slot is needed again (MADICP_ERR_CAPACITY otherwise).  Bit-identical to madicp_icp_register on the same inputs. */
int madicp_stream_submit(madicp_ctx* ctx, const double* leaf_means, int32_t L, const int* tree_ids, int K,
                         const double X0[12], const madicp_icp_params* params, int n_iters, int* out_ticket);
/* The same with the moving set taken from a tree that is already resident: the scan's own MAD-tree, uploaded for the
 * frame window (pipeline.cpp:140-144: current_leaves_ ARE its leaves, in getLeafs() order, sensor frame). */
int madicp_stream_submit_tree(madicp_ctx* ctx, int moving_tree_id, const int* tree_ids, int K, const double X0[12],
                              const madicp_icp_params* params, int n_iters, int* out_ticket);
/* Any output may be NULL; out_matched holds L bytes. */
int madicp_stream_collect(madicp_ctx* ctx, int ticket, double out_X[12], double out_H[36], double out_b[6],
                          uint8_t* out_matched, int32_t* out_n_matched, uint64_t* out_visits);

/* (Measurement and test aids — madicp_icp_time_*, madicp_nn_time_descend, madicp_debug_* — are declared in
 * include/madicp_hip_measure.h: exported by the same library for bench.py and tests/, not part of the drop-in boundary.) */

/* ---- device front-end: scans resident in HBM, ingest + deskew, MAD-tree build (row f-1) -- */
/* Additive (the reference has no such interface: its Pipeline takes a host vector and builds on the CPU).  A "cloud" is
 * an (n,3) float64 scan in device memory, referred to by id.  Everything here runs on the context's copy stream, i.e.
 * concurrently with a registration in flight; results are handed to the compute stream by events. */
int madicp_cloud_upload(madicp_ctx* ctx, const double* xyz, int64_t n, int* out_cloud_id);
int madicp_cloud_release(madicp_ctx* ctx, int cloud_id);
int madicp_cloud_size(madicp_ctx* ctx, int cloud_id, int64_t* out_n);
int madicp_cloud_download(madicp_ctx* ctx, int cloud_id, double* out_xyz, int64_t n);
/* apps/cpp_runners/bin_runner.cpp:126-166: float32 records (x, y, z, intensity ...; `stride_floats` floats apart) ->
 * fp64 points in input order, dropping |p| < min_range, |p| > max_range (|p| in float, as there) and NaN coordinates;
 * kitti_correction != 0 applies the VERTICAL_ANGLE_OFFSET rotation of :153-158.  Synchronises once (the number of
 * surviving points sizes the cloud). */
int madicp_cloud_ingest_f32(madicp_ctx* ctx, const float* records, int64_t n_records, int stride_floats, double min_range,
                            double max_range, int kitti_correction, int* out_cloud_id, int64_t* out_n);
/* Pipeline::deskew (mad_icp/src/odometry/pipeline.cpp:79-123) in place: azimuth sort, then every point moved by the pose
 * of its time chunk.  velocity = naive_vel of :82-86 ([translation; logMapSO3(rotation)] of T_prev^-1 T_now, divided by
 * the scan period), which the caller computes (six numbers, host libm like the reference); the thresholds, times and
 * chunk poses are tabulated on the host with the reference's own running arithmetic.  The cloud ends up in azimuth
 * order, like the reference's.  Points whose azimuths tie exactly may come out in a different order than std::sort
 * leaves them (it is not stable), and the device atan2 may differ from libm's in the last bit: the result is the
 * reference's up to the order of such ties.  out_chunks (n, optional): the time chunk of every point in walk order
 * (largest azimuth first) — synchronises when given. */
int madicp_cloud_deskew(madicp_ctx* ctx, int cloud_id, const double velocity[6], double sensor_hz, int32_t* out_chunks);
/* MADtree::build + getLeafs + the upload, all on the device (mad_tree.cpp:47-142,154-163): the tree of the cloud becomes
 * a resident tree exactly like one given to madicp_tree_upload (same node format, madicp_tree_download returns it).
 * Same decisions as the reference node by node, the reference's member order (the permutation utils.h:37-52 leaves), and
 * for the nodes of at most 32 points its summation order: same topology and leaf representatives as the host builder on
 * every scan tried, centroids and covariances of small nodes bit for bit.  Not the same bits everywhere: larger nodes add in
 * a parallel shape and the eigen-solver's trigonometry comes from the device library (mad_icp_amd/csrc/hip/tree_build.hip.h);
 * bit-reproducible run to run.  The cloud is left untouched.  Synchronises the copy stream once (the leaf count sizes the tree). */
int madicp_tree_build(madicp_ctx* ctx, int cloud_id, double b_max, double b_min, int* out_tree_id, int32_t* out_n_leaves);
/* The same construction as a look-ahead, for a caller that has the NEXT scan in hand while the current one registers
 * (what Pipeline::prefetch does with the device front-end on).  _begin copies the (n,3) float64 scan to the device and
 * enqueues the whole level loop on a stream of its own — the library's BUILD stream, so that neither the registration on
 * the compute stream nor its feed on the copy stream queues behind it — and returns without waiting; _end waits for the
 * leaf count, sizes and emits the tree and returns its id.  One look-ahead per context: a second _begin, or
 * madicp_tree_build / madicp_cloud_ingest_f32 / madicp_cloud_deskew / madicp_tree_build_stats before the _end, return
 * MADICP_ERR_CAPACITY (they share the builder's scratch).  Registrations, uploads, transforms, searches and releases are
 * free to run in between.  The tree is the one madicp_cloud_upload + madicp_tree_build give for the same scan, bit for bit. */
int madicp_tree_build_begin(madicp_ctx* ctx, const double* xyz, int64_t n, double b_max, double b_min);
int madicp_tree_build_end(madicp_ctx* ctx, int* out_tree_id, int32_t* out_n_leaves);
/* drops a look-ahead whose scan never came (waits for the device work, frees the copy of the scan); no-op without one */
int madicp_tree_build_cancel(madicp_ctx* ctx);
int madicp_tree_info(madicp_ctx* ctx, int tree_id, int32_t* out_n_nodes, int32_t* out_n_leaves);
/* diagnostics of the last madicp_tree_build on this context: out[0] deepest level, out[1] nodes handled one-per-lane, out[2..65] nodes handled one-per-wavefront per level, out[66..129] nodes handled chip-wide per level */
int madicp_tree_build_stats(madicp_ctx* ctx, int32_t out[130]);

/* ---- multi-GPU: keyframe trees sharded across ranks, one all-reduce of (H,b) per GN round ---------- */
/* Peer-mapped mailboxes: keyframe sharding WITHOUT a collective between two rounds (option "shard_p2p" = 1, default 0; additive,
 * one node, at most 8 ranks).  The per-round join of the ranks' adders — the reference's serial sum of mad_icp.cpp:106-109, 30
 * doubles per scan — is then done inside the next round kernel's prologue: workgroup 0 writes this rank's sums as tagged
 * granules into every peer's mailbox, every workgroup polls the peers' rows in the own mailbox and adds them in rank order
 * (all ranks the same bits).  The matched flags of the last round (a leaf is an inlier if ANY keyframe on ANY rank matched it:
 * mad_icp.cpp:85, pipeline.cpp:197-204) travel the same way, 32 flags per tagged word, OR-ed by the closing kernel — for moving
 * sets of up to 131 072 leaves; larger ones OR them through the communicator as before.  A sharded registration is then the
 * single-GPU launch sequence with NO collective and no host step: it is captured in a hipGraph and streams its results out like
 * a single-GPU one.
 *   madicp_p2p_export: allocates this rank's mailbox (the first time), ZEROES it and returns its 64-byte hipIpcMemHandle_t.
 *                      Fine-grained device memory; where the runtime cannot export that, the call fails unless option
 *                      "p2p_allow_coarse" = 1 (coarse-grained memory promises no visibility of a peer's stores to a running
 *                      kernel: acceptable only for ranks that share ONE device).  get_option "p2p_fine_grained" says which.
 *   madicp_p2p_attach: `handles` = n_ranks x 64 bytes in rank order (gathered by the caller — torch.distributed, MPI ...);
 *                      needs madicp_comm_init / madicp_comm_init_host first, with the same n_ranks / rank, and an export SINCE
 *                      the last attach: a session is export -> gather (a point every rank passes) -> attach, so that every
 *                      mailbox is clean before any rank can begin a registration of the session, whose counter restarts at 0.
 *   madicp_p2p_detach: unmaps the peers' mailboxes (madicp_comm_destroy does it too).
 * A peer whose row does not arrive within "comm_timeout_ms" fails the registration with MADICP_ERR_COMM — once per
 * registration, not once per round — and the session is over (the ranks' counters may disagree from there on): later sharded
 * registrations fail with MADICP_ERR_COMM until every rank has exported and attached again.  Every rank must submit the same
 * sequence of sharded registrations (as with any collective). */
int madicp_p2p_export(madicp_ctx* ctx, uint8_t out_handle[64]);
int madicp_p2p_attach(madicp_ctx* ctx, const uint8_t* handles, int n_ranks, int rank);
int madicp_p2p_detach(madicp_ctx* ctx);

/* Replaces the serial sum of per-thread adders at mad_icp.cpp:106-109.  unique_id: the 128-byte
 * ncclUniqueId produced by madicp_comm_unique_id on rank 0 and distributed by the caller (e.g. a
 * torch.distributed broadcast).  After this call every registration on the context all-reduces
 * [H(21) b(6) n] over the communicator after each round and ORs the matched flags after the last.  With a
 * communicator a registration may be given K = 0 trees (a rank that owns no keyframe still joins every collective,
 * contributing zeros). */
int madicp_comm_unique_id(uint8_t out_id[128]);
int madicp_comm_init(madicp_ctx* ctx, const uint8_t unique_id[128], int n_ranks, int rank);
int madicp_comm_destroy(madicp_ctx* ctx);

/* The same sharded registration over a HOST-STAGED transport supplied by the caller (MPI, gloo, a socket ...) instead of
 * RCCL: after every round's icp_reduce the library copies this rank's [H(21) b(6) n v w] per scan to pinned host memory,
 * waits for that copy (only this rank's own kernels are in front of it: no peer can stall it), calls `fn` — which must
 * leave the element-wise reduction over all ranks in `buf` on every rank, and whose own time-out is the only bound on a
 * peer that never joins ("comm_timeout_ms" covers RCCL's collectives, not the caller's transport) — and copies the totals
 * back in front of the next round; the matched flags go through the same
 * call once with MADICP_REDUCE_MAX_U8.  Kernel sequence, summation order inside a rank and the K = 0 rank are exactly
 * those of the RCCL path: it is the way to run (and test) the multi-rank product path where RCCL cannot form a
 * communicator — e.g. two ranks sharing one GPU — and a fallback where there is no xGMI/RDMA path between the ranks.
 * One host round trip per round, so not capturable in a hipGraph and slower than RCCL (tens of microseconds per round).
 * `fn` returns 0 on success; anything else fails the registration with MADICP_ERR_COMM.  madicp_comm_destroy ends it. */
#define MADICP_REDUCE_SUM_F64 0
#define MADICP_REDUCE_MAX_U8 1
typedef int (*madicp_host_allreduce_fn)(void* user, void* buf, int64_t count, int kind);
int madicp_comm_init_host(madicp_ctx* ctx, int n_ranks, int rank, madicp_host_allreduce_fn fn, void* user);

#ifdef __cplusplus
}
#endif
#endif /* MADICP_HIP_H */
