/* Measurement and test aids — NOT part of the drop-in boundary (include/madicp_hip.h) and NOT in the product library: the
 * default build of libmadicp_hip.so exports exactly include/madicp_hip.h (tests/test_abi.py).  These entry points exist only
 * in the MEASUREMENT build of the same sources (-DMADICP_MEASURE; mad_icp_amd/_build.py puts it into mad_icp_amd/_measure/,
 * mad_icp_amd.capi.measure_variant() loads it): what bench.py needs to time kernels with HIP events on the library's own
 * stream and to calibrate rocprofv3's counters, and what a few tests need to look inside a device tree build.  Nothing in the
 * host layer (csrc/host) or in the pybind modules calls them. */
#ifndef MADICP_HIP_MEASURE_H
#define MADICP_HIP_MEASURE_H
#include "madicp_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Measurement aid (bench.py's roofline): n_launches back-to-back launches of the dominant kernel (icp_round, as
 * round 0: every pair walked, no solve prologue) for this batch at pose X0, replayed as one captured graph between two
 * hipEvents on the context's stream.  out_avg_us = time per launch (a dependent dispatch's launch overhead included, as
 * in a profiler trace of the registration graph); out_visits_per_launch (n_scans, optional) = internal nodes visited. */
int madicp_icp_time_linearize(madicp_ctx* ctx, int n_scans, const int* moving_ids, const int* tree_ids, int K,
                              const double* X0, const madicp_icp_params* params, int n_launches, double* out_avg_us,
                              uint64_t* out_visits_per_launch);

/* Measurement aid: the registration exactly as madicp_icp_register_batch_enqueue runs it (captured graph of n_iters
 * icp_round launches + icp_final, correspondence reuse and all), `reps` times between two hipEvents, then icp_final
 * alone.  out_linearize_avg_us = (registration - icp_final) / n_iters = average icp_round launch over the rounds of a
 * registration (what a profiler's kernel trace of the registration averages to); out_solve_avg_us = one icp_final;
 * out_visits_per_launch (n_scans) = internal nodes visited per round, averaged over the rounds, counted like the
 * reference's descent would (a reused correspondence counts its cached depth); out_walked_per_launch (n_scans, optional)
 * = the nodes the kernel really walked per round (correspondence reuse excluded). */
int madicp_icp_time_registration(madicp_ctx* ctx, int n_scans, const int* moving_ids, const int* tree_ids, int K,
                                 const double* X0, const madicp_icp_params* params, int n_iters, int reps,
                                 double* out_linearize_avg_us, double* out_solve_avg_us, uint64_t* out_visits_per_launch,
                                 uint64_t* out_walked_per_launch);

/* Measurement aid: `reps` launches of nn_descend (batched bestMatchingLeafFast, mad_tree.cpp:144-152 as
 * mad_tree_wrapper.h:48-67 loops it) over n host queries already copied to the device, between two hipEvents.
 * out_avg_us per launch; out_depth_sum = internal nodes visited by one launch. */
int madicp_nn_time_descend(madicp_ctx* ctx, int tree_id, const double* queries, int64_t n, int reps, double* out_avg_us,
                           uint64_t* out_depth_sum);
/* Measurement aid: a plain 16-byte-per-lane device-to-device copy of `bytes` bytes, `reps` times; out_gbs = read +
 * written bytes per second / 1e9.  The measured HBM rate of this box, and the known byte count on which bench.py
 * calibrates rocprofv3's FETCH_SIZE / WRITE_SIZE. */
int madicp_debug_stream_copy(madicp_ctx* ctx, int64_t bytes, int reps, double* out_gbs);
/* Measurement aid: `reps` launches of n_gathers random 16-byte loads over a region of region_bytes (gather g of launch r reads
 * the 16 bytes at 16 * ((g * 0x9E3779B97F4A7C15 + seed + r) mod (region_bytes / 16)): a set of lines the caller can enumerate).
 * icp_round's own access pattern with a known byte count — what bench.py calibrates rocprofv3's FETCH_SIZE on for that
 * kernel.  out_avg_us per launch. */
int madicp_debug_gather16(madicp_ctx* ctx, int64_t region_bytes, int64_t n_gathers, uint64_t seed, int reps, double* out_avg_us);

/* diagnostics (tests): the (n,3) points of the last madicp_tree_build on this context in the order the construction left
 * them — every leaf's members as the splits above it ordered them, i.e. the caller's container after the reference's
 * MADtree::build (mad_tree.cpp:95-97 with utils.h:37-52; the reference additionally overwrites a leaf's first member with
 * the leaf's representative, mad_tree.cpp:76-84).  Valid until the next build, ingest or deskew on the context. */
int madicp_debug_tree_build_points(madicp_ctx* ctx, double* out_xyz, int64_t n);

#ifdef __cplusplus
}
#endif
#endif
