/* madicp_host.h — C ABI of libmadicp_host.so: the host-side (CPU) half that FEEDS the HIP path.
 * Only the MAD-tree construction lives here (it stays on the CPU in this slice — SURVEY §8 row f-1 moves it
 * to the GPU later); nearest-neighbour search and registration are in madicp_hip.h and have no CPU path.
 */
#ifndef MADICP_HOST_H
#define MADICP_HOST_H

#include <stdint.h>

#include "madicp_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct madicp_host_tree madicp_host_tree;

/* MADtree constructor / build (mad_icp/src/tools/mad_tree.cpp:35-130) + getLeafs (:154-163).
 * points: (n,3) float64, NOT modified (the library works on a private copy, as the reference's by-value
 * ContainerType arguments do: pipeline.h:71, mad_tree_wrapper.h:36).  Returns NULL if n <= 0. */
madicp_host_tree* madicp_host_tree_build(const double* points, int64_t n, double b_max, double b_min,
                                         int max_parallel_level);
void madicp_host_tree_free(madicp_host_tree* t);
int32_t madicp_host_tree_num_nodes(const madicp_host_tree* t);
int32_t madicp_host_tree_num_leaves(const madicp_host_tree* t);
/* the linear node array, ready for madicp_tree_upload */
const madicp_node* madicp_host_tree_nodes(const madicp_host_tree* t);
/* node index of every leaf in getLeafs() order */
const int32_t* madicp_host_tree_leaf_nodes(const madicp_host_tree* t);
/* leaf means (L,3) in getLeafs() order — what MADicp::setMoving consumes (mad_icp.cpp:53-55, :78) */
void madicp_host_tree_leaf_means(const madicp_host_tree* t, double* out);
/* MADtree::applyTransform on the host copy (mad_tree.cpp:165-172); R row-major */
void madicp_host_tree_transform(madicp_host_tree* t, const double R[9], const double tr[3]);

/* MADicp::updateState after the adders have been joined (mad_icp.cpp:111-116): dx = LDLT(H).solve(-b),
 * X <- X * [expSO3(dx[3:6]), dx[0:3]].  H row-major 36 (lower triangle read), X: R row-major 9 + t 3, in place.
 * Host counterpart of the device-side solve; used by the staged multi-rank driver (mad_icp_amd/sharded.py). */
void madicp_host_gn_update(const double H[36], const double b[6], double X[12]);
/* det(H^-1), the keyframe weight of pipeline.cpp:223 */
double madicp_host_det_of_inverse6(const double H[36]);

/* max |mean_i - mean_0|_2 over the internal nodes with finite means — the rho2 argument of madicp_tree_upload_trusted */
double madicp_host_tree_rho2(const madicp_host_tree* t);

/* The thread budget of the host tree builder (the caller included), process-wide — what Pipeline's num_threads
 * argument sets, like the reference's omp_set_num_threads(num_threads) (pipeline.cpp:64-65). */
void madicp_host_set_threads(int n);

/* Test hook: `split` of utils.h:37-52 about the plane (mean, normal) applied to points (n,3) IN PLACE, by the
 * reference's own loop (impl 0) or by the builder's flag-driven closed form (impl 1); returns the split position
 * (first point of the right part), -1 on bad arguments.  Both must leave the same permutation. */
int64_t madicp_host_debug_partition(double* points, int64_t n, const double mean[3], const double normal[3], int impl);
/* Test hook: MADtree::build on the CALLER's buffer — points (n,3) are permuted in place exactly like the reference permutes
 * its private copy (utils.h:37-52 at every internal node, and mad_tree.cpp:76-84 at every leaf: the representative is
 * written over the leaf's first member); returns the number of leaves, -1 on bad arguments. */
int64_t madicp_host_debug_tree_points(double* points, int64_t n, double b_max, double b_min, int max_parallel_level);
/* Pipeline::deskew (mad_icp/src/odometry/pipeline.cpp:79-123) on its own, in place, output in azimuth order; poses as 12
 * doubles (R row-major, t).  route 0: the azimuth order from the task pool (unique when the azimuths are distinct), the
 * reference's serial std::sort of (azimuth, point) pairs when two of them tie; route 1: always the reference's route.
 * Returns 1 when the parallel order was used, 0 when the serial route ran, < 0 on bad arguments.  out_velocity6 (optional):
 * naive_vel of pipeline.cpp:82-86. */
int madicp_host_debug_deskew(double* points, int64_t n, const double T_prev[12], const double T_now[12], double sensor_hz, int route,
                             double* out_velocity6);

#ifdef __cplusplus
}
#endif
#endif /* MADICP_HOST_H */
