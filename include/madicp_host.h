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

#ifdef __cplusplus
}
#endif
#endif /* MADICP_HOST_H */
