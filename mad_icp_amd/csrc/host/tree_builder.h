// MAD-tree construction on the host, emitting the linear 64-byte node array the GPU kernels walk.
// Replaces MADtree::build / makeSubtree / getLeafs (reference mad_tree.cpp:47-142,154-163 and the helpers
// in utils.h:37-97).  Same decisions and the same fp64 operation order as the reference, different data
// structure: no per-node heap objects, no parent pointers — a DFS-preorder array with relative child
// offsets, so sub-trees built by different threads are position independent and are spliced by memcpy.
#pragma once
#include <cstdint>
#include <memory>
#include <new>
#include <type_traits>
#include <utility>
#include <vector>

#include "madicp_hip.h"

namespace madicp_host {

// std::vector that leaves trivially-constructible elements uninitialised on resize (the builder overwrites every one;
// zero-filling 3 MB first is a tenth of a whole build)
template <class T>
struct DefaultInitAllocator : std::allocator<T> {
  template <class U>
  struct rebind {
    using other = DefaultInitAllocator<U>;
  };
  template <class U>
  void construct(U* p) noexcept(std::is_nothrow_default_constructible<U>::value) {
    ::new (static_cast<void*>(p)) U;
  }
  template <class U, class... Args>
  void construct(U* p, Args&&... args) {
    ::new (static_cast<void*>(p)) U(std::forward<Args>(args)...);
  }
};
using NodeVec = std::vector<madicp_node, DefaultInitAllocator<madicp_node>>;
using IndexVec = std::vector<int32_t, DefaultInitAllocator<int32_t>>;

struct LinearTree {
  NodeVec nodes;        // DFS preorder; left child = i + 1, right child = i + nodes[i].right
  IndexVec leaf_nodes;  // node index of leaf `leaf_id`, i.e. getLeafs() order
  // max |mean_i - mean_0|_2 over the internal nodes with finite means (what the device's screening bound needs,
  // madicp_tree_upload computes it while it validates the array): -1 = not known.  Rotation invariant up to rounding.
  double rho2 = -1.0;
  int32_t num_leaves() const { return static_cast<int32_t>(leaf_nodes.size()); }
  int32_t num_nodes() const { return static_cast<int32_t>(nodes.size()); }
};

// points: n x 3 doubles, permuted in place exactly like the reference permutes its private copy.
// max_parallel_level: 0 = build on the calling thread; > 0 = fork tasks for the top of the tree (reference: the
// `level >= max_parallel_level` test at mad_tree.cpp:99; here the same argument forks two levels deeper, smaller
// tasks, see tree_builder.cpp "Task policy").  The result does not depend on it.
LinearTree build_tree(double* points, int64_t n, double b_max, double b_min, int max_parallel_level);

// test hook: see tree_builder.cpp
int64_t debug_partition(double* pts, int64_t n, const double* mean, const double* normal, int impl);

// MADtree::applyTransform (mad_tree.cpp:165-172) on the linear form; R row-major.
void transform_tree(LinearTree& tree, const double* R, const double* t);

// descent on the host copy (used for single-point MADtree.search when no device round trip is wanted
// is NOT provided on purpose: every search goes through the HIP path).

}  // namespace madicp_host
