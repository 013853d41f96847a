// MADtree — host-side owner of one MAD-tree: the linear node array on the host and its resident copy in HBM.
// Mirrors the reference's `struct MADtree` (mad_icp/src/tools/mad_tree.h:47-102) at the granularity the
// callers use it: build from a cloud, getLeafs, applyTransform, bestMatchingLeafFast — the latter only ever
// runs on the GPU (madicp_nn_search).
#pragma once
#include <cstdint>
#include <vector>

#include "linalg.h"
#include "tree_builder.h"
#include "types.h"

namespace madicp_host {

struct LeafMatch {  // what MADtreeWrapper::search* hands back (mad_tree_wrapper.h:42-67)
  Vector3d point;
  Vector3d normal;
  double dist;
  uint32_t leaf_idx;  // getLeafs() ordinal of the matched leaf
};

class MADtree {
 public:
  // build (mad_tree.cpp:35-130); the cloud is taken by value and permuted, as in the reference
  MADtree(ContainerType cloud, double b_max, double b_min, int max_parallel_level);
  // adopt a tree that was built elsewhere (Pipeline::prefetch builds the next scan's tree on another thread)
  explicit MADtree(LinearTree&& built);
  ~MADtree();
  MADtree(const MADtree&) = delete;
  MADtree& operator=(const MADtree&) = delete;

  int numLeaves() const { return tree_.num_leaves(); }
  int numNodes() const { return tree_.num_nodes(); }
  const madicp_node& leaf(int leaf_id) { return linear().nodes[tree_.leaf_nodes[leaf_id]]; }
  const LinearTree& linear();       // host copy, with any pending transform applied
  ContainerType leafMeans();        // getLeafs() order (mad_tree.cpp:154-163)

  // mad_tree.cpp:165-172.  The resident copy is transformed on the device at once (stream-ordered, no host wait);
  // the host copy lazily — it is only read again by leafMeans() / search() result decoding (bit-identical results).
  void applyTransform(const double* R, const double* t);

  // batched bestMatchingLeafFast on the device (mad_tree.cpp:144-152)
  std::vector<LeafMatch> search(const ContainerType& queries, bool with_dist);

  int deviceId();  // uploads on first use (asynchronous: staged, then copied on the context's copy stream)
  bool resident() const { return dev_id_ >= 0; }

 private:
  void flushTransform();
  LinearTree tree_;
  int dev_id_ = -1;
  unsigned dev_gen_ = 0;  // context generation dev_id_ belongs to
  bool pending_ = false;  // host copy still to be transformed by pending_R_, pending_t_
  double pending_R_[9], pending_t_[3];
};

}  // namespace madicp_host
