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
};

class MADtree {
 public:
  // build (mad_tree.cpp:35-130); the cloud is taken by value and permuted, as in the reference
  MADtree(ContainerType cloud, double b_max, double b_min, int max_parallel_level);
  ~MADtree();
  MADtree(const MADtree&) = delete;
  MADtree& operator=(const MADtree&) = delete;

  int numLeaves() const { return tree_.num_leaves(); }
  int numNodes() const { return tree_.num_nodes(); }
  const madicp_node& leaf(int leaf_id) const { return tree_.nodes[tree_.leaf_nodes[leaf_id]]; }
  const LinearTree& linear() const { return tree_; }
  ContainerType leafMeans() const;  // getLeafs() order (mad_tree.cpp:154-163)

  // mad_tree.cpp:165-172 — host copy and, if resident, the device copy (bit-identical results)
  void applyTransform(const double* R, const double* t);

  // batched bestMatchingLeafFast on the device (mad_tree.cpp:144-152)
  std::vector<LeafMatch> search(const ContainerType& queries, bool with_dist);

  int deviceId();  // uploads on first use
  bool resident() const { return dev_id_ >= 0; }

 private:
  LinearTree tree_;
  int dev_id_ = -1;
};

}  // namespace madicp_host
