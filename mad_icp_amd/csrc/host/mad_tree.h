// MADtree — host-side owner of one MAD-tree: the linear node array on the host and its resident copy in HBM.
// Mirrors the reference's `struct MADtree` (mad_icp/src/tools/mad_tree.h:47-102) at the granularity the
// callers use it: build from a cloud, getLeafs, applyTransform, bestMatchingLeafFast — the latter only ever
// runs on the GPU (madicp_nn_search).
#pragma once
#include <cstdint>
#include <memory>
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
  // build ON THE DEVICE from a cloud that is already resident there (madicp_cloud_*); the host keeps
  // no copy of the tree unless somebody asks for one (linear(), leafMeans(), search result decoding download it)
  struct DeviceCloud {
    int cloud_id;
  };
  MADtree(DeviceCloud cloud, double b_max, double b_min);
  // ... as a look-ahead (madicp_tree_build_begin / _end): beginDeviceBuild() copies the scan and starts its construction on
  // the library's build stream and returns a ticket (0: the one slot of the process-wide device context is taken);
  // collectDeviceBuild(ticket) waits for it and returns the tree — or null when somebody cancelled it in between;
  // cancelDeviceBuild(ticket) drops it (ticket 0: whatever is in flight — what a synchronous device build does first, the
  // builder's scratch has one owner at a time)
  static unsigned beginDeviceBuild(const ContainerType& cloud, double b_max, double b_min);
  static std::unique_ptr<MADtree> collectDeviceBuild(unsigned ticket);
  static void cancelDeviceBuild(unsigned ticket);
  ~MADtree();
  MADtree(const MADtree&) = delete;
  MADtree& operator=(const MADtree&) = delete;

  int numLeaves() const { return n_leaves_; }
  int numNodes() const { return n_nodes_; }
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
  MADtree() = default;    // (collectDeviceBuild fills one in)
  void flushTransform();
  void fetchHostCopy();   // device-built trees: download the node array on first host-side use
  LinearTree tree_;
  bool host_copy_ = true; // tree_ holds the nodes (false: device-built and not downloaded yet)
  int n_leaves_ = 0, n_nodes_ = 0;
  int dev_id_ = -1;
  unsigned dev_gen_ = 0;  // context generation dev_id_ belongs to
  bool pending_ = false;  // host copy still to be transformed by pending_R_, pending_t_
  double pending_R_[9], pending_t_[3];
};

}  // namespace madicp_host
