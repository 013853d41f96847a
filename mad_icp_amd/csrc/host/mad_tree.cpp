#include "mad_tree.h"

#include <cstdlib>
#include <cstring>
#include <mutex>

#include "device.h"

namespace madicp_host {

namespace {
std::mutex g_mu;
madicp_ctx* g_ctx = nullptr;
}  // namespace

madicp_ctx* Device::ctx() {
  std::lock_guard<std::mutex> lock(g_mu);
  if (!g_ctx) {
    int dev = 0;
    if (const char* e = std::getenv("MAD_ICP_DEVICE")) dev = std::atoi(e);
    check(madicp_ctx_create(dev, nullptr, &g_ctx), "madicp_ctx_create (the HIP path has no CPU fallback)");
  }
  return g_ctx;
}

void Device::shutdown() {
  std::lock_guard<std::mutex> lock(g_mu);
  if (g_ctx) madicp_ctx_destroy(g_ctx);
  g_ctx = nullptr;
}

MADtree::MADtree(ContainerType cloud, double b_max, double b_min, int max_parallel_level) {
  if (cloud.empty()) throw std::invalid_argument("MADtree: empty cloud");
  tree_ = build_tree(cloud.front().data(), static_cast<int64_t>(cloud.size()), b_max, b_min, max_parallel_level);
}

MADtree::~MADtree() {
  if (dev_id_ >= 0) madicp_tree_release(Device::ctx(), dev_id_);
}

ContainerType MADtree::leafMeans() const {
  ContainerType out(tree_.leaf_nodes.size());
  for (size_t i = 0; i < out.size(); ++i) std::memcpy(out[i].data(), tree_.nodes[tree_.leaf_nodes[i]].mean, 24);
  return out;
}

int MADtree::deviceId() {
  if (dev_id_ < 0)
    check(madicp_tree_upload(Device::ctx(), tree_.nodes.data(), tree_.num_nodes(), tree_.num_leaves(), &dev_id_),
          "madicp_tree_upload");
  return dev_id_;
}

void MADtree::applyTransform(const double* R, const double* t) {
  transform_tree(tree_, R, t);
  if (dev_id_ >= 0) check(madicp_tree_transform(Device::ctx(), dev_id_, R, t), "madicp_tree_transform");
}

std::vector<LeafMatch> MADtree::search(const ContainerType& queries, bool with_dist) {
  std::vector<LeafMatch> out(queries.size());
  if (queries.empty()) return out;
  std::vector<uint32_t> node(queries.size());
  std::vector<double> dist(with_dist ? queries.size() : 0);
  check(madicp_nn_search(Device::ctx(), deviceId(), queries.front().data(), static_cast<int64_t>(queries.size()), nullptr,
                         node.data(), with_dist ? dist.data() : nullptr, nullptr),
        "madicp_nn_search");
  for (size_t i = 0; i < out.size(); ++i) {
    const madicp_node& n = tree_.nodes[node[i]];
    std::memcpy(out[i].point.data(), n.mean, 24);
    std::memcpy(out[i].normal.data(), n.dir, 24);
    out[i].dist = with_dist ? dist[i] : 0.0;
  }
  return out;
}

}  // namespace madicp_host
