#include "mad_tree.h"

#include <cstdlib>
#include <cstring>
#include <mutex>

#include "device.h"

namespace madicp_host {

namespace {
std::recursive_mutex g_mu;
madicp_ctx* g_ctx = nullptr;
unsigned g_gen = 1;
}  // namespace

std::recursive_mutex& Device::mutex() { return g_mu; }

madicp_ctx* Device::ctx() {
  DeviceLock lock(g_mu);
  if (!g_ctx) {
    int dev = 0;
    if (const char* e = std::getenv("MAD_ICP_DEVICE")) dev = std::atoi(e);
    check(madicp_ctx_create(dev, nullptr, &g_ctx), "madicp_ctx_create (the HIP path has no CPU fallback)");
#ifdef MADICP_MEASURE
    // MADICP_PUBLISH_SIDE=0|1 (probes: tools/lookahead_matrix.sh): the library option of the same name for the shared context
    // (measurement aids: only in the measurement build, -DMADICP_MEASURE)
    if (const char* e = std::getenv("MADICP_PUBLISH_SIDE")) madicp_ctx_set_option(g_ctx, "publish_side", e[0] == '1' ? 1 : 0);
    if (const char* e = std::getenv("MADICP_BUILD_AFTER_REG")) madicp_ctx_set_option(g_ctx, "build_after_registration", e[0] == '1' ? 1 : 0);
#endif
  }
  return g_ctx;
}

madicp_ctx* Device::current(unsigned generation) {
  DeviceLock lock(g_mu);
  return generation == g_gen ? g_ctx : nullptr;
}

unsigned Device::generation() {
  DeviceLock lock(g_mu);
  return g_gen;
}

void Device::shutdown() {
  DeviceLock lock(g_mu);
  if (g_ctx) madicp_ctx_destroy(g_ctx);
  g_ctx = nullptr;
  ++g_gen;
}

MADtree::MADtree(ContainerType cloud, double b_max, double b_min, int max_parallel_level) {
  if (cloud.empty()) throw std::invalid_argument("MADtree: empty cloud");
  tree_ = build_tree(cloud.front().data(), static_cast<int64_t>(cloud.size()), b_max, b_min, max_parallel_level);
  n_leaves_ = tree_.num_leaves();
  n_nodes_ = tree_.num_nodes();
}

MADtree::MADtree(DeviceCloud cloud, double b_max, double b_min) {
  DeviceLock lock(Device::mutex());
  madicp_ctx* c = Device::ctx();
  cancelDeviceBuild(0);  // (a look-ahead of another Pipeline: its owner finds its ticket stale and builds when its scan comes)
  int32_t leaves = 0;
  check(madicp_tree_build(c, cloud.cloud_id, b_max, b_min, &dev_id_, &leaves), "madicp_tree_build");
  dev_gen_ = Device::generation();
  n_leaves_ = leaves;
  n_nodes_ = 2 * leaves - 1;
  host_copy_ = false;
}

namespace {
unsigned g_lookahead = 0;       // ticket of the construction in flight on the context's build stream (0: none)
unsigned g_lookahead_next = 1;
unsigned g_lookahead_gen = 0;   // context generation it belongs to
}  // namespace

unsigned MADtree::beginDeviceBuild(const ContainerType& cloud, double b_max, double b_min) {
  if (cloud.empty()) throw std::invalid_argument("MADtree: empty cloud");
  DeviceLock lock(Device::mutex());
  madicp_ctx* c = Device::ctx();
  if (g_lookahead && g_lookahead_gen != Device::generation()) g_lookahead = 0;  // (went down with its context)
  if (g_lookahead) return 0;  // another Pipeline of this process is looking ahead: build when the scan comes
  const int rc = madicp_tree_build_begin(c, cloud.front().data(), static_cast<int64_t>(cloud.size()), b_max, b_min);
  if (rc == MADICP_ERR_CAPACITY) return 0;
  check(rc, "madicp_tree_build_begin");
  g_lookahead = g_lookahead_next++;
  if (g_lookahead_next == 0) g_lookahead_next = 1;
  g_lookahead_gen = Device::generation();
  return g_lookahead;
}

void MADtree::cancelDeviceBuild(unsigned ticket) {
  DeviceLock lock(Device::mutex());
  if (!g_lookahead || (ticket && ticket != g_lookahead)) return;
  g_lookahead = 0;
  if (madicp_ctx* c = Device::current(g_lookahead_gen)) madicp_tree_build_cancel(c);
}

std::unique_ptr<MADtree> MADtree::collectDeviceBuild(unsigned ticket) {
  DeviceLock lock(Device::mutex());
  if (!ticket || ticket != g_lookahead) return nullptr;  // cancelled in between (a synchronous build needed the scratch)
  g_lookahead = 0;
  madicp_ctx* c = Device::current(g_lookahead_gen);
  if (!c) return nullptr;
  std::unique_ptr<MADtree> t(new MADtree());
  int32_t leaves = 0;
  check(madicp_tree_build_end(c, &t->dev_id_, &leaves), "madicp_tree_build_end");
  t->dev_gen_ = g_lookahead_gen;
  t->n_leaves_ = leaves;
  t->n_nodes_ = 2 * leaves - 1;
  t->host_copy_ = false;
  return t;
}

MADtree::MADtree(LinearTree&& built) : tree_(std::move(built)) {
  if (tree_.nodes.empty()) throw std::invalid_argument("MADtree: empty tree");
  n_leaves_ = tree_.num_leaves();
  n_nodes_ = tree_.num_nodes();
}

MADtree::~MADtree() {
  // never create a context from a destructor, never release an id into a context that did not issue it
  if (dev_id_ < 0) return;
  DeviceLock lock(Device::mutex());
  if (madicp_ctx* c = Device::current(dev_gen_)) madicp_tree_release(c, dev_id_);
}

void MADtree::fetchHostCopy() {
  if (host_copy_) return;
  DeviceLock lock(Device::mutex());
  madicp_ctx* c = Device::current(dev_gen_);
  if (!c || dev_id_ < 0) throw std::runtime_error("MADtree: the device context that built this tree is gone");
  tree_.nodes.resize(static_cast<size_t>(n_nodes_));
  check(madicp_tree_download(c, dev_id_, tree_.nodes.data(), n_nodes_), "madicp_tree_download");
  tree_.leaf_nodes.resize(static_cast<size_t>(n_leaves_));
  for (int32_t i = 0; i < n_nodes_; ++i)
    if (tree_.nodes[i].right == 0) tree_.leaf_nodes[tree_.nodes[i].leaf_id] = i;
  host_copy_ = true;
  pending_ = false;  // the download is the tree as it stands on the device, transforms included
}

void MADtree::flushTransform() {
  fetchHostCopy();
  if (!pending_) return;
  transform_tree(tree_, pending_R_, pending_t_);
  pending_ = false;
}

const LinearTree& MADtree::linear() {
  flushTransform();
  return tree_;
}

ContainerType MADtree::leafMeans() {
  flushTransform();
  ContainerType out(tree_.leaf_nodes.size());
  for (size_t i = 0; i < out.size(); ++i) std::memcpy(out[i].data(), tree_.nodes[tree_.leaf_nodes[i]].mean, 24);
  return out;
}

int MADtree::deviceId() {
  DeviceLock lock(Device::mutex());
  if (dev_id_ >= 0 && !Device::current(dev_gen_)) dev_id_ = -1;  // the context that held it is gone
  if (dev_id_ < 0) {
    if (!host_copy_) throw std::runtime_error("MADtree: device-built tree lost with its context");
    flushTransform();
    madicp_ctx* c = Device::ctx();
    // a tree from this library's own builder needs no structure check; anything else (a downloaded / caller-made
    // array: rho2 < 0) goes through the validating entry
    if (tree_.rho2 >= 0.0)
      check(madicp_tree_upload_trusted(c, tree_.nodes.data(), tree_.num_nodes(), tree_.num_leaves(), tree_.rho2, &dev_id_),
            "madicp_tree_upload_trusted");
    else
      check(madicp_tree_upload(c, tree_.nodes.data(), tree_.num_nodes(), tree_.num_leaves(), &dev_id_), "madicp_tree_upload");
    dev_gen_ = Device::generation();
  }
  return dev_id_;
}

void MADtree::applyTransform(const double* R, const double* t) {
  DeviceLock lock(Device::mutex());
  const bool on_device = dev_id_ >= 0 && Device::current(dev_gen_);
  if (on_device) check(madicp_tree_transform(Device::ctx(), dev_id_, R, t), "madicp_tree_transform");
  if (!host_copy_) return;         // device-built, never downloaded: the device copy is the only one
  if (pending_) flushTransform();  // (a second transform before the first was needed on the host: compose by applying)
  if (on_device) {
    std::memcpy(pending_R_, R, sizeof(pending_R_));
    std::memcpy(pending_t_, t, sizeof(pending_t_));
    pending_ = true;
  } else {
    transform_tree(tree_, R, t);
  }
}

std::vector<LeafMatch> MADtree::search(const ContainerType& queries, bool with_dist) {
  std::vector<LeafMatch> out(queries.size());
  if (queries.empty()) return out;
  std::vector<uint32_t> node(queries.size()), leaf(queries.size());
  std::vector<double> dist(with_dist ? queries.size() : 0);
  {
    DeviceLock lock(Device::mutex());
    const int id = deviceId();
    check(madicp_nn_search(Device::ctx(), id, queries.front().data(), static_cast<int64_t>(queries.size()), leaf.data(),
                           node.data(), with_dist ? dist.data() : nullptr, nullptr),
          "madicp_nn_search");
  }
  flushTransform();
  for (size_t i = 0; i < out.size(); ++i) {
    const madicp_node& n = tree_.nodes[node[i]];
    std::memcpy(out[i].point.data(), n.mean, 24);
    std::memcpy(out[i].normal.data(), n.dir, 24);
    out[i].dist = with_dist ? dist[i] : 0.0;
    out[i].leaf_idx = leaf[i];
  }
  return out;
}

}  // namespace madicp_host
