#include "mad_icp.h"

#include <cstring>

#include "device.h"

namespace madicp_host {

MADicp::MADicp(double min_ball, double rho_ker, double b_ratio, int num_threads)
  : rho_ker_(rho_ker), min_ball_(min_ball), b_ratio_(b_ratio), num_threads_(num_threads) {
  X_ = Pose::identity();
  std::memset(H_adder_, 0, sizeof(H_adder_));
  std::memset(b_adder_, 0, sizeof(b_adder_));
}

MADicp::~MADicp() { releaseMoving(); }

void MADicp::releaseMoving() {
  if (moving_id_ >= 0) madicp_moving_release(Device::ctx(), moving_id_);
  moving_id_ = -1;
  L_ = 0;
}

void MADicp::setMoving(const ContainerType& leaf_means) {
  releaseMoving();
  if (leaf_means.empty()) throw std::invalid_argument("MADicp::setMoving: no leaves");
  check(madicp_moving_upload(Device::ctx(), leaf_means.front().data(), static_cast<int32_t>(leaf_means.size()),
                             &moving_id_),
        "madicp_moving_upload");
  L_ = static_cast<int>(leaf_means.size());
  matched_.assign(L_, 0);
}

void MADicp::setMoving(const MADtree& scan_tree) { setMoving(scan_tree.leafMeans()); }

void MADicp::compute(const std::vector<MADtree*>& fixed, int n_iters) {
  if (moving_id_ < 0) throw std::runtime_error("MADicp::compute: setMoving was not called");
  if (fixed.empty()) throw std::runtime_error("MADicp::compute: no fixed tree");
  if (n_iters < 1) return;
  std::vector<int> ids;
  ids.reserve(fixed.size());
  for (MADtree* t : fixed) ids.push_back(t->deviceId());
  double X[12];
  std::memcpy(X, X_.R, sizeof(X_.R));
  std::memcpy(X + 9, X_.t, sizeof(X_.t));
  const madicp_icp_params p{min_ball_, rho_ker_, b_ratio_};
  check(madicp_icp_register(Device::ctx(), moving_id_, ids.data(), static_cast<int>(ids.size()), X, &p, n_iters,
                            H_adder_, b_adder_, matched_.data(), nullptr, &visits_),
        "madicp_icp_register");
  std::memcpy(X_.R, X, sizeof(X_.R));
  std::memcpy(X_.t, X + 9, sizeof(X_.t));
}

int MADicp::numMatched() const {
  int n = 0;
  for (uint8_t m : matched_) n += m ? 1 : 0;
  return n;
}

}  // namespace madicp_host
