#include "vel_estimator.h"

#include <cmath>
#include <cstring>

namespace madicp_host {

static constexpr double E_THRESHOLD_VEL = 0.3162;  // tools/constants.h:33

VelEstimator::VelEstimator(double sensor_hz) : ts_(1. / sensor_hz) {
  std::memset(X_, 0, sizeof(X_));
  std::memset(H_adder_, 0, sizeof(H_adder_));
  std::memset(b_adder_, 0, sizeof(b_adder_));
}

void VelEstimator::init(const double* velocity6) { std::memcpy(X_, velocity6, sizeof(X_)); }

// vel_estimator.cpp:45-79: error = predicted relative motion minus measured one (translation, xyz Euler
// angles read off the relative rotation); Jacobian = delta_t * I, so only the diagonal of H is touched.
void VelEstimator::update(const Pose& T_now, const Pose& T_prev, double delta_t, double weight) {
  const Pose rel = compose(inverse(T_prev), T_now);
  double e[6];
  for (int i = 0; i < 3; ++i) e[i] = delta_t * X_[i] - rel.t[i];
  const double* L = rel.R;
  const double angles[3] = {std::atan2(-L[5], L[8]), std::asin(L[2]), std::atan2(-L[1], L[0])};
  for (int i = 0; i < 3; ++i) e[3 + i] = delta_t * X_[3 + i] - angles[i];

  // squaredNorm of a contiguous 6-vector: three packets of two summed first, lanes added last
  const double lane0 = e[0] * e[0] + (e[2] * e[2] + e[4] * e[4]);
  const double lane1 = e[1] * e[1] + (e[3] * e[3] + e[5] * e[5]);
  const double chi = std::sqrt(lane0 + lane1);
  double scale = 1.;
  if (chi > E_THRESHOLD_VEL) scale = E_THRESHOLD_VEL / chi;
  const double swj = (scale * weight) * delta_t;
  for (int i = 0; i < 6; ++i) {
    H_adder_[i * 6 + i] += swj * delta_t;
    b_adder_[i] += swj * e[i];
  }
}

// vel_estimator.cpp:81-97
void VelEstimator::oneRound() {
  std::memset(H_adder_, 0, sizeof(H_adder_));
  std::memset(b_adder_, 0, sizeof(b_adder_));
  const size_t n = odometry_.size();
  const Pose T_now = odometry_.back();
  for (size_t i = 0; i + 1 < n; ++i) {
    const double delta_t = (n - 1 - i) * ts_;
    const double weight = 1.f - double(n - 2 - i) / double(n - 1);
    update(T_now, odometry_[i], delta_t, weight);
  }
  double nb[6], dx[6];
  for (int i = 0; i < 6; ++i) nb[i] = -b_adder_[i];
  ldlt6_solve(H_adder_, nb, dx);
  for (int i = 0; i < 6; ++i) X_[i] += dx[i];
}

}  // namespace madicp_host
