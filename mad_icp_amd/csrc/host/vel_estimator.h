// VelEstimator — 6-dof constant-velocity fit over the last poses, one Gauss-Newton round per frame.
// Mirrors mad_icp/src/odometry/vel_estimator.{h:39-58,cpp:32-97}.  O(10) 6x6 operations per frame: stays on
// the CPU (out of the GPU scope) but Pipeline::compute needs it.
#pragma once
#include <vector>

#include "linalg.h"

namespace madicp_host {

struct VelEstimator {
  explicit VelEstimator(double sensor_hz);
  void init(const double* velocity6);
  void setOdometry(const std::vector<Pose>& odometry) { odometry_ = odometry; }
  void oneRound();

  double X_[6];
  double H_adder_[36];
  double b_adder_[6];
  std::vector<Pose> odometry_;
  double ts_;

 private:
  void update(const Pose& T_now, const Pose& T_prev, double delta_t, double weight);
};

}  // namespace madicp_host
