// Pipeline::deskew (mad_icp/src/odometry/pipeline.cpp:79-123) on the host, split where the data dependence is: the azimuth of
// every point and the ascending order of the azimuths depend on the scan alone (:91-97 — atan2 per point, std::sort of
// (azimuth, point) pairs: 3-5 ms of one thread at 120 k points, most of a deskewed frame), the motion compensation on the two
// previous poses (:82-86, :99-122).  The first half is computed by the task pool — and ahead of time by Pipeline::prefetch —
// the second walks the sorted points once.
//
// Bit-identical to the reference by construction, not by luck: with DISTINCT azimuths the sorted order is unique, so any
// correct sort gives std::sort's permutation; when two azimuths compare equal (std::sort is not stable: the order it leaves
// among them is a property of its implementation and of the whole input) `ties` is set and deskew_cloud() takes the
// reference's own route — serial std::sort of the pairs — instead.
#pragma once
#include <cstdint>
#include <vector>

#include "linalg.h"
#include "types.h"

namespace madicp_host {

struct DeskewOrder {
  std::vector<double> azimuth;   // atan2(y, x), input order
  std::vector<uint32_t> order;   // order[i] = index of the point with the i-th smallest azimuth (valid when !ties)
  bool ties = false;             // two azimuths compare equal (or one is NaN): only std::sort itself knows its order
};

DeskewOrder deskew_order(const ContainerType& cloud);  // parallel (task pool)

// in place, output in azimuth order like the reference's; `prep` (optional): deskew_order(cloud) computed earlier
void deskew_cloud(ContainerType& cloud, const Pose& T_prev, const Pose& T_now, double sensor_hz, const DeskewOrder* prep,
                  double* out_velocity6 /* optional */);

// pipeline.cpp:82-86: [translation; logMapSO3(rotation)] of T_prev^-1 T_now over one scan period
void naive_velocity(const Pose& T_prev, const Pose& T_now, double sensor_hz, double* vel6);

}  // namespace madicp_host
