#include "deskew.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <functional>
#include <utility>

#include "pipeline.h"  // CHUNKS
#include "task_pool.h"

namespace madicp_host {

void naive_velocity(const Pose& T_prev, const Pose& T_now, double sensor_hz, double* vel) {
  const double ts = 1. / sensor_hz;
  const Pose rel = compose(inverse(T_prev), T_now);
  double w[3];
  log_so3(rel.R, w);
  for (int i = 0; i < 3; ++i) {
    vel[i] = rel.t[i] / ts;
    vel[3 + i] = w[i] / ts;
  }
}

namespace {

// merge of two sorted index runs a[0..na), b[0..nb) into out, by azimuth; a's elements first among equals (irrelevant for the
// result that is used: equal keys set `ties` and the order is discarded)
void merge_runs(const double* az, const uint32_t* a, size_t na, const uint32_t* b, size_t nb, uint32_t* out) {
  std::merge(a, a + na, b, b + nb, out, [az](uint32_t x, uint32_t y) { return az[x] < az[y]; });
}

// the same merge cut into `parts` independent pieces (co-ranking by binary search on the longer run's quantiles)
void merge_runs_parallel(TaskPool& pool, const double* az, const uint32_t* a, size_t na, const uint32_t* b, size_t nb, uint32_t* out,
                         int parts) {
  if (parts <= 1 || na + nb < 16384) {
    merge_runs(az, a, na, b, nb, out);
    return;
  }
  auto less = [az](uint32_t x, uint32_t y) { return az[x] < az[y]; };
  std::vector<std::function<void()>> fns;
  size_t a0 = 0, b0 = 0;
  for (int p = 1; p <= parts; ++p) {
    size_t a1, b1;
    if (p == parts) {
      a1 = na;
      b1 = nb;
    } else {
      a1 = na * static_cast<size_t>(p) / static_cast<size_t>(parts);
      // everything of b that sorts strictly before a[a1] goes with the earlier piece (a first among equals)
      b1 = a1 < na ? static_cast<size_t>(std::lower_bound(b, b + nb, a[a1], less) - b) : nb;
      if (b1 < b0) b1 = b0;
    }
    const size_t o = a0 + b0;
    fns.emplace_back([=] { merge_runs(az, a + a0, a1 - a0, b + b0, b1 - b0, out + o); });
    a0 = a1;
    b0 = b1;
  }
  for (const TaskPool::Handle& j : pool.submit_batch(std::move(fns))) pool.wait(j);
}

}  // namespace

DeskewOrder deskew_order(const ContainerType& cloud) {
  DeskewOrder d;
  const size_t n = cloud.size();
  d.azimuth.resize(n);
  d.order.resize(n);
  if (n == 0) return d;
  TaskPool& pool = TaskPool::instance();
  const int runs = static_cast<int>(std::max<size_t>(1, std::min<size_t>(16, n / 4096)));
  std::vector<size_t> cut(static_cast<size_t>(runs) + 1);
  for (int r = 0; r <= runs; ++r) cut[static_cast<size_t>(r)] = n * static_cast<size_t>(r) / static_cast<size_t>(runs);
  std::vector<uint32_t> other(n);
  double* az = d.azimuth.data();
  uint32_t* buf[2] = {d.order.data(), other.data()};
  std::atomic<bool> run_tie{false};
  {  // azimuths and sorted runs
    std::vector<std::function<void()>> fns;
    for (int r = 0; r < runs; ++r) {
      const size_t lo = cut[static_cast<size_t>(r)], hi = cut[static_cast<size_t>(r) + 1];
      fns.emplace_back([&cloud, &run_tie, az, lo, hi, out = buf[0]] {
        for (size_t i = lo; i < hi; ++i) {
          az[i] = std::atan2(cloud[i][1], cloud[i][0]);  // pipeline.cpp:93
          out[i] = static_cast<uint32_t>(i);
        }
        std::sort(out + lo, out + hi, [az](uint32_t x, uint32_t y) { return az[x] < az[y]; });
        bool tie = false;  // a tie inside a run is a tie of the scan: the merges would be wasted work
        for (size_t i = lo; i + 1 < hi; ++i) tie |= !(az[out[i]] < az[out[i + 1]]);
        if (tie) run_tie.store(true, std::memory_order_relaxed);
      });
    }
    for (const TaskPool::Handle& j : pool.submit_batch(std::move(fns))) pool.wait(j);
  }
  if (run_tie.load(std::memory_order_relaxed)) {  // (the order is not used with ties: deskew_cloud takes the serial route)
    d.ties = true;
    return d;
  }
  int src = 0;
  for (int width = 1; width < runs; width *= 2) {  // pairwise merges, ping-pong between the two buffers
    std::vector<std::function<void()>> fns;
    const int pairs = (runs + 2 * width - 1) / (2 * width);
    const int parts = std::max(1, 16 / pairs);  // the fewer merges a level has, the more pieces each is cut into
    for (int r = 0; r < runs; r += 2 * width) {
      const size_t lo = cut[static_cast<size_t>(r)], mid = cut[static_cast<size_t>(std::min(r + width, runs))],
                   hi = cut[static_cast<size_t>(std::min(r + 2 * width, runs))];
      const uint32_t* s = buf[src];
      uint32_t* o = buf[src ^ 1];
      fns.emplace_back([&pool, az, s, o, lo, mid, hi, parts] {
        if (mid == hi)
          std::copy(s + lo, s + hi, o + lo);
        else
          merge_runs_parallel(pool, az, s + lo, mid - lo, s + mid, hi - mid, o + lo, parts);
      });
    }
    for (const TaskPool::Handle& j : pool.submit_batch(std::move(fns))) pool.wait(j);
    src ^= 1;
  }
  if (src == 1) d.order.swap(other);
  const uint32_t* ord = d.order.data();
  bool ties = false;
  for (size_t i = 0; i + 1 < n; ++i) ties |= !(az[ord[i]] < az[ord[i + 1]]);  // equal, or unordered (NaN)
  if (n == 1) ties = std::isnan(az[0]);
  d.ties = ties;
  return d;
}

void deskew_cloud(ContainerType& cloud, const Pose& T_prev, const Pose& T_now, double sensor_hz, const DeskewOrder* prep,
                  double* out_velocity6) {
  const double ts = 1. / sensor_hz;
  double vel[6];
  naive_velocity(T_prev, T_now, sensor_hz, vel);
  if (out_velocity6)
    for (int i = 0; i < 6; ++i) out_velocity6[i] = vel[i];
  const size_t n = cloud.size();
  DeskewOrder local;
  if (!prep || prep->order.size() != n) {
    local = deskew_order(cloud);
    prep = &local;
  }
  TaskPool& pool = TaskPool::instance();
  const int pieces = static_cast<int>(std::max<size_t>(1, std::min<size_t>(16, n / 8192)));
  auto in_pieces = [&](const std::function<void(size_t, size_t)>& body) {
    if (pieces == 1) {
      body(0, n);
      return;
    }
    std::vector<std::function<void()>> fns;
    for (int p = 0; p < pieces; ++p) {
      const size_t lo = n * static_cast<size_t>(p) / static_cast<size_t>(pieces), hi = n * static_cast<size_t>(p + 1) / static_cast<size_t>(pieces);
      fns.emplace_back([&body, lo, hi] { body(lo, hi); });
    }
    for (const TaskPool::Handle& j : pool.submit_batch(std::move(fns))) pool.wait(j);
  };
  // the points and their azimuths in ascending azimuth order
  std::vector<double> key(n);
  ContainerType pts(n);
  if (!prep->ties) {
    const DeskewOrder& d = *prep;
    in_pieces([&](size_t lo, size_t hi) {
      for (size_t i = lo; i < hi; ++i) {
        key[i] = d.azimuth[d.order[i]];
        pts[i] = cloud[d.order[i]];
      }
    });
  } else {  // the reference's own route (pipeline.cpp:88-97): only std::sort knows what it does with equal keys
    using AzimuthPair = std::pair<double, Vector3d>;
    std::vector<AzimuthPair> sorted(n);
    for (size_t i = 0; i < n; ++i) sorted[i] = std::make_pair(prep->azimuth[i], cloud[i]);
    std::sort(sorted.begin(), sorted.end(), [](const AzimuthPair& a, const AzimuthPair& b) -> bool { return a.first < b.first; });
    for (size_t i = 0; i < n; ++i) {
      key[i] = sorted[i].first;
      pts[i] = sorted[i].second;
    }
  }
  // pipeline.cpp:99-122.  The walk from the largest azimuth down is a running state (threshold, time: the reference's own
  // repeated subtraction / addition) — one serial pass over the keys that only records WHICH pose a point gets and tabulates
  // the poses in the order they arise; applying them is independent per point.
  const double resolution = 2 * M_PI / double(CHUNKS);
  const double delta = ts / double(CHUNKS - 1);
  double t = -ts;
  auto pose_at = [&](double tt) {
    const double dx[6] = {vel[0] * tt, vel[1] * tt, vel[2] * tt, vel[3] * tt, vel[4] * tt, vel[5] * tt};
    return motion_from_twist(dx);
  };
  std::vector<Pose> poses;
  poses.reserve(CHUNKS + 2);
  poses.push_back(pose_at(t));
  std::vector<uint32_t> which(n);
  double angle = M_PI - resolution;
  for (long i = static_cast<long>(n) - 1; i >= 0; --i) {
    if (key[static_cast<size_t>(i)] < angle) {
      angle -= resolution;
      t += delta;
      poses.push_back(pose_at(t));
    }
    which[static_cast<size_t>(i)] = static_cast<uint32_t>(poses.size() - 1);
  }
  in_pieces([&](size_t lo, size_t hi) {
    for (size_t i = lo; i < hi; ++i) apply(poses[which[i]], pts[i].data(), cloud[i].data());
  });
}

}  // namespace madicp_host
