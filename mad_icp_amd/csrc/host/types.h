// Public value types of the host C++ layer.
//
// The reference's surface is written against Eigen (ContainerType = std::vector<Eigen::Vector3d>,
// Eigen::Matrix4d poses: mad_tree.h:42, pipeline.h:60-62).  Where Eigen is installed these aliases ARE the
// Eigen types, so the reference's bin_runner.cpp compiles against this Pipeline unchanged; where it is not
// (this build image) they are layout-identical PODs: 3 contiguous doubles, and a column-major 4x4.
// Nothing in the implementation uses Eigen arithmetic either way.
#pragma once
#include <cstddef>
#include <vector>

#if __has_include(<Eigen/Core>)
#include <Eigen/Core>
namespace madicp_host {
using Vector3d = Eigen::Vector3d;
using Matrix4d = Eigen::Matrix4d;
}  // namespace madicp_host
#else
namespace madicp_host {
struct Vector3d {
  double v[3];
  double& operator[](std::size_t i) { return v[i]; }
  const double& operator[](std::size_t i) const { return v[i]; }
  double& operator()(std::size_t i) { return v[i]; }
  const double& operator()(std::size_t i) const { return v[i]; }
  double* data() { return v; }
  const double* data() const { return v; }
  bool operator==(const Vector3d& o) const { return v[0] == o.v[0] && v[1] == o.v[1] && v[2] == o.v[2]; }
};
struct Matrix4d {  // column-major, like Eigen::Matrix4d
  double m[16];
  double& operator()(std::size_t r, std::size_t c) { return m[c * 4 + r]; }
  const double& operator()(std::size_t r, std::size_t c) const { return m[c * 4 + r]; }
  double* data() { return m; }
  const double* data() const { return m; }
};
}  // namespace madicp_host
#endif

namespace madicp_host {
static_assert(sizeof(Vector3d) == 24, "Vector3d must be 3 contiguous doubles (eigen_stl_bindings.h:73-81)");
using ContainerType = std::vector<Vector3d>;  // mad_tree.h:42
}  // namespace madicp_host
