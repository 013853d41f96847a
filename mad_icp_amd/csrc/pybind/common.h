// Shared pieces of the four pybind11 modules (pyvector, pymadtree, pymadicp, pypeline): the opaque
// VectorEigen3d class and the numpy converters that stand in for <pybind11/eigen.h> (Eigen is not required).
//
// VectorEigen3d mirrors the reference binding (mad_icp/src/pybind/eigen_stl_bindings.h:25-97): an opaque
// std::vector of 3-double points, constructible from a C-contiguous (N,3) float64 array (forcecast; anything
// else raises cast_error), exposing the buffer protocol with shape (N,3) / strides (24,8), list-like
// modifiers and accessors, __len__, __bool__, __repr__, __copy__, __deepcopy__.
#pragma once
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>
#include <pybind11/stl_bind.h>

#include <cstring>
#include <string>
#include <vector>

#include "types.h"

namespace py = pybind11;
using madicp_host::ContainerType;
using madicp_host::Matrix4d;
using madicp_host::Vector3d;

#if !__has_include(<Eigen/Core>)
// element caster: a point crosses to Python as a float64 array of shape (3,), like an Eigen::Vector3d would
namespace pybind11 {
namespace detail {
template <>
struct type_caster<Vector3d> {
 public:
  PYBIND11_TYPE_CASTER(Vector3d, const_name("numpy.ndarray[numpy.float64[3]]"));
  bool load(handle src, bool convert) {
    if (!convert && !array_t<double>::check_(src)) return false;
    auto arr = array_t<double, array::c_style | array::forcecast>::ensure(src);
    if (!arr || arr.size() != 3) return false;
    std::memcpy(value.v, arr.data(), 24);
    return true;
  }
  static handle cast(const Vector3d& v, return_value_policy, handle) {
    array_t<double> a(3);
    std::memcpy(a.mutable_data(), v.v, 24);
    return a.release();
  }
};
template <>
struct type_caster<Matrix4d> {
 public:
  PYBIND11_TYPE_CASTER(Matrix4d, const_name("numpy.ndarray[numpy.float64[4, 4]]"));
  bool load(handle src, bool convert) {
    if (!convert && !array_t<double>::check_(src)) return false;
    auto arr = array_t<double, array::c_style | array::forcecast>::ensure(src);
    if (!arr || arr.ndim() != 2 || arr.shape(0) != 4 || arr.shape(1) != 4) return false;
    for (int r = 0; r < 4; ++r)
      for (int c = 0; c < 4; ++c) value(r, c) = arr.at(r, c);
    return true;
  }
  static handle cast(const Matrix4d& M, return_value_policy, handle) {
    array_t<double> a({4, 4});
    auto w = a.mutable_unchecked<2>();
    for (int r = 0; r < 4; ++r)
      for (int c = 0; c < 4; ++c) w(r, c) = M(r, c);
    return a.release();
  }
};
}  // namespace detail
}  // namespace pybind11
#else
#include <pybind11/eigen.h>
#endif

PYBIND11_MAKE_OPAQUE(std::vector<Vector3d>);

inline ContainerType container_from_array(py::array_t<double, py::array::c_style | py::array::forcecast> array) {
  if (array.ndim() != 2 || array.shape(1) != 3) throw py::cast_error();  // eigen_stl_bindings.h:48-50
  ContainerType out(static_cast<size_t>(array.shape(0)));
  if (!out.empty()) std::memcpy(out.front().data(), array.data(), out.size() * sizeof(Vector3d));
  return out;
}

// an (N,3) float64 C-contiguous array read in place as N points (same shape rule as container_from_array)
inline const Vector3d* points_of_array(const py::array_t<double, py::array::c_style | py::array::forcecast>& array) {
  static_assert(sizeof(Vector3d) == 3 * sizeof(double), "a point is three packed doubles");
  if (array.ndim() != 2 || array.shape(1) != 3) throw py::cast_error();
  return reinterpret_cast<const Vector3d*>(array.data());
}

template <typename... Extra>
inline void bind_vector_eigen3d(py::module_& m, const Extra&... extra) {
  // same recipe as the reference (eigen_stl_bindings.h:25-35,64-97): a plain class_ with the buffer protocol,
  // then pybind11's list-like vector helpers (bind_vector itself would demand a numpy dtype for the element)
  using Class_ = py::class_<ContainerType, std::unique_ptr<ContainerType>>;
  Class_ vec(m, "VectorEigen3d", py::buffer_protocol(), extra...);
  vec.def(py::init<>());
  vec.def("__bool__", [](const ContainerType& v) -> bool { return !v.empty(); }, "Check whether the list is nonempty");
  vec.def("__len__", &ContainerType::size);
  vec.def(py::init(&container_from_array));
  vec.def_buffer([](ContainerType& v) -> py::buffer_info {
    return py::buffer_info(v.empty() ? nullptr : v.front().data(), sizeof(double), py::format_descriptor<double>::format(), 2,
                           {v.size(), size_t(3)}, {sizeof(Vector3d), sizeof(double)});
  });
  vec.def("__repr__", [](const ContainerType& v) {
    return std::string("std::vector<Eigen::Vector3d> with ") + std::to_string(v.size()) +
           std::string(" elements.\nUse numpy.asarray() to access data.");
  });
  vec.def("__copy__", [](ContainerType& v) { return ContainerType(v); });
  vec.def("__deepcopy__", [](ContainerType& v, py::dict&) { return ContainerType(v); });
  py::detail::vector_if_copy_constructible<ContainerType, Class_>(vec);
  py::detail::vector_if_equal_operator<ContainerType, Class_>(vec);
  py::detail::vector_modifiers<ContainerType, Class_>(vec);
  py::detail::vector_accessor<ContainerType, Class_>(vec);
}
