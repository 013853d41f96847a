// pyvector — `VectorEigen3d` (reference: mad_icp/src/pybind/pyvector.cpp:51-54)
#include "common.h"

PYBIND11_MODULE(pyvector, m) {
  m.doc() = "mad_icp_amd: opaque (N,3) float64 point container, drop-in for mad_icp.src.pybind.pyvector";
  bind_vector_eigen3d(m);
}
