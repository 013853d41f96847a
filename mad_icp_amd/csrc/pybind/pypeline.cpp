// pypeline — `Pipeline` + `VectorEigen3d` (reference: mad_icp/src/pybind/pypeline.cpp:57-74).
#include "common.h"
#include "pipeline.h"

PYBIND11_MODULE(pypeline, m) {
  m.doc() = "mad_icp_amd: MAD-ICP odometry pipeline with the registration on MI355X, drop-in for mad_icp.src.pybind.pypeline";
  // the reference registers the same container in pyvector and pypeline; module_local lets both be imported
  bind_vector_eigen3d(m, py::module_local());
  py::class_<Pipeline>(m, "Pipeline")
    .def(py::init<double, bool, double, double, double, double, double, int, int, bool>(), py::arg("sensor_hz"),
         py::arg("deskew"), py::arg("b_max"), py::arg("rho_ker"), py::arg("p_th"), py::arg("b_min"), py::arg("b_ratio"),
         py::arg("num_keyframes"), py::arg("num_threads"), py::arg("realtime"))
    .def("currentPose", &Pipeline::currentPose)
    .def("trajectory", &Pipeline::trajectory)
    .def("keyframePose", &Pipeline::keyframePose)
    .def("isInitialized", &Pipeline::isInitialized)
    .def("isMapUpdated", &Pipeline::isMapUpdated)
    .def("currentID", &Pipeline::currentID)
    .def("keyframeID", &Pipeline::keyframeID)
    .def("modelLeaves", &Pipeline::modelLeaves)
    .def("currentLeaves", &Pipeline::currentLeaves)
    .def("compute", &Pipeline::compute)
    // additive: an (N,3) float64 array directly — one copy instead of the two of VectorEigen3d(points) + by-value call
    .def("compute",
         [](Pipeline& self, double stamp, py::array_t<double, py::array::c_style | py::array::forcecast> cloud) {
           self.compute(stamp, container_from_array(std::move(cloud)));
         })
    // additive look-ahead: start building the next scan's MAD-tree while this frame is registered
    .def("prefetch", &Pipeline::prefetch, py::arg("next_cloud"))
    .def("prefetch",
         [](Pipeline& self, py::array_t<double, py::array::c_style | py::array::forcecast> cloud) {
           self.prefetch(container_from_array(std::move(cloud)));
         })
    // instrumentation, not in the reference
    .def("lastInliersRatio", &Pipeline::lastInliersRatio)
    .def("lastIcpMs", &Pipeline::lastIcpMs)
    .def("lastBuildMs", &Pipeline::lastBuildMs)
    .def("numKeyframes", &Pipeline::numKeyframes);
}
