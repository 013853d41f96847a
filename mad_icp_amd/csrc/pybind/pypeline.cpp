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
    // compute(stamp, VectorEigen3d): the reference binds the by-value member (pypeline.cpp:69), which has pybind copy the whole
    // container per call; bound here on a reference and handed on as a view — the same call for Python, no 3 MB allocation +
    // copy in front of every frame (the host path makes the one copy its tree keeps, inside)
    .def("compute", [](Pipeline& self, double stamp, const ContainerType& cloud) { self.computeView(stamp, cloud.data(), cloud.size()); })
    // additive: an (N,3) float64 array directly (C-contiguous float64 input is read in place)
    .def("compute",
         [](Pipeline& self, double stamp, py::array_t<double, py::array::c_style | py::array::forcecast> cloud) {
           const Vector3d* pts = points_of_array(cloud);
           self.computeView(stamp, pts, static_cast<size_t>(cloud.shape(0)));
         })
    // additive look-ahead: start building the next scan's MAD-tree while this frame is registered
    .def("lookAheadHits", &Pipeline::lookAheadHits)
    .def("prefetch", [](Pipeline& self, const ContainerType& cloud) { self.prefetchView(cloud.data(), cloud.size()); },
         py::arg("next_cloud"))
    .def("prefetch",
         [](Pipeline& self, py::array_t<double, py::array::c_style | py::array::forcecast> cloud) {
           const Vector3d* pts = points_of_array(cloud);
           self.prefetchView(pts, static_cast<size_t>(cloud.shape(0)));
         })
    // additive: deskew + MAD-tree construction on the device — the default; MAD_ICP_GPU_BUILD=0 or
    // setDeviceFrontEnd(False) keep the host builder
    .def("setDeviceFrontEnd", &Pipeline::setDeviceFrontEnd, py::arg("on"))
    .def("deviceFrontEnd", &Pipeline::deviceFrontEnd)
    // additive: a frame straight from sensor records, (n, >=3) float32 (a KITTI .bin is (n,4)): range filter, optional
    // KITTI correction (apps/cpp_runners/bin_runner.cpp:126-166), deskew, build and registration on the device
    .def("computeRecords",
         [](Pipeline& self, double stamp, py::array_t<float, py::array::c_style | py::array::forcecast> rec, double min_range,
            double max_range, bool kitti) {
           if (rec.ndim() != 2 || rec.shape(1) < 3) throw py::cast_error("records must be an (n, >=3) float32 array");
           self.computeRecords(stamp, rec.data(), static_cast<size_t>(rec.shape(0)), static_cast<int>(rec.shape(1)), min_range,
                               max_range, kitti);
         },
         py::arg("stamp"), py::arg("records"), py::arg("min_range"), py::arg("max_range"), py::arg("kitti_correction") = false)
    // instrumentation, not in the reference
    .def("lastInliersRatio", &Pipeline::lastInliersRatio)
    .def("lastRounds", &Pipeline::lastRounds)
#ifdef MADICP_MEASURE  // (test seam of the realtime budget rule: measurement build only)
    .def("setTimingForTest", &Pipeline::setTimingForTest, py::arg("pre_ms"), py::arg("round_ms"))
#endif
    .def("lastIcpMs", &Pipeline::lastIcpMs)
    .def("lastBuildMs", &Pipeline::lastBuildMs)
    .def("lastIcpPhasesMs", &Pipeline::lastIcpPhasesMs)
    .def("numKeyframes", &Pipeline::numKeyframes);
}
