// pymadicp — `MADicp` pairwise registration tool (reference: mad_icp/src/pybind/tools/pymadicp.cpp:36-52 over
// tools/mad_icp_wrapper.h:33-112).  Trees are built on the host; the GN loop runs on the MI355X.
#include <sys/time.h>

#include <cmath>
#include <iostream>
#include <memory>
#include <stdexcept>

#include "common.h"
#include "mad_icp.h"
#include "mad_tree.h"
#include "pipeline.h"

using madicp_host::MADicp;
using madicp_host::MADtree;
using madicp_host::Pose;

class MADicpWrapper {
 public:
  // the reference initialises num_threads_ from itself (mad_icp_wrapper.h:35, quirk Q2); the argument is used here
  explicit MADicpWrapper(const int num_threads) : num_threads_(num_threads) {
    max_parallel_levels_ = num_threads_ > 0 ? static_cast<int>(std::log2(num_threads_)) : 0;
  }
  void setQueryCloud(ContainerType query, const double b_max, const double b_min) {
    query_tree_ = std::make_unique<MADtree>(std::move(query), b_max, b_min, max_parallel_levels_);  // quirk Q4: replaced, not appended
  }
  void setReferenceCloud(ContainerType reference, const double b_max, const double b_min) {
    ref_b_max_ = b_max;
    ref_tree_ = std::make_unique<MADtree>(std::move(reference), b_max, b_min, max_parallel_levels_);
  }
  Matrix4d compute(const Matrix4d& T, const size_t max_icp_iterations, const double rho_ker, double b_ratio,
                   const bool print_stats) {
    if (!ref_tree_ || !query_tree_) throw std::runtime_error("MADicp: setReferenceCloud/setQueryCloud first");
    mad_icp_ = std::make_unique<MADicp>(ref_b_max_, rho_ker, b_ratio, 1);
    mad_icp_->setMoving(*query_tree_);
    Pose X;
    for (int r = 0; r < 3; ++r) {
      for (int c = 0; c < 3; ++c) X.R[3 * r + c] = T(r, c);
      X.t[r] = T(r, 3);
    }
    mad_icp_->init(X);
    struct timeval t0, t1;
    gettimeofday(&t0, nullptr);
    mad_icp_->compute({ref_tree_.get()}, static_cast<int>(max_icp_iterations));
    gettimeofday(&t1, nullptr);
    if (print_stats) {
      const float ms = float(t1.tv_sec - t0.tv_sec) * 1000.f + 1e-3f * float(t1.tv_usec - t0.tv_usec);
      const int matched = mad_icp_->numMatched();
      std::cout << "MADicp|compute time " << ms << " [ms] " << std::endl;
      std::cout << "MADicp|inliers ratio " << double(matched) / double(mad_icp_->numMoving()) << std::endl;
      std::cout << "--MADicp|matched leaves " << matched << std::endl;
      std::cout << "--MADicp|total num leaves " << mad_icp_->numMoving() << std::endl;
    }
    return madicp_host::Pipeline::toMatrix(mad_icp_->X_);
  }

 protected:
  std::unique_ptr<MADicp> mad_icp_;
  std::unique_ptr<MADtree> ref_tree_;
  std::unique_ptr<MADtree> query_tree_;
  double ref_b_max_ = 0.2;
  int max_parallel_levels_;
  int num_threads_;
};

PYBIND11_MODULE(pymadicp, m) {
  m.doc() = "mad_icp_amd: MAD-ICP pairwise registration on MI355X, drop-in for mad_icp.src.pybind.pymadicp";
  py::class_<MADicpWrapper>(m, "MADicp")
    .def(py::init<int>(), py::arg("num_threads"))
    .def("setQueryCloud", &MADicpWrapper::setQueryCloud, py::arg("query"), py::arg("b_max") = 0.2, py::arg("b_min") = 0.1)
    .def("setReferenceCloud", &MADicpWrapper::setReferenceCloud, py::arg("reference"), py::arg("b_max") = 0.2,
         py::arg("b_min") = 0.1)
    .def("compute", &MADicpWrapper::compute, py::arg("T"), py::arg("icp_iterations") = madicp_host::MAX_ICP_ITS,
         py::arg("rho_ker") = 0.1, py::arg("b_ratio") = 0.02, py::arg("print_stats") = false);
}
