// pymadtree — `MADtree` NN tool (reference: mad_icp/src/pybind/tools/pymadtree.cpp:36-48 over
// tools/mad_tree_wrapper.h:34-71).  build() runs on the host; every search runs on the MI355X.
// searchCloudArrays is an additive bulk form: the legacy methods build N Python tuples.
#include <memory>
#include <stdexcept>

#include "common.h"
#include "mad_tree.h"

using madicp_host::LeafMatch;
using madicp_host::MADtree;

class MADtreeWrapper {
 public:
  void build(ContainerType vec, const double b_max, const double b_min, const int max_parallel_level) {
    tree_ = std::make_unique<MADtree>(std::move(vec), b_max, b_min, max_parallel_level);
  }
  std::pair<Vector3d, Vector3d> search(const Vector3d& query) {
    const std::vector<LeafMatch> m = need().search(ContainerType{query}, false);
    return std::make_pair(m[0].point, m[0].normal);
  }
  std::vector<std::pair<Vector3d, Vector3d>> searchCloud(const ContainerType& query_cloud) {
    const std::vector<LeafMatch> m = need().search(query_cloud, false);
    std::vector<std::pair<Vector3d, Vector3d>> out(m.size());
    for (size_t i = 0; i < m.size(); ++i) out[i] = std::make_pair(m[i].point, m[i].normal);
    return out;
  }
  std::vector<std::tuple<Vector3d, Vector3d, double>> searchCloudDist(const ContainerType& query_cloud) {
    const std::vector<LeafMatch> m = need().search(query_cloud, true);
    std::vector<std::tuple<Vector3d, Vector3d, double>> out(m.size());
    for (size_t i = 0; i < m.size(); ++i) out[i] = std::make_tuple(m[i].point, m[i].normal, m[i].dist);
    return out;
  }
  py::tuple searchCloudArrays(const ContainerType& query_cloud) {
    const std::vector<LeafMatch> m = need().search(query_cloud, true);
    const py::ssize_t n = static_cast<py::ssize_t>(m.size());
    py::array_t<double> pts({n, py::ssize_t(3)}), nrm({n, py::ssize_t(3)}), dist(n);
    py::array_t<uint32_t> leaf(n);
    auto p = pts.mutable_unchecked<2>();
    auto q = nrm.mutable_unchecked<2>();
    auto d = dist.mutable_unchecked<1>();
    auto li = leaf.mutable_unchecked<1>();
    for (py::ssize_t i = 0; i < n; ++i) {
      for (int k = 0; k < 3; ++k) {
        p(i, k) = m[i].point[k];
        q(i, k) = m[i].normal[k];
      }
      d(i) = m[i].dist;
      li(i) = m[i].leaf_idx;
    }
    return py::make_tuple(pts, nrm, dist, leaf);  // (points, normals, dist, leaf_idx)
  }
  int numLeaves() { return need().numLeaves(); }

 private:
  MADtree& need() {
    // the reference dereferences a null unique_ptr here (mad_tree_wrapper.h:43); raise instead
    if (!tree_) throw std::runtime_error("MADtree: build() has not been called");
    return *tree_;
  }
  std::unique_ptr<MADtree> tree_;
};

PYBIND11_MODULE(pymadtree, m) {
  m.doc() = "mad_icp_amd: MAD-tree nearest-neighbour/normal lookup on MI355X, drop-in for mad_icp.src.pybind.pymadtree";
  py::class_<MADtreeWrapper>(m, "MADtree")
    .def(py::init<>())
    .def("build", &MADtreeWrapper::build, py::arg("vec"), py::arg("b_max") = 1e-5, py::arg("b_min") = 0.1,
         py::arg("max_parallel_level") = 2)
    .def("search", &MADtreeWrapper::search, py::arg("query"))
    .def("searchCloud", &MADtreeWrapper::searchCloud, py::arg("query_cloud"))
    .def("searchCloudDist", &MADtreeWrapper::searchCloudDist, py::arg("query_cloud"))
    .def("searchCloudArrays", &MADtreeWrapper::searchCloudArrays, py::arg("query_cloud"))
    .def("numLeaves", &MADtreeWrapper::numLeaves);
}
