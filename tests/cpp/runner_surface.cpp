// C++ boundary proof: a caller written like the reference's apps/cpp_runners/bin_runner.cpp, making
// exactly the calls it makes against the Pipeline surface — construction (bin_runner.cpp:106-107), `ContainerType
// cloud` / `Matrix4d lidar_to_world` declarations (:117-118), currentID() (:121), the .bin decoding loop that fills the
// cloud (float32 x,y,z,i records, range filter: :126-166), compute(time, cloud) with the cloud passed BY VALUE (:174) and
// currentPose() (:180) — compiled with g++ against mad_icp_amd/csrc/host/pipeline.h and linked with libmadicp_host.so /
// libmadicp_hip.so.  The only line that differs from the reference's runner is the matrix type's spelling: where Eigen
// is installed `Matrix4d` IS Eigen::Matrix4d (csrc/host/types.h), here it is the layout-identical POD.
//
// usage: runner_surface <dir with NNNNNN.bin files> <out estimate.txt>   (yaml parsing, which the reference's runner does
// with yaml-cpp, is not part of the Pipeline surface: parameters are default.cfg / kitti.cfg's, fixed below)
#include <sys/time.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <filesystem>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <memory>
#include <string>
#include <vector>

#include "pipeline.h"

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  const std::string data_path = argv[1], estimate_file = argv[2];
  // configurations/default.cfg:2-7 and configurations/datasets/kitti.cfg:2-11
  const double sensor_hz = 10., b_max = 0.2, rho_ker = 0.1, p_th = 0.8, b_min = 0.1, b_ratio = 0.02;
  const double min_range = 0.7, max_range = 120.;
  const bool deskew = false, realtime = false;
  const int num_keyframes = 4, num_cores = 4;

  std::vector<std::string> files_in_directory;
  for (const auto& e : std::filesystem::directory_iterator(data_path))
    if (e.path().extension() == ".bin") files_in_directory.push_back(e.path().string());
  std::sort(files_in_directory.begin(), files_in_directory.end());

  std::unique_ptr<Pipeline> pipeline =
    std::make_unique<Pipeline>(sensor_hz, deskew, b_max, rho_ker, p_th, b_min, b_ratio, num_keyframes, num_cores, realtime);

  double time = 0.;
  const double time_incr = 1. / sensor_hz;

  std::ofstream os;
  os.open(estimate_file);
  os << std::fixed << std::setprecision(12);

  ContainerType cloud;
  madicp_host::Matrix4d lidar_to_world;

  for (const std::string& filename : files_in_directory) {
    std::cout << "Loading frame # " << pipeline->currentID() << std::endl;
    int32_t num = 1000000;
    float* data = (float*) malloc(num * sizeof(float));
    float* px = data + 0;
    float* py = data + 1;
    float* pz = data + 2;
    cloud.clear();
    cloud.reserve(num);
    FILE* stream = fopen(filename.c_str(), "rb");
    num = fread(data, sizeof(float), num, stream) / 4;
    for (int32_t i = 0; i < num; i++) {
      const float x = *px, y = *py, z = *pz;
      px += 4;
      py += 4;
      pz += 4;
      const float norm = std::sqrt(x * x + y * y + z * z);
      if (norm < min_range || norm > max_range || std::isnan(x) || std::isnan(y) || std::isnan(z)) continue;
      madicp_host::Vector3d p;
      p[0] = double(x);
      p[1] = double(y);
      p[2] = double(z);
      cloud.push_back(p);
    }
    cloud.shrink_to_fit();
    fclose(stream);
    free(data);

    struct timeval t_start, t_end, t_delta;
    gettimeofday(&t_start, nullptr);
    pipeline->compute(time, cloud);
    gettimeofday(&t_end, nullptr);
    timersub(&t_end, &t_start, &t_delta);
    std::cout << std::fixed << std::setprecision(4)
              << "Time for odometry estimation [ms]: " << double(t_delta.tv_sec) * 1000. + 1e-3 * t_delta.tv_usec << std::endl;

    lidar_to_world = pipeline->currentPose();
    time += time_incr;
    // KITTI format row (bin_runner.cpp:253-269): the 3x4 pose, row-major
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 4; ++c) os << lidar_to_world(r, c) << ((r == 2 && c == 3) ? "\n" : " ");
  }
  os.close();
  return 0;
}
