#!/bin/sh
# ORACLE — test infrastructure only.  Compiles the REFERENCE's own C++ caller, apps/cpp_runners/bin_runner.cpp, UNCHANGED
# from where it lies under /root/reference, against the PRODUCT: its `#include <odometry/pipeline.h>` / `<tools/mad_tree.h>`
# resolve (oracle/bin_runner_shim) to mad_icp_amd/csrc/host/{pipeline,mad_tree}.h, its `<Eigen/Core>` / `<Eigen/Dense>` to the
# Eigen stand-in (oracle/eigen_standin: this image has no Eigen), its `<yaml-cpp/yaml.h>` to a stub of the four calls it makes
# (this image has no yaml-cpp), its `cpp_utils/parse_cmd_line.h` to the reference's own header.  With an <Eigen/Core> on the
# include path the product's public value types ARE the Eigen types (csrc/host/types.h, the `__has_include` branch), so the
# product's host sources are compiled here a second time in that mode and linked into the runner; the GPU half is the
# shipped libmadicp_hip.so, untouched.  Output: oracle/_ref/bin_runner (git-ignored, travels to the GPU box with the
# snapshot; tests/test_gpu_bin_runner.py runs it on a KITTI-format directory and compares estimate.txt with the oracle
# pipeline's poses).  No reference source is copied into this repository.
# Usage:  oracle/build_bin_runner.sh [REFERENCE_ROOT]
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(dirname "$HERE")"
REF="${1:-/root/reference}"
APP="$REF/mad_icp/apps/cpp_runners"
OUT="$HERE/_ref"
HOST="$ROOT/mad_icp_amd/csrc/host"
[ -f "$APP/bin_runner.cpp" ] || { echo "build_bin_runner.sh: $APP/bin_runner.cpp not found (the reference is not on this machine)" >&2; exit 4; }
[ -f "$ROOT/mad_icp_amd/libmadicp_hip.so" ] || { echo "build_bin_runner.sh: build the product first (mad_icp_amd/_build.py)" >&2; exit 5; }
mkdir -p "$OUT"
# the reference's flag set (mad_icp/CMakeLists.txt:6-8,38-40) + what the product's host library is built with
g++ -O3 -std=c++17 -fopenmp -ffp-contract=off -DNDEBUG -DMADICP_STANDIN_TYPES_ONLY \
  -I"$HERE/bin_runner_shim" -I"$HERE/eigen_standin" -I"$APP" -I"$ROOT/include" -I"$HOST" \
  "$APP/bin_runner.cpp" \
  "$HOST/tree_builder.cpp" "$HOST/mad_tree.cpp" "$HOST/mad_icp.cpp" "$HOST/vel_estimator.cpp" "$HOST/pipeline.cpp" "$HOST/deskew.cpp" \
  -o "$OUT/bin_runner" -L"$ROOT/mad_icp_amd" -lmadicp_hip -Wl,-rpath,'$ORIGIN/../../mad_icp_amd' -pthread
echo "build_bin_runner.sh: $OUT/bin_runner (reference caller from $APP, product host layer in its Eigen-typed mode, libmadicp_hip.so)"
