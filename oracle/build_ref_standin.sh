#!/bin/sh
# ORACLE — test infrastructure only.  Compiles the REFERENCE's own translation units of the hot path (tools/mad_tree.cpp,
# odometry/{mad_icp,vel_estimator,pipeline}.cpp) from where they lie under /root/reference, with g++ and the reference's
# flag set (Release -O3, C++17, -fopenmp, no -march, no fast-math: mad_icp/CMakeLists.txt:6-8,38-40), against the Eigen
# STAND-IN of oracle/eigen_standin (this image has no Eigen; oracle/build_ref.sh is the recipe for a real one), behind the
# oracle's own C ABI (oracle/ref_standin_capi.cpp).  Output: oracle/_ref/libmad_ref_standin.so — git-ignored, travels to
# the GPU box with the snapshot, never copied into the product.  No reference source is copied into this repository.
# What this pins: the oracle's control flow, mechanically (tests/test_reference_structure_pin.py).  What it does not:
# Eigen's own arithmetic — see oracle/eigen_standin/standin.h.
# Usage:  oracle/build_ref_standin.sh [REFERENCE_ROOT]
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
REF="${1:-/root/reference}"
SRC="$REF/mad_icp/src"
OUT="$HERE/_ref"
[ -d "$SRC" ] || { echo "build_ref_standin.sh: $SRC not found (the reference is not on this machine)" >&2; exit 4; }
mkdir -p "$OUT"
g++ -O3 -std=c++17 -fopenmp -fPIC -shared -ffp-contract=off -DNDEBUG -DMADICP_REDUX_SCALAR_ONLY \
  -I"$HERE/eigen_standin" -I"$SRC" \
  "$HERE/ref_standin_capi.cpp" "$SRC/tools/mad_tree.cpp" "$SRC/odometry/mad_icp.cpp" "$SRC/odometry/vel_estimator.cpp" \
  "$SRC/odometry/pipeline.cpp" -o "$OUT/libmad_ref_standin.so" -pthread
echo "build_ref_standin.sh: $OUT/libmad_ref_standin.so (reference sources from $SRC, Eigen stand-in)"
