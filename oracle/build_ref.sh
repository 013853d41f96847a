#!/bin/sh
# ORACLE — test infrastructure only.  Builds the REFERENCE's own pybind modules (pyvector, pypeline, pymadtree, pymadicp)
# from the sources where they lie under /root/reference, with g++ and the reference's flag set (Release -O3, C++17,
# -fopenmp, no -march, no fast-math: mad_icp/CMakeLists.txt:6-8,38-40), into oracle/_ref/ — git-ignored, never copied
# into the product.  With those modules present tests/test_reference_pin.py checks the oracle restatement against the
# reference itself and oracle/_ref pins what is today "parity unpinned".
#
# It needs Eigen 3.3/3.4 headers, which this image does not ship (no network either): the script looks for them and
# says so when they are missing.  Usage:  oracle/build_ref.sh [EIGEN_INCLUDE_DIR] [REFERENCE_ROOT]
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
REF="${2:-/root/reference}"
SRC="$REF/mad_icp/src"
OUT="$HERE/_ref"
EIGEN="$1"
if [ -z "$EIGEN" ]; then
  for d in /usr/include/eigen3 /usr/local/include/eigen3 /opt/eigen3 "$HOME/eigen3" "$HERE/_eigen"; do
    [ -f "$d/Eigen/Core" ] && EIGEN="$d" && break
  done
fi
if [ -z "$EIGEN" ] || [ ! -f "$EIGEN/Eigen/Core" ]; then
  echo "build_ref.sh: no Eigen headers found (looked for <dir>/Eigen/Core); pass the include directory as \$1." >&2
  echo "build_ref.sh: without them the reference cannot be compiled: the oracle stays 'parity unpinned'." >&2
  exit 3
fi
[ -d "$SRC" ] || { echo "build_ref.sh: $SRC not found" >&2; exit 4; }
PY="${PYTHON:-python3}"
PYINC="$($PY -c 'import sysconfig; print(sysconfig.get_paths()["include"])')"
PBINC="$($PY -c 'import pybind11; print(pybind11.get_include())')"
SUFFIX="$($PY -c 'import sysconfig; print(sysconfig.get_config_var("EXT_SUFFIX"))')"
FLAGS="-O3 -std=c++17 -fopenmp -fPIC -shared -fvisibility=hidden -DNDEBUG"
INC="-I$EIGEN -I$SRC -I$SRC/.. -I$PYINC -I$PBINC"
CORE="$SRC/tools/mad_tree.cpp"
ODOM="$SRC/odometry/mad_icp.cpp $SRC/odometry/vel_estimator.cpp $SRC/odometry/pipeline.cpp"
mkdir -p "$OUT"
set -x
g++ $FLAGS $INC "$SRC/pybind/pyvector.cpp" -o "$OUT/pyvector$SUFFIX"
g++ $FLAGS $INC "$SRC/pybind/tools/pymadtree.cpp" $CORE -o "$OUT/pymadtree$SUFFIX"
g++ $FLAGS $INC "$SRC/pybind/tools/pymadicp.cpp" $CORE $ODOM -o "$OUT/pymadicp$SUFFIX"
g++ $FLAGS $INC "$SRC/pybind/pypeline.cpp" $CORE $ODOM -o "$OUT/pypeline$SUFFIX"
set +x
echo "build_ref.sh: reference modules in $OUT (Eigen from $EIGEN)"
