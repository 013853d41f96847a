#!/bin/sh
# The reference's two example scripts (apps/utils/tools/nn_search.py, mad_registration.py) and the fixture module they import
# (tools_utils.py) are the only known-answer material the reference ships (apps/utils/tools/README.md:9-10).  They need a GPU
# (every search / registration runs on the HIP path) and /root/reference does not exist on the GPU box — so, like
# oracle/_ref/bin_runner, they travel inside oracle/_ref/ (git-ignored: never part of this repository's history), copied
# UNCHANGED from where they lie by __graft_entry__.build().  tests/test_reference_scripts.py runs them from there.
set -e
REF=${1:-/root/reference/mad_icp}
HERE=$(cd "$(dirname "$0")" && pwd)
DST=$HERE/_ref/tools/mad_icp/apps/utils/tools
[ -f "$REF/apps/utils/tools/nn_search.py" ] || { echo "ship_ref_tools.sh: no reference tree at $REF"; exit 3; }
mkdir -p "$DST"
[ -f "$REF/apps/__init__.py" ] && cp "$REF/apps/__init__.py" "$HERE/_ref/tools/mad_icp/apps/__init__.py"
for f in nn_search.py mad_registration.py tools_utils.py; do cp "$REF/apps/utils/tools/$f" "$DST/$f"; done
echo "ship_ref_tools.sh: $DST (nn_search.py, mad_registration.py, tools_utils.py: byte-for-byte copies from $REF/apps/utils/tools)"
# The reference's Python LAUNCHER (apps/mad_icp.py: typer CLI -> dataset reader -> Pipeline.compute -> estimate.txt), the
# readers and helpers it imports and the configuration tables it looks datasets up in — the other caller north_star names
# ("drops into bin_runner and the Python launcher unchanged").  Same rule: byte for byte, git-ignored, test infrastructure.
cp "$REF/apps/mad_icp.py" "$HERE/_ref/tools/mad_icp/apps/mad_icp.py"
for f in utils.py kitti_reader.py ros_reader.py ros2_reader.py mcap_reader.py point_cloud2.py visualizer.py; do
  cp "$REF/apps/utils/$f" "$HERE/_ref/tools/mad_icp/apps/utils/$f"
done
mkdir -p "$HERE/_ref/tools/mad_icp/configurations/datasets"
cp "$REF/configurations/default.cfg" "$REF/configurations/mad_params.py" "$HERE/_ref/tools/mad_icp/configurations/"
cp "$REF"/configurations/datasets/*.py "$REF"/configurations/datasets/*.cfg "$HERE/_ref/tools/mad_icp/configurations/datasets/"
echo "ship_ref_tools.sh: $HERE/_ref/tools/mad_icp/apps/mad_icp.py + apps/utils/*.py + configurations/ (the Python launcher, unchanged)"
