"""Import-path shim: the reference's callers do `from mad_icp.src.pybind.pypeline import Pipeline, VectorEigen3d`
(mad_icp/apps/mad_icp.py:49, apps/utils/tools/*.py:38-40).  These modules re-export the MI355X implementation in
mad_icp_amd.pybind under those exact paths, so launcher and tool scripts run unchanged."""
