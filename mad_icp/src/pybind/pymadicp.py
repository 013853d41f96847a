"""Drop-in for the reference's mad_icp.src.pybind.pymadicp — re-exports mad_icp_amd.pybind.pymadicp (MI355X implementation)."""
from mad_icp_amd.pybind.pymadicp import *  # noqa: F401,F403
