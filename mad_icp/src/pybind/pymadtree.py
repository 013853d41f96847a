"""Drop-in for the reference's mad_icp.src.pybind.pymadtree — re-exports mad_icp_amd.pybind.pymadtree (MI355X implementation)."""
from mad_icp_amd.pybind.pymadtree import *  # noqa: F401,F403
