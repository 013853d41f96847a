"""Drop-in for the reference's mad_icp.src.pybind.pypeline — re-exports mad_icp_amd.pybind.pypeline (MI355X implementation)."""
from mad_icp_amd.pybind.pypeline import *  # noqa: F401,F403
