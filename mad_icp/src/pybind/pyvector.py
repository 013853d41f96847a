"""Drop-in for the reference's mad_icp.src.pybind.pyvector — re-exports mad_icp_amd.pybind.pyvector (MI355X implementation)."""
from mad_icp_amd.pybind.pyvector import *  # noqa: F401,F403
