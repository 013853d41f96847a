"""mad_icp_amd — MI355X (gfx950) implementation of MAD-ICP's data-association + registration hot path.

Layout:
  csrc/hip/      hand-written HIP kernels + the C ABI (include/madicp_hip.h)  -> libmadicp_hip.so
  csrc/host/     host C++ mirroring the reference's MADtree / MADicp / Pipeline -> libmadicp_host.so
  csrc/pybind/   pybind11 modules with the reference's names                   -> pybind/*.so
  capi.py        ctypes plumbing over the two C ABIs (tests, bench)
  synth.py       seeded KITTI-shaped scan generator (benchmarks / tests)
  sharded.py     keyframe-sharded multi-GPU registration driver (torch.distributed plumbing)
"""
__all__ = ["capi", "synth"]
