"""The round kernel's registers are a contract with the side stream (profiles/r6_prerejection_negative.md): the
one-leaf-per-lane instantiations of icp_round must stay at or below 160 VGPRs and spill nothing.  A 768-thread workgroup is three
wavefronts per SIMD; registers are allocated in granules of 8, so 161 become 168, three wavefronts of 168 leave 8 of a SIMD's 512,
and the small kernels of the side stream (icp_publish: a streamed registration's results out) no longer fit beside a resident
round kernel — measured in round 6: the streamed headline fell from 4 759 to 3 153 registrations/s on a build with 161.
hipcc cross-compiles for gfx950 without a GPU; this test reads its kernel-resource-usage remarks."""
import os
import re
import subprocess

from mad_icp_amd import _build


def test_round_kernel_leaves_room_for_the_side_stream(tmp_path):
    src = os.path.join(_build.CSRC, "hip", "madicp_capi.hip")
    cmd = [_build.HIPCC] + [f for f in _build.HIP_FLAGS if f != "-shared"] + ["-c", "-I" + _build.INC, "-I" + os.path.join(_build.CSRC, "hip"),
                                                                               src, "-o", str(tmp_path / "capi.o"),
                                                                               "-Rpass-analysis=kernel-resource-usage"]
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, check=True).stdout
    seen = 0
    for block in out.split("Function Name: ")[1:]:
        name = block.split()[0]
        if "icp_roundILi1E" not in name:  # (icp_round<1, ...>: every instantiation with one leaf per lane — the default)
            continue
        vgprs = int(re.search(r"VGPRs: (\d+)", block).group(1))
        scratch = int(re.search(r"ScratchSize \[bytes/lane\]: (\d+)", block).group(1))
        seen += 1
        assert vgprs <= 160, (name, vgprs)
        assert scratch == 0, (name, scratch)
    assert seen >= 6, seen
