"""The per-point closed form of the permutation the reference's `split` leaves (mad_icp/src/tools/utils.h:37-52), which
the device tree builder's three regimes use (mad_icp_amd/csrc/common/split_order.h), against that loop itself: every
left/right pattern of up to 16 points, random patterns through the 32-bit select form and through whole-node and chunked
rank tables (tests/cpp/split_order_check.cpp).  CPU only."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_split_order_closed_form_matches_the_reference_loop(tmp_path):
    exe = str(tmp_path / "split_order_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "mad_icp_amd", "csrc", "common"),
                           os.path.join(ROOT, "tests", "cpp", "split_order_check.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "split order ok" in out.stdout
