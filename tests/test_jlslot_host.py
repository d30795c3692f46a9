"""csrc/jlslot.hpp -- the Julia 1.7 slot-order model the device front-end runs per lane / per wavefront -- compiled for the
host and compared with csrc/jlorder.hpp (jl::SlotTable, pinned to the reference's dumps by test_julia_order.py) on 8 000 random
insertion sequences: duplicates, growth steps, the 64 000-key threshold, keys around 2^32, interleaved storage."""
import os
import subprocess
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))


def test_jlslot_matches_host_model():
    exe = os.path.join(tempfile.gettempdir(), "ecne_jlslot_check_%d" % os.getuid())
    subprocess.check_call(["g++", "-O2", "-std=c++17", os.path.join(HERE, "native", "jlslot_check.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "0 mismatches" in out.stdout
