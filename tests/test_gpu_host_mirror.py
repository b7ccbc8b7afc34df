"""The host-side C++ mirror of chisel::Chisel / ChunkManager / Atlas (texturefusion_amd/host/tf_chisel.hpp),
driven like GCFusion/MobileFusion.cpp drives the reference, against the oracle -- a compiled C++ program."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_mirror_header_compiles():
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True, stdout=subprocess.DEVNULL)
    subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "cpp")], check=True, stdout=subprocess.DEVNULL)
    assert os.path.exists(os.path.join(ROOT, "tests", "cpp", "host_mirror_parity"))


@pytest.mark.gpu
def test_host_mirror_parity(gpu_required):
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True, stdout=subprocess.DEVNULL)
    subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "cpp")], check=True, stdout=subprocess.DEVNULL)
    r = subprocess.run([os.path.join(ROOT, "tests", "cpp", "host_mirror_parity")], capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "HOST MIRROR PARITY OK" in r.stdout
