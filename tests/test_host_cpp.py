"""The C++ host façade (flate_amd/host/flate.hpp) mirrors the reference's public interface
(flate.zig:9-71).  CPU: it compiles and links against the C-ABI library.  GPU: the reference's
"public interface" test (flate.zig:356-481) runs through it."""
import os
import subprocess

import pytest

from conftest import ROOT

SRC = os.path.join(ROOT, "tests", "host_cpp", "test_facade.cpp")
BIN = os.path.join(ROOT, "tests", "host_cpp", "test_facade")


def _build():
    from flate_amd import _capi
    assert os.path.exists(_capi.LIB_PATH)
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-o", BIN, SRC, "-L" + os.path.dirname(_capi.LIB_PATH),
                    "-lflate_hip", "-Wl,-rpath," + os.path.dirname(_capi.LIB_PATH)], check=True)


def test_facade_compiles_and_links():
    _build()
    assert os.path.exists(BIN)


@pytest.mark.gpu
def test_facade_public_interface_on_gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    _build()
    r = subprocess.run([BIN], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "facade ok" in r.stdout
