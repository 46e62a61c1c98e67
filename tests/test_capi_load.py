"""CPU-side checks of the product boundary: the C-ABI library loads and exports every
symbol include/flate_hip.h declares; without a GPU the product path fails loudly
(no CPU fallback).  No compute calls."""
import os
import re

import pytest

from conftest import ROOT


def test_library_exports_every_declared_symbol():
    from flate_amd import _capi
    hdr = open(os.path.join(ROOT, "include", "flate_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(flate_hip_[a-z0-9_]+)\s*\(", hdr)))
    assert declared == sorted(_capi.SYMBOLS)
    L = _capi.lib()
    for s in declared:
        assert hasattr(L, s), s
    assert b"gfx950" in L.flate_hip_version()
    assert _capi.status_name(14) == "InvalidDynamicBlockHeader"
    assert _capi.status_name(101) == "ChunkTooLarge"


def test_status_names_match_oracle_numbering():
    import _oracle as O
    from flate_amd import _capi
    for code, name in O.STATUS.items():
        assert _capi.status_name(code) == name


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import flate_amd
    from flate_amd import gzip
    with pytest.raises(flate_amd._capi.FlateHipError):
        flate_amd.Engine(0)
    import io
    with pytest.raises(flate_amd._capi.FlateHipError):
        gzip.compress(io.BytesIO(b"hello"), io.BytesIO())


def test_product_package_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "flate_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".hpp")):
                src = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "flate_oracle" not in src and "_oracle" not in src and "oracle/" not in src, f


def test_synth_text_properties():
    import zlib
    import numpy as np
    from flate_amd import synth
    a = synth.text(synth.SEED_TEXT, 1 << 21)
    assert a.dtype == np.uint8 and a.size == 1 << 21
    r = len(zlib.compress(a.tobytes(), 6)) / a.size
    assert 0.30 <= r <= 0.40, r  # enwik-like
    b = synth.text_torch(synth.SEED_TEXT, 1 << 21, device="cpu").numpy()
    assert np.array_equal(a, b)
    s = synth.silesia_like(synth.SEED_SILESIA, 1 << 21)
    assert s.size == 1 << 21
    off = synth.split_offsets(200000, 65535)
    assert list(off) == [0, 65535, 131070, 196605, 200000]


def test_every_tuning_variable_the_library_reads_is_documented_and_reloadable():
    """The library reads its FLATE_HIP_* variables once per handle (flate_hip.hip, read_knobs): every one of them is described in
    INTEGRATION.md section 7 and is in the list the Python engine watches (Engine._KNOBS: a change between two calls makes the
    handle read them again -- tests and probes rely on it)."""
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "flate_amd", "csrc", "flate_hip.hip")).read()
    names = sorted(set(re.findall(r'getenv\("(FLATE_HIP_[A-Z_0-9]+)"\)', src)))
    assert len(names) >= 12, names
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    from flate_amd.engine import Engine
    missing_doc = [n for n in names if n not in doc]
    missing_knob = [n for n in names if n not in Engine._KNOBS]
    assert not missing_doc, missing_doc
    assert not missing_knob, missing_knob
