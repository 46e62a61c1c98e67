"""The oracle's window-slide path (inputs beyond 64 KiB: deflate.zig:291-321, SlidingWindow.zig:36-44, Lookup.zig:43-51,
and the raw slice a block loses after a slide, SlidingWindow.zig:119-123) against fixtures made by an INDEPENDENT
pure-Python model of the reference (tests/golden/make_slide_fixtures.py, written from the Zig sources, not from the
oracle): token lists by SHA-256 and, per flushTokens, what the block writer was handed.  No vector the reference
itself holds is longer than 65 535 bytes; these six inputs of 150-300 KB at levels 4 / 6 / 9 are the second opinion."""
import hashlib
import json
import os

import numpy as np
import pytest

import _oracle as O
from conftest import GOLDEN

with open(os.path.join(GOLDEN, "slide", "fixtures.json")) as f:
    FIX = json.load(f)


@pytest.mark.parametrize("key", sorted(FIX))
def test_oracle_tokens_and_blocks_beyond_64k(key):
    name, level = key.split("@")
    want = FIX[key]
    with open(os.path.join(GOLDEN, "slide", name + ".bin"), "rb") as f:
        data = f.read()
    assert len(data) == want["bytes"]
    d = O.Deflate(O.RAW, int(level), log_tokens=True)
    d.write(data)
    d.flush()
    toks = d.tokens()
    blocks = d.blocks()
    d.close()
    assert len(toks) == want["tokens"]
    assert hashlib.sha256(np.ascontiguousarray(toks, dtype="<u4").tobytes()).hexdigest() == want["sha256"]
    assert [list(b) for b in blocks] == want["blocks"]
    # (the same stream through the one-shot entry point inflates to the input)
    import zlib
    assert zlib.decompress(O.compress(data, O.RAW, int(level)), -15) == data
