#!/usr/bin/env python3
"""Write the 'sparse zeros' segment type of the Silesia-like mix (one random byte per ~97 zeros) to a file: mk_sparse.py PATH MIB"""
import os, sys
os.environ.setdefault("FLATE_HIP_PRELOAD_TORCH_HIP", "1")  # one HIP runtime per process: torch, imported later, brings its own (flate_amd/_capi.py)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from flate_amd import synth
path, n = sys.argv[1], int(sys.argv[2]) << 20
z = np.zeros(n, dtype=np.uint8); k = n // 97 + 1
where = (synth.splitmix64(4242, k) % np.uint64(n)).astype(np.int64)
z[where] = (synth.splitmix64(4243, k) & np.uint64(0xFF)).astype(np.uint8)
z.tofile(path)
