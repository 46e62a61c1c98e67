#!/usr/bin/env python3
"""gzip <file>: writes <file>.gz with the MI355X engine -- the reference's bin/gzip.zig:20
(`gzip.compress(input_file.reader(), output_file.writer(), .{})`, comparable to `gzip -kfn`).
Options beyond the reference's tool: -l LEVEL (4..9), --huffman, --store."""
import argparse
import os
os.environ.setdefault("FLATE_HIP_PRELOAD_TORCH_HIP", "1")  # one HIP runtime per process: torch, imported later, brings its own (flate_amd/_capi.py)
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__)
    ap.add_argument("input_file")
    ap.add_argument("-l", dest="level", type=int, default=6, choices=range(4, 10))
    ap.add_argument("--huffman", action="store_true")
    ap.add_argument("--store", action="store_true")
    a = ap.parse_args(argv)
    from flate_amd import gzip
    with open(a.input_file, "rb") as src, open(a.input_file + ".gz", "wb") as dst:
        if a.huffman:
            gzip.huffman.compress(src, dst)
        elif a.store:
            gzip.store.compress(src, dst)
        else:
            gzip.compress(src, dst, gzip.Options(level=a.level))
    return 0


if __name__ == "__main__":
    sys.exit(main())
