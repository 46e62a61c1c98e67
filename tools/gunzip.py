#!/usr/bin/env python3
"""gunzip <file>.gz: writes <file> with the MI355X engine -- the reference's bin/gunzip.zig:25-27
(`gzip.decompress(br.reader(), output_file.writer())`; refuses names without the .gz suffix, :15-19).
Concatenated members are decoded one after the other (Inflate.reset, inflate.zig:301-309)."""
import os
os.environ.setdefault("FLATE_HIP_PRELOAD_TORCH_HIP", "1")  # one HIP runtime per process: torch, imported later, brings its own (flate_amd/_capi.py)
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    if len(argv) != 1:
        print("usage: gunzip.py <file>.gz", file=sys.stderr)
        return 2
    name = argv[0]
    if not name.endswith(".gz"):
        print("not a .gz file", file=sys.stderr)
        return 1
    from flate_amd import gzip
    # decoded into a temporary file beside the target: an archive that fails to decode leaves an existing
    # file of that name as it was (and no partial output behind)
    out_name, tmp_name = name[:-3], name[:-3] + ".gunzip-tmp"
    try:
        with open(name, "rb") as src, open(tmp_name, "wb") as dst:
            d = gzip.decompressor(src)
            d.decompress(dst)
            while d.more_input():  # further members of the same file
                d.reset()
                d.decompress(dst)
        os.replace(tmp_name, out_name)
    except BaseException:
        if os.path.exists(tmp_name):
            os.unlink(tmp_name)
        raise
    return 0


if __name__ == "__main__":
    sys.exit(main())
