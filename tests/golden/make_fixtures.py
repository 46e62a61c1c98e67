#!/usr/bin/env python3
"""Regenerate tests/golden/ from the reference tree (run in the build container only).

What it produces (all of it is DATA -- inputs and expected outputs the
reference's own tests hold -- never reference source text):

  block_writer/*      the 52 data files of src/flate/testdata/block_writer/
                      (9 inputs + 43 expected block encodings,
                      block_writer.zig:599-706)
  fuzz/*              the 48 files of src/flate/testdata/fuzz/
                      (inflate.zig:481-563, deflate.zig:636-643)
  rfc1951.txt         src/flate/testdata/rfc1951.txt (deflate.zig:620, flate.zig:101)
  fixed_codes.bin     the 297 expected bytes of huffman_encoder.zig:497-536
  block_writer_tokens.json
                      the 9 token lists of src/flate/testdata/block_writer.zig,
                      transcribed into a neutral format:
                      {"input","want","want_no_input","tokens":[[lit] | [dist,len]]}

The GPU box has no /root/reference; tests read only the committed copies.
"""
import json
import os
import re
import shutil
import sys

REF = os.environ.get("FLATE_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))

_ESC = {"n": 10, "t": 9, "r": 13, "\\": 92, "'": 39, '"': 34, "0": 0}


def _char_literal(body):
    if body.startswith("\\x"):
        return int(body[2:], 16)
    if body.startswith("\\"):
        return _ESC[body[1]]
    assert len(body.encode()) == 1, body
    return ord(body)


def parse_token_lists(text):
    cases = []
    # one TestCase{...} per case; fields are simple string/array initialisers
    for m in re.finditer(r"TestCase\{(.*?)\n        \},", text, re.S):
        blk = m.group(1)
        fields = {}
        for key in ("input", "want", "want_no_input"):
            fm = re.search(r"\.%s = \"([^\"]*)\"" % key, blk)
            fields[key] = fm.group(1) if fm else ""
        tm = re.search(r"\.tokens = &\[_\]Token\{(.*)\}", blk, re.S)
        toks = []
        src = tm.group(1)
        i = 0
        tok_re = re.compile(r"L\('((?:\\.|\\x[0-9a-fA-F]{2}|[^'\\]))'\)|L\((0x[0-9a-fA-F]+|\d+)\)|M\((\d+),\s*(\d+)\)|\bml\b")
        for t in tok_re.finditer(src):
            if t.group(1) is not None:
                toks.append([_char_literal(t.group(1))])
            elif t.group(2) is not None:
                toks.append([int(t.group(2), 0)])
            elif t.group(3) is not None:
                toks.append([int(t.group(3)), int(t.group(4))])
            else:
                toks.append([1, 258])  # ml = M(1, 258)
        fields["tokens"] = toks
        cases.append(fields)
    return cases


def main():
    src = os.path.join(REF, "src/flate/testdata")
    if not os.path.isdir(src):
        sys.exit("reference tree not found at %s" % REF)
    for sub in ("block_writer", "fuzz"):
        dst = os.path.join(HERE, sub)
        os.makedirs(dst, exist_ok=True)
        for name in sorted(os.listdir(os.path.join(src, sub))):
            shutil.copyfile(os.path.join(src, sub, name), os.path.join(dst, name))
            os.chmod(os.path.join(dst, name), 0o644)
    shutil.copyfile(os.path.join(src, "rfc1951.txt"), os.path.join(HERE, "rfc1951.txt"))
    os.chmod(os.path.join(HERE, "rfc1951.txt"), 0o644)
    with open(os.path.join(src, "block_writer.zig")) as f:
        cases = parse_token_lists(f.read())
    assert len(cases) == 9, len(cases)
    with open(os.path.join(HERE, "block_writer_tokens.json"), "w") as f:
        json.dump(cases, f, separators=(",", ":"))
    # the 297-byte known-answer bitstream of the fixed literal code
    # (huffman_encoder.zig:497-536 `fixed_codes`): expected OUTPUT bytes only.
    with open(os.path.join(REF, "src/flate/huffman_encoder.zig")) as f:
        enc = f.read()
    body = enc[enc.index("pub const fixed_codes"):]
    body = body[: body.index("};")]
    fixed = bytes(int(b, 2) for b in re.findall(r"0b([01]{8})", body))
    assert len(fixed) == 297, len(fixed)
    with open(os.path.join(HERE, "fixed_codes.bin"), "wb") as f:
        f.write(fixed)
    print("token lists:", [(c["input"] or c["want_no_input"], len(c["tokens"])) for c in cases])


if __name__ == "__main__":
    main()
