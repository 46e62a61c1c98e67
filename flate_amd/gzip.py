"""`gzip` container: drop-in twin of the reference's src/gzip.zig (same function set)."""
from . import _capi
from .api import ContainerModule, Level, Options  # noqa: F401

_m = ContainerModule(_capi.GZIP)
compress = _m.compress
compressor = _m.compressor
Compressor = _m.Compressor
decompress = _m.decompress
decompressor = _m.decompressor
Decompressor = _m.Decompressor
huffman = _m.huffman
store = _m.store
