"""flate_amd -- MI355X (gfx950) DEFLATE engine behind ianic/flate's API.

    from flate_amd import flate, gzip, zlib        # the three container modules
    gzip.compress(reader, writer, gzip.Options(level=gzip.Level.default))

The compute path is libflate_hip.so (hand-written HIP kernels) reached through the
C ABI of include/flate_hip.h; importing this package needs no GPU, calling it does.
"""
from . import _capi  # noqa: F401
from .api import (ChunkTooLarge, FlateError, Level, Options)  # noqa: F401
from .engine import Engine, default_engine  # noqa: F401

__all__ = ["flate", "gzip", "zlib", "Engine", "default_engine", "Level", "Options", "FlateError"]


def __getattr__(name):
    if name in ("flate", "gzip", "zlib", "synth", "sharded"):
        import importlib
        return importlib.import_module("." + name, __name__)
    raise AttributeError(name)
