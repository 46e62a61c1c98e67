"""ctypes binding of libflate_hip.so (the C ABI declared in include/flate_hip.h).

The product path.  It never imports the CPU oracle and has no CPU fallback: if the
library is missing or no HIP device is usable, calls raise FlateHipError.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (FLATE_HIP_LIB: another build of the same library, e.g. one with tuning counters -- tools/ only)
LIB_PATH = os.environ.get("FLATE_HIP_LIB") or os.path.join(_HERE, "lib", "libflate_hip.so")

RAW, GZIP, ZLIB = 0, 1, 2
MODE_STORE, MODE_HUFFMAN = 0, 1
MEM_HOST, MEM_DEVICE = 0, 1
MAX_LZ_CHUNK = 65535
INFLATE_STRICT_Q6 = 1
DEFLATE_REPAIR_Q1 = 1          # flate_hip_set_flags
ST_REFERENCE_Q1_STREAM = 102   # compress status: the reference's bytes, which do not inflate to the input (include/flate_hip.h)

# every symbol include/flate_hip.h declares
SYMBOLS = [
    "flate_hip_create", "flate_hip_destroy", "flate_hip_set_stream", "flate_hip_set_sync", "flate_hip_set_flags",
    "flate_hip_compress_bound", "flate_hip_compress_batch", "flate_hip_decompress_batch",
    "flate_hip_status_name", "flate_hip_last_error", "flate_hip_version",
    "flate_hip_profile_enable", "flate_hip_profile_read", "flate_hip_profile_reset",
    "flate_hip_debug_tokens", "flate_hip_gather_streams", "flate_hip_debug_phase_cycles",
    "flate_hip_compress_flush", "flate_hip_debug_write_block",
    "flate_hip_compress_batch_sharded", "flate_hip_decompress_batch_sharded",
    "flate_hip_plan_compress", "flate_hip_compress_planned", "flate_hip_plan_destroy",
    "flate_hip_checksum", "flate_hip_checksum_combine", "flate_hip_debug_reload_env",
]


class FlateHipError(RuntimeError):
    pass


_lib = None


def _share_torchs_hip_runtime():
    """A process that also runs PyTorch-ROCm must have ONE HIP runtime: torch ships its own libamdhip64, and if this
    library is loaded first it brings in /opt/rocm's -- the second runtime to come up then finds no usable device
    (measured: flate_hip_create fails with NO_DEVICE after `import torch`).  When torch is imported already, its runtime
    is the one the library binds to (same SONAME) and nothing is to do.  Otherwise torch's copy is loaded first ONLY when
    FLATE_HIP_PRELOAD_TORCH_HIP=1 asks for it (bench.py and the test suite set it: they import torch later); a process
    that never imports torch gets /opt/rocm's runtime and no side effect.  torch itself is never imported here."""
    import sys
    if "torch" in sys.modules or os.environ.get("FLATE_HIP_PRELOAD_TORCH_HIP", "0") in ("", "0"):
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        for d in (spec.submodule_search_locations or []) if spec else []:
            cand = os.path.join(d, "lib", "libamdhip64.so")
            if os.path.exists(cand):
                C.CDLL(cand, mode=C.RTLD_GLOBAL)
                return
    except Exception:  # noqa: BLE001 -- without torch there is nothing to share
        pass


def lib():
    """Load the shared library (pure dlopen: no GPU needed until flate_hip_create)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FlateHipError("libflate_hip.so not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                            "or `make -C flate_amd/csrc` (%s)" % LIB_PATH)
    _share_torchs_hip_runtime()
    L = C.CDLL(LIB_PATH)
    vp, u64p, i32p = C.c_void_p, C.c_void_p, C.c_void_p
    L.flate_hip_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    L.flate_hip_create.restype = C.c_int
    L.flate_hip_destroy.argtypes = [vp]
    L.flate_hip_set_stream.argtypes = [vp, vp]
    L.flate_hip_set_sync.argtypes = [vp, C.c_int]
    L.flate_hip_set_flags.argtypes = [vp, C.c_uint32]
    L.flate_hip_set_flags.restype = C.c_int
    L.flate_hip_compress_bound.argtypes = [C.c_size_t, C.c_int, C.c_int]
    L.flate_hip_compress_bound.restype = C.c_size_t
    L.flate_hip_compress_batch.argtypes = [vp, vp, u64p, C.c_uint32, C.c_int, C.c_int, vp, u64p, u64p, i32p, C.c_int]
    L.flate_hip_compress_batch.restype = C.c_int
    L.flate_hip_decompress_batch.argtypes = [vp, vp, u64p, C.c_uint32, C.c_int, C.c_int, vp, u64p, u64p, i32p, u64p,
                                             C.c_int]
    L.flate_hip_decompress_batch.restype = C.c_int
    L.flate_hip_status_name.argtypes = [C.c_int]
    L.flate_hip_status_name.restype = C.c_char_p
    L.flate_hip_last_error.argtypes = [vp]
    L.flate_hip_last_error.restype = C.c_char_p
    L.flate_hip_version.restype = C.c_char_p
    L.flate_hip_profile_enable.argtypes = [vp, C.c_int]
    L.flate_hip_profile_reset.argtypes = [vp]
    L.flate_hip_profile_read.argtypes = [vp, vp, vp, vp, C.c_int]
    L.flate_hip_profile_read.restype = C.c_int
    L.flate_hip_gather_streams.argtypes = [vp, vp, u64p, u64p, C.c_uint32, vp, u64p]
    L.flate_hip_gather_streams.restype = C.c_int
    L.flate_hip_debug_phase_cycles.argtypes = [vp, u64p, C.c_int]
    L.flate_hip_debug_phase_cycles.restype = C.c_int
    L.flate_hip_debug_reload_env.argtypes = [vp]
    L.flate_hip_debug_reload_env.restype = C.c_int
    L.flate_hip_compress_flush.argtypes = [vp, vp, C.c_uint64, vp, C.c_uint32, C.c_int, C.c_int, C.c_int, vp,
                                           C.c_uint64, vp, vp, C.c_int]
    L.flate_hip_compress_flush.restype = C.c_int
    L.flate_hip_debug_write_block.argtypes = [vp, vp, C.c_uint32, vp, C.c_uint32, C.c_int, C.c_int, vp, C.c_uint64, vp]
    L.flate_hip_debug_write_block.restype = C.c_int
    L.flate_hip_compress_batch_sharded.argtypes = [vp, vp, C.c_int, C.c_int, vp, u64p, C.c_uint32, C.c_int, C.c_int, vp,
                                                   u64p, u64p, i32p, vp, C.c_uint64, u64p, u64p]
    L.flate_hip_compress_batch_sharded.restype = C.c_int
    L.flate_hip_decompress_batch_sharded.argtypes = [vp, vp, C.c_int, C.c_int, vp, u64p, C.c_uint32, C.c_int, C.c_int, vp,
                                                     C.c_uint64, u64p, u64p, i32p, u64p]
    L.flate_hip_decompress_batch_sharded.restype = C.c_int
    L.flate_hip_plan_compress.argtypes = [vp, u64p, u64p, C.c_uint32, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    L.flate_hip_plan_compress.restype = C.c_int
    L.flate_hip_compress_planned.argtypes = [vp, vp, vp, vp, u64p, i32p]
    L.flate_hip_compress_planned.restype = C.c_int
    L.flate_hip_plan_destroy.argtypes = [vp, vp]
    L.flate_hip_checksum.argtypes = [vp, vp, C.c_uint64, C.c_int, vp]
    L.flate_hip_checksum.restype = C.c_int
    L.flate_hip_checksum_combine.argtypes = [C.c_int, C.c_uint32, C.c_uint32, C.c_uint64]
    L.flate_hip_checksum_combine.restype = C.c_uint32
    L.flate_hip_debug_tokens.argtypes = [vp, C.c_uint32, vp, C.c_uint64]
    L.flate_hip_debug_tokens.restype = C.c_int64
    _lib = L
    return L


def status_name(code):
    return lib().flate_hip_status_name(int(code)).decode()
