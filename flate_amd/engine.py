"""Engine: one handle of the HIP DEFLATE engine on one device (one process per GPU).

Host-buffer entry points (`compress_many`, `decompress_many`) take / return Python
bytes; device entry points take raw device pointers (e.g. torch tensors'
`data_ptr()`), keep everything resident in HBM and run on the caller's stream.
"""
import ctypes as C
import os

import numpy as np

from . import _capi
from ._capi import FlateHipError, MEM_DEVICE, MEM_HOST


class Engine:
    def __init__(self, device=0):
        self._L = _capi.lib()
        self._h = C.c_void_p()
        rc = self._L.flate_hip_create(int(device), C.byref(self._h))
        if rc != 0:
            import sys
            hint = ""
            if "torch" in sys.modules and os.environ.get("FLATE_HIP_PRELOAD_TORCH_HIP", "0") in ("", "0"):
                # (two HIP runtimes in one process: the second to come up finds no device -- flate_amd/_capi.py)
                hint = ("; this process also runs PyTorch, which ships its own libamdhip64: import torch BEFORE flate_amd, "
                        "or set FLATE_HIP_PRELOAD_TORCH_HIP=1 before importing flate_amd, so that both use one runtime")
            raise FlateHipError("flate_hip_create(device=%d) failed with %d: no usable MI355X / HIP device "
                                "(there is no CPU fallback)%s" % (device, rc, hint))
        self.device = device

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._L.flate_hip_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            err = self._L.flate_hip_last_error(self._h).decode()
            raise FlateHipError("%s failed with %d (%s)" % (what, rc, err))

    # ---- configuration ----
    _KNOBS = ("FLATE_HIP_MAX_PASS_CHUNKS", "FLATE_HIP_HOST_PASS_CHUNKS", "FLATE_HIP_MAX_STREAM_PASS_MIB",
              "FLATE_HIP_INFLATE_SPANS", "FLATE_HIP_SPAN_DEBUG", "FLATE_HIP_SPAN_TWIN", "FLATE_HIP_NO_PIN_MIRROR",
              "FLATE_HIP_NO_RAMP", "FLATE_HIP_INFLATE_PAR", "FLATE_HIP_INFLATE_RING", "FLATE_HIP_RECT", "FLATE_HIP_STREAM_WINDOWS", "FLATE_HIP_SIMPLE_CK_INLINE", "FLATE_HIP_STREAM_GROUP", "FLATE_HIP_SPAN_TWO_RUNS", "FLATE_HIP_MEMSET_INLINE", "FLATE_HIP_ONE_COMPUTE_STREAM")

    def _sync_env(self):
        """The library reads its FLATE_HIP_* tuning variables once, when the handle is made.  Tests and probes change them
        between calls of one engine: when this process's view of them has changed, the handle is told to read them again."""
        now = tuple(os.environ.get(k) for k in self._KNOBS)
        if now != getattr(self, "_knob_state", None):
            if getattr(self, "_knob_state", None) is not None or any(v is not None for v in now):
                self._L.flate_hip_debug_reload_env(self._h)
            self._knob_state = now

    def set_stream(self, stream_ptr):
        self._L.flate_hip_set_stream(self._h, C.c_void_p(stream_ptr or 0))

    def set_sync(self, flag):
        self._L.flate_hip_set_sync(self._h, int(bool(flag)))

    def set_flags(self, flags):
        """_capi.DEFLATE_REPAIR_Q1: levels 4..9 hand every block the bytes its tokens cover (streams that always inflate
        to their input; they differ from the reference's only where the reference's own stream is broken: status 102)."""
        self._check(self._L.flate_hip_set_flags(self._h, int(flags)), "flate_hip_set_flags")

    def compress_bound(self, n, container=0, mode=6):
        return self._L.flate_hip_compress_bound(int(n), container, mode)

    # ---- host buffers ----
    def compress_many(self, chunks, container=0, mode=6):
        """chunks: sequence of bytes-like.  Returns (list of bytes, list of status codes)."""
        self._sync_env()
        n = len(chunks)
        if n == 0:
            return [], []
        lens = np.array([len(c) for c in chunks], dtype=np.uint64)
        in_off = np.zeros(n + 1, dtype=np.uint64)
        np.cumsum(lens, out=in_off[1:])
        blob = np.frombuffer(b"".join(bytes(c) for c in chunks), dtype=np.uint8)
        if blob.size == 0:
            blob = np.zeros(1, dtype=np.uint8)
        caps = np.array([(self.compress_bound(int(l), container, mode) + 7) & ~7 for l in lens], dtype=np.uint64)
        out_off = np.zeros(n + 1, dtype=np.uint64)
        np.cumsum(caps, out=out_off[1:])
        out = np.zeros(int(out_off[-1]) + 8, dtype=np.uint8)
        out_len = np.zeros(n, dtype=np.uint64)
        status = np.zeros(n, dtype=np.int32)
        rc = self._L.flate_hip_compress_batch(self._h, blob.ctypes.data, in_off.ctypes.data, n, container, mode,
                                              out.ctypes.data, out_off.ctypes.data, out_len.ctypes.data,
                                              status.ctypes.data, MEM_HOST)
        self._check(rc, "flate_hip_compress_batch")
        res = [out[int(out_off[i]): int(out_off[i]) + int(out_len[i])].tobytes() for i in range(n)]
        return res, [int(s) for s in status]

    def compress_flush(self, data, flush_points, finish=True, container=0, mode=6):
        """One stream with sync-flush points (Compressor.write / flush / finish, deflate.zig:335-367).
        Returns (bytes, status)."""
        self._sync_env()
        blob = np.frombuffer(bytes(data), dtype=np.uint8) if len(data) else np.zeros(1, dtype=np.uint8)
        fp = np.array(list(flush_points), dtype=np.uint64)
        cap = (self.compress_bound(len(data), container, mode) + 64 * (len(fp) + 1) + 7) & ~7
        out = np.zeros(cap + 8, dtype=np.uint8)
        out_len = np.zeros(1, dtype=np.uint64)
        status = np.zeros(1, dtype=np.int32)
        rc = self._L.flate_hip_compress_flush(self._h, blob.ctypes.data, len(data), fp.ctypes.data if len(fp) else None,
                                              len(fp), 1 if finish else 0, container, mode, out.ctypes.data, cap,
                                              out_len.ctypes.data, status.ctypes.data, MEM_HOST)
        self._check(rc, "flate_hip_compress_flush")
        return out[: int(out_len[0])].tobytes(), int(status[0])

    def checksum(self, data, container):
        """CRC-32 (container 1) / Adler-32 (container 2) of a host buffer, by the checksum kernels."""
        blob = np.frombuffer(bytes(data), dtype=np.uint8) if len(data) else np.zeros(1, dtype=np.uint8)
        v = np.zeros(1, dtype=np.uint32)
        rc = self._L.flate_hip_checksum(self._h, blob.ctypes.data, len(data), container, v.ctypes.data)
        self._check(rc, "flate_hip_checksum")
        return int(v[0])

    def checksum_combine(self, container, a, b, len_b):
        return int(self._L.flate_hip_checksum_combine(container, a, b, len_b))

    def decompress_many(self, streams, container=0, flags=0, caps=None):
        """streams: sequence of bytes-like.  caps: output capacity per stream.  Default: for gzip the ISIZE
        field of the stream's last 8 bytes (container.zig:92-96) plus slack, and whatever then reports
        OutputTooSmall (more members behind the first, a damaged footer) is decoded again with the worst
        case of 1100 output bytes per input byte; raw / zlib streams get the worst case at once.
        Returns (list of bytes, list of status codes, list of consumed input bytes)."""
        self._sync_env()
        n = len(streams)
        if n == 0:
            return [], [], []
        if caps is None and container == 1:  # gzip
            worst = [max(1 << 16, len(c) * 1100 + 1024) for c in streams]
            guess = [min(w, int.from_bytes(bytes(c[-4:]), "little") + 64) if len(c) >= 18 else w
                     for c, w in zip(streams, worst)]
            res, st, cons = self.decompress_many(streams, container, flags, guess)
            redo = [i for i in range(n) if st[i] == 100 and guess[i] < worst[i]]  # FLATE_HIP_ST_OUTPUT_TOO_SMALL
            if redo:
                r2, s2, c2 = self.decompress_many([streams[i] for i in redo], container, flags, [worst[i] for i in redo])
                for k, i in enumerate(redo):
                    res[i], st[i], cons[i] = r2[k], s2[k], c2[k]
            return res, st, cons
        lens = np.array([len(c) for c in streams], dtype=np.uint64)
        in_off = np.zeros(n + 1, dtype=np.uint64)
        np.cumsum(lens, out=in_off[1:])
        blob = np.frombuffer(b"".join(bytes(c) for c in streams), dtype=np.uint8)
        if blob.size == 0:
            blob = np.zeros(1, dtype=np.uint8)
        if caps is None:
            caps = [max(1 << 16, int(l) * 1100 + 1024) for l in lens]
        caps = np.array([(int(c) + 7) & ~7 for c in caps], dtype=np.uint64)
        out_off = np.zeros(n + 1, dtype=np.uint64)
        np.cumsum(caps, out=out_off[1:])
        out = np.zeros(int(out_off[-1]) + 8, dtype=np.uint8)
        out_len = np.zeros(n, dtype=np.uint64)
        status = np.zeros(n, dtype=np.int32)
        consumed = np.zeros(n, dtype=np.uint64)
        rc = self._L.flate_hip_decompress_batch(self._h, blob.ctypes.data, in_off.ctypes.data, n, container, flags,
                                                out.ctypes.data, out_off.ctypes.data, out_len.ctypes.data,
                                                status.ctypes.data, consumed.ctypes.data, MEM_HOST)
        self._check(rc, "flate_hip_decompress_batch")
        res = [out[int(out_off[i]): int(out_off[i]) + int(out_len[i])].tobytes() for i in range(n)]
        return res, [int(s) for s in status], [int(c) for c in consumed]

    # ---- device buffers (raw pointers; everything already in HBM) ----
    def compress_device(self, in_ptr, in_off_ptr, n_chunks, container, mode, out_ptr, out_off_ptr, out_len_ptr,
                        status_ptr):
        self._sync_env()
        rc = self._L.flate_hip_compress_batch(self._h, in_ptr, in_off_ptr, n_chunks, container, mode, out_ptr,
                                              out_off_ptr, out_len_ptr, status_ptr, MEM_DEVICE)
        self._check(rc, "flate_hip_compress_batch")

    def plan_compress(self, in_off, out_off, container, mode):
        """Plan a device batch whose layout repeats (host offset arrays, n + 1 entries each); returns a
        handle for compress_planned / plan_destroy."""
        self._sync_env()
        a = np.ascontiguousarray(in_off, dtype=np.uint64)
        b = np.ascontiguousarray(out_off, dtype=np.uint64)
        plan = C.c_void_p()
        rc = self._L.flate_hip_plan_compress(self._h, a.ctypes.data, b.ctypes.data, a.size - 1, container, mode,
                                             C.byref(plan))
        self._check(rc, "flate_hip_plan_compress")
        return plan

    def compress_planned(self, plan, in_ptr, out_ptr, out_len_ptr, status_ptr):
        """Enqueue one planned batch: kernels only, nothing touches the host."""
        rc = self._L.flate_hip_compress_planned(self._h, plan, in_ptr, out_ptr, out_len_ptr, status_ptr)
        self._check(rc, "flate_hip_compress_planned")

    def plan_destroy(self, plan):
        self._L.flate_hip_plan_destroy(self._h, plan)

    def decompress_device(self, in_ptr, in_off_ptr, n_chunks, container, flags, out_ptr, out_off_ptr, out_len_ptr,
                          status_ptr, consumed_ptr=None):
        self._sync_env()
        rc = self._L.flate_hip_decompress_batch(self._h, in_ptr, in_off_ptr, n_chunks, container, flags, out_ptr,
                                                out_off_ptr, out_len_ptr, status_ptr, consumed_ptr, MEM_DEVICE)
        self._check(rc, "flate_hip_decompress_batch")

    def gather_streams_device(self, out_ptr, out_off_ptr, out_len_ptr, n_chunks, dst_ptr, dst_off_ptr):
        """Pack the produced streams back to back in device memory (dst_off gets n_chunks + 1 entries)."""
        rc = self._L.flate_hip_gather_streams(self._h, out_ptr, out_off_ptr, out_len_ptr, n_chunks, dst_ptr,
                                              dst_off_ptr)
        self._check(rc, "flate_hip_gather_streams")

    # ---- measurement / test seams ----
    def profile_enable(self, flag=True):
        self._L.flate_hip_profile_enable(self._h, int(bool(flag)))

    def profile_reset(self):
        self._L.flate_hip_profile_reset(self._h)

    def profile_read(self):
        """{kernel name: (total_ms, launches)}"""
        cap = 32
        names = (C.c_char_p * cap)()
        ms = (C.c_double * cap)()
        cnt = (C.c_uint64 * cap)()
        n = self._L.flate_hip_profile_read(self._h, names, ms, cnt, cap)
        return {names[i].decode(): (ms[i], int(cnt[i])) for i in range(max(n, 0))}

    def phase_cycles(self):
        """Shader-clock timestamps of workgroup 0's phases in the last tokenizer launch (tuning aid)."""
        buf = np.zeros(160, dtype=np.uint64)
        self._L.flate_hip_debug_phase_cycles(self._h, buf.ctypes.data, 160)
        return buf

    def debug_write_block(self, tokens, input_bytes, eof, dynamic_only=False):
        """One block from a token list through the device planner / offset scan / bit packer
        (input_bytes None = the Zig null).  Returns the block's bytes."""
        tok = np.ascontiguousarray(tokens, dtype=np.uint32)
        inp = None if input_bytes is None else np.frombuffer(bytes(input_bytes), dtype=np.uint8)
        cap = 8 * tok.size + (0 if inp is None else inp.size) + 1024
        out = np.zeros(cap + 8, dtype=np.uint8)
        out_len = np.zeros(1, dtype=np.uint64)
        keep = np.zeros(1, dtype=np.uint8)  # a non-NULL pointer for an empty input
        rc = self._L.flate_hip_debug_write_block(
            self._h, tok.ctypes.data if tok.size else None, tok.size,
            None if inp is None else (inp.ctypes.data if inp.size else keep.ctypes.data),
            0 if inp is None else inp.size, int(bool(eof)), int(bool(dynamic_only)), out.ctypes.data, cap,
            out_len.ctypes.data)
        self._check(rc, "flate_hip_debug_write_block")
        return out[: int(out_len[0])].tobytes()

    def debug_tokens(self, chunk):
        """Token list the tokenizer kernels produced for `chunk` of the last level 4..9 call."""
        buf = np.zeros(65536, dtype=np.uint32)
        n = self._L.flate_hip_debug_tokens(self._h, int(chunk), buf.ctypes.data, buf.size)
        if n > buf.size:  # whole-stream pass: the count comes back even when the buffer is short
            buf = np.zeros(n, dtype=np.uint32)
            n = self._L.flate_hip_debug_tokens(self._h, int(chunk), buf.ctypes.data, buf.size)
        if n < 0:
            raise FlateHipError("flate_hip_debug_tokens failed with %d" % n)
        return buf[:n].copy()


_default = None


def default_engine():
    global _default
    if _default is None:
        _default = Engine(0)
    return _default
