/*
 * flate_hip.h -- C ABI of libflate_hip.so: the MI355X (gfx950) DEFLATE engine that
 * stands behind ianic/flate's compress / decompress / compressor / decompressor
 * API (reference: /root/reference/src/flate.zig:9-71, src/gzip.zig:4-66,
 * src/zlib.zig:4-66).
 *
 * The boundary is batch-shaped: one call processes n independent chunks
 * ("chunk" = one complete deflate/gzip/zlib stream, i.e. what one
 * `compress(reader, writer, options)` call of the reference produces, or what
 * one `decompress(reader, writer)` call consumes).  Plain pointers and sizes
 * only; no torch / HIP types in the signatures (hipStream_t travels as void*).
 *
 * What each entry point replaces in the reference:
 *   flate_hip_compress_batch    deflate.compress            deflate.zig:56-60
 *                               (= compressor + compress + finish, :138,304,344)
 *                               deflate.huffman.compress    deflate.zig:402-406
 *                               deflate.store.compress      deflate.zig:421-425
 *                               via flate.zig:28-30,44-47,59-62 and the gzip /
 *                               zlib twins (gzip.zig:23-25, zlib.zig:23-25)
 *   flate_hip_compress_flush    Compressor.write / flush / finish with LZ history kept
 *                               across flushes              deflate.zig:335-337, 344-347,
 *                               363-367 (sync flush: 00 00 ff ff, :276-278)
 *   flate_hip_decompress_batch  inflate.decompress          inflate.zig:14-17
 *                               via flate.zig:10-12, gzip.zig:5-7, zlib.zig:5-7
 *   flate_hip_compress_bound    (no reference twin: the Zig writer grows)
 *   status codes                the reference's error set, 1:1
 *                               (bit_reader.zig:29, container.zig:45-51,
 *                               huffman_decoder.zig:35-40, inflate.zig:72-78)
 *
 * All work runs on the GPU.  There is no CPU fallback: if no HIP device is
 * usable every call fails with FLATE_HIP_E_NO_DEVICE.
 */
#ifndef FLATE_HIP_H
#define FLATE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct flate_hip_ctx* flate_hip_handle;
typedef struct flate_hip_plan* flate_hip_plan_t;

/* container tag -- container.zig:18-21 */
enum { FLATE_HIP_RAW = 0, FLATE_HIP_GZIP = 1, FLATE_HIP_ZLIB = 2 };

/* mode: 0 = store-only (deflate.zig:420-434), 1 = huffman-only (:401-415),
 * 4..9 = Level.level_4 .. level_9 (:23-32; fast=4, default=6, best=9). */
enum { FLATE_HIP_MODE_STORE = 0, FLATE_HIP_MODE_HUFFMAN = 1, FLATE_HIP_MODE_DEFAULT = 6 };

/* where in / out / offsets / lengths / status live */
enum { FLATE_HIP_MEM_HOST = 0, FLATE_HIP_MEM_DEVICE = 1 };

/* call-level return codes */
enum {
    FLATE_HIP_OK = 0,
    FLATE_HIP_E_NO_DEVICE = -1,   /* no usable HIP device / kernel image */
    FLATE_HIP_E_INVALID_ARG = -2,
    FLATE_HIP_E_ALLOC = -3,       /* workspace allocation failed */
    FLATE_HIP_E_LAUNCH = -4,      /* a HIP call failed; see flate_hip_last_error */
    FLATE_HIP_E_UNSUPPORTED = -5  /* reserved */
};

/* per-chunk status: the reference's error names, same numbering as the oracle */
enum {
    FLATE_HIP_ST_OK = 0,
    FLATE_HIP_ST_END_OF_STREAM = 1,
    FLATE_HIP_ST_BAD_GZIP_HEADER = 2,
    FLATE_HIP_ST_BAD_ZLIB_HEADER = 3,
    FLATE_HIP_ST_WRONG_GZIP_CHECKSUM = 4,
    FLATE_HIP_ST_WRONG_GZIP_SIZE = 5,
    FLATE_HIP_ST_WRONG_ZLIB_CHECKSUM = 6,
    FLATE_HIP_ST_INVALID_CODE = 7,
    FLATE_HIP_ST_OVERSUBSCRIBED_HUFFMAN_TREE = 8,
    FLATE_HIP_ST_INCOMPLETE_HUFFMAN_TREE = 9,
    FLATE_HIP_ST_MISSING_END_OF_BLOCK_CODE = 10,
    FLATE_HIP_ST_INVALID_MATCH = 11,
    FLATE_HIP_ST_INVALID_BLOCK_TYPE = 12,
    FLATE_HIP_ST_WRONG_STORED_BLOCK_NLEN = 13,
    FLATE_HIP_ST_INVALID_DYNAMIC_BLOCK_HEADER = 14,
    FLATE_HIP_ST_OUTPUT_TOO_SMALL = 100,
    FLATE_HIP_ST_CHUNK_TOO_LARGE = 101, /* reserved; not produced any more: long inputs take the whole-stream path */
    /* Compress, levels 4..9, informational: out / out_len hold exactly the reference's bytes for this input, and those
     * bytes DO NOT inflate to the input.  The reference hands its block writer a full block of 32768 tokens before its
     * window has advanced over the last token's match (deflate.zig:227-230 -> 268-270 -> SlidingWindow.zig:119-123; the
     * advance is at deflate.zig:193), so the raw slice of that block ends up to 258 bytes early and the next block's slice
     * starts there; when exactly ONE of the two blocks goes out stored (block_writer.zig:369) the stream loses those bytes
     * (the first block stored) or holds them twice (the second).  Every inflater -- the reference's included -- then
     * reports a wrong checksum / size for gzip and zlib, and returns other bytes than went in for a raw stream.
     * See FLATE_HIP_DEFLATE_REPAIR_Q1. */
    FLATE_HIP_ST_REFERENCE_Q1_STREAM = 102
};

/* handle flags (flate_hip_set_flags).  bit0: compress at levels 4..9 hands every block the bytes its tokens cover -- the
 * reference with its window advanced before the tokens are flushed.  Streams differ from the reference's ONLY for inputs on
 * which a full token block ends in a match (the slices that decide "stored or Huffman" move by the match's length); they
 * always inflate to the input and status FLATE_HIP_ST_REFERENCE_Q1_STREAM is never produced.  Default 0: the reference's
 * bytes, whatever they decode to. */
enum { FLATE_HIP_DEFLATE_REPAIR_Q1 = 1 };
int flate_hip_set_flags(flate_hip_handle h, uint32_t flags);

/* decompress flags.  bit0: reference-strict dynamic block header (quirk Q6,
 * inflate.zig:161-180: a code-length repeat that crosses the HLIT/HDIST
 * boundary is rejected, although RFC 1951 3.2.7 allows it and the reference's
 * own encoder emits it).  Default 0: such a header is accepted when valid. */
enum { FLATE_HIP_INFLATE_STRICT_Q6 = 1 };

/* Levels 4..9: an input of up to this many bytes never slides the reference's window and takes
 * the batched chunk path (one workgroup per input); a longer one is compressed as ONE stream by
 * the whole-stream path, byte-identical to Deflate.compress over the whole input
 * (deflate.zig:304-321 with SlidingWindow.zig:36-44, Lookup.zig:43-51).  One level-4..9 input may be
 * at most 0xfff00000 bytes (stream positions are 32-bit); other modes 0xfffffff0. */
#define FLATE_HIP_MAX_LZ_CHUNK 65535u

int flate_hip_create(int device, flate_hip_handle* h);
int flate_hip_destroy(flate_hip_handle h);

/* run on this hipStream_t (NULL = the handle's own stream).  All calls are
 * synchronous on return for FLATE_HIP_MEM_HOST; for FLATE_HIP_MEM_DEVICE the
 * work is enqueued on the stream and the call returns without a device sync
 * unless flate_hip_set_sync(h, 1) (default 1). */
int flate_hip_set_stream(flate_hip_handle h, void* hip_stream);
int flate_hip_set_sync(flate_hip_handle h, int sync_on_return);

/* upper bound of one chunk's output for n input bytes */
size_t flate_hip_compress_bound(size_t n, int container, int mode);

/*
 * Compress n_chunks independent chunks.
 *   in       all input bytes; chunk i is in[in_off[i] .. in_off[i+1])
 *   in_off   n_chunks+1 offsets (bytes)
 *   out      output buffer; chunk i owns the slot out[out_off[i] .. out_off[i+1])
 *   out_off  n_chunks+1 slot starts (bytes), ascending; the slots must not
 *            overlap and `out` must be 4-byte aligned
 *   out_len  n_chunks produced lengths (bytes)
 *   status   n_chunks FLATE_HIP_ST_* codes
 * The bytes of chunk i are exactly what the reference writes for the same
 * input with the same container and level/mode (status 0, or 102 where the reference's own stream is
 * broken: FLATE_HIP_ST_REFERENCE_Q1_STREAM).
 * Host buffers (FLATE_HIP_MEM_HOST): a slot's bytes beyond out_len[i] are unspecified (zeros or
 * what the caller had there; a pinned `out` is written in place by the DMA engine and a kernel).
 */
int flate_hip_compress_batch(flate_hip_handle h, const uint8_t* in, const uint64_t* in_off,
                             uint32_t n_chunks, int container, int mode, uint8_t* out,
                             const uint64_t* out_off, uint64_t* out_len, int32_t* status,
                             int memkind);

/*
 * Asynchronous device batches.  flate_hip_compress_batch(MEM_DEVICE) has to read the offset arrays
 * back and build its chunk / block tables on the host before it can launch (two short blocking
 * copies per call).  A caller whose batch layout repeats -- the same chunk sizes and output slots call
 * after call, as in a pipeline that compresses batch k + 1 while batch k's output is on the wire --
 * plans it once from HOST copies of the offsets; flate_hip_compress_planned then only enqueues kernels on
 * the handle's stream (no allocation, no copy from or to the host, no wait; with set_sync(0) it returns
 * at once).  Inputs of more than 65535 bytes at levels 4..9 (the whole-stream path) are not plannable:
 * FLATE_HIP_E_UNSUPPORTED.  Same output bytes as flate_hip_compress_batch.
 */
int flate_hip_plan_compress(flate_hip_handle h, const uint64_t* in_off_host, const uint64_t* out_off_host,
                            uint32_t n_chunks, int container, int mode, flate_hip_plan_t* plan);
int flate_hip_compress_planned(flate_hip_handle h, flate_hip_plan_t plan, const uint8_t* in, uint8_t* out,
                               uint64_t* out_len, int32_t* status);
int flate_hip_plan_destroy(flate_hip_handle h, flate_hip_plan_t plan);

/*
 * One stream with sync-flush points (levels 4..9, huffman-only, store-only): what a Compressor
 * / SimpleCompressor of the reference (deflate.zig:335-337, 474-478) has
 * written after   write(in[0 .. p0]); flush(); write(in[p0 .. p1]); flush(); ...   and, when
 * `finish` is non-zero, write(rest); finish().  flush_pos: n_flush ascending stream positions
 * (<= n; equal neighbours = flush called twice).  Without `finish` the last flush point must be
 * n and the output ends with that flush's marker.  Output of a shorter prefix of the same call
 * sequence is a prefix of this output, so a streaming wrapper emits only what is new.
 * The LZ77 history survives a flush (deflate.zig:335-337): matches reach back across it, but
 * never run over it.  Host buffers only (memkind FLATE_HIP_MEM_HOST).
 */
int flate_hip_compress_flush(flate_hip_handle h, const uint8_t* in, uint64_t n, const uint64_t* flush_pos,
                             uint32_t n_flush, int finish, int container, int mode, uint8_t* out,
                             uint64_t out_cap, uint64_t* out_len, int32_t* status, int memkind);

/*
 * Decompress n_chunks independent streams (same argument shape).  out_len[i] is the
 * number of bytes produced; status[i] the reference's error name for a bad stream.
 * consumed (optional, may be NULL) receives the input bytes used by each stream, so a
 * caller can walk concatenated members the way Inflate.reset() does (inflate.zig:301-309).
 * Stream order: the call always waits once for the offsets; beyond that a batch of short streams is
 * only enqueued (set_sync(0)).  A batch with long streams (>= 128 KiB, at most 140 of them) is cut
 * into spans and BLOCKS ON THE HOST three to four times between its kernels (the chain of spans is
 * followed on the CPU): such a call cannot be captured in a graph or overlapped behind other work on the
 * stream.  FLATE_HIP_INFLATE_SPANS=0 (environment) or flag bit 0 keeps the old one-workgroup-per-stream path.
 * Host buffers (FLATE_HIP_MEM_HOST): a slot's bytes beyond out_len[i] may be overwritten with zeros.
 */
int flate_hip_decompress_batch(flate_hip_handle h, const uint8_t* in, const uint64_t* in_off,
                               uint32_t n_chunks, int container, int flags, uint8_t* out,
                               const uint64_t* out_off, uint64_t* out_len, int32_t* status,
                               uint64_t* consumed, int memkind);

/*
 * Pack the n_chunks produced streams back to back (device memory only): copies
 * out[out_off[i] .. out_off[i] + out_len[i]) to dst[dst_off[i] ..) with
 * dst_off[i] = sum of out_len[k] for k < i; dst_off has n_chunks + 1 entries (the last
 * one is the packed size).  This is the "reassemble the output bitstream" step that
 * precedes the RCCL all-gather of a sharded job; the Zig writer of the reference does
 * the same thing implicitly by writing streams one after another.
 */
int flate_hip_gather_streams(flate_hip_handle h, const uint8_t* out, const uint64_t* out_off,
                             const uint64_t* out_len, uint32_t n_chunks, uint8_t* dst, uint64_t* dst_off);

/*
 * Multi-GPU (one process per GPU): the caller shards the job's chunks over the ranks
 * (contiguous ranges balanced by bytes; inflate: by ISIZE); each rank hands ITS chunks to these entry
 * points together with an RCCL communicator (ncclComm_t as void*, made by the RCCL of the process --
 * torch.distributed's or the caller's own; the library binds to it at run time, it does not link
 * RCCL).  No collective touches the data path; the exchange is the reassembly of the output:
 *
 * compress: the local chunks are compressed exactly as flate_hip_compress_batch(MEM_DEVICE) does, the
 *   produced streams are packed back to back at gathered + rank * slice_bytes (dst_off: n_chunks + 1
 *   offsets inside the slice, device memory), the packed sizes are all-gathered into sizes[world]
 *   (device memory), and every slice goes to every peer in ONE grouped batch of ncclSend / ncclRecv
 *   (xGMI is point to point: each link carries one peer's shard, no ring).  slice_bytes must bound
 *   every rank's packed shard (e.g. the sum of its compress bounds) and be the same on all ranks.
 *   Every peer is sent the largest packed shard's worth of bytes (the sizes are read on the host once per call), not
 *   the slice's capacity.  Afterwards gathered[r * slice_bytes .. + sizes[r]) is rank r's output on every rank -- what the
 *   reference's writer would hold after compressing the ranks' chunks one after the other.
 * decompress: the local streams are inflated straight into this rank's slice (out_off relative to
 *   the slice), then the slices are exchanged the same way.
 * All device memory; work is enqueued on the handle's stream (set_sync decides about the final wait).
 * The reference has no counterpart: it is single-threaded; these replace the loop a
 * caller would write around compress() / decompress() per file.
 */
int flate_hip_compress_batch_sharded(flate_hip_handle h, void* nccl_comm, int rank, int world,
                                     const uint8_t* in, const uint64_t* in_off, uint32_t n_chunks,
                                     int container, int mode, uint8_t* out, const uint64_t* out_off,
                                     uint64_t* out_len, int32_t* status, uint8_t* gathered,
                                     uint64_t slice_bytes, uint64_t* sizes, uint64_t* dst_off);
int flate_hip_decompress_batch_sharded(flate_hip_handle h, void* nccl_comm, int rank, int world,
                                       const uint8_t* in, const uint64_t* in_off, uint32_t n_chunks,
                                       int container, int flags, uint8_t* gathered, uint64_t slice_bytes,
                                       const uint64_t* out_off, uint64_t* out_len, int32_t* status,
                                       uint64_t* consumed);

/* The containers' checksums on their own (container.zig:168-206: std.hash.Crc32 / Adler32 over the raw
 * input): container 1 = CRC-32, 2 = Adler-32 of a host buffer, computed by the checksum kernels; and the
 * checksum of a concatenation from the checksums of its parts (value_b over len_b bytes follows value_a).
 * The streaming compressor wrappers use them to write the footer of a stream they compressed piece by
 * piece (flate_hip_compress_flush on the tail of the stream only). */
int flate_hip_checksum(flate_hip_handle h, const uint8_t* data, uint64_t n, int container, uint32_t* value);
uint32_t flate_hip_checksum_combine(int container, uint32_t value_a, uint32_t value_b, uint64_t len_b);

const char* flate_hip_status_name(int status);
const char* flate_hip_last_error(flate_hip_handle h);
const char* flate_hip_version(void);

/* ---- measurement hooks (bench.py / tests) ---- */
/* When enabled every kernel launch is bracketed by hipEvents on the launch stream.
 * flate_hip_profile_read synchronises, folds the pending events into per-kernel
 * totals and returns them: names[i], total_ms[i], launches[i] for i < return value
 * (at most cap).  flate_hip_profile_reset clears the totals. */
int flate_hip_profile_enable(flate_hip_handle h, int enable);
int flate_hip_profile_read(flate_hip_handle h, const char** names, double* total_ms,
                           uint64_t* launches, int cap);
int flate_hip_profile_reset(flate_hip_handle h);

/* ---- test seam (the reference's own seam is the BlockWriterType parameter of
 * Deflate, deflate.zig:118-121: "so we can change that in test to test just
 * tokenization part").  After a level 4..9 compress_batch call, copies the token
 * list the tokenizer kernels produced for chunk `chunk` (same 32-bit token
 * encoding as the oracle: bit23 kind, bits15-22 len-3 / literal, bits0-14
 * dist-1) into host memory.  Returns the token count or a negative error. */
int64_t flate_hip_debug_tokens(flate_hip_handle h, uint32_t chunk, uint32_t* tokens, uint64_t cap);

/* ---- test seam for the block writer alone (the reference tests it the same way: token lists
 * straight into BlockWriter.write / dynamicBlock, block_writer.zig:599-706).  Runs the DEVICE
 * planner, offset scan and bit packer on a caller-supplied token list (<= 32768 tokens, encoding as
 * above) and optional raw input (`input` NULL = the Zig `null`: the block cannot be stored) and
 * writes the block, padded to a byte as BlockWriter.flush does, to host memory.  dynamic_only = 1
 * plans the block as BlockWriter.dynamicBlock does (block_writer.zig:395-432). */
int flate_hip_debug_write_block(flate_hip_handle h, const uint32_t* tokens, uint32_t n_tokens,
                                const uint8_t* input, uint32_t input_len, int eof, int dynamic_only,
                                uint8_t* out, uint64_t out_cap, uint64_t* out_len);

/* Tuning aid: shader-clock timestamps that workgroup 0 of the tokenizer kernels took at its
 * phase boundaries during the last call (slots: sort 0-7, match 8-10, parse 16-23). */
int flate_hip_debug_phase_cycles(flate_hip_handle h, uint64_t* out, int n);

/* The FLATE_HIP_* tuning variables are read once, in flate_hip_create: no call path reads the
 * environment.  This reads them again (the test suite changes them between calls of one handle). */
int flate_hip_debug_reload_env(flate_hip_handle h);

#ifdef __cplusplus
}
#endif
#endif
