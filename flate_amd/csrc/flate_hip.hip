// flate_hip.hip -- host side of libflate_hip.so: the C ABI of include/flate_hip.h.
// Owns the device workspace, builds the chunk / block tables, launches the
// kernels on the caller's HIP stream.  There is no CPU execution path: without a
// usable gfx950 device every entry point fails.
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <chrono>
#include <deque>
#include <vector>
#include <thread>
#include <functional>

#include "../../include/flate_hip.h"
#include "kernels_block.h"
#include "kernels_common.h"
#include "kernels_inflate.h"
#include "kernels_inflate_par.h"
#include "kernels_lz.h"
#include "kernels_parse.h"
#include "kernels_walk.h"
#include "kernels_stream.h"
#include "stream_tables.h"

namespace {

enum KernelId {
    K_MEMSET = 0,
    K_BYTE_HIST,
    K_CHECKSUM,
    K_LZ_SORT,
    K_LZ_MATCH,
    K_LZ_CHAIN,
    K_LZ_PARSE,
    K_LZ_LINKS,
    K_LZ_WALK,
    K_LZ_EMIT,
    K_ST_PARSE,
    K_ST_EMIT,
    K_PLAN,
    K_OFFSETS,
    K_ENCODE,
    K_INFLATE,
    K_INFLATE_PAR,
    K_SPAN_SCAN,
    K_INFLATE_SPAN,
    K_GATHER,
    K_COUNT
};
const char* const kKernelNames[K_COUNT] = {"memset_out", "k_byte_hist", "k_checksum", "k_lz_sort", "k_lz_match",
                                           "k_lz_chain", "k_lz_parse", "k_lz_links", "k_lz_walk", "k_lz_emit",
                                           "k_st_parse", "k_st_emit", "k_plan",
                                           "k_offsets",  "k_encode",    "k_inflate",  "k_inflate_par", "k_span_scan", "k_inflate_span",
                                           "k_gather"};

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
};

}  // namespace

// Offsets and per-pass tables of a compress batch whose layout does not change from call to call
// (flate_hip_plan_compress): with them a call only enqueues kernels.
struct flate_hip_plan {
    std::vector<uint64_t> hin, hout;
    uint32_t n_chunks = 0;
    int container = 0, mode = 0;
    struct Pass {
        uint32_t nc = 0, nb = 0;
        void* chunks = nullptr;     // fl_chunk[nc]
        void* blk_chunk = nullptr;  // uint32_t[nb]
    };
    std::vector<Pass> passes;
    bool ready = false;
};

// The tuning knobs of the environment, read ONCE when the handle is made -- no getenv on any call path
// (flate_hip_debug_reload_env reads them again: the test suite's seam).
struct fl_knobs {
    size_t max_pass_chunks = 32768;       // FLATE_HIP_MAX_PASS_CHUNKS
    size_t host_pass_chunks = 1024;       // FLATE_HIP_HOST_PASS_CHUNKS
    uint64_t stream_pass_bytes = 4096ull << 20;  // FLATE_HIP_MAX_STREAM_PASS_MIB
    uint64_t span_min_bytes = 0;          // FLATE_HIP_INFLATE_SPANS (0 = never); set to the default below
    bool span_debug = false;              // FLATE_HIP_SPAN_DEBUG
    int span_twin = -1;                   // FLATE_HIP_SPAN_TWIN: -1 unset, 0 never, 2..950 where to cut
    bool memset_inline = false;           // FLATE_HIP_MEMSET_INLINE: the output slots are cleared in the caller's stream (round 4's way; tuning)
    bool span_two_runs = false;           // FLATE_HIP_SPAN_TWO_RUNS: round 4's two decodes per span instead of one in symbols (tests, tuning)
    bool no_pin_mirror = false;           // FLATE_HIP_NO_PIN_MIRROR
    bool no_ramp = false;                 // FLATE_HIP_NO_RAMP
    uint32_t stream_group = 0;            // FLATE_HIP_STREAM_GROUP: windows per group of the whole-stream path (0: by the number of streams; tuning / tests)
    int stream_windows = -1;              // FLATE_HIP_STREAM_WINDOWS: -1 unset (by estimate), 0 never, 1 whenever possible (kernels_parse.h, k_lz_parse<true>)
    bool simple_ck_inline = false;        // FLATE_HIP_SIMPLE_CK_INLINE: the simple modes' checksum on the compute stream (round 4's way)
    bool one_stream = false;              // FLATE_HIP_ONE_COMPUTE_STREAM: the pinned path's sub-batches one after the other on the caller's stream (round 5's way; tuning)
    int rect = -1;                        // FLATE_HIP_RECT: 1 = half of every slot goes home by the DMA engine's rectangle copy (off by default: see there)
    int64_t inflate_par = -1;             // FLATE_HIP_INFLATE_PAR: -1 unset, 0 never, else the minimum stream size
    int64_t inflate_ring = -1;            // FLATE_HIP_INFLATE_RING: -1 unset
};

struct flate_hip_ctx {
    int device = 0;
    fl_knobs knobs;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    bool sync = true;
    uint32_t flags = 0;  // flate_hip_set_flags
    std::string last_error;
    fl_crc_consts crc{};
    // device workspace (grown on demand, reused across calls)
    DevBuf chunks, blk_chunk, plans, hist, cks, S, NC, rec, desc, marks, tokens, ntok, cflag, links, shard_sz;
    DevBuf wexit;  // per window and sub-pass: where the path left it; per group: its exit, its entry; a flag (kernels_parse.h, fix launch)
    DevBuf wchunks, swins;  // whole-stream passes on k_lz_parse<true>: the streams' windows as chunks, a table entry per stream
    void* pin_in = nullptr;   // pinned mirrors of pageable host buffers (compress_impl)
    void* pin_out = nullptr;
    void* pin_len = nullptr;  // out_len of a sub-batch on its way home (mirror_out reads it before the call ends)
    void* pin_tab = nullptr;  // the pinned path's per-pass tables on their way to the device (a copy from PAGEABLE memory is staged by the runtime)
    size_t pin_tab_cap = 0;
    bool in_mirror = false;   // compress_impl is running on the mirrors (no second level of them)
    size_t pin_in_cap = 0, pin_out_cap = 0, pin_len_cap = 0;
    // pageable callers: a sub-batch's input is copied into the mirror right before its H2D copy is enqueued, its produced
    // bytes out of the mirror as soon as they have landed -- while the GPU works on the other sub-batches
    std::function<void(uint32_t, uint32_t)> mirror_in, mirror_out;
    uint32_t n_cu = 0;  // of the device (spans)
    DevBuf sp_points, sp_found, sp_spans, sp_res, sp_cand, sp_candoff, sp_tails, sp_tails_b, sp_chain, sp_chainoff,
        sp_pool, sp_pooltab, sp_poolctl, sp_items, sp_part, sp_footoff, sp_foot, sp_fin, sp_chainpos, sp_rs;  // inflate of long streams by spans
    DevBuf tiles, segs, pieces, fpts, zones, nsorted, jmp, exitmap, entry, segtok, tokbase, bound;  // whole-stream passes
    DevBuf sgroups, sgroup0, gmap, gentry, sblocks;
    DevBuf st_in, st_out, st_inoff, st_outlen, st_status, st_consumed, st_pack, st_packoff, st_slot;
    // host-buffer calls with pinned memory: copy streams beside the compute stream, events between them
    hipStream_t s_in = nullptr, s_out = nullptr;
    std::vector<hipEvent_t> xfer_events;
    // the container's checksum runs on a stream of its own beside the tokenizer (it reads the input and nothing else;
    // k_offsets waits for it)
    hipStream_t s_ck = nullptr;
    hipEvent_t ck_ev0 = nullptr, ck_ev1 = nullptr;
    hipStream_t s_c2 = nullptr;  // pinned path, round 6: every other sub-batch's kernels run here, beside the caller's stream (compress_impl)
    hipEvent_t c2_ev0 = nullptr, c2_ev1 = nullptr;
    hipStream_t s_ms = nullptr;  // the output slots are cleared beside the tokenizer / the histograms (round 5)
    hipEvent_t ms_ev0 = nullptr, ms_ev1 = nullptr;
    bool ms_pending = false;
    bool ck_pending = false;
    // last level 4..9 call, for the debug seam
    uint32_t dbg_pass_chunks = 0;
    uint32_t dbg_first_chunk = 0;
    std::vector<uint64_t> dbg_pos_off;
    std::vector<fl_chunk> dbg_chunks;   // whole-stream pass: chunk and piece tables of the last pass
    std::vector<fl_piece> dbg_pieces;
    // profiling
    bool prof = false;
    struct Pending {
        int kid;
        hipEvent_t a, b;
    };
    std::vector<Pending> pending;
    std::vector<hipEvent_t> free_events;
    double prof_ms[K_COUNT] = {0};
    uint64_t prof_n[K_COUNT] = {0};
};

namespace {

#define HIP_OK(h, expr)                                                                                  \
    do {                                                                                                 \
        hipError_t e_ = (expr);                                                                          \
        if (e_ != hipSuccess) {                                                                          \
            (h)->last_error = std::string(#expr) + ": " + hipGetErrorString(e_);                         \
            return FLATE_HIP_E_LAUNCH;                                                                   \
        }                                                                                                \
    } while (0)

int ensure(flate_hip_ctx* h, DevBuf& b, size_t bytes) {
    if (bytes <= b.cap) return FLATE_HIP_OK;
    if (b.p) {
        (void)hipStreamSynchronize(h->stream);
        (void)hipFree(b.p);
        b.p = nullptr;
        b.cap = 0;
    }
    size_t want = bytes + bytes / 8 + 256;
    hipError_t e = hipMalloc(&b.p, want);
    if (e != hipSuccess) {
        want = bytes;
        e = hipMalloc(&b.p, want);
    }
    if (e != hipSuccess) {
        h->last_error = std::string("hipMalloc(") + std::to_string(bytes) + "): " + hipGetErrorString(e);
        b.p = nullptr;
        return FLATE_HIP_E_ALLOC;
    }
    b.cap = want;
    return FLATE_HIP_OK;
}

hipEvent_t get_event(flate_hip_ctx* h) {
    if (!h->free_events.empty()) {
        hipEvent_t e = h->free_events.back();
        h->free_events.pop_back();
        return e;
    }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
}

struct ProfScope {
    flate_hip_ctx* h;
    int kid;
    hipStream_t s;
    hipEvent_t a{}, b{};
    ProfScope(flate_hip_ctx* h_, int kid_, hipStream_t s_ = nullptr) : h(h_), kid(kid_), s(s_ ? s_ : h_->stream) {
        if (h->prof) {
            a = get_event(h);
            b = get_event(h);
            (void)hipEventRecord(a, s);
        }
    }
    ~ProfScope() {
        if (h->prof) {
            (void)hipEventRecord(b, s);
            h->pending.push_back({kid, a, b});
        }
    }
};

// k_checksum beside the kernels that follow on the compute stream; enqueue_back_end joins it
int launch_checksum_side(flate_hip_ctx* h, uint32_t nb, const uint8_t* d_in, const fl_chunk* dch, const uint32_t* dbc,
                         const fl_sblock* dsb, const fl_params& prm);

void fold_profile(flate_hip_ctx* h) {
    if (h->pending.empty()) return;
    (void)hipStreamSynchronize(h->stream);
    for (auto& p : h->pending) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            h->prof_ms[p.kid] += ms;
            h->prof_n[p.kid] += 1;
        }
        h->free_events.push_back(p.a);
        h->free_events.push_back(p.b);
    }
    h->pending.clear();
}

void init_crc_consts(fl_crc_consts& cc) {
    cc.xpow8[0] = 0x00800000u;  // x^8 in the reflected representation (x^0 = 0x80000000)
    for (int j = 1; j < 32; j++) cc.xpow8[j] = fl_crc_mulmod(cc.xpow8[j - 1], cc.xpow8[j - 1]);
    for (int m = 0; m < 64; m++) cc.pow1024[m] = fl_crc_xpow8n(cc.xpow8, 1024ull * m);
    cc.pow65535 = fl_crc_xpow8n(cc.xpow8, 65535);
}

bool level_args(int mode, fl_params& p) {  // deflate.zig:41-52
    switch (mode) {
        case 4: p.good = 4; p.lazy = 4; p.nice = 16; p.chain = 16; return true;
        case 5: p.good = 8; p.lazy = 16; p.nice = 32; p.chain = 32; return true;
        case 6: p.good = 8; p.lazy = 16; p.nice = 128; p.chain = 128; return true;
        case 7: p.good = 8; p.lazy = 32; p.nice = 128; p.chain = 256; return true;
        case 8: p.good = 32; p.lazy = 128; p.nice = 258; p.chain = 1024; return true;
        case 9: p.good = 32; p.lazy = 258; p.nice = 258; p.chain = 4096; return true;
        default: p.good = p.lazy = p.nice = p.chain = 0; return mode == 0 || mode == 1;
    }
}

void read_knobs(fl_knobs& k, uint64_t span_default) {
    k = fl_knobs();
    const char* e;
    if ((e = getenv("FLATE_HIP_MAX_PASS_CHUNKS")) && atoi(e) > 0) k.max_pass_chunks = (size_t)atoi(e);
    if ((e = getenv("FLATE_HIP_HOST_PASS_CHUNKS")) && atoi(e) > 0) k.host_pass_chunks = (size_t)atoi(e);
    if ((e = getenv("FLATE_HIP_MAX_STREAM_PASS_MIB")) && atoll(e) > 0) k.stream_pass_bytes = (uint64_t)atoll(e) << 20;
    e = getenv("FLATE_HIP_INFLATE_SPANS");
    k.span_min_bytes = e ? (uint64_t)atoll(e) : span_default;
    k.span_debug = getenv("FLATE_HIP_SPAN_DEBUG") != nullptr;
    if ((e = getenv("FLATE_HIP_SPAN_TWIN"))) k.span_twin = atoi(e);
    k.no_pin_mirror = getenv("FLATE_HIP_NO_PIN_MIRROR") != nullptr;
    k.no_ramp = getenv("FLATE_HIP_NO_RAMP") != nullptr;
    if ((e = getenv("FLATE_HIP_RECT"))) k.rect = atoi(e) != 0;
    k.one_stream = getenv("FLATE_HIP_ONE_COMPUTE_STREAM") != nullptr;
    k.simple_ck_inline = getenv("FLATE_HIP_SIMPLE_CK_INLINE") != nullptr;
    if ((e = getenv("FLATE_HIP_SPAN_TWO_RUNS"))) k.span_two_runs = atoi(e) != 0;
    if ((e = getenv("FLATE_HIP_MEMSET_INLINE"))) k.memset_inline = atoi(e) != 0;
    if ((e = getenv("FLATE_HIP_STREAM_WINDOWS"))) k.stream_windows = atoi(e) != 0;
    if ((e = getenv("FLATE_HIP_STREAM_GROUP")) && atoi(e) > 0) k.stream_group = (uint32_t)atoi(e);
    if ((e = getenv("FLATE_HIP_INFLATE_PAR"))) k.inflate_par = atoll(e);
    if ((e = getenv("FLATE_HIP_INFLATE_RING"))) k.inflate_ring = atoll(e);
}
size_t pass_chunk_limit(const flate_hip_ctx* h) { return h->knobs.max_pass_chunks; }
// host-buffer calls whose buffers are pinned run in sub-batches of this many chunks, so that the H2D copy of
// sub-batch k + 1 and the D2H copy of k - 1 overlap the kernels of k
size_t host_pass_chunk_limit(const flate_hip_ctx* h) { return h->knobs.host_pass_chunks; }
bool is_pinned_host(const void* p) {
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError();  // plain pageable memory: not an error of ours
        return false;
    }
    return a.type == hipMemoryTypeHost;
}
// The copy streams of the host-buffer paths.  HIP multiplexes the streams of one priority onto a few hardware queues (four by
// default), torch's and the caller's included, and two streams that land on one queue run IN ORDER: with the input stream and the
// output stream on one queue, the input copy of sub-batch k + 1 sits behind the output copy of k, which waits for the kernels of
// k -- nothing overlaps any more (rocprofv3 timeline, profiles/r05_host_path.txt: H2D 1.19 + kernels 2.15 + D2H 0.6 ms one after
// the other, 16.6 ms per 256 MiB instead of 10.4; which way a process went was luck).  Queues are pooled per PRIORITY: the
// input stream gets the highest, the output stream the lowest, the kernels stay on the caller's (normal): three pools.
hipError_t create_copy_stream(hipStream_t* s, bool input) {
    int least = 0, greatest = 0;  // (numerically: greatest priority = lowest number)
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess || least == greatest) {
        (void)hipGetLastError();
        return hipStreamCreateWithFlags(s, hipStreamNonBlocking);
    }
    return hipStreamCreateWithPriority(s, hipStreamNonBlocking, input ? greatest : least);
}

int xfer_event(flate_hip_ctx* h, size_t k, hipEvent_t* ev) {
    while (h->xfer_events.size() <= k) {
        hipEvent_t e;
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return FLATE_HIP_E_ALLOC;
        h->xfer_events.push_back(e);
    }
    *ev = h->xfer_events[k];
    return FLATE_HIP_OK;
}

// whole-stream passes: uncompressed bytes per pass (about 30 bytes of scratch per input byte)
uint64_t stream_pass_byte_limit(const flate_hip_ctx* h) { return h->knobs.stream_pass_bytes; }

// Levels 4..9, whole-stream pass: tokenizer kernels (kernels_stream.h).  Leaves tokens,
// histograms and the block table for the shared back end.
#ifndef FL_STREAM_FIX_MAX
#define FL_STREAM_FIX_MAX 3u  // fix launches of a grouped whole-stream pass before it is handed to the sort / match tiles
#endif
int compress_stream_pass(flate_hip_ctx* h, const uint8_t* d_in, const fl_params& prm, uint32_t nc, uint32_t nb,
                         const StreamTables& t, const fl_chunk* hch /* the pass's chunks, host copy */) {
    hipStream_t st = h->stream;
    int rc;
    const uint32_t nseg = (uint32_t)t.segs.size(), npc = (uint32_t)t.pieces.size();
    const uint64_t npos = t.npos;
    const size_t tile_limit = pass_chunk_limit(h);
    const size_t tiles_per_launch = std::min(t.tiles.size(), tile_limit);
    if ((rc = ensure(h, h->tiles, sizeof(fl_tile) * t.tiles.size()))) return rc;
    if ((rc = ensure(h, h->segs, sizeof(fl_seg) * (t.segs.size() + 1)))) return rc;
    if ((rc = ensure(h, h->pieces, sizeof(fl_piece) * npc))) return rc;
    if ((rc = ensure(h, h->fpts, sizeof(uint32_t) * (t.fpts.size() + 1)))) return rc;
    if ((rc = ensure(h, h->zones, sizeof(uint32_t) * (t.zones.size() + 1)))) return rc;
    if ((rc = ensure(h, h->nsorted, sizeof(uint32_t) * tiles_per_launch))) return rc;
    if ((rc = ensure(h, h->cflag, sizeof(uint32_t) * tiles_per_launch))) return rc;
    if ((rc = ensure(h, h->S, tiles_per_launch * FL_CHUNK_STRIDE * sizeof(uint16_t)))) return rc;
    if ((rc = ensure(h, h->desc, npos * sizeof(uint32_t)))) return rc;
    if ((rc = ensure(h, h->tokens, npos * sizeof(uint32_t)))) return rc;
    if ((rc = ensure(h, h->marks, npos / 8))) return rc;
    // what only the sort / match tiles and the stitch over records need (about 14 bytes per input byte): not reserved for a
    // pass that takes the windows of k_lz_parse<true>
    auto ensure_tile_workspace = [&]() -> int {
        int r;
        if ((r = ensure(h, h->NC, tiles_per_launch * FL_CHUNK_STRIDE * sizeof(uint32_t)))) return r;
        if ((r = ensure(h, h->rec, npos * 2 * sizeof(uint32_t) + 64))) return r;
        if ((r = ensure(h, h->jmp, npos * sizeof(uint16_t)))) return r;
        if ((r = ensure(h, h->exitmap, ((size_t)nseg + 1) * FL_SEG_ENTRIES * sizeof(uint16_t)))) return r;
        // positions a flush keeps out of the hash table get no record from the match finder
        if (t.any_flush && hipMemsetAsync(h->rec.p, 0, npos * 2 * sizeof(uint32_t), st) != hipSuccess) return FLATE_HIP_E_LAUNCH;
        return 0;
    };
    if ((rc = ensure(h, h->entry, sizeof(uint32_t) * (nseg + 1)))) return rc;
    if ((rc = ensure(h, h->segtok, sizeof(uint32_t) * (nseg + 1)))) return rc;
    if ((rc = ensure(h, h->tokbase, sizeof(uint32_t) * (nseg + 1)))) return rc;
    if ((rc = ensure(h, h->bound, sizeof(uint32_t) * 2 * nb))) return rc;  // [nb] window positions + [nb] Q1 gaps (k_st_emit)
    if ((rc = ensure(h, h->ntok, sizeof(uint32_t) * npc))) return rc;
    HIP_OK(h, hipMemcpyAsync(h->tiles.p, t.tiles.data(), sizeof(fl_tile) * t.tiles.size(), hipMemcpyHostToDevice, st));
    if (nseg) HIP_OK(h, hipMemcpyAsync(h->segs.p, t.segs.data(), sizeof(fl_seg) * nseg, hipMemcpyHostToDevice, st));
    HIP_OK(h, hipMemcpyAsync(h->pieces.p, t.pieces.data(), sizeof(fl_piece) * npc, hipMemcpyHostToDevice, st));
    if (!t.fpts.empty())
        HIP_OK(h, hipMemcpyAsync(h->fpts.p, t.fpts.data(), sizeof(uint32_t) * t.fpts.size(), hipMemcpyHostToDevice, st));
    if (!t.zones.empty())
        HIP_OK(h, hipMemcpyAsync(h->zones.p, t.zones.data(), sizeof(uint32_t) * t.zones.size(), hipMemcpyHostToDevice, st));
    HIP_OK(h, hipMemsetAsync(h->hist.p, 0, sizeof(uint32_t) * 320 * (size_t)nb, st));
    HIP_OK(h, hipMemsetAsync(h->marks.p, 0, npos / 8, st));
    HIP_OK(h, hipStreamSynchronize(st));  // the host vectors must outlive the async copies

    const fl_chunk* dch = (const fl_chunk*)h->chunks.p;
    const fl_seg* dsg = (const fl_seg*)h->segs.p;
    const fl_piece* dpc = (const fl_piece*)h->pieces.p;
    const uint32_t* dfp = (const uint32_t*)h->fpts.p;
    // Round 5: many streams, none of them with flush points, levels 4-7: the demand-driven tokenizer of the chunk path walks
    // every stream's windows in order, a workgroup per stream (kernels_parse.h, k_lz_parse<true>) -- no sort, no records for
    // every position.  A window costs a workgroup about 0.28 ms, the sort / match pair about 0.061 ms per MiB of all CUs.
    // Round 6: levels 8 and 9 the same way on k_lz_links / k_lz_walk<true, true> (kernels_walk.h).
    const bool deep_walk = prm.chain >= FL_BULK_MIN_CHAIN;
    // (sync-flush points: levels 4-7 only -- k_lz_chain keeps the three positions before a flush point out of the table and
    // k_lz_parse<true> ends every match there; the run logic of k_lz_walk counts on runs whose every position is in the table)
    bool windows = !(t.any_flush && deep_walk) && h->knobs.stream_windows != 0 && nseg;
    std::vector<fl_chunk> wch;
    std::vector<fl_swin> sws;
    bool grouped = false;  // some stream is split into groups of windows: a fix launch follows (kernels_parse.h)
    uint32_t round_cap = 0;  // rounds of a window's stitch after which the pass is given to the tiles (0: never)
    if (windows) {
        if (h->n_cu == 0) {
            int v = 0;
            if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, h->device) != hipSuccess || v <= 0) v = 256;
            h->n_cu = (uint32_t)v;
        }
        uint64_t bytes = 0, total_win = 0;
        for (uint32_t i = 0; i < nc; i++) {
            bytes += hch[i].in_len;
            total_win += hch[i].n_slides + 1u;
        }
        // Many streams: a workgroup walks a whole stream (no guess, no fix).  Fewer than the chip holds workgroups at a time (one a
        // CU; levels 8-9: two): groups of G windows, as many groups as it holds, so that they all end together and the fix launch
        // -- which parses one window (a sub-pass) per group again -- is one round: G + 1 window times.  (Round 5 took four groups
        // per CU of at least four windows: one 1 MiB stream at level 6 cost 1.9 ms against 1.05 as 32 groups of one window, and
        // 1.34 on the sort / match tiles; one 177 MB stream at level 9 as 675 groups of 8 took 2 x 8 + 2 window times, as 491
        // groups of 11 it takes 11 + 1.  tools/small_stream_round.sh)
        const uint64_t slots = deep_walk ? 2ull * h->n_cu : (uint64_t)h->n_cu;
        uint32_t G = ~0u;
        if (nc < slots) G = (uint32_t)std::max<uint64_t>(1, (total_win + slots - 1) / slots);
        if (h->knobs.stream_group) G = h->knobs.stream_group;
        for (uint32_t i = 0; i < nc; i++) {
            const fl_chunk& c = hch[i];
            const uint32_t nw = c.n_slides + 1u, w0 = (uint32_t)wch.size();
            for (uint32_t j = 0; j < nw; j++) {
                fl_chunk w{};
                w.in_off = c.in_off + (uint64_t)FL_SEG * j;
                w.in_len = (uint32_t)std::min<uint64_t>(65536u, (uint64_t)c.in_len - (uint64_t)FL_SEG * j);
                w.pad_ = 1u;  // (a window: k_lz_chain builds its chains whatever it holds)
                w.piece0 = FL_SEG * j;  // the window's position in its stream, and the stream's flush points (k_lz_chain)
                w.flush_off = c.flush_off;
                w.n_flush = c.n_flush;
                // (+ the bytes behind the window that the lazy calls of its last anchor look at: kernels_parse.h, kernels_walk.h)
                w.pad_ |= (uint32_t)std::min<uint64_t>(264u, (uint64_t)c.in_len - (uint64_t)FL_SEG * j - w.in_len) << 8;
                wch.push_back(w);
            }
            uint32_t prev = ~0u;
            for (uint32_t j = 0; j < nw; j += G) {
                fl_swin sw{};
                sw.chunk = i;
                sw.win0 = w0 + j;
                sw.nwin = std::min(G, nw - j);
                sw.wfirst = j;
                sw.prev = prev;
                prev = (uint32_t)sws.size();
                if (j) grouped = true;
                sws.push_back(sw);
                if (G == ~0u) break;
            }
        }
        const uint32_t ng = (uint32_t)sws.size();
        const uint32_t gmax = G == ~0u ? (uint32_t)0 : G;
        uint32_t max_win = 0;
        for (const fl_swin& sw : sws) max_win = std::max(max_win, sw.nwin);
        (void)gmax;
        // A window costs a workgroup about 0.28 ms (levels 8-9: 1.2, text 1.1, TAR-like 1.6), the fix launch one more per round and
        // a host wait; sort / match cost about 0.061 ms per MiB on all CUs (levels 8-9: 0.11 text, 0.20 TAR-like) and 0.8 (1.3) ms
        // of kernel latencies whatever the size.
        if (bytes < (64ull << 20) && nc < slots) round_cap = 24u;
        const uint64_t rounds = (ng + slots - 1) / slots;
        const double win_ms = deep_walk ? 1.5 : 0.28;
        const double est_new = (double)rounds * max_win * win_ms + (grouped ? (double)rounds * win_ms + 0.1 : 0.0);
        const double est_old = (double)bytes / 1048576.0 * (deep_walk ? 0.11 : 0.061) + (deep_walk ? 1.3 : 0.8);
        if (h->knobs.stream_windows < 0 && est_new >= est_old) windows = false;
        if (wch.size() > tile_limit) windows = false;  // (the chain links of all windows at once: 128 KiB each)
        // (levels 8-9: 512 KiB of links a window -- 16 GiB for a 1 GiB stream; a device that cannot give them: the tiles)
        if (windows && deep_walk && ensure(h, h->links, wch.size() * FL_CHUNK_STRIDE * 4 * sizeof(uint16_t))) {
            (void)hipGetLastError();
            windows = false;
        }
    }
    if (windows) {
        const uint32_t nw = (uint32_t)wch.size(), ng = (uint32_t)sws.size();
        if ((rc = ensure(h, h->wchunks, sizeof(fl_chunk) * nw))) return rc;
        if ((rc = ensure(h, h->swins, sizeof(fl_swin) * ng))) return rc;
        if (deep_walk) {
            if ((rc = ensure(h, h->links, (size_t)nw * FL_CHUNK_STRIDE * 4 * sizeof(uint16_t)))) return rc;
        } else {
            if ((rc = ensure(h, h->S, (size_t)nw * FL_CHUNK_STRIDE * sizeof(uint16_t)))) return rc;
        }
        if ((rc = ensure(h, h->cflag, sizeof(uint32_t) * nw))) return rc;
        if ((rc = ensure(h, h->wexit, sizeof(uint32_t) * (2 * (size_t)nw + 2 * (size_t)ng + 4)))) return rc;
        uint32_t* d_wexit = (uint32_t*)h->wexit.p;
        uint32_t* d_gexit = d_wexit + 2 * (size_t)nw;
        uint32_t* d_gentry = d_gexit + ng;
        uint32_t* d_dirty = d_gentry + ng;
        HIP_OK(h, hipMemcpyAsync(h->wchunks.p, wch.data(), sizeof(fl_chunk) * nw, hipMemcpyHostToDevice, st));
        HIP_OK(h, hipMemcpyAsync(h->swins.p, sws.data(), sizeof(fl_swin) * ng, hipMemcpyHostToDevice, st));
        HIP_OK(h, hipStreamSynchronize(st));  // the host vectors must outlive the async copies
        const fl_chunk* dwc = (const fl_chunk*)h->wchunks.p;
        // the tokenizer over the groups of windows: the first launch (fix = 0), or one from the groups' true entries (fix = 1)
        // (a small pass of few streams -- where round 5 took the tiles -- gives a window up after 24 rounds of its stitch: periodic data)
        auto launch_tokenizer = [&](uint32_t fix) {
            fix |= round_cap << 8;
            if (deep_walk) {
                ProfScope ps(h, K_LZ_WALK);
                wk_stream sp{(const fl_swin*)h->swins.p, dch, (const uint32_t*)h->zones.p, d_gexit, d_gentry, d_wexit, d_dirty, fix};
                hipLaunchKernelGGL((k_lz_walk<true, true>), dim3(ng), dim3(WK_THREADS), 0, st, d_in, dwc, prm, (const uint16_t*)h->links.p,
                                   (const uint32_t*)h->cflag.p, (uint32_t*)h->desc.p, (uint32_t*)h->marks.p, sp);
            } else {
                ProfScope ps(h, K_LZ_PARSE);
                hipLaunchKernelGGL(k_lz_parse<true>, dim3(ng), dim3(PZ_THREADS), 0, st, d_in, dwc, prm,
                                   (const uint16_t*)h->S.p, (const uint32_t*)h->cflag.p, (uint32_t*)h->desc.p, (uint32_t*)h->marks.p,
                                   (const fl_swin*)h->swins.p, dch, (const uint32_t*)h->zones.p, d_gexit, d_gentry, d_wexit, d_dirty, fix,
                                   t.any_flush ? dfp : (const uint32_t*)nullptr);
            }
        };
        if (deep_walk) {
            ProfScope ps(h, K_LZ_LINKS);
            hipLaunchKernelGGL(k_lz_links<0>, dim3(nw), dim3(64 * FL_CHAIN_WAVES), 0, st, d_in, dwc, (uint16_t*)h->links.p, (uint32_t*)h->cflag.p);
            hipLaunchKernelGGL(k_lz_links<1>, dim3(nw), dim3(64 * FL_CHAIN_WAVES), 0, st, d_in, dwc, (uint16_t*)h->links.p, (uint32_t*)h->cflag.p);
            hipLaunchKernelGGL(k_lz_links<2>, dim3(nw), dim3(64 * FL_CHAIN_WAVES), 0, st, d_in, dwc, (uint16_t*)h->links.p, (uint32_t*)h->cflag.p);
            hipLaunchKernelGGL(k_lz_links<3>, dim3(nw), dim3(64 * FL_CHAIN_WAVES), 0, st, d_in, dwc, (uint16_t*)h->links.p, (uint32_t*)h->cflag.p);
        } else {
            ProfScope ps(h, K_LZ_CHAIN);
            if (t.any_flush)
                hipLaunchKernelGGL(k_lz_chain<true>, dim3(nw), dim3(64 * FL_CHAIN_WAVES), 0, st, d_in, dwc,
                                   (uint16_t*)h->S.p, (uint32_t*)h->cflag.p, (uint32_t*)nullptr, dfp);
            else
                hipLaunchKernelGGL(k_lz_chain<false>, dim3(nw), dim3(64 * FL_CHAIN_WAVES), 0, st, d_in, dwc,
                                   (uint16_t*)h->S.p, (uint32_t*)h->cflag.p, (uint32_t*)nullptr, (const uint32_t*)nullptr);
        }
        HIP_OK(h, hipMemsetAsync(d_dirty, 0, sizeof(uint32_t), st));
        launch_tokenizer(0u);
        // the groups that were parsed from a guess: again from where the group before them leaves, until nothing moves any more
        // (one launch in practice: a parse falls in step within a few bytes; the loop is what makes it exact).  NOT on
        // periodic data: in a stream of one repeated byte every anchor is a 258-byte match, a parse from a guessed entry never
        // meets the true one, and every launch settles one more group (about ng launches).  After FL_STREAM_FIX_MAX
        // launches the pass is given to the sort / match tiles, which cost the same whatever the data.
        bool gave_up = false;
        {
            // (bit 31: some window's stitch did not settle within the round cap -- periodic data -- and its workgroup has stopped)
            uint32_t flag = 0;
            HIP_OK(h, hipMemcpyAsync(&flag, d_dirty, sizeof flag, hipMemcpyDeviceToHost, st));
            HIP_OK(h, hipStreamSynchronize(st));
            gave_up = (flag & 0x80000000u) != 0;
        }
        for (uint32_t it = 0; grouped && !gave_up; it++) {
            HIP_OK(h, hipMemsetAsync(d_dirty, 0, sizeof(uint32_t), st));
            launch_tokenizer(1u);
            uint32_t flag = 0;
            HIP_OK(h, hipMemcpyAsync(&flag, d_dirty, sizeof flag, hipMemcpyDeviceToHost, st));
            HIP_OK(h, hipStreamSynchronize(st));
            if (!flag) break;
            // (`flag` = the groups that were parsed to their end with a new exit.  Text: none after the first fix launch.  More than
            // an eighth of them: periodic data -- every launch settles one group per stream -- and no reason to try twice more)
            if (it + 1 >= FL_STREAM_FIX_MAX || flag > ng / 8 + 1) {  // (incl. bit 31: a window that does not settle)
                gave_up = true;
                break;
            }
        }
        if (gave_up) {
            windows = false;  // the tiles below; the anchors marked so far go
            HIP_OK(h, hipMemsetAsync(h->marks.p, 0, npos / 8, st));
            if ((rc = ensure(h, h->nsorted, sizeof(uint32_t) * tiles_per_launch))) return rc;
            if ((rc = ensure(h, h->cflag, sizeof(uint32_t) * tiles_per_launch))) return rc;
            if ((rc = ensure(h, h->S, tiles_per_launch * FL_CHUNK_STRIDE * sizeof(uint16_t)))) return rc;
        } else {
            ProfScope ps(h, K_ST_PARSE);
            hipLaunchKernelGGL(k_st_count, dim3(nseg), dim3(FL_PARSE_THREADS), 0, st, dch, dpc, dsg, (const uint32_t*)h->desc.p,
                               (const uint32_t*)h->marks.p, (uint32_t*)h->segtok.p);
        }
    }
    if (!windows && (rc = ensure_tile_workspace())) return rc;
    for (size_t t0 = 0; !windows && t0 < t.tiles.size(); t0 += tiles_per_launch) {
        const uint32_t nt = (uint32_t)std::min(tiles_per_launch, t.tiles.size() - t0);
        const fl_tile* dti = (const fl_tile*)h->tiles.p + t0;
        {
            ProfScope ps(h, K_LZ_SORT);
            hipLaunchKernelGGL(k_lz_sort<true>, dim3(nt), dim3(FL_SORT_THREADS), 0, st, d_in, dch, dti, dfp,
                               (uint32_t*)h->nsorted.p, (uint16_t*)h->S.p,
                               (uint32_t*)h->cflag.p);
        }
        {
            ProfScope ps(h, K_LZ_MATCH);
            const uint32_t* cf = (const uint32_t*)h->cflag.p;
            hipLaunchKernelGGL((k_lz_match<true, true, false>), dim3(nt), dim3(FL_MATCH_THREADS), 0, st, d_in, dch, dti,
                               dfp, (const uint32_t*)h->nsorted.p, prm, (const uint16_t*)h->S.p, (uint32_t*)h->NC.p,
                               (uint32_t*)h->rec.p, cf);
            // the tiles k_lz_sort marked runny
            hipLaunchKernelGGL((k_lz_match<true, true, true>), dim3(nt), dim3(FL_MATCH_THREADS), 0, st, d_in, dch,
                               dti, dfp, (const uint32_t*)h->nsorted.p, prm, (const uint16_t*)h->S.p,
                               (uint32_t*)h->NC.p, (uint32_t*)h->rec.p, cf);
        }
    }
    if (nseg && !windows) {
        ProfScope ps(h, K_ST_PARSE);
        hipLaunchKernelGGL(k_st_parse1, dim3(nseg), dim3(FL_PARSE_THREADS), 0, st, dch, dpc, dsg, prm,
                           (const uint32_t*)h->rec.p, (uint32_t*)h->desc.p, (uint16_t*)h->jmp.p,
                           (uint16_t*)h->exitmap.p);
        // the walk from segment to segment: directly for short pieces, over groups of segments
        // when some piece is long (kernels_stream.h)
        uint32_t max_seg = 0;
        for (const fl_piece& pc : t.pieces) max_seg = std::max(max_seg, pc.n_seg);
        if (max_seg <= 4 * FL_STITCH_GROUP) {
            hipLaunchKernelGGL(k_st_stitch, dim3((npc + 63) / 64), dim3(64), 0, st, dpc, npc, dsg,
                               (const uint16_t*)h->exitmap.p, (uint32_t*)h->entry.p);
        } else {
            std::vector<fl_sgroup> groups;
            std::vector<uint32_t> group0(npc);
            for (uint32_t i = 0; i < npc; i++) {
                group0[i] = (uint32_t)groups.size();
                for (uint32_t g = 0; g * FL_STITCH_GROUP < t.pieces[i].n_seg; g++) groups.push_back(fl_sgroup{i, g});
            }
            const uint32_t ngr = (uint32_t)groups.size();
            if ((rc = ensure(h, h->sgroups, sizeof(fl_sgroup) * (ngr + 1)))) return rc;
            if ((rc = ensure(h, h->sgroup0, sizeof(uint32_t) * npc))) return rc;
            if ((rc = ensure(h, h->gmap, ((size_t)ngr + 1) * FL_SEG_ENTRIES * sizeof(uint16_t)))) return rc;
            if ((rc = ensure(h, h->gentry, sizeof(uint32_t) * (ngr + 1)))) return rc;
            HIP_OK(h, hipMemcpyAsync(h->sgroups.p, groups.data(), sizeof(fl_sgroup) * ngr, hipMemcpyHostToDevice, st));
            HIP_OK(h, hipMemcpyAsync(h->sgroup0.p, group0.data(), sizeof(uint32_t) * npc, hipMemcpyHostToDevice, st));
            HIP_OK(h, hipStreamSynchronize(st));
            if (ngr) {
                hipLaunchKernelGGL(k_st_stitch_a, dim3(ngr), dim3(FL_SEG_ENTRIES), 0, st, dpc,
                                   (const fl_sgroup*)h->sgroups.p, dsg, (const uint16_t*)h->exitmap.p,
                                   (uint16_t*)h->gmap.p);
                hipLaunchKernelGGL(k_st_stitch_b, dim3((npc + 63) / 64), dim3(64), 0, st, dpc, npc,
                                   (const uint32_t*)h->sgroup0.p, dsg, (const uint16_t*)h->gmap.p,
                                   (uint32_t*)h->gentry.p);
                hipLaunchKernelGGL(k_st_stitch_c, dim3((ngr + 63) / 64), dim3(64), 0, st, dpc,
                                   (const fl_sgroup*)h->sgroups.p, ngr, dsg, (const uint16_t*)h->exitmap.p,
                                   (const uint32_t*)h->gentry.p, (uint32_t*)h->entry.p);
            }
        }
        hipLaunchKernelGGL(k_st_parse2, dim3(nseg), dim3(FL_PARSE_THREADS), 0, st, dch, dpc, dsg,
                           (const uint32_t*)h->desc.p, (const uint16_t*)h->jmp.p, (const uint32_t*)h->entry.p,
                           (uint32_t*)h->marks.p, (uint32_t*)h->segtok.p);
    }
    {
        ProfScope ps(h, K_ST_EMIT);
        hipLaunchKernelGGL(k_st_scan, dim3(npc), dim3(64), 0, st, dpc, (const uint32_t*)h->segtok.p,
                           (uint32_t*)h->tokbase.p, (uint32_t*)h->ntok.p);
        if (nseg)
            hipLaunchKernelGGL(k_st_emit, dim3(nseg), dim3(FL_EMIT_THREADS), 0, st, d_in, dch, dpc, dsg, prm,
                               (const uint32_t*)h->desc.p, (const uint32_t*)h->marks.p,
                               (const uint32_t*)h->tokbase.p, (uint32_t*)h->tokens.p, (uint32_t*)h->hist.p,
                               (uint32_t*)h->bound.p, (uint32_t*)h->bound.p + nb);
        hipLaunchKernelGGL(k_st_blocks, dim3(npc), dim3(64), 0, st, dch, dpc, (const uint32_t*)h->ntok.p,
                           (const uint32_t*)h->bound.p, (const uint32_t*)h->bound.p + nb, prm.flags & FL_PRM_REPAIR_Q1,
                           (const uint32_t*)h->zones.p, (fl_block_plan*)h->plans.p);
    }
    (void)nc;
    return FLATE_HIP_OK;
}

// k_gather_copy: about 4096 workgroups whatever the number of streams (one long stream is cut into slices)
dim3 gather_grid(uint32_t n) { return dim3(n, std::max(1u, std::min(1024u, 4096u / std::max(n, 1u)))); }

// Host-buffer calls: bring back only what was produced.  The slots are sized for the worst case (an
// inflate caller may reserve 1000x the input); when the produced bytes are a small part of the slot
// range they are packed on the device first (k_scan_lens + k_gather_copy), cross PCIe once, and are put
// into their slots by the host; otherwise the range up to the last produced byte is copied as it is:
// bytes of a slot beyond out_len[i] are then whatever the staging buffer held there (the callers clear it:
// zeros), in the packed case they are not touched.
int copy_out_host(flate_hip_ctx* h, const uint8_t* d_out, const uint64_t* d_outlen, uint32_t n,
                  const std::vector<uint64_t>& hout, uint64_t out_shift, uint8_t* out, const uint64_t* out_len) {
    hipStream_t st = h->stream;
    const uint64_t out_lo = hout[0], out_hi = hout[n];
    if (out_hi <= out_lo) return FLATE_HIP_OK;
    uint64_t total = 0;
    for (uint32_t i = 0; i < n; i++) total += out_len[i];
    if (total == 0) return FLATE_HIP_OK;
    if (total * 2 >= out_hi - out_lo) {
        // dense: the range from the first slot to the end of the last produced byte
        uint64_t end = out_lo;
        for (uint32_t i = 0; i < n; i++)
            if (out_len[i]) end = std::max(end, hout[i] + out_len[i]);
        HIP_OK(h, hipMemcpyAsync(out + out_lo, d_out + (out_lo - out_shift), end - out_lo, hipMemcpyDeviceToHost, st));
        HIP_OK(h, hipStreamSynchronize(st));
        return FLATE_HIP_OK;
    }
    int rc;
    if ((rc = ensure(h, h->st_pack, total + 16))) return rc;
    if ((rc = ensure(h, h->st_packoff, sizeof(uint64_t) * ((size_t)n + 1)))) return rc;
    if ((rc = ensure(h, h->st_slot, sizeof(uint64_t) * ((size_t)n + 1)))) return rc;
    std::vector<uint64_t> slot(n + 1);
    for (uint32_t i = 0; i <= n; i++) slot[i] = hout[i] - out_shift;
    HIP_OK(h, hipMemcpyAsync(h->st_slot.p, slot.data(), sizeof(uint64_t) * (n + 1), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_scan_lens, dim3(1), dim3(1024), 0, st, d_outlen, n, (uint64_t*)h->st_packoff.p);
    hipLaunchKernelGGL(k_gather_copy, gather_grid(n), dim3(256), 0, st, d_out, (const uint64_t*)h->st_slot.p, d_outlen,
                       (uint8_t*)h->st_pack.p, (const uint64_t*)h->st_packoff.p);
    HIP_OK(h, hipGetLastError());
    std::vector<uint8_t> packed(total);
    HIP_OK(h, hipMemcpyAsync(packed.data(), h->st_pack.p, total, hipMemcpyDeviceToHost, st));
    HIP_OK(h, hipStreamSynchronize(st));
    uint64_t at = 0;
    for (uint32_t i = 0; i < n; i++) {
        if (out_len[i]) memcpy(out + hout[i], packed.data() + at, out_len[i]);
        at += out_len[i];
    }
    return FLATE_HIP_OK;
}

// Fetch the n+1 offsets to the host (the block tables are built there).
int fetch_offsets(flate_hip_ctx* h, const uint64_t* off, uint32_t n, int memkind, std::vector<uint64_t>& host) {
    host.resize((size_t)n + 1);
    if (memkind == FLATE_HIP_MEM_HOST) {
        memcpy(host.data(), off, sizeof(uint64_t) * ((size_t)n + 1));
    } else {
        HIP_OK(h, hipMemcpyAsync(host.data(), off, sizeof(uint64_t) * ((size_t)n + 1), hipMemcpyDeviceToHost,
                                 h->stream));
        HIP_OK(h, hipStreamSynchronize(h->stream));
    }
    for (uint32_t i = 0; i < n; i++)
        if (host[i + 1] < host[i]) return FLATE_HIP_E_INVALID_ARG;
    return FLATE_HIP_OK;
}


// workspace of a chunk-path pass of nc chunks (levels 4..9)
int ensure_lz_workspace(flate_hip_ctx* h, uint32_t nc, uint32_t chain) {
    int rc;
    const size_t per = (size_t)nc * FL_CHUNK_STRIDE;
    if (chain >= FL_BULK_MIN_CHAIN) {
        if ((rc = ensure(h, h->links, per * 4 * sizeof(uint16_t)))) return rc;  // per chunk [L4 | L6 | L8 | RK] (kernels_walk.h)
    } else {
        if ((rc = ensure(h, h->S, per * sizeof(uint16_t)))) return rc;          // chain links (kernels_parse.h)
    }
    if ((rc = ensure(h, h->desc, per * sizeof(uint32_t)))) return rc;   // anchor descriptors
    if ((rc = ensure(h, h->marks, per / 8))) return rc;                 // true anchors, one bit per position
    if ((rc = ensure(h, h->tokens, per * sizeof(uint32_t)))) return rc;
    if ((rc = ensure(h, h->ntok, sizeof(uint32_t) * nc))) return rc;
    if ((rc = ensure(h, h->cflag, sizeof(uint32_t) * nc))) return rc;
    return FLATE_HIP_OK;
}

// shared back end of every pass: block planner, offset scan, bit packer
int launch_checksum_side(flate_hip_ctx* h, uint32_t nb, const uint8_t* d_in, const fl_chunk* dch, const uint32_t* dbc,
                         const fl_sblock* dsb, const fl_params& prm) {
    hipStream_t st = h->stream;
    if (!h->s_ck) {  // (all three or none: a half-made set would be skipped by every later call)
        hipStream_t s = nullptr;
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&e0, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&e1, hipEventDisableTiming) != hipSuccess) {
            if (e0) (void)hipEventDestroy(e0);
            if (e1) (void)hipEventDestroy(e1);
            if (s) (void)hipStreamDestroy(s);
            (void)hipGetLastError();
            return FLATE_HIP_E_ALLOC;
        }
        h->s_ck = s;
        h->ck_ev0 = e0;
        h->ck_ev1 = e1;
    }
    // behind everything enqueued so far (the pass's tables and input, the kernels that read the checksums of the pass before)
    HIP_OK(h, hipEventRecord(h->ck_ev0, st));
    HIP_OK(h, hipStreamWaitEvent(h->s_ck, h->ck_ev0, 0));
    {
        ProfScope ps(h, K_CHECKSUM, h->s_ck);
        hipLaunchKernelGGL(k_checksum, dim3(nb), dim3(64), 0, h->s_ck, d_in, dch, dbc, dsb, prm, h->crc, (uint32_t*)h->cks.p);
    }
    HIP_OK(h, hipEventRecord(h->ck_ev1, h->s_ck));
    h->ck_pending = true;
    return FLATE_HIP_OK;
}

#ifndef FL_ENC_WAVE_MIN_SLOTS_STREAM
#define FL_ENC_WAVE_MIN_SLOTS_STREAM 32768u
#endif
int enqueue_back_end(flate_hip_ctx* h, const fl_params& prm, uint32_t nc, uint32_t nb, uint32_t c0, const fl_chunk* dch,
                     const uint32_t* dbc, const fl_sblock* dsb, const uint8_t* d_in, uint8_t* d_out, uint64_t* d_outlen,
                     int32_t* d_status) {
    hipStream_t st = h->stream;
    const int mode = prm.mode;
    fl_block_plan* dpl = (fl_block_plan*)h->plans.p;
    uint32_t* dhist = (uint32_t*)h->hist.p;
    uint32_t* dcks = (uint32_t*)h->cks.p;
    {
        ProfScope ps(h, K_PLAN);
        if (mode == 0)
            hipLaunchKernelGGL(k_plan_store, dim3((nb + 255) / 256), dim3(256), 0, st, dch, dbc, dsb, nb, dpl);
        else
            hipLaunchKernelGGL(k_plan, dim3((nb + FL_PLAN_WAVES - 1) / FL_PLAN_WAVES), dim3(64 * FL_PLAN_WAVES), 0, st,
                               dch, dbc, dsb, prm, (const uint32_t*)dhist, dpl);
    }
    if (h->ck_pending) {  // the checksums of this pass (launch_checksum_side)
        HIP_OK(h, hipStreamWaitEvent(st, h->ck_ev1, 0));
        h->ck_pending = false;
    }
    if (h->ms_pending) {  // the output slots have been cleared (compress_impl)
        HIP_OK(h, hipStreamWaitEvent(st, h->ms_ev1, 0));
        h->ms_pending = false;
    }
    {
        ProfScope ps(h, K_OFFSETS);
        // (a wave per chunk; sixteen when the chunks are long streams of many blocks: config #4's one stream has 2049)
        hipLaunchKernelGGL(k_offsets, dim3(nc), dim3((uint64_t)nb >= 32ull * nc ? 64 * FL_OFFS_MAX_WAVES : 64), 0, st, dch, prm, h->crc, dpl, (const uint32_t*)dcks,
                           d_out, d_outlen + c0, d_status + c0);
    }
    {
        ProfScope ps(h, K_ENCODE);
        // many blocks (the chunk path: two plan slots per chunk): a wave per block, one pass over its tokens; few blocks
        // (long streams): a workgroup per block, its waves share the block
        // (whole-stream passes: a block holds 32768 tokens and most slots are empty -- 256 streams of 1 MiB have 3072 blocks in 8448
        // slots: four waves a block are faster than one until the blocks are many)
        if (mode >= 4 && nb >= (prm.stream ? FL_ENC_WAVE_MIN_SLOTS_STREAM : 8192u))
            hipLaunchKernelGGL(k_encode_wave<true>, dim3((nb + FL_ENC_WAVES - 1) / FL_ENC_WAVES), dim3(64 * FL_ENC_WAVES), 0, st,
                               d_in, dch, dbc, (const fl_block_plan*)dpl, (const uint32_t*)h->tokens.p, (uint32_t*)d_out, nb,
                               (prm.stream || (nb & 1u)) ? 0u : 1u);
        else if (mode >= 4)
            hipLaunchKernelGGL(k_encode<true>, dim3(nb), dim3(64 * FL_ENC_WAVES), 0, st, d_in, dch, dbc,
                               (const fl_block_plan*)dpl, (const uint32_t*)h->tokens.p, (uint32_t*)d_out);
        else
            hipLaunchKernelGGL(k_encode<false>, dim3(nb), dim3(64 * FL_ENC_WAVES), 0, st, d_in, dch, dbc,
                               (const fl_block_plan*)dpl, (const uint32_t*)nullptr, (uint32_t*)d_out);
    }
    HIP_OK(h, hipGetLastError());
    return FLATE_HIP_OK;
}

// one pass of the chunk path (levels 4..9, inputs <= 65535 bytes) or of a simple mode: checksums,
// tokenizer / histograms, back end.  Only enqueues.
int enqueue_pass(flate_hip_ctx* h, const fl_params& prm, uint32_t nc, uint32_t nb, uint32_t c0, const fl_chunk* dch,
                 const uint32_t* dbc, const fl_sblock* dsb, const uint8_t* d_in, uint8_t* d_out, uint64_t* d_outlen,
                 int32_t* d_status) {
    hipStream_t st = h->stream;
    const int mode = prm.mode, container = prm.container;
    int rc;
    fl_block_plan* dpl = (fl_block_plan*)h->plans.p;
    uint32_t* dhist = (uint32_t*)h->hist.p;
    // The container's checksum reads the input and nothing else.  Levels 4-7: on a stream of its own beside k_lz_parse, which
    // lives in LDS (1 GiB gzip level 6: 30.1 -> 29.5 ms; the tokenizer pays 1.1 ms for a neighbour that takes 1.8) -- not
    // beside k_lz_chain (2.98 ms instead of 1.26 with the checksum next to it) and not beside k_lz_walk, whose gathers
    // wait for the same memory system (config #3: 9.95 -> 11.9 ms): there it runs first, on the compute stream.
    if (container != 0 && ((mode >= 4 && prm.chain >= FL_BULK_MIN_CHAIN) || (mode < 4 && h->knobs.simple_ck_inline))) {
        ProfScope ps(h, K_CHECKSUM);
        hipLaunchKernelGGL(k_checksum, dim3(nb), dim3(64), 0, st, d_in, dch, dbc, dsb, prm, h->crc, (uint32_t*)h->cks.p);
    }
    // simple modes: beside the histograms and the planner, which are short chains of latency (config #4: 0.15 of 1.2 ms in line)
    if (container != 0 && mode < 4 && !h->knobs.simple_ck_inline && (rc = launch_checksum_side(h, nb, d_in, dch, dbc, dsb, prm))) return rc;
    if (mode >= 4) {
        if ((rc = ensure_lz_workspace(h, nc, prm.chain))) return rc;
        if (prm.chain >= FL_BULK_MIN_CHAIN) {
            HIP_OK(h, hipMemsetAsync(h->marks.p, 0, (size_t)nc * FL_CHUNK_STRIDE / 8, st));  // k_lz_walk stores the anchors in (k_lz_chain clears its chunk's itself)
            // levels 8 and 9 (chains of 1024 / 4096 candidates): the reference's chain and two sparser ones in global
            // memory, the automaton over them (kernels_walk.h): a walk is 1.4 steps per byte instead of 14
            {
                ProfScope ps(h, K_LZ_LINKS);
                hipLaunchKernelGGL(k_lz_links<0>, dim3(nc), dim3(64 * FL_CHAIN_WAVES), 0, st, d_in, dch, (uint16_t*)h->links.p,
                                   (uint32_t*)h->cflag.p);
                hipLaunchKernelGGL(k_lz_links<1>, dim3(nc), dim3(64 * FL_CHAIN_WAVES), 0, st, d_in, dch, (uint16_t*)h->links.p,
                                   (uint32_t*)h->cflag.p);
                hipLaunchKernelGGL(k_lz_links<2>, dim3(nc), dim3(64 * FL_CHAIN_WAVES), 0, st, d_in, dch, (uint16_t*)h->links.p,
                                   (uint32_t*)h->cflag.p);
                hipLaunchKernelGGL(k_lz_links<3>, dim3(nc), dim3(64 * FL_CHAIN_WAVES), 0, st, d_in, dch, (uint16_t*)h->links.p,
                                   (uint32_t*)h->cflag.p);
            }
            {
                ProfScope ps(h, K_LZ_WALK);
                hipLaunchKernelGGL(k_lz_walk<false>, dim3(nc), dim3(WK_THREADS), 0, st, d_in, dch, prm, (const uint16_t*)h->links.p,
                                   (const uint32_t*)h->cflag.p, (uint32_t*)h->desc.p, (uint32_t*)h->marks.p, wk_stream{});
                // the chunks k_lz_links<0> found a run of one byte in (workgroups of the other kind return at once)
                hipLaunchKernelGGL(k_lz_walk<true>, dim3(nc), dim3(WK_THREADS), 0, st, d_in, dch, prm, (const uint16_t*)h->links.p,
                                   (const uint32_t*)h->cflag.p, (uint32_t*)h->desc.p, (uint32_t*)h->marks.p, wk_stream{});
            }
        } else {
            // levels 4..7: the reference's chain in LDS, the automaton per segment (kernels_parse.h)
            {
                ProfScope ps(h, K_LZ_CHAIN);
                hipLaunchKernelGGL(k_lz_chain<false>, dim3(nc), dim3(64 * FL_CHAIN_WAVES), 0, st, d_in, dch, (uint16_t*)h->S.p,
                                   (uint32_t*)h->cflag.p, (uint32_t*)h->marks.p, (const uint32_t*)nullptr);
            }
            if (container != 0 && (rc = launch_checksum_side(h, nb, d_in, dch, dbc, dsb, prm))) return rc;
            {
                ProfScope ps(h, K_LZ_PARSE);
                hipLaunchKernelGGL(k_lz_parse<false>, dim3(nc), dim3(PZ_THREADS), 0, st, d_in, dch, prm, (const uint16_t*)h->S.p,
                                   (const uint32_t*)h->cflag.p, (uint32_t*)h->desc.p, (uint32_t*)h->marks.p,
                                   (const fl_swin*)nullptr, (const fl_chunk*)nullptr, (const uint32_t*)nullptr,
                                   (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, 0u, (const uint32_t*)nullptr);
            }
        }
        {
            ProfScope ps(h, K_LZ_EMIT);
            hipLaunchKernelGGL(k_lz_emit, dim3(nc), dim3(FL_EMITZ_THREADS), 0, st, d_in, dch, prm,
                               (const uint32_t*)h->desc.p, (const uint32_t*)h->marks.p, (uint32_t*)h->tokens.p, dhist,
                               dpl, (uint32_t*)h->ntok.p);
        }
        h->dbg_pass_chunks = nc;
        h->dbg_first_chunk = c0;
        h->dbg_pos_off.resize(nc);
        h->dbg_pieces.clear();
        for (uint32_t i = 0; i < nc; i++) h->dbg_pos_off[i] = (uint64_t)i * FL_CHUNK_STRIDE;
    } else if (mode == 1) {
        ProfScope ps(h, K_BYTE_HIST);
        hipLaunchKernelGGL(k_byte_hist, dim3(nb), dim3(256), 0, st, d_in, dch, dbc, dsb, dhist);
    }
    return enqueue_back_end(h, prm, nc, nb, c0, dch, dbc, dsb, d_in, d_out, d_outlen, d_status);
}


// Long streams, few of them: cut each at block starts into spans and decode the spans at once, twice
// (kernels_inflate_par.h, "spans").  Streams that come out whole get their status / out_len / consumed here and
// chunks[i].skip = 1 (the kernels that follow leave them alone); everything else stays as it was.
// Returns the number of streams finished, or a negative FLATE_HIP_E_*.
#define FL_SPAN_MIN_BYTES (128u * 1024u)   // streams shorter than this are not cut (8-32 members of 1 MiB, 130-1000 KB each: 7.0-8.0 -> 5.1-6.2 ms)
#define FL_SPAN_BYTES (64u * 1024u)        // compressed bytes per span, at least
#define FL_SPAN_MAX 1024u                  // spans per call
#ifndef FL_SPAN_STREAMS
#define FL_SPAN_STREAMS 32u
#endif
int try_span_inflate(flate_hip_ctx* h, hipStream_t st, const uint8_t* d_in, std::vector<fl_chunk>& chunks, int container,
                     int flags, uint8_t* d_out, uint64_t* d_outlen, int32_t* d_status, uint64_t* d_consumed) {
    const uint32_t n_chunks = (uint32_t)chunks.size();
    const uint64_t min_bytes = h->knobs.span_min_bytes;  // FLATE_HIP_INFLATE_SPANS -- 0: never; else the minimum stream size in bytes
    if (!min_bytes || (flags & 1)) return 0;
    const bool dbg = h->knobs.span_debug;
    const auto t_begin = std::chrono::steady_clock::now();
    auto since = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count(); };
    // Worth it (every span is decoded twice) when the long streams of the batch are too few to fill the chip with
    // a workgroup each: at most FL_SPAN_STREAMS of them (-DFL_SPAN_STREAMS: tuning)
    std::vector<uint32_t> elig;
    uint32_t n_long = 0;
    for (uint32_t i = 0; i < n_chunks; i++) n_long += chunks[i].in_len >= 32768u ? 1u : 0u;
    if (h->n_cu == 0) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, h->device) != hipSuccess || v <= 0) v = 256;
        h->n_cu = (uint32_t)v;
    }
    // More long streams than that, but fewer than CUs (config #5: 128 members on 256 CUs): every long stream is cut
    // ONCE, where about f = 2 S / (CUs + S) of it lies before the cut, and both runs of the second spans go in one launch
    // with the first spans (TWIN): the CUs the first spans leave free decode the second spans twice in the time the
    // first spans take.  (Cut in two by the rule above, the second run would have half the chip to itself.)  A stream
    // without a place to cut at is one span, decoded in place by run A: as before, but in the same launch.  Config #5,
    // cut at 56 / 62 / 66.6 / 72 / 78 % of the compressed bytes: 11.7 / 7.9 / 6.3 / 6.9 / 7.1 ms -- second spans that
    // are too long cost more than first spans that are: 1.5 % are added to f.
    const int etw = h->knobs.span_twin;  // FLATE_HIP_SPAN_TWIN -- -1: unset; 0: never; 2..950: where to cut, in thousandths of the stream (tuning)
    const bool sym = !h->knobs.span_two_runs;  // one decode per span, in symbols (kernels_inflate_par.h)
    const bool twin = n_long > FL_SPAN_STREAMS && etw != 0 && (size_t)n_long * 20 <= (size_t)h->n_cu * 11;  // (40 / 64 / 96 / 160 / 200 one-MiB members: 7.9 / 7.9 / 13.0 / 13.0 / 13.0 ms a workgroup each, 6.1 / 6.1 / 9.1 / 13.3 / 13.4 this way)
    if (n_long > FL_SPAN_STREAMS && !twin) return 0;
    const uint64_t elig_bytes = twin ? std::min<uint64_t>(min_bytes, 32768u) : min_bytes;
    for (uint32_t i = 0; i < n_chunks; i++)
        if (chunks[i].in_len >= elig_bytes) elig.push_back(i);
    if (elig.empty() || elig.size() > 256) return 0;
    int rc;
    // ---- where spans may start
    std::vector<fl_scan_point> points;
    std::vector<uint32_t> pt_first(elig.size() + 1, 0);
    // A span is a workgroup, a workgroup has a CU to itself: as many spans as CUs (a few less: what else runs), or
    // a multiple of that for long inputs -- 313 spans on 256 CUs take as long as 499 (measured: 170 MiB of text as
    // 249 / 313 / 374 / 499 / 703 spans: 12.3 / 17.5 / 16.2 / 14.3 / 16.3 ms), and every span costs k_span_scan and
    // k_span_resolve a step.
    // What a span costs goes with the bytes it makes; a stream's share of the spans goes with the room its caller
    // gave it (exact for gzip members whose ISIZE was read; 32 bytes per compressed byte at most), a span has at
    // least FL_SPAN_BYTES / 4 compressed bytes.
    auto weight = [&](const fl_chunk& c) { return std::max<uint64_t>(c.in_len, std::min<uint64_t>(c.out_cap, 32ull * c.in_len)); };
    uint64_t elig_w = 0;
    for (uint32_t ci : elig) elig_w += weight(chunks[ci]);
    const uint64_t rounds = std::min<uint64_t>(4, std::max<uint64_t>(1, (elig_w + h->n_cu * 393216ull) / (h->n_cu * 786432ull)));
    const uint64_t want = std::min<uint64_t>({(uint64_t)FL_SPAN_MAX, rounds * h->n_cu - std::min<uint32_t>(8u, h->n_cu / 2), std::max<uint64_t>(2, elig_w / (3 * FL_SPAN_BYTES))});
    for (size_t k = 0; k < elig.size(); k++) {
        const fl_chunk& c = chunks[elig[k]];
        const uint64_t bits = (uint64_t)c.in_len * 8;
        const uint32_t P = twin ? 2u : (uint32_t)std::max<uint64_t>(2, std::min<uint64_t>(want * weight(c) / elig_w, c.in_len / (FL_SPAN_BYTES / 4)));
        for (uint32_t j = 1; j < P; j++) {
            fl_scan_point pt;
            // (two runs: f = 2 S / (CUs + S) of the stream before the cut, + 1.5 %; one decode in symbols, 1.25 x the cost of a decode in
            // bytes: f = 1.25 S / (CUs + S / 4))
            const uint64_t auto_f = sym ? std::min<uint64_t>(900, std::max<uint64_t>(500, 1250ull * elig.size() / (h->n_cu + elig.size() / 4) + 10))
                                        : std::min<uint64_t>(900, std::max<uint64_t>(500, 2000ull * elig.size() / (h->n_cu + elig.size()) + 15));
            pt.from_bit = twin ? bits / 1000 * (etw > 1 ? (uint64_t)std::min(950, etw) : auto_f) : bits / P * j;
            pt.limit_bit = j + 1 < P ? bits / P * (j + 1) : bits;
            pt.stream = elig[k];
            pt.pad = 0;
            points.push_back(pt);
        }
        pt_first[k + 1] = (uint32_t)points.size();
    }
    const uint32_t npts = (uint32_t)points.size();
    if ((rc = ensure(h, h->sp_points, sizeof(fl_scan_point) * npts))) { (void)hipGetLastError(); return 0; }  // (no room: the old way)
    if ((rc = ensure(h, h->sp_found, sizeof(uint64_t) * npts))) { (void)hipGetLastError(); return 0; }  // (no room: the old way)
    if (hipMemcpyAsync(h->sp_points.p, points.data(), sizeof(fl_scan_point) * npts, hipMemcpyHostToDevice, st) != hipSuccess) return -1;
    const fl_chunk* dch = (const fl_chunk*)h->chunks.p;
    {
        ProfScope ps(h, K_SPAN_SCAN);
        hipLaunchKernelGGL(k_span_scan, dim3(npts), dim3(FP_THREADS), 0, st, d_in, dch, flags,
                           (const fl_scan_point*)h->sp_points.p, (uint64_t*)h->sp_found.p);
    }
    std::vector<uint64_t> found(npts);
    if (hipMemcpyAsync(found.data(), h->sp_found.p, sizeof(uint64_t) * npts, hipMemcpyDeviceToHost, st) != hipSuccess) return -1;
    if (hipStreamSynchronize(st) != hipSuccess) return -1;
    if (dbg) fprintf(stderr, "[spans] %.3f ms: sync 1 done\n", since());
    // ---- the spans: the stream start, then every distinct position found
    std::vector<fl_span> spans;
    std::vector<uint64_t> cand;
    std::vector<uint32_t> cand_off(n_chunks + 1, 0), sp_first(elig.size() + 1, 0);
    {
        size_t k = 0;
        for (uint32_t i = 0; i < n_chunks; i++) {
            cand_off[i] = (uint32_t)cand.size();
            if (k < elig.size() && elig[k] == i) {
                fl_span s0;
                s0.start_bit = 0;
                s0.wp = 0;
                s0.stream = i;
                s0.first = 1;
                s0.prev = FP_NO_SPAN;
                s0.live = 0;
                spans.push_back(s0);
                uint64_t last = 0;
                for (uint32_t j = pt_first[k]; j < pt_first[k + 1]; j++) {
                    if (found[j] == ~0ull || found[j] <= last) continue;  // (ascending: the targets are)
                    last = found[j];
                    fl_span s1 = s0;
                    s1.start_bit = found[j];
                    s1.first = 0;
                    spans.push_back(s1);
                    cand.push_back(found[j]);
                }
                k++;
                sp_first[k] = (uint32_t)spans.size();
            }
        }
        cand_off[n_chunks] = (uint32_t)cand.size();
    }
    const uint32_t nsp = (uint32_t)spans.size();
    if (dbg) fprintf(stderr, "[spans] %zu streams, %u scan points, %u spans\n", elig.size(), npts, nsp);
    if (nsp == (uint32_t)elig.size()) return 0;  // nothing to cut
    if ((rc = ensure(h, h->sp_spans, sizeof(fl_span) * nsp))) { (void)hipGetLastError(); return 0; }  // (no room: the old way)
    if ((rc = ensure(h, h->sp_res, sizeof(fl_span_res) * 2 * nsp))) { (void)hipGetLastError(); return 0; }  // (no room: the old way)
    if ((rc = ensure(h, h->sp_cand, sizeof(uint64_t) * (cand.size() + 1)))) { (void)hipGetLastError(); return 0; }  // (no room: the old way)
    if ((rc = ensure(h, h->sp_candoff, sizeof(uint32_t) * (n_chunks + 1)))) { (void)hipGetLastError(); return 0; }  // (no room: the old way)
    if ((rc = ensure(h, h->sp_tails, (size_t)FP_TAIL * nsp))) { (void)hipGetLastError(); return 0; }  // (no room: the old way)
    if ((rc = ensure(h, h->sp_tails_b, (size_t)FP_TAIL * nsp))) { (void)hipGetLastError(); return 0; }  // (no room: the old way)
    if (hipMemcpyAsync(h->sp_spans.p, spans.data(), sizeof(fl_span) * nsp, hipMemcpyHostToDevice, st) != hipSuccess) return -1;
    if (!cand.empty() && hipMemcpyAsync(h->sp_cand.p, cand.data(), sizeof(uint64_t) * cand.size(), hipMemcpyHostToDevice, st) != hipSuccess) return -1;
    if (hipMemcpyAsync(h->sp_candoff.p, cand_off.data(), sizeof(uint32_t) * (n_chunks + 1), hipMemcpyHostToDevice, st) != hipSuccess) return -1;
    // ---- the pool run A writes to: what the streams can hold (at most 32 bytes for every compressed one: beyond
    // that a stream goes the old way), two pieces of slack per span
    fl_span_pool pool;
    {
        uint64_t bytes = 0;
        for (uint32_t ci : elig) bytes += std::min<uint64_t>(chunks[ci].out_cap, 32ull * chunks[ci].in_len);
        const uint64_t pieces = ((bytes >> FP_PIECE_LOG) + 2ull * nsp + 1) * ((twin || sym) ? 2 : 1);
        // (a pool the device cannot give -- callers who reserve the worst case for gigabytes of input: the old way.  The
        // pool is a second copy of the decoded output: it may take a quarter of what is free next to what it holds
        // already, never more than 16 GiB, so that an allocator that shares the device -- PyTorch's -- is not starved.)
        if (pieces * FP_PIECE > (16ull << 30)) return 0;
        if (pieces * FP_PIECE + 16 > h->sp_pool.cap) {
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) {
                (void)hipGetLastError();
                return 0;
            }
            if (pieces * FP_PIECE + 16 > h->sp_pool.cap + free_b / 4) return 0;
        }
        if (ensure(h, h->sp_pool, (size_t)(pieces * FP_PIECE + 16))) {
            (void)hipGetLastError();  // (not this call's failure)
            return 0;
        }
        if ((rc = ensure(h, h->sp_pooltab, sizeof(uint32_t) * (size_t)FP_MAX_PIECES * nsp * 2))) { (void)hipGetLastError(); return 0; }  // (no room: the old way)
        if ((rc = ensure(h, h->sp_poolctl, 16))) { (void)hipGetLastError(); return 0; }  // (no room: the old way)
        if (hipMemsetAsync(h->sp_poolctl.p, 0, 16, st) != hipSuccess) return -1;
        pool.base = (uint8_t*)h->sp_pool.p;
        pool.next = (uint32_t*)h->sp_poolctl.p;
        pool.tab = (uint32_t*)h->sp_pooltab.p;
        pool.pieces = (uint32_t)pieces;
        pool.pad = 0;
    }
    // ---- run A
    {
        ProfScope ps(h, K_INFLATE_SPAN);
        hipLaunchKernelGGL(k_inflate_span, dim3(twin && !sym ? 2 * nsp : nsp), dim3(FP_THREADS), 0, st, d_in, dch, container, flags, h->crc, d_out,
                           (const fl_span*)h->sp_spans.p, (fl_span_res*)h->sp_res.p, (const uint64_t*)h->sp_cand.p,
                           (const uint32_t*)h->sp_candoff.p, (uint8_t*)h->sp_tails.p, 0u, pool, twin && !sym ? nsp : 0u,
                           (uint8_t*)h->sp_tails_b.p, sym ? nsp : 0u);
    }
    const bool both = twin || sym;  // what run B makes is there after this launch
    std::vector<fl_span_res> r1(nsp), r2(nsp);
    if (hipMemcpyAsync(r1.data(), h->sp_res.p, sizeof(fl_span_res) * nsp, hipMemcpyDeviceToHost, st) != hipSuccess) return -1;
    if (twin && !sym && hipMemcpyAsync(r2.data(), (const fl_span_res*)h->sp_res.p + nsp, sizeof(fl_span_res) * nsp, hipMemcpyDeviceToHost, st) != hipSuccess) return -1;
    if (hipStreamSynchronize(st) != hipSuccess) return -1;
    if (sym) r2 = r1;  // (one decode: what it says holds for both planes)
    if (dbg) fprintf(stderr, "[spans] %.3f ms: sync 2 done\n", since());
    // ---- the chain of every stream
    struct StreamPlan {
        bool ok = false;
        std::vector<uint32_t> chain;
        uint64_t total = 0, end_bit = 0;
    };
    std::vector<StreamPlan> plans(elig.size());
    bool any = false;
    for (size_t k = 0; k < elig.size(); k++) {
        StreamPlan& pl = plans[k];
        const fl_chunk& c = chunks[elig[k]];
        uint32_t cur = sp_first[k];
        uint64_t acc = 0;
        bool ok = true;
        for (uint32_t guard = 0; guard <= nsp; guard++) {
            const fl_span_res& r = r1[cur];
            if (dbg && guard < 6) fprintf(stderr, "[spans] pass 1 span %u: start %llu status %u end %llu out %llu final %u\n", cur, (unsigned long long)spans[cur].start_bit, r.status, (unsigned long long)r.end_bit, (unsigned long long)r.out_len, r.final_seen);
            if (r.status != 0) { ok = false; break; }
            // (run B of a twin launch does not know where its span starts: a distance that reaches before the start of
            // the OUTPUT -- possible in the first 32 KiB only -- is the old path's to find)
            if (both && !spans[cur].first && r.uses_hist && acc < FP_TAIL) { ok = false; break; }
            spans[cur].live = 1;
            spans[cur].wp = acc;
            spans[cur].prev = pl.chain.empty() ? FP_NO_SPAN : pl.chain.back();
            pl.chain.push_back(cur);
            acc += r.out_len;
            if (acc > c.out_cap) { ok = false; break; }
            if (r.final_seen) { pl.end_bit = r.end_bit; break; }
            // the span that starts where this one stopped
            uint32_t lo = sp_first[k] + 1, hi = sp_first[k + 1];
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (spans[mid].start_bit < r.end_bit) lo = mid + 1; else hi = mid;
            }
            if (lo >= sp_first[k + 1] || spans[lo].start_bit != r.end_bit || lo <= cur) { ok = false; break; }
            cur = lo;
        }
        if (ok && (pl.chain.empty() || !r1[pl.chain.back()].final_seen)) ok = false;
        if (dbg) fprintf(stderr, "[spans] stream %u: chain of %zu spans, %llu bytes, ok=%d\n", elig[k], pl.chain.size(), (unsigned long long)acc, (int)ok);
        if (dbg)
            for (uint32_t j = sp_first[k]; j < sp_first[k + 1]; j++)
                fprintf(stderr, "[spans]    span %u: start %llu status %u (0: decoded; else why it gave up) end %llu out %llu final %u blocks / pieces %u\n", j,
                        (unsigned long long)spans[j].start_bit, r1[j].status, (unsigned long long)r1[j].end_bit, (unsigned long long)r1[j].out_len, r1[j].final_seen, r1[j].n_pieces);
        if (!ok)
            for (uint32_t j = sp_first[k]; j < sp_first[k + 1]; j++) spans[j].live = 0;
        pl.ok = ok;
        pl.total = acc;
        any = any || ok;
    }
    if (!any) return 0;
    // ---- run B with the other filling of the history (live spans, in place), the true tails, k_span_fix
    std::vector<uint32_t> chain_all, chain_off(1, 0);
    for (size_t k = 0; k < elig.size(); k++) {
        if (plans[k].ok) chain_all.insert(chain_all.end(), plans[k].chain.begin(), plans[k].chain.end());
        chain_off.push_back((uint32_t)chain_all.size());
    }
    // (streams without a single copy that reaches before its span -- huffman-only, store-only ones -- need neither)
    bool uses_hist = false;
    for (uint32_t si : chain_all) uses_hist = uses_hist || r1[si].uses_hist != 0;
    if (dbg) fprintf(stderr, "[spans] history needed: %d\n", (int)uses_hist);
    std::vector<fl_fix_item> items;
    std::vector<uint32_t> item_first(elig.size() + 1, 0);
    for (size_t k = 0; k < elig.size(); k++) {
        if (plans[k].ok) {
            for (uint32_t si : plans[k].chain) {
                const uint64_t n = r1[si].out_len;
                for (uint64_t o = 0; o < n; o += FP_FIX_ITEM) {
                    fl_fix_item it;
                    it.dst = chunks[elig[k]].out_off + spans[si].wp + o;
                    it.span = si;
                    it.local = (uint32_t)o;
                    it.len = (uint32_t)std::min<uint64_t>(FP_FIX_ITEM, n - o);
                    it.kind = spans[si].first ? 0u : uses_hist ? (both ? 3u : 2u) : 1u;
                    it.prev = spans[si].prev;
                    it.pad = both ? nsp : 0u;
                    items.push_back(it);
                }
                // (a workgroup of k_span_fix takes FP_FIX_WAVES items of ONE span: empty ones fill the last)
                while (items.size() % FP_FIX_WAVES) {
                    fl_fix_item it = items.back();
                    it.local = 0;
                    it.len = 0;
                    items.push_back(it);
                }
            }
        }
        item_first[k + 1] = (uint32_t)items.size();
    }
    const uint32_t n_items = (uint32_t)items.size();
    if ((rc = ensure(h, h->sp_chain, sizeof(uint32_t) * (chain_all.size() + 1)))) { (void)hipGetLastError(); return 0; }  // (no room: the old way)
    if ((rc = ensure(h, h->sp_chainoff, sizeof(uint32_t) * chain_off.size()))) { (void)hipGetLastError(); return 0; }  // (no room: the old way)
    if ((rc = ensure(h, h->sp_items, sizeof(fl_fix_item) * ((size_t)n_items + 1)))) { (void)hipGetLastError(); return 0; }  // (no room: the old way)
    if ((rc = ensure(h, h->sp_part, sizeof(uint32_t) * 2 * ((size_t)n_items + 1)))) { (void)hipGetLastError(); return 0; }  // (no room: the old way)
    if (hipMemcpyAsync(h->sp_spans.p, spans.data(), sizeof(fl_span) * nsp, hipMemcpyHostToDevice, st) != hipSuccess) return -1;
    if (hipMemcpyAsync(h->sp_chain.p, chain_all.data(), sizeof(uint32_t) * chain_all.size(), hipMemcpyHostToDevice, st) != hipSuccess) return -1;
    if (hipMemcpyAsync(h->sp_chainoff.p, chain_off.data(), sizeof(uint32_t) * chain_off.size(), hipMemcpyHostToDevice, st) != hipSuccess) return -1;
    // (symbols) every span's place in its chain, the longest chain, room for two sets of tails in symbols; no room: span after span
    std::vector<uint32_t> chain_pos;
    uint32_t longest = 0;
    bool rs_ok = false;
    if (sym && uses_hist) {
        for (size_t k = 0; k < elig.size(); k++)
            if (plans[k].ok) {
                for (uint32_t j = 0; j < plans[k].chain.size(); j++) chain_pos.push_back(j);
                longest = std::max<uint32_t>(longest, (uint32_t)plans[k].chain.size());
            }
        rs_ok = !chain_pos.empty() && ensure(h, h->sp_chainpos, sizeof(uint32_t) * chain_pos.size()) == 0 &&
                ensure(h, h->sp_rs, 2 * chain_pos.size() * (size_t)FP_TAIL * sizeof(uint16_t)) == 0;
        if (!rs_ok) (void)hipGetLastError();
        if (rs_ok && hipMemcpyAsync(h->sp_chainpos.p, chain_pos.data(), sizeof(uint32_t) * chain_pos.size(), hipMemcpyHostToDevice, st) != hipSuccess) return -1;
    }
    if (n_items && hipMemcpyAsync(h->sp_items.p, items.data(), sizeof(fl_fix_item) * n_items, hipMemcpyHostToDevice, st) != hipSuccess) return -1;
    {
        ProfScope ps(h, K_INFLATE_SPAN);
        if (uses_hist) {
            if (!both)
                hipLaunchKernelGGL(k_inflate_span, dim3(nsp), dim3(FP_THREADS), 0, st, d_in, dch, container, flags, h->crc, d_out,
                                   (const fl_span*)h->sp_spans.p, (fl_span_res*)h->sp_res.p, (const uint64_t*)h->sp_cand.p,
                                   (const uint32_t*)h->sp_candoff.p, (uint8_t*)h->sp_tails_b.p, 1u, pool, 0u, (uint8_t*)nullptr, 0u);
            if (sym && rs_ok) {
                // the true tails of all spans at once: log2(longest chain) steps (kernels_inflate_par.h, k_span_rs_*)
                const uint32_t ng = (uint32_t)chain_all.size();
                uint16_t* S0 = (uint16_t*)h->sp_rs.p;
                uint16_t* S1 = S0 + (size_t)ng * FP_TAIL;
                const dim3 grid(ng, FP_TAIL / (256 * 8));
                hipLaunchKernelGGL(k_span_rs_init, grid, dim3(256), 0, st, (const uint32_t*)h->sp_chain.p, (const uint32_t*)h->sp_chainpos.p,
                                   (const uint8_t*)h->sp_tails.p, (const uint8_t*)h->sp_tails_b.p, S0);
                for (uint32_t D = 1; D < longest; D *= 2) {
                    hipLaunchKernelGGL(k_span_rs_step, grid, dim3(256), 0, st, (const uint32_t*)h->sp_chainpos.p, (const uint16_t*)S0, S1, D);
                    std::swap(S0, S1);
                }
                hipLaunchKernelGGL(k_span_rs_out, grid, dim3(256), 0, st, (const uint32_t*)h->sp_chain.p, (const uint16_t*)S0, (uint8_t*)h->sp_tails.p);
            } else {
                hipLaunchKernelGGL(k_span_resolve, dim3((uint32_t)elig.size()), dim3(FP_THREADS), 0, st, (const uint32_t*)h->sp_chain.p,
                                   (const uint32_t*)h->sp_chainoff.p, (uint8_t*)h->sp_tails.p, (const uint8_t*)h->sp_tails_b.p);
            }
        }
        if (n_items)
            hipLaunchKernelGGL(k_span_fix, dim3(n_items / FP_FIX_WAVES), dim3(64 * FP_FIX_WAVES), 0, st, (const fl_fix_item*)h->sp_items.p, pool,
                               (const uint8_t*)h->sp_tails.p, d_out, container, h->crc, (uint32_t*)h->sp_part.p);
    }
    // the footers of the streams whose chain is whole (one small gather instead of a copy per stream)
    const uint32_t flen = container == 1 ? 8u : container == 2 ? 4u : 0u;
    const uint32_t nel = (uint32_t)elig.size();
    std::vector<uint64_t> foot_off(nel, ~0ull);
    std::vector<uint8_t> foot(8 * (size_t)nel, 0);
    if (flen) {
        for (size_t k = 0; k < elig.size(); k++) {
            const fl_chunk& c = chunks[elig[k]];
            const uint64_t fb = (plans[k].end_bit + 7) >> 3;
            if (plans[k].ok && fb + flen <= c.in_len) foot_off[k] = c.in_off + fb;
        }
        if ((rc = ensure(h, h->sp_footoff, sizeof(uint64_t) * nel))) { (void)hipGetLastError(); return 0; }  // (no room: the old way)
        if ((rc = ensure(h, h->sp_foot, 8 * (size_t)nel))) { (void)hipGetLastError(); return 0; }  // (no room: the old way)
        if (hipMemcpyAsync(h->sp_footoff.p, foot_off.data(), sizeof(uint64_t) * nel, hipMemcpyHostToDevice, st) != hipSuccess) return -1;
        hipLaunchKernelGGL(k_span_footers, dim3((nel + 63) / 64), dim3(64), 0, st, d_in, (const uint64_t*)h->sp_footoff.p, nel, flen,
                           (uint8_t*)h->sp_foot.p);
        if (hipMemcpyAsync(foot.data(), h->sp_foot.p, 8 * (size_t)nel, hipMemcpyDeviceToHost, st) != hipSuccess) return -1;
    }
    std::vector<uint32_t> part(2 * (size_t)n_items);
    if (uses_hist && !both && hipMemcpyAsync(r2.data(), h->sp_res.p, sizeof(fl_span_res) * nsp, hipMemcpyDeviceToHost, st) != hipSuccess) return -1;
    if (n_items && hipMemcpyAsync(part.data(), h->sp_part.p, sizeof(uint32_t) * 2 * n_items, hipMemcpyDeviceToHost, st) != hipSuccess) return -1;
    if (hipStreamSynchronize(st) != hipSuccess) return -1;
    if (dbg) fprintf(stderr, "[spans] %.3f ms: sync 3 done\n", since());
    // ---- run B as run A, the checksum, the footer
    const uint32_t pow_piece = fl_crc_xpow8n(h->crc.xpow8, FP_FIX_ITEM);
    int done = 0;
    std::vector<fl_span_fin> fin;
    for (size_t k = 0; k < elig.size(); k++) {
        StreamPlan& pl = plans[k];
        if (!pl.ok) continue;
        fl_chunk& c = chunks[elig[k]];
        bool ok = true;
        for (size_t j = 0; j < pl.chain.size() && ok && uses_hist; j++) {
            const uint32_t si = pl.chain[j];
            if (spans[si].first) continue;  // (run A was final)
            const fl_span_res &a = r1[si], &b = r2[si];
            if (b.status != 0 || b.out_len != a.out_len || b.end_bit != a.end_bit) ok = false;
            if (dbg && !ok) fprintf(stderr, "[spans] span %u (chain %zu): run B status %u len %llu/%llu end %llu/%llu\n", si, j, b.status, (unsigned long long)b.out_len, (unsigned long long)a.out_len, (unsigned long long)b.end_bit, (unsigned long long)a.end_bit);
        }
        uint32_t crc = 0;
        uint64_t adA = 0, adB = 0;  // Adler-32 with a = b = 0 start
        for (uint32_t q = item_first[k]; q < item_first[k + 1] && ok; q++) {
            const uint32_t pc = part[2 * (size_t)q], plen = part[2 * (size_t)q + 1];
            if (container == 1) {
                crc = fl_crc_mulmod(crc, plen == FP_FIX_ITEM ? pow_piece : fl_crc_xpow8n(h->crc.xpow8, plen)) ^ pc;
            } else if (container == 2) {
                adB = (adB + adA * plen + (pc >> 16)) % 65521u;
                adA = (adA + (pc & 0xffff)) % 65521u;
            }
        }
        const uint64_t fb = (pl.end_bit + 7) >> 3;
        if (ok && fb + flen > c.in_len) ok = false;  // truncated: the old way names it
        if (ok && flen) {
            const uint8_t* f = &foot[8 * k];
            if (container == 1) {
                const uint32_t fcrc = (uint32_t)f[0] | ((uint32_t)f[1] << 8) | ((uint32_t)f[2] << 16) | ((uint32_t)f[3] << 24);
                const uint32_t fsz = (uint32_t)f[4] | ((uint32_t)f[5] << 8) | ((uint32_t)f[6] << 16) | ((uint32_t)f[7] << 24);
                ok = fcrc == crc && fsz == (uint32_t)pl.total;
            } else {
                const uint32_t a = (uint32_t)((1 + adA) % 65521u);
                const uint32_t b = (uint32_t)((pl.total % 65521u + adB) % 65521u);
                const uint32_t fad = ((uint32_t)f[0] << 24) | ((uint32_t)f[1] << 16) | ((uint32_t)f[2] << 8) | (uint32_t)f[3];
                ok = fad == ((b << 16) | a);
            }
        }
        if (dbg) fprintf(stderr, "[spans] stream %u after k_span_fix: ok=%d crc %08x\n", elig[k], (int)ok, crc);
        if (!ok) continue;  // (the bytes written so far are written again by the kernels that follow)
        fl_span_fin fi;
        fi.total = pl.total;
        fi.used = fb + flen;
        fi.chunk = elig[k];
        fi.pad = 0;
        fin.push_back(fi);
        c.skip = 1;
        done++;
    }
    if (done) {
        if ((rc = ensure(h, h->sp_fin, sizeof(fl_span_fin) * fin.size()))) { (void)hipGetLastError(); return 0; }  // (no room: the old way)
        if (hipMemcpyAsync(h->sp_fin.p, fin.data(), sizeof(fl_span_fin) * fin.size(), hipMemcpyHostToDevice, st) != hipSuccess) return -1;
        hipLaunchKernelGGL(k_span_finish, dim3(((uint32_t)fin.size() + 63) / 64), dim3(64), 0, st, (const fl_span_fin*)h->sp_fin.p,
                           (uint32_t)fin.size(), d_status, d_outlen, d_consumed);
        if (hipStreamSynchronize(st) != hipSuccess) return -1;  // (the source is on this stack)
    }
    if (dbg) fprintf(stderr, "[spans] %.3f ms: return\n", since());
    return done;
}

}  // namespace

extern "C" {

const char* flate_hip_version(void) { return "flate_hip 0.1 (gfx950)"; }

const char* flate_hip_status_name(int s) {
    switch (s) {
        case 0: return "Ok";
        case 1: return "EndOfStream";
        case 2: return "BadGzipHeader";
        case 3: return "BadZlibHeader";
        case 4: return "WrongGzipChecksum";
        case 5: return "WrongGzipSize";
        case 6: return "WrongZlibChecksum";
        case 7: return "InvalidCode";
        case 8: return "OversubscribedHuffmanTree";
        case 9: return "IncompleteHuffmanTree";
        case 10: return "MissingEndOfBlockCode";
        case 11: return "InvalidMatch";
        case 12: return "InvalidBlockType";
        case 13: return "WrongStoredBlockNlen";
        case 14: return "InvalidDynamicBlockHeader";
        case 100: return "OutputTooSmall";
        case 101: return "ChunkTooLarge";
        case 102: return "ReferenceQ1Stream";
        default: return "Unknown";
    }
}

int flate_hip_create(int device, flate_hip_handle* out) {
    if (!out) return FLATE_HIP_E_INVALID_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev)
        return FLATE_HIP_E_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return FLATE_HIP_E_NO_DEVICE;
    flate_hip_ctx* h = new flate_hip_ctx();
    h->device = device;
    if (hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking) != hipSuccess) {
        delete h;
        return FLATE_HIP_E_NO_DEVICE;
    }
    h->stream = h->own_stream;
    init_crc_consts(h->crc);
    read_knobs(h->knobs, FL_SPAN_MIN_BYTES);
    *out = h;
    return FLATE_HIP_OK;
}

// Debug / test seam: read the FLATE_HIP_* tuning variables again (they are read once, when the handle is made).
int flate_hip_debug_reload_env(flate_hip_handle h) {
    if (!h) return FLATE_HIP_E_INVALID_ARG;
    read_knobs(h->knobs, FL_SPAN_MIN_BYTES);
    return FLATE_HIP_OK;
}

int flate_hip_destroy(flate_hip_handle h) {
    if (!h) return FLATE_HIP_E_INVALID_ARG;
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    fold_profile(h);
    if (h->pin_in) (void)hipHostFree(h->pin_in);
    if (h->pin_len) (void)hipHostFree(h->pin_len);
    if (h->pin_tab) (void)hipHostFree(h->pin_tab);
    if (h->pin_out) (void)hipHostFree(h->pin_out);
    for (DevBuf* b : {&h->sp_points, &h->sp_found, &h->sp_spans, &h->sp_res, &h->sp_cand, &h->sp_candoff, &h->sp_tails,
                      &h->sp_tails_b, &h->sp_chain, &h->sp_chainoff, &h->sp_pool, &h->sp_pooltab, &h->sp_poolctl, &h->sp_items,
                      &h->sp_part, &h->sp_footoff, &h->sp_foot, &h->sp_fin, &h->sp_chainpos, &h->sp_rs})
        if (b->p) (void)hipFree(b->p);
    for (DevBuf* b : {&h->chunks, &h->blk_chunk, &h->plans, &h->hist, &h->cks, &h->S, &h->NC, &h->rec, &h->desc, &h->marks,
                      &h->tokens, &h->ntok, &h->cflag, &h->links, &h->wchunks, &h->swins, &h->wexit, &h->shard_sz, &h->tiles, &h->segs, &h->pieces, &h->fpts, &h->zones, &h->nsorted, &h->jmp,
                      &h->exitmap, &h->entry, &h->segtok, &h->tokbase, &h->bound, &h->sgroups, &h->sgroup0, &h->gmap, &h->gentry,
                      &h->sblocks, &h->st_in, &h->st_out, &h->st_inoff, &h->st_outlen, &h->st_status,
                      &h->st_consumed, &h->st_pack, &h->st_packoff, &h->st_slot})
        if (b->p) (void)hipFree(b->p);
    for (hipEvent_t e : h->free_events) (void)hipEventDestroy(e);
    for (hipEvent_t e : h->xfer_events) (void)hipEventDestroy(e);
    if (h->c2_ev0) (void)hipEventDestroy(h->c2_ev0);
    if (h->c2_ev1) (void)hipEventDestroy(h->c2_ev1);
    if (h->s_c2) (void)hipStreamDestroy(h->s_c2);
    if (h->ms_ev0) (void)hipEventDestroy(h->ms_ev0);
    if (h->ms_ev1) (void)hipEventDestroy(h->ms_ev1);
    if (h->s_ms) (void)hipStreamDestroy(h->s_ms);
    if (h->ck_ev0) (void)hipEventDestroy(h->ck_ev0);
    if (h->ck_ev1) (void)hipEventDestroy(h->ck_ev1);
    if (h->s_ck) (void)hipStreamDestroy(h->s_ck);
    if (h->s_in) (void)hipStreamDestroy(h->s_in);
    if (h->s_out) (void)hipStreamDestroy(h->s_out);
    (void)hipStreamDestroy(h->own_stream);
    delete h;
    return FLATE_HIP_OK;
}

int flate_hip_set_stream(flate_hip_handle h, void* hip_stream) {
    if (!h) return FLATE_HIP_E_INVALID_ARG;
    hipStream_t next = hip_stream ? (hipStream_t)hip_stream : h->own_stream;
    if (next != h->stream) {
        // work enqueued on the old stream still uses the workspace: drain it before the next call
        // (which may grow, i.e. free, those buffers) runs on another stream
        (void)hipSetDevice(h->device);
        (void)hipStreamSynchronize(h->stream);
        fold_profile(h);
    }
    h->stream = next;
    return FLATE_HIP_OK;
}
int flate_hip_set_sync(flate_hip_handle h, int s) {
    if (!h) return FLATE_HIP_E_INVALID_ARG;
    h->sync = s != 0;
    return FLATE_HIP_OK;
}
const char* flate_hip_last_error(flate_hip_handle h) { return h ? h->last_error.c_str() : "null handle"; }

size_t flate_hip_compress_bound(size_t n, int container, int mode) {
    (void)mode;
    const size_t hdr = container == 1 ? 18 : (container == 2 ? 6 : 0);
    // stored blocks: 5 bytes per 65535 (+ a possibly empty trailing one); a Huffman block never
    // beats that bound by more than its header; slack for the Q1 seam.
    return n + (n / 32768 + 2) * 16 + hdr + 64 + n / 64;
}

int flate_hip_profile_enable(flate_hip_handle h, int enable) {
    if (!h) return FLATE_HIP_E_INVALID_ARG;
    h->prof = enable != 0;
    return FLATE_HIP_OK;
}
int flate_hip_profile_reset(flate_hip_handle h) {
    if (!h) return FLATE_HIP_E_INVALID_ARG;
    fold_profile(h);
    for (int i = 0; i < K_COUNT; i++) {
        h->prof_ms[i] = 0;
        h->prof_n[i] = 0;
    }
    return FLATE_HIP_OK;
}
int flate_hip_profile_read(flate_hip_handle h, const char** names, double* total_ms, uint64_t* launches, int cap) {
    if (!h) return FLATE_HIP_E_INVALID_ARG;
    fold_profile(h);
    int n = 0;
    for (int i = 0; i < K_COUNT && n < cap; i++) {
        if (!h->prof_n[i]) continue;
        names[n] = kKernelNames[i];
        total_ms[n] = h->prof_ms[i];
        launches[n] = h->prof_n[i];
        n++;
    }
    return n;
}

}  // extern "C"

namespace {
int compress_impl(flate_hip_handle h, const uint8_t* in, const uint64_t* in_off, uint32_t n_chunks, int container,
                  int mode, uint8_t* out, const uint64_t* out_off, uint64_t* out_len, int32_t* status, int memkind,
                  const FlushSpec* fs, flate_hip_plan* pl = nullptr) {
    // pl: the batch's layout is known (plan): no offsets are fetched; while the plan is being built
    // (!pl->ready) the per-pass tables are made and kept and nothing is launched, afterwards they are
    // used as they are and the call only enqueues work
    const bool planning = pl && !pl->ready;
    if (!h || (!pl && (!in_off || !out_off)) || (!planning && (!out_len || !status))) return FLATE_HIP_E_INVALID_ARG;
    if (container < 0 || container > 2) return FLATE_HIP_E_INVALID_ARG;
    fl_params prm{};
    if (!level_args(mode, prm)) return FLATE_HIP_E_INVALID_ARG;
    if (memkind != FLATE_HIP_MEM_HOST && memkind != FLATE_HIP_MEM_DEVICE) return FLATE_HIP_E_INVALID_ARG;
    if (n_chunks == 0) return FLATE_HIP_OK;
    if (hipSetDevice(h->device) != hipSuccess) return FLATE_HIP_E_NO_DEVICE;
    prm.container = container;
    prm.mode = mode;
    prm.flags = (h->flags & FLATE_HIP_DEFLATE_REPAIR_Q1) ? FL_PRM_REPAIR_Q1 : 0u;
    hipStream_t st = h->stream;

    std::vector<uint64_t> hin_, hout_;
    int rc = 0;
    if (!pl) {
        rc = fetch_offsets(h, in_off, n_chunks, memkind, hin_);
        if (rc) return rc;
        rc = fetch_offsets(h, out_off, n_chunks, memkind, hout_);
        if (rc) return rc;
    }
    const std::vector<uint64_t>& hin = pl ? pl->hin : hin_;
    const std::vector<uint64_t>& hout = pl ? pl->hout : hout_;
    const uint64_t in_lo = hin[0], in_hi = hin[n_chunks];
    const uint64_t out_lo = hout[0], out_hi = hout[n_chunks];

    // Pageable host buffers of some size (what every caller of the facade's compress(reader, writer) has): a
    // staged hipMemcpy each way moves about 10 GB/s and nothing overlaps.  Instead a few host threads copy the
    // input into a pinned mirror, the call runs on the mirrors (sub-batches, DMA copies on their own streams
    // beside the kernels: the pinned path below), and the threads copy the produced bytes out of the mirror.
    // (Pinned buffers are used where they lie: no host thread touches the data.  On a shared host both ways have their
    // bad minutes -- the direct one 16.6 instead of 10.1 ms in one process of five, the mirrors 17.6 instead of 11.3 when
    // the neighbours keep the memory system busy -- and the direct one needs no CPU.)
    if (memkind == FLATE_HIP_MEM_HOST && !fs && !pl && (in_hi - in_lo) >= (8ull << 20) && in && out && !h->in_mirror &&
        !is_pinned_host(in + in_lo) && !is_pinned_host(out + out_lo) && !h->knobs.no_pin_mirror) {
        auto grow = [&](void*& p, size_t& cap, size_t want) -> bool {
            if (want <= cap) return true;
            if (p) (void)hipHostFree(p);
            p = nullptr;
            cap = 0;
            const size_t sz = want + want / 8 + 4096;
            if (hipHostMalloc(&p, sz, hipHostMallocDefault) != hipSuccess) {
                (void)hipGetLastError();
                p = nullptr;
                return false;
            }
            cap = sz;
            return true;
        };
        if (grow(h->pin_in, h->pin_in_cap, (in_hi - in_lo) + 16) && grow(h->pin_out, h->pin_out_cap, (out_hi - out_lo) + 16)) {
            const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
            const unsigned nt = std::min(8u, hw);
            auto parallel = [&](uint64_t items, const std::function<void(uint64_t, uint64_t)>& fn) {
                std::vector<std::thread> th;
                const uint64_t per = (items + nt - 1) / nt;
                for (unsigned t = 1; t < nt; t++) {
                    const uint64_t a = std::min(items, t * per), b = std::min(items, a + per);
                    if (b > a) th.emplace_back(fn, a, b);
                }
                fn(0, std::min(items, per));
                for (auto& x : th) x.join();
            };
            const uint8_t* src = in + in_lo;
            uint8_t* pin = (uint8_t*)h->pin_in;
            const uint8_t* pout = (const uint8_t*)h->pin_out;
            // the mirrors take the place of the caller's buffers: same offsets.  A sub-batch's input is copied right before
            // its H2D copy is enqueued, its produced bytes leave the mirror as soon as they have landed.
            h->mirror_in = [&](uint32_t c0, uint32_t nc) {
                const uint64_t a = hin[c0] - in_lo, b = hin[c0 + nc] - in_lo;
                parallel(b - a, [&](uint64_t x, uint64_t y) { memcpy(pin + a + x, src + a + x, y - x); });
            };
            bool landed = false;
            h->mirror_out = [&](uint32_t c0, uint32_t nc) {
                landed = true;
                parallel(nc, [&](uint64_t x, uint64_t y) {
                    for (uint64_t i = c0 + x; i < c0 + y; i++) {
                        const uint64_t n = std::min<uint64_t>(out_len[i], hout[i + 1] - hout[i]);
                        if (n) memcpy(out + hout[i], pout + (hout[i] - out_lo), n);
                    }
                });
            };
            h->in_mirror = true;
            rc = compress_impl(h, pin - in_lo, in_off, n_chunks, container, mode, (uint8_t*)h->pin_out - out_lo,
                               out_off, out_len, status, memkind, nullptr, nullptr);
            h->in_mirror = false;
            auto take_out = h->mirror_out;
            h->mirror_in = nullptr;
            h->mirror_out = nullptr;
            if (rc) return rc;
            if (!landed) take_out(0, n_chunks);  // (the call did not take the overlapped path)
            // (large mirrors do not stay pinned with the handle)
            if (h->pin_in_cap > (1ull << 30)) {
                (void)hipHostFree(h->pin_in);
                h->pin_in = nullptr;
                h->pin_in_cap = 0;
            }
            if (h->pin_out_cap > (1ull << 30)) {
                (void)hipHostFree(h->pin_out);
                h->pin_out = nullptr;
                h->pin_out_cap = 0;
            }
            return FLATE_HIP_OK;
        }
    }

    // device views of the caller's buffers
    const uint8_t* d_in = in;
    uint8_t* d_out = out;
    uint64_t* d_outlen = out_len;
    int32_t* d_status = status;
    uint64_t in_shift = 0, out_shift = 0;  // subtracted from offsets when staging host buffers
    bool pin_in = false, pin_out = false;
    if (memkind == FLATE_HIP_MEM_HOST) {
        if ((rc = ensure(h, h->st_in, (in_hi - in_lo) + 16))) return rc;
        if ((rc = ensure(h, h->st_out, (out_hi - out_lo) + 16))) return rc;
        if ((rc = ensure(h, h->st_outlen, sizeof(uint64_t) * n_chunks))) return rc;
        if ((rc = ensure(h, h->st_status, sizeof(int32_t) * n_chunks))) return rc;
        // Pinned host buffers (hipHostMalloc / hipHostRegister, torch pin_memory): the copies are real DMA
        // and run on their own streams, a sub-batch at a time (below).  Pageable: one staged copy each way.
        pin_in = !fs && is_pinned_host(in + in_lo);
        pin_out = !fs && is_pinned_host(out + out_lo);
        if ((pin_in && !h->s_in && create_copy_stream(&h->s_in, true) != hipSuccess) ||
            (pin_out && !h->s_out && create_copy_stream(&h->s_out, false) != hipSuccess))
            return FLATE_HIP_E_ALLOC;
        if (in_hi > in_lo && !pin_in)
            HIP_OK(h, hipMemcpyAsync(h->st_in.p, in + in_lo, in_hi - in_lo, hipMemcpyHostToDevice, st));
        d_in = (const uint8_t*)h->st_in.p;
        d_out = (uint8_t*)h->st_out.p;
        d_outlen = (uint64_t*)h->st_outlen.p;
        d_status = (int32_t*)h->st_status.p;
        in_shift = in_lo;
        out_shift = out_lo;
    } else if (((uintptr_t)out & 3) != 0) {
        return FLATE_HIP_E_INVALID_ARG;
    }

    // chunk table + initial status
    std::vector<fl_chunk> chunks(n_chunks);
    for (uint32_t i = 0; i < n_chunks; i++) {
        fl_chunk& c = chunks[i];
        const uint64_t len = hin[i + 1] - hin[i];
        c.in_off = hin[i] - in_shift;
        c.out_off = hout[i] - out_shift;
        c.out_cap = hout[i + 1] - hout[i];
        c.skip = 0;
        // stream positions are 32-bit and the last match tile looks 64 KiB + 258 bytes ahead of its start
        if (len > (mode >= 4 ? 0xfff00000ull : 0xfffffff0ull)) return FLATE_HIP_E_INVALID_ARG;
        c.in_len = (uint32_t)len;
        c.pos_off = 0;
        c.piece0 = c.n_piece = c.flush_off = c.n_flush = c.zone_off = c.n_slides = 0;
        c.unfinished = (fs && !fs->finish) ? 1u : 0u;
        c.pad_ = 0;
    }
    if (!planning) {
        HIP_OK(h, hipMemsetAsync(d_outlen, 0, sizeof(uint64_t) * n_chunks, st));
        HIP_OK(h, hipMemsetAsync(d_status, 0, sizeof(int32_t) * n_chunks, st));
    }

    // the bit packer ORs into the output: clear the slots first -- on a stream of its own, behind what the caller's stream holds
    // so far; the back end's first kernel that writes the slots waits for it (enqueue_back_end).  The tokenizer / the histograms
    // and the planner do not touch the slots: 0.08 ms of config #4's 0.98 and 0.15 ms of the headline's 26.5 are hidden.
    struct MsGuard {  // (whatever way this call ends, the caller's stream is behind the clearing)
        flate_hip_ctx* h;
        ~MsGuard() {
            if (h->ms_pending) {
                (void)hipStreamWaitEvent(h->stream, h->ms_ev1, 0);
                h->ms_pending = false;
            }
        }
    } ms_guard{h};
    if (out_hi > out_lo && !planning) {
        if (!h->s_ms) {  // (all three or none)
            hipStream_t s = nullptr;
            hipEvent_t e0 = nullptr, e1 = nullptr;
            if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) == hipSuccess && hipEventCreateWithFlags(&e0, hipEventDisableTiming) == hipSuccess &&
                hipEventCreateWithFlags(&e1, hipEventDisableTiming) == hipSuccess) {
                h->s_ms = s;
                h->ms_ev0 = e0;
                h->ms_ev1 = e1;
            } else {
                if (e0) (void)hipEventDestroy(e0);
                if (e1) (void)hipEventDestroy(e1);
                if (s) (void)hipStreamDestroy(s);
                (void)hipGetLastError();
            }
        }
        if (h->s_ms && !h->knobs.memset_inline) {
            HIP_OK(h, hipEventRecord(h->ms_ev0, st));
            HIP_OK(h, hipStreamWaitEvent(h->s_ms, h->ms_ev0, 0));
            {
                ProfScope ps(h, K_MEMSET, h->s_ms);
                HIP_OK(h, hipMemsetAsync(d_out + (out_lo - out_shift), 0, out_hi - out_lo, h->s_ms));
            }
            HIP_OK(h, hipEventRecord(h->ms_ev1, h->s_ms));
            h->ms_pending = true;
        } else {
            ProfScope ps(h, K_MEMSET);
            HIP_OK(h, hipMemsetAsync(d_out + (out_lo - out_shift), 0, out_hi - out_lo, st));
        }
    }

    // A pass is a run of consecutive chunks of one kind: at levels 4..9 inputs of up to 65535 bytes
    // take the chunk path (kernels_lz.h), longer ones the whole-stream path (kernels_stream.h).
    const size_t pass_limit = (pin_in || pin_out) ? std::min(pass_chunk_limit(h), host_pass_chunk_limit(h)) : pass_chunk_limit(h);
    const uint64_t stream_pass_bytes = stream_pass_byte_limit(h);
    if (pin_out) HIP_OK(h, hipStreamSynchronize(h->s_out));  // (nothing of an earlier call may still read st_out)
    // Pinned output: the link is shared by both directions (57 GB/s one way, 28.6 each way at once: tools/pcie_probe.py),
    // so what must not cross it is the unused part of the slots.  A copy kernel on the output stream writes the
    // produced bytes of every slot straight into the caller's (device-visible) pinned memory: 52 GB/s for the bytes
    // that matter (tools/zero_copy_probe.py) instead of 57 GB/s for 2.5 x as many.  Bytes of a slot beyond out_len[i]
    // are then not touched at all.
    uint8_t* zc_out = nullptr;  // device view of out + out_lo
    std::vector<uint64_t> zc_slot;
    if (pin_out && out_hi > out_lo) {
        void* dp = nullptr;
        if (hipHostGetDevicePointer(&dp, out + out_lo, 0) == hipSuccess && dp) {
            zc_out = (uint8_t*)dp;
            zc_slot.resize((size_t)n_chunks + 1);
            for (uint32_t i = 0; i <= n_chunks; i++) zc_slot[i] = hout[i] - out_lo;
            if ((rc = ensure(h, h->st_slot, sizeof(uint64_t) * ((size_t)n_chunks + 1)))) return rc;
            HIP_OK(h, hipMemcpyAsync(h->st_slot.p, zc_slot.data(), sizeof(uint64_t) * ((size_t)n_chunks + 1), hipMemcpyHostToDevice, st));
        } else {
            (void)hipGetLastError();
        }
    }
    // Pinned path: the tables of a pass (chunk table, block -> chunk) are copies from pageable vectors, which the runtime
    // stages and moves with a blit kernel -- 0.3-0.4 ms each time, on the compute stream between two passes (rocprofv3
    // timeline, tools/e2e_timeline.py).  They go on the input stream instead, into a slice of their own per pass, ahead of
    // the pass's input; the compute stream waits for one event per pass.
    const bool pinned_passes = (pin_in || pin_out) && !pl;
    uint64_t blk_total = 0, blk_base = 0;
    if (pinned_passes) {
        if (!h->s_in && create_copy_stream(&h->s_in, true) != hipSuccess) return FLATE_HIP_E_ALLOC;
        for (uint32_t i = 0; i < n_chunks; i++)
            blk_total += mode >= 4 ? 2u : (uint64_t)(chunks[i].in_len / FL_BLOCK_BYTES + 1);
        if ((rc = ensure(h, h->chunks, sizeof(fl_chunk) * (size_t)n_chunks))) return rc;
        if ((rc = ensure(h, h->blk_chunk, sizeof(uint32_t) * (size_t)std::max<uint64_t>(blk_total, 1)))) return rc;
        HIP_OK(h, hipStreamSynchronize(h->s_in));  // (nothing of an earlier call may still write the tables)
        // the tables leave from pinned memory: [fl_chunk x n_chunks | uint32 x blk_total]
        const size_t tab_need = sizeof(fl_chunk) * (size_t)n_chunks + sizeof(uint32_t) * (size_t)std::max<uint64_t>(blk_total, 1);
        if (h->pin_tab_cap < tab_need) {
            if (h->pin_tab) (void)hipHostFree(h->pin_tab);
            h->pin_tab = nullptr;
            h->pin_tab_cap = 0;
            const size_t sz = tab_need + tab_need / 4 + 4096;
            if (hipHostMalloc(&h->pin_tab, sz, hipHostMallocDefault) != hipSuccess) {
                (void)hipGetLastError();
                h->pin_tab = nullptr;
                return FLATE_HIP_E_ALLOC;
            }
            h->pin_tab_cap = sz;
        }
    }
    uint32_t nc = 0;
    size_t pass_count = 0;
    std::deque<std::vector<uint32_t>> keep_blk;
    std::deque<std::vector<fl_sblock>> keep_sb;
    // On the pinned path the passes are enqueued without a host wait in between and the copies on the input stream read the
    // vectors above: whichever way this function is left (an error in the middle included), they are outlived by the copies.
    struct PassGuard {
        flate_hip_ctx* h;
        bool on;
        ~PassGuard() {
            if (on) {
                if (h->s_in) (void)hipStreamSynchronize(h->s_in);
                if (h->s_c2) (void)hipStreamSynchronize(h->s_c2);
                (void)hipStreamSynchronize(h->stream);
            }
            h->ck_pending = false;  // (an error between the checksum launch and the back end must not leave a stale wait)
        }
    } pass_guard{h, pinned_passes};
    // Round 6: TWO compute streams.  A sub-batch's six kernels each end in a tail that fills the chip less and less (1024 chunks are
    // four per CU: 2.15 ms a sub-batch where a quarter of the whole batch's 6.6 ms would be 1.65) and the next sub-batch's first
    // kernel waits behind the last one's tail.  The sub-batches alternate between the caller's stream and a second one -- every
    // per-pass buffer is a slice of a batch-wide workspace, so two passes never touch the same bytes -- and the kernels of sub-batch
    // k + 1 start as soon as its input is there and run into the tails of k's: 10.3 -> 10.0 ms per 256 MiB in one process, 11.7 ->
    // 10.1 in another (bench e2e_host).  Levels 4-7, every input on the chunk path.  What lost (profiles/r06_host_path.txt): the
    // tokenizers of all sub-batches in a row on one stream with the chains and the back ends on streams of higher priority beside
    // them, 11.0 ms -- k_lz_parse takes a CU's whole LDS, a single small workgroup on a CU keeps the next tokenizer workgroup off it,
    // and every kernel ran a quarter to a half longer than alone.
    // (planned device batches of more than one pass take the same way: flate_hip_compress_planned)
    const bool planned_run = pl && pl->ready;
    if (planned_run) {
        blk_total = 0;
        for (const flate_hip_plan::Pass& pp : pl->passes) blk_total += pp.nb;
    }
    bool two = ((pinned_passes && (size_t)n_chunks > pass_limit) || (planned_run && pl->passes.size() > 1)) && mode >= 4 && !fs && !h->knobs.one_stream;
    for (uint32_t i = 0; two && i < n_chunks; i++) two = chunks[i].in_len <= FLATE_HIP_MAX_LZ_CHUNK;
    if (two && !h->s_c2) {
        hipStream_t s2 = nullptr;
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (hipStreamCreateWithFlags(&s2, hipStreamNonBlocking) == hipSuccess && hipEventCreateWithFlags(&e0, hipEventDisableTiming) == hipSuccess &&
            hipEventCreateWithFlags(&e1, hipEventDisableTiming) == hipSuccess) {
            h->s_c2 = s2;
            h->c2_ev0 = e0;
            h->c2_ev1 = e1;
        } else {
            if (e0) (void)hipEventDestroy(e0);
            if (e1) (void)hipEventDestroy(e1);
            if (s2) (void)hipStreamDestroy(s2);
            (void)hipGetLastError();
            two = false;
        }
    }
    if (two) {
        // the batch-wide workspace (the passes' slices: c0 chunks / the blocks before the pass into every buffer)
        if ((rc = ensure_lz_workspace(h, n_chunks, prm.chain))) return rc;
        if ((rc = ensure(h, h->plans, sizeof(fl_block_plan) * (size_t)blk_total))) return rc;
        if ((rc = ensure(h, h->hist, sizeof(uint32_t) * 320 * (size_t)blk_total))) return rc;
        if ((rc = ensure(h, h->cks, sizeof(uint32_t) * 2 * (size_t)blk_total))) return rc;
        // the second stream behind what the caller's stream holds so far (the cleared lengths and statuses)
        HIP_OK(h, hipEventRecord(h->c2_ev0, st));
        HIP_OK(h, hipStreamWaitEvent(h->s_c2, h->c2_ev0, 0));
        if (h->ms_pending) HIP_OK(h, hipStreamWaitEvent(h->s_c2, h->ms_ev1, 0));  // ... and behind the clearing of the slots
    }
    struct WsShift {  // a pass's view of the batch-wide workspace: every buffer from its slice on (undone when the pass is enqueued)
        std::vector<std::pair<DevBuf*, size_t>> undo;
        void add(DevBuf& b, size_t bytes) {
            b.p = (uint8_t*)b.p + bytes;
            b.cap -= bytes;
            undo.emplace_back(&b, bytes);
        }
        ~WsShift() {
            for (auto& u : undo) {
                u.first->p = (uint8_t*)u.first->p - u.second;
                u.first->cap += u.second;
            }
        }
    };
    // a pass over its slice of every per-pass buffer (c0 chunks / b0 blocks into the batch-wide workspace), its kernels on `sq`
    auto enqueue_sliced = [&](hipStream_t sq, uint32_t nc_, uint32_t nb_, uint32_t c0_, size_t b0, const fl_chunk* dch_, const uint32_t* dbc_,
                              const fl_sblock* dsb_) -> int {
        WsShift ws;
        const size_t pos0 = (size_t)c0_ * FL_CHUNK_STRIDE;
        ws.add(h->plans, sizeof(fl_block_plan) * b0);
        ws.add(h->hist, sizeof(uint32_t) * 320 * b0);
        ws.add(h->cks, sizeof(uint32_t) * 2 * b0);
        if (prm.chain >= FL_BULK_MIN_CHAIN) ws.add(h->links, pos0 * 4 * sizeof(uint16_t)); else ws.add(h->S, pos0 * sizeof(uint16_t));
        ws.add(h->desc, pos0 * sizeof(uint32_t));
        ws.add(h->marks, pos0 / 8);
        ws.add(h->tokens, pos0 * sizeof(uint32_t));
        ws.add(h->ntok, sizeof(uint32_t) * c0_);
        ws.add(h->cflag, sizeof(uint32_t) * c0_);
        hipStream_t saved = h->stream;
        h->stream = sq;
        const int r = enqueue_pass(h, prm, nc_, nb_, c0_, dch_, dbc_, dsb_, d_in, d_out, d_outlen, d_status);
        h->stream = saved;
        return r;
    };
    std::vector<uint32_t> pass_c0;  // first chunk of every pass (mirror_out)
    const bool landing = pin_out && (bool)h->mirror_out;
    if (landing && h->pin_len_cap < sizeof(uint64_t) * (size_t)n_chunks) {
        if (h->pin_len) (void)hipHostFree(h->pin_len);
        h->pin_len = nullptr;
        h->pin_len_cap = 0;
        const size_t sz = sizeof(uint64_t) * ((size_t)n_chunks + n_chunks / 4 + 64);
        if (hipHostMalloc(&h->pin_len, sz, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            h->pin_len = nullptr;
            return FLATE_HIP_E_ALLOC;
        }
        h->pin_len_cap = sz;
    }
    for (uint32_t c0 = 0; c0 < n_chunks; c0 += nc) {
        const bool stream = mode >= 4 && (fs || chunks[c0].in_len > FLATE_HIP_MAX_LZ_CHUNK);
        uint64_t pass_bytes = 0;
        // (a sub-batch of the pinned path costs about 0.9 ms whatever it holds: a tail of less than half a sub-batch
        // goes with the one before it)
        size_t limit = pass_limit;
        if (pinned_passes && (size_t)(n_chunks - c0) < pass_limit + pass_limit / 2) limit = std::min(pass_chunk_limit(h), (size_t)(n_chunks - c0));
        // (the GPU idles until the first sub-batch has crossed the link: the first two are a quarter and a half)
        const bool ramp = !h->knobs.no_ramp;
        if (ramp && pinned_passes && n_chunks >= 3 * pass_limit && pass_count < 2) limit = std::min(pass_limit, std::max<size_t>(64, pass_limit >> (2 - pass_count)));  // (never above the configured bound: FLATE_HIP_MAX_PASS_CHUNKS)
        for (nc = 0; c0 + nc < n_chunks; nc++) {
            const fl_chunk& c = chunks[c0 + nc];
            if ((mode >= 4 && (fs || c.in_len > FLATE_HIP_MAX_LZ_CHUNK)) != stream) break;
            if (stream ? (nc > 0 && pass_bytes + c.in_len > stream_pass_bytes) : nc >= limit) break;
            pass_bytes += c.in_len;
        }
        if (pl && stream) return FLATE_HIP_E_UNSUPPORTED;  // (whole-stream passes build more tables per call)
        const size_t pass_index = pass_count++;
        if (h->mirror_in) h->mirror_in(c0, nc);
        pass_c0.push_back(c0);
        hipEvent_t ev_in = nullptr;
        const bool sliced = pinned_passes && !stream;  // this pass's tables: a slice of their own, filled on the input stream
        if (pin_in && !sliced) {  // this sub-batch's input: in flight while the previous sub-batch is computed
            const uint64_t a = hin[c0], b = hin[c0 + nc];
            if ((rc = xfer_event(h, 4 * pass_index, &ev_in))) return rc;
            if (b > a)
                HIP_OK(h, hipMemcpyAsync((uint8_t*)h->st_in.p + (a - in_lo), in + a, b - a, hipMemcpyHostToDevice, h->s_in));
            HIP_OK(h, hipEventRecord(ev_in, h->s_in));
        }
        if (pl && pl->ready) {
            // planned batch: the tables of this pass are on the device already
            const flate_hip_plan::Pass& pp = pl->passes[pass_index];
            const uint32_t nb = pp.nb;
            prm.n_chunks = nc;
            prm.n_blocks = nb;
            prm.stream = 0;
            if ((rc = ensure(h, h->plans, sizeof(fl_block_plan) * (size_t)nb))) return rc;
            if ((rc = ensure(h, h->hist, sizeof(uint32_t) * 320 * (size_t)nb))) return rc;
            if ((rc = ensure(h, h->cks, sizeof(uint32_t) * 2 * (size_t)nb))) return rc;
            if (two) {
                hipStream_t stq = (pass_index & 1u) ? h->s_c2 : st;
                if ((rc = enqueue_sliced(stq, nc, nb, c0, (size_t)blk_base, (const fl_chunk*)pp.chunks, (const uint32_t*)pp.blk_chunk, nullptr))) return rc;
                blk_base += nb;
            } else if ((rc = enqueue_pass(h, prm, nc, nb, c0, (const fl_chunk*)pp.chunks, (const uint32_t*)pp.blk_chunk, nullptr,
                                          d_in, d_out, d_outlen, d_status))) {
                return rc;
            }
            continue;
        }
        // block table of this pass (kept until the end of the call: on the pinned path the passes are enqueued without a
        // host wait in between -- 0.27 ms per pass, tools/e2e_probe.py -- and the copies below read these vectors)
        keep_blk.emplace_back();
        keep_sb.emplace_back();
        std::vector<uint32_t>& blk_chunk = keep_blk.back();
        std::vector<fl_sblock>& sblocks = keep_sb.back();  // huffman-only / store-only with flush points
        StreamTables tabs;
        uint32_t nb = 0;
        for (uint32_t i = 0; i < nc; i++) {
            fl_chunk& c = chunks[c0 + i];
            c.first_block = nb;
            if (stream) {
                add_stream_chunk(tabs, c, i, nb, fs);
            } else if (fs) {
                // SimpleCompressor (deflate.zig:449-529): the 65535-byte buffer goes out as soon as it is
                // full; flush / finish write what it holds then, an empty block if nothing (474-484)
                uint32_t prev = 0;
                auto piece = [&](uint32_t start, uint32_t end, bool last) {
                    uint32_t p = start;
                    for (; end - p >= FL_BLOCK_BYTES; p += FL_BLOCK_BYTES) sblocks.push_back(fl_sblock{p, FL_BLOCK_BYTES, 0u});
                    sblocks.push_back(fl_sblock{p, end - p, last ? 1u : 0u});
                    if (!last) sblocks.push_back(fl_sblock{0u, 0u, 2u});
                };
                for (uint32_t k = 0; k < fs->n; k++) {
                    piece(prev, (uint32_t)fs->pos[k], false);
                    prev = (uint32_t)fs->pos[k];
                }
                if (fs->finish) piece(prev, c.in_len, true);
                c.n_blocks = (uint32_t)sblocks.size();
                c.pos_off = 0;
            } else {
                c.n_blocks = mode >= 4 ? 2u : (uint32_t)(c.in_len / FL_BLOCK_BYTES + 1);  // deflate.zig:498-511, 480-484
                c.pos_off = (uint64_t)i * FL_CHUNK_STRIDE;
            }
            for (uint32_t k = 0; k < c.n_blocks; k++) blk_chunk.push_back(i);
            nb += c.n_blocks;
        }
        prm.n_chunks = nc;
        prm.n_blocks = nb;
        prm.stream = stream ? 1u : 0u;
        if (!sliced) {
            if ((rc = ensure(h, h->chunks, sizeof(fl_chunk) * nc))) return rc;
            if ((rc = ensure(h, h->blk_chunk, sizeof(uint32_t) * nb))) return rc;
        }
        if ((rc = ensure(h, h->plans, sizeof(fl_block_plan) * (size_t)nb))) return rc;
        if ((rc = ensure(h, h->hist, sizeof(uint32_t) * 320 * (size_t)nb))) return rc;
        if ((rc = ensure(h, h->cks, sizeof(uint32_t) * 2 * (size_t)nb))) return rc;
        fl_chunk* tab_chunks = (fl_chunk*)h->chunks.p + (sliced ? c0 : 0u);
        uint32_t* tab_blk = (uint32_t*)h->blk_chunk.p + (sliced ? blk_base : 0u);
        if (sliced) {
            if (blk_base + nb > blk_total) return FLATE_HIP_E_LAUNCH;
            blk_base += nb;
            // (through the pinned table buffer: a copy from a pageable vector is staged by the runtime, and the staged copy of
            // sub-batch k + 1 did not start before the way home of sub-batch k was over -- rocprofv3 timeline, round 5)
            fl_chunk* pt_chunks = (fl_chunk*)h->pin_tab + c0;
            uint32_t* pt_blk = (uint32_t*)((fl_chunk*)h->pin_tab + n_chunks) + (blk_base - nb);
            memcpy(pt_chunks, &chunks[c0], sizeof(fl_chunk) * nc);
            memcpy(pt_blk, blk_chunk.data(), sizeof(uint32_t) * nb);
            HIP_OK(h, hipMemcpyAsync(tab_chunks, pt_chunks, sizeof(fl_chunk) * nc, hipMemcpyHostToDevice, h->s_in));
            HIP_OK(h, hipMemcpyAsync(tab_blk, pt_blk, sizeof(uint32_t) * nb, hipMemcpyHostToDevice, h->s_in));
            const uint64_t a = hin[c0], b = hin[c0 + nc];
            if (pin_in && b > a)
                HIP_OK(h, hipMemcpyAsync((uint8_t*)h->st_in.p + (a - in_lo), in + a, b - a, hipMemcpyHostToDevice, h->s_in));
            if ((rc = xfer_event(h, 4 * pass_index, &ev_in))) return rc;
            HIP_OK(h, hipEventRecord(ev_in, h->s_in));
        } else {
            HIP_OK(h, hipMemcpyAsync(tab_chunks, &chunks[c0], sizeof(fl_chunk) * nc, hipMemcpyHostToDevice, st));
            HIP_OK(h, hipMemcpyAsync(tab_blk, blk_chunk.data(), sizeof(uint32_t) * nb, hipMemcpyHostToDevice, st));
        }
        const fl_sblock* dsb = nullptr;
        if (!sblocks.empty()) {
            if ((rc = ensure(h, h->sblocks, sizeof(fl_sblock) * sblocks.size()))) return rc;
            HIP_OK(h, hipMemcpyAsync(h->sblocks.p, sblocks.data(), sizeof(fl_sblock) * sblocks.size(), hipMemcpyHostToDevice, st));
            dsb = (const fl_sblock*)h->sblocks.p;
        }
        // the host vectors must outlive the async copies
        if (!(pin_in || pin_out) || stream || planning) HIP_OK(h, hipStreamSynchronize(st));
        const bool alt = two && (pass_index & 1u) != 0;  // every other sub-batch on the second compute stream
        hipStream_t stp = alt ? h->s_c2 : st;
        if (ev_in) HIP_OK(h, hipStreamWaitEvent(stp, ev_in, 0));
        if (planning) {
            // keep this pass's tables in the plan; size the workspace now so that planned calls never allocate
            flate_hip_plan::Pass pp;
            pp.nc = nc;
            pp.nb = nb;
            if (hipMalloc(&pp.chunks, sizeof(fl_chunk) * nc) != hipSuccess ||
                hipMalloc(&pp.blk_chunk, sizeof(uint32_t) * std::max(nb, 1u)) != hipSuccess) {
                if (pp.chunks) (void)hipFree(pp.chunks);
                if (pp.blk_chunk) (void)hipFree(pp.blk_chunk);
                return FLATE_HIP_E_ALLOC;
            }
            pl->passes.push_back(pp);  // the plan owns the tables from here on (flate_hip_plan_destroy frees them)
            HIP_OK(h, hipMemcpy(pp.chunks, h->chunks.p, sizeof(fl_chunk) * nc, hipMemcpyDeviceToDevice));
            HIP_OK(h, hipMemcpy(pp.blk_chunk, h->blk_chunk.p, sizeof(uint32_t) * nb, hipMemcpyDeviceToDevice));
            if (mode >= 4 && (rc = ensure_lz_workspace(h, nc, prm.chain))) return rc;
            continue;
        }

        const fl_chunk* dch = tab_chunks;
        const uint32_t* dbc = tab_blk;
        if (stream) {
            if (container != 0) {
                ProfScope ps(h, K_CHECKSUM);
                hipLaunchKernelGGL(k_checksum, dim3(nb), dim3(64), 0, st, d_in, dch, dbc, dsb, prm, h->crc,
                                   (uint32_t*)h->cks.p);
            }
            if ((rc = compress_stream_pass(h, d_in, prm, nc, nb, tabs, &chunks[c0]))) return rc;
            h->dbg_pass_chunks = nc;
            h->dbg_first_chunk = c0;
            h->dbg_pos_off.clear();
            h->dbg_chunks.assign(chunks.begin() + c0, chunks.begin() + c0 + nc);
            h->dbg_pieces = tabs.pieces;
            if ((rc = enqueue_back_end(h, prm, nc, nb, c0, dch, dbc, dsb, d_in, d_out, d_outlen, d_status))) return rc;
            if (pinned_passes) {
                // A whole-stream pass keeps its tables at the START of the table buffers (its block count is not known when
                // the slices are laid out), where the slices of the chunk passes lie: its kernels are only enqueued here, so
                // the input stream -- which fills the next chunk pass's slice -- waits for them first.
                hipEvent_t ev_tab;
                if ((rc = xfer_event(h, 4 * pass_index + 3, &ev_tab))) return rc;
                HIP_OK(h, hipEventRecord(ev_tab, st));
                HIP_OK(h, hipStreamWaitEvent(h->s_in, ev_tab, 0));
            }
        } else if (two) {
            if ((rc = enqueue_sliced(stp, nc, nb, c0, (size_t)(blk_base - nb), dch, dbc, dsb))) return rc;
        } else {
            if ((rc = enqueue_pass(h, prm, nc, nb, c0, dch, dbc, dsb, d_in, d_out, d_outlen, d_status))) return rc;
        }
        HIP_OK(h, hipGetLastError());
        if (pin_out) {  // this sub-batch's output slots go home while the next sub-batch is computed
            hipEvent_t ev_out;
            if ((rc = xfer_event(h, 4 * pass_index + 1, &ev_out))) return rc;
            HIP_OK(h, hipEventRecord(ev_out, stp));
            HIP_OK(h, hipStreamWaitEvent(h->s_out, ev_out, 0));
            const uint64_t a = hout[c0], b = hout[c0 + nc];
            bool len_by_kernel = false;
            if (b > a && zc_out) {
                // Slots of one size (the usual case: compress_bound of one chunk size): the first half of every slot goes
                // home by the DMA engine's rectangle copy -- what lies beyond out_len[i] there is the zeros the slots were
                // cleared with --, and only what a chunk produced BEYOND that half by the copy kernel: a few workgroups that
                // take the slots in turn.  The kernel is not free beside the next sub-batch's kernels: its stores wait for the
                // link and the stores of the other kernels wait behind them (k_lz_chain took 0.47 ms instead of 0.09 beside
                // it, started behind it the tokenizer paid the same); the DMA engine costs them nothing (11.7 -> 10.5 ms for
                // 256 MiB of text, where no chunk reaches the second half).
                uint32_t urows = 0;
                uint64_t pitch = 0, half = 0;
                if (nc > 1) {
                    pitch = hout[c0 + 1] - hout[c0];
                    for (urows = 1; urows < nc && hout[c0 + urows + 1] - hout[c0 + urows] == pitch; urows++) {}
                    half = (pitch / 2) & ~(uint64_t)15;
                    // (Round 5: OFF unless FLATE_HIP_RECT=1.  The rectangle copy is a DMA copy that waits for the sub-batch's kernels
                    // -- and the input copy of the NEXT sub-batch, submitted behind it, lands in the same in-order DMA queue in most
                    // calls of a process: then nothing overlaps any more, 16.6 ms per 256 MiB instead of 10.4 (rocprofv3 timelines,
                    // profiles/r05_host_path.txt: the 10.4 only ever showed in the second call of a process).  The copy kernel
                    // alone is 11.4 ms in every call.)
                    if (urows < 64 || half < 4096 || h->knobs.rect != 1) urows = 0;
                }
                if (urows)
                    HIP_OK(h, hipMemcpy2DAsync(out + a, pitch, d_out + (a - out_shift), pitch, half, urows, hipMemcpyDeviceToHost, h->s_out));
                uint64_t* len_dev = nullptr;  // device view of pin_len + c0 (the mirror path reads the lengths as the passes land)
                if (landing) {
                    void* dp = nullptr;
                    if (hipHostGetDevicePointer(&dp, (uint64_t*)h->pin_len + c0, 0) == hipSuccess && dp) len_dev = (uint64_t*)dp;
                    else (void)hipGetLastError();
                }
                len_by_kernel = len_dev != nullptr;
                hipLaunchKernelGGL(k_copy_slots, dim3(std::min(64u, nc)), dim3(256), 0, h->s_out, d_out,
                                   (const uint64_t*)h->st_slot.p + c0, (const uint64_t*)d_outlen + c0, zc_out,
                                   (const uint64_t*)h->st_slot.p + c0, nc, urows, half, len_dev);
                HIP_OK(h, hipGetLastError());
            } else if (b > a) {
                HIP_OK(h, hipMemcpyAsync(out + a, d_out + (a - out_shift), b - a, hipMemcpyDeviceToHost, h->s_out));
            }
            if (landing) {  // this pass's lengths and an event behind its way home
                hipEvent_t ev_done;
                if ((rc = xfer_event(h, 4 * pass_index + 2, &ev_done))) return rc;
                if (!len_by_kernel)
                    HIP_OK(h, hipMemcpyAsync((uint64_t*)h->pin_len + c0, d_outlen + c0, sizeof(uint64_t) * nc, hipMemcpyDeviceToHost, h->s_out));
                HIP_OK(h, hipEventRecord(ev_done, h->s_out));
            }
        }
    }
    if (two) {  // the caller's stream behind the second one
        HIP_OK(h, hipEventRecord(h->c2_ev1, h->s_c2));
        HIP_OK(h, hipStreamWaitEvent(st, h->c2_ev1, 0));
    }
    if (landing) {
        // everything is enqueued: take the passes' output out of the mirror as they land
        for (size_t k = 0; k < pass_c0.size(); k++) {
            const uint32_t a = pass_c0[k], b = k + 1 < pass_c0.size() ? pass_c0[k + 1] : n_chunks;
            hipEvent_t ev_done;
            if ((rc = xfer_event(h, 4 * k + 2, &ev_done))) return rc;
            HIP_OK(h, hipEventSynchronize(ev_done));
            memcpy(out_len + a, (const uint64_t*)h->pin_len + a, sizeof(uint64_t) * (b - a));
            h->mirror_out(a, b - a);
        }
    }

    if (memkind == FLATE_HIP_MEM_HOST) {
        HIP_OK(h, hipMemcpyAsync(out_len, d_outlen, sizeof(uint64_t) * n_chunks, hipMemcpyDeviceToHost, st));
        HIP_OK(h, hipMemcpyAsync(status, d_status, sizeof(int32_t) * n_chunks, hipMemcpyDeviceToHost, st));
        HIP_OK(h, hipStreamSynchronize(st));
        if (pin_out)
            HIP_OK(h, hipStreamSynchronize(h->s_out));
        else if ((rc = copy_out_host(h, d_out, d_outlen, n_chunks, hout, out_shift, out, out_len)))
            return rc;
    } else if (h->sync) {
        HIP_OK(h, hipStreamSynchronize(st));
    }
    return FLATE_HIP_OK;
}

}  // namespace

extern "C" {

int flate_hip_compress_batch(flate_hip_handle h, const uint8_t* in, const uint64_t* in_off, uint32_t n_chunks,
                             int container, int mode, uint8_t* out, const uint64_t* out_off, uint64_t* out_len,
                             int32_t* status, int memkind) {
    return compress_impl(h, in, in_off, n_chunks, container, mode, out, out_off, out_len, status, memkind, nullptr);
}

int flate_hip_compress_flush(flate_hip_handle h, const uint8_t* in, uint64_t n, const uint64_t* flush_pos,
                             uint32_t n_flush, int finish, int container, int mode, uint8_t* out, uint64_t out_cap,
                             uint64_t* out_len, int32_t* status, int memkind) {
    if (!h || !out_len || !status || (n_flush && !flush_pos)) return FLATE_HIP_E_INVALID_ARG;
    if (!(mode == 0 || mode == 1 || (mode >= 4 && mode <= 9))) return FLATE_HIP_E_INVALID_ARG;
    if (memkind != FLATE_HIP_MEM_HOST) return FLATE_HIP_E_UNSUPPORTED;
    if (n > 0xfff00000ull) return FLATE_HIP_E_INVALID_ARG;
    uint64_t prev = 0;
    for (uint32_t k = 0; k < n_flush; k++) {
        if (flush_pos[k] < prev || flush_pos[k] > n) return FLATE_HIP_E_INVALID_ARG;
        prev = flush_pos[k];
    }
    // without finish() the stream ends with the marker of the last flush: nothing may follow it
    if (!finish && (n_flush == 0 || flush_pos[n_flush - 1] != n)) return FLATE_HIP_E_INVALID_ARG;
    const uint64_t in_off[2] = {0, n}, out_off[2] = {0, out_cap};
    const FlushSpec fs{flush_pos, n_flush, finish != 0};
    return compress_impl(h, in, in_off, 1, container, mode, out, out_off, out_len, status, memkind, &fs);
}

int flate_hip_decompress_batch(flate_hip_handle h, const uint8_t* in, const uint64_t* in_off, uint32_t n_chunks,
                               int container, int flags, uint8_t* out, const uint64_t* out_off, uint64_t* out_len,
                               int32_t* status, uint64_t* consumed, int memkind) {
    if (!h || !in_off || !out_off || !out_len || !status) return FLATE_HIP_E_INVALID_ARG;
    if (container < 0 || container > 2) return FLATE_HIP_E_INVALID_ARG;
    if (memkind != FLATE_HIP_MEM_HOST && memkind != FLATE_HIP_MEM_DEVICE) return FLATE_HIP_E_INVALID_ARG;
    if (n_chunks == 0) return FLATE_HIP_OK;
    if (hipSetDevice(h->device) != hipSuccess) return FLATE_HIP_E_NO_DEVICE;
    hipStream_t st = h->stream;

    std::vector<uint64_t> hin, hout;
    int rc = fetch_offsets(h, in_off, n_chunks, memkind, hin);
    if (rc) return rc;
    rc = fetch_offsets(h, out_off, n_chunks, memkind, hout);
    if (rc) return rc;
    const uint64_t in_lo = hin[0], in_hi = hin[n_chunks];
    const uint64_t out_lo = hout[0], out_hi = hout[n_chunks];

    const uint8_t* d_in = in;
    uint8_t* d_out = out;
    uint64_t* d_outlen = out_len;
    int32_t* d_status = status;
    uint64_t* d_consumed = consumed;
    uint64_t in_shift = 0, out_shift = 0;
    bool pin_io = false;
    if (memkind == FLATE_HIP_MEM_HOST) {
        if ((rc = ensure(h, h->st_in, (in_hi - in_lo) + 16))) return rc;
        if ((rc = ensure(h, h->st_out, (out_hi - out_lo) + 16))) return rc;
        if ((rc = ensure(h, h->st_outlen, sizeof(uint64_t) * n_chunks))) return rc;
        if ((rc = ensure(h, h->st_status, sizeof(int32_t) * n_chunks))) return rc;
        if ((rc = ensure(h, h->st_consumed, sizeof(uint64_t) * n_chunks))) return rc;
        // Pinned host buffers whose output slots are not much larger than what goes in (a caller who knows
        // the sizes): sub-batches with the copies on their own streams, as in compress_impl.  Otherwise one
        // staged copy in, and only the produced bytes come back (copy_out_host).
        pin_io = is_pinned_host(in + in_lo) && is_pinned_host(out + out_lo) &&
                 (out_hi - out_lo) <= 8 * (in_hi - in_lo) + (1ull << 20) && n_chunks > 4 * host_pass_chunk_limit(h);
        if (pin_io && ((!h->s_in && create_copy_stream(&h->s_in, true) != hipSuccess) ||
                       (!h->s_out && create_copy_stream(&h->s_out, false) != hipSuccess)))
            return FLATE_HIP_E_ALLOC;
        if (in_hi > in_lo && !pin_io)
            HIP_OK(h, hipMemcpyAsync(h->st_in.p, in + in_lo, in_hi - in_lo, hipMemcpyHostToDevice, st));
        d_in = (const uint8_t*)h->st_in.p;
        d_out = (uint8_t*)h->st_out.p;
        d_outlen = (uint64_t*)h->st_outlen.p;
        d_status = (int32_t*)h->st_status.p;
        d_consumed = (uint64_t*)h->st_consumed.p;
        in_shift = in_lo;
        out_shift = out_lo;
    }
    std::vector<fl_chunk> chunks(n_chunks);
    for (uint32_t i = 0; i < n_chunks; i++) {
        fl_chunk& c = chunks[i];
        const uint64_t len = hin[i + 1] - hin[i];
        if (len > 0xfffffff0ull) return FLATE_HIP_E_INVALID_ARG;
        c.in_off = hin[i] - in_shift;
        c.out_off = hout[i] - out_shift;
        c.out_cap = hout[i + 1] - hout[i];
        c.in_len = (uint32_t)len;
        c.first_block = 0;
        c.n_blocks = 0;
        c.skip = 0;
    }
    if ((rc = ensure(h, h->chunks, sizeof(fl_chunk) * n_chunks))) return rc;
    HIP_OK(h, hipMemcpyAsync(h->chunks.p, chunks.data(), sizeof(fl_chunk) * n_chunks, hipMemcpyHostToDevice, st));
    HIP_OK(h, hipStreamSynchronize(st));
    // (what goes back to the caller beyond out_len[i] is zeros, never bytes of an earlier call; the copies of the
    // call before are done: it waited for them)
    if (memkind == FLATE_HIP_MEM_HOST && out_hi > out_lo) HIP_OK(h, hipMemsetAsync(h->st_out.p, 0, out_hi - out_lo, st));
    // A few long streams: each by many workgroups at once (spans); what comes out whole is skipped below.
    if (!pin_io) {
        const int done = try_span_inflate(h, st, d_in, chunks, container, flags, d_out, d_outlen, d_status, d_consumed);
        // (the pool holds a second copy of the decoded output: a large one does not stay with the handle)
        if (h->sp_pool.cap > (2ull << 30)) {
            (void)hipStreamSynchronize(st);
            (void)hipFree(h->sp_pool.p);
            h->sp_pool.p = nullptr;
            h->sp_pool.cap = 0;
        }
        if (done < 0) {
            h->last_error = std::string("inflate by spans: ") + hipGetErrorString(hipGetLastError());
            return FLATE_HIP_E_LAUNCH;
        }
        if (done > 0) {
            HIP_OK(h, hipMemcpyAsync(h->chunks.p, chunks.data(), sizeof(fl_chunk) * n_chunks, hipMemcpyHostToDevice, st));
            HIP_OK(h, hipStreamSynchronize(st));
        }
    }
    // Long streams of a batch that has few of them: a workgroup per stream (kernels_inflate_par.h).  It marks
    // what it does not finish (short streams, anything irregular) FL_PAR_REDO, and k_inflate takes those.
    bool use_par = false;
    uint32_t par_min_bytes = 0;
    {
        const uint32_t min_bytes = h->knobs.inflate_par >= 0 ? (uint32_t)h->knobs.inflate_par : 32768u;  // FLATE_HIP_INFLATE_PAR -- 0: never; else the minimum stream size in bytes
        uint32_t n_big = 0;
        for (uint32_t i = 0; i < n_chunks; i++)  // long input, or an output slot that says the output is long
            n_big += (chunks[i].in_len >= min_bytes || chunks[i].out_cap >= 16ull * min_bytes) ? 1u : 0u;
        use_par = min_bytes && n_big && n_big <= 2048u && !(flags & 1);
        par_min_bytes = min_bytes;
    }
    // few streams: the latency of one stream decides, give each the large LDS ring (3 per CU);
    // many streams: the small ring keeps 20 per CU in flight
    const bool large = h->knobs.inflate_ring >= 0 ? h->knobs.inflate_ring >= (int64_t)FL_INF_RING_LARGE : n_chunks <= 4u * 256u;  // (FLATE_HIP_INFLATE_RING; measured crossover)
    if (pin_io) HIP_OK(h, hipStreamSynchronize(h->s_out));  // (nothing of an earlier call may still read st_out)
    // (a wave per stream: a sub-batch must still fill the chip -- 20 streams per CU -- or the kernel's latency per
    // stream, not the copies, decides; measured: sub-batches of 1024 streams are slower than no overlap at all)
    // sub-batches of one size (4097 streams used to be 4096 + 1, and the launch for the one cost a stream's whole latency)
    uint32_t sub = n_chunks;
    if (pin_io) {
        // (a sub-batch has to fill the chip -- 20 streams per CU -- or the latency of a stream decides: two sub-batches of
        // 2049 streams take as long as one batch, five of 3277 are 33 GB/s against 24, tools/e2e_inflate_probe.py)
        const uint32_t target = 3u * (uint32_t)host_pass_chunk_limit(h);
        const uint32_t nsub = std::max(1u, n_chunks / target);
        sub = (n_chunks + nsub - 1) / nsub;
    }
    size_t pass_index = 0;
    if (pin_io) {
        // Every sub-batch's input copy is submitted BEFORE any output copy: an output copy waits for its sub-batch's kernels,
        // and an input copy submitted behind it can land in the same in-order DMA queue and wait with it -- then nothing
        // overlaps (round 5, profiles/r05_host_path.txt: what happened to the compress path's rectangle copy).
        size_t k = 0;
        for (uint32_t c0 = 0; c0 < n_chunks; c0 += sub, k++) {
            const uint32_t nc = std::min(sub, n_chunks - c0);
            hipEvent_t ev_in;
            if ((rc = xfer_event(h, 2 * k, &ev_in))) return rc;
            const uint64_t a = hin[c0], b = hin[c0 + nc];
            if (b > a)
                HIP_OK(h, hipMemcpyAsync((uint8_t*)h->st_in.p + (a - in_lo), in + a, b - a, hipMemcpyHostToDevice, h->s_in));
            HIP_OK(h, hipEventRecord(ev_in, h->s_in));
        }
    }
    for (uint32_t c0 = 0; c0 < n_chunks; c0 += sub, pass_index++) {
        const uint32_t nc = std::min(sub, n_chunks - c0);
        if (pin_io) {  // this sub-batch's input: it came in while the sub-batches before it were decoded
            hipEvent_t ev_in;
            if ((rc = xfer_event(h, 2 * pass_index, &ev_in))) return rc;
            HIP_OK(h, hipStreamWaitEvent(st, ev_in, 0));
        }
        const fl_chunk* dch = (const fl_chunk*)h->chunks.p + c0;
        // Long streams of a batch that has few of them: a workgroup per stream (kernels_inflate_par.h).  It marks
        // what it does not finish (short streams, anything irregular) FL_PAR_REDO, and k_inflate takes those.
        const int32_t* redo_only = nullptr;
        if (use_par) {
            ProfScope ps(h, K_INFLATE_PAR);
            hipLaunchKernelGGL(k_inflate_par, dim3(nc), dim3(FP_THREADS), 0, st, d_in, dch, container, flags, par_min_bytes,
                               h->crc, d_out, d_outlen + c0, d_status + c0, d_consumed ? d_consumed + c0 : nullptr);
            redo_only = d_status + c0;
        }
        {
            ProfScope ps(h, K_INFLATE);
            if (large)
                hipLaunchKernelGGL(k_inflate<FL_INF_RING_LARGE>, dim3(nc), dim3(64), 0, st, d_in, dch, container, flags,
                                   h->crc, d_out, d_outlen + c0, d_status + c0,
                                   d_consumed ? d_consumed + c0 : nullptr, redo_only);
            else
                hipLaunchKernelGGL(k_inflate<FL_INF_RING_SMALL>, dim3(nc), dim3(64), 0, st, d_in, dch, container, flags,
                                   h->crc, d_out, d_outlen + c0, d_status + c0,
                                   d_consumed ? d_consumed + c0 : nullptr, redo_only);
        }
        HIP_OK(h, hipGetLastError());
        if (pin_io) {  // this sub-batch's output slots go home while the next sub-batch is decoded
            hipEvent_t ev_out;
            if ((rc = xfer_event(h, 2 * pass_index + 1, &ev_out))) return rc;
            HIP_OK(h, hipEventRecord(ev_out, st));
            HIP_OK(h, hipStreamWaitEvent(h->s_out, ev_out, 0));
            const uint64_t a = hout[c0], b = hout[c0 + nc];
            if (b > a)
                HIP_OK(h, hipMemcpyAsync(out + a, d_out + (a - out_shift), b - a, hipMemcpyDeviceToHost, h->s_out));
        }
    }
    HIP_OK(h, hipGetLastError());
    if (memkind == FLATE_HIP_MEM_HOST) {
        HIP_OK(h, hipMemcpyAsync(out_len, d_outlen, sizeof(uint64_t) * n_chunks, hipMemcpyDeviceToHost, st));
        HIP_OK(h, hipMemcpyAsync(status, d_status, sizeof(int32_t) * n_chunks, hipMemcpyDeviceToHost, st));
        if (consumed)
            HIP_OK(h, hipMemcpyAsync(consumed, d_consumed, sizeof(uint64_t) * n_chunks, hipMemcpyDeviceToHost, st));
        HIP_OK(h, hipStreamSynchronize(st));
        if (pin_io)
            HIP_OK(h, hipStreamSynchronize(h->s_out));
        else if ((rc = copy_out_host(h, d_out, d_outlen, n_chunks, hout, out_shift, out, out_len)))
            return rc;
    } else if (h->sync) {
        HIP_OK(h, hipStreamSynchronize(st));
    }
    return FLATE_HIP_OK;
}

int flate_hip_gather_streams(flate_hip_handle h, const uint8_t* out, const uint64_t* out_off, const uint64_t* out_len,
                             uint32_t n_chunks, uint8_t* dst, uint64_t* dst_off) {
    if (!h || !out || !out_off || !out_len || !dst || !dst_off) return FLATE_HIP_E_INVALID_ARG;
    if (hipSetDevice(h->device) != hipSuccess) return FLATE_HIP_E_NO_DEVICE;
    hipStream_t st = h->stream;
    {
        ProfScope ps(h, K_GATHER);
        hipLaunchKernelGGL(k_scan_lens, dim3(1), dim3(1024), 0, st, out_len, n_chunks, dst_off);
        if (n_chunks)
            hipLaunchKernelGGL(k_gather_copy, gather_grid(n_chunks), dim3(256), 0, st, out, out_off, out_len, dst,
                               (const uint64_t*)dst_off);
    }
    HIP_OK(h, hipGetLastError());
    if (h->sync) HIP_OK(h, hipStreamSynchronize(st));
    return FLATE_HIP_OK;
}

int flate_hip_set_flags(flate_hip_handle h, uint32_t flags) {
    if (!h || (flags & ~(uint32_t)FLATE_HIP_DEFLATE_REPAIR_Q1)) return FLATE_HIP_E_INVALID_ARG;
    h->flags = flags;
    return FLATE_HIP_OK;
}

int flate_hip_debug_phase_cycles(flate_hip_handle h, uint64_t* out, int n) {
    if (!h || !out || n <= 0) return FLATE_HIP_E_INVALID_ARG;
    if (hipSetDevice(h->device) != hipSuccess) return FLATE_HIP_E_NO_DEVICE;
    if (hipStreamSynchronize(h->stream) != hipSuccess) return FLATE_HIP_E_LAUNCH;
    uint64_t tmp[FL_PROF_SLOTS];
    if (hipMemcpyFromSymbol(tmp, HIP_SYMBOL(g_fl_prof), sizeof tmp) != hipSuccess) return FLATE_HIP_E_LAUNCH;
    for (int i = 0; i < n && i < FL_PROF_SLOTS; i++) out[i] = tmp[i];
    return FLATE_HIP_OK;
}

int64_t flate_hip_debug_tokens(flate_hip_handle h, uint32_t chunk, uint32_t* tokens, uint64_t cap) {
    if (!h || !tokens) return FLATE_HIP_E_INVALID_ARG;
    if (chunk < h->dbg_first_chunk || chunk >= h->dbg_first_chunk + h->dbg_pass_chunks) return FLATE_HIP_E_INVALID_ARG;
    if (hipSetDevice(h->device) != hipSuccess) return FLATE_HIP_E_NO_DEVICE;
    const uint32_t local = chunk - h->dbg_first_chunk;
    if (hipStreamSynchronize(h->stream) != hipSuccess) return FLATE_HIP_E_LAUNCH;
    if (!h->dbg_pieces.empty()) {
        // whole-stream pass: the token lists of the chunk's pieces, one after the other
        const fl_chunk& c = h->dbg_chunks[local];
        uint64_t total = 0;
        for (uint32_t i = 0; i < c.n_piece; i++) {
            const fl_piece& pc = h->dbg_pieces[c.piece0 + i];
            uint32_t m = 0;
            if (hipMemcpy(&m, (uint32_t*)h->ntok.p + c.piece0 + i, sizeof m, hipMemcpyDeviceToHost) != hipSuccess)
                return FLATE_HIP_E_LAUNCH;
            if (total + m <= cap && m &&
                hipMemcpy(tokens + total, (uint32_t*)h->tokens.p + c.pos_off + pc.start, (size_t)m * sizeof(uint32_t),
                          hipMemcpyDeviceToHost) != hipSuccess)
                return FLATE_HIP_E_LAUNCH;
            total += m;
        }
        return (int64_t)total;
    }
    uint32_t n = 0;
    if (hipMemcpy(&n, (uint32_t*)h->ntok.p + local, sizeof n, hipMemcpyDeviceToHost) != hipSuccess)
        return FLATE_HIP_E_LAUNCH;
    const uint64_t k = std::min<uint64_t>(n, cap);
    if (k && hipMemcpy(tokens, (uint32_t*)h->tokens.p + h->dbg_pos_off[local], k * sizeof(uint32_t),
                       hipMemcpyDeviceToHost) != hipSuccess)
        return FLATE_HIP_E_LAUNCH;
    return (int64_t)n;
}

int flate_hip_debug_write_block(flate_hip_handle h, const uint32_t* tokens, uint32_t n_tokens, const uint8_t* input,
                                uint32_t input_len, int eof, int dynamic_only, uint8_t* out, uint64_t out_cap,
                                uint64_t* out_len) {
    if (!h || !out || !out_len || (n_tokens && !tokens) || n_tokens > FL_MAX_TOKENS) return FLATE_HIP_E_INVALID_ARG;
    const bool has_input = input != nullptr;
    if (has_input && input_len > 0xfffffff0u) return FLATE_HIP_E_INVALID_ARG;
    if (hipSetDevice(h->device) != hipSuccess) return FLATE_HIP_E_NO_DEVICE;
    hipStream_t st = h->stream;
    int rc;
    const uint32_t in_len = has_input ? input_len : 0u;
    const uint64_t cap4 = (out_cap + 3) & ~3ull;
    if ((rc = ensure(h, h->chunks, sizeof(fl_chunk)))) return rc;
    if ((rc = ensure(h, h->blk_chunk, sizeof(uint32_t)))) return rc;
    if ((rc = ensure(h, h->plans, sizeof(fl_block_plan)))) return rc;
    if ((rc = ensure(h, h->hist, sizeof(uint32_t) * 320))) return rc;
    if ((rc = ensure(h, h->cks, sizeof(uint32_t) * 2))) return rc;
    if ((rc = ensure(h, h->tokens, sizeof(uint32_t) * ((size_t)n_tokens + 1)))) return rc;
    if ((rc = ensure(h, h->st_in, (size_t)in_len + 16))) return rc;
    if ((rc = ensure(h, h->st_out, cap4 + 16))) return rc;
    if ((rc = ensure(h, h->st_outlen, sizeof(uint64_t)))) return rc;
    if ((rc = ensure(h, h->st_status, sizeof(int32_t)))) return rc;
    fl_chunk ck{};
    ck.in_off = 0;
    ck.out_off = 0;
    ck.out_cap = out_cap;
    ck.in_len = in_len;
    ck.n_blocks = 1;
    fl_block_plan plan{};
    plan.valid = 1;
    plan.tok_count = n_tokens;
    plan.in_len = has_input ? in_len : FL_NO_INPUT;  // Zig null: the block cannot be stored
    plan.final_block = eof ? 1u : 0u;
    const uint32_t zero = 0;
    fl_params prm{};
    level_args(6, prm);
    prm.n_chunks = 1;
    prm.n_blocks = 1;
    prm.container = 0;
    prm.mode = 6;
    prm.stream = 1;  // plain block numbering in k_plan
    prm.plan_dynamic_only = dynamic_only ? 1u : 0u;
    HIP_OK(h, hipMemcpyAsync(h->chunks.p, &ck, sizeof ck, hipMemcpyHostToDevice, st));
    HIP_OK(h, hipMemcpyAsync(h->blk_chunk.p, &zero, sizeof zero, hipMemcpyHostToDevice, st));
    HIP_OK(h, hipMemcpyAsync(h->plans.p, &plan, sizeof plan, hipMemcpyHostToDevice, st));
    if (n_tokens)
        HIP_OK(h, hipMemcpyAsync(h->tokens.p, tokens, sizeof(uint32_t) * n_tokens, hipMemcpyHostToDevice, st));
    if (in_len) HIP_OK(h, hipMemcpyAsync(h->st_in.p, input, in_len, hipMemcpyHostToDevice, st));
    HIP_OK(h, hipMemsetAsync(h->st_out.p, 0, cap4 + 16, st));
    HIP_OK(h, hipStreamSynchronize(st));  // the host structs must outlive the async copies
    hipLaunchKernelGGL(k_dbg_token_hist, dim3(1), dim3(256), 0, st, (const uint32_t*)h->tokens.p, n_tokens,
                       (uint32_t*)h->hist.p);
    hipLaunchKernelGGL(k_plan, dim3(1), dim3(64 * FL_PLAN_WAVES), 0, st, (const fl_chunk*)h->chunks.p,
                       (const uint32_t*)h->blk_chunk.p, (const fl_sblock*)nullptr, prm, (const uint32_t*)h->hist.p,
                       (fl_block_plan*)h->plans.p);
    hipLaunchKernelGGL(k_offsets, dim3(1), dim3(64), 0, st, (const fl_chunk*)h->chunks.p, prm, h->crc,
                       (fl_block_plan*)h->plans.p, (const uint32_t*)h->cks.p, (uint8_t*)h->st_out.p,
                       (uint64_t*)h->st_outlen.p, (int32_t*)h->st_status.p);
    hipLaunchKernelGGL(k_encode<true>, dim3(1), dim3(64 * FL_ENC_WAVES), 0, st, (const uint8_t*)h->st_in.p,
                       (const fl_chunk*)h->chunks.p, (const uint32_t*)h->blk_chunk.p,
                       (const fl_block_plan*)h->plans.p, (const uint32_t*)h->tokens.p, (uint32_t*)h->st_out.p);
    HIP_OK(h, hipGetLastError());
    int32_t status = 0;
    HIP_OK(h, hipMemcpyAsync(out_len, h->st_outlen.p, sizeof(uint64_t), hipMemcpyDeviceToHost, st));
    HIP_OK(h, hipMemcpyAsync(&status, h->st_status.p, sizeof status, hipMemcpyDeviceToHost, st));
    HIP_OK(h, hipStreamSynchronize(st));
    if (status != 0) return FLATE_HIP_E_INVALID_ARG;  // out_cap too small
    if (*out_len) HIP_OK(h, hipMemcpy(out, h->st_out.p, *out_len, hipMemcpyDeviceToHost));
    return FLATE_HIP_OK;
}

// ---- multi-GPU: the local shard through the single-GPU path, then the reassembly over RCCL ----
// RCCL is not linked: the entry points bind to the RCCL the process already uses (the communicator
// the caller hands over was made by it), or load librccl.so when none is loaded yet.
namespace {
struct RcclApi {
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    int (*Send)(const void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    bool ok = false;
};
const RcclApi& rccl() {
    static RcclApi api = [] {
        RcclApi a;
        void* lib = RTLD_DEFAULT;
        if (!dlsym(lib, "ncclAllGather")) lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (lib) {
            a.AllGather = (decltype(a.AllGather))dlsym(lib, "ncclAllGather");
            a.Send = (decltype(a.Send))dlsym(lib, "ncclSend");
            a.Recv = (decltype(a.Recv))dlsym(lib, "ncclRecv");
            a.GroupStart = (decltype(a.GroupStart))dlsym(lib, "ncclGroupStart");
            a.GroupEnd = (decltype(a.GroupEnd))dlsym(lib, "ncclGroupEnd");
            a.ok = a.AllGather && a.Send && a.Recv && a.GroupStart && a.GroupEnd;
        }
        return a;
    }();
    return api;
}
enum { kNcclUint8 = 1, kNcclUint64 = 5 };  // ncclDataType_t (rccl.h)

// every rank's slice of `gathered` to every peer: one grouped batch of sends and receives, so that
// each of a GPU's xGMI links carries one peer's shard (no ring)
int exchange_slices(flate_hip_ctx* h, void* comm, int rank, int world, uint8_t* gathered, uint64_t slice_bytes,
                    uint64_t send_bytes) {
    const RcclApi& r = rccl();
    if (world == 1) return FLATE_HIP_OK;
    if (r.GroupStart() != 0) return FLATE_HIP_E_LAUNCH;
    for (int d = 1; d < world; d++) {
        const int to = (rank + d) % world, from = (rank - d + world) % world;
        if (r.Send(gathered + (uint64_t)rank * slice_bytes, send_bytes, kNcclUint8, to, comm, h->stream) != 0 ||
            r.Recv(gathered + (uint64_t)from * slice_bytes, send_bytes, kNcclUint8, from, comm, h->stream) != 0) {
            (void)r.GroupEnd();
            h->last_error = "ncclSend / ncclRecv failed";
            return FLATE_HIP_E_LAUNCH;
        }
    }
    if (r.GroupEnd() != 0) {
        h->last_error = "ncclGroupEnd failed";
        return FLATE_HIP_E_LAUNCH;
    }
    return FLATE_HIP_OK;
}
}  // namespace

int flate_hip_compress_batch_sharded(flate_hip_handle h, void* nccl_comm, int rank, int world, const uint8_t* in,
                                     const uint64_t* in_off, uint32_t n_chunks, int container, int mode, uint8_t* out,
                                     const uint64_t* out_off, uint64_t* out_len, int32_t* status, uint8_t* gathered,
                                     uint64_t slice_bytes, uint64_t* sizes, uint64_t* dst_off) {
    if (!h || !nccl_comm || world <= 0 || rank < 0 || rank >= world || !gathered || !sizes || !dst_off)
        return FLATE_HIP_E_INVALID_ARG;
    if (!rccl().ok) {
        h->last_error = "RCCL (librccl.so) not available";
        return FLATE_HIP_E_UNSUPPORTED;
    }
    const bool was_sync = h->sync;
    h->sync = false;  // everything below is ordered on the handle's stream
    int rc = compress_impl(h, in, in_off, n_chunks, container, mode, out, out_off, out_len, status,
                           FLATE_HIP_MEM_DEVICE, nullptr);
    // this rank's streams back to back at the head of its slice; dst_off[n_chunks] = packed size
    if (!rc) rc = flate_hip_gather_streams(h, out, out_off, out_len, n_chunks, gathered + (uint64_t)rank * slice_bytes, dst_off);
    if (!rc && rccl().AllGather(dst_off + n_chunks, sizes, 1, kNcclUint64, nccl_comm, h->stream) != 0) {
        h->last_error = "ncclAllGather failed";
        rc = FLATE_HIP_E_LAUNCH;
    }
    // every peer gets the largest packed shard's worth of bytes, not the slice's capacity (the sizes are read once:
    // one wait on the stream; a caller needs them anyway to use gathered[])
    uint64_t send_bytes = slice_bytes;
    if (!rc && world > 1) {
        std::vector<uint64_t> hs((size_t)world);
        if (hipMemcpyAsync(hs.data(), sizes, sizeof(uint64_t) * (size_t)world, hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
            hipStreamSynchronize(h->stream) != hipSuccess) {
            h->last_error = "reading the packed sizes failed";
            rc = FLATE_HIP_E_LAUNCH;
        } else {
            uint64_t mx = 0;
            for (uint64_t v : hs) mx = std::max(mx, v);
            if (mx > slice_bytes) {
                h->last_error = "a packed shard is larger than slice_bytes";
                rc = FLATE_HIP_E_INVALID_ARG;
            } else {
                send_bytes = std::min<uint64_t>(slice_bytes, (mx + 15) & ~(uint64_t)15);
            }
        }
    }
    if (!rc) rc = exchange_slices(h, nccl_comm, rank, world, gathered, slice_bytes, send_bytes);
    h->sync = was_sync;
    if (!rc && h->sync) HIP_OK(h, hipStreamSynchronize(h->stream));
    return rc;
}

int flate_hip_decompress_batch_sharded(flate_hip_handle h, void* nccl_comm, int rank, int world, const uint8_t* in,
                                       const uint64_t* in_off, uint32_t n_chunks, int container, int flags,
                                       uint8_t* gathered, uint64_t slice_bytes, const uint64_t* out_off,
                                       uint64_t* out_len, int32_t* status, uint64_t* consumed) {
    if (!h || !nccl_comm || world <= 0 || rank < 0 || rank >= world || !gathered) return FLATE_HIP_E_INVALID_ARG;
    if (!rccl().ok) {
        h->last_error = "RCCL (librccl.so) not available";
        return FLATE_HIP_E_UNSUPPORTED;
    }
    const bool was_sync = h->sync;
    h->sync = false;
    // the outputs of this rank's streams go straight into its slice (out_off is relative to the slice)
    int rc = flate_hip_decompress_batch(h, in, in_off, n_chunks, container, flags, gathered + (uint64_t)rank * slice_bytes,
                                        out_off, out_len, status, consumed, FLATE_HIP_MEM_DEVICE);
    // every peer gets what the fullest slice holds (the end of its last slot), not the slices' capacity: the ends are
    // all-gathered and read once (one wait on the stream; decompress has waited for its offsets already)
    uint64_t send_bytes = slice_bytes;
    if (!rc && world > 1) {
        std::vector<uint64_t> hs((size_t)world + 1, 0);
        if ((rc = ensure(h, h->shard_sz, sizeof(uint64_t) * ((size_t)world + 1))) == 0) {
            uint64_t* dsz = (uint64_t*)h->shard_sz.p;
            if (hipMemcpyAsync(dsz + world, out_off + n_chunks, sizeof(uint64_t), hipMemcpyDeviceToDevice, h->stream) != hipSuccess ||
                rccl().AllGather(dsz + world, dsz, 1, kNcclUint64, nccl_comm, h->stream) != 0 ||
                hipMemcpyAsync(hs.data(), dsz, sizeof(uint64_t) * (size_t)world, hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
                hipStreamSynchronize(h->stream) != hipSuccess) {
                h->last_error = "exchanging the slice sizes failed";
                rc = FLATE_HIP_E_LAUNCH;
            } else {
                uint64_t mx = 0;
                for (int i = 0; i < world; i++) mx = std::max(mx, hs[(size_t)i]);
                if (mx > slice_bytes) {
                    h->last_error = "a rank's slots end beyond slice_bytes";
                    rc = FLATE_HIP_E_INVALID_ARG;
                } else {
                    send_bytes = std::min<uint64_t>(slice_bytes, (mx + 15) & ~(uint64_t)15);
                }
            }
        }
    }
    if (!rc) rc = exchange_slices(h, nccl_comm, rank, world, gathered, slice_bytes, send_bytes);
    h->sync = was_sync;
    if (!rc && h->sync) HIP_OK(h, hipStreamSynchronize(h->stream));
    return rc;
}

int flate_hip_plan_compress(flate_hip_handle h, const uint64_t* in_off, const uint64_t* out_off, uint32_t n_chunks,
                            int container, int mode, flate_hip_plan_t* plan) {
    if (!h || !in_off || !out_off || !plan || n_chunks == 0) return FLATE_HIP_E_INVALID_ARG;
    *plan = nullptr;
    flate_hip_plan* pl = new flate_hip_plan();
    pl->hin.assign(in_off, in_off + n_chunks + 1);
    pl->hout.assign(out_off, out_off + n_chunks + 1);
    pl->n_chunks = n_chunks;
    pl->container = container;
    pl->mode = mode;
    for (uint32_t i = 0; i < n_chunks; i++)
        if (pl->hin[i + 1] < pl->hin[i] || pl->hout[i + 1] < pl->hout[i]) {
            delete pl;
            return FLATE_HIP_E_INVALID_ARG;
        }
    const int rc = compress_impl(h, nullptr, nullptr, n_chunks, container, mode, nullptr, nullptr, nullptr, nullptr,
                                 FLATE_HIP_MEM_DEVICE, nullptr, pl);
    if (rc) {
        (void)flate_hip_plan_destroy(h, pl);
        return rc;
    }
    pl->ready = true;
    *plan = pl;
    return FLATE_HIP_OK;
}

int flate_hip_compress_planned(flate_hip_handle h, flate_hip_plan_t plan, const uint8_t* in, uint8_t* out,
                               uint64_t* out_len, int32_t* status) {
    if (!h || !plan || !plan->ready || !in || !out) return FLATE_HIP_E_INVALID_ARG;
    return compress_impl(h, in, nullptr, plan->n_chunks, plan->container, plan->mode, out, nullptr, out_len, status,
                         FLATE_HIP_MEM_DEVICE, nullptr, plan);
}

int flate_hip_plan_destroy(flate_hip_handle h, flate_hip_plan_t plan) {
    if (!h || !plan) return FLATE_HIP_E_INVALID_ARG;
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    for (auto& pp : plan->passes) {
        if (pp.chunks) (void)hipFree(pp.chunks);
        if (pp.blk_chunk) (void)hipFree(pp.blk_chunk);
    }
    delete plan;
    return FLATE_HIP_OK;
}

int flate_hip_checksum(flate_hip_handle h, const uint8_t* data, uint64_t n, int container, uint32_t* value) {
    if (!h || !value || (n && !data) || (container != 1 && container != 2) || n > 0xfffffff0ull) return FLATE_HIP_E_INVALID_ARG;
    if (hipSetDevice(h->device) != hipSuccess) return FLATE_HIP_E_NO_DEVICE;
    hipStream_t st = h->stream;
    int rc;
    // one huffman-only style chunk: block j = bytes [65535 j, ...)
    fl_chunk ck{};
    ck.in_len = (uint32_t)n;
    ck.n_blocks = (uint32_t)(n / FL_BLOCK_BYTES + 1);
    const uint32_t nb = ck.n_blocks;
    std::vector<uint32_t> blk_chunk(nb, 0u);
    fl_params prm{};
    prm.n_chunks = 1;
    prm.n_blocks = nb;
    prm.container = container;
    prm.mode = 1;
    if ((rc = ensure(h, h->chunks, sizeof(fl_chunk)))) return rc;
    if ((rc = ensure(h, h->blk_chunk, sizeof(uint32_t) * nb))) return rc;
    if ((rc = ensure(h, h->cks, sizeof(uint32_t) * 2 * (size_t)nb + 16))) return rc;
    if ((rc = ensure(h, h->st_in, n + 16))) return rc;
    if ((rc = ensure(h, h->st_status, 16))) return rc;
    HIP_OK(h, hipMemcpyAsync(h->chunks.p, &ck, sizeof ck, hipMemcpyHostToDevice, st));
    HIP_OK(h, hipMemcpyAsync(h->blk_chunk.p, blk_chunk.data(), sizeof(uint32_t) * nb, hipMemcpyHostToDevice, st));
    if (n) HIP_OK(h, hipMemcpyAsync(h->st_in.p, data, n, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_checksum, dim3(nb), dim3(64), 0, st, (const uint8_t*)h->st_in.p, (const fl_chunk*)h->chunks.p,
                       (const uint32_t*)h->blk_chunk.p, (const fl_sblock*)nullptr, prm, h->crc, (uint32_t*)h->cks.p);
    hipLaunchKernelGGL(k_fold_checksum, dim3(1), dim3(64), 0, st, (const uint32_t*)h->cks.p, nb, container, h->crc,
                       (uint32_t*)h->st_status.p);
    HIP_OK(h, hipGetLastError());
    HIP_OK(h, hipMemcpyAsync(value, h->st_status.p, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    HIP_OK(h, hipStreamSynchronize(st));
    return FLATE_HIP_OK;
}

uint32_t flate_hip_checksum_combine(int container, uint32_t a, uint32_t b, uint64_t len_b) {
    if (container == 1) {
        // crc(A || B) = crc(A) * x^(8 |B|) + crc(B) in the reflected representation (as zlib's crc32_combine)
        fl_crc_consts cc;
        init_crc_consts(cc);
        return fl_crc_mulmod(a, fl_crc_xpow8n(cc.xpow8, len_b)) ^ b;
    }
    // Adler-32: a = a1 + a2 - 1, b = b1 + b2 + |B| (a1 - 1)   (mod 65521)
    const uint32_t a1 = a & 0xffff, b1 = a >> 16, a2 = b & 0xffff, b2 = b >> 16;
    const uint32_t rem = (uint32_t)(len_b % 65521u);
    const uint32_t an = (a1 + a2 + 65521u - 1u) % 65521u;
    const uint32_t bn = (uint32_t)(((uint64_t)b1 + b2 + (uint64_t)rem * ((a1 + 65521u - 1u) % 65521u)) % 65521u);
    return an | (bn << 16);
}

}  // extern "C"
