// flate.hpp -- C++ host-side mirror of ianic/flate's public interface over the C ABI of
// libflate_hip.so (include/flate_hip.h).  The reference is Zig; no Zig toolchain exists in this
// environment, so the compiled-language façade is C++ (the Zig façades a maintainer would add are
// in INTEGRATION.md).  Same names, argument meaning and error behaviour as
//   src/flate.zig:9-71   namespace flate_hip::flate   (raw deflate)
//   src/gzip.zig:4-66    namespace flate_hip::gzip
//   src/zlib.zig:4-66    namespace flate_hip::zlib
//
//   compress(reader, writer, Options{level})      flate.zig:28-30
//   Compressor<Writer> / compressor(writer, opt)  flate.zig:33-40   write / compress / finish
//   decompress(reader, writer)                    flate.zig:10-12
//   Decompressor<Reader> / decompressor(reader)   flate.zig:15-22   decompress / next / read / reset
//   huffman::{compress,Compressor,compressor}     flate.zig:44-56
//   store::{compress,Compressor,compressor}       flate.zig:59-71
//
// Reader: anything with  size_t read(uint8_t* buf, size_t n)  (0 = end; Zig's readAll contract is
// met by looping).  Writer: anything with  void write(const uint8_t* buf, size_t n).
// Errors: flate_hip::Error carrying the reference's error name (inflate.zig:72-78 etc.).
//
// One-shot semantics run on the GPU, for inputs of any length: at levels 4..9 an input longer than
// 65535 bytes is compressed as one stream by the whole-stream path (same bytes as the reference's
// sliding-window compressor).  Compressor::flush (history-preserving sync flush,
// deflate.zig:335-337, 474-478) runs on the GPU as well.  There is no CPU fallback.
#pragma once
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/flate_hip.h"

namespace flate_hip {

enum class Level : int {  // deflate.zig:23-32
    fast = 4, level_4 = 4, level_5 = 5, default_ = 6, level_6 = 6, level_7 = 7, level_8 = 8, best = 9, level_9 = 9
};
struct Options {  // deflate.zig:15-17
    Level level = Level::default_;
};

struct Error : std::runtime_error {
    int status;
    explicit Error(int st) : std::runtime_error(flate_hip_status_name(st)), status(st) {}
    Error(int st, const std::string& what) : std::runtime_error(what), status(st) {}
};

// One engine handle per process/device (one process per GPU).
class Engine {
   public:
    explicit Engine(int device = 0) {
        const int rc = flate_hip_create(device, &h_);
        if (rc != FLATE_HIP_OK) throw Error(rc, "flate_hip_create failed: no usable MI355X (no CPU fallback)");
    }
    ~Engine() {
        if (h_) flate_hip_destroy(h_);
    }
    Engine(const Engine&) = delete;
    Engine& operator=(const Engine&) = delete;
    flate_hip_handle handle() const { return h_; }
    // FLATE_HIP_DEFLATE_REPAIR_Q1: every block is handed the bytes its tokens cover (streams that always inflate to
    // their input; they differ from the reference's only where the reference's own stream is broken, below)
    void set_flags(uint32_t flags) {
        const int rc = flate_hip_set_flags(h_, flags);
        if (rc != FLATE_HIP_OK) throw Error(rc, "flate_hip_set_flags");
    }
    // streams written so far that are byte for byte the reference's and do not inflate to their input (quirk Q1,
    // FLATE_HIP_ST_REFERENCE_Q1_STREAM in include/flate_hip.h): the reference writes them without a word, so does the
    // facade -- and counts them
    uint64_t reference_q1_streams() const { return q1_streams_; }
    static Engine& instance() {
        static Engine e(0);
        return e;
    }

    std::vector<uint8_t> compress_one(const std::vector<uint8_t>& in, int container, int mode) {
        const uint64_t in_off[2] = {0, in.size()};
        const size_t cap = (flate_hip_compress_bound(in.size(), container, mode) + 7) & ~size_t(7);
        const uint64_t out_off[2] = {0, cap};
        std::vector<uint8_t> out(cap + 8);
        uint64_t out_len = 0;
        int32_t status = 0;
        const uint8_t dummy = 0;
        const int rc = flate_hip_compress_batch(h_, in.empty() ? &dummy : in.data(), in_off, 1, container, mode,
                                                out.data(), out_off, &out_len, &status, FLATE_HIP_MEM_HOST);
        if (rc != FLATE_HIP_OK) throw Error(rc, std::string("flate_hip_compress_batch: ") + flate_hip_last_error(h_));
        if (status == FLATE_HIP_ST_REFERENCE_Q1_STREAM) q1_streams_++;
        else if (status) throw Error(status);
        out.resize(out_len);
        return out;
    }
    uint32_t checksum(const uint8_t* p, size_t n, int container) {
        uint32_t v = 0;
        static const uint8_t dummy = 0;
        const int rc = flate_hip_checksum(h_, n ? p : &dummy, n, container, &v);
        if (rc != FLATE_HIP_OK) throw Error(rc, std::string("flate_hip_checksum: ") + flate_hip_last_error(h_));
        return v;
    }
    // the stream a Compressor has written after write/flush/.../[finish] (flate_hip_compress_flush)
    std::vector<uint8_t> compress_flush(const std::vector<uint8_t>& in, const std::vector<uint64_t>& flushes, bool finish,
                                        int container, int mode) {
        const size_t cap = (flate_hip_compress_bound(in.size(), container, mode) + 64 * (flushes.size() + 1) + 7) & ~size_t(7);
        std::vector<uint8_t> out(cap + 8);
        uint64_t out_len = 0;
        int32_t status = 0;
        const uint8_t dummy = 0;
        const int rc = flate_hip_compress_flush(h_, in.empty() ? &dummy : in.data(), in.size(),
                                                flushes.empty() ? nullptr : flushes.data(), (uint32_t)flushes.size(),
                                                finish ? 1 : 0, container, mode, out.data(), cap, &out_len, &status,
                                                FLATE_HIP_MEM_HOST);
        if (rc != FLATE_HIP_OK) throw Error(rc, std::string("flate_hip_compress_flush: ") + flate_hip_last_error(h_));
        if (status == FLATE_HIP_ST_REFERENCE_Q1_STREAM) q1_streams_++;
        else if (status) throw Error(status);
        out.resize(out_len);
        return out;
    }
    // returns bytes consumed from `in`
    size_t decompress_one(const uint8_t* in, size_t n, int container, std::vector<uint8_t>& out) {
        size_t cap = n * 8 + (1 << 16);
        for (;;) {
            const uint64_t in_off[2] = {0, n};
            const uint64_t out_off[2] = {0, cap};
            out.assign(cap + 8, 0);
            uint64_t out_len = 0, consumed = 0;
            int32_t status = 0;
            const uint8_t dummy = 0;
            const int rc = flate_hip_decompress_batch(h_, n ? in : &dummy, in_off, 1, container, 0, out.data(), out_off,
                                                      &out_len, &status, &consumed, FLATE_HIP_MEM_HOST);
            if (rc != FLATE_HIP_OK)
                throw Error(rc, std::string("flate_hip_decompress_batch: ") + flate_hip_last_error(h_));
            if (status == FLATE_HIP_ST_OUTPUT_TOO_SMALL && cap < (size_t(1) << 36)) {
                cap *= 8;
                continue;
            }
            if (status) throw Error(status);
            out.resize(out_len);
            return consumed;
        }
    }

   private:
    flate_hip_handle h_ = nullptr;
    uint64_t q1_streams_ = 0;
};

namespace detail {

template <class Reader>
inline void read_all(Reader& r, std::vector<uint8_t>& buf) {
    uint8_t tmp[65536];
    for (;;) {
        const size_t k = r.read(tmp, sizeof tmp);
        if (k == 0) break;
        buf.insert(buf.end(), tmp, tmp + k);
    }
}

// Deflate (deflate.zig:121-373) / SimpleCompressor (:449-529) seen from the caller.
// One-shot use is one GPU call.  With flush() (deflate.zig:335-337: pending tokens out, then an empty
// stored block; the LZ history stays) the object works incrementally: what the reference emits after a
// flush point F depends only on the stream from a 32 KiB-aligned position B <= F - 96 KiB on (its 64 KiB
// window has slid past everything older, and the slide schedule is periodic in 32 KiB), so every flush /
// finish runs flate_hip_compress_flush on the retained tail [B, now) only and hands the writer the bytes
// after the previous flush's marker: O(new bytes + 128 KiB) per flush.  The container header goes out
// once, the footer's checksum is folded from per-piece checksums (flate_hip_checksum / _combine).
template <class Writer>
class CompressorImpl {
   public:
    CompressorImpl(Writer& w, int container, int mode) : wrt_(&w), container_(container), mode_(mode) {}
    size_t write(const uint8_t* p, size_t n) {  // deflate.zig:363-367
        live();
        buf_.insert(buf_.end(), p, p + n);
        total_ += n;
        return n;
    }
    template <class Reader>
    void compress(Reader& r) {  // deflate.zig:304-321
        live();
        const size_t before = buf_.size();
        read_all(r, buf_);
        total_ += buf_.size() - before;
    }
    void flush() {
        live();
        fold_checksum();
        const bool first = nflush_ == 0;
        flushes_.push_back(total_);
        nflush_++;
        if (first) {
            rel_emitted_ = 0;
            have_rel_ = true;
            static const uint8_t gz[10] = {0x1f, 0x8b, 0x08, 0, 0, 0, 0, 0, 0, 0x03};  // container.zig:64
            static const uint8_t zl[2] = {0x78, 0x9c};                                 // container.zig:78
            if (container_ == 1) wrt_->write(gz, sizeof gz);
            if (container_ == 2) wrt_->write(zl, sizeof zl);
        }
        run_tail(false);
        // drop the history no later piece can depend on
        const uint64_t last = flushes_.back();
        if (last >= 98304) {
            const uint64_t nb = ((last - 98304) / 32768) * 32768;
            if (nb > base_) {
                buf_.erase(buf_.begin(), buf_.begin() + (nb - base_));
                std::vector<uint64_t> keep;
                for (uint64_t f : flushes_)
                    if (f >= nb) keep.push_back(f);
                flushes_.swap(keep);
                base_ = nb;
                have_rel_ = false;
            }
        }
    }
    void setWriter(Writer& w) { wrt_ = &w; }  // deflate.zig:351-354
    void finish() {                           // deflate.zig:344-347
        if (done_) return;
        if (nflush_ == 0) {
            const std::vector<uint8_t> out = Engine::instance().compress_one(buf_, container_, mode_);
            wrt_->write(out.data(), out.size());
        } else {
            fold_checksum();
            run_tail(true);
            if (container_ == 1) {  // container.zig:92-96
                uint8_t f[8];
                for (int i = 0; i < 4; i++) f[i] = (uint8_t)(cks_ >> (8 * i));
                for (int i = 0; i < 4; i++) f[4 + i] = (uint8_t)(total_ >> (8 * i));
                wrt_->write(f, 8);
            } else if (container_ == 2) {  // container.zig:104
                uint8_t f[4];
                for (int i = 0; i < 4; i++) f[i] = (uint8_t)(cks_ >> (8 * (3 - i)));
                wrt_->write(f, 4);
            }
        }
        done_ = true;
    }

   private:
    void live() const {
        // (the reference has no such check: writing after finish() emits a broken stream)
        if (done_) throw Error(FLATE_HIP_E_INVALID_ARG, "compressor used after finish()");
    }
    void fold_checksum() {
        if (container_ == 0) return;
        const size_t from = (size_t)(cks_pos_ - base_);
        const uint32_t v = Engine::instance().checksum(buf_.data() + from, buf_.size() - from, container_);
        cks_ = have_cks_ ? flate_hip_checksum_combine(container_, cks_, v, buf_.size() - from) : v;
        have_cks_ = true;
        cks_pos_ = total_;
    }
    void run_tail(bool finish) {
        std::vector<uint64_t> rel;
        for (uint64_t f : flushes_) rel.push_back(f - base_);
        Engine& e = Engine::instance();
        if (!have_rel_) {
            // how much of the tail's output went out already: the tail up to the previous flush
            const uint64_t prev = finish ? rel.back() : rel[rel.size() - 2];
            std::vector<uint64_t> r2(rel.begin(), finish ? rel.end() : rel.end() - 1);
            const std::vector<uint8_t> head(buf_.begin(), buf_.begin() + prev);
            rel_emitted_ = e.compress_flush(head, r2, false, 0, mode_).size();
            have_rel_ = true;
        }
        const std::vector<uint8_t> out = e.compress_flush(buf_, rel, finish, 0, mode_);
        wrt_->write(out.data() + rel_emitted_, out.size() - rel_emitted_);
        rel_emitted_ = out.size();
    }
    Writer* wrt_;
    int container_, mode_;
    std::vector<uint8_t> buf_;      // stream bytes from absolute position base_ on
    uint64_t base_ = 0, total_ = 0;
    std::vector<uint64_t> flushes_;  // absolute flush points >= base_
    size_t nflush_ = 0;
    size_t rel_emitted_ = 0;
    bool have_rel_ = false;
    uint32_t cks_ = 0;
    bool have_cks_ = false;
    uint64_t cks_pos_ = 0;
    bool done_ = false;
};

// Inflate (inflate.zig:43-355) seen from the caller
// Inflate (inflate.zig:43-355) seen from the caller.  The reader is consumed as far as the current stream
// needs it, in steps that double: a decode that runs out of input (EndOfStream) while the reader still has
// bytes is repeated with twice as much; what was read past the end of a stream stays buffered for reset().
template <class Reader>
class DecompressorImpl {
   public:
    DecompressorImpl(Reader& r, int container) : rd_(&r), container_(container) {}
    // next(): slices of at most 64 KiB, empty = end of stream (inflate.zig:315-336)
    std::pair<const uint8_t*, size_t> next() {
        decode();
        const size_t n = std::min<size_t>(out_.size() - rp_, 65536);
        const uint8_t* p = out_.data() + rp_;
        rp_ += n;
        if (n == 0) ended_ = true;
        return {p, n};
    }
    size_t read(uint8_t* buf, size_t n) {  // inflate.zig:343-347
        decode();
        const size_t k = std::min(n, out_.size() - rp_);
        memcpy(buf, out_.data() + rp_, k);
        rp_ += k;
        if (k == 0) ended_ = true;
        return k;
    }
    template <class Writer>
    void decompress(Writer& w) {  // inflate.zig:292-296
        for (;;) {
            auto s = next();
            if (s.second == 0) break;
            w.write(s.first, s.second);
        }
    }
    void reset() {  // inflate.zig:301-309: next stream of the same reader
        if (!ended_) throw Error(102, "InvalidState");
        pos_ += used_;
        decoded_ = false;
        ended_ = false;
        rp_ = 0;
        out_.clear();
    }

   private:
    void fill(size_t want) {  // `want` bytes of the current stream buffered, or the reader at its end
        uint8_t tmp[65536];
        while (!eof_ && in_.size() - pos_ < want) {
            const size_t k = rd_->read(tmp, std::min(sizeof tmp, want - (in_.size() - pos_)));
            if (k == 0) {
                eof_ = true;
                break;
            }
            in_.insert(in_.end(), tmp, tmp + k);
        }
    }
    void decode() {
        if (decoded_) return;
        size_t want = 65536;
        for (;;) {
            fill(want);
            try {
                used_ = Engine::instance().decompress_one(in_.data() + pos_, in_.size() - pos_, container_, out_);
                break;
            } catch (const Error& e) {
                if (e.status != 1 /* EndOfStream */ || eof_) throw;
                want = 2 * std::max(want, in_.size() - pos_);  // input still to come: not an error yet
            }
        }
        decoded_ = true;
    }
    Reader* rd_;
    int container_;
    std::vector<uint8_t> in_, out_;
    size_t pos_ = 0, used_ = 0, rp_ = 0;
    bool decoded_ = false, ended_ = false, eof_ = false;
};

}  // namespace detail

#define FLATE_HIP_CONTAINER_NS(NS, TAG)                                                                      \
    namespace NS {                                                                                           \
    using Options = ::flate_hip::Options;                                                                    \
    using Level = ::flate_hip::Level;                                                                        \
    template <class Writer>                                                                                  \
    struct Compressor : detail::CompressorImpl<Writer> {                                                     \
        Compressor(Writer& w, Options o = {}) : detail::CompressorImpl<Writer>(w, TAG, (int)o.level) {}      \
    };                                                                                                       \
    template <class Writer>                                                                                  \
    Compressor<Writer> compressor(Writer& w, Options o = {}) {                                               \
        return Compressor<Writer>(w, o);                                                                     \
    }                                                                                                        \
    template <class Reader, class Writer>                                                                    \
    void compress(Reader& r, Writer& w, Options o = {}) {                                                    \
        auto c = compressor(w, o);                                                                           \
        c.compress(r);                                                                                       \
        c.finish();                                                                                          \
    }                                                                                                        \
    template <class Reader>                                                                                  \
    struct Decompressor : detail::DecompressorImpl<Reader> {                                                 \
        explicit Decompressor(Reader& r) : detail::DecompressorImpl<Reader>(r, TAG) {}                       \
    };                                                                                                       \
    template <class Reader>                                                                                  \
    Decompressor<Reader> decompressor(Reader& r) {                                                           \
        return Decompressor<Reader>(r);                                                                      \
    }                                                                                                        \
    template <class Reader, class Writer>                                                                    \
    void decompress(Reader& r, Writer& w) {                                                                  \
        decompressor(r).decompress(w);                                                                       \
    }                                                                                                        \
    namespace huffman {                                                                                      \
    template <class Writer>                                                                                  \
    struct Compressor : detail::CompressorImpl<Writer> {                                                     \
        explicit Compressor(Writer& w) : detail::CompressorImpl<Writer>(w, TAG, FLATE_HIP_MODE_HUFFMAN) {}   \
    };                                                                                                       \
    template <class Writer>                                                                                  \
    Compressor<Writer> compressor(Writer& w) {                                                               \
        return Compressor<Writer>(w);                                                                        \
    }                                                                                                        \
    template <class Reader, class Writer>                                                                    \
    void compress(Reader& r, Writer& w) {                                                                    \
        auto c = compressor(w);                                                                              \
        c.compress(r);                                                                                       \
        c.finish();                                                                                          \
    }                                                                                                        \
    }                                                                                                        \
    namespace store {                                                                                        \
    template <class Writer>                                                                                  \
    struct Compressor : detail::CompressorImpl<Writer> {                                                     \
        explicit Compressor(Writer& w) : detail::CompressorImpl<Writer>(w, TAG, FLATE_HIP_MODE_STORE) {}     \
    };                                                                                                       \
    template <class Writer>                                                                                  \
    Compressor<Writer> compressor(Writer& w) {                                                               \
        return Compressor<Writer>(w);                                                                        \
    }                                                                                                        \
    template <class Reader, class Writer>                                                                    \
    void compress(Reader& r, Writer& w) {                                                                    \
        auto c = compressor(w);                                                                              \
        c.compress(r);                                                                                       \
        c.finish();                                                                                          \
    }                                                                                                        \
    }                                                                                                        \
    }

FLATE_HIP_CONTAINER_NS(flate, FLATE_HIP_RAW)
FLATE_HIP_CONTAINER_NS(gzip, FLATE_HIP_GZIP)
FLATE_HIP_CONTAINER_NS(zlib, FLATE_HIP_ZLIB)

// fixed-buffer reader / growing writer, the C++ twins of std.io.fixedBufferStream / ArrayList writer
struct BufferReader {
    const uint8_t* p;
    size_t n, pos = 0;
    BufferReader(const uint8_t* p_, size_t n_) : p(p_), n(n_) {}
    size_t read(uint8_t* buf, size_t k) {
        k = std::min(k, n - pos);
        memcpy(buf, p + pos, k);
        pos += k;
        return k;
    }
};
struct VectorWriter {
    std::vector<uint8_t> data;
    void write(const uint8_t* buf, size_t k) { data.insert(data.end(), buf, buf + k); }
};

}  // namespace flate_hip
