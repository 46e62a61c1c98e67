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
        if (status) throw Error(status);
        out.resize(out_len);
        return out;
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
        if (status) throw Error(status);
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

// Deflate (deflate.zig:121-373) / SimpleCompressor (:449-529) seen from the caller
template <class Writer>
class CompressorImpl {
   public:
    CompressorImpl(Writer& w, int container, int mode) : wrt_(&w), container_(container), mode_(mode) {}
    size_t write(const uint8_t* p, size_t n) {  // deflate.zig:363-367
        buf_.insert(buf_.end(), p, p + n);
        return n;
    }
    template <class Reader>
    void compress(Reader& r) {  // deflate.zig:304-321
        read_all(r, buf_);
    }
    // deflate.zig:335-337: pending tokens out, then an empty stored block; the LZ history stays.
    // The stream so far is re-run with its flush points; what a shorter prefix of the calls has
    // produced is a prefix of it, so only the new bytes go to the writer.
    void flush() {
        flushes_.push_back(buf_.size());
        emit(Engine::instance().compress_flush(buf_, flushes_, false, container_, mode_));
    }
    void setWriter(Writer& w) { wrt_ = &w; }  // deflate.zig:351-354
    void finish() {                           // deflate.zig:344-347
        if (done_) return;
        if (flushes_.empty())
            emit(Engine::instance().compress_one(buf_, container_, mode_));
        else
            emit(Engine::instance().compress_flush(buf_, flushes_, true, container_, mode_));
        done_ = true;
    }

   private:
    void emit(const std::vector<uint8_t>& out) {
        wrt_->write(out.data() + emitted_, out.size() - emitted_);
        emitted_ = out.size();
    }
    Writer* wrt_;
    int container_, mode_;
    std::vector<uint8_t> buf_;
    std::vector<uint64_t> flushes_;
    size_t emitted_ = 0;
    bool done_ = false;
};

// Inflate (inflate.zig:43-355) seen from the caller
template <class Reader>
class DecompressorImpl {
   public:
    DecompressorImpl(Reader& r, int container) : container_(container) { read_all(r, in_); }
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
    void decode() {
        if (decoded_) return;
        used_ = Engine::instance().decompress_one(in_.data() + pos_, in_.size() - pos_, container_, out_);
        decoded_ = true;
    }
    int container_;
    std::vector<uint8_t> in_, out_;
    size_t pos_ = 0, used_ = 0, rp_ = 0;
    bool decoded_ = false, ended_ = false;
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
