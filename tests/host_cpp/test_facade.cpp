// "flate public interface" (flate.zig:356-481) through the C++ façade.  Needs a GPU to run.
#include <cstdio>
#include <cstdlib>

#include "../../flate_amd/host/flate.hpp"

using namespace flate_hip;

#define CHECK(c)                                                         \
    do {                                                                 \
        if (!(c)) {                                                      \
            fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); \
            exit(1);                                                     \
        }                                                                \
    } while (0)

template <class F>
static std::vector<uint8_t> run(F f, const std::vector<uint8_t>& in) {
    BufferReader r(in.data(), in.size());
    VectorWriter w;
    f(r, w);
    return w.data;
}

int main() {
    const std::vector<uint8_t> plain = {'H', 'e', 'l', 'l', 'o', ' ', 'w', 'o', 'r', 'l', 'd', 0x0a};
    std::vector<uint8_t> block = {0x01, 0x0c, 0x00, 0xf3, 0xff};
    block.insert(block.end(), plain.begin(), plain.end());
    std::vector<uint8_t> gz = {0x1f, 0x8b, 0x08, 0, 0, 0, 0, 0, 0, 0x03};
    gz.insert(gz.end(), block.begin(), block.end());
    for (uint8_t b : {0xd5, 0xe0, 0x39, 0xb7, 0x0c, 0x00, 0x00, 0x00}) gz.push_back(b);
    std::vector<uint8_t> zl = {0x78, 0x9c};
    zl.insert(zl.end(), block.begin(), block.end());
    for (uint8_t b : {0x1c, 0xf2, 0x04, 0x47}) zl.push_back(b);

    CHECK(run([](auto& r, auto& w) { gzip::decompress(r, w); }, gz) == plain);
    CHECK(run([](auto& r, auto& w) { zlib::decompress(r, w); }, zl) == plain);
    CHECK(run([](auto& r, auto& w) { flate::decompress(r, w); }, block) == plain);
    CHECK(run([](auto& r, auto& w) { gzip::store::compress(r, w); }, plain) == gz);
    CHECK(run([](auto& r, auto& w) { zlib::store::compress(r, w); }, plain) == zl);
    CHECK(run([](auto& r, auto& w) { flate::store::compress(r, w); }, plain) == block);
    // compress / decompress, compressor / decompressor, huffman (flate.zig:399-447)
    auto c1 = run([](auto& r, auto& w) { gzip::compress(r, w, gzip::Options{}); }, plain);
    CHECK(run([](auto& r, auto& w) { gzip::decompress(r, w); }, c1) == plain);
    VectorWriter cw;
    auto cmp = zlib::compressor(cw, zlib::Options{Level::best});
    cmp.write(plain.data(), 5);
    cmp.write(plain.data() + 5, plain.size() - 5);
    cmp.finish();
    BufferReader rr(cw.data.data(), cw.data.size());
    auto dcp = zlib::decompressor(rr);
    uint8_t buf[64];
    CHECK(dcp.read(buf, sizeof buf) == plain.size() && memcmp(buf, plain.data(), plain.size()) == 0);
    // the decompressor reads its reader as it goes (inflate.zig:283-353): two members, then 4 MiB of other bytes
    {
        std::vector<uint8_t> a(150000), b(90);
        for (size_t i = 0; i < a.size(); i++) a[i] = (uint8_t)("stream one "[i % 11] + (i / 3000) % 5);
        for (size_t i = 0; i < b.size(); i++) b[i] = (uint8_t)('a' + i % 7);
        auto ca = run([](auto& r, auto& w) { gzip::compress(r, w); }, a);
        auto cb = run([](auto& r, auto& w) { gzip::compress(r, w); }, b);
        std::vector<uint8_t> blob(ca);
        blob.insert(blob.end(), cb.begin(), cb.end());
        blob.resize(blob.size() + (4u << 20), 0);
        BufferReader br(blob.data(), blob.size());
        auto d = gzip::decompressor(br);
        VectorWriter w1, w2;
        d.decompress(w1);
        d.reset();
        d.decompress(w2);
        CHECK(w1.data == a && w2.data == b);
        CHECK(br.pos < ca.size() + cb.size() + (1u << 20));
    }
    auto h1 = run([](auto& r, auto& w) { flate::huffman::compress(r, w); }, plain);
    CHECK(run([](auto& r, auto& w) { flate::decompress(r, w); }, h1) == plain);
    // error names (flate.zig:267-295)
    try {
        run([](auto& r, auto& w) { zlib::decompress(r, w); }, std::vector<uint8_t>{0x79, 0x94});
        CHECK(false);
    } catch (const Error& e) {
        CHECK(std::string(e.what()) == "BadZlibHeader");
    }
    // inputs longer than one 64 KiB window go through the whole-stream path as ONE stream
    {
        std::vector<uint8_t> big(200000);
        for (size_t i = 0; i < big.size(); i++) big[i] = (uint8_t)("flate on gfx950 "[i % 16] + (i / 4096) % 7);
        auto c = run([](auto& r, auto& w) { gzip::compress(r, w); }, big);
        CHECK(c.size() < big.size() / 4);
        CHECK(run([](auto& r, auto& w) { gzip::decompress(r, w); }, c) == big);
    }
    // write / flush / write / finish (flate.zig:33-40, deflate.zig:335-367): the flushed part decodes on its own
    {
        std::vector<uint8_t> big(90000);
        for (size_t i = 0; i < big.size(); i++) big[i] = (uint8_t)("sync flush keeps history "[i % 25] + (i / 5000) % 3);
        VectorWriter w;
        auto c = zlib::compressor(w);
        c.write(big.data(), 40000);
        c.flush();
        const size_t n1 = w.data.size();
        CHECK(n1 > 4 && w.data[n1 - 4] == 0x00 && w.data[n1 - 3] == 0x00 && w.data[n1 - 2] == 0xff && w.data[n1 - 1] == 0xff);
        c.write(big.data() + 40000, big.size() - 40000);
        c.finish();
        CHECK(w.data.size() > n1);
        BufferReader r2(w.data.data(), w.data.size());
        VectorWriter back;
        zlib::decompress(r2, back);
        CHECK(back.data == big);
    }
    // many flushes over a stream several windows long: the compressor keeps only the tail, and the
    // stream equals the one-call stream with the same flush points (incremental == whole re-run)
    {
        std::vector<uint8_t> big(700000);
        uint32_t x = 12345;
        for (size_t i = 0; i < big.size(); i++) {
            x = x * 1664525u + 1013904223u;
            big[i] = (uint8_t)("the quick brown fox jumps over the lazy dog "[(i + (x >> 28)) % 44]);
        }
        VectorWriter w;
        auto c = gzip::compressor(w, gzip::Options{Level::level_6});
        std::vector<uint64_t> fl;
        size_t pos = 0;
        for (size_t step : {1000u, 64535u, 1u, 32768u, 200000u, 99999u, 131072u, 70000u}) {
            c.write(big.data() + pos, step);
            pos += step;
            c.flush();
            fl.push_back(pos);
        }
        c.write(big.data() + pos, big.size() - pos);
        c.finish();
        const std::vector<uint8_t> whole = Engine::instance().compress_flush(big, fl, true, 1, 6);
        CHECK(w.data == whole);
        BufferReader r3(w.data.data(), w.data.size());
        VectorWriter back;
        gzip::decompress(r3, back);
        CHECK(back.data == big);
    }
    printf("facade ok\n");
    return 0;
}
