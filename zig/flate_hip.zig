//! Zig host side of the MI355X DEFLATE engine: the API of ianic/flate's src/flate.zig,
//! src/gzip.zig and src/zlib.zig (compress / decompress / Compressor / Decompressor /
//! huffman.* / store.*) over the C ABI of include/flate_hip.h.
//!
//! NOT COMPILED IN THIS REPOSITORY'S IMAGE (there is no Zig toolchain here); written against
//! the Zig 0.12-era std the reference itself targets.  The same façade in C++
//! (flate_amd/host/flate.hpp) and Python (flate_amd/api.py) is compiled / run and tested.
//! Deviation from the reference: the reference keeps all state inline and needs no allocator;
//! this façade buffers in `std.heap.page_allocator` memory what the GPU entry points need at once
//! (a one-shot stream, or the retained tail of a stream between sync flushes, see DESIGN.md 1).
const std = @import("std");

// ---- include/flate_hip.h ----
pub const Handle = ?*anyopaque;
pub extern "c" fn flate_hip_create(device: c_int, h: *Handle) c_int;
pub extern "c" fn flate_hip_destroy(h: Handle) c_int;
pub extern "c" fn flate_hip_compress_bound(n: usize, container: c_int, mode: c_int) usize;
pub extern "c" fn flate_hip_compress_batch(h: Handle, in: [*]const u8, in_off: [*]const u64, n_chunks: u32, container: c_int, mode: c_int, out: [*]u8, out_off: [*]const u64, out_len: [*]u64, status: [*]i32, memkind: c_int) c_int;
pub extern "c" fn flate_hip_compress_flush(h: Handle, in: [*]const u8, n: u64, flush_pos: ?[*]const u64, n_flush: u32, finish: c_int, container: c_int, mode: c_int, out: [*]u8, out_cap: u64, out_len: *u64, status: *i32, memkind: c_int) c_int;
pub extern "c" fn flate_hip_decompress_batch(h: Handle, in: [*]const u8, in_off: [*]const u64, n_chunks: u32, container: c_int, flags: c_int, out: [*]u8, out_off: [*]const u64, out_len: [*]u64, status: [*]i32, consumed: ?[*]u64, memkind: c_int) c_int;

/// inflate.zig:72-78, huffman_decoder.zig:35-40, container.zig:45-51, bit_reader.zig:29
pub const Error = error{
    EndOfStream,
    BadGzipHeader,
    BadZlibHeader,
    WrongGzipChecksum,
    WrongGzipSize,
    WrongZlibChecksum,
    InvalidCode,
    OversubscribedHuffmanTree,
    IncompleteHuffmanTree,
    MissingEndOfBlockCode,
    InvalidMatch,
    InvalidBlockType,
    WrongStoredBlockNlen,
    InvalidDynamicBlockHeader,
    OutputTooSmall,
    DeviceError,
    OutOfMemory,
};

fn statusToError(st: i32) Error!void {
    return switch (st) {
        0 => {},
        1 => error.EndOfStream,
        2 => error.BadGzipHeader,
        3 => error.BadZlibHeader,
        4 => error.WrongGzipChecksum,
        5 => error.WrongGzipSize,
        6 => error.WrongZlibChecksum,
        7 => error.InvalidCode,
        8 => error.OversubscribedHuffmanTree,
        9 => error.IncompleteHuffmanTree,
        10 => error.MissingEndOfBlockCode,
        11 => error.InvalidMatch,
        12 => error.InvalidBlockType,
        13 => error.WrongStoredBlockNlen,
        14 => error.InvalidDynamicBlockHeader,
        100 => error.OutputTooSmall,
        // 102 (FLATE_HIP_ST_REFERENCE_Q1_STREAM) is not an error here: the bytes are the reference's own for this input
        // (deflate.zig:227-230 flushes a full token block before :193 advances the window) -- runCompress counts them
        else => error.DeviceError,
    };
}

/// deflate.zig:23-32
pub const DeflateLevel = enum(u4) {
    fast = 0xb,
    level_4 = 4,
    level_5 = 5,
    default = 0xc,
    level_6 = 6,
    level_7 = 7,
    level_8 = 8,
    best = 0xd,
    level_9 = 9,
};
/// deflate.zig:15-17
pub const DeflateOptions = struct { level: DeflateLevel = .default };

fn modeOf(l: DeflateLevel) c_int {
    return switch (l) {
        .fast, .level_4 => 4,
        .level_5 => 5,
        .default, .level_6 => 6,
        .level_7 => 7,
        .level_8 => 8,
        .best, .level_9 => 9,
    };
}
const mode_huffman: c_int = 1;
const mode_store: c_int = 0;

const gpa = std.heap.page_allocator;
var g_handle: Handle = null;

/// Streams written so far that are byte for byte the reference's and do not inflate to their input (quirk Q1: include/flate_hip.h,
/// FLATE_HIP_ST_REFERENCE_Q1_STREAM).  `setRepairQ1(true)` makes every stream inflate to its input instead (bytes then differ
/// from the reference's on exactly those inputs).
pub var reference_q1_streams: u64 = 0;
pub extern "c" fn flate_hip_set_flags(h: Handle, flags: u32) c_int;
pub fn setRepairQ1(on: bool) Error!void {
    const h = try engine();
    if (flate_hip_set_flags(h, if (on) 1 else 0) != 0) return error.DeviceError;
}

/// One engine per process, created on first use.  There is no CPU fallback: without a usable
/// MI355X every call fails with error.DeviceError.
fn engine() Error!Handle {
    if (g_handle == null) {
        if (flate_hip_create(0, &g_handle) != 0) return error.DeviceError;
    }
    return g_handle;
}

/// everything a Compressor has written after the calls so far (flate_hip_compress_flush), or the
/// one-shot stream when there was no flush
fn runCompress(input: []const u8, flushes: []const u64, finish: bool, container: c_int, mode: c_int) Error![]u8 {
    const h = try engine();
    const cap = flate_hip_compress_bound(input.len, container, mode) + 64 * (flushes.len + 1);
    const out = gpa.alloc(u8, cap + 8) catch return error.OutOfMemory;
    errdefer gpa.free(out);
    var out_len: u64 = 0;
    var status: i32 = 0;
    const dummy = [1]u8{0};
    const in_ptr: [*]const u8 = if (input.len == 0) &dummy else input.ptr;
    if (flushes.len == 0 and finish) {
        const in_off = [2]u64{ 0, input.len };
        const out_off = [2]u64{ 0, cap };
        var lens = [1]u64{0};
        var sts = [1]i32{0};
        if (flate_hip_compress_batch(h, in_ptr, &in_off, 1, container, mode, out.ptr, &out_off, &lens, &sts, 0) != 0)
            return error.DeviceError;
        out_len = lens[0];
        status = sts[0];
    } else {
        if (flate_hip_compress_flush(h, in_ptr, input.len, if (flushes.len == 0) null else flushes.ptr, @intCast(flushes.len), @intFromBool(finish), container, mode, out.ptr, cap, &out_len, &status, 0) != 0)
            return error.DeviceError;
    }
    if (status == 102) {
        reference_q1_streams += 1;
        status = 0;
    }
    try statusToError(status);
    return out[0..@intCast(out_len)]; // caller frees the whole allocation via `gpa.free(slice.ptr[0 .. cap + 8])`: see CompressorImpl.emit
}

fn CompressorImpl(comptime container: c_int, comptime WriterType: type) type {
    return struct {
        wrt: WriterType,
        mode: c_int,
        buf: std.ArrayList(u8),
        flushes: std.ArrayList(u64),
        emitted: usize = 0,
        done: bool = false,

        const Self = @This();
        pub const Writer = std.io.Writer(*Self, Error || WriterType.Error, write);

        /// deflate.zig:138
        pub fn init(wrt: WriterType, mode: c_int) Self {
            return .{ .wrt = wrt, .mode = mode, .buf = std.ArrayList(u8).init(gpa), .flushes = std.ArrayList(u64).init(gpa) };
        }
        pub fn deinit(self: *Self) void {
            self.buf.deinit();
            self.flushes.deinit();
        }
        /// deflate.zig:363-367
        pub fn write(self: *Self, input: []const u8) !usize {
            self.buf.appendSlice(input) catch return error.OutOfMemory;
            return input.len;
        }
        /// deflate.zig:369-371
        pub fn writer(self: *Self) Writer {
            return .{ .context = self };
        }
        /// deflate.zig:304-321
        pub fn compress(self: *Self, reader: anytype) !void {
            var tmp: [65536]u8 = undefined;
            while (true) {
                const n = try reader.readAll(&tmp);
                self.buf.appendSlice(tmp[0..n]) catch return error.OutOfMemory;
                if (n < tmp.len) break;
            }
        }
        fn emit(self: *Self, finish: bool) !void {
            const cap = flate_hip_compress_bound(self.buf.items.len, container, self.mode) + 64 * (self.flushes.items.len + 1);
            const out = try runCompress(self.buf.items, self.flushes.items, finish, container, self.mode);
            defer gpa.free(out.ptr[0 .. cap + 8]);
            // the stream after more calls extends the stream after fewer: hand over what is new
            try self.wrt.writeAll(out[self.emitted..]);
            self.emitted = out.len;
        }
        /// deflate.zig:335-337 (levels 4-9) / 474-478 (huffman-only, store-only): pending data goes
        /// out as a block of its own, then an empty stored block; the LZ77 history stays
        pub fn flush(self: *Self) !void {
            self.flushes.append(self.buf.items.len) catch return error.OutOfMemory;
            try self.emit(false);
        }
        /// deflate.zig:344-347
        pub fn finish(self: *Self) !void {
            if (self.done) return;
            try self.emit(true);
            self.done = true;
        }
        /// deflate.zig:351-354
        pub fn setWriter(self: *Self, new_writer: WriterType) void {
            self.wrt = new_writer;
        }
    };
}

fn DecompressorImpl(comptime container: c_int, comptime ReaderType: type) type {
    return struct {
        rdr: ReaderType,
        input: std.ArrayList(u8),
        pos: usize = 0, // start of the current stream in `input`
        out: ?[]u8 = null,
        out_cap: usize = 0,
        used: usize = 0,
        rp: usize = 0,
        loaded: bool = false,
        ended: bool = false,

        const Self = @This();
        pub const Reader = std.io.Reader(*Self, Error || ReaderType.Error, read);

        /// inflate.zig:80
        pub fn init(rdr: ReaderType) Self {
            return .{ .rdr = rdr, .input = std.ArrayList(u8).init(gpa) };
        }
        pub fn deinit(self: *Self) void {
            if (self.out) |o| gpa.free(o.ptr[0..self.out_cap]);
            self.input.deinit();
        }
        /// at least `want` bytes of the current stream buffered, or the reader at its end
        fn fill(self: *Self, want: usize) !void {
            var tmp: [65536]u8 = undefined;
            while (!self.loaded and self.input.items.len - self.pos < want) {
                const n = try self.rdr.read(&tmp);
                if (n == 0) {
                    self.loaded = true; // end of the reader
                    break;
                }
                self.input.appendSlice(tmp[0..n]) catch return error.OutOfMemory;
            }
        }
        /// The reader is consumed as far as the stream needs it, in doubling steps: EndOfStream while the
        /// reader still has bytes means "read more and decode again" (inflate.zig:283-353 reads as it goes).
        fn decode(self: *Self) !void {
            if (self.out != null) return;
            var want: usize = 65536;
            while (true) {
                try self.fill(want);
                self.decodeBuffered() catch |e| {
                    if (e == error.EndOfStream and !self.loaded) {
                        want = 2 * @max(want, self.input.items.len - self.pos);
                        continue;
                    }
                    return e;
                };
                return;
            }
        }
        fn decodeBuffered(self: *Self) !void {
            const h = try engine();
            const data = self.input.items[self.pos..];
            var cap: usize = @max(@as(usize, 1) << 16, data.len * 8);
            const dummy = [1]u8{0};
            while (true) {
                const buf = gpa.alloc(u8, cap + 8) catch return error.OutOfMemory;
                const in_off = [2]u64{ 0, data.len };
                const out_off = [2]u64{ 0, cap };
                var lens = [1]u64{0};
                var sts = [1]i32{0};
                var cons = [1]u64{0};
                const rc = flate_hip_decompress_batch(h, if (data.len == 0) &dummy else data.ptr, &in_off, 1, container, 0, buf.ptr, &out_off, &lens, &sts, &cons, 0);
                if (rc != 0) {
                    gpa.free(buf);
                    return error.DeviceError;
                }
                if (sts[0] == 100 and cap < (@as(usize, 1) << 36)) { // OutputTooSmall: the stream expands more
                    gpa.free(buf);
                    cap *= 8;
                    continue;
                }
                statusToError(sts[0]) catch |e| {
                    gpa.free(buf);
                    return e;
                };
                self.out = buf[0..@intCast(lens[0])];
                self.out_cap = cap + 8;
                self.used = @intCast(cons[0]);
                return;
            }
        }
        /// inflate.zig:326-336: up to `limit` bytes (0 = up to 64 KiB), empty slice at the end of the stream
        pub fn get(self: *Self, limit: usize) ![]const u8 {
            try self.decode();
            const o = self.out.?;
            const n = @min(o.len - self.rp, if (limit == 0) @as(usize, 65536) else limit);
            const s = o[self.rp .. self.rp + n];
            self.rp += n;
            if (n == 0) self.ended = true;
            return s;
        }
        /// inflate.zig:315-319
        pub fn next(self: *Self) !?[]const u8 {
            const s = try self.get(0);
            return if (s.len == 0) null else s;
        }
        /// inflate.zig:343-347
        pub fn read(self: *Self, buffer: []u8) !usize {
            const s = try self.get(buffer.len);
            @memcpy(buffer[0..s.len], s);
            return s.len;
        }
        /// inflate.zig:349-351
        pub fn reader(self: *Self) Reader {
            return .{ .context = self };
        }
        /// inflate.zig:292-296
        pub fn decompress(self: *Self, w: anytype) !void {
            while (try self.next()) |buf| try w.writeAll(buf);
        }
        /// inflate.zig:301-309: go on with the next stream of the same reader (concatenated members)
        pub fn reset(self: *Self) void {
            self.pos += self.used;
            if (self.out) |o| gpa.free(o.ptr[0..self.out_cap]);
            self.out = null;
            self.rp = 0;
            self.ended = false;
        }
        /// inflate.zig:283-288
        pub fn setReader(self: *Self, new_reader: ReaderType) void {
            self.rdr = new_reader;
            self.input.clearRetainingCapacity();
            self.loaded = false;
            self.pos = 0;
            self.reset();
            self.pos = 0;
        }
    };
}

/// The API of one of the reference's three modules: container 0 = src/flate.zig (raw deflate),
/// 1 = src/gzip.zig, 2 = src/zlib.zig.
pub fn Module(comptime container: c_int) type {
    return struct {
        pub const Options = DeflateOptions;
        pub const Level = DeflateLevel;

        /// flate.zig:10-12
        pub fn decompress(reader: anytype, writer: anytype) !void {
            var d = decompressor(reader);
            defer d.deinit();
            try d.decompress(writer);
        }
        /// flate.zig:15-17
        pub fn Decompressor(comptime ReaderType: type) type {
            return DecompressorImpl(container, ReaderType);
        }
        /// flate.zig:20-22
        pub fn decompressor(reader: anytype) Decompressor(@TypeOf(reader)) {
            return Decompressor(@TypeOf(reader)).init(reader);
        }
        /// flate.zig:28-30
        pub fn compress(reader: anytype, writer: anytype, options: Options) !void {
            var c = try compressor(writer, options);
            defer c.deinit();
            try c.compress(reader);
            try c.finish();
        }
        /// flate.zig:33-35
        pub fn Compressor(comptime WriterType: type) type {
            return CompressorImpl(container, WriterType);
        }
        /// flate.zig:38-40
        pub fn compressor(writer: anytype, options: Options) !Compressor(@TypeOf(writer)) {
            return Compressor(@TypeOf(writer)).init(writer, modeOf(options.level));
        }
        /// flate.zig:44-56
        pub const huffman = Simple(mode_huffman);
        /// flate.zig:59-71
        pub const store = Simple(mode_store);

        fn Simple(comptime mode: c_int) type {
            return struct {
                pub fn compress(reader: anytype, writer: anytype) !void {
                    var c = try @This().compressor(writer);
                    defer c.deinit();
                    try c.compress(reader);
                    try c.finish();
                }
                pub fn Compressor(comptime WriterType: type) type {
                    return CompressorImpl(container, WriterType);
                }
                pub fn compressor(writer: anytype) !@This().Compressor(@TypeOf(writer)) {
                    return @This().Compressor(@TypeOf(writer)).init(writer, mode);
                }
            };
        }
    };
}
