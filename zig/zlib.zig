//! Drop-in for ianic/flate's src/zlib.zig on the MI355X engine.
pub usingnamespace @import("flate_hip.zig").Module(2);
