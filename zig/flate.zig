//! Drop-in for ianic/flate's src/flate.zig (raw deflate) on the MI355X engine.
pub usingnamespace @import("flate_hip.zig").Module(0);
