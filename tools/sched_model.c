// Round 6: how k_lz_parse's speculative phase would schedule under other hand-outs of the segments (tooling only).
// The unit of time is one trip of a wave through its loop (a burst of at most 18 chain steps + one visit of the slow block):
// a lane's call costs one trip per piece of at most 18 steps between two candidates that pass the filter, at least one.
// A wave is as slow as its slowest lane; the workgroup as its slowest wave.  The model parses every segment from its start
// (SPEC) and from the exit of the segment before it until it meets SPEC's anchors (FIX), like the kernel's round 0.
//
// Build: gcc -O2 -o /tmp/sched_model tools/sched_model.c      Run: /tmp/sched_model FILE [level=6] [nchunks=64]
#include "../oracle/flate_oracle.c"
#include <stdio.h>

static uint8_t buf[65536 + 600];
static uint16_t prv[65536];
static int N, good, lazy, nice, chainmax;
#define BURST 18
static unsigned long long g_trips;  // trips of the current lane

static int lcp(int q, int p, int maxlen) {
    int i = 0;
    while (i < maxlen && buf[q + i] == buf[p + i]) i++;
    return i;
}
static int find_match(int p, int min_len, int* dist) {
    unsigned trips = 0, since = 0;
    int found = 0;
    if (N - p >= 4) {
        int len = min_len, ch = chainmax;
        if (len >= good) ch >>= 2;
        const int maxlen = N - p < 258 ? N - p : 258;
        int q = prv[p];
        while (q > 0 && ch > 0 && maxlen > len) {
            if (p - q > 32768) break;
            since++;
            const int off = len ? len - 3 : 0;
            if (memcmp(buf + q + off, buf + p + off, 4) == 0) {  // the kernel's filter: the four bytes that end at offset len
                trips += (since + BURST - 1) / BURST;
                since = 0;
                int l = lcp(q, p, maxlen);
                if (l < 4) l = 0;
                if (l > len) {
                    found = l;
                    *dist = p - q;
                    len = l;
                    if (l >= nice) break;
                }
            }
            q = prv[q];
            ch--;
        }
        trips += (since + BURST - 1) / BURST;
    }
    g_trips += trips ? trips : 1;
    return found;
}
static int eval_anchor(int a) {  // next anchor
    int dist = 0;
    int len = find_match(a, 0, &dist);
    if (!len) return a + 1;
    int j = 0;
    while (len < lazy) {
        int d2 = 0;
        const int l2 = find_match(a + j + 1, len, &d2);
        if (!l2) break;
        len = l2;
        j++;
    }
    return a + j + len;
}

#define MAXSEG 8192
static unsigned spec[MAXSEG], fixc[MAXSEG];
static int segX[MAXSEG];
static uint8_t isanchor[65536 + 600];

// SPEC and FIX trips of the segments of size S over targets [t0, t1): spec[k], fixc[k]; returns the number of segments
static int costs(int t0, int t1, int S) {
    const int end = t1 < N ? t1 : N;
    int n = 0;
    memset(isanchor, 0, sizeof isanchor);
    for (int s = t0; s < end; s += S, n++) {
        const int e = s + S < end ? s + S : end;
        g_trips = 0;
        int a = s;
        while (a < e) {
            isanchor[a] = 1;
            a = eval_anchor(a);
        }
        segX[n] = a;
        spec[n] = (unsigned)g_trips;
    }
    for (int k = 0; k < n; k++) {
        fixc[k] = 0;
        if (!k) continue;
        const int s = t0 + k * S, e = s + S < end ? s + S : end;
        int a = segX[k - 1];
        if (a < s || a >= e) continue;
        g_trips = 0;
        // (anchors of this segment's own parse only: isanchor holds every segment's, and a is inside this one)
        while (a < e && !isanchor[a]) a = eval_anchor(a);
        fixc[k] = (unsigned)g_trips + 1;  // + the trip in which the exit of the lane before is seen
    }
    return n;
}

typedef struct { double wave_mean, wg_max, lane_mean; } res_t;
// static hand-out: wave w takes segments [64 w, 64 w + 64), a lane one segment
static void sched_static(int n, res_t* r) {
    double sum = 0, mx = 0, lsum = 0;
    int nw = 0;
    for (int w = 0; w < n; w += 64, nw++) {
        unsigned m = 0;
        for (int k = w; k < n && k < w + 64; k++) {
            const unsigned c = spec[k] + fixc[k];
            lsum += c;
            if (c > m) m = c;
        }
        sum += m;
        if (m > mx) mx = m;
    }
    r->wave_mean += sum / nw;
    r->wg_max += mx;
    r->lane_mean += lsum / n;
}
// dynamic inside a wave: wave w owns segments [G w, G w + G) (G = 64 R), its lanes take them in order as they come free;
// FIX afterwards, R segments a lane
static void sched_dyn_wave(int n, int R, res_t* r) {
    const int G = 64 * R;
    double sum = 0, mx = 0;
    int nw = 0;
    for (int w = 0; w < n; w += G, nw++) {
        unsigned lane[64] = {0};
        for (int k = w; k < n && k < w + G; k++) {
            int best = 0;
            for (int l = 1; l < 64; l++) if (lane[l] < lane[best]) best = l;
            lane[best] += spec[k];
        }
        unsigned m = 0;
        for (int l = 0; l < 64; l++) if (lane[l] > m) m = lane[l];
        unsigned f = 0;
        for (int l = 0; l < 64; l++) {
            unsigned c = 0;
            for (int j = 0; j < R; j++) if (w + l * R + j < n) c += fixc[w + l * R + j];
            if (c > f) f = c;
        }
        m += f;
        sum += m;
        if (m > mx) mx = m;
    }
    r->wave_mean += sum / nw;
    r->wg_max += mx;
}
// dynamic over the workgroup: blocks of 64 segments (a lane one segment) go to the 16 waves as they come free
static void sched_dyn_blocks(int n, res_t* r) {
    double wave[16] = {0};
    for (int w = 0; w < n; w += 64) {
        unsigned m = 0;
        for (int k = w; k < n && k < w + 64; k++) if (spec[k] + fixc[k] > m) m = spec[k] + fixc[k];
        int best = 0;
        for (int i = 1; i < 16; i++) if (wave[i] < wave[best]) best = i;
        wave[best] += m;
    }
    double mx = 0, sum = 0;
    for (int i = 0; i < 16; i++) { sum += wave[i]; if (wave[i] > mx) mx = wave[i]; }
    r->wave_mean += sum / 16;
    r->wg_max += mx;
}

// helpers (round 6, the item that is left): a lane that is done takes over the second half of what the lane with the most left
// still has to parse; `levels` = how often one segment's rest may be split (1: one helper a segment); OV trips for the helper's
// own synchronisation and the hand-over
static void sched_help(int n, int levels, int OV, int T, res_t* r) {
    double sum = 0, mx = 0;
    int nw = 0;
    for (int w = 0; w < n; w += 64, nw++) {
        int rem[64], help[64], busy[64];
        int cnt = 0;
        for (int k = w; k < n && k < w + 64; k++, cnt++) { rem[cnt] = (int)(spec[k] + fixc[k]); help[cnt] = 0; busy[cnt] = 0; }
        int t = 0;
        for (;; t++) {
            int alive = 0;
            for (int l = 0; l < cnt; l++) alive |= rem[l] > 0 || busy[l] > 0;
            if (!alive) break;
            for (int b = 0; b < cnt; b++) {
                if (rem[b] > 0 || busy[b] > 0) continue;
                int best = -1;
                for (int a = 0; a < cnt; a++) if (rem[a] >= T && help[a] < levels && (best < 0 || rem[a] > rem[best])) best = a;
                if (best < 0) break;
                const int half = rem[best] / 2;
                rem[best] = rem[best] - half + OV;  // (the owner waits for the helper in the end: the later of the two)
                busy[b] = half + OV;
                if (busy[b] > rem[best]) rem[best] = busy[b];
                help[best]++;
            }
            for (int l = 0; l < cnt; l++) { if (rem[l] > 0) rem[l]--; if (busy[l] > 0) busy[l]--; }
        }
        sum += t;
        if (t > mx) mx = t;
    }
    r->wave_mean += sum / nw;
    r->wg_max += mx;
}

int main(int argc, char** argv) {
    FILE* f = fopen(argv[1], "rb");
    const int level = argc > 2 ? atoi(argv[2]) : 6;
    const int nchunks = argc > 3 ? atoi(argv[3]) : 64;
    const level_args_t la = level_args(level);
    good = la.good; lazy = la.lazy; nice = la.nice; chainmax = la.chain;
    static uint16_t head[32768];
    res_t h1 = {0}, h2 = {0}, h3 = {0}, hb1 = {0}, hb3 = {0}, st48 = {0}, st32 = {0}, st24 = {0}, dw24 = {0}, dw16 = {0}, dw12 = {0}, db24 = {0}, db16 = {0}, db32 = {0}, bst32 = {0}, bdw16 = {0}, bdw8 = {0}, bst16 = {0};
    int c;
    for (c = 0; c < nchunks; c++) {
        N = (int)fread(buf, 1, 65535, f);
        if (N <= 0) break;
        memset(buf + N, 0, 600);
        memset(head, 0, sizeof head);
        for (int p = 0; p < N; p++) {
            if (p + 4 <= N) { const uint32_t h = fo_hash4(buf + p); prv[p] = head[h]; head[h] = (uint16_t)p; }
            else prv[p] = 0;
        }
        int n;
        n = costs(0, 49152, 48); sched_static(n, &st48); sched_help(n, 1, 3, 8, &h1); sched_help(n, 2, 3, 8, &h2); sched_help(n, 3, 3, 8, &h3);
        n = costs(0, 49152, 32); sched_static(n, &st32); sched_dyn_blocks(n, &db32);
        n = costs(0, 49152, 24); sched_static(n, &st24); sched_dyn_wave(n, 2, &dw24); sched_dyn_blocks(n, &db24);
        n = costs(0, 49152, 16); sched_dyn_wave(n, 3, &dw16); sched_dyn_blocks(n, &db16);
        n = costs(0, 49152, 12); sched_dyn_wave(n, 4, &dw12);
        n = costs(49152, 65536, 32); sched_static(n, &bst32); sched_help(n, 1, 3, 8, &hb1); sched_help(n, 3, 3, 8, &hb3);
        n = costs(49152, 65536, 16); sched_static(n, &bst16); sched_dyn_wave(n, 2, &bdw16);
        n = costs(49152, 65536, 8); sched_dyn_wave(n, 4, &bdw8);
    }
    printf("level %d, %d chunks; trips of a wave's loop, mean over the chunks (sub-pass A, targets [0, 49152))\n", level, c);
#define ROW(name, r) printf("  %-58s mean wave %7.1f   slowest wave %7.1f   mean lane %6.1f\n", name, r.wave_mean / c, r.wg_max / c, r.lane_mean / c)
    ROW("static, 48-byte segments, 16 waves (today)", st48);
    ROW("... + a lane that is done takes half of the longest rest (once a segment, 3 trips of overhead)", h1);
    ROW("... twice a segment", h2);
    ROW("... three times a segment", h3);
    ROW("static, 32-byte segments, 24 blocks", st32);
    ROW("static, 24-byte segments, 32 blocks", st24);
    ROW("blocks of 64 x 32 bytes to the waves as they come free", db32);
    ROW("blocks of 64 x 24 bytes to the waves as they come free", db24);
    ROW("blocks of 64 x 16 bytes to the waves as they come free", db16);
    ROW("a wave's 3072 bytes as 128 x 24, lanes take them in order", dw24);
    ROW("a wave's 3072 bytes as 192 x 16, lanes take them in order", dw16);
    ROW("a wave's 3072 bytes as 256 x 12, lanes take them in order", dw12);
    printf("sub-pass B, targets [49152, 65536)\n");
    ROW("static, 32-byte segments, 8 waves (today)", bst32);
    ROW("... + helpers, once a segment", hb1);
    ROW("... + helpers, three times a segment", hb3);
    ROW("static, 16-byte segments, 16 waves", bst16);
    ROW("16 waves, a wave's 1024 bytes as 128 x 16, lanes take them", bdw16);
    ROW("16 waves, a wave's 1024 bytes as 256 x 8, lanes take them", bdw8);
    return 0;
}
