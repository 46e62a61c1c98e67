// CPU model of the round-3 tokenizer (k_lz_chain + k_lz_parse): the lazy-matching automaton
// (deflate.zig:154-205) run speculatively from every segment start, then stitched into the one true
// parse.  The model executes the same phases as the kernel, lane by lane, and checks the token
// list against the oracle's (oracle/flate_oracle.c).  It also reports the quantities the kernel's
// cost depends on (calls, candidate steps, rounds, load balance across lanes).  Tooling only.
//
// Build: gcc -O2 -o /tmp/spec_model tools/spec_parse_model.c
// Run:   /tmp/spec_model FILE [level=6] [nchunks=64] [S=64] [chunk=65535]
#include "../oracle/flate_oracle.c"
#include <stdio.h>

static uint8_t buf[65536 + 600];
static uint16_t prv[65536];
static int N, good, lazy, nice, chainmax;
#define DESC_LIT 0x40000000u
static uint32_t desc_arr[65536];

// per evaluation counters
static unsigned long long n_calls, n_steps, n_measure, n_ext;

static int lcp(int q, int p, int maxlen) {
    int i = 0;
    while (i < maxlen && buf[q + i] == buf[p + i]) i++;
    return i;
}
// deflate.zig:233-266 over the chain array; returns len (0 = none)
static int find_match(int p, int min_len, int* dist) {
    n_calls++;
    if (N - p < 4) return 0;
    int len = min_len, found = 0, ch = chainmax;
    if (len >= good) ch >>= 2;
    const int maxlen = N - p < 258 ? N - p : 258;
    int q = prv[p];
    while (q > 0 && ch > 0) {
        if (p - q > 32768) break;
        n_steps++;
        int l = 0;
        if (!(len > 0 && maxlen <= len)) {
            if (len == 0 || buf[q + len] == buf[p + len]) {
                n_measure++;
                l = lcp(q, p, maxlen);
                if (l >= 8) n_ext++;
                if (l < 4) l = 0;
            }
        }
        if (l > len) {
            found = l;
            *dist = p - q;
            len = l;
            if (l >= nice) return found;
        }
        q = prv[q];
        ch--;
    }
    return found;
}
// the automaton at an anchor (a position visited with no pending match): descriptor and next anchor
static uint32_t eval_anchor(int a, int* next) {
    int dist = 0;
    int len = find_match(a, 0, &dist);
    if (!len) {
        *next = a + 1;
        return DESC_LIT;
    }
    int j = 0;
    while (len < lazy) {  // deflate.zig:171-178
        int d2 = 0;
        const int l2 = find_match(a + j + 1, len, &d2);
        if (!l2) break;
        len = l2;
        dist = d2;
        j++;
    }
    *next = a + j + len;
    return 0x80000000u | ((uint32_t)j << 23) | ((uint32_t)(len - 3) << 15) | (uint32_t)(dist - 1);
}

#define MAXSEG 4096
static int S = 64;
static uint64_t segA[MAXSEG], segF[MAXSEG], segTRUE[MAXSEG];
static int segX[MAXSEG], res_entry[MAXSEG], res_exit[MAXSEG], segZ[MAXSEG], entry[MAXSEG];
static unsigned long long lane_steps[MAXSEG];  // cost of the speculative phase per lane (steps + calls)
static unsigned long long st_rounds, st_subpasses, st_fix_segments, st_spec_lane_sum, st_spec_wave_max, st_fix_wave_max, st_maxrounds;

// one sub-pass over targets [t0, t1) entered at the true anchor y0; returns the true exit (first true anchor >= t1 or >= N)
static int sub_pass(int t0, int t1, int y0) {
    const int end = t1 < N ? t1 : N;
    if (y0 >= end) return y0;
    const int m0 = t0 / S, m1 = (end + S - 1) / S;  // segments [m0, m1)
    const int me = y0 / S;
    st_subpasses++;
    // P0: speculative parse of every segment from its start
    for (int m = m0; m < m1; m++) {
        const int s = m * S, e = (s + S < end) ? s + S : end;
        segA[m] = segF[m] = segTRUE[m] = 0;
        res_entry[m] = -1;
        segZ[m] = -1;
        lane_steps[m] = 0;
        if (e <= y0) { segX[m] = -1; continue; }
        int a = m == me ? y0 : s;
        const unsigned long long c0 = n_calls + n_steps + n_measure;
        while (a < e) {
            int nx;
            desc_arr[a] = eval_anchor(a, &nx);
            segA[m] |= 1ull << (a - s);
            a = nx;
        }
        segX[m] = a;
        lane_steps[m] = n_calls + n_steps + n_measure - c0;
        if (m == me) { res_entry[m] = y0; res_exit[m] = a; segZ[m] = y0; }
    }
    for (int w = m0; w < m1; w += 64) {  // load balance of the speculative phase: a wave runs as long as its slowest lane
        unsigned long long mx = 0;
        for (int m = w; m < m1 && m < w + 64; m++) { st_spec_lane_sum += lane_steps[m]; if (lane_steps[m] > mx) mx = lane_steps[m]; }
        st_spec_wave_max += mx * 64;
    }
    // rounds: provisional path from the entry, fix-ups of the segments whose entry is not resolved yet
    unsigned long long rounds = 0;
    int y_exit = 0;
    for (;;) {
        rounds++;
        for (int m = m0; m < m1; m++) entry[m] = -1;
        int y = y0;
        while (y < end) {
            const int m = y / S;
            entry[m] = y;
            y = res_entry[m] >= 0 ? res_exit[m] : segX[m];
        }
        y_exit = y;
        int changed = 0;
        unsigned long long mx = 0;
        for (int m = m0; m < m1; m++) {
            if (entry[m] < 0 || entry[m] == res_entry[m]) continue;
            changed++;
            st_fix_segments++;
            const int s = m * S, e = (s + S < end) ? s + S : end;
            int a = entry[m];
            uint64_t F = 0;
            const unsigned long long c0 = n_calls + n_steps + n_measure;
            while (a < e && !((segA[m] >> (a - s)) & 1)) {
                int nx;
                desc_arr[a] = eval_anchor(a, &nx);
                F |= 1ull << (a - s);
                a = nx;
            }
            const unsigned long long c = n_calls + n_steps + n_measure - c0;
            if (c > mx) mx = c;
            segF[m] = F;
            res_entry[m] = entry[m];
            if (a < e) { segZ[m] = a; res_exit[m] = segX[m]; }
            else { segZ[m] = -1; res_exit[m] = a; }
        }
        st_fix_wave_max += mx * 64 * ((m1 - m0 + 63) / 64);
        if (!changed) break;
    }
    st_rounds += rounds;
    if (rounds > st_maxrounds) st_maxrounds = rounds;
    for (int m = m0; m < m1; m++) {
        if (entry[m] < 0) continue;
        const int s = m * S;
        uint64_t t = segF[m];
        if (segZ[m] >= 0) t |= segA[m] & (~0ull << (segZ[m] - s));
        segTRUE[m] = t;
    }
    return y_exit;
}

int main(int argc, char** argv) {
    FILE* f = fopen(argv[1], "rb");
    const int level = argc > 2 ? atoi(argv[2]) : 6;
    const int nchunks = argc > 3 ? atoi(argv[3]) : 64;
    S = argc > 4 ? atoi(argv[4]) : 64;
    const size_t chunk = argc > 5 ? (size_t)atol(argv[5]) : 65535;
    const level_args_t la = level_args(level);
    good = la.good; lazy = la.lazy; nice = la.nice; chainmax = la.chain;
    static uint32_t toks[65536 + 16], mine[65536 + 16];
    static uint16_t head[32768];
    unsigned long long bad = 0, total = 0, tot_tokens = 0;
    int c;
    for (c = 0; c < nchunks; c++) {
        N = (int)fread(buf, 1, chunk, f);
        if (N <= 0) break;
        memset(buf + N, 0, 600);
        total += N;
        size_t nt = 0;
        fo_tokenize(buf, N, level, toks, 65536 + 16, &nt);
        // k_lz_chain: prev[p] = nearest earlier position with the same hash (Lookup.zig:35-40), 0 = none
        memset(head, 0, sizeof head);
        for (int p = 0; p < N; p++) {
            if (p + 4 <= N) { const uint32_t h = fo_hash4(buf + p); prv[p] = head[h]; head[h] = (uint16_t)p; }
            else prv[p] = 0;
        }
        const int TA = 49152;
        int y = sub_pass(0, TA, 0);
        y = sub_pass(TA, 65536, y);
        // tokens from the true anchors
        size_t k = 0;
        for (int m = 0; m * S < N; m++)
            for (int b = 0; b < S; b++)
                if ((segTRUE[m] >> b) & 1) {
                    const int p = m * S + b;
                    const uint32_t d = desc_arr[p];
                    if (d & DESC_LIT) mine[k++] = FO_TOK_LIT(buf[p]);
                    else {
                        const int j = (d >> 23) & 0xff;
                        for (int x = 0; x < j; x++) mine[k++] = FO_TOK_LIT(buf[p + x]);
                        mine[k++] = (1u << 23) | (d & 0x7fffffu);
                    }
                }
        tot_tokens += nt;
        if (k != nt || memcmp(mine, toks, nt * 4)) {
            bad++;
            size_t i = 0;
            while (i < k && i < nt && mine[i] == toks[i]) i++;
            printf("chunk %d: MISMATCH at token %zu (mine %zu tokens, oracle %zu)\n", c, i, k, nt);
        }
    }
    printf("level %d S %d: %d chunks, %llu bytes, %llu tokens, mismatching chunks: %llu\n", level, S, c, total, tot_tokens, bad);
    printf("calls %.4f/byte  candidate steps %.3f/byte  measures %.3f/byte  (lcp>=8: %.4f/byte)\n", (double)n_calls / total, (double)n_steps / total,
           (double)n_measure / total, (double)n_ext / total);
    printf("sub-passes %llu, rounds/sub-pass %.2f (max %llu), fixed-up segments/sub-pass %.1f\n", st_subpasses, (double)st_rounds / st_subpasses, st_maxrounds,
           (double)st_fix_segments / st_subpasses);
    printf("speculative phase: lane-steps %llu, wave-slots %llu -> lane efficiency %.1f%%; fix-up phase wave-slots %llu (%.1f%% of spec)\n", st_spec_lane_sum,
           st_spec_wave_max, 100.0 * st_spec_lane_sum / st_spec_wave_max, st_fix_wave_max, 100.0 * st_fix_wave_max / st_spec_wave_max);
    return bad != 0;
}
