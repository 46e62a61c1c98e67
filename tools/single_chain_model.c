// CPU model for round 5's question: how many chain steps does findMatch (deflate.zig:233-266)
// need when the chain a step follows is ONE sparser chain L_K (positions that share a hash of K bytes) instead of
// the reference's 4-byte chain, and what is left for the reference's own chain L4?
//
// With `len` bytes in hand a candidate that can still change the result shares at least len + 1 bytes with the
// position, so the walk may follow L_K as soon as len >= K - 1 (every such candidate is on it, in the same order).
// Before that ("L4 phase": len < K - 1) it follows the reference's chain.  What the reference counts down per
// candidate (`chain`, a quarter of it from `good` on) becomes a bound on the candidate's POSITION:
// B_ch[p] = the ch-th previous member of p's L4 bucket (0 if there are fewer), so that
// "looked at by the reference" == "q >= max(1, p - 32768, B_ch[p])" -- one value per call instead of a count per step.
//
// The model runs the whole lazy-matching automaton with this finder, compares the token list with the oracle's,
// and prints per input byte: calls, L4-phase steps (and calls that have one), L_K steps, skipped entries, measures;
// and the load per 48-byte segment (mean, slowest of a 1024-segment group).  Tooling only.
//
// Build: gcc -O2 -o /tmp/sc_model tools/single_chain_model.c
// Run:   /tmp/sc_model FILE [level=6] [nchunks=64] [K=6] [hashbits=15] [segment=48] [lkfirst=0]
#include "../oracle/flate_oracle.c"
#include <stdio.h>

static uint8_t buf[65536 + 600];
static uint16_t L4[65536], LK[65536], B1[65536], B2[65536], RK[65536];
static int N, Mpos, good, lazy, nice, chainmax, K, hbits;
static unsigned long long n_calls, n_l4phase, n_l4steps, n_lksteps, n_skip, n_meas, n_fresh, n_lazy, n_l4_by_len[3];
static unsigned long long hist_best[10];  // reference's candidates by bytes in hand: 0, 4, 5, 6, 7, 8-11, 12-15, 16+
static unsigned long long seg_cost[2048], pa_seg[2048], n_pa_steps, n_pa_meas, n_pa_e4steps, n_pa_skip, n_pa_walk;
static double pa_mean, pa_max;

static uint32_t ld32(const uint8_t* b) { uint32_t v; memcpy(&v, b, 4); return v; }
static uint64_t ld64(const uint8_t* b) { uint64_t v; memcpy(&v, b, 8); return v; }
static uint32_t hashK(const uint8_t* b) {
    const uint64_t v = ld64(b) & (K >= 8 ? ~0ull : ((1ull << (8 * K)) - 1));
    return (uint32_t)((v * 0x9E3779B97F4A7C15ull) >> (64 - hbits));
}

static void build_links(void) {
    static uint16_t head[65536];
    static uint16_t* members[32768];
    static uint32_t cnt[32768];
    static uint16_t pool[65536];
    memset(head, 0, sizeof head);
    memset(cnt, 0, sizeof cnt);
    for (int p = 0; p < Mpos; p++) cnt[fo_hash4(buf + p)]++;
    uint32_t off = 0;
    for (int h = 0; h < 32768; h++) { members[h] = pool + off; off += cnt[h]; cnt[h] = 0; }
    for (int p = 0; p < N; p++) {
        L4[p] = 0; B1[p] = 0; B2[p] = 0; RK[p] = 0;
        if (p < Mpos) {
            const uint32_t h = fo_hash4(buf + p);
            L4[p] = head[h]; head[h] = (uint16_t)p;
            const uint32_t r = cnt[h]++;
            RK[p] = (uint16_t)r;
            members[h][r] = (uint16_t)p;
            // the ch-th previous member (1 = the nearest); a member at position 0 is the chain's null anyway
            if (r >= (uint32_t)chainmax) B1[p] = members[h][r - chainmax];
            if (r >= (uint32_t)(chainmax >> 2)) B2[p] = members[h][r - (chainmax >> 2)];
        }
    }
    memset(head, 0, sizeof head);
    for (int p = 0; p < N; p++) { LK[p] = 0; if (p < Mpos) { const uint32_t h = hashK(buf + p); LK[p] = head[h]; head[h] = (uint16_t)p; } }
}
static int lcp(int q, int p, int maxlen) { int i = 0; while (i < maxlen && buf[q + i] == buf[p + i]) i++; return i; }

static double ad_mean[3], ad_max[3]; static unsigned long long ad_n[3]; static int ad_chunks;
#ifndef EST_BASE
#define EST_BASE 1.0
#endif
#ifndef SEG_MIN
#define SEG_MIN 16
#endif
static int cur_seg, cur_a;
static uint32_t pos_cost[65536 + 16];
#define COST_ADD(V) (seg_cost[cur_seg] += (V), pos_cost[cur_a] += (V))
static int find_match_impl(int p, int len0, int* dist) {
    n_calls++;
    COST_ADD(3);  // (a call costs a visit of the slow block: counted as three steps)
    if (len0) n_lazy++; else n_fresh++;
    if (p >= Mpos) return 0;
    const int maxlen = N - p < 258 ? N - p : 258;
    if (len0 > 0 && maxlen <= len0) return 0;
    const int quarter = len0 >= good;
    int lo = p > 32768 ? p - 32768 : 1;
    const int Bp = quarter ? B2[p] : B1[p];
    if (Bp > lo) lo = Bp;  // B is the last one looked at: q >= B
    int len = len0, found = 0, last = p;
    // ---- L4 phase
    if (len < K - 1 && K > 4) {
        n_l4phase++;
        int q = L4[p];
        while (q >= lo) {
            n_l4steps++; COST_ADD(1);
            n_l4_by_len[len == 0 ? 0 : len == 4 ? 1 : 2]++;
            last = q;
            const int fo = len ? len - 3 : 0;
            if (ld32(buf + q + fo) == ld32(buf + p + fo)) {
                n_meas++; COST_ADD(2);
                const int l = lcp(q, p, maxlen);
                if (l >= 4 && l > len) {
                    found = l; *dist = p - q; len = l;
                    if (l >= nice || l >= maxlen) return found;
                    if (len >= K - 1) break;
                }
            }
            q = L4[q];
        }
        if (len < K - 1) return found;  // the chain ended in the L4 phase
    }
    // ---- the sparser chain
    const uint16_t* C = K == 4 ? L4 : LK;
    int q = C[p];
    while (q >= lo) {
        if (q >= last) { n_skip++; COST_ADD(1); q = C[q]; continue; }
        n_lksteps++; COST_ADD(1);
        const int fo = len ? len - 3 : 0;
        if (ld32(buf + q + fo) == ld32(buf + p + fo)) {
            n_meas++; COST_ADD(2);
            const int l = lcp(q, p, maxlen);
            if (l >= 4 && l > len) {
                found = l; *dist = p - q; len = l;
                if (l >= nice || l >= maxlen) return found;
            }
        }
        q = C[q];
    }
    return found;
}


// ---- "LK first": every call walks L_K at once, accepting only candidates longer than max(len0, K - 1); if nothing was
// accepted and len0 < K - 1, the shorter answers come from the heads of the lower chains: H_j = nearest candidate that shares
// at least j bytes, j = K - 1 down to max(len0 + 1, 4).  Here they are found by walking L4 (fallback steps counted).
static int lkfirst;
static unsigned long long n_rkcheck, n_rkcut, n_waste;
static unsigned long long n_fb_calls, n_fb_steps, n_fb_long, n_fb_max, n_fb_found, n_fb_hist[8];
static int find_match_lkfirst(int p, int len0, int* dist) {
    n_calls++;
    COST_ADD(3);
    if (len0) n_lazy++; else n_fresh++;
    if (p >= Mpos) return 0;
    const int maxlen = N - p < 258 ? N - p : 258;
    if (len0 > 0 && maxlen <= len0) return 0;
    const int quarter = len0 >= good;
    int lo = p > 32768 ? p - 32768 : 1;
    const int Bp = quarter ? B2[p] : B1[p];
    const int lo_exact = Bp > lo ? Bp : lo;
    const int ch = quarter ? chainmax >> 2 : chainmax;
    if (lkfirst != 2) lo = lo_exact;
    int len = len0 > K - 1 ? len0 : K - 1, found = 0;
    if (maxlen > len) {
        int q = LK[p];
        while (q >= lo) {
            n_lksteps++; COST_ADD(1);
            const int fo = len - 3;
            if (ld32(buf + q + fo) == ld32(buf + p + fo)) {
                n_meas++; COST_ADD(2);
                const int l = lcp(q, p, maxlen);
                if (l > len) {
                    if (lkfirst == 2 && RK[p] > ch) { n_rkcheck++; if ((int)RK[p] - (int)RK[q] > ch) { n_rkcut++; break; } }
                    found = l; *dist = p - q; len = l;
                    if (l >= nice || l >= maxlen) return found;
                }
            }
            if (q < lo_exact) n_waste++;
            q = LK[q];
        }
    }
    lo = lo_exact;
    if (found || len0 >= K - 1) return found;
    // fallback: the nearest candidate with lcp in (len0, K - 1], the longest first -- i.e. the reference's walk restricted
    // to what L_K cannot see.  Nothing of at least K bytes exists within the bounds.
    n_fb_calls++;
    COST_ADD(3);
    int q = L4[p], steps = 0;
    len = len0;
    while (q >= lo) {
        steps++; n_fb_steps++; COST_ADD(3);
        const int fo = len ? len - 3 : 0;
        if (ld32(buf + q + fo) == ld32(buf + p + fo)) {
            const int l = lcp(q, p, maxlen);
            if (l >= 4 && l > len) { found = l; *dist = p - q; len = l; if (l >= K - 1 || l >= maxlen) break; }
        }
        q = L4[q];
    }
    if (found) n_fb_found++;
    if (steps > 8) n_fb_long += steps;
    if ((unsigned long long)steps > n_fb_max) n_fb_max = steps;
    n_fb_hist[steps == 0 ? 0 : steps == 1 ? 1 : steps == 2 ? 2 : steps <= 4 ? 3 : steps <= 8 ? 4 : steps <= 16 ? 5 : steps <= 64 ? 6 : 7]++;
    return found;
}

static int ref_find_match(int p, int min_len, int* dist) {
    if (N - p < 4) return 0;
    int len = min_len, found = 0, ch = chainmax;
    if (len >= good) ch >>= 2;
    const int maxlen = N - p < 258 ? N - p : 258;
    int q = L4[p];
    while (q > 0 && ch > 0) {
        if (p - q > 32768) break;
        hist_best[len == 0 ? 0 : len < 8 ? len - 3 : len < 12 ? 5 : len < 16 ? 6 : 7]++;
        int l = 0;
        if (!(len > 0 && maxlen <= len)) { if (len == 0 || buf[q + len] == buf[p + len]) { l = lcp(q, p, maxlen); if (l < 4) l = 0; } }
        if (l > len) { found = l; *dist = p - q; len = l; if (l >= nice) break; }
        q = L4[q]; ch--;
    }
    return found;
}
static unsigned long long n_bad_calls;
static int find_match(int p, int len0, int* dist) {
    int d0 = 0, d1 = 0;
    const int l0 = ref_find_match(p, len0, &d0);
    const int l1 = lkfirst ? find_match_lkfirst(p, len0, &d1) : find_match_impl(p, len0, &d1);
    if (l0 != l1 || (l0 && d0 != d1)) {
        if (n_bad_calls++ < 5) printf("call p=%d len0=%d: reference (%d,%d) model (%d,%d)\n", p, len0, l0, d0, l1, d1);
    }
    *dist = d0;
    return l0;
}

int main(int argc, char** argv) {
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 2; }
    const int level = argc > 2 ? atoi(argv[2]) : 6;
    const int nchunks = argc > 3 ? atoi(argv[3]) : 64;
    K = argc > 4 ? atoi(argv[4]) : 6;
    hbits = argc > 5 ? atoi(argv[5]) : 15;
    const int SEG = argc > 6 ? atoi(argv[6]) : 48;
    lkfirst = argc > 7 ? atoi(argv[7]) : 0;
    const size_t chunk = 65535;
    const level_args_t la = level_args(level);
    good = la.good; lazy = la.lazy; nice = la.nice; chainmax = la.chain;
    static uint32_t toks[65536 + 16], mine[65536 + 16];
    unsigned long long bad = 0, total = 0;
    double sum_mean = 0, sum_max = 0, sum_wavemax = 0;
    int c;
    for (c = 0; c < nchunks; c++) {
        N = (int)fread(buf, 1, chunk, f);
        if (N <= 0) break;
        memset(buf + N, 0, 600);
        Mpos = N >= 4 ? N - 3 : 0;
        total += N;
        size_t nt = 0, k = 0;
        fo_tokenize(buf, N, level, toks, 65536 + 16, &nt);
        build_links();
        memset(seg_cost, 0, sizeof seg_cost);
        memset(pos_cost, 0, sizeof pos_cost);
        for (int p = 0; p < Mpos; p++) {  // phase A: E4 / E5 of every position
            int q = L4[p], cn = chainmax, e4 = 0; const int lo = p > 32768 ? p - 32768 : 1; const int maxlen = N - p < 258 ? N - p : 258;
            unsigned st = 0;
            {   // positions whose nearest L_K member is a true K-sharer within the bounds need no fallback: phase B accepts it
                const int q6 = LK[p]; const int lob = B1[p] > lo ? B1[p] : lo;
                if (q6 >= lob && maxlen >= K && lcp(q6, p, maxlen) >= K) { n_pa_skip++; continue; }
                n_pa_walk++;
            }
            while (q >= lo && cn > 0) {
                n_pa_steps++; st++;
                const int fo = e4 ? 1 : 0;
                if (ld32(buf + q + fo) == ld32(buf + p + fo)) { n_pa_meas++; const int l = lcp(q, p, maxlen); if (l >= 4) { if (!e4) n_pa_e4steps += st; e4 = 1; } if (l >= K - 1) break; }
                q = L4[q]; cn--;
            }
            if (!e4) n_pa_e4steps += st;
            pa_seg[p / SEG] += st;
        }
        { unsigned long long mx = 0, sm = 0; const int nseg = (N + SEG - 1) / SEG, G = nseg < 1024 ? nseg : 1024; for (int i = 0; i < G; i++) { sm += pa_seg[i]; if (pa_seg[i] > mx) mx = pa_seg[i]; } pa_mean += (double)sm / G; pa_max += (double)mx; memset(pa_seg, 0, sizeof pa_seg); }
        int a = 0;
        while (a < N) {  // deflate.zig:154-205
            cur_seg = a / SEG; cur_a = a;
            int dist = 0, len = find_match(a, 0, &dist);
            if (!len) { mine[k++] = FO_TOK_LIT(buf[a]); a++; continue; }
            int j = 0;
            while (len < lazy) {
                int d2 = 0;
                const int l2 = find_match(a + j + 1, len, &d2);
                if (!l2) break;
                len = l2; dist = d2; j++;
            }
            for (int x = 0; x < j; x++) mine[k++] = FO_TOK_LIT(buf[a + x]);
            mine[k++] = (1u << 23) | ((uint32_t)(len - 3) << 15) | (uint32_t)(dist - 1);
            a += j + len;
        }
        if (k != nt || memcmp(mine, toks, nt * 4)) bad++;
        // load balance: groups of 1024 segments (sub-pass A of k_lz_parse), waves of 64 consecutive segments
        const int nseg = (N + SEG - 1) / SEG, G = nseg < 1024 ? nseg : 1024;
        unsigned long long mx = 0, sm = 0; double wm = 0;
        for (int w = 0; w < G; w += 64) { unsigned long long m2 = 0; for (int i = w; i < w + 64 && i < G; i++) { sm += seg_cost[i]; if (seg_cost[i] > m2) m2 = seg_cost[i]; } if (m2 > mx) mx = m2; wm += (double)m2; }
        sum_mean += (double)sm / G; sum_max += (double)mx; sum_wavemax += wm / ((G + 63) / 64);
        if (N >= 49152) {
            // Segments of sub-pass A ([0, 49152), 1024 lanes) cut by an ESTIMATE of the load instead of every 48 bytes: a position whose
            // chain link is d bytes away sits in a bucket with about 32768 / d members in the window; a call there walks min(that, chain)
            // candidates.  Segments of 8 .. 64 bytes (the anchors of a segment are one 64-bit mask), at most 1024 of them.
            static double est[49152];
            for (int p = 0; p < 49152; p++) {
                const int d = L4[p] ? p - L4[p] : 0;
                double e = d ? 32768.0 / d : 0.0;
                if (e > chainmax) e = chainmax;
                est[p] = EST_BASE + e;
            }
            for (int mode = 0; mode < 2; mode++) {  // 0: by the estimate, 1: by the true load (the bound of the method)
                double lo_t = 0, hi_t = 1e9;
                int bounds[1100], nb = 0;
                for (int it = 0; it < 60; it++) {
                    const double T = 0.5 * (lo_t + hi_t);
                    nb = 0; int s0 = 0; double acc = 0;
                    for (int p = 0; p < 49152 && nb <= 1024; p++) {
                        acc += mode ? (double)pos_cost[p] + 0.01 : est[p];
                        const int len = p + 1 - s0;
                        if (len >= 64 || (acc >= T && len >= SEG_MIN) || p == 49151) { if (nb < 1100) bounds[nb] = p + 1; nb++; s0 = p + 1; acc = 0; }
                    }
                    if (nb > 1024) lo_t = T; else hi_t = T;
                }
                {   // (with the last feasible T)
                    const double T = hi_t; nb = 0; int s0 = 0; double acc = 0;
                    for (int p = 0; p < 49152; p++) {
                        acc += mode ? (double)pos_cost[p] + 0.01 : est[p];
                        const int len = p + 1 - s0;
                        if (len >= 64 || (acc >= T && len >= SEG_MIN) || p == 49151) { bounds[nb++] = p + 1; s0 = p + 1; acc = 0; }
                    }
                }
                unsigned long long mxa = 0, sma = 0; int s0 = 0; double wmax = 0;
                for (int i = 0; i < nb; i++) {
                    unsigned long long cs = 0;
                    for (int p = s0; p < bounds[i]; p++) cs += pos_cost[p];
                    s0 = bounds[i];
                    sma += cs; if (cs > mxa) mxa = cs;
                }
                (void)wmax;
                ad_mean[mode] += (double)sma / 1024; ad_max[mode] += (double)mxa; ad_n[mode] += nb;
            }
            {   // fixed 48-byte segments over the same range
                unsigned long long mxa = 0, sma = 0;
                for (int i = 0; i < 1024; i++) { unsigned long long cs = 0; for (int p = 48 * i; p < 48 * i + 48; p++) cs += pos_cost[p]; sma += cs; if (cs > mxa) mxa = cs; }
                ad_mean[2] += (double)sma / 1024; ad_max[2] += (double)mxa; ad_n[2] += 1024;
            }
            ad_chunks++;
        }
    }
    printf("level %d K %d hash bits %d: %d chunks, %llu bytes, mismatching chunks %llu, mismatching calls %llu\n", level, K, hbits, c, total, bad, n_bad_calls);
    printf("per byte: calls %.3f (fresh %.3f lazy %.3f)  L4-phase calls %.3f  L4-phase steps %.3f (len 0: %.3f, 4: %.3f, 5+: %.3f)  L%d steps %.3f  skipped %.3f  measures %.3f  all steps %.3f\n",
           (double)n_calls / total, (double)n_fresh / total, (double)n_lazy / total, (double)n_l4phase / total, (double)n_l4steps / total,
           (double)n_l4_by_len[0] / total, (double)n_l4_by_len[1] / total, (double)n_l4_by_len[2] / total, K,
           (double)n_lksteps / total, (double)n_skip / total, (double)n_meas / total, (double)(n_l4steps + n_lksteps + n_skip) / total);
    if (lkfirst) printf("LK first: fallback calls %.3f per byte (found something: %.3f), fallback L4 steps %.3f per byte, in walks of more than 8: %.3f, longest %llu; walks of 0/1/2/3-4/5-8/9-16/17-64/65+ steps: %llu %llu %llu %llu %llu %llu %llu %llu\n",
           (double)n_fb_calls / total, (double)n_fb_found / total, (double)n_fb_steps / total, (double)n_fb_long / total, n_fb_max,
           n_fb_hist[0], n_fb_hist[1], n_fb_hist[2], n_fb_hist[3], n_fb_hist[4], n_fb_hist[5], n_fb_hist[6], n_fb_hist[7]);
    printf("phase A (E4 .. E(K-1) on L4 of every position whose L_K head is not a K-sharer in bounds: %.3f of them per byte): %.3f steps per byte (until E4 alone: %.3f), %.3f measures; per %d-byte segment: mean %.0f, slowest of 1024: %.0f\n", (double)n_pa_walk / total, (double)n_pa_steps / total, (double)n_pa_e4steps / total, (double)n_pa_meas / total, SEG, pa_mean / c, pa_max / c);
    if (lkfirst == 2) printf("rank checks at accepts %.4f per byte (walks cut by them %.4f), steps beyond the budget %.4f per byte\n", (double)n_rkcheck / total, (double)n_rkcut / total, (double)n_waste / total);
    printf("reference's candidates by bytes in hand (per byte): 0: %.3f  4: %.3f  5: %.3f  6: %.3f  7: %.3f  8-11: %.3f  12-15: %.3f  16+: %.3f\n",
           (double)hist_best[0] / total, (double)hist_best[1] / total, (double)hist_best[2] / total, (double)hist_best[3] / total,
           (double)hist_best[4] / total, (double)hist_best[5] / total, (double)hist_best[6] / total, (double)hist_best[7] / total);
    printf("load per %d-byte segment (steps + 3 per call + 2 per measure): mean %.0f, slowest of a wave %.0f (%.2f x), slowest of 1024 %.0f (%.2f x)\n",
           SEG, sum_mean / c, sum_wavemax / c, sum_wavemax / sum_mean, sum_max / c, sum_max / sum_mean);
    if (ad_chunks) {
        const char* nm[3] = {"cut by the estimate", "cut by the true load", "every 48 bytes"};
        for (int m = 0; m < 3; m++) printf("sub-pass A, segments %s: %.0f segments, load per lane: mean %.0f, slowest %.0f (%.2f x)\n", nm[m], (double)ad_n[m] / ad_chunks, ad_mean[m] / ad_chunks, ad_max[m] / ad_chunks, ad_max[m] / ad_mean[m]);
    }
    return bad != 0;
}
