// CPU model of the round-4 match finder (k_lz_links + k_lz_walk): findMatch (deflate.zig:233-266)
// walks the reference's 4-byte hash chain only while the match in hand is shorter than 5 bytes; from then on it
// walks a SPARSER chain -- the positions that share a hash of 6 (match of 5 or 6 bytes in hand) or 8 bytes (7 and
// more) with the call's position.  A candidate that can still change the result agrees with the position on more
// bytes than the match in hand, so it is on that chain too, in the same order; the candidates left out are exactly
// ones the reference looks at and drops.  What the reference counts down per candidate (`chain`) is checked per
// ACCEPTED candidate instead, with rank[p] = number of earlier positions in p's 4-byte hash bucket: a candidate q
// of p is within the budget iff rank[p] - rank[q] <= chain.
// The model runs the whole lazy-matching automaton with this finder and compares the token list with the
// oracle's (oracle/flate_oracle.c); it also counts what the kernel's cost depends on.  Tooling only.
//
// Build: gcc -O2 -o /tmp/ml_model tools/multilevel_model.c
// Run:   /tmp/ml_model FILE [level=6] [nchunks=64] [chunk=65535] [offset_mode=0]
#include "../oracle/flate_oracle.c"
#include <stdio.h>

static uint8_t buf[65536 + 600];
static uint16_t L4[65536], RK[65536], L6[65536], L8[65536];
static int N, Mpos, good, lazy, nice, chainmax, offset_mode;
static unsigned long long n_calls, n_top, n_step[3], n_skip, n_meas, n_rank, n_iter;

static uint32_t ld32(const uint8_t* b) { uint32_t v; memcpy(&v, b, 4); return v; }
// the hashes of the upper levels (any function would do: the chains only have to contain what matches)
static uint32_t hash6(const uint8_t* b) { return ((ld32(b) * 0x9E3779B1u) ^ ((ld32(b + 4) & 0xffffu) * 0x85EBCA6Bu)) >> 17; }
static uint32_t hash8(const uint8_t* b) { return ((ld32(b) * 0x9E3779B1u) ^ (ld32(b + 4) * 0x85EBCA6Bu)) >> 17; }

static void build_links(void) {
    static uint16_t head[32768], cnt[32768];
    memset(head, 0, sizeof head);
    memset(cnt, 0, sizeof cnt);
    for (int p = 0; p < N; p++) {
        if (p < Mpos) { const uint32_t h = fo_hash4(buf + p); L4[p] = head[h]; RK[p] = cnt[h]++; head[h] = (uint16_t)p; }
        else { L4[p] = 0; RK[p] = 0; }
    }
    memset(head, 0, sizeof head);
    for (int p = 0; p < N; p++) { if (p < Mpos) { const uint32_t h = hash6(buf + p); L6[p] = head[h]; head[h] = (uint16_t)p; } else L6[p] = 0; }
    memset(head, 0, sizeof head);
    for (int p = 0; p < N; p++) { if (p < Mpos) { const uint32_t h = hash8(buf + p); L8[p] = head[h]; head[h] = (uint16_t)p; } else L8[p] = 0; }
}
static int lcp(int q, int p, int maxlen) { int i = 0; while (i < maxlen && buf[q + i] == buf[p + i]) i++; return i; }
#define PROBE 8
static int level_of(int len) { return len < 5 ? 4 : (len < 7 ? 6 : 8); }

// deflate.zig:233-266 with the sparser chains; returns len (0 = none)
static int find_match(int p, int len0, int* dist) {
    n_calls++;
    if (p >= Mpos) return 0;
    const int maxlen = N - p < 258 ? N - p : 258;
    if (len0 > 0 && maxlen <= len0) return 0;  // (the reference walks on and finds nothing)
    const int B = len0 >= good ? chainmax >> 2 : chainmax;
    const int lo = p > 32768 ? p - 32768 : 1;
    int len = len0, found = 0, last = p, K = level_of(len0), cnt = B, off = 0, probe = PROBE;
    for (;;) {  // one level (and, in offset mode, one offset) per trip
        off = (offset_mode && K > 4) ? len + 1 - K : 0;
        probe = PROBE;
        n_top++; n_iter++;
        int q = (K == 4 ? L4 : K == 6 ? L6 : L8)[p + off];
        int switched = 0;
        for (;;) {
            if (q < lo + off) break;
            if (K == 4 && cnt == 0) break;
            n_iter++;
            const int qc = q - off;  // the candidate
            if (qc >= last) n_skip++;
            else {
                n_step[K == 4 ? 0 : K == 6 ? 1 : 2]++;
                if (K != 4 && probe) probe--;
                const int fo = off ? 0 : (len ? len - 3 : 0);  // filter: four bytes the candidate must share
                if (ld32(buf + qc + fo) == ld32(buf + p + fo)) {
                    n_meas++;
                    const int l = lcp(qc, p, maxlen);
                    if (l >= 4 && l > len) {
                        if (K != 4) { n_rank++; n_iter++; if ((int)RK[p] - (int)RK[qc] > B) return found; }
                        found = l; *dist = p - qc; len = l; last = qc;
                        if (l >= nice || l >= maxlen) return found;
                        if (level_of(len) != K || off != ((offset_mode && K > 4) ? len + 1 - K : 0)) { K = level_of(len); switched = 1; break; }
                    }
                }
                else if (K != 4 && probe == 0 && ld32(buf + qc) == ld32(buf + p)) {
                    // a walk on L6 / L8 is not counted down: every PROBE steps a candidate of p's own bucket is asked for its rank
                    n_rank++; n_iter++; probe = PROBE;
                    if ((int)RK[p] - (int)RK[qc] > B) return found;
                }
                if (K == 4) last = qc;
            }
            if (K == 4) cnt--;
            q = (K == 4 ? L4 : K == 6 ? L6 : L8)[q];
        }
        if (!switched) return found;
    }
}

int main(int argc, char** argv) {
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 2; }
    const int level = argc > 2 ? atoi(argv[2]) : 6;
    const int nchunks = argc > 3 ? atoi(argv[3]) : 64;
    const size_t chunk = argc > 4 ? (size_t)atol(argv[4]) : 65535;
    offset_mode = argc > 5 ? atoi(argv[5]) : 0;
    const level_args_t la = level_args(level);
    good = la.good; lazy = la.lazy; nice = la.nice; chainmax = la.chain;
    static uint32_t toks[65536 + 16], mine[65536 + 16];
    unsigned long long bad = 0, total = 0;
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
        int a = 0;
        while (a < N) {  // deflate.zig:154-205
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
        if (k != nt || memcmp(mine, toks, nt * 4)) {
            bad++;
            size_t i = 0;
            while (i < k && i < nt && mine[i] == toks[i]) i++;
            printf("chunk %d: MISMATCH at token %zu (mine %zu tokens, oracle %zu)\n", c, i, k, nt);
        }
    }
    printf("level %d offset %d: %d chunks, %llu bytes, mismatching chunks: %llu\n", level, offset_mode, c, total, bad);
    printf("per byte: calls %.3f  tops %.3f  steps L4 %.3f L6 %.3f L8 %.3f  skipped %.3f  measures %.3f  rank checks %.3f  gather rounds %.3f\n",
           (double)n_calls / total, (double)n_top / total, (double)n_step[0] / total, (double)n_step[1] / total, (double)n_step[2] / total,
           (double)n_skip / total, (double)n_meas / total, (double)n_rank / total, (double)n_iter / total);
    return bad != 0;
}
