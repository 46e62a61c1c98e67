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
static uint16_t L4[65536], RK[65536], L6[65536], L8[65536], RS[65536];
static int N, Mpos, good, lazy, nice, chainmax, offset_mode;
static unsigned long long n_calls, n_top, n_step[3], n_skip, n_meas, n_rank, n_iter, n_short;
static int shortcuts = 1;

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
    for (int p = 0; p < N; p++) RS[p] = (p && buf[p - 1] == buf[p]) ? RS[p - 1] : (uint16_t)p;  // start of the run of equal bytes p lies in
    memset(head, 0, sizeof head);
    for (int p = 0; p < N; p++) { if (p < Mpos) { const uint32_t h = hash8(buf + p); L8[p] = head[h]; head[h] = (uint16_t)p; } else L8[p] = 0; }
}
static int lcp(int q, int p, int maxlen) { int i = 0; while (i < maxlen && buf[q + i] == buf[p + i]) i++; return i; }
static int all_b(int q, int n, uint8_t b) { for (int i = 0; i < n; i++) if (buf[q + i] != b) return 0; return 1; }
#define PROBE 8
// offset mode: with a match of at least offset_mode bytes in hand (and p not at a run) the walk follows the chain of p + off,
// off = len + 1 - 8: its members, moved back by off, are the positions that share the LAST 8 of the len + 1 bytes a better candidate must share
#define OFFSET_OF(LEN) ((offset_mode && (LEN) >= offset_mode && (LEN) >= 7 && !prun && !all_b(p + (LEN) + 1 - 8, 8, buf[p + (LEN) + 1 - 8])) ? (LEN) + 1 - 8 : 0)
#define RUNSKIP 512
static int run_skip = 1;
static unsigned long long n_runskip;

static int level_of(int len) { return len < 5 ? 4 : (len < 7 ? 6 : 8); }

// deflate.zig:233-266 with the sparser chains; returns len (0 = none)
static const uint16_t* chain_of(int K) { return K == 4 ? L4 : K == 6 ? L6 : L8; }
static int find_match_impl(int p, int len0, int* dist) {
    n_calls++;
    if (p >= Mpos) return 0;
    const int maxlen = N - p < 258 ? N - p : 258;
    if (len0 > 0 && maxlen <= len0) return 0;  // (the reference walks on and finds nothing)
    const int B = len0 >= good ? chainmax >> 2 : chainmax;
    const int lo = p > 32768 ? p - 32768 : 1;
    int len = len0, found = 0, last = p, K = level_of(len0), cnt = B, off = 0, probe = PROBE;
    int prun = 0;  // bytes equal to the first from p on (capped at maxlen) if at least 4
    if (run_skip) { int r = 1; while (r < maxlen && buf[p + r] == buf[p]) r++; if (r >= 4) prun = r; }
    off = OFFSET_OF(len);
    n_top++; n_iter++;
    int q = chain_of(K)[p + off];
    for (;;) {
        if (q < lo + off) return found;
        if (K == 4 && cnt == 0) return found;
        n_iter++;
        const int qc = q - off;  // the candidate
        if (run_skip && qc >= last && all_b(q, 8, buf[q]) && !(prun && !off && buf[q] == buf[p])) {
            // (a foreign run among what has been looked at already: skipped in one step as well, nothing to count)
            n_runskip++; n_iter++;
            int t = q; while (q - t < RUNSKIP && t != 0 && buf[t - 1] == buf[q]) t--;
            q = chain_of(K)[t]; continue;
        }
        if (qc >= last) { n_skip++; q = chain_of(K)[q]; continue; }  // looked at before the walk changed chains
        n_step[K == 4 ? 0 : K == 6 ? 1 : 2]++;
        if (K == 4) { cnt--; last = qc; } else if (probe) probe--;
        int accept_l = 0, acc = -1;   // candidate to accept (after the rank check on L6 / L8)
        int next_from = qc;           // the walk goes on with the link of this chain member
        if (run_skip && all_b(q, 8, buf[q]) && !(prun && !off && buf[q] == buf[p])) {
            // ---- the chain member lies in a run of one byte, and it is not the case below (p itself at a run of that byte): a
            // collision of the hash with a crowded bucket (zero padding).  No member of that run shares with p + off the bytes
            // the chain stands for: what is known to be run is skipped, and counts as looked at.
            n_runskip++; n_iter++;
            int t = q; while (q - t < RUNSKIP && t != 0 && buf[t - 1] == buf[q]) t--;
            if (K == 4) { const int need = q - t; if (cnt < need) return found; cnt -= need; last = t; } else probe = 0;
            next_from = t - off;
        } else if (prun && !off && all_b(qc, 8, buf[p])) {
            // ---- a run of p's byte b: [s, E) with qc inside.  q' in it matches p over min(prun, E - q') bytes (more only at
            // q* = E - prun), so with c bytes in hand the only positions that can help are U = [max(s, q*), min(qc, max(E - c - 1, q*))]:
            // the walk would take every one of them in turn, each a byte longer than the last; it ends up at the lowest one
            // (or at the first that reaches `nice`), unless the budget or the distance ends it before.
            // (round 6) a walk on L6 / L8 that left the run before this one with nothing accepted asks the ranks before it enters
            // this one: beyond the budget the call ends, as the reference's does (the member lies in p's own bucket)
            if (K != 4 && probe == 0) {
                n_rank++; n_iter++; probe = PROBE;
                if ((int)RK[p] - (int)RK[qc] > B) return found;
            }
            n_runskip++; n_iter++;
            const uint8_t bb = buf[p];
            int leave = 0;  // the walk leaves this run at its start s0
            const int c = len, r = prun, cap = (c > r ? c : r) + 1;
            int d = 8; while (d <= cap && buf[qc + d] == bb) d++;   // bytes b from qc on, counted up to cap + 1
            const int s0 = RS[qc];
            int target = -1, hi = -1;
            if (d <= cap) {
                const int E = qc + d;
                hi = E - c - 1 > E - r ? E - c - 1 : E - r;  // (q* = E - r itself may go on behind the runs: always worth a look)
                if (hi > qc) hi = qc;
                const int lo_u = s0 > E - r ? s0 : E - r;
                if (hi >= lo_u) {
                    target = hi < E - nice ? hi : E - nice;
                    if (target < lo_u) target = lo_u;
                }
            }
            // (below q* every position matches exactly prun bytes: the first one met helps if less than that is in hand)
            if (target < 0 && len < prun && d > prun) { target = qc; hi = qc; }
            if (target < 0) {
                // nothing in this run can help: on below its start; what is skipped counts as looked at
                if (K == 4) { const int need = qc - s0; if (cnt < need) return found; cnt -= need; last = s0; } else probe = shortcuts ? (probe > 2 ? probe - 2 : 0) : 0;
                next_from = s0;
                leave = 1;
            } else {
                if (target < lo) target = lo;            // (beyond the distance nothing is looked at)
                if (K == 4) { if (cnt < qc - target) target = qc - cnt; cnt -= qc - target; }
                if (target > hi) return found;           // budget / distance end among positions that cannot help
                if (K == 4) last = target; else probe = 0;
                n_meas++;
                const int l = lcp(target, p, maxlen);
                if (l > len) { accept_l = l; acc = target; }
                next_from = target;
                if (shortcuts && !(l > len)) {
                    // (round 6) the run's best position did not beat what is in hand: below it every position of the run matches
                    // exactly prun <= len bytes -- on below the run's start at once; what is skipped counts as looked at
                    if (K == 4) { const int need = target - s0; if (cnt < need) return found; cnt -= need; last = s0; } else probe = probe > 2 ? probe - 2 : 0;
                    next_from = s0;
                    leave = 1;
                }
                if (K != 4 && l > len) {
                    n_rank++; n_iter++;
                    const int dt = (int)RK[p] - (int)RK[target];
                    if (dt > B) {
                        // the budget ends inside the run (its members have consecutive ranks): the last one within it
                        const int t2 = target + (dt - B);
                        if (t2 > hi) return found;
                        n_meas++;
                        const int l2 = lcp(t2, p, maxlen);
                        if (l2 > len) { found = l2; *dist = p - t2; }
                        return found;
                    }
                    found = l; *dist = p - target; len = l; last = target;
                    if (l >= nice || l >= maxlen) return found;
                    accept_l = 0;  // (done here)
                    if (level_of(len) != K || OFFSET_OF(len) != off) { K = level_of(len); off = OFFSET_OF(len); probe = PROBE; n_top++; n_iter++; q = chain_of(K)[p + off]; continue; }
                }
            }
            if (shortcuts && leave && !accept_l && K == 8 && prun >= K && s0 >= 2) {
                // (round 6) on L8 with p itself at K = 8 bytes b the chain's next member below s0 is the top of the run before --
                // found in the window when at most 8 other bytes lie between (sparse data: one), no link fetched.  Members of
                // other strings with the same hash are left out: they share fewer than K bytes with p and K - 1 are in hand.
                int e = s0 - 1;                          // (buf[s0 - 1] != b: the run starts at s0)
                while (e > 0 && s0 - e <= 8 && buf[e - 1] != bb) e--;
                if (e >= K && buf[e - 1] == bb && all_b(e - K, K, bb) && e - K >= lo) {
                    n_short++;
                    q = e - K;
                    continue;
                }
            }
        } else {
            const int fo = off ? 0 : (len ? len - 3 : 0);  // filter: four bytes the candidate must share
            if (ld32(buf + qc + fo) == ld32(buf + p + fo)) {
                n_meas++;
                const int l = lcp(qc, p, maxlen);
                if (l >= 4 && l > len) { accept_l = l; acc = qc; }
            } else if (K != 4 && probe == 0 && ld32(buf + qc) == ld32(buf + p)) {
                // a walk on L6 / L8 is not counted down: every PROBE steps a candidate of p's own bucket is asked for its rank
                n_rank++; n_iter++; probe = PROBE;
                if ((int)RK[p] - (int)RK[qc] > B) return found;
            }
        }
        if (accept_l) {
            if (K != 4) { n_rank++; n_iter++; probe = PROBE; if ((int)RK[p] - (int)RK[acc] > B) return found; }
            found = accept_l; *dist = p - acc; len = accept_l; last = acc;
            if (len >= nice || len >= maxlen) return found;
            if (level_of(len) != K || OFFSET_OF(len) != off) { K = level_of(len); off = OFFSET_OF(len); probe = PROBE; n_top++; n_iter++; q = chain_of(K)[p + off]; continue; }
        }
        q = chain_of(K)[next_from + off];
    }
}

static int dbg_compare;
static int ref_find_match(int p, int min_len, int* dist) {
    if (N - p < 4) return 0;
    int len = min_len, found = 0, ch = chainmax;
    if (len >= good) ch >>= 2;
    const int maxlen = N - p < 258 ? N - p : 258;
    int q = L4[p];
    while (q > 0 && ch > 0) {
        if (p - q > 32768) break;
        int l = 0;
        if (!(len > 0 && maxlen <= len)) { if (len == 0 || buf[q + len] == buf[p + len]) { l = lcp(q, p, maxlen); if (l < 4) l = 0; } }
        if (l > len) { found = l; *dist = p - q; len = l; if (l >= nice) break; }
        q = L4[q]; ch--;
    }
    return found;
}
static unsigned long long worst_iter; static int worst_p, worst_len0, worst_chunk, cur_chunk;
static int find_match(int p, int len0, int* dist) {
    if (dbg_compare == 3) {
        const unsigned long long i0 = n_iter;
        const int l = find_match_impl(p, len0, dist);
        if (n_iter - i0 > worst_iter) { worst_iter = n_iter - i0; worst_p = p; worst_len0 = len0; worst_chunk = cur_chunk; }
        return l;
    }
    if (!dbg_compare) return find_match_impl(p, len0, dist);
    int d0 = 0, d1 = 0;
    const int l0 = ref_find_match(p, len0, &d0);   // deflate.zig:233-266 as written
    const int l1 = find_match_impl(p, len0, &d1);
    if (l0 != l1 || (l0 && d0 != d1)) {
        int r = 1; while (r < 258 && buf[p + r] == buf[p]) r++;
        printf("call p=%d len0=%d: reference (%d,%d) sparse chains (%d,%d)  prun %d byte %02x N %d\n", p, len0, l0, d0, l1, d1, r, buf[p], N);
        dbg_compare = 2;
    }
    *dist = d0;
    return l0;
}

int main(int argc, char** argv) {
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 2; }
    const int level = argc > 2 ? atoi(argv[2]) : 6;
    const int nchunks = argc > 3 ? atoi(argv[3]) : 64;
    const size_t chunk = argc > 4 ? (size_t)atol(argv[4]) : 65535;
    offset_mode = argc > 5 ? atoi(argv[5]) : 0;
    run_skip = argc > 6 ? atoi(argv[6]) : 1;
    dbg_compare = argc > 7 ? atoi(argv[7]) : 0;
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
        cur_chunk = c;
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
    printf("run skips %.4f/B\n", (double)n_runskip / total);
    if (dbg_compare == 3) printf("worst call: chunk %d p=%d len0=%d: %llu gather rounds\n", worst_chunk, worst_p, worst_len0, worst_iter);
    return bad != 0;
}
