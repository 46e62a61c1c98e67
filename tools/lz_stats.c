// Counts what the reference's tokenizer does on a file cut into 65535-byte chunks: findMatch calls by
// kind (no pending match / pending < good / pending >= good), candidates walked, their distribution.
// Build: gcc -O2 -DFO_STATS -o /tmp/lz_stats tools/lz_stats.c   (includes the oracle source; tooling only)
#include "../oracle/flate_oracle.c"
#include <stdio.h>
int main(int argc, char** argv) {
    FILE* f = fopen(argv[1], "rb");
    int level = argc > 2 ? atoi(argv[2]) : 6;
    size_t chunk = argc > 3 ? (size_t)atol(argv[3]) : 65535;
    uint8_t* buf = malloc(chunk);
    uint32_t* toks = malloc(4 * (chunk + 16));
    size_t n, total = 0, ntok = 0, nch = 0;
    while ((n = fread(buf, 1, chunk, f)) > 0) {
        size_t nt = 0;
        fo_tokenize(buf, n, level, toks, chunk + 16, &nt);
        total += n; ntok += nt; nch++;
    }
    printf("bytes %zu chunks %zu tokens %zu (%.3f per byte)\n", total, nch, ntok, (double)ntok / total);
    const char* kn[3] = {"min_len=0 (full)", "0<min_len<good (full)", "min_len>=good (quarter)"};
    unsigned long long tc = 0, tk = 0;
    for (int k = 0; k < 3; k++) {
        printf("%-26s calls %llu (%.4f/byte)  cands %llu (%.2f/call, %.3f/byte)\n   hist[<1,<2,<4,...]:", kn[k], fo_stat_calls[k],
               (double)fo_stat_calls[k] / total, fo_stat_cands[k], fo_stat_calls[k] ? (double)fo_stat_cands[k] / fo_stat_calls[k] : 0.0, (double)fo_stat_cands[k] / total);
        for (int b = 0; b < 14; b++) printf(" %llu", fo_stat_hist[k][b]);
        printf("\n");
        tc += fo_stat_calls[k]; tk += fo_stat_cands[k];
    }
    printf("total calls %.4f/byte cands %.3f/byte; per 64KiB chunk: %.0f calls %.0f cands; cands with lcp>=8: %.3f/byte\n", (double)tc / total, (double)tk / total,
           (double)tc / nch, (double)tk / nch, (double)fo_stat_cmp8 / total);
    return 0;
}
