/*
 * oracle/binseq_oracle.c -- TEST INFRASTRUCTURE ONLY.  See binseq_oracle.h.
 * Pinned against the reference's own include/kmer.h (oracle/ref_shim.cpp: ref_kmer_stats) and
 * against the `-f` table of the reference binary run with `-b` (tests/golden/e2e_toy_bias).
 */
#include "binseq_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

static unsigned to_dna(uint8_t c)          /* kmer.h:106-124 */
{
    switch (c) {
    case 'C': case 'c': return 1u;
    case 'G': case 'g': return 2u;
    case 'T': case 't': return 3u;
    default: return 0u;
    }
}

static unsigned to_dna2(uint8_t c)         /* kmer.h:91-104 */
{
    switch (c) {
    case 'C': case 'c': case 'G': case 'g': case 1u: case 2u: return 1u;
    default: return 0u;
    }
}

double sbo_gc_ratio(const uint8_t *seq, int64_t len)
{
    int gc = 0, total = 0;
    for (int64_t i = 0; i < len; ++i) { ++total; gc += (int)to_dna2(seq[i]); }
    return (double)gc / total;
}

static int cmp_u64(const void *a, const void *b)
{
    uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
    return x < y ? -1 : x > y;
}

double sbo_kmer_entropy(const uint8_t *seq, int64_t len, int k)
{
    if (len < k) return 0.0;               /* the reference asserts len > k, kmer.h:20 */
    int64_t total = len - k + 1;
    uint64_t *km = (uint64_t *)malloc((size_t)total * sizeof *km);
    uint64_t mask = (k < 32) ? ((1ull << (2 * k)) - 1) : ~0ull, cur = 0;
    for (int i = 0; i < k; ++i) cur = (cur << 2) | to_dna(seq[i]);      /* :24-31 */
    km[0] = cur;
    for (int64_t i = k; i < len; ++i) {                                 /* :35-41 */
        cur = ((cur << 2) | to_dna(seq[i])) & mask;
        km[i - k + 1] = cur;
    }
    qsort(km, (size_t)total, sizeof *km, cmp_u64);                      /* :42 */
    double counter = 1.0, sum = 0.0;                                    /* :50-64 */
    for (int64_t i = 1; i < total; ++i) {
        if (km[i] != km[i - 1]) {
            double p = counter / total;
            sum -= p * log(p);
            counter = 1.0;
        } else {
            counter += 1.0;
        }
    }
    double p = counter / total;
    sum -= p * log(p);
    free(km);
    return sum;
}

int sbo_high_gc_stretch(const uint8_t *seq, int64_t len, int w, double cutoff)
{
    for (int64_t b = 0; b + w <= len; ++b)                              /* :83-86 */
        if (sbo_gc_ratio(seq + b, w) > cutoff) return 1;
    return 0;
}

void sbo_binseq_batch(const uint8_t *genome, int64_t genome_start, int64_t n_bins, const int64_t *seg_off,
                      const uint32_t *seg_left, const uint32_t *seg_right, double *gc, double *entropy,
                      uint8_t *flags)
{
    static const int W[4] = {20, 20, 40, 40};
    static const double CUT[4] = {0.8, 0.9, 0.8, 0.9};
    for (int64_t b = 0; b < n_bins; ++b) {
        int64_t len = 0;
        for (int64_t s = seg_off[b]; s < seg_off[b + 1]; ++s) len += (int64_t)seg_right[s] - seg_left[s] + 1;
        uint8_t *seq = (uint8_t *)malloc((size_t)(len > 0 ? len : 1));
        int64_t at = 0;
        for (int64_t s = seg_off[b]; s < seg_off[b + 1]; ++s) {         /* isoform.h:177-180 */
            int64_t n = (int64_t)seg_right[s] - seg_left[s] + 1;
            memcpy(seq + at, genome + ((int64_t)seg_left[s] - genome_start), (size_t)n);
            at += n;
        }
        gc[b] = sbo_gc_ratio(seq, len);
        entropy[b] = sbo_kmer_entropy(seq, len, 6);
        uint8_t f = 0;
        for (int q = 0; q < 4; ++q) f |= (uint8_t)(sbo_high_gc_stretch(seq, len, W[q], CUT[q]) << q);
        flags[b] = f;
        free(seq);
    }
}
