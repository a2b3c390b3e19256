/*
 * oracle/binseq_oracle.h -- TEST INFRASTRUCTURE ONLY (see em_oracle.h).
 *
 * Plain-C restatement of the per-bin sequence statistics the reference writes into the last six
 * columns of its `-f` table when a genome is given with `-b` (the reference's
 * "bias" code computes nothing else: src/bias.cpp is comments only).
 *   src/alignments.cpp:1622-1636   which statistics, in which order
 *   include/isoform.h:173-182      the bin's sequence = its segments' bases, concatenated
 *   src/fasta.cpp:195-200          fetchSeq: 1-based start, bytes as they are in the FASTA
 *   include/kmer.h:14-135          SortedKmer / Entropy / GCRatio / HighGCStrech / ToDna / ToDna2
 * The reference's asserts are live in its release build (CMakeLists.txt:84 has no -DNDEBUG): a bin
 * of 40 bases or fewer aborts it (kmer.h:82 `assert(w < len)`), fewer than 7 at kmer.h:20.  The
 * functions below continue naturally there (no window -> flag 0; fewer than k bases -> entropy 0).
 */
#ifndef SB_BINSEQ_ORACLE_H_
#define SB_BINSEQ_ORACLE_H_
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* Kmer<string>::GCRatio, kmer.h:67-76 (ToDna2 :91-104: C c G g and the bytes 1, 2 count). */
double sbo_gc_ratio(const uint8_t *seq, int64_t len);

/* Kmer<string>::Entropy(seq, k), kmer.h:16-65: Shannon entropy (natural log) of the k-mer
 * spectrum; every byte that is not ACGTacgt codes as A (ToDna :106-124).                   */
double sbo_kmer_entropy(const uint8_t *seq, int64_t len, int k);

/* Kmer<string>::HighGCStrech(b, e, w, cutoff), kmer.h:78-88: some window of w bases has a
 * GC ratio above the cutoff.                                                                */
int sbo_high_gc_stretch(const uint8_t *seq, int64_t len, int w, double cutoff);

/* The six columns for every bin of a batch: genome[0] is base `genome_start` (1-based) of the
 * chromosome; bin b is the concatenation of segments seg_off[b]..seg_off[b+1]-1 (closed
 * coordinates, in std::set order).  gc, entropy: doubles; flags bit 0..3 = stretch (20, 0.8),
 * (20, 0.9), (40, 0.8), (40, 0.9) -- alignments.cpp:1626-1629.                              */
void sbo_binseq_batch(const uint8_t *genome, int64_t genome_start, int64_t n_bins, const int64_t *seg_off,
                      const uint32_t *seg_left, const uint32_t *seg_right, double *gc, double *entropy,
                      uint8_t *flags);

#ifdef __cplusplus
}
#endif
#endif
