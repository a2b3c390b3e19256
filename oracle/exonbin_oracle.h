/*
 * oracle/exonbin_oracle.h -- TEST INFRASTRUCTURE ONLY (see em_oracle.h).
 *
 * Plain-C restatement of the integer part of the reference's exon-bin assignment
 * (SURVEY.md 8(a) A5): is a read compatible with an isoform, and which disjoint exon
 * segments does it touch.  Features are flat: code 0 MATCH / 1 INTRON / 2 GAP (Match_t,
 * include/contig.h:26-31), closed coordinates left..right.
 */
#ifndef SB_EXONBIN_ORACLE_H_
#define SB_EXONBIN_ORACLE_H_
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* Contig::is_compatible(read, isoform), src/contig.cpp:547-599.  The isoform is given by its
 * exons (sorted, closed); its introns are the gaps between consecutive exons, which is what
 * its _genomic_feats hold at odd positions.                                               */
int sbo_is_compatible(int n_feat, const uint8_t *code, const uint32_t *left, const uint32_t *right,
                      int n_exon, const uint32_t *exon_left, const uint32_t *exon_right);

/* LocusContext::overlap_exons, src/estimate.cpp:115-131: key_out[k] = 1 iff some MATCH block of
 * the read overlaps segment k (GenomicFeature::overlaps, src/contig.cpp:98-102).          */
void sbo_overlap_key(int n_feat, const uint8_t *code, const uint32_t *left, const uint32_t *right,
                     int n_seg, const uint32_t *seg_left, const uint32_t *seg_right, uint8_t *key_out);

/* The two tests for every hit of a batch, results as the bit words the GPU kernel writes
 * (include/sbgpu.h, sbgpu_exonbin_device): compat[h*cw + w] bit b <=> hit h compatible with
 * isoform 32*w+b of its locus; key[h*kw + w] bit b <=> hit h overlaps segment 32*w+b.
 * The loop nest is LocusContext::assign_exon_bin's (src/estimate.cpp:135-198).            */
void sbo_exonbin_batch(const int64_t *iso_off, const int64_t *exon_off, const uint32_t *exon_left,
                       const uint32_t *exon_right, const int64_t *seg_off, const uint32_t *seg_left,
                       const uint32_t *seg_right, int64_t n_hits, const int32_t *hit_locus,
                       const int64_t *feat_off, const uint8_t *feat_code, const uint32_t *feat_left,
                       const uint32_t *feat_right, int32_t compat_words, int32_t key_words,
                       uint32_t *compat, uint32_t *key);

#ifdef __cplusplus
}
#endif
#endif
