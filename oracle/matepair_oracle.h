/* oracle/matepair_oracle.h -- TEST INFRASTRUCTURE ONLY (see matepair_oracle.c). */
#ifndef SBO_MATEPAIR_ORACLE_H
#define SBO_MATEPAIR_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
/* One cluster's alignment records, in arrival order -> its read pairs in the order they are completed.
 * flags: bit 0 reverse strand, bit 1 partner on another reference, bits 2-3 XS strand (0 unknown, 1 +, 2 -).
 * Out, per pair: left_rec / right_rec (record indices, -1: no such mate), mass; counts[4] = complete pairs, single
 * reads, refused records, records that never found their mate.  Returns the number of pairs.                   */
int sbo_pair_mates(int n_reads, const uint64_t *read_id, const int64_t *block_off, const uint32_t *block_left,
                   const uint32_t *block_right, const uint32_t *partner_pos, const uint8_t *flags, const int32_t *nh,
                   int32_t *left_rec, int32_t *right_rec, double *mass, int32_t counts[4]);
/* Sample::nextClusterRefDemand's pass (src/alignments.cpp:1145-1187), literally: read_cluster[i] = the cluster record
 * i joins or -1; off[k] = the record the pass stands at when cluster k begins (off[n_clusters]: where it ends).
 * r_xs: 0 unknown, 1 +, 2 -.                                                                                  */
void sbo_assign_reads(int n_clusters, const int32_t *c_ref, const uint32_t *c_left, const uint32_t *c_right, const uint8_t *c_strand,
                      int64_t n_reads, const int32_t *r_ref, const uint32_t *r_left, const uint32_t *r_right, const uint8_t *r_xs,
                      int32_t *read_cluster, int64_t *off);
#ifdef __cplusplus
}
#endif
#endif
