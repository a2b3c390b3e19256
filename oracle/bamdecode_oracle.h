/* oracle/bamdecode_oracle.h -- TEST INFRASTRUCTURE ONLY (see bamdecode_oracle.c). */
#ifndef SB_BAMDECODE_ORACLE_H
#define SB_BAMDECODE_ORACLE_H
#include <stdint.h>

typedef struct {
   int32_t min_intron;  /* kMinIntronLength (src/common.cpp:21; -j)             */
   int32_t max_intron;  /* kMaxIntronLength (src/common.cpp:20; -J)             */
   int32_t unique_only; /* use_only_unique_hits (src/common.cpp:67; --allow-multimapped-hits clears it) */
   int32_t library;     /* 0 unstranded, 1 fr_strand, 2 rf_strand (src/common.cpp:68-69)     */
   int32_t n_ref;       /* references in the header (records with a larger id are refused); <= 0: not checked */
} sbo_bam_opts;

/* why a record is not a ReadHit (0: it is one) -- the order of getHitFromBuf's tests */
enum {
   SBO_BAM_OK = 0,
   SBO_BAM_UNMAPPED = 1,      /* flag 0x4 or no reference (:508)                     */
   SBO_BAM_BAD_REF = 2,       /* reference id not in the header (:531)               */
   SBO_BAM_ZERO_OP = 3,       /* a CIGAR operation of length 0 (:540)                */
   SBO_BAM_OP = 4,            /* an operation the reference does not take (=, X, B, ...) (:585) */
   SBO_BAM_INTRON_LONG = 5,   /* N longer than max_intron (:575)                     */
   SBO_BAM_INTRON_SHORT = 6,  /* N shorter than min_intron (:579)                    */
   SBO_BAM_INDEL = 7,         /* I / D not between two M, or among the first two kept operations (:592-599) */
   SBO_BAM_SHORT = 8,         /* at most one aligned base (:601)                     */
   SBO_BAM_MULTI = 9,         /* NH > 1 or a secondary alignment while unique_only (:670) */
   SBO_BAM_TRUNCATED = 10     /* (not in the reference) the record's own lengths run past its block_size */
};

/* the records' offsets in an uncompressed BAM record stream (each record: int32 block_size + block_size bytes).
 * Returns the number of records, -1 when the stream ends inside a record or `cap` is too small. */
int64_t sbo_bam_index(const uint8_t *bytes, int64_t n_bytes, int64_t *rec_off, int64_t cap);

/* Per record r (rec_off[r] .. rec_off[r + 1]), file order.  Arrays of n (or n + 1 for the CSR offsets); the flattened
 * arrays need as many entries as the records have CIGAR operations in all.
 *   status            the enum above
 *   read_id           FNV-1 of the read name (ReadTable::get_id)
 *   ref, left, right  the ReadHit's interval: reference id (= the file's), 1-based closed ends
 *   strand            0 unknown, 1 plus, 2 minus (XS tag, else the library type)
 *   partner_same_ref, partner_pos   mate's reference equals the read's; mate's 1-based start (0: none)
 *   nm, nh, sam_flag  NM (as the reference's unsigned char), NH (1 when absent), the record's flag word
 *   singleton, mass   ReadHit::is_singleton, 1 / NH or 0.5 / NH
 *   read_len          ReadHit::read_len(): M + S + I lengths of the kept CIGAR
 *   cig_*             the CIGAR the ReadHit keeps (no H, no P): BAM operation codes
 *   feat_*            readhit_2_genomicFeats: code 0 MATCH / 1 INTRON, closed coordinates
 * *any_paired: some accepted-so-far record had flag 0x1 (the reference clears SINGLE_END_EXP at :607). */
void sbo_bam_decode(const uint8_t *bytes, const int64_t *rec_off, int64_t n, const sbo_bam_opts *o, uint8_t *status,
                    uint64_t *read_id, int32_t *ref, uint32_t *left, uint32_t *right, uint8_t *strand, uint8_t *partner_same_ref,
                    uint32_t *partner_pos, int32_t *nm, int32_t *nh, uint32_t *sam_flag, uint8_t *singleton, double *mass,
                    int32_t *read_len, int64_t *cig_off, uint8_t *cig_type, uint32_t *cig_len, int64_t *feat_off,
                    uint8_t *feat_code, uint32_t *feat_left, uint32_t *feat_right, int32_t *any_paired);
#endif
