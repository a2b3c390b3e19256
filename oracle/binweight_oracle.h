/*
 * oracle/binweight_oracle.h -- TEST INFRASTRUCTURE ONLY (see em_oracle.h).
 *
 * Plain-C restatement of the reference's bin-weight model:
 * what fills the EM matrix F.  References are to /root/reference.
 */
#ifndef SB_BINWEIGHT_ORACLE_H_
#define SB_BINWEIGHT_ORACLE_H_
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* Insert-size law, include/read.hpp:176-192.  use_emp==0: Gaussian(mean, sd).
 * use_emp!=0: histogram emp_hist[0 .. end_offset-start_offset] of counts (as
 * double, like _emp_dist), total_reads; falls back to the Gaussian where the
 * histogram is zero (src/read.cpp:276-289).                                 */
typedef struct {
   double mean, sd;
   int32_t use_emp;
   int32_t start_offset, end_offset;
   int32_t total_reads;
   const double *emp_hist;
} sbo_insert_t;

/* InsertSize::emp_dist_pdf, src/read.cpp:274-297; normal_pdf include/common.h:92-99 */
double sbo_insert_pdf(const sbo_insert_t *is, uint32_t fl);

/* ExonBin::no_gap_ef / gap_ef, include/isoform.h:105-129 */
int sbo_no_gap_ef(int l_left, int l_right, int l_int, int fl);
int sbo_gap_ef(int l_left, int l_right, int l_int, int rl, int gap);

/* ExonBin::effective_len, include/isoform.h:419-516.
 * seg_lens[nseg] = lengths of the isoform's segments spanned by the bin,
 * implicit_idx[nimp] = indices (into seg_lens) of segments lying in the mate
 * gap (ExonBin::bin_under_iso, include/isoform.h:363-411).                  */
int sbo_effective_len(int nseg, const uint32_t *seg_lens, int nimp,
                      const uint32_t *implicit_idx, int fl, int rl);

/* One (bin, isoform) entry of F: LocusContext::set_theory_bin_weight,
 * src/estimate.cpp:209-230.                                                 */
double sbo_bin_weight(int nseg, const uint32_t *seg_lens, int nimp,
                      const uint32_t *implicit_idx, int iso_len, int rl,
                      const sbo_insert_t *is);

#ifdef __cplusplus
}
#endif
#endif
