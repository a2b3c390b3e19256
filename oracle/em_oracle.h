/*
 * oracle/em_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C CPU restatement of the reference's per-locus EM hot path
 * (ruolin/strawberry v1.1.2).  It exists to CHECK the HIP product path; it is
 * never linked into, imported by, or called from the product
 * (strawberry_amd/ and libsbgpu.so).  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may use it.
 *
 * Parity pinning: every function here is validated in tests/ against
 *   (1) the known-answer vectors captured from the reference (SURVEY.md
 *       "Known-answer tests"), and
 *   (2) golden vectors produced by the reference's own EmSolver compiled
 *       unmodified from /root/reference (oracle/_ref, see oracle/Makefile),
 *       committed under tests/golden/.
 *
 * Each function cites the reference lines it follows (paths relative to
 * /root/reference).
 */
#ifndef SB_EM_ORACLE_H_
#define SB_EM_ORACLE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* per-locus status (same numbering as include/sbgpu.h) */
#define SBO_EM_OK 0         /* init()==true, run()==true, converged before the cap  */
#define SBO_EM_INIT_EMPTY 1 /* init()==false: no row has a weight > 1e-5; caller drops the locus */
#define SBO_EM_DENOM_ZERO 2 /* run()==false: a row denominator was exactly 0; theta stays theta0 */
#define SBO_EM_MAXITER 3    /* run()==true after all 1000 iterations (no break)     */

#define SBO_EM_MAX_ITER 1000     /* include/estimate.hpp:237 */
#define SBO_EM_THETA_LIMIT 1e-2  /* include/estimate.hpp:241 */
#define SBO_EM_ROW_EPS 1e-5      /* src/estimate.cpp:380     */

/* EmSolver::init + EmSolver::run for one locus.
 *   count[nrow]          per-bin fragment counts (vector<int> n)
 *   F[nrow*niso]         bin weights, row-major (vector<vector<double>> alpha)
 *   theta_out[niso]      em._theta after init/run
 *   iters_out            number of E-steps started (0 when init fails)
 * returns one of SBO_EM_*.                                               */
int sbo_em_locus(int nrow, int niso, const int32_t *count, const double *F,
                 double *theta_out, int32_t *iters_out);

/* Same over a CSR-of-loci batch (layout of include/sbgpu.h):
 *   row_off[n_loci+1]  first row of each locus in count[]
 *   iso_off[n_loci+1]  first isoform of each locus in theta_out[]
 *   f_off[n_loci+1]    first element of each locus' row-major F block
 * Loci [lo, hi) are processed; nothing outside them is touched, so the
 * caller may run disjoint ranges on several threads.                      */
void sbo_em_batch(int64_t lo, int64_t hi, const int64_t *row_off,
                  const int64_t *iso_off, const int64_t *f_off,
                  const int32_t *count, const double *F, double *theta_out,
                  int32_t *status_out, int32_t *iters_out);

/* LocusContext::estimate_abundances epilogue, src/estimate.cpp:314-355.
 *   theta[niso], length[niso] (exonic length L_j), total_mapped_reads (int),
 *   effective_len_norm / insert_mean            (src/estimate.cpp:317-327)
 *   fpkm_out[niso], frac_out[niso]; keep_out[niso]=0 where the isoform is
 *   erased by the kMinIsoformFrac filter (src/estimate.cpp:346-355) or is
 *   "NA" (negative effective length).  Returns the locus' sum of FPKM.    */
double sbo_abundance_locus(int niso, const double *theta, const int32_t *length,
                           int32_t total_mapped_reads, int effective_len_norm,
                           double insert_mean, int filter_by_expression,
                           double min_isoform_frac, double *fpkm_out,
                           double *frac_out, int32_t *keep_out);

/* Sample::procSample tail, src/alignments.cpp:1821-1829: TPM over the
 * isoforms that survived (keep!=0), in array order.                        */
double sbo_tpm(int64_t n, const double *fpkm, const int32_t *keep, double *tpm_out);

#ifdef __cplusplus
}
#endif
#endif
