/*
 * oracle/em_oracle.c -- TEST INFRASTRUCTURE ONLY (see em_oracle.h).
 *
 * CPU restatement of EmSolver::init / EmSolver::run and of the abundance
 * epilogue of ruolin/strawberry v1.1.2.  Plain C, one thread, fp64, same
 * order of operations as the reference's loops (Eigen's packet reductions
 * differ from the sequential sums used here by O(1e-16) relative; the tests
 * pin the agreement with the real EmSolver at 1e-12).
 *
 * Build flags mirror the reference (-Ofast, CMakeLists.txt:84) in
 * oracle/Makefile.
 */
#include "em_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* src/estimate.cpp:366-391 (init) and :411-488 (run). */
int sbo_em_locus(int nrow, int niso, const int32_t *count, const double *F,
                 double *theta_out, int32_t *iters_out)
{
   int i, j, it;
   /* estimate.cpp:374-375 -- theta0 = (sum over ALL rows, as double) / niso,
    * computed before any row is dropped. */
   double total = 0.0;
   for (i = 0; i < nrow; ++i) total += (double)count[i];
   for (j = 0; j < niso; ++j) theta_out[j] = total / (double)niso;
   if (iters_out) *iters_out = 0;

   /* estimate.cpp:376-390 -- keep row i iff some weight > 1e-5. */
   int *keep = (int *)malloc(sizeof(int) * (size_t)(nrow > 0 ? nrow : 1));
   int nk = 0;
   for (i = 0; i < nrow; ++i) {
      int remove = 1;
      for (j = 0; j < niso; ++j)
         if (F[(size_t)i * niso + j] > SBO_EM_ROW_EPS) remove = 0;
      if (!remove) keep[nk++] = i;
   }
   /* estimate.cpp:391 */
   if (nk == 0) {
      free(keep);
      return SBO_EM_INIT_EMPTY;
   }

   /* estimate.cpp:417-442 -- working copies */
   size_t ne = (size_t)nk * (size_t)niso;
   double *Fw = (double *)malloc(sizeof(double) * ne);     /* F    */
   double *U = (double *)malloc(sizeof(double) * ne);      /* U    */
   double *obs = (double *)malloc(sizeof(double) * (size_t)nk);
   double *theta = (double *)malloc(sizeof(double) * (size_t)niso);
   double *next_theta = (double *)malloc(sizeof(double) * (size_t)niso);
   for (i = 0; i < nk; ++i) {
      obs[i] = (double)count[keep[i]];
      memcpy(Fw + (size_t)i * niso, F + (size_t)keep[i] * niso, sizeof(double) * (size_t)niso);
   }
   for (j = 0; j < niso; ++j) {
      theta[j] = theta_out[j];
      next_theta[j] = 0.0;
   }

   int status = SBO_EM_MAXITER;
   for (it = 0; it < SBO_EM_MAX_ITER; ++it) { /* estimate.cpp:444 */
      if (iters_out) *iters_out = it + 1;
      /* E-step, estimate.cpp:449-458 */
      for (i = 0; i < nk; ++i) {
         const double *Fi = Fw + (size_t)i * niso;
         double denom = 0.0;
         for (j = 0; j < niso; ++j) denom += Fi[j] * theta[j];
         if (denom == 0) { /* :451-453 -- return false, _theta untouched (= theta0) */
            status = SBO_EM_DENOM_ZERO;
            goto done;
         }
         for (j = 0; j < niso; ++j) {
            double num = obs[i] * Fi[j] * theta[j];
            U[(size_t)i * niso + j] = num / denom;
         }
      }
      /* M-step, estimate.cpp:462-464 */
      for (j = 0; j < niso; ++j) {
         double s = 0.0;
         for (i = 0; i < nk; ++i) s += U[(size_t)i * niso + j];
         next_theta[j] = s;
      }
      /* column renormalisation of F, estimate.cpp:466-478; a zero column
       * stays zero (newF was zero-initialised, `newF(i,j)==0;` is a no-op) */
      for (j = 0; j < niso; ++j) {
         double denom = 0.0;
         for (i = 0; i < nk; ++i) denom += Fw[(size_t)i * niso + j];
         for (i = 0; i < nk; ++i) {
            if (denom == 0)
               Fw[(size_t)i * niso + j] = 0.0;
            else
               Fw[(size_t)i * niso + j] = Fw[(size_t)i * niso + j] / denom;
         }
      }
      /* estimate.cpp:479-481 -- break BEFORE theta = next_theta */
      double d2 = 0.0;
      for (j = 0; j < niso; ++j) {
         double d = next_theta[j] - theta[j];
         d2 += d * d;
      }
      if (sqrt(d2) < SBO_EM_THETA_LIMIT) {
         status = SBO_EM_OK;
         break;
      }
      for (j = 0; j < niso; ++j) theta[j] = next_theta[j];
   }
   /* estimate.cpp:484-486 */
   for (j = 0; j < niso; ++j) theta_out[j] = theta[j];
done:
   free(keep);
   free(Fw);
   free(U);
   free(obs);
   free(theta);
   free(next_theta);
   return status;
}

void sbo_em_batch(int64_t lo, int64_t hi, const int64_t *row_off,
                  const int64_t *iso_off, const int64_t *f_off,
                  const int32_t *count, const double *F, double *theta_out,
                  int32_t *status_out, int32_t *iters_out)
{
   int64_t l;
   for (l = lo; l < hi; ++l) {
      int nrow = (int)(row_off[l + 1] - row_off[l]);
      int niso = (int)(iso_off[l + 1] - iso_off[l]);
      int32_t iters = 0;
      int st = sbo_em_locus(nrow, niso, count + row_off[l], F + f_off[l],
                            theta_out + iso_off[l], &iters);
      if (status_out) status_out[l] = st;
      if (iters_out) iters_out[l] = iters;
   }
}

/* src/estimate.cpp:314-355 */
double sbo_abundance_locus(int niso, const double *theta, const int32_t *length,
                           int32_t total_mapped_reads, int effective_len_norm,
                           double insert_mean, int filter_by_expression,
                           double min_isoform_frac, double *fpkm_out,
                           double *frac_out, int32_t *keep_out)
{
   int j;
   double sum_fpkm = 0.0;
   for (j = 0; j < niso; ++j) {
      double kb;
      fpkm_out[j] = 0.0; /* Isoform::_FPKM default, include/isoform.h:53 */
      frac_out[j] = 0.0; /* Isoform::_frac default, include/isoform.h:52 */
      keep_out[j] = 1;
      if (effective_len_norm) { /* :317-324 */
         kb = (double)length[j] - insert_mean;
         if (kb < 0) {
            keep_out[j] = 2; /* "NA" */
            continue;
         }
         kb = 1e3 / kb;
      } else {
         kb = 1e3 / (double)length[j]; /* :326 */
      }
      double rpm = 1e6 / (double)total_mapped_reads; /* :328 (int -> double) */
      double fpkm = theta[j] * rpm * kb;             /* :329 */
      fpkm_out[j] = fpkm;
      sum_fpkm += fpkm;
   }
   for (j = 0; j < niso; ++j) { /* :337-345 */
      if (keep_out[j] == 2) continue;
      frac_out[j] = fpkm_out[j] / sum_fpkm;
   }
   if (filter_by_expression) { /* :346-355 */
      for (j = 0; j < niso; ++j)
         if (frac_out[j] < min_isoform_frac) keep_out[j] = 0;
   }
   return sum_fpkm;
}

/* src/alignments.cpp:1821-1829 */
double sbo_tpm(int64_t n, const double *fpkm, const int32_t *keep, double *tpm_out)
{
   int64_t k;
   double total_fpkm = 0.0;
   for (k = 0; k < n; ++k)
      if (keep[k]) total_fpkm += fpkm[k];
   for (k = 0; k < n; ++k) tpm_out[k] = keep[k] ? 1e6 * fpkm[k] / total_fpkm : 0.0;
   return total_fpkm;
}
