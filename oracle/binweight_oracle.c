/*
 * oracle/binweight_oracle.c -- TEST INFRASTRUCTURE ONLY (see binweight_oracle.h).
 */
#include "binweight_oracle.h"

#include <math.h>

static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }

/* include/common.h:92-99 */
static double normal_pdf(double x, double m, double s)
{
   static const double inv_sqrt_2pi = 0.3989422804014327;
   double a = (x - m) / s;
   return inv_sqrt_2pi / s * exp(-0.5 * a * a);
}

/* src/read.cpp:274-297 */
double sbo_insert_pdf(const sbo_insert_t *is, uint32_t fl)
{
   if (is->use_emp) {
      double ret = 0.0;
      /* uint-vs-int comparisons as in the reference (offsets are positive) */
      if (fl < (uint32_t)is->start_offset || fl > (uint32_t)is->end_offset) {
      } else {
         ret = is->emp_hist[fl - (uint32_t)is->start_offset] / is->total_reads;
      }
      if (ret != 0.0) return ret;
   }
   double p = normal_pdf((double)fl, is->mean, is->sd);
   return p > 0 ? p : 0.0;
}

/* include/isoform.h:105-115 */
int sbo_no_gap_ef(int l_left, int l_right, int l_int, int fl)
{
   if (fl < l_int + 2) return 0;
   if (fl > l_left + l_right + l_int) return 0;
   int mid = fl - l_int - 1;
   return imin(l_left, mid) + imin(l_right, mid) - mid;
}

/* include/isoform.h:117-129 */
int sbo_gap_ef(int l_left, int l_right, int l_int, int rl, int gap)
{
   if (2 * rl + gap < l_int + 2) return 0;
   if (2 * rl + gap > l_left + l_right + l_int) return 0;
   int start = imax(rl, l_left + l_int - gap - 1);
   int end = imin(l_left, l_left + l_right + l_int - gap - rl);
   return imax(0, end - start);
}

/* include/isoform.h:419-516.  The reference mixes uint segment lengths with
 * int arithmetic; results wrap mod 2^32 and are returned as int, which is what
 * the two's-complement int arithmetic below produces as well.               */
int sbo_effective_len(int nseg, const uint32_t *seg_lens, int nimp,
                      const uint32_t *implicit_idx, int fl, int rl)
{
   const uint32_t *s = seg_lens;
   int gap = fl - 2 * rl;
   if (nseg == 1) return (int)(s[0] - (uint32_t)fl + 1u); /* :427-429 */
   if (nseg == 2) return sbo_no_gap_ef((int)s[0], (int)s[1], 0, fl); /* :430-432 */
   if (nseg == 3) { /* :435-447 */
      if (nimp == 1) return sbo_gap_ef((int)s[0], (int)s[2], (int)s[1], rl, gap);
      if (nimp == 0)
         return sbo_no_gap_ef((int)s[0], (int)s[2], (int)s[1], fl) -
                sbo_gap_ef((int)s[0], (int)s[2], (int)s[1], rl, gap);
      return 0; /* assert(false) in the reference */
   }
   if (nseg == 4) { /* :448-475 */
      int hit14 = sbo_gap_ef((int)s[0], (int)s[3], (int)(s[2] + s[1]), rl, gap);
      int hit24 = sbo_gap_ef((int)s[3], (int)s[1], (int)s[2], rl, gap);
      int hit124 = sbo_gap_ef((int)(s[0] + s[1]), (int)s[3], (int)s[2], rl, gap);
      int hit13 = sbo_gap_ef((int)s[0], (int)s[2], (int)s[1], rl, gap);
      int hit134 = sbo_gap_ef((int)s[0], (int)(s[2] + s[3]), (int)s[1], rl, gap);
      if (nimp == 0) {
         int hit_all_124 = hit124 - hit14 - hit24;
         int hit_all_134 = hit134 - hit14 - hit13;
         int total = sbo_no_gap_ef((int)s[0], (int)s[3], (int)(s[1] + s[2]), fl);
         return total - hit_all_124 - hit_all_134 - hit14;
      }
      if (nimp == 2) return hit14;
      if (implicit_idx[0] == 1) return hit134 - hit14 - hit13;
      return hit124 - hit14 - hit24;
   }
   /* >= 5 segments, :476-515: scan the start position in the first segment */
   {
      uint32_t num_inners = (uint32_t)nseg - 2u;
      uint32_t num_pos = 0;
      uint32_t target = (uint32_t)(pow(2.0, (double)nseg) - 1);
      int k, i, inner = 0;
      for (k = 0; k < nimp; ++k) target &= ~(1u << implicit_idx[k]);
      for (k = 1; k < nseg - 1; ++k) inner += (int)s[k];
      for (i = 1; (uint32_t)i != s[0] + 1u; ++i) {
         uint32_t hit = 1;
         int bp_last = fl - i - inner;
         if ((uint32_t)bp_last > s[nseg - 1]) continue; /* int-vs-uint compare: negatives continue too */
         if (bp_last == 0) break;
         hit |= (1u << (nseg - 1));
         /* right-end cover */
         int last_rest_bp = rl - bp_last;
         uint32_t j = num_inners;
         while (last_rest_bp > 0 && j > 0) {
            hit |= (1u << j);
            last_rest_bp = (int)((uint32_t)last_rest_bp - s[j]);
            j = j - 1;
         }
         /* left-end cover */
         int first_rest_bp = rl - i;
         j = 1;
         while (first_rest_bp > 0 && j <= num_inners) {
            hit |= (1u << j);
            first_rest_bp = (int)((uint32_t)first_rest_bp - s[j]);
            j = j + 1;
         }
         if (hit == target) num_pos++;
      }
      return (int)num_pos;
   }
}

/* src/estimate.cpp:209-230 */
double sbo_bin_weight(int nseg, const uint32_t *seg_lens, int nimp,
                      const uint32_t *implicit_idx, int iso_len, int rl,
                      const sbo_insert_t *is)
{
   int k, fl;
   int lmax = 0, inner = 0;
   for (k = 0; k < nseg; ++k) lmax += (int)seg_lens[k];
   for (k = 1; k < nseg - 1; ++k) inner += (int)seg_lens[k];
   int lmin = is->use_emp ? is->start_offset : rl; /* :214-219 */
   if (nseg > 2) lmin = imax(lmin, inner);         /* :220-221 */
   double weight = 0.0;
   for (fl = lmin; fl <= lmax; ++fl) { /* :223-227 */
      double le_eff = (double)sbo_effective_len(nseg, seg_lens, nimp, implicit_idx, fl, rl);
      double tmp = sbo_insert_pdf(is, (uint32_t)fl) * le_eff / (double)(iso_len - fl + 1);
      weight += tmp;
   }
   return weight;
}
