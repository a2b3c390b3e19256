/* oracle/collapse_oracle.c -- TEST INFRASTRUCTURE ONLY: a plain-C restatement of the reference's duplicate collapse,
 * HitCluster::collapseAndFilterHits (/root/reference/src/alignments.cpp:658-703), with what it calls:
 *   phi                      include/common.h:112-133   (Abramowitz & Stegun 7.1.26)
 *   getMeanAndSd             include/common.h:100-110   over HitCluster::_read_ref_span (alignments.cpp:527: one entry
 *                            per READ, in the order the reads arrive -- coordinate order of the BAM)
 *   PairedHit::operator<     src/read.cpp:917-923       (left_pos, right_pos)
 *   PairedHit::left_pos/right_pos  src/read.cpp:797-819
 *   PairedHit::operator==    src/read.cpp:897-910, ReadHit::operator== :196-207 (same start, same CIGAR, per mate)
 *   ReadHit mass             src/read.cpp:49-53         0.5 / NH per mate of a pair, 1 / NH for a singleton
 * Pinned: tests/test_collapse_oracle.py compares it with the reference's own HitCluster (oracle/ref_shim.cpp:
 * ref_collapse_cluster, every read through addOpenHit, then the reference's collapse) on random clusters, and with
 * the reference binary's runs through the toy goldens' unique-hit counts.
 *
 * One deliberate choice: the reference sorts its pairs with std::sort, whose order among pairs of equal (left,
 * right) is unspecified; here such pairs keep their input order (a stable sort).  Where tied pairs are all equal
 * fragments (PCR duplicates) the result is the same up to WHICH duplicate represents the unique hit; different
 * fragments with equal ends would make the reference's own output depend on its sort's internals.
 * Only tests/ (and nothing under strawberry_amd/) may use this file. */
#include "collapse_oracle.h"

#include <math.h>
#include <stdlib.h>

double sbo_phi(double x) /* common.h:112-133 */
{
   const double a1 = 0.254829592, a2 = -0.284496736, a3 = 1.421413741, a4 = -1.453152027, a5 = 1.061405429, p = 0.3275911;
   int sign = 1;
   if (x < 0) sign = -1;
   x = fabs(x) / sqrt(2.0);
   const double t = 1.0 / (1.0 + p * x);
   const double y = 1.0 - (((((a5 * t + a4) * t) + a3) * t + a2) * t + a1) * t * exp(-x * x);
   return 0.5 * (1.0 + sign * y);
}

typedef struct {
   uint32_t lpos, rpos; /* PairedHit::left_pos / right_pos */
   int32_t pair;        /* input index: the stable tie-break */
} sbo_key;

static int key_cmp(const void *a, const void *b)
{
   const sbo_key *x = (const sbo_key *)a, *y = (const sbo_key *)b;
   if (x->lpos != y->lpos) return x->lpos < y->lpos ? -1 : 1;   /* read.cpp:917-923 */
   if (x->rpos != y->rpos) return x->rpos < y->rpos ? -1 : 1;
   return x->pair < y->pair ? -1 : (x->pair > y->pair ? 1 : 0);
}

typedef struct {
   uint32_t pos;
   int32_t pair, side;
} sbo_read;

static int read_cmp(const void *a, const void *b)
{
   const sbo_read *x = (const sbo_read *)a, *y = (const sbo_read *)b;
   if (x->pos != y->pos) return x->pos < y->pos ? -1 : 1;
   if (x->pair != y->pair) return x->pair < y->pair ? -1 : 1;
   return x->side - y->side;
}

/* ReadHit::operator==: same left end, same CIGAR -- for M / N CIGARs: the same blocks */
static int mate_equal(const uint32_t *al, const uint32_t *ar, int64_t an, const uint32_t *bl, const uint32_t *br, int64_t bn)
{
   if (an != bn) return 0;
   for (int64_t i = 0; i < an; ++i)
      if (al[i] != bl[i] || ar[i] != br[i]) return 0;
   return 1;
}

int sbo_collapse_cluster(int n_pairs, const int64_t *lo, const uint32_t *ll, const uint32_t *lr, const int64_t *ro,
                         const uint32_t *rl, const uint32_t *rr, const int32_t *nh, int32_t *uniq_pair,
                         double *uniq_mass, double *cluster_mass, int32_t *n_filtered)
{
   *cluster_mass = 0.0;
   *n_filtered = 0;
   if (n_pairs <= 0) return 0;
   sbo_key *key = (sbo_key *)malloc((size_t)n_pairs * sizeof(sbo_key));
   sbo_read *reads = (sbo_read *)malloc((size_t)n_pairs * 2 * sizeof(sbo_read));
   int n_reads = 0;
   for (int p = 0; p < n_pairs; ++p) {
      const int64_t nl = lo[p + 1] - lo[p], nr = ro[p + 1] - ro[p];
      if (nl <= 0 && nr <= 0) {
         free(key);
         free(reads);
         return -1;
      }
      /* read.cpp:797-819: both mates: min of the left ends / max of the right ends; else the one mate's */
      uint32_t lp, rp;
      if (nl > 0 && nr > 0) {
         lp = ll[lo[p]] < rl[ro[p]] ? ll[lo[p]] : rl[ro[p]];
         rp = lr[lo[p + 1] - 1] > rr[ro[p + 1] - 1] ? lr[lo[p + 1] - 1] : rr[ro[p + 1] - 1];
      } else if (nl > 0) {
         lp = ll[lo[p]], rp = lr[lo[p + 1] - 1];
      } else {
         lp = rl[ro[p]], rp = rr[ro[p + 1] - 1];
      }
      key[p].lpos = lp, key[p].rpos = rp, key[p].pair = p;
      if (nl > 0) reads[n_reads].pos = ll[lo[p]], reads[n_reads].pair = p, reads[n_reads++].side = 0;
      if (nr > 0) reads[n_reads].pos = rl[ro[p]], reads[n_reads].pair = p, reads[n_reads++].side = 1;
   }
   /* _read_ref_span: one span per read, in arrival (coordinate) order; getMeanAndSd, common.h:100-110 */
   qsort(reads, (size_t)n_reads, sizeof(sbo_read), read_cmp);
   double sum = 0.0;
   for (int k = 0; k < n_reads; ++k) {
      const int p = reads[k].pair;
      const int span = reads[k].side ? (int)(rr[ro[p + 1] - 1] - rl[ro[p]] + 1) : (int)(lr[lo[p + 1] - 1] - ll[lo[p]] + 1);
      sum += (double)span; /* std::accumulate(..., 0.0) over ints */
   }
   const double mean = sum / (double)n_reads;
   double sq = 0.0;
   for (int k = 0; k < n_reads; ++k) {
      const int p = reads[k].pair;
      const int span = reads[k].side ? (int)(rr[ro[p + 1] - 1] - rl[ro[p]] + 1) : (int)(lr[lo[p + 1] - 1] - ll[lo[p]] + 1);
      const double d = (double)span - mean;
      sq += d * d; /* std::inner_product(diff, diff, 0.0) */
   }
   const double sd = sqrt(sq / (double)n_reads) * 5; /* alignments.cpp:668 */
   qsort(key, (size_t)n_pairs, sizeof(sbo_key), key_cmp); /* :660 (ties: input order, see the header) */
   int n_uniq = 0, last = -1;
   for (int q = 0; q < n_pairs; ++q) {
      const int p = key[q].pair;
      const int64_t nl = lo[p + 1] - lo[p], nr = ro[p + 1] - ro[p];
      /* :671-682: a mate whose span is an outlier skips the pair */
      int skip = 0;
      if (nl > 0 && sbo_phi(((double)(uint32_t)(lr[lo[p + 1] - 1] - ll[lo[p]] + 1) - mean) / sd) > 0.999) skip = 1;
      if (!skip && nr > 0 && sbo_phi(((double)(uint32_t)(rr[ro[p + 1] - 1] - rl[ro[p]] + 1) - mean) / sd) > 0.999) skip = 1;
      if (skip) {
         ++*n_filtered;
         continue;
      }
      /* :683-684: the pair's mass = its reads' masses (read.cpp:49-53, 734-741) */
      const int single = !(nl > 0 && nr > 0);
      double m = 0.0;
      if (nl > 0) m += (single ? 1.0 : 0.5) / nh[p];
      if (nr > 0) m += (single ? 1.0 : 0.5) / nh[p];
      *cluster_mass += m;
      /* :685-696: equal to the latest unique hit? */
      int same = 0;
      if (last >= 0) {
         const int64_t ml = lo[last + 1] - lo[last], mr = ro[last + 1] - ro[last];
         same = ((ml > 0) == (nl > 0)) && ((mr > 0) == (nr > 0)) &&
                (nl <= 0 || mate_equal(ll + lo[last], lr + lo[last], ml, ll + lo[p], lr + lo[p], nl)) &&
                (nr <= 0 || mate_equal(rl + ro[last], rr + ro[last], mr, rl + ro[p], rr + ro[p], nr));
      }
      if (same) {
         uniq_mass[n_uniq - 1] += m;
      } else {
         uniq_pair[n_uniq] = p;
         uniq_mass[n_uniq] = m;
         ++n_uniq;
         last = p;
      }
   }
   free(key);
   free(reads);
   return n_uniq;
}
