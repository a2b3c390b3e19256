/*
 * oracle/sam_writer.c -- TEST INFRASTRUCTURE ONLY (see em_oracle.h).
 *
 * Turns unique hits (the layout of sbgpu_hits_t: MATCH / INTRON features of the left mate, one GAP, the right mate's
 * features) back into the coordinate-sorted SAM text of the read pairs behind them, so that the REFERENCE PROGRAM
 * (oracle/_ref/strawberry_ref, after our sam2bam) can be run on a sample that was drawn as hits: bench.py's
 * cpu_baseline / parity leg and tools/dropin_timing.py.  No reference code: the SAM columns are the format's.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* CIGAR and aligned length of features [x0, x1): code 0 = MATCH -> M, anything else (INTRON) -> N */
static int mate_cigar(const uint8_t *code, const uint32_t *left, const uint32_t *right, int64_t x0, int64_t x1, char *out, int *bases)
{
   int n = 0, b = 0;
   for (int64_t i = x0; i < x1; ++i) {
      const long ln = (long)right[i] - (long)left[i] + 1;
      n += sprintf(out + n, "%ld%c", ln, code[i] == 0 ? 'M' : 'N');
      if (code[i] == 0) b += (int)ln;
   }
   *bases = b;
   return n;
}

/* order[n_rec]: record indices in output order; record r = 2 * pair + mate, pair p belongs to the hit k with
 * pair_off[k] <= p < pair_off[k + 1] (a unique hit of mass m stands for m read pairs: copies c = p - pair_off[k]).
 * gap_idx[k]: index of hit k's GAP feature.  Returns 0, or -1 when the file cannot be written.                       */
int sbo_write_sam(const char *path, int64_t chrom_len, int64_t n_hits, const int64_t *feat_off, const uint8_t *code,
                  const uint32_t *left, const uint32_t *right, const int64_t *gap_idx, const int64_t *pair_off,
                  int64_t n_rec, const int64_t *order)
{
   FILE *f = fopen(path, "w");
   if (!f) return -1;
   static char buf[1 << 22];
   setvbuf(f, buf, _IOFBF, sizeof buf);
   fprintf(f, "@HD\tVN:1.0\tSO:coordinate\n@SQ\tSN:chr1\tLN:%lld\n", (long long)chrom_len);
   char cl[512], cr[512], seq[4096], qual[4096];
   memset(seq, 'A', sizeof seq);
   memset(qual, 'I', sizeof qual);
   for (int64_t i = 0; i < n_rec; ++i) {
      const int64_t r = order[i], p = r >> 1;
      int64_t lo = 0, hi = n_hits; /* last k with pair_off[k] <= p */
      while (hi - lo > 1) {
         const int64_t mid = (lo + hi) / 2;
         if (pair_off[mid] <= p) lo = mid;
         else hi = mid;
      }
      const int64_t k = lo, c = p - pair_off[k], f0 = feat_off[k], f1 = feat_off[k + 1], g = gap_idx[k];
      int nl, nr;
      mate_cigar(code, left, right, f0, g, cl, &nl);
      mate_cigar(code, left, right, g + 1, f1, cr, &nr);
      if (nl > 4000 || nr > 4000) {
         fclose(f);
         return -1;
      }
      const long pl = (long)left[f0], pr = (long)left[g + 1], tlen = (long)right[f1 - 1] - pl + 1;
      if ((r & 1) == 0)
         fprintf(f, "r%lld_%lld\t99\tchr1\t%ld\t255\t%s\t=\t%ld\t%ld\t%.*s\t%.*s\tNH:i:1\tXS:A:+\n", (long long)k, (long long)c, pl, cl, pr, tlen, nl, seq, nl, qual);
      else
         fprintf(f, "r%lld_%lld\t147\tchr1\t%ld\t255\t%s\t=\t%ld\t%ld\t%.*s\t%.*s\tNH:i:1\tXS:A:+\n", (long long)k, (long long)c, pr, cr, pl, -tlen, nr, seq, nr, qual);
   }
   return fclose(f) == 0 ? 0 : -1;
}
