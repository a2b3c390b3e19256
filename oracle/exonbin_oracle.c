/*
 * oracle/exonbin_oracle.c -- TEST INFRASTRUCTURE ONLY (see exonbin_oracle.h).
 */
#include "exonbin_oracle.h"

/* src/contig.cpp:547-599 */
int sbo_is_compatible(int n_feat, const uint8_t *code, const uint32_t *left, const uint32_t *right,
                      int n_exon, const uint32_t *exon_left, const uint32_t *exon_right)
{
   int e, i, it;
   /* :560-566 lower_bound: first exon whose right >= first feature's left */
   for (e = 0; e < n_exon; ++e)
      if (!(exon_right[e] < left[0])) break;
   if (e == n_exon) return 0;
   /* :568 first exon must contain the first feature (GenomicFeature::contains, :127-132) */
   if (!(exon_left[e] <= left[0] && exon_right[e] >= right[0])) return 0;
   it = e;
   for (i = 1; i < n_feat; ++i) {
      if (code[i] == 2) continue; /* S_GAP :572-574 */
      if (code[i] == 1) {         /* S_INTRON :575-581: must equal the intron after exon `it` */
         /* next_intron_offset = 2*it + 1 must be inside the isoform's 2*n_exon-1 features */
         if (2 * it + 1 >= 2 * n_exon - 1) return 0;
         if (!(left[i] == exon_right[it] + 1 && right[i] == exon_left[it + 1] - 1)) return 0;
      } else { /* S_MATCH :582-591: find_if from `it` on for an exon containing the block */
         int k;
         for (k = it; k < n_exon; ++k)
            if (exon_left[k] <= left[i] && exon_right[k] >= right[i]) break;
         if (k == n_exon) return 0;
         it = k;
      }
   }
   return 1;
}

/* src/estimate.cpp:115-131 */
void sbo_overlap_key(int n_feat, const uint8_t *code, const uint32_t *left, const uint32_t *right,
                     int n_seg, const uint32_t *seg_left, const uint32_t *seg_right, uint8_t *key_out)
{
   int k, f;
   for (k = 0; k < n_seg; ++k) {
      key_out[k] = 0;
      for (f = 0; f < n_feat; ++f) {
         if (code[f] != 0) continue;
         if (left[f] <= seg_right[k] && seg_left[k] <= right[f]) key_out[k] = 1; /* contig.cpp:98-102 */
      }
   }
}

/* src/estimate.cpp:135-198: for every hit, every isoform of its locus */
void sbo_exonbin_batch(const int64_t *iso_off, const int64_t *exon_off, const uint32_t *exon_left,
                       const uint32_t *exon_right, const int64_t *seg_off, const uint32_t *seg_left,
                       const uint32_t *seg_right, int64_t n_hits, const int32_t *hit_locus,
                       const int64_t *feat_off, const uint8_t *feat_code, const uint32_t *feat_left,
                       const uint32_t *feat_right, int32_t compat_words, int32_t key_words,
                       uint32_t *compat, uint32_t *key)
{
   int64_t h;
   for (h = 0; h < n_hits; ++h) {
      const int32_t loc = hit_locus[h];
      const int64_t f0 = feat_off[h];
      const int nf = (int)(feat_off[h + 1] - f0);
      const int64_t i0 = iso_off[loc], s0 = seg_off[loc];
      const int niso = (int)(iso_off[loc + 1] - i0), nseg = (int)(seg_off[loc + 1] - s0);
      int j, k, w;
      for (w = 0; w < compat_words; ++w) compat[h * compat_words + w] = 0;
      for (w = 0; w < key_words; ++w) key[h * key_words + w] = 0;
      if (nf == 0) continue;
      for (j = 0; j < niso && j < 32 * compat_words; ++j) {
         const int64_t e0 = exon_off[i0 + j];
         const int ne = (int)(exon_off[i0 + j + 1] - e0);
         if (ne > 0 && sbo_is_compatible(nf, feat_code + f0, feat_left + f0, feat_right + f0, ne, exon_left + e0,
                                         exon_right + e0))
            compat[h * compat_words + (j >> 5)] |= 1u << (j & 31);
      }
      for (k = 0; k < nseg && k < 32 * key_words; ++k) {
         uint8_t bit;
         sbo_overlap_key(nf, feat_code + f0, feat_left + f0, feat_right + f0, 1, seg_left + s0 + k, seg_right + s0 + k,
                         &bit);
         if (bit) key[h * key_words + (k >> 5)] |= 1u << (k & 31);
      }
   }
}
