// strawberry_amd/csrc/exonbin_device.h -- exon-bin assignment, integer part.
//
// For every fragment ("hit") of a locus the reference asks, isoform by isoform, whether the
// fragment is compatible with it (Contig::is_compatible, /root/reference/src/contig.cpp:547-599)
// and which disjoint exon segments its MATCH blocks touch (LocusContext::overlap_exons,
// src/estimate.cpp:115-131); the set of touched segments is the fragment's exon bin.  Both are
// pure integer interval tests, independent per hit: one lane per hit, results as bit words
//   compat[h][w] bit b  <=>  hit h is compatible with isoform 32*w + b of its locus
//   key[h][w]    bit b  <=>  hit h overlaps segment 32*w + b of its locus
// The bookkeeping that follows (bins keyed by `key`, LocusContext::set_maps,
// include/estimate.hpp:29-52) groups hits by equal words and stays on the host.
//
// HBM-bound by construction: a hit brings 9 B per feature in and 4*(cw+kw) B out; the locus
// tables (isoform exons, segments) are shared by all hits of a locus and are served from L2.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sb {

struct ExonBinArgs {
   // annotation, CSR over loci -> isoforms -> exons, and loci -> segments
   const int64_t *iso_off;
   const int64_t *exon_off;
   const uint32_t *exon_left, *exon_right;
   const int64_t *seg_off;
   const uint32_t *seg_left, *seg_right;
   // hits, CSR over features
   int64_t n_hits;
   const int32_t *hit_locus;
   const int64_t *feat_off;
   const uint8_t *feat_code; // 0 MATCH, 1 INTRON, 2 GAP (Match_t, include/contig.h:26-31)
   const uint32_t *feat_left, *feat_right;
   int32_t compat_words, key_words;
   uint32_t *compat, *key;
};

constexpr int kExonBinRegFeats = 8; // features a hit may have and still be held in registers

// One hit's features: the common short hit sits in registers (every loop over it is fully
// unrolled, so the arrays are indexed statically), a longer one is re-read from L1/L2.
struct RegHit {
   uint8_t c[kExonBinRegFeats];
   uint32_t l[kExonBinRegFeats], r[kExonBinRegFeats];
   int nf;
};
struct MemHit {
   const uint8_t *__restrict__ c;
   const uint32_t *__restrict__ l, *__restrict__ r;
   int nf;
};

// Contig::is_compatible for one isoform (its `ne` exons start at xl/xr).
#define SB_COMPAT_STEP(C, L, R)                                                                   \
   {                                                                                              \
      if ((C) == 1) { /* the isoform's intron after exon `it` must be this one (:575-581) */      \
         if (it + 1 >= ne) return false;                                                          \
         if (!((L) == xr[it] + 1 && (R) == xl[it + 1] - 1)) return false;                         \
      } else if ((C) == 0) { /* a later exon must contain the block (:582-591) */                 \
         int k = it;                                                                              \
         while (k < ne && !(xl[k] <= (L) && xr[k] >= (R))) ++k;                                   \
         if (k == ne) return false;                                                               \
         it = k;                                                                                  \
      } /* S_GAP between the mates: anything goes (:572-574) */                                   \
   }

__device__ __forceinline__ int first_exon(uint32_t l, uint32_t r, const uint32_t *__restrict__ xl,
                                          const uint32_t *__restrict__ xr, int ne)
{
   // contig.cpp:560-568: lower_bound for the first exon with right >= the first block's left;
   // it must contain that block
   int it = 0;
   while (it < ne && xr[it] < l) ++it;
   if (it == ne) return -1;
   return (xl[it] <= l && xr[it] >= r) ? it : -1;
}

__device__ __forceinline__ bool hit_compatible(const RegHit &h, const uint32_t *__restrict__ xl,
                                               const uint32_t *__restrict__ xr, int ne)
{
   int it = first_exon(h.l[0], h.r[0], xl, xr, ne);
   if (it < 0) return false;
#pragma unroll
   for (int i = 1; i < kExonBinRegFeats; ++i) {
      if (i >= h.nf) break;
      SB_COMPAT_STEP(h.c[i], h.l[i], h.r[i]);
   }
   return true;
}
__device__ __forceinline__ bool hit_compatible(const MemHit &h, const uint32_t *__restrict__ xl,
                                               const uint32_t *__restrict__ xr, int ne)
{
   int it = first_exon(h.l[0], h.r[0], xl, xr, ne);
   if (it < 0) return false;
   for (int i = 1; i < h.nf; ++i) SB_COMPAT_STEP(h.c[i], h.l[i], h.r[i]);
   return true;
}
#undef SB_COMPAT_STEP

// GenomicFeature::overlaps(read block, segment), contig.cpp:98-102, over the hit's MATCH blocks
__device__ __forceinline__ bool hit_overlaps(const RegHit &h, uint32_t sl, uint32_t sr)
{
   bool hit = false;
#pragma unroll
   for (int i = 0; i < kExonBinRegFeats; ++i)
      hit |= (i < h.nf) && (h.c[i] == 0) && h.l[i] <= sr && sl <= h.r[i];
   return hit;
}
__device__ __forceinline__ bool hit_overlaps(const MemHit &h, uint32_t sl, uint32_t sr)
{
   bool hit = false;
   for (int i = 0; i < h.nf; ++i) hit |= (h.c[i] == 0) && h.l[i] <= sr && sl <= h.r[i];
   return hit;
}

template <class Hit>
__device__ __forceinline__ void exonbin_hit(const ExonBinArgs &a, int64_t hidx, const Hit &h)
{
   const int32_t loc = a.hit_locus[hidx];
   const int64_t i0 = a.iso_off[loc];
   const int niso = (int)(a.iso_off[loc + 1] - i0);
   const int64_t s0 = a.seg_off[loc];
   const int nseg = (int)(a.seg_off[loc + 1] - s0);
   uint32_t *__restrict__ cout = a.compat + hidx * a.compat_words;
   uint32_t *__restrict__ kout = a.key + hidx * a.key_words;
   for (int w = 0; w < a.compat_words; ++w) {
      uint32_t word = 0;
      const int hi = niso - 32 * w < 32 ? niso - 32 * w : 32;
      for (int b = 0; b < hi; ++b) {
         const int64_t iso = i0 + 32 * w + b;
         const int64_t e0 = a.exon_off[iso];
         const int ne = (int)(a.exon_off[iso + 1] - e0);
         if (h.nf > 0 && ne > 0 && hit_compatible(h, a.exon_left + e0, a.exon_right + e0, ne)) word |= 1u << b;
      }
      cout[w] = word;
   }
   for (int w = 0; w < a.key_words; ++w) {
      uint32_t word = 0;
      const int hi = nseg - 32 * w < 32 ? nseg - 32 * w : 32;
      for (int b = 0; b < hi; ++b)
         if (hit_overlaps(h, a.seg_left[s0 + 32 * w + b], a.seg_right[s0 + 32 * w + b])) word |= 1u << b;
      kout[w] = word;
   }
}

__global__ __launch_bounds__(256) void exonbin_kernel(ExonBinArgs a)
{
   const int64_t stride = (int64_t)gridDim.x * blockDim.x;
   for (int64_t hidx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; hidx < a.n_hits; hidx += stride) {
      const int64_t f0 = a.feat_off[hidx];
      const int nf = (int)(a.feat_off[hidx + 1] - f0);
      if (nf <= kExonBinRegFeats) {
         RegHit h;
         h.nf = nf;
#pragma unroll
         for (int i = 0; i < kExonBinRegFeats; ++i) {
            const bool in = i < nf;
            h.c[i] = in ? a.feat_code[f0 + i] : (uint8_t)2;
            h.l[i] = in ? a.feat_left[f0 + i] : 0u;
            h.r[i] = in ? a.feat_right[f0 + i] : 0u;
         }
         exonbin_hit(a, hidx, h);
      } else {
         MemHit h = {a.feat_code + f0, a.feat_left + f0, a.feat_right + f0, nf};
         exonbin_hit(a, hidx, h);
      }
   }
}

} // namespace sb
