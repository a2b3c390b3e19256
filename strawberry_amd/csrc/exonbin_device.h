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
// tables (isoform exons, segments) are shared by all hits of a locus and stay in cache.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_common.h"

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
   // optional (null: not written): what the grouping (bins_device.h) wants to know about a hit without touching its
   // features again -- span = first left end << 32 | last right end (HitCluster's sort key, src/read.cpp:917-923),
   // fhash = a hash of the (left, right) sequence: equal fragments (std::set<Contig>, isoform.h:133,267) have equal
   // spans and hashes, and only those candidates are compared feature by feature
   uint64_t *span;
   uint32_t *fhash;
};

__device__ __forceinline__ uint64_t hit_sig_step(uint64_t h, uint32_t l, uint32_t r)
{
   h = (h ^ l) * 0x9E3779B97F4A7C15ull;
   h = (h ^ r) * 0xC2B2AE3D27D4EB4Full;
   return h ^ (h >> 31);
}
constexpr uint64_t kHitSigSeed = 0x243F6A8885A308D3ull;
__device__ __forceinline__ uint32_t hit_sig_fold(uint64_t h) { return (uint32_t)(h ^ (h >> 32)); }

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

// ------------------------------------------------------------------ per-lane form
// Every lane walks the tables of its own hit's locus: divergent loops, vector loads.  The wave
// form below falls back to it for hits it does not cover; it is also a kernel of its own
// (SBGPU_EXONBIN_LANE=1, A/B measurements).
__global__ __launch_bounds__(256) void exonbin_lane_kernel(ExonBinArgs a)
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

// ------------------------------------------------------------------ wave form
// Hits arrive sorted by locus and position, so the 64 hits of a wave nearly always share one
// locus and a short stretch of it.  The wave then walks the locus' tables ONCE, uniformly:
// exon and segment coordinates come through the scalar cache into SGPRs (constant address
// space loads), every lane tests its own register-resident hit against them, and only the
// exons / segments that overlap the wave's span [lo, hi] are visited at all.  No vector loads
// and no divergent loops in the hot part.
//
// A regular hit -- MATCH blocks separated by exactly one INTRON or GAP each, which is what
// Contig(PairedHit) produces for every well-formed pair -- is held as up to kExonBinBlocks
// blocks plus the connector in front of each.  is_compatible is then one forward pass over the
// exons k (same answers as the reference's walk): `stage` = blocks placed so far.  Block 0 is
// placed at the first exon whose right end reaches it, and must lie inside it.  The moment block
// j-1 is placed at exon k, an INTRON connector j must equal the intron that follows exon k (a GAP
// asks nothing); block j is then placed at the first exon from k on that contains it.
// Compatible <=> nothing failed and all blocks are placed.  Irregular or longer hits take the
// per-lane walk above.
#define SB_AS4 __attribute__((address_space(4)))
template <class T>
__device__ __forceinline__ const SB_AS4 T *scalar_ptr(const T *p)
{
   return (const SB_AS4 T *)p; // tables are read-only for the whole launch: let uniform reads use s_load
}
// first index in [0, n) with right[idx] >= v (right ends ascend); uniform arguments
__device__ __forceinline__ int first_reaching(const SB_AS4 uint32_t *right, int n, uint32_t v)
{
   int lo = 0, hi = n;
   while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (right[mid] < v) lo = mid + 1;
      else hi = mid;
   }
   return lo;
}

constexpr int kExonBinBlocks = (kExonBinRegFeats + 1) / 2; // 4 blocks = up to 7 features
constexpr int kExonBinWaveLoci = 4; // loci a wave serves uniformly before its stragglers go per-lane

struct BlockHit {
   uint32_t l[kExonBinBlocks], r[kExonBinBlocks];   // MATCH blocks; padding: l > r, inside nothing
   uint32_t cl[kExonBinBlocks], cr[kExonBinBlocks]; // connector in front of block j >= 1
   bool intron[kExonBinBlocks];                     // ... is an INTRON (else a GAP)
   int nb;
};

__device__ __forceinline__ void exonbin_locus_uniform(const ExonBinArgs &a, int loc, bool mine, const BlockHit &h,
                                                      int64_t hidx)
{
   const SB_AS4 int64_t *iso_off = scalar_ptr(a.iso_off), *exon_off = scalar_ptr(a.exon_off), *seg_off = scalar_ptr(a.seg_off);
   const SB_AS4 uint32_t *XL = scalar_ptr(a.exon_left), *XR = scalar_ptr(a.exon_right);
   const SB_AS4 uint32_t *SL = scalar_ptr(a.seg_left), *SR = scalar_ptr(a.seg_right);
   const int64_t i0 = iso_off[loc];
   const int niso = (int)(iso_off[loc + 1] - i0);
   const int64_t s0 = seg_off[loc];
   const int nseg = (int)(seg_off[loc + 1] - s0);
   const bool live = mine && h.nb > 0;
   uint32_t rmax = 0;
#pragma unroll
   for (int j = 0; j < kExonBinBlocks; ++j) rmax = (j < h.nb) ? max(rmax, h.r[j]) : rmax;
   const uint32_t lo = wave_min_u32(live ? h.l[0] : 0xffffffffu);
   const uint32_t hi = wave_max_u32(live ? rmax : 0u);
   const int nbmax = (int)wave_max_u32(live ? (uint32_t)h.nb : 0u);
   uint32_t *__restrict__ cout = a.compat + hidx * a.compat_words;
   uint32_t *__restrict__ kout = a.key + hidx * a.key_words;

   for (int w = 0; w < a.compat_words; ++w) {
      uint32_t word = 0;
      const int nbits = niso - 32 * w < 32 ? niso - 32 * w : 32;
      for (int b = 0; b < nbits; ++b) {
         const int64_t iso = i0 + 32 * w + b;
         const int64_t e0 = exon_off[iso];
         const int ne = (int)(exon_off[iso + 1] - e0);
         if (ne == 0 || nbmax == 0) continue;
         // Divergent bools are lane masks in SGPRs; bitwise (not short-circuit) operators keep every
         // condition a mask, so the logic runs in SALU and VALU only does the coordinate compares.
         bool ok = live;
         int stage = 0;
         for (int k = first_reaching(XR + e0, ne, lo); k < ne; ++k) {
            const uint32_t xl = XL[e0 + k];
            if (xl > hi) break; // nothing of this wave reaches that far
            const uint32_t xr = XR[e0 + k];
            const bool has_next = k + 1 < ne;
            const uint32_t in_l = xr + 1, in_r = (has_next ? XL[e0 + k + 1] : 0u) - 1; // the intron after exon k
            // block 0: contig.cpp:560-568
            bool placed = ok & (stage == 0) & !(xr < h.l[0]);
            ok = ok & (!placed | ((xl <= h.l[0]) & (xr >= h.r[0])));
            stage += placed ? 1 : 0;
#pragma unroll
            for (int j = 1; j < kExonBinBlocks; ++j) {
               if (j >= nbmax) break;
               const bool same_intron = has_next & (h.cl[j] == in_l) & (h.cr[j] == in_r); // :575-581
               ok = ok & !(placed & h.intron[j] & !same_intron);
               placed = ok & (stage == j) & (xl <= h.l[j]) & (xr >= h.r[j]);              // :582-591
               stage += placed ? 1 : 0;
            }
            if (!__ballot(ok & (stage < h.nb))) break; // every lane has its answer
         }
         word |= (ok & (stage >= h.nb)) ? (1u << b) : 0u;
      }
      if (mine) cout[w] = word;
   }
   // bin key: the segments the wave's span touches, each against every block
   const int k_lo = first_reaching(SR + s0, nseg, lo);
   for (int w = 0; w < a.key_words; ++w) {
      uint32_t word = 0;
      const int nbits = nseg - 32 * w < 32 ? nseg - 32 * w : 32;
      for (int b = (k_lo > 32 * w ? k_lo - 32 * w : 0); b < nbits; ++b) {
         const uint32_t sl = SL[s0 + 32 * w + b];
         if (sl > hi) break;
         const uint32_t sr = SR[s0 + 32 * w + b];
         bool touch = false;
#pragma unroll
         for (int j = 0; j < kExonBinBlocks; ++j) {
            if (j >= nbmax) break;
            touch = touch | ((h.l[j] <= sr) & (sl <= h.r[j])); // contig.cpp:98-102
         }
         word |= (live & touch) ? (1u << b) : 0u;
      }
      if (mine) kout[w] = word;
   }
}

__global__ __launch_bounds__(256) void exonbin_kernel(ExonBinArgs a)
{
   const int64_t hidx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; // one hit per lane, no loop
   const bool active = hidx < a.n_hits;
   int64_t f0 = 0;
   int nf = 0, my_loc = -1;
   if (active) {
      f0 = a.feat_off[hidx];
      nf = (int)(a.feat_off[hidx + 1] - f0);
      my_loc = a.hit_locus[hidx];
   }
   const bool is_long = nf > kExonBinRegFeats;
   RegHit h;
   h.nf = is_long ? 0 : nf;
   bool regular = (h.nf & 1) != 0; // M (x M)*: odd count, MATCH exactly at the even positions
#pragma unroll
   for (int i = 0; i < kExonBinRegFeats; ++i) {
      const bool in = i < h.nf;
      h.c[i] = in ? a.feat_code[f0 + i] : (uint8_t)2;
      h.l[i] = in ? a.feat_left[f0 + i] : 0u;
      h.r[i] = in ? a.feat_right[f0 + i] : 0u;
      regular = regular & (!in | ((h.c[i] == 0) == ((i & 1) == 0)));
   }
   if (a.span && active) {
      uint64_t sig = kHitSigSeed;
      uint64_t sp = 0;
      if (is_long) {
         for (int i = 0; i < nf; ++i) sig = hit_sig_step(sig, a.feat_left[f0 + i], a.feat_right[f0 + i]);
         sp = ((uint64_t)a.feat_left[f0] << 32) | a.feat_right[f0 + nf - 1];
      } else if (nf > 0) {
         uint32_t last_r = 0;
#pragma unroll
         for (int i = 0; i < kExonBinRegFeats; ++i)
            if (i < nf) {
               sig = hit_sig_step(sig, h.l[i], h.r[i]);
               last_r = h.r[i];
            }
         sp = ((uint64_t)h.l[0] << 32) | last_r;
      }
      a.span[hidx] = sp; // 0: a hit without features (it carries no position)
      a.fhash[hidx] = hit_sig_fold(sig);
   }
   BlockHit bh;
   bh.nb = regular ? (h.nf + 1) / 2 : 0;
#pragma unroll
   for (int j = 0; j < kExonBinBlocks; ++j) {
      const bool in = j < bh.nb;
      bh.l[j] = in ? h.l[2 * j] : 0xffffffffu;
      bh.r[j] = in ? h.r[2 * j] : 0u;
      bh.cl[j] = (in && j) ? h.l[2 * j - 1] : 0u;
      bh.cr[j] = (in && j) ? h.r[2 * j - 1] : 0u;
      bh.intron[j] = in && j && h.c[2 * j - 1] == 1;
   }
   // hits without features are "regular" with no blocks: all-zero words, written by the wave form
   bool todo = active && (regular || nf == 0);
   for (int round = 0; round < kExonBinWaveLoci; ++round) {
      const uint64_t m = __ballot(todo);
      if (!m) break;
      const int loc = __builtin_amdgcn_readlane(my_loc, __ffsll((long long)m) - 1);
      const bool mine = todo && my_loc == loc;
      exonbin_locus_uniform(a, loc, mine, bh, hidx);
      todo = todo && !mine;
   }
   // per-lane walk: more features than the registers hold, an irregular feature list, or a wave
   // spread over many small loci
   if (active && is_long) {
      MemHit m = {a.feat_code + f0, a.feat_left + f0, a.feat_right + f0, nf};
      exonbin_hit(a, hidx, m);
   } else if (active && (todo || !(regular || nf == 0))) {
      exonbin_hit(a, hidx, h);
   }
}

// spans and hashes alone (a caller that made the words elsewhere: sbgpu_bins_create_device)
__global__ __launch_bounds__(256) void hit_signature_kernel(int64_t n_hits, const int64_t *feat_off, const uint32_t *feat_left,
                                                            const uint32_t *feat_right, uint64_t *span, uint32_t *fhash)
{
   const int64_t stride = (int64_t)gridDim.x * blockDim.x;
   for (int64_t h = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; h < n_hits; h += stride) {
      const int64_t f0 = feat_off[h], f1 = feat_off[h + 1];
      uint64_t sig = kHitSigSeed;
      for (int64_t i = f0; i < f1; ++i) sig = hit_sig_step(sig, feat_left[i], feat_right[i]);
      span[h] = f1 > f0 ? (((uint64_t)feat_left[f0] << 32) | feat_right[f1 - 1]) : 0ull;
      fhash[h] = hit_sig_fold(sig);
   }
}

} // namespace sb
