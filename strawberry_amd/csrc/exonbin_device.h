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
   // optional (null: the exon walk everywhere): the isoforms in the SEGMENT basis (iso_masks_kernel below) -- bit s of
   // iso_member[i]: segment s of the locus lies in isoform i; bit s of iso_start[i]: one of its exons starts there -- and
   // per locus whether that form may be used (<= 64 segments, every exon a run of adjacent segments)
   const uint64_t *iso_member, *iso_start;
   const uint32_t *locus_seg_ok; // 0: no (the exon walk), 1: yes, up to 32 segments, 2: yes, up to 64, 3: yes, 65-128 segments
                                 // or isoforms: bits 64-127 of the masks in the *_hi arrays, served by exonbin_seg128_kernel
   const uint64_t *locus_adj; // per locus: bit s = segment s begins right behind segment s - 1
   const uint64_t *iso_member_hi, *iso_start_hi, *locus_adj_hi;
};

// (32 bits, and no multiply per feature: integer multiplies run at a quarter of the vector rate and exonbin_kernel is
// bound by its vector instructions -- a step is xor, rotate, xor, shift-add, each a bijection of h for given (l, r),
// the two ends entering at different rotations; the two multiplies of the avalanche come once per hit.  Collisions
// only cost a feature-by-feature compare.)
__device__ __forceinline__ uint32_t hit_sig_step(uint32_t h, uint32_t l, uint32_t r)
{
   h = __builtin_rotateleft32(h ^ l, 13) ^ r;
   return h + (h << 3);
}
constexpr uint32_t kHitSigSeed = 0x243F6A88u;
__device__ __forceinline__ uint32_t hit_sig_fold(uint32_t h)
{
   h ^= h >> 16;
   h *= 0x85EBCA6Bu;
   h ^= h >> 13;
   h *= 0xC2B2AE35u;
   return h ^ (h >> 16);
}

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
   bool intron[kExonBinBlocks];                     // ... is an INTRON (else a GAP)
   int nb;
};

__device__ __forceinline__ void exonbin_locus_uniform(const ExonBinArgs &a, int loc, bool mine, const BlockHit &h,
                                                      int64_t hidx, int64_t f0)
{
   // the connector in front of block j >= 1, read again here (feature 2j - 1 of the hit): the segment-basis form, which
   // serves nearly every locus, does not need them, and eight registers held for this path cost it a wave per SIMD
   uint32_t cl[kExonBinBlocks], cr[kExonBinBlocks];
#pragma unroll
   for (int j = 0; j < kExonBinBlocks; ++j) {
      const bool in = mine && j > 0 && j < h.nb;
      cl[j] = in ? a.feat_left[f0 + 2 * j - 1] : 0u;
      cr[j] = in ? a.feat_right[f0 + 2 * j - 1] : 0u;
   }
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
               const bool same_intron = has_next & (cl[j] == in_l) & (cr[j] == in_r); // :575-581
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

// ------------------------------------------------------------------ segment-basis form
// The exon walk above costs ~1000 scalar instructions per wave (mask logic per isoform, exon and block) and the
// kernel is bound by the CU's one scalar unit.  For loci of up to 64 segments the same answers come from bit masks.
// The segments are the disjoint pieces of the union of the locus' exons (IRanges::disjoint), so every exon is a run of
// adjacent segments, and for a hit whose blocks ascend (M (x M)*, every INTRON connector filling the gap between its
// two blocks exactly):
//   * a block lies inside ONE exon of isoform i  <=>  the segments it touches cover it completely (a statement about
//     the hit alone: the overlap lengths add up to the block's length), all of them are members of i, and none but
//     the first is an exon start of i.  Exons of an isoform are disjoint and sorted and the blocks ascend, so "the
//     first exon reaching block 0 contains it" (contig.cpp:560-568) and "an exon from `it` on contains block j"
//     (:582-591) both say just that.
//   * an INTRON connector between blocks j-1 and j equals the intron after the exon holding block j-1 (:575-581)
//     <=>  block j-1 ends where its last segment ends, block j starts where its first segment starts (the hit alone),
//     that first segment is an exon start of i, and no segment strictly between the two is a member of i.
// Per hit: masks need_member / forbid_start / need_start / forbid_member from ONE uniform walk over the segments in
// the wave's span (the walk the bin key needs anyway); per isoform: four AND tests against two 64-bit words.
// iso_masks_kernel checks the premise per locus (every exon exactly a run of adjacent segments) and leaves the exon
// walk to the loci that fail it or have more than 64 segments.
// Two launches: iso_masks_locus_kernel (a thread per locus: the shape test, the adjacency words) and iso_masks_kernel (a
// thread per ISOFORM: its member / start words; an exon that is not a run of adjacent segments clears the locus' form).  A
// thread per locus for everything took 3.9 ms on 1 500 loci of a hundred segments and a dozen isoforms each.
__global__ __launch_bounds__(256) void iso_masks_locus_kernel(ExonBinArgs a, int64_t n_loci, uint32_t *seg_ok, uint64_t *locus_adj, uint64_t *locus_adj_hi)
{
   const int64_t stride = (int64_t)gridDim.x * blockDim.x;
   for (int64_t l = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; l < n_loci; l += stride) {
      const int64_t ni = a.iso_off[l + 1] - a.iso_off[l], s0 = a.seg_off[l];
      const int nseg = (int)(a.seg_off[l + 1] - s0);
      const bool ok = nseg >= 1 && nseg <= 128 && ni <= 128; // (exonbin_seg128_kernel: two segments per lane, masks of 128 bits)
      const bool small = nseg <= 64 && ni <= 64;             // (exonbin_locus_segbasis: a lane per segment and per isoform)
      uint64_t adj[2] = {0, 0};
      for (int sg = 1; sg < nseg && sg < 128; ++sg)
         if (a.seg_left[s0 + sg] == a.seg_right[s0 + sg - 1] + 1u) adj[sg >> 6] |= 1ull << (sg & 63);
      locus_adj[l] = adj[0];
      locus_adj_hi[l] = adj[1];
      seg_ok[l] = ok ? (small ? (nseg <= 32 ? 1u : 2u) : 3u) : 0u;
   }
}
__global__ __launch_bounds__(256) void iso_masks_kernel(ExonBinArgs a, int64_t n_loci, int64_t n_iso, uint64_t *member, uint64_t *start, uint32_t *seg_ok,
                                                        uint64_t *member_hi, uint64_t *start_hi)
{
   const int64_t stride = (int64_t)gridDim.x * blockDim.x;
   for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_iso; i += stride) {
      int64_t lo = 0, hi = n_loci; // the isoform's locus: the last l with iso_off[l] <= i
      while (hi - lo > 1) {
         const int64_t mid = (lo + hi) >> 1;
         if (a.iso_off[mid] <= i) lo = mid;
         else hi = mid;
      }
      const int64_t l = lo, s0 = a.seg_off[l];
      const int nseg = (int)(a.seg_off[l + 1] - s0);
      bool ok = seg_ok[l] != 0u; // (the shape; another isoform's thread may clear it meanwhile: then these words are not read)
      uint64_t m[2] = {0, 0}, st[2] = {0, 0};
      int sidx = 0;
      for (int64_t e = a.exon_off[i]; ok && e < a.exon_off[i + 1]; ++e) {
         const uint32_t xl = a.exon_left[e], xr = a.exon_right[e];
         while (sidx < nseg && a.seg_left[s0 + sidx] < xl) ++sidx;
         if (sidx >= nseg || a.seg_left[s0 + sidx] != xl) {
            ok = false;
            break;
         }
         st[sidx >> 6] |= 1ull << (sidx & 63);
         uint32_t reach = xl - 1;
         while (sidx < nseg && a.seg_left[s0 + sidx] == reach + 1 && a.seg_right[s0 + sidx] <= xr) {
            m[sidx >> 6] |= 1ull << (sidx & 63);
            reach = a.seg_right[s0 + sidx];
            ++sidx;
         }
         if (reach != xr) ok = false;
      }
      member[i] = m[0];
      start[i] = st[0];
      member_hi[i] = m[1];
      start_hi[i] = st[1];
      if (!ok && seg_ok[l] != 0u) seg_ok[l] = 0u; // an exon that is no run of adjacent segments: the exon walk for this locus
   }
}

// The locus' segments (<= 64) are read ONCE per wave, lane s holding segment s: two coalesced loads in flight together.  Everything after that is register traffic --
// v_readlane for the uniform walks, a ballot for "the first segment reaching the wave's span", ds_bpermute for a
// block's first / last segment -- where the scalar-load form waited for some twenty dependent s_loads per wave (a
// binary search, two loads per segment step, two per isoform) and the kernel sat 60 % of its cycles in s_waitcnt.
__device__ __forceinline__ uint32_t lane_bcast(uint32_t v, int lane) { return (uint32_t)__builtin_amdgcn_readlane((int)v, lane); }
__device__ __forceinline__ uint64_t lane_bcast(uint64_t v, int lane)
{
   return ((uint64_t)lane_bcast((uint32_t)(v >> 32), lane) << 32) | lane_bcast((uint32_t)v, lane);
}
__device__ __forceinline__ uint32_t lane_gather(uint32_t v, uint32_t lane) { return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(lane << 2), (int)v); }

// The locus' numbers (first isoform / segment, counts, adjacency word) arrive as arguments: exonbin_kernel has them in
// lanes (vector loads issued beside the feature loads) and hands the round's locus over with v_readlane -- the scalar
// loads this function made itself (five dependent s_loads before the first segment could be asked for) were a memory
// round trip per locus on the wave's critical path.  The isoforms' member / start words sit in lanes too (lane i:
// isoform i), loaded together with the segments.
template <class M>
__device__ __forceinline__ void exonbin_locus_segbasis(const ExonBinArgs &a, int64_t i0, int niso, int64_t s0, int nseg, uint64_t adj64,
                                                       bool mine, const BlockHit &h, int64_t hidx)
{
   const M adj = (M)adj64; // bit s: segment s begins right behind segment s - 1
   const int lane = (int)(threadIdx.x & 63u);
   const uint32_t my_sl = lane < nseg ? a.seg_left[s0 + lane] : 0xffffffffu;
   const uint32_t my_sr = lane < nseg ? a.seg_right[s0 + lane] : 0xffffffffu; // (past the end: reaches anything)
   const M my_mem = lane < niso ? (M)a.iso_member[i0 + lane] : (M)0;
   const M my_sta = lane < niso ? (M)a.iso_start[i0 + lane] : (M)0;
   const bool live = mine && h.nb > 0;
   uint32_t rmax = 0;
#pragma unroll
   for (int j = 0; j < kExonBinBlocks; ++j) rmax = (j < h.nb) ? max(rmax, h.r[j]) : rmax;
   const uint32_t lo = wave_min_u32(live ? h.l[0] : 0xffffffffu);
   const uint32_t hi = wave_max_u32(live ? rmax : 0u);
   const int nbmax = (int)wave_max_u32(live ? (uint32_t)h.nb : 0u);
   // A block's touched segments are an index range [sa, sb] (segments are sorted and disjoint): sa = the number of
   // segments that end before it, sb + 1 = the number that begin no later than it ends -- two compares and two carry
   // adds per (segment, block) in the uniform walk, nothing else.
   uint32_t n_cnt[kExonBinBlocks]; // low half: segments ending before the block; high half: segments beginning no later than its end
#pragma unroll
   for (int j = 0; j < kExonBinBlocks; ++j) n_cnt[j] = 0u;
   const uint64_t reaching = __ballot(my_sr >= lo); // (lanes from nseg on always do: k_lo <= nseg)
   const int k_lo = nbmax > 0 ? (int)__ffsll((long long)reaching) - 1 : nseg;
   for (int sg = k_lo; sg < nseg; ++sg) {
      const uint32_t sl = lane_bcast(my_sl, sg);
      if (sl > hi) break;
      const uint32_t sr = lane_bcast(my_sr, sg);
#pragma unroll
      for (int j = 0; j < kExonBinBlocks; ++j) {
         if (j >= nbmax) break;
         n_cnt[j] += ((sr < h.l[j]) ? 1u : 0u) + ((sl <= h.r[j]) ? 0x10000u : 0u);
      }
   }
   bool valid = live;
   M need_member = 0, forbid_start = 0, need_start = 0, forbid_member = 0;
   constexpr uint32_t kTop = 8 * sizeof(M) - 1;
   constexpr M kOne = 1, kTwo = 2, kAll = ~(M)0;
   uint32_t sb_prev = 0;
   bool ends_prev = false;
   // (blocks no lane of the wave has are skipped: they would change nothing -- every statement below is under `in` --, and
   // the typical wave's longest hit has two blocks of the four the loop is unrolled for)
#pragma unroll
   for (int j = 0; j < kExonBinBlocks; ++j) {
      if (j >= nbmax) break; // (uniform)
      const bool in = live & (j < h.nb);
      const uint32_t sa = (uint32_t)k_lo + (n_cnt[j] & 0xffffu), sb1 = (uint32_t)k_lo + (n_cnt[j] >> 16); // [sa, sb1)
      const bool any = in & (sb1 > sa);
      const uint32_t sb = any ? sb1 - 1u : 0u, sa_c = any ? sa : 0u;
      // the block's own ends against its first and last segment
      const uint32_t first_l = lane_gather(my_sl, sa_c), last_r = lane_gather(my_sr, sb);
      const M upto_sb = (sb >= kTop) ? kAll : (M)((kTwo << sb) - kOne);
      const M mask = any ? (M)(upto_sb & ~(M)((kOne << sa_c) - kOne)) : (M)0; // bits sa .. sb
      const M inner = mask & (M)(mask - kOne);                                // all but the first
      // covered completely: it begins and ends inside its first / last segment and the segments between are adjacent
      valid = valid & (!in | (any & (h.l[j] >= first_l) & (h.r[j] <= last_r) & ((M)(inner & ~adj) == (M)0)));
      need_member |= mask;
      forbid_start |= inner;
      if (j > 0) {
         const bool intr = in & h.intron[j];
         valid = valid & (!intr | (any & ends_prev & (h.l[j] == first_l)));
         const M first_j = any ? (M)(kOne << sa_c) : (M)0, upto_prev = (sb_prev >= kTop) ? kAll : (M)((kTwo << sb_prev) - kOne);
         need_start |= intr ? first_j : (M)0;
         forbid_member |= intr ? (M)((M)(first_j - kOne) & ~upto_prev) : (M)0; // strictly between block j-1's last and block j's first
      }
      sb_prev = sb;
      ends_prev = any & (h.r[j] == last_r);
   }
   uint32_t *__restrict__ cout = a.compat + hidx * a.compat_words;
   uint32_t *__restrict__ kout = a.key + hidx * a.key_words;
   for (int w = 0; w < a.compat_words; ++w) {
      uint32_t word = 0;
      const int nbits = niso - 32 * w < 32 ? niso - 32 * w : 32;
      for (int b = 0; b < nbits; ++b) {
         const M m = lane_bcast(my_mem, 32 * w + b), st = lane_bcast(my_sta, 32 * w + b);
         const M bad = (need_member & ~m) | (forbid_start & st) | (need_start & ~st) | (forbid_member & m);
         word |= (valid & (bad == (M)0) & (m != (M)0)) ? (1u << b) : 0u; // (m == 0: an isoform without exons is compatible with nothing)
      }
      if (mine) cout[w] = word;
   }
   const uint64_t keybits = live ? (uint64_t)need_member : 0ull;
   if (mine) {
      if (a.key_words > 0) kout[0] = (uint32_t)keybits;
      if (a.key_words > 1) kout[1] = (uint32_t)(keybits >> 32);
      for (int w = 2; w < a.key_words; ++w) kout[w] = 0u;
   }
}

// A lane's features, all kExonBinRegFeats slots of the three arrays, with FIVE loads in flight together: two
// global_load_dwordx4 per coordinate array and one 8-byte load of the codes, from the hit's first feature on (the
// slots past its last feature read the next hits' -- masked off afterwards).  The form before asked for every slot
// under its own `i < nf` branch, and the codes were consumed as they came: five dependent memory round trips at the
// head of every wave, most of a wave's 8 us.  Reading eight slots from f0 must stay inside the arrays: `wide` says so
// (every lane's f0 + 8 <= the feature count; only the last wave of a launch fails it and loads slot by slot, clamped).
struct __attribute__((packed, aligned(4))) PackedU32x4 {
   uint32_t v[4];
};
struct __attribute__((packed, aligned(1))) PackedU64 {
   uint64_t v;
};
static_assert(kExonBinRegFeats == 8, "the wide feature loads are written for eight slots");

__device__ __forceinline__ void load_hit_slots(const ExonBinArgs &a, int64_t f0, int64_t n_feat, bool wide, uint64_t &codes,
                                               uint32_t (&L)[kExonBinRegFeats], uint32_t (&R)[kExonBinRegFeats])
{
   if (wide) { // (uniform)
      const PackedU32x4 l0 = *reinterpret_cast<const PackedU32x4 *>(a.feat_left + f0), l1 = *reinterpret_cast<const PackedU32x4 *>(a.feat_left + f0 + 4);
      const PackedU32x4 r0 = *reinterpret_cast<const PackedU32x4 *>(a.feat_right + f0), r1 = *reinterpret_cast<const PackedU32x4 *>(a.feat_right + f0 + 4);
      const PackedU64 c0 = *reinterpret_cast<const PackedU64 *>(a.feat_code + f0);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
         L[i] = l0.v[i];
         L[i + 4] = l1.v[i];
         R[i] = r0.v[i];
         R[i + 4] = r1.v[i];
      }
      codes = c0.v; // byte i: the code of slot i
   } else if (n_feat > 0) {
      codes = 0;
#pragma unroll
      for (int i = 0; i < kExonBinRegFeats; ++i) {
         const int64_t idx = f0 + i < n_feat ? f0 + i : n_feat - 1;
         codes |= (uint64_t)a.feat_code[idx] << (8 * i);
         L[i] = a.feat_left[idx];
         R[i] = a.feat_right[idx];
      }
   } else {
      codes = 0;
#pragma unroll
      for (int i = 0; i < kExonBinRegFeats; ++i) L[i] = R[i] = 0u;
   }
}

__device__ __forceinline__ void exonbin_tile(const ExonBinArgs &a, const int64_t hidx, const int64_t n_feat)
{
   const bool active = hidx < a.n_hits;
   // first level: the hit's offsets and its locus, together (idle lanes of the last wave read hit 0 -- no branch around
   // the loads, so none of them is waited for before the other is on its way)
   const int64_t hq = active ? hidx : 0;
   int loc_q = a.hit_locus[hq];
   int64_t f0_q = a.feat_off[hq], f1_q = a.feat_off[hq + 1];
   // (all three are wanted before the second level goes out: without this pin the compiler waits for the offsets only,
   // and later -- at the join of the two feature-load forms -- for everything in flight, to get at the locus)
   asm volatile("" : "+v"(loc_q), "+v"(f0_q), "+v"(f1_q));
   const int64_t f0 = active ? f0_q : 0;
   const int nf = active ? (int)(f1_q - f0_q) : 0;
   const int my_loc = active ? loc_q : -1;
   // second level, all in flight together: the hit's feature slots and its locus' numbers (most lanes of a wave read
   // the same locus: one cache line)
   int64_t my_i0 = 0, my_i1 = 0, my_s0 = 0, my_s1 = 0;
   uint32_t my_form = 0;
   uint64_t my_adj = 0;
   if (a.locus_seg_ok) { // (uniform; idle lanes read locus 0 -- no branch around the loads, nothing waits for them here)
      const int lq = active ? my_loc : 0;
      my_form = a.locus_seg_ok[lq];
      my_i0 = a.iso_off[lq];
      my_i1 = a.iso_off[lq + 1];
      my_s0 = a.seg_off[lq];
      my_s1 = a.seg_off[lq + 1];
      my_adj = a.locus_adj[lq];
   }
   const bool wide = !__ballot(f0 + kExonBinRegFeats > n_feat);
   uint64_t codes;
   uint32_t L[kExonBinRegFeats], R[kExonBinRegFeats]; // (slots past the hit's last feature hold the next hits': every use asks i < nf)
   load_hit_slots(a, f0, n_feat, wide, codes, L, R);
   const bool is_long = nf > kExonBinRegFeats;
   const int nfr = is_long ? 0 : nf; // the features held in registers
   // M (x M)*: an odd count, MATCH (code 0) exactly at the even positions -- asked of the eight code bytes at once: the
   // bytes of the hit's own slots, "byte is non-zero" as bit 7 of every byte (the carry of 0x7f + the low seven bits,
   // or bit 7 itself)
   constexpr uint64_t kEvenBytes = 0x00ff00ff00ff00ffull, kOddTop = 0x8000800080008000ull, kLow7 = 0x7f7f7f7f7f7f7f7full;
   const uint64_t own = nfr >= 8 ? ~0ull : ((1ull << (8 * nfr)) - 1ull);
   const uint64_t cw = codes & own;
   const uint64_t nonzero = (((cw & kLow7) + kLow7) | cw) & ~kLow7;
   const bool regular = ((nfr & 1) != 0) & ((cw & kEvenBytes) == 0ull) & ((nonzero & kOddTop) == (own & kOddTop));
   if (a.span && active) {
      uint32_t sig = kHitSigSeed;
      uint64_t sp = 0;
      if (is_long) {
         for (int i = 0; i < nf; ++i) sig = hit_sig_step(sig, a.feat_left[f0 + i], a.feat_right[f0 + i]);
         sp = ((uint64_t)a.feat_left[f0] << 32) | a.feat_right[f0 + nf - 1];
      } else if (nf > 0) {
         uint32_t last_r = 0;
#pragma unroll
         for (int i = 0; i < kExonBinRegFeats; ++i) {
            const bool in = i < nf;
            if (!__ballot(in)) break; // (uniform: the wave's longest list of features is typically three of the eight slots)
            const uint32_t nx = hit_sig_step(sig, L[i], R[i]);
            sig = in ? nx : sig;
            last_r = in ? R[i] : last_r;
         }
         sp = ((uint64_t)L[0] << 32) | last_r;
      }
      a.span[hidx] = sp; // 0: a hit without features (it carries no position)
      a.fhash[hidx] = hit_sig_fold(sig);
   }
   BlockHit bh;
   bh.nb = regular ? (nfr + 1) / 2 : 0;
#pragma unroll
   for (int j = 0; j < kExonBinBlocks; ++j) {
      const bool in = j < bh.nb;
      bh.l[j] = in ? L[2 * j] : 0xffffffffu;
      bh.r[j] = in ? R[2 * j] : 0u;
      bh.intron[j] = in && j && (uint32_t)((cw >> (8 * (2 * j - 1 > 0 ? 2 * j - 1 : 0))) & 0xffull) == 1u;
   }
   // the segment-basis form also wants the blocks ascending and every INTRON connector to fill its gap exactly
   bool seg_regular = regular;
#pragma unroll
   for (int j = 1; j < kExonBinBlocks; ++j) {
      const bool in = j < bh.nb;
      seg_regular = seg_regular & (!in | (bh.l[j] > bh.r[j - 1]));
      seg_regular = seg_regular & (!(in & bh.intron[j]) | ((L[2 * j - 1] == bh.r[j - 1] + 1u) & (R[2 * j - 1] + 1u == bh.l[j])));
   }
   // hits without features are "regular" with no blocks: all-zero words, written by the wave form
   bool todo = active && (regular || nf == 0);
   bool per_lane = false; // a regular hit the segment-basis form does not take, in a locus that uses it
   for (int round = 0; round < kExonBinWaveLoci; ++round) {
      const uint64_t m = __ballot(todo);
      if (!m) break;
      const int src = __ffsll((long long)m) - 1; // the first lane of the round's locus hands its numbers over
      const int loc = __builtin_amdgcn_readlane(my_loc, src);
      const bool mine = todo && my_loc == loc;
      const uint32_t form = lane_bcast(my_form, src); // 0: exon walk; 1: <= 32 segments; 2: <= 64
      if (form == 3u) {
         // 65-128 segments or isoforms: the regular hits are exonbin_seg128_kernel's (launched behind this one); the others
         // take the per-lane walk below
         per_lane = per_lane || (mine && !(seg_regular || nf == 0));
      } else if (form) {
         per_lane = per_lane || (mine && !(seg_regular || nf == 0));
         const int64_t i0 = (int64_t)lane_bcast((uint64_t)my_i0, src), s0 = (int64_t)lane_bcast((uint64_t)my_s0, src);
         const int niso = (int)(lane_bcast((uint32_t)my_i1, src) - (uint32_t)i0), nseg = (int)(lane_bcast((uint32_t)my_s1, src) - (uint32_t)s0); // (<= 64 each)
         // (a uint32_t instantiation for the loci of up to 32 segments was measured twice, in round 3 and on this form of
         // the kernel: 4.18 ms either way on the chain sample -- one copy of the code)
         exonbin_locus_segbasis<uint64_t>(a, i0, niso, s0, nseg, lane_bcast(my_adj, src), mine && (seg_regular || nf == 0), bh, hidx);
      } else {
         exonbin_locus_uniform(a, loc, mine, bh, hidx, f0);
      }
      todo = todo && !mine;
   }
   // per-lane walk: more features than the registers hold, an irregular feature list, or a wave
   // spread over many small loci
   // (its features are read again from memory: keeping the register copy alive across the loop above costs 24
   // registers and two waves per SIMD of occupancy, for the few lanes that come here)
   if (active && (is_long || todo || per_lane || !(regular || nf == 0))) {
      MemHit m = {a.feat_code + f0, a.feat_left + f0, a.feat_right + f0, nf};
      exonbin_hit(a, hidx, m);
   }
}

// One hit per lane, a workgroup per 256 hits, no loop.  (Workgroups that loop over tiles were measured on the chain sample:
// 4.8 ms with five resident workgroups per CU, 4.3 ms with 160, against 4.2 ms for a workgroup per tile -- the short-lived
// waves are not what the kernel waits for; its vector instructions are, 78 % of the SIMDs' cycles.)
__global__ __launch_bounds__(256) void exonbin_kernel(ExonBinArgs a)
{
   const int64_t n_feat = scalar_ptr(a.feat_off)[a.n_hits]; // (an s_load beside the lanes' own offsets)
   exonbin_tile(a, (int64_t)blockIdx.x * blockDim.x + threadIdx.x, n_feat);
}

// ------------------------------------------------------------------ segment basis, 65-128 segments or isoforms
// The same statements as exonbin_locus_segbasis with masks of 128 bits: a lane holds two segments (s and s + 64), the
// uniform walk and the blocks' first / last segments pick the register by the index' bit 6.  A kernel of its own, launched
// behind exonbin_kernel only when the annotation has such a locus (key or compat words beyond two): the 64-bit form, which
// serves nearly every locus, keeps its registers and its code as they are.  Hits of other loci and hits the segment basis
// does not take (irregular feature lists, more than 7 features) return at once: exonbin_kernel has served them.
typedef unsigned __int128 u128;
__device__ __forceinline__ u128 make_u128(uint64_t lo, uint64_t hi) { return ((u128)hi << 64) | lo; }

__device__ __forceinline__ void exonbin_locus_seg128(const ExonBinArgs &a, int loc, bool mine, const BlockHit &h, int64_t hidx)
{
   const SB_AS4 int64_t *iso_off = scalar_ptr(a.iso_off), *seg_off = scalar_ptr(a.seg_off);
   const int64_t i0 = iso_off[loc];
   const int niso = (int)(iso_off[loc + 1] - i0);
   const int64_t s0 = seg_off[loc];
   const int nseg = (int)(seg_off[loc + 1] - s0);
   const u128 adj = make_u128(scalar_ptr(a.locus_adj)[loc], scalar_ptr(a.locus_adj_hi)[loc]);
   const int lane = (int)(threadIdx.x & 63u);
   uint32_t my_sl[2], my_sr[2];
#pragma unroll
   for (int w = 0; w < 2; ++w) {
      my_sl[w] = lane + 64 * w < nseg ? a.seg_left[s0 + lane + 64 * w] : 0xffffffffu;
      my_sr[w] = lane + 64 * w < nseg ? a.seg_right[s0 + lane + 64 * w] : 0xffffffffu; // (past the end: reaches anything)
   }
   const SB_AS4 uint64_t *MEM = scalar_ptr(a.iso_member), *STA = scalar_ptr(a.iso_start);
   const SB_AS4 uint64_t *MEMH = scalar_ptr(a.iso_member_hi), *STAH = scalar_ptr(a.iso_start_hi);
   const bool live = mine && h.nb > 0;
   uint32_t rmax = 0;
#pragma unroll
   for (int j = 0; j < kExonBinBlocks; ++j) rmax = (j < h.nb) ? max(rmax, h.r[j]) : rmax;
   const uint32_t lo = wave_min_u32(live ? h.l[0] : 0xffffffffu);
   const uint32_t hi = wave_max_u32(live ? rmax : 0u);
   const int nbmax = (int)wave_max_u32(live ? (uint32_t)h.nb : 0u);
   uint32_t n_cnt[kExonBinBlocks];
#pragma unroll
   for (int j = 0; j < kExonBinBlocks; ++j) n_cnt[j] = 0u;
   const uint64_t reach0 = __ballot(my_sr[0] >= lo), reach1 = __ballot(my_sr[1] >= lo);
   const int k_lo = nbmax > 0 ? (reach0 ? (int)__ffsll((long long)reach0) - 1 : 64 + (int)__ffsll((long long)reach1) - 1) : nseg;
   for (int sg = k_lo; sg < nseg; ++sg) {
      const uint32_t sl = sg < 64 ? lane_bcast(my_sl[0], sg) : lane_bcast(my_sl[1], sg - 64);
      if (sl > hi) break;
      const uint32_t sr = sg < 64 ? lane_bcast(my_sr[0], sg) : lane_bcast(my_sr[1], sg - 64);
#pragma unroll
      for (int j = 0; j < kExonBinBlocks; ++j) {
         if (j >= nbmax) break;
         n_cnt[j] += ((sr < h.l[j]) ? 1u : 0u) + ((sl <= h.r[j]) ? 0x10000u : 0u);
      }
   }
   bool valid = live;
   u128 need_member = 0, forbid_start = 0, need_start = 0, forbid_member = 0;
   constexpr u128 kOne = 1, kAll = ~(u128)0;
   uint32_t sb_prev = 0;
   bool ends_prev = false;
   // (blocks no lane of the wave has are skipped: they would change nothing -- every statement below is under `in` --, and
   // the typical wave's longest hit has two blocks of the four the loop is unrolled for)
#pragma unroll
   for (int j = 0; j < kExonBinBlocks; ++j) {
      if (j >= nbmax) break; // (uniform)
      const bool in = live & (j < h.nb);
      const uint32_t sa = (uint32_t)k_lo + (n_cnt[j] & 0xffffu), sb1 = (uint32_t)k_lo + (n_cnt[j] >> 16); // [sa, sb1)
      const bool any = in & (sb1 > sa);
      const uint32_t sb = any ? sb1 - 1u : 0u, sa_c = any ? sa : 0u;
      const uint32_t fl0 = lane_gather(my_sl[0], sa_c & 63u), fl1 = lane_gather(my_sl[1], sa_c & 63u);
      const uint32_t lr0 = lane_gather(my_sr[0], sb & 63u), lr1 = lane_gather(my_sr[1], sb & 63u);
      const uint32_t first_l = sa_c < 64u ? fl0 : fl1, last_r = sb < 64u ? lr0 : lr1;
      const u128 upto_sb = (sb >= 127u) ? kAll : (u128)((kOne << (sb + 1u)) - kOne);
      const u128 mask = any ? (u128)(upto_sb & ~(u128)((kOne << sa_c) - kOne)) : (u128)0; // bits sa .. sb
      const u128 inner = mask & (u128)(mask - kOne);                                      // all but the first
      valid = valid & (!in | (any & (h.l[j] >= first_l) & (h.r[j] <= last_r) & ((u128)(inner & ~adj) == (u128)0)));
      need_member |= mask;
      forbid_start |= inner;
      if (j > 0) {
         const bool intr = in & h.intron[j];
         valid = valid & (!intr | (any & ends_prev & (h.l[j] == first_l)));
         const u128 first_j = any ? (u128)(kOne << sa_c) : (u128)0, upto_prev = (sb_prev >= 127u) ? kAll : (u128)((kOne << (sb_prev + 1u)) - kOne);
         need_start |= intr ? first_j : (u128)0;
         forbid_member |= intr ? (u128)((u128)(first_j - kOne) & ~upto_prev) : (u128)0;
      }
      sb_prev = sb;
      ends_prev = any & (h.r[j] == last_r);
   }
   uint32_t *__restrict__ cout = a.compat + hidx * a.compat_words;
   uint32_t *__restrict__ kout = a.key + hidx * a.key_words;
   for (int w = 0; w < a.compat_words; ++w) {
      uint32_t word = 0;
      const int nbits = niso - 32 * w < 32 ? niso - 32 * w : 32;
      for (int b = 0; b < nbits; ++b) {
         const u128 m = make_u128(MEM[i0 + 32 * w + b], MEMH[i0 + 32 * w + b]), st = make_u128(STA[i0 + 32 * w + b], STAH[i0 + 32 * w + b]);
         const u128 bad = (need_member & ~m) | (forbid_start & st) | (need_start & ~st) | (forbid_member & m);
         word |= (valid & (bad == (u128)0) & (m != (u128)0)) ? (1u << b) : 0u;
      }
      if (mine) cout[w] = word;
   }
   const u128 keybits = live ? need_member : (u128)0;
   if (mine)
      for (int w = 0; w < a.key_words; ++w) kout[w] = w < 4 ? (uint32_t)(keybits >> (32 * w)) : 0u;
}

__global__ __launch_bounds__(256) void exonbin_seg128_kernel(ExonBinArgs a)
{
   const int64_t hidx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
   const bool active = hidx < a.n_hits;
   const int my_loc = active ? a.hit_locus[hidx] : -1;
   const bool big = active && a.locus_seg_ok[my_loc] == 3u;
   if (!__ballot(big)) return; // (the usual wave: no hit of such a locus)
   int64_t f0 = 0;
   int nf = 0;
   if (big) {
      f0 = a.feat_off[hidx];
      nf = (int)(a.feat_off[hidx + 1] - f0);
   }
   // the hit as exonbin_kernel reads it: M (x M)* with the MATCH blocks at the even positions, at most kExonBinRegFeats features
   const bool fits = big && nf <= kExonBinRegFeats;
   RegHit h;
   h.nf = fits ? nf : 0;
   bool regular = (h.nf & 1) != 0;
#pragma unroll
   for (int i = 0; i < kExonBinRegFeats; ++i) {
      const bool in = i < h.nf;
      h.c[i] = in ? a.feat_code[f0 + i] : (uint8_t)2;
      h.l[i] = in ? a.feat_left[f0 + i] : 0u;
      h.r[i] = in ? a.feat_right[f0 + i] : 0u;
      regular = regular & (!in | ((h.c[i] == 0) == ((i & 1) == 0)));
   }
   BlockHit bh;
   bh.nb = regular ? (h.nf + 1) / 2 : 0;
#pragma unroll
   for (int j = 0; j < kExonBinBlocks; ++j) {
      const bool in = j < bh.nb;
      bh.l[j] = in ? h.l[2 * j] : 0xffffffffu;
      bh.r[j] = in ? h.r[2 * j] : 0u;
      bh.intron[j] = in && j && h.c[2 * j - 1] == 1;
   }
   bool seg_regular = regular;
#pragma unroll
   for (int j = 1; j < kExonBinBlocks; ++j) {
      const bool in = j < bh.nb;
      seg_regular = seg_regular & (!in | (bh.l[j] > bh.r[j - 1]));
      seg_regular = seg_regular & (!(in & bh.intron[j]) | ((h.l[2 * j - 1] == bh.r[j - 1] + 1u) & (h.r[2 * j - 1] + 1u == bh.l[j])));
   }
   // exactly the hits exonbin_kernel left: regular for the segment basis (or without features), in a locus of form 3
   bool todo = big && ((fits && seg_regular) || nf == 0);
   for (int round = 0; round < 64; ++round) { // (one locus per round; a wave rarely spans two)
      const uint64_t m = __ballot(todo);
      if (!m) break;
      const int loc = __builtin_amdgcn_readlane(my_loc, __ffsll((long long)m) - 1);
      const bool mine = todo && my_loc == loc;
      exonbin_locus_seg128(a, loc, mine, bh, hidx);
      todo = todo && !mine;
   }
}

// spans and hashes alone (a caller that made the words elsewhere: sbgpu_bins_create_device)
__global__ __launch_bounds__(256) void hit_signature_kernel(int64_t n_hits, const int64_t *feat_off, const uint32_t *feat_left,
                                                            const uint32_t *feat_right, uint64_t *span, uint32_t *fhash)
{
   const int64_t stride = (int64_t)gridDim.x * blockDim.x;
   for (int64_t h = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; h < n_hits; h += stride) {
      const int64_t f0 = feat_off[h], f1 = feat_off[h + 1];
      uint32_t sig = kHitSigSeed;
      for (int64_t i = f0; i < f1; ++i) sig = hit_sig_step(sig, feat_left[i], feat_right[i]);
      span[h] = f1 > f0 ? (((uint64_t)feat_left[f0] << 32) | feat_right[f1 - 1]) : 0ull;
      fhash[h] = hit_sig_fold(sig);
   }
}

} // namespace sb
