// strawberry_amd/csrc/bins_device.h -- grouping the hits of a locus into exon bins on the GPU
// (SURVEY 8(a) A5): LocusContext::set_maps (/root/reference/include/estimate.hpp:29-52) and
// ExonBin::read_count (include/isoform.h:285-296) for the common case.
//
// One workgroup per locus.  A hash table in LDS maps a hit's key words (the set of exon segments
// it touches, exonbin_device.h) to a bin; bins are then ranked by their first hit (the reference
// numbers them in order of first appearance, UniqPushAndReturnIdx), every hit gets its bin, and
// the bin's compat union and mass are accumulated with atomics.
//
// Exactness.  The reference keeps a bin's fragments in a std::set<Contig> ordered by the
// (offset, length) feature sequence and adds their float masses in that order.  Two things make
// the order irrelevant here, and the kernel VERIFIES both and reports when they do not hold
// (the caller then groups on the host, locus_bins.cpp):
//   - hits of a locus come sorted by (left end, right end) -- HitCluster's own order -- so equal
//     feature sequences are neighbours: a hit is dropped as a duplicate iff an earlier hit of its
//     (left, right) run has the same sequence and the same bin;
//   - all masses are whole numbers below 2^24 per bin (no multi-mapped reads: the reference's
//     default), so the float accumulation is exact in any order.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sb {

// Two table sizes: loci of up to 256 hits (most of them) take 1024 slots -- 18 KB of LDS, so that many
// workgroups share a CU and hide each other's memory latency -- the others 8192 (140 KB, one per CU).
constexpr int kBinsSlotsSmall = 1024, kBinsMaxSmall = 256;   // at most one bin per hit
constexpr int kBinsSlotsMid = 2048, kBinsMaxMid = 1400;      // larger loci first try this one (36 KB: four workgroups per CU);
                                                             // a locus with more bins than that is redone with the big table
constexpr int kBinsSlotsBig = 8192, kBinsMaxBig = 5600;      // ~0.7 load
constexpr int kBinsSmallHits = 256;
constexpr int kBinsThreads = 256;      // the small table's workgroup
constexpr int kBinsThreadsMid = 512;   // the two-pass form's
constexpr int kBinsThreadsAccum = 1024; // the single-pass form's middle table: two workgroups of 16 waves per CU
constexpr int kBinsThreadsBig = 1024;  // the big table fills a CU's LDS: one workgroup per CU, so it brings 16 waves
enum : int32_t { kBinsUnsorted = 1, kBinsFractional = 2, kBinsTableFull = 4, kBinsMassOverflow = 8, kBinsRunTooLong = 128 };

struct BinsArgs {
   int64_t n_loci;               // loci in `loci`
   const int32_t *loci;          // the loci this launch serves (small or big ones)
   const int64_t *locus_hit_off; // [all loci + 1]: hits are grouped by locus
   const int64_t *feat_off;
   const uint32_t *feat_left, *feat_right;
   const float *mass;
   int32_t compat_words, key_words;
   const uint32_t *compat, *key;
   const uint64_t *span;    // [n_hits] first left end << 32 | last right end; 0 for a hit without features (exonbin_device.h)
   const uint32_t *fhash;   // [n_hits] hash of the (left, right) sequence
   // hit-indexed scratch: bin b of locus l lives at index locus_hit_off[l] + b
   int32_t *hit_bin_local;  // [n_hits] rank of the hit's bin inside its locus, -1: dropped
   int32_t *bin_rep;        // [n_hits] a member hit of the bin (its key words are the bin's)
   int32_t *bin_count;      // [n_hits] zeroed by the caller
   uint32_t *bin_compat;    // [n_hits * compat_words] zeroed by the caller
   uint8_t *dup;            // [n_hits] 1: an equal fragment of the same bin came earlier (std::set keeps that one)
   int32_t *n_bins;         // [n_loci]
   int32_t *n_used;         // [n_loci] hits that landed in a bin
   int32_t *flags;          // [1] OR of kBins*
};

__device__ __forceinline__ uint32_t bins_hash(const uint32_t *k, int kw)
{
   uint64_t h = 1469598103934665603ull;
   for (int w = 0; w < kw; ++w) h = (h ^ k[w]) * 1099511628211ull;
   h ^= h >> 29;
   return (uint32_t)h | 1u; // never 0: a zero tag means "free"
}

// slot of the bin with these key words; INSERT: claim a free slot if there is none.
// A slot's tag = hash << 32 | (member hit - first hit of the locus + 1), written by one atomicCAS, so a
// reader never sees a claimed slot without its member.
template <bool INSERT>
__device__ __forceinline__ int bins_find(unsigned long long *tag, int slots, const BinsArgs &a, int64_t q0, int64_t h,
                                         const uint32_t *kh)
{
   const int kw = a.key_words;
   const uint32_t hv = bins_hash(kh, kw);
   const unsigned long long mine = ((unsigned long long)hv << 32) | (unsigned long long)(h - q0 + 1);
   int slot = (int)(hv >> 1) & (slots - 1);
   for (int probe = 0; probe < slots; ++probe) {
      unsigned long long t = tag[slot];
      if (t == 0ull && INSERT) {
         t = atomicCAS(&tag[slot], 0ull, mine);
         if (t == 0ull) return slot; // claimed
      }
      if (t == 0ull) return -1;
      if ((uint32_t)(t >> 32) == hv) {
         const uint32_t *kr = a.key + (q0 + (int64_t)(uint32_t)t - 1) * kw;
         bool same = true;
         for (int w = 0; w < kw; ++w) same &= kr[w] == kh[w];
         if (same) return slot;
      }
      slot = (slot + 1) & (slots - 1);
   }
   return -1;
}

__device__ __forceinline__ bool bins_same_fragment(const BinsArgs &a, int64_t x, int64_t y)
{
   const int64_t fx = a.feat_off[x], fy = a.feat_off[y];
   const int64_t n = a.feat_off[x + 1] - fx;
   if (a.feat_off[y + 1] - fy != n) return false;
   for (int64_t i = 0; i < n; ++i)
      if (a.feat_left[fx + i] != a.feat_left[fy + i] || a.feat_right[fx + i] != a.feat_right[fy + i]) return false;
   return true;
}

// RETRY: a locus whose bins do not fit this table is not an error -- it is marked (n_bins = -1, nothing else
// written that the retry does not overwrite) and the caller runs it again with the next larger table.
template <int kBinsSlots, int kBinsMaxPerLocus, int kThreads, bool RETRY = false>
__global__ __launch_bounds__(kThreads) void bins_locus_kernel(BinsArgs a)
{
   __shared__ unsigned long long tag[kBinsSlots];
   __shared__ int first[kBinsSlots]; // first hit of the bin (min), later its rank
   __shared__ int used[kBinsMaxPerLocus], used_rank[kBinsMaxPerLocus];
   // A bin's mass and compat union are accumulated in LDS and written once (hundreds of hits of a locus add to the
   // same few bins: as global atomics those adds serialise at the memory side).  The big table leaves no LDS for
   // this, and compat unions of more than two words do not fit either: those cases keep the global atomics.
   constexpr bool kLdsAcc = kBinsSlots <= kBinsSlotsMid;
   __shared__ int acc_count[kLdsAcc ? kBinsMaxPerLocus : 1];
   __shared__ unsigned acc_compat[kLdsAcc ? 2 * kBinsMaxPerLocus : 1];
   __shared__ int n_used_slots, n_hits_in, bad;
   const int tid = threadIdx.x;
   for (int64_t li = blockIdx.x; li < a.n_loci; li += gridDim.x) {
      const int64_t l = a.loci[li];
      const int64_t q0 = a.locus_hit_off[l], q1 = a.locus_hit_off[l + 1];
      // the table is as large as the locus needs (at most one bin per hit, load <= 1/4): small loci neither
      // clear nor scan 4096 slots
      int slots = 64;
      while (slots < kBinsSlots && slots < 4 * (q1 - q0)) slots <<= 1;
      for (int s = tid; s < slots; s += kThreads) {
         tag[s] = 0ull;
         first[s] = 0x7fffffff;
      }
      if (tid == 0) n_used_slots = 0, n_hits_in = 0, bad = 0;
      __syncthreads();
      const int cw = a.compat_words, kw = a.key_words;
      // ---- pass 1: every hit that has a compatible isoform enters the table
      int my_bad = 0;
      for (int64_t h = q0 + tid; h < q1; h += kThreads) {
         const int64_t f0 = a.feat_off[h], f1 = a.feat_off[h + 1];
         if (h > q0 && f1 > f0) { // sorted by (left, right)?  (empty hits carry no position)
            const int64_t g0 = a.feat_off[h - 1], g1 = f0;
            if (g1 > g0) {
               const uint32_t pl = a.feat_left[g0], pr = a.feat_right[g1 - 1], cl = a.feat_left[f0], cr = a.feat_right[f1 - 1];
               if (pl > cl || (pl == cl && pr > cr)) my_bad |= kBinsUnsorted;
            }
         }
         uint32_t any_c = 0, any_k = 0;
         for (int w = 0; w < cw; ++w) any_c |= a.compat[h * cw + w];
         for (int w = 0; w < kw; ++w) any_k |= a.key[h * kw + w];
         if (!any_c || !any_k) continue;
         const float m = a.mass[h];
         if (!(m >= 0.0f && m < 16777216.0f && (float)(int)m == m)) my_bad |= kBinsFractional;
         const int slot = bins_find<true>(tag, slots, a, q0, h, a.key + h * kw);
         if (slot < 0) {
            my_bad |= kBinsTableFull;
            continue;
         }
         atomicMin(&first[slot], (int)(h - q0));
      }
      if (my_bad) atomicOr(&bad, my_bad);
      __syncthreads();
      // ---- the bins, ranked by their first hit
      for (int s = tid; s < slots; s += kThreads)
         if (tag[s] != 0ull) {
            const int k = atomicAdd(&n_used_slots, 1);
            if (k < kBinsMaxPerLocus) used[k] = s;
         }
      __syncthreads();
      int nb = n_used_slots;
      bool overflow = nb > kBinsMaxPerLocus || (bad & kBinsTableFull);
      if (overflow) {
         if (tid == 0 && !RETRY) atomicOr(&bad, (int)kBinsTableFull);
         nb = 0; // the locus is not written; the flag sends the whole batch to the host (or, RETRY, to the next table)
      }
      for (int k = tid; k < nb; k += kThreads) {
         const int fk = first[used[k]];
         int r = 0;
         for (int v = 0; v < nb; ++v) r += first[used[v]] < fk; // first hits are distinct: ranks are a permutation
         used_rank[k] = r;
      }
      __syncthreads();
      const bool lds_acc = kLdsAcc && cw <= 2;
      for (int k = tid; k < nb; k += kThreads) {
         const int s = used[k];
         a.bin_rep[q0 + used_rank[k]] = (int32_t)(uint32_t)tag[s] - 1; // a member hit, relative to q0
         first[s] = used_rank[k];                                      // from now on: the bin's rank
         if (lds_acc) {
            acc_count[k] = 0;
            acc_compat[2 * k] = acc_compat[2 * k + 1] = 0u;
         }
      }
      __syncthreads();
      // ---- pass 2: every hit learns its bin; the first of equal fragments adds its mass
      int my_used = 0;
      for (int64_t h = q0 + tid; h < q1 && nb > 0; h += kThreads) {
         a.hit_bin_local[h] = -1;
         uint32_t any_c = 0, any_k = 0;
         for (int w = 0; w < cw; ++w) any_c |= a.compat[h * cw + w];
         for (int w = 0; w < kw; ++w) any_k |= a.key[h * kw + w];
         if (!any_c || !any_k) continue;
         const int slot = bins_find<false>(tag, slots, a, q0, h, a.key + h * kw);
         if (slot < 0) continue; // table was full
         const int b = first[slot];
         a.hit_bin_local[h] = b;
         ++my_used;
         for (int w = 0; w < cw; ++w) {
            const uint32_t cv = a.compat[h * cw + w];
            if (!cv) continue;
            if (lds_acc) {
               if ((acc_compat[2 * b + w] & cv) != cv) atomicOr(&acc_compat[2 * b + w], cv);
            } else {
               atomicOr(&a.bin_compat[(q0 + b) * cw + w], cv);
            }
         }
         // std::set<Contig>: an equal fragment already in this bin?  Only its (left, right) run can hold one.
         const int64_t f0 = a.feat_off[h], f1 = a.feat_off[h + 1];
         const uint32_t cl = a.feat_left[f0], cr = a.feat_right[f1 - 1];
         bool dup = false;
         for (int64_t p = h - 1; p >= q0 && !dup; --p) {
            const int64_t g0 = a.feat_off[p], g1 = a.feat_off[p + 1];
            if (g1 <= g0) continue;
            if (a.feat_left[g0] != cl || a.feat_right[g1 - 1] != cr) break;
            if (!bins_same_fragment(a, p, h)) continue;
            uint32_t pc = 0, pk = 0;
            for (int w = 0; w < cw; ++w) pc |= a.compat[p * cw + w];
            for (int w = 0; w < kw; ++w) pk |= a.key[p * kw + w];
            if (!pc || !pk) continue;
            const int ps = bins_find<false>(tag, slots, a, q0, p, a.key + p * kw);
            dup = ps >= 0 && first[ps] == b;
         }
         a.dup[h] = dup ? 1 : 0;
         if (!dup) {
            if (lds_acc) atomicAdd(&acc_count[b], (int)a.mass[h]);
            else atomicAdd(&a.bin_count[q0 + b], (int)a.mass[h]);
         }
      }
      if (lds_acc) {
         __syncthreads();
         for (int b = tid; b < nb; b += kThreads) {
            a.bin_count[q0 + b] = acc_count[b];
            for (int w = 0; w < cw; ++w) a.bin_compat[(q0 + b) * cw + w] = acc_compat[2 * b + w];
         }
      }
      if (nb == 0)
         for (int64_t h = q0 + tid; h < q1; h += kThreads) a.hit_bin_local[h] = -1;
      if (my_used) atomicAdd(&n_hits_in, my_used);
      __syncthreads();
      if (tid == 0) {
         a.n_bins[l] = (RETRY && overflow) ? -1 : nb;
         a.n_used[l] = n_hits_in;
         if (RETRY && overflow) bad &= ~(int)kBinsTableFull;
         if (bad) atomicOr(a.flags, bad);
      }
      __syncthreads();
   }
}


// ------------------------------------------------------------------ the single-pass form (round 3)
// The two-pass kernel above reads a hit's feature offsets and end coordinates twice (gathers: ~150 bytes of traffic
// per hit) and finds the bin of every earlier hit of a (left, right) run to apply the std::set<Contig> rule.  Two
// observations remove both: (i) fragments with equal feature sequences have equal key and compat words, hence the
// same bin, so "an equal fragment already in this bin" is "an equal fragment earlier in the run" -- a question about
// hits alone; (ii) the exon-bin kernel, which has every hit's features in registers anyway, can leave the run key
// (span) and a hash of the sequence per hit.  This kernel then reads 25 coalesced bytes per hit -- key, compat, mass,
// span, hash -- ONCE: the table entry of a hit's key accumulates mass and compat union by SLOT while the bins are
// still unranked, the ranking follows, and the per-bin results are written from the slots.  Features are touched
// only to confirm a hash match (i.e. for true duplicates).  hit -> bin is a kernel of its own (bins_assign_kernel),
// run only where somebody wants it.  Compat unions of up to two words; wider loci and loci of more bins than the
// middle table holds keep the two-pass kernel.
// The table of this form holds the key words themselves (at most two: 64 bits, never 0 for a hit that enters), so a
// probe compares in LDS and never reads a member hit's words from global memory.
__device__ __forceinline__ int bins_slot(unsigned long long *keyw, int slots, unsigned long long key, bool insert)
{
   unsigned long long x = key * 0x9E3779B97F4A7C15ull;
   int slot = (int)(x >> 40) & (slots - 1);
   for (int probe = 0; probe < slots; ++probe) {
      unsigned long long t = keyw[slot];
      if (t == key) return slot;
      if (t == 0ull) {
         if (!insert) return -1;
         t = atomicCAS(&keyw[slot], 0ull, key);
         if (t == 0ull || t == key) return slot;
      }
      slot = (slot + 1) & (slots - 1);
   }
   return -1;
}

constexpr int kBinsBatch = 4; // hits a thread has in flight per trip: their loads are issued together

template <int kBinsSlots, int kBinsMaxPerLocus, int kThreads, int CW, bool RETRY = false>
__global__ __launch_bounds__(kThreads) void bins_accum_kernel(BinsArgs a)
{
   __shared__ unsigned long long keyw[kBinsSlots];
   __shared__ int first[kBinsSlots];          // first hit of the slot's bin (min)
   __shared__ int acc_count[kBinsSlots];      // mass of the slot's bin
   __shared__ unsigned acc_compat[CW * kBinsSlots];
   __shared__ int used[kBinsMaxPerLocus], used_rank[kBinsMaxPerLocus];
   __shared__ int n_used_slots, n_hits_in, bad;
   const int tid = threadIdx.x;
   const int kw = a.key_words; // 1 or 2; compat words: CW
   for (int64_t li = blockIdx.x; li < a.n_loci; li += gridDim.x) {
      const int64_t l = a.loci[li];
      const int64_t q0 = a.locus_hit_off[l], q1 = a.locus_hit_off[l + 1];
      int slots = 64;
      while (slots < kBinsSlots && slots < 4 * (q1 - q0)) slots <<= 1;
      for (int s = tid; s < slots; s += kThreads) {
         keyw[s] = 0ull;
         first[s] = 0x7fffffff;
         acc_count[s] = 0;
         for (int w = 0; w < CW; ++w) acc_compat[CW * s + w] = 0u;
      }
      if (tid == 0) n_used_slots = 0, n_hits_in = 0, bad = 0;
      __syncthreads();
      int my_bad = 0, my_used = 0;
      for (int64_t base = q0 + tid; base < q1; base += (int64_t)kBinsBatch * kThreads) {
         // ---- the batch's loads, all in flight together
         uint64_t sp[kBinsBatch], sp1[kBinsBatch];
         uint32_t fh[kBinsBatch], fh1[kBinsBatch], cv[kBinsBatch][CW];
         unsigned long long key[kBinsBatch];
         float m[kBinsBatch];
#pragma unroll
         for (int i = 0; i < kBinsBatch; ++i) {
            const int64_t h = base + (int64_t)i * kThreads;
            const bool in = h < q1;
            const int64_t hc = in ? h : q0;
            sp[i] = in ? a.span[hc] : 0ull;
            fh[i] = a.fhash[hc];
            const int64_t hp = hc > q0 ? hc - 1 : q0;
            sp1[i] = hc > q0 ? a.span[hp] : 0ull;
            fh1[i] = a.fhash[hp];
            for (int w = 0; w < CW; ++w) cv[i][w] = in ? a.compat[hc * CW + w] : 0u;
            key[i] = in ? (kw == 2 ? (((unsigned long long)a.key[hc * 2 + 1] << 32) | a.key[hc * 2]) : (unsigned long long)a.key[hc]) : 0ull;
            m[i] = a.mass[hc];
         }
#pragma unroll
         for (int i = 0; i < kBinsBatch; ++i) {
            const int64_t h = base + (int64_t)i * kThreads;
            if (h >= q1) break;
            if (sp[i] && sp1[i] > sp[i]) my_bad |= kBinsUnsorted; // sorted by (left, right)?  (hits without features carry no position: 0)
            uint32_t any_c = 0;
            for (int w = 0; w < CW; ++w) any_c |= cv[i][w];
            if (!any_c || !key[i]) {
               a.dup[h] = 0;
               continue;
            }
            if (!(m[i] >= 0.0f && m[i] < 16777216.0f && (float)(int)m[i] == m[i])) my_bad |= kBinsFractional;
            const int slot = bins_slot(keyw, slots, key[i], true);
            if (slot < 0) {
               my_bad |= kBinsTableFull;
               a.dup[h] = 0;
               continue;
            }
            if (first[slot] > (int)(h - q0)) atomicMin(&first[slot], (int)(h - q0));
            ++my_used;
            for (int w = 0; w < CW; ++w)
               if (cv[i][w] && (acc_compat[CW * slot + w] & cv[i][w]) != cv[i][w]) atomicOr(&acc_compat[CW * slot + w], cv[i][w]);
            // std::set<Contig>: an equal fragment earlier in this (left, right) run?  (equal fragments share bin and words)
            bool dup = false;
            if (sp1[i] == sp[i] || sp1[i] == 0ull) { // (the common case -- the hit before starts another run -- costs no load)
               for (int64_t p = h - 1; p >= q0 && !dup; --p) {
                  const uint64_t ps = p == h - 1 ? sp1[i] : a.span[p];
                  if (ps != sp[i]) {
                     if (ps == 0ull) continue; // a hit without features sits anywhere
                     break;
                  }
                  const uint32_t pf = p == h - 1 ? fh1[i] : a.fhash[p];
                  dup = pf == fh[i] && bins_same_fragment(a, p, h);
               }
            }
            a.dup[h] = dup ? 1 : 0;
            if (!dup) atomicAdd(&acc_count[slot], (int)m[i]);
         }
      }
      if (my_bad) atomicOr(&bad, my_bad);
      if (my_used) atomicAdd(&n_hits_in, my_used);
      __syncthreads();
      // ---- the bins, ranked by their first hit
      for (int s = tid; s < slots; s += kThreads)
         if (keyw[s] != 0ull) {
            const int k = atomicAdd(&n_used_slots, 1);
            if (k < kBinsMaxPerLocus) used[k] = s;
         }
      __syncthreads();
      int nb = n_used_slots;
      const bool overflow = nb > kBinsMaxPerLocus || (bad & kBinsTableFull);
      if (overflow) {
         if (tid == 0 && !RETRY) atomicOr(&bad, (int)kBinsTableFull);
         nb = 0; // the locus is not written; the flag sends the batch to the host (or, RETRY, the locus to the two-pass kernel)
      }
      for (int k = tid; k < nb; k += kThreads) {
         const int fk = first[used[k]];
         int r = 0;
         for (int v = 0; v < nb; ++v) r += first[used[v]] < fk; // first hits are distinct: ranks are a permutation
         used_rank[k] = r;
      }
      __syncthreads();
      for (int k = tid; k < nb; k += kThreads) {
         const int s = used[k], r = used_rank[k];
         a.bin_rep[q0 + r] = first[s]; // a member hit (the first), relative to q0
         a.bin_count[q0 + r] = acc_count[s];
         for (int w = 0; w < CW; ++w) a.bin_compat[(q0 + r) * CW + w] = acc_compat[CW * s + w];
      }
      if (tid == 0) {
         a.n_bins[l] = (RETRY && overflow) ? -1 : nb;
         a.n_used[l] = n_hits_in;
         if (RETRY && overflow) bad &= ~(int)kBinsTableFull;
         if (bad) atomicOr(a.flags, bad);
      }
      __syncthreads();
   }
}

// hit -> rank of its bin inside the locus (-1: in no bin), from the bins' member hits: the table is rebuilt from the
// nb keys, then every hit looks its key up.  Key words: at most two.
template <int kBinsSlots, int kThreads>
__global__ __launch_bounds__(kThreads) void bins_assign_kernel(BinsArgs a)
{
   __shared__ unsigned long long keyw[kBinsSlots];
   __shared__ int rank_of[kBinsSlots];
   const int tid = threadIdx.x;
   const int cw = a.compat_words, kw = a.key_words;
   auto key_of = [&](int64_t h) -> unsigned long long {
      return kw == 2 ? (((unsigned long long)a.key[h * 2 + 1] << 32) | a.key[h * 2]) : (unsigned long long)a.key[h];
   };
   for (int64_t li = blockIdx.x; li < a.n_loci; li += gridDim.x) {
      const int64_t l = a.loci[li];
      const int64_t q0 = a.locus_hit_off[l], q1 = a.locus_hit_off[l + 1];
      const int nb = a.n_bins[l];
      if (nb <= 0) {
         for (int64_t h = q0 + tid; h < q1; h += kThreads) a.hit_bin_local[h] = -1;
         continue;
      }
      int slots = 64;
      while (slots < kBinsSlots && slots < 4 * nb) slots <<= 1;
      for (int s = tid; s < slots; s += kThreads) keyw[s] = 0ull;
      __syncthreads();
      for (int b = tid; b < nb; b += kThreads) {
         const int slot = bins_slot(keyw, slots, key_of(q0 + a.bin_rep[q0 + b]), true);
         if (slot >= 0) rank_of[slot] = b;
      }
      __syncthreads();
      for (int64_t h = q0 + tid; h < q1; h += kThreads) {
         uint32_t any_c = 0;
         for (int w = 0; w < cw; ++w) any_c |= a.compat[h * cw + w];
         const unsigned long long key = key_of(h);
         int b = -1;
         if (any_c && key) {
            const int slot = bins_slot(keyw, slots, key, false);
            if (slot >= 0) b = rank_of[slot];
         }
         a.hit_bin_local[h] = b;
      }
      __syncthreads();
   }
}

// ------------------------------------------------------------------ fractional masses
// With multi-mapped reads (--allow-multimapped-hits) a hit's mass is a sum of 1, 1/2, 1/3, ...; ExonBin::read_count
// (/root/reference/include/isoform.h:285-296) adds the masses of a bin's distinct fragments IN FLOAT, in the order
// of its std::set<Contig> -- Contig::operator< (src/contig.cpp:342-347): the feature lists compared (offset, length)
// by (offset, length), a proper prefix first -- and LocusContext truncates the sum to int (src/estimate.cpp:288).
// Float addition is not associative and the truncation can turn a last-bit difference into a different count, so
// the order is reproduced: hits come sorted by their left end, which is also the set order's leading key, so the
// set order differs from the hit order only INSIDE a run of hits with the same left end.  One thread per bin walks
// the locus' hits (every thread of the workgroup reads the same hit at the same time: the loads broadcast),
// keeps the members of its bin of the current run, and when the run ends sorts those few by the set's order and
// adds their masses.  A bin with more than kBinsRunCap members in one run raises a flag (the caller then groups on
// the host).
constexpr int kBinsRunCap = 48;

__device__ __forceinline__ int bins_frag_cmp(const BinsArgs &a, int64_t x, int64_t y)
{
   const int64_t fx = a.feat_off[x], fy = a.feat_off[y];
   const int64_t nx = a.feat_off[x + 1] - fx, ny = a.feat_off[y + 1] - fy, n = nx < ny ? nx : ny;
   for (int64_t i = 0; i < n; ++i) {
      const uint32_t lx = a.feat_left[fx + i], ly = a.feat_left[fy + i];
      if (lx != ly) return lx < ly ? -1 : 1;
      const uint32_t wx = a.feat_right[fx + i] - lx, wy = a.feat_right[fy + i] - ly;
      if (wx != wy) return wx < wy ? -1 : 1;
   }
   return nx == ny ? 0 : (nx < ny ? -1 : 1);
}

__global__ __launch_bounds__(256) void bins_ordered_mass_kernel(BinsArgs a)
{
   for (int64_t li = blockIdx.x; li < a.n_loci; li += gridDim.x) {
      const int64_t l = a.loci[li];
      const int64_t q0 = a.locus_hit_off[l], q1 = a.locus_hit_off[l + 1];
      const int nb = a.n_bins[l];
      for (int b0 = 0; b0 < nb; b0 += 256) {
         const int b = b0 + (int)threadIdx.x; // this thread's bin (threads beyond nb walk along idle)
         int run[kBinsRunCap];
         int n_run = 0, too_long = 0;
         uint32_t run_left = 0;
         float sum = 0.0f;
         auto flush = [&]() {
            for (int k = 0; k < n_run; ++k) sum += a.mass[q0 + run[k]]; // the set's order inside the run
            n_run = 0;
         };
         for (int64_t h = q0; h < q1; ++h) {
            const int hb = a.hit_bin_local[h]; // workgroup-uniform loads: every thread looks at the same hit
            if (hb < 0 || a.dup[h]) continue;
            if (hb != b) continue;
            const uint32_t left = a.feat_left[a.feat_off[h]];
            if (n_run && left != run_left) flush();
            run_left = left;
            if (n_run == kBinsRunCap) {
               too_long = 1;
               continue;
            }
            // stable insertion by Contig::operator<: behind every member that is not greater
            int pos = n_run;
            while (pos > 0 && bins_frag_cmp(a, q0 + run[pos - 1], h) > 0) {
               run[pos] = run[pos - 1];
               --pos;
            }
            run[pos] = (int)(h - q0);
            ++n_run;
         }
         flush();
         if (b < nb) a.bin_count[q0 + b] = (int)sum; // (int) float: estimate.cpp:288
         if (too_long) atomicOr(a.flags, (int)kBinsRunTooLong);
      }
   }
}

// bins of all loci into their final, dense arrays (row_off = scan of n_bins, done on the host)
struct BinsPackArgs {
   int64_t n_loci;
   const int64_t *locus_hit_off, *row_off;
   int32_t compat_words, key_words;
   const uint32_t *key;
   const int32_t *hit_bin_local, *bin_rep, *bin_count_in;
   const uint32_t *bin_compat_in;
   int32_t *count;       // [n_bins]
   uint32_t *bin_key;    // [n_bins * key_words]
   uint32_t *bin_compat; // [n_bins * compat_words]
   int64_t *hit_bin;     // [n_hits] global bin or -1 (may be null)
   int32_t *flags;
};

__global__ __launch_bounds__(256) void bins_pack_kernel(BinsPackArgs a)
{
   for (int64_t l = blockIdx.x; l < a.n_loci; l += gridDim.x) {
      const int64_t q0 = a.locus_hit_off[l], q1 = a.locus_hit_off[l + 1], b0 = a.row_off[l];
      const int nb = (int)(a.row_off[l + 1] - b0);
      for (int b = threadIdx.x; b < nb; b += blockDim.x) {
         const int32_t c = a.bin_count_in[q0 + b];
         if (c >= 16777216) atomicOr(a.flags, (int)kBinsMassOverflow);
         a.count[b0 + b] = c;
         const int64_t rep = q0 + a.bin_rep[q0 + b];
         for (int w = 0; w < a.key_words; ++w) a.bin_key[(b0 + b) * a.key_words + w] = a.key[rep * a.key_words + w];
         for (int w = 0; w < a.compat_words; ++w)
            a.bin_compat[(b0 + b) * a.compat_words + w] = a.bin_compat_in[(q0 + b) * a.compat_words + w];
      }
      if (a.hit_bin)
         for (int64_t h = q0 + threadIdx.x; h < q1; h += blockDim.x)
            a.hit_bin[h] = a.hit_bin_local[h] < 0 ? -1 : b0 + a.hit_bin_local[h];
   }
}

// the count and compat entries of the listed loci, zeroed (before bins_locus_kernel's global atomics run on loci whose
// entries a single-pass kernel owned before)
__global__ __launch_bounds__(256) void bins_zero_loci_kernel(BinsArgs a)
{
   for (int64_t i = blockIdx.x; i < a.n_loci; i += gridDim.x) {
      const int64_t l = a.loci[i], q0 = a.locus_hit_off[l], q1 = a.locus_hit_off[l + 1];
      for (int64_t h = q0 + threadIdx.x; h < q1; h += blockDim.x) {
         a.bin_count[h] = 0;
         for (int w = 0; w < a.compat_words; ++w) a.bin_compat[h * a.compat_words + w] = 0;
      }
   }
}

// ------------------------------------------------------------------ (bin, isoform) pairs
// ExonBin::bin_under_iso (/root/reference/include/isoform.h:363-411) for every pair LocusContext::
// set_theory_bin_weight visits (src/estimate.cpp:203-213): the isoform's segments from the bin's first to
// its last, and which of the inner ones the bin does not hold ("implicit": under the mate gap).
// One lane per isoform, walking the bins of its locus: the pairs come out ordered by (isoform, bin), the
// order of the host code.  Pass 1 counts, pass 2 (after bins_scan_kernel<1>) fills.
enum : int32_t { kPairsNotUnder = 16, kPairsForeignSegment = 32, kPairsWide = 64 };

struct PairsArgs {
   int64_t n_iso;
   const int32_t *iso_locus;      // [n_iso]
   const int64_t *iso_off;        // [n_loci + 1]
   const int64_t *row_off, *f_off; // [n_loci + 1]
   const int64_t *seg_off;        // [n_loci + 1]
   const uint32_t *seg_left, *seg_right;
   const int64_t *iso_seg_off;    // [n_iso + 1]: Isoform::_exon_segs as local segment indices
   const int32_t *iso_seg_idx;
   const int32_t *iso_len;        // [n_iso]
   int32_t compat_words, key_words;
   const uint32_t *bin_key, *bin_compat;
   int32_t *pair_cnt, *seg_cnt;        // pass 1: [n_iso]
   const int64_t *pair_off, *pseg_off; // pass 2: their exclusive scans, [n_iso + 1]
   int64_t *pair_seg_off;
   uint32_t *pair_seg_lens, *pair_mask;
   int32_t *pair_iso_len;
   int64_t *pair_out_index;
   int32_t *flags;
   int32_t only_wide_loci; // thread-per-isoform kernel: serve only loci of more than 64 isoforms (the wave-per-locus kernel has the others)
};

// Exclusive prefix sums between the grouping's kernels, so that they follow each other in the stream without a
// host round trip.  MODE 0, over the loci: the bin counts -> row_off, bins x isoforms -> f_off (a count < 0 -- a
// locus the middle table could not hold, redone later -- stands for 0).  MODE 1, over the isoforms: the pair and
// segment counts -> pair_off, pseg_off.  Three small launches: tile sums (4096 entries per workgroup, 16 per lane:
// one cache line of a lane's own), the scan of the tile sums (one workgroup), the tiles again with their bases.
// 250 000 isoforms: 61 tiles, ~15 us together -- against a D2H, a host loop and an H2D.
constexpr int kScanTile = 4096, kScanPerLane = 16;

// Zero fill, 16 bytes per lane and step (`bytes` a multiple of 16, `p` 16-byte aligned): the runtime's
// hipMemsetAsync kernel reaches 60 GB/s on this -- 0.2 ms for the 12 MB of 1.5 M hits -- this one the memory's rate.
__global__ __launch_bounds__(256) void bins_zero_kernel(uint4 *p, int64_t n16)
{
   const int64_t stride = (int64_t)gridDim.x * blockDim.x;
   for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) p[i] = make_uint4(0, 0, 0, 0);
}

struct ScanArgs {
   int64_t n;
   const int32_t *a, *b;     // MODE 0: a = bins per locus (b unused); MODE 1: pair counts, segment counts
   const int64_t *iso_off;   // MODE 0
   int64_t *out_a, *out_b;   // [n + 1]
   int64_t *part_a, *part_b; // [tiles]: tile sums, then their exclusive scan
   int64_t *totals;          // [2]
};

template <int MODE>
__device__ __forceinline__ void scan_value(const ScanArgs &s, int64_t i, int64_t &va, int64_t &vb)
{
   if (MODE == 0) {
      va = s.a[i] > 0 ? s.a[i] : 0;
      vb = va * (s.iso_off[i + 1] - s.iso_off[i]);
   } else {
      va = s.a[i];
      vb = s.b[i];
   }
}

// sums over the workgroup's 256 lanes, inclusive per lane, through LDS (8 steps)
__device__ __forceinline__ void scan_block256(int64_t *sa, int64_t *sb_, int t, int64_t &pa, int64_t &pb)
{
   sa[t] = pa;
   sb_[t] = pb;
   __syncthreads();
   for (int d = 1; d < 256; d <<= 1) {
      const int64_t xa = t >= d ? sa[t - d] : 0, xb = t >= d ? sb_[t - d] : 0;
      __syncthreads();
      sa[t] += xa;
      sb_[t] += xb;
      __syncthreads();
   }
   pa = sa[t];
   pb = sb_[t];
}

template <int MODE>
__global__ __launch_bounds__(256) void bins_scan_tiles_kernel(ScanArgs s)
{
   __shared__ int64_t sa[256], sb_[256];
   const int t = threadIdx.x;
   const int64_t i0 = (int64_t)blockIdx.x * kScanTile + (int64_t)t * kScanPerLane;
   int64_t pa = 0, pb = 0;
   for (int k = 0; k < kScanPerLane; ++k)
      if (i0 + k < s.n) {
         int64_t va, vb;
         scan_value<MODE>(s, i0 + k, va, vb);
         pa += va;
         pb += vb;
      }
   scan_block256(sa, sb_, t, pa, pb);
   if (t == 255) {
      s.part_a[blockIdx.x] = pa;
      s.part_b[blockIdx.x] = pb;
   }
}

// the tile sums -> their exclusive scan, in place; the grand totals -> out[n] and totals[0..1]
__global__ __launch_bounds__(256) void bins_scan_parts_kernel(ScanArgs s, int64_t n_tiles)
{
   __shared__ int64_t sa[256], sb_[256];
   const int t = threadIdx.x;
   int64_t base_a = 0, base_b = 0;
   for (int64_t c0 = 0; c0 < n_tiles; c0 += 256) { // (256 tiles = 10^6 entries per round)
      const int64_t i = c0 + t;
      const int64_t va = i < n_tiles ? s.part_a[i] : 0, vb = i < n_tiles ? s.part_b[i] : 0;
      int64_t pa = va, pb = vb;
      scan_block256(sa, sb_, t, pa, pb);
      if (i < n_tiles) {
         s.part_a[i] = base_a + pa - va;
         s.part_b[i] = base_b + pb - vb;
      }
      base_a += sa[255];
      base_b += sb_[255];
      __syncthreads();
   }
   if (t == 0) {
      s.out_a[s.n] = base_a;
      s.out_b[s.n] = base_b;
      s.totals[0] = base_a;
      s.totals[1] = base_b;
   }
}

template <int MODE>
__global__ __launch_bounds__(256) void bins_scan_apply_kernel(ScanArgs s)
{
   __shared__ int64_t sa[256], sb_[256];
   const int t = threadIdx.x;
   const int64_t i0 = (int64_t)blockIdx.x * kScanTile + (int64_t)t * kScanPerLane;
   int64_t va[kScanPerLane], vb[kScanPerLane];
   int64_t pa = 0, pb = 0;
#pragma unroll
   for (int k = 0; k < kScanPerLane; ++k) {
      va[k] = vb[k] = 0;
      if (i0 + k < s.n) scan_value<MODE>(s, i0 + k, va[k], vb[k]);
      pa += va[k];
      pb += vb[k];
   }
   const int64_t own_a = pa, own_b = pb;
   scan_block256(sa, sb_, t, pa, pb);
   int64_t ra = s.part_a[blockIdx.x] + pa - own_a, rb = s.part_b[blockIdx.x] + pb - own_b;
#pragma unroll
   for (int k = 0; k < kScanPerLane; ++k)
      if (i0 + k < s.n) {
         s.out_a[i0 + k] = ra;
         s.out_b[i0 + k] = rb;
         ra += va[k];
         rb += vb[k];
      }
}

template <bool FILL>
__global__ __launch_bounds__(256) void bins_pairs_kernel(PairsArgs a)
{
   const int64_t stride = (int64_t)gridDim.x * blockDim.x;
   if (FILL && !a.only_wide_loci && blockIdx.x == 0 && threadIdx.x == 0) a.pair_seg_off[a.pair_off[a.n_iso]] = a.pseg_off[a.n_iso]; // the CSR's last entry
   for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n_iso; i += stride) {
      const int l = a.iso_locus[i];
      const int64_t i0 = a.iso_off[l];
      const int j = (int)(i - i0), niso = (int)(a.iso_off[l + 1] - i0);
      if (a.only_wide_loci && niso <= 64) continue;
      const int64_t b0 = a.row_off[l], b1 = a.row_off[l + 1], s0 = a.seg_off[l];
      const int32_t *is = a.iso_seg_idx + a.iso_seg_off[i];
      const int nis = (int)(a.iso_seg_off[i + 1] - a.iso_seg_off[i]);
      const int cw = a.compat_words, kw = a.key_words;
      int np = 0, ns = 0, bad = 0;
      int64_t p = FILL ? a.pair_off[i] : 0, so = FILL ? a.pseg_off[i] : 0;
      for (int64_t b = b0; b < b1; ++b) {
         if (!((a.bin_compat[b * cw + (j >> 5)] >> (j & 31)) & 1u)) continue;
         const uint32_t *key = a.bin_key + b * kw;
         int bf = -1, bl = -1, nbits = 0; // first / last segment of the bin, number of its segments
         for (int w = 0; w < kw; ++w) {
            const uint32_t k = key[w];
            if (!k) continue;
            if (bf < 0) bf = 32 * w + __ffs(k) - 1;
            bl = 32 * w + 31 - __clz(k);
            nbits += __popc(k);
         }
         // isoform.h:381-391: lower_bound of the bin's first and last segment among the isoform's
         int low = 0, up = 0;
         {
            int lo = 0, hi = nis;
            while (lo < hi) {
               const int mid = (lo + hi) >> 1;
               if (is[mid] < bf) lo = mid + 1;
               else hi = mid;
            }
            low = lo;
            hi = nis;
            while (lo < hi) {
               const int mid = (lo + hi) >> 1;
               if (is[mid] < bl) lo = mid + 1;
               else hi = mid;
            }
            up = lo;
         }
         if (low >= nis || up >= nis || up < low) {
            bad |= kPairsNotUnder;
            continue;
         }
         const int n = up - low + 1;
         uint32_t mask = 0;
         int n_eff = n;
         if (n > 32) {
            n_eff = 0; // see locus_bins.cpp: such a pair carries no segments (long-read workflow only)
            bad |= kPairsWide;
         } else {
            // :393-409: an inner isoform segment the bin does not hold is implicit; every segment the bin
            // holds between its ends must be one of the isoform's
            int inside = 0;
            for (int q = 1; q + 1 < n; ++q) {
               const int sidx = is[low + q];
               if ((key[sidx >> 5] >> (sidx & 31)) & 1u) ++inside;
               else mask |= 1u << q;
            }
            // the bin's other segments (all but its first and last) must be the ones counted: a mismatch is
            // either the reference's assert(false) or a case its walk lets pass -- the host code decides
            if (nbits >= 2 && inside != nbits - 2) bad |= kPairsForeignSegment;
         }
         if (FILL) {
            a.pair_seg_off[p] = so;
            for (int q = 0; q < n_eff; ++q) {
               const int sidx = is[low + q];
               a.pair_seg_lens[so + q] = a.seg_right[s0 + sidx] - a.seg_left[s0 + sidx] + 1;
            }
            a.pair_mask[p] = mask;
            a.pair_iso_len[p] = a.iso_len[i];
            a.pair_out_index[p] = a.f_off[l] + (b - b0) * niso + j;
            ++p;
            so += n_eff;
         }
         ++np;
         ns += n_eff;
      }
      if (!FILL) {
         a.pair_cnt[i] = np;
         a.seg_cnt[i] = ns;
      }
      if (bad) atomicOr(a.flags, bad);
   }
}

// The same pairs, one WAVE per locus (loci of up to 64 isoforms: a lane per isoform holds its running offsets): lanes
// over the locus' bins, 64 at a time; a bin's key is analysed once (first / last segment, number of segments), not once
// per isoform; per isoform the lanes whose bin is compatible do the two lower_bounds, and a ballot / prefix count gives
// every pair its place, so the pairs of one isoform leave as consecutive words.  The thread-per-isoform kernel above
// reads every bin's words once per isoform with one lane per address and writes its pairs as partial lines (1 GB of
// traffic for 73 MB of pairs); it stays for loci of more isoforms.
__device__ __forceinline__ int wave_excl_scan(int v, int lane)
{
   int x = v;
#pragma unroll
   for (int d = 1; d < 64; d <<= 1) {
      const int y = __builtin_amdgcn_ds_bpermute(((lane - d) & 63) << 2, x);
      x += lane >= d ? y : 0;
   }
   return x - v;
}

template <bool FILL>
__global__ __launch_bounds__(256) void bins_pairs_locus_kernel(PairsArgs a, int64_t n_loci)
{
   const int lane = threadIdx.x & 63;
   const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
   const int cw = a.compat_words, kw = a.key_words;
   if (FILL && blockIdx.x == 0 && threadIdx.x == 0) a.pair_seg_off[a.pair_off[a.n_iso]] = a.pseg_off[a.n_iso]; // the CSR's last entry
   for (int64_t l = wave; l < n_loci; l += n_waves) {
      const int64_t i0 = a.iso_off[l];
      const int niso = (int)(a.iso_off[l + 1] - i0);
      if (niso > 64) continue; // (the launcher sends such loci to the other kernel)
      const int64_t b0 = a.row_off[l], b1 = a.row_off[l + 1], s0 = a.seg_off[l], f0 = a.f_off[l];
      // lane j: isoform i0 + j
      const bool has_iso = lane < niso;
      const int64_t my_seg0 = has_iso ? a.iso_seg_off[i0 + lane] : 0;
      const int my_nis = has_iso ? (int)(a.iso_seg_off[i0 + lane + 1] - my_seg0) : 0;
      const int my_len = has_iso ? a.iso_len[i0 + lane] : 0;
      int64_t my_p = (FILL && has_iso) ? a.pair_off[i0 + lane] : 0, my_so = (FILL && has_iso) ? a.pseg_off[i0 + lane] : 0;
      int my_np = 0, my_ns = 0, bad = 0;
      for (int64_t c0 = b0; c0 < b1; c0 += 64) {
         const int64_t b = c0 + lane;
         const bool has_bin = b < b1;
         // the bin's key, once: first / last segment and how many
         int bf = -1, bl = -1, nbits = 0;
         uint32_t k0 = 0, k1 = 0; // the first two key words (all there are for loci of up to 64 segments)
         for (int w = 0; w < kw; ++w) {
            const uint32_t k = has_bin ? a.bin_key[b * kw + w] : 0u;
            if (w == 0) k0 = k;
            if (w == 1) k1 = k;
            if (!k) continue;
            if (bf < 0) bf = 32 * w + __ffs(k) - 1;
            bl = 32 * w + 31 - __clz(k);
            nbits += __popc(k);
         }
         uint32_t c0w = (has_bin && cw > 0) ? a.bin_compat[b * cw] : 0u, c1w = (has_bin && cw > 1) ? a.bin_compat[b * cw + 1] : 0u;
         for (int j = 0; j < niso; ++j) {
            const bool set = has_bin && (((j < 32 ? c0w : c1w) >> (j & 31)) & 1u);
            const uint64_t m = __ballot(set);
            if (!m) continue;
            const int64_t seg0 = (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(my_seg0 >> 32), j) << 32) |
                                           (uint32_t)__builtin_amdgcn_readlane((int)my_seg0, j));
            const int nis = __builtin_amdgcn_readlane(my_nis, j);
            const int32_t *is = a.iso_seg_idx + seg0;
            int n_eff = 0, ok = 0;
            uint32_t mask = 0;
            int low = 0;
            if (set) {
               // isoform.h:381-391: lower_bound of the bin's first and last segment among the isoform's
               int lo = 0, hi = nis;
               while (lo < hi) {
                  const int mid = (lo + hi) >> 1;
                  if (is[mid] < bf) lo = mid + 1;
                  else hi = mid;
               }
               low = lo;
               hi = nis;
               while (lo < hi) {
                  const int mid = (lo + hi) >> 1;
                  if (is[mid] < bl) lo = mid + 1;
                  else hi = mid;
               }
               const int up = lo;
               if (low >= nis || up >= nis || up < low) {
                  bad |= kPairsNotUnder;
               } else {
                  ok = 1;
                  const int n = up - low + 1;
                  n_eff = n;
                  if (n > 32) {
                     n_eff = 0; // see locus_bins.cpp: such a pair carries no segments (long-read workflow only)
                     bad |= kPairsWide;
                  } else {
                     // :393-409: an inner isoform segment the bin does not hold is implicit; every segment the bin
                     // holds between its ends must be one of the isoform's
                     int inside = 0;
                     for (int q = 1; q + 1 < n; ++q) {
                        const int sidx = is[low + q];
                        const uint32_t kwd = sidx < 32 ? k0 : (sidx < 64 ? k1 : a.bin_key[b * kw + (sidx >> 5)]);
                        if ((kwd >> (sidx & 31)) & 1u) ++inside;
                        else mask |= 1u << q;
                     }
                     if (nbits >= 2 && inside != nbits - 2) bad |= kPairsForeignSegment;
                  }
               }
            }
            const uint64_t okm = __ballot(ok != 0);
            const int n_here = __popcll(okm);
            const int rank = __popcll(okm & ((1ull << lane) - 1ull));
            const int seg_before = wave_excl_scan(ok ? n_eff : 0, lane);
            const int seg_here = __builtin_amdgcn_readlane(seg_before + (ok ? n_eff : 0), 63);
            if (FILL) {
               const int len_j = __builtin_amdgcn_readlane(my_len, j);
               const int64_t pb = (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(my_p >> 32), j) << 32) |
                                            (uint32_t)__builtin_amdgcn_readlane((int)my_p, j));
               const int64_t sb = (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(my_so >> 32), j) << 32) |
                                            (uint32_t)__builtin_amdgcn_readlane((int)my_so, j));
               if (ok) {
                  const int64_t p = pb + rank, so = sb + seg_before;
                  a.pair_seg_off[p] = so;
                  for (int q = 0; q < n_eff; ++q) {
                     const int sidx = is[low + q];
                     a.pair_seg_lens[so + q] = a.seg_right[s0 + sidx] - a.seg_left[s0 + sidx] + 1;
                  }
                  a.pair_mask[p] = mask;
                  a.pair_iso_len[p] = len_j;
                  a.pair_out_index[p] = f0 + (b - b0) * niso + j;
               }
            }
            if (lane == j) {
               my_p += n_here, my_so += seg_here;
               my_np += n_here, my_ns += seg_here;
            }
         }
      }
      if (!FILL && has_iso) {
         a.pair_cnt[i0 + lane] = my_np;
         a.seg_cnt[i0 + lane] = my_ns;
      }
      if (bad) atomicOr(a.flags, bad);
   }
}

} // namespace sb
