// strawberry_amd/csrc/fraglen_device.h -- pass 1 of the reference on the device: the sample of the empirical insert-size law.
//
// Sample::fragLenDist (/root/reference/src/alignments.cpp:1363-1410): every unique hit of a cluster that is compatible with
// EXACTLY ONE of the cluster's transcripts contributes ONE fragment length -- Contig::exonic_overlaps_len(transcript, hit.left(),
// hit.right()) (/root/reference/src/contig.cpp:412-426): the transcript's exonic bases between the hit's two ends -- to
// ReadTable::_frag_dist, whatever the hit's collapse mass.  InsertSize(frag_lens) (/root/reference/src/read.cpp:238-262) then
// reads the sample's size, sum, sum of squares, extremes and histogram -- all of them functions of the HISTOGRAM alone, so the
// histogram (exact integers, any order) is all the device makes; sbgpu_quantify_* turns it into the law on the host.
//
// The compat words are the exon-bin kernel's (one bit per isoform of the hit's locus), the hit's two ends its `span` word
// (first left end << 32 | last right end, 0 for a hit without features: such a hit has ref_id -1 and is skipped, :1376-1379).
// 16 bytes per hit, coalesced; the isoform's exons are gathered (the annotation lives in L2).  A workgroup counts lengths
// below kFragLenLdsBins in LDS (nearly all of them land on a few hundred addresses: one global atomic per hit would queue at
// the L2) and adds its non-zero counters to the global histogram at its end.
#pragma once

#include "device_common.h"

namespace sb {

constexpr int kFragLenLdsBins = 8192; // 32 KB of LDS per workgroup

struct FragLenArgs {
   int64_t n_hits;
   int32_t compat_words;
   const int32_t *hit_locus;   // [n_hits]
   const uint32_t *compat;     // [n_hits * compat_words]
   const uint64_t *span;       // [n_hits]
   const int64_t *iso_off;     // [n_loci + 1]
   const int64_t *exon_off;    // [n_iso + 1]
   const uint32_t *exon_left, *exon_right;
   int64_t hist_len;           // fragment lengths 0 .. hist_len - 1
   unsigned long long *hist;   // [hist_len + 1], zeroed by the caller; [hist_len] counts lengths beyond the table (an error)
};

// exonic bases of isoform `iso` inside [left, right], a lane by itself (gathers)
__device__ __forceinline__ int64_t fraglen_by_lane(const FragLenArgs &a, int64_t iso, uint32_t left, uint32_t right)
{
   int64_t len = 0;
   for (int64_t e = a.exon_off[iso], e1 = a.exon_off[iso + 1]; e < e1; ++e) {
      const uint32_t xl = a.exon_left[e], xr = a.exon_right[e];
      if (xl <= right && left <= xr) len += (int64_t)min(xr, right) - (int64_t)max(xl, left) + 1; // GenomicFeature::overlap_len_in_genome
   }
   return len;
}

// A wave takes 64 CONSECUTIVE hits, a workgroup a contiguous range of the hits, trip after trip.  The hits come grouped by
// locus, so nearly every wave's hits share ONE locus -- and that locus is usually the previous trip's.  The wave keeps the
// locus' exon table in registers (lane e: exon e of the locus, lane j: where isoform j's exons begin; loci of up to 63
// isoforms and 64 exons) and walks it uniformly with v_readlane: the overlap arithmetic for all lanes at once, an isoform
// only when some lane's hit is compatible with it alone; the table is loaded when the locus changes.  Bigger loci walk
// the same way through scalar loads; a wave that straddles loci lets every lane gather for itself.  (One trip used to cost
// a chain of dependent loads -- hit -> locus -> isoform -> exons, 2.5 ms per 1.76e8 hits with one hit per lane and trip --
// now only its own words, loaded one trip ahead.)
constexpr int kFragLenThreads = 1024;

__global__ __launch_bounds__(kFragLenThreads) void fraglen_hist_kernel(FragLenArgs a)
{
   __shared__ unsigned lds[kFragLenLdsBins];
   for (int i = (int)threadIdx.x; i < kFragLenLdsBins; i += kFragLenThreads) lds[i] = 0u;
   __syncthreads();
   // this workgroup's hits: [h_begin, h_end), whole tiles of kFragLenThreads
   const int64_t tiles = (a.n_hits + kFragLenThreads - 1) / kFragLenThreads, per = (tiles + gridDim.x - 1) / gridDim.x;
   const int64_t h_begin = (int64_t)blockIdx.x * per * kFragLenThreads;
   const int64_t h_end = min(a.n_hits, h_begin + per * kFragLenThreads);
   const int lane = (int)(threadIdx.x & 63u);
   int64_t h = h_begin + threadIdx.x;
   // (0: no features -- Contig(hit).ref_id() == -1 -- or no hit)
   uint64_t sp_n = h < h_end ? a.span[h] : 0ull;
   int32_t loc_n = h < h_end ? a.hit_locus[h] : -1;
   uint32_t bits_n = h < h_end ? a.compat[h * a.compat_words] : 0u;
   // the table of the locus in hand
   int32_t t_loc = -1;
   int64_t t_j0 = 0, t_niso = 0;
   bool t_small = false;
   int t_eoff = 0;
   uint32_t t_xl = 0, t_xr = 0;
   for (; h - lane < h_end; h += kFragLenThreads) {
      const bool in = h < h_end;
      const uint64_t sp = sp_n;
      const int32_t loc = loc_n;
      const uint32_t bits0 = bits_n;
      {  // the next trip's words
         const int64_t hn = h + kFragLenThreads;
         sp_n = hn < h_end ? a.span[hn] : 0ull;
         loc_n = hn < h_end ? a.hit_locus[hn] : -1;
         bits_n = hn < h_end ? a.compat[hn * a.compat_words] : 0u;
      }
      const int32_t loc0 = __builtin_amdgcn_readfirstlane(loc); // (lane 0 of a wave that is here has a hit)
      const bool uniform = __ballot(in && loc != loc0) == 0ull;
      if (uniform && loc0 != t_loc) { // another locus: its table
         t_loc = loc0;
         t_j0 = a.iso_off[loc0];
         t_niso = a.iso_off[loc0 + 1] - t_j0;
         const int64_t ex0 = a.exon_off[t_j0], nex = a.exon_off[t_j0 + t_niso] - ex0;
         t_small = t_niso <= 63 && nex <= 64;
         if (t_small) {
            t_eoff = lane <= t_niso ? (int)(a.exon_off[t_j0 + lane] - ex0) : 0;
            t_xl = lane < nex ? a.exon_left[ex0 + lane] : 0u;
            t_xr = lane < nex ? a.exon_right[ex0 + lane] : 0u;
         }
      }
      const int64_t j0 = uniform ? t_j0 : (in ? a.iso_off[loc] : 0);
      const int64_t niso = uniform ? t_niso : ((in ? a.iso_off[loc + 1] : 0) - j0);
      // compatible with exactly one transcript (:1383-1392)
      int counter = 0;
      int mark = 0;
      for (int w = 0; w < a.compat_words; ++w) {
         uint32_t bits = w == 0 ? bits0 : (sp ? a.compat[h * a.compat_words + w] : 0u);
         const int64_t left_in_word = niso - 32 * (int64_t)w;
         if (left_in_word < 32) bits = left_in_word > 0 ? (bits & ((1u << left_in_word) - 1u)) : 0u;
         if (bits) {
            counter += __popc(bits);
            mark = 32 * w + (31 - __clz((int)bits));
         }
      }
      const bool one = sp != 0ull && counter == 1;
      const uint32_t left = (uint32_t)(sp >> 32), right = (uint32_t)sp;
      int64_t len = 0;
      if (uniform) {
         unsigned long long pending = __ballot(one);
         while (pending) {
            const int j = __builtin_amdgcn_readlane(mark, __ffsll((long long)pending) - 1); // an isoform somebody needs
            int64_t lj = 0;
            if (t_small) {
               const int e0 = __builtin_amdgcn_readlane(t_eoff, j), e1 = __builtin_amdgcn_readlane(t_eoff, j + 1);
               for (int e = e0; e < e1; ++e) {
                  const uint32_t xl = (uint32_t)__builtin_amdgcn_readlane((int)t_xl, e), xr = (uint32_t)__builtin_amdgcn_readlane((int)t_xr, e);
                  if (xl <= right && left <= xr) lj += (int64_t)min(xr, right) - (int64_t)max(xl, left) + 1; // GenomicFeature::overlap_len_in_genome
               }
            } else {
               const int64_t iso = j0 + j;
               const int64_t e0 = a.exon_off[iso], e1 = a.exon_off[iso + 1]; // (uniform: scalar loads)
               for (int64_t e = e0; e < e1; e += 4) {
                  uint32_t xl[4], xr[4];
#pragma unroll
                  for (int u = 0; u < 4; ++u) { // (an index past the isoform's last exon reads that exon again and is not counted)
                     const int64_t eu = e + u < e1 ? e + u : e1 - 1;
                     xl[u] = a.exon_left[eu], xr[u] = a.exon_right[eu];
                  }
#pragma unroll
                  for (int u = 0; u < 4; ++u)
                     if (e + u < e1 && xl[u] <= right && left <= xr[u]) lj += (int64_t)min(xr[u], right) - (int64_t)max(xl[u], left) + 1;
               }
            }
            const bool mine = one && mark == j;
            if (mine) len = lj;
            pending &= ~__ballot(mine);
         }
      } else if (one) {
         len = fraglen_by_lane(a, j0 + mark, left, right);
      }
      if (one) {
         if (len < kFragLenLdsBins) atomicAdd(&lds[len], 1u);
         else atomicAdd(&a.hist[len < a.hist_len ? len : a.hist_len], 1ull);
      }
   }
   __syncthreads();
   for (int i = (int)threadIdx.x; i < kFragLenLdsBins; i += kFragLenThreads) {
      const unsigned v = lds[i];
      if (v) atomicAdd(&a.hist[i < a.hist_len ? i : a.hist_len], (unsigned long long)v);
   }
}

} // namespace sb
