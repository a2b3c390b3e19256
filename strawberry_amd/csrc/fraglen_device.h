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

__global__ __launch_bounds__(256) void fraglen_hist_kernel(FragLenArgs a)
{
   __shared__ unsigned lds[kFragLenLdsBins];
   for (int i = (int)threadIdx.x; i < kFragLenLdsBins; i += 256) lds[i] = 0u;
   __syncthreads();
   const int64_t stride = (int64_t)gridDim.x * 256;
   for (int64_t h = (int64_t)blockIdx.x * 256 + threadIdx.x; h < a.n_hits; h += stride) {
      const uint64_t sp = a.span[h];
      if (sp == 0ull) continue; // no features: Contig(hit).ref_id() == -1
      const int32_t loc = a.hit_locus[h];
      const int64_t j0 = a.iso_off[loc], niso = a.iso_off[loc + 1] - j0;
      // compatible with exactly one transcript (:1383-1392)
      int counter = 0;
      int64_t mark = 0;
      for (int w = 0; w < a.compat_words; ++w) {
         uint32_t bits = a.compat[h * a.compat_words + w];
         const int64_t left_in_word = niso - 32 * (int64_t)w;
         if (left_in_word < 32) bits = left_in_word > 0 ? (bits & ((1u << left_in_word) - 1u)) : 0u;
         if (bits) {
            counter += __popc(bits);
            mark = 32 * (int64_t)w + (31 - __clz((int)bits));
         }
      }
      if (counter != 1) continue;
      const uint32_t left = (uint32_t)(sp >> 32), right = (uint32_t)sp;
      const int64_t iso = j0 + mark;
      int64_t len = 0;
      for (int64_t e = a.exon_off[iso], e1 = a.exon_off[iso + 1]; e < e1; ++e) {
         const uint32_t xl = a.exon_left[e], xr = a.exon_right[e];
         if (xl <= right && left <= xr) len += (int64_t)min(xr, right) - (int64_t)max(xl, left) + 1; // GenomicFeature::overlap_len_in_genome
      }
      if (len < kFragLenLdsBins) atomicAdd(&lds[len], 1u);
      else atomicAdd(&a.hist[len < a.hist_len ? len : a.hist_len], 1ull);
   }
   __syncthreads();
   for (int i = (int)threadIdx.x; i < kFragLenLdsBins; i += 256) {
      const unsigned v = lds[i];
      if (v) atomicAdd(&a.hist[i < a.hist_len ? i : a.hist_len], (unsigned long long)v);
   }
}

} // namespace sb
