// strawberry_amd/csrc/binweight_device.h
//
// The bin-weight model that fills the EM matrix F: for every
// (exon bin, isoform) pair
//     F = sum_{fl = lmin}^{lmax} pdf(fl) * eff(fl) / (L - fl + 1)
// LocusContext::set_theory_bin_weight, /root/reference/src/estimate.cpp:201-234;
// ExonBin::effective_len (+ no_gap_ef / gap_ef), include/isoform.h:105-129,419-516.
// `eff` is integer arithmetic and must be exact; the sum is fp64.
//
// One wave per pair, lanes over fragment lengths fl; the pair's
// segment lengths sit in LDS for the data-dependent indexing of the >= 5-segment
// scan.  pdf(fl) comes from a table built on the host exactly like
// InsertSize::emp_dist_pdf (src/read.cpp:274-297).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sb {

struct BinWeightArgs {
   int64_t n_pairs;
   const int64_t *seg_off;        // [n_pairs+1] into seg_lens
   const uint32_t *seg_lens;      // lengths of the isoform's segments spanned by the bin
   const uint32_t *implicit_mask; // bit k: segment k lies in the mate gap (bin_under_iso, isoform.h:363-411)
   const int32_t *iso_len;        // exonic length L_j of the pair's isoform
   const int64_t *out_index;      // where the weight goes in `out` (NULL: out[pair])
   const double *pdf;             // pdf[fl], fl in [0, pdf_len)
   double *out;
   int32_t pdf_len;
   int32_t read_len;              // rl = read_len_mode()
   int32_t lmin_base;             // _use_emp ? _start_offset : rl   (estimate.cpp:214-219)
   int32_t long_read;             // set_bin_weight_without_frag_dist: F = 1/L (estimate.cpp:236-247)
};

// include/isoform.h:105-115
__device__ __forceinline__ int no_gap_ef(int l_left, int l_right, int l_int, int fl)
{
   if (fl < l_int + 2) return 0;
   if (fl > l_left + l_right + l_int) return 0;
   const int mid = fl - l_int - 1;
   return min(l_left, mid) + min(l_right, mid) - mid;
}

// include/isoform.h:117-129
__device__ __forceinline__ int gap_ef(int l_left, int l_right, int l_int, int rl, int gap)
{
   if (2 * rl + gap < l_int + 2) return 0;
   if (2 * rl + gap > l_left + l_right + l_int) return 0;
   const int start = max(rl, l_left + l_int - gap - 1);
   const int end = min(l_left, l_left + l_right + l_int - gap - rl);
   return max(0, end - start);
}

// include/isoform.h:419-516.  s = segment lengths (LDS), nseg >= 1.
__device__ __forceinline__ int effective_len(const uint32_t *s, int nseg, uint32_t imask, int nimp, int inner,
                                             int fl, int rl)
{
   const int gap = fl - 2 * rl;
   if (nseg == 1) return (int)(s[0] - (uint32_t)fl + 1u);          // :427-429 (uint arithmetic wraps)
   if (nseg == 2) return no_gap_ef((int)s[0], (int)s[1], 0, fl);   // :430-432
   if (nseg == 3) {                                                // :435-447
      const int g = gap_ef((int)s[0], (int)s[2], (int)s[1], rl, gap);
      if (nimp == 1) return g;
      if (nimp == 0) return no_gap_ef((int)s[0], (int)s[2], (int)s[1], fl) - g;
      return 0; // assert(false) in the reference
   }
   if (nseg == 4) {                                                // :448-475
      const int hit14 = gap_ef((int)s[0], (int)s[3], (int)(s[2] + s[1]), rl, gap);
      const int hit24 = gap_ef((int)s[3], (int)s[1], (int)s[2], rl, gap);
      const int hit124 = gap_ef((int)(s[0] + s[1]), (int)s[3], (int)s[2], rl, gap);
      const int hit13 = gap_ef((int)s[0], (int)s[2], (int)s[1], rl, gap);
      const int hit134 = gap_ef((int)s[0], (int)(s[2] + s[3]), (int)s[1], rl, gap);
      if (nimp == 0) {
         const int hit_all_124 = hit124 - hit14 - hit24;
         const int hit_all_134 = hit134 - hit14 - hit13;
         const int total = no_gap_ef((int)s[0], (int)s[3], (int)(s[1] + s[2]), fl);
         return total - hit_all_124 - hit_all_134 - hit14;
      }
      if (nimp == 2) return hit14;
      if (imask & 2u) return hit134 - hit14 - hit13; // implicit_idx[0] == 1
      return hit124 - hit14 - hit24;
   }
   // >= 5 segments, :476-515: count the start positions i in the first segment whose
   // mates cover exactly the non-implicit inner segments.  The reference walks
   // i = 1..s[0] and skips (int vs uint compare) every i with bp_last outside
   // [1, s_last], stopping at bp_last == 0; only that window is visited here.
   const uint32_t num_inners = (uint32_t)nseg - 2u;
   const uint32_t target = ((nseg >= 32) ? 0xFFFFFFFFu : ((1u << nseg) - 1u)) & ~imask;
   const int s_last = (int)s[nseg - 1];
   int i_lo = fl - inner - s_last; // bp_last == s_last
   if (i_lo < 1) i_lo = 1;
   int i_hi = fl - inner - 1;      // bp_last == 1
   if (i_hi > (int)s[0]) i_hi = (int)s[0];
   uint32_t num_pos = 0;
   for (int i = i_lo; i <= i_hi; ++i) {
      const int bp_last = fl - i - inner;
      uint32_t hit = 1u | (1u << (nseg - 1));
      // right-end cover
      int last_rest_bp = rl - bp_last;
      uint32_t j = num_inners;
      while (last_rest_bp > 0 && j > 0) {
         hit |= (1u << j);
         last_rest_bp = (int)((uint32_t)last_rest_bp - s[j]);
         j = j - 1;
      }
      // left-end cover
      int first_rest_bp = rl - i;
      j = 1;
      while (first_rest_bp > 0 && j <= num_inners) {
         hit |= (1u << j);
         first_rest_bp = (int)((uint32_t)first_rest_bp - s[j]);
         j = j + 1;
      }
      num_pos += (hit == target) ? 1u : 0u;
   }
   return (int)num_pos;
}

constexpr int kBinWeightMaxSeg = 32; // the reference's `1u << idx` masks stop at 32 segments too

__global__ __launch_bounds__(64) void binweight_kernel(BinWeightArgs a)
{
   __shared__ uint32_t s_seg[kBinWeightMaxSeg];
   const int lane = threadIdx.x;
   // IEEE mode on purpose: the fp64 division below needs denormal support to be exact.
   // The reference's FTZ arithmetic is mirrored where it is observable, in the pdf table
   // (subnormal densities are 0, sbgpu_insert_pdf_table).
   // one wave per pair, pairs strided over the grid (the caller interleaves heavy and
   // light pairs; a pair's cost is known only after reading its segments)
   for (int64_t p = blockIdx.x; p < a.n_pairs; p += gridDim.x) {
      const int64_t off = a.seg_off[p];
      const int nseg = (int)(a.seg_off[p + 1] - off);
      const int L = a.iso_len[p];
      const int64_t dst = a.out_index ? a.out_index[p] : p;
      double acc = 0.0;
      if (a.long_read) { // estimate.cpp:236-247
         acc = 1.0 / (double)L;
      } else {
         __syncthreads(); // the previous pair's readers are done with s_seg
         if (lane < nseg) s_seg[lane] = a.seg_lens[off + lane];
         __syncthreads();
         const uint32_t imask = a.implicit_mask[p];
         const int nimp = __popc(imask);
         int lmax = 0, inner = 0;
         for (int k = 0; k < nseg; ++k) {
            lmax += (int)s_seg[k];
            if (k >= 1 && k < nseg - 1) inner += (int)s_seg[k];
         }
         int lmin = a.lmin_base;                // estimate.cpp:214-219
         if (nseg > 2) lmin = max(lmin, inner); // :220-221
         for (int fl = lmin + lane; fl <= lmax; fl += 64) { // :223-227, lanes over fl
            const int e = effective_len(s_seg, nseg, imask, nimp, inner, fl, a.read_len);
            const double pdfv = (fl >= 0 && fl < a.pdf_len) ? a.pdf[fl] : 0.0;
            acc += pdfv * (double)e / (double)(L - fl + 1);
         }
         // wave sum (order differs from the reference's sequential loop by rounding only)
         for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m);
      }
      if (lane == 0) a.out[dst] = acc;
   }
}

} // namespace sb
