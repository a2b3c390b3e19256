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
// segment lengths and their prefix sums sit in LDS for the data-dependent indexing of the
// >= 5-segment case.  pdf(fl) comes from a table built on the host exactly like
// InsertSize::emp_dist_pdf (src/read.cpp:274-297).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "em_device.h" // wave_group_sum: the 64-lane sum by DPP steps and one matrix instruction

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
   // [5] pdf_support_kernel: [0], [1] first and last fl with pdf[fl] != 0 -- the terms outside are exact zeros and are skipped;
   // [2], [3] first and last fl with pdf[fl] >= kBwTiny; [4] != 0: a tiny non-zero density lies strictly between them.
   // A pair whose fragment lengths stay inside [2]..[3] (and [4] == 0) cannot produce a subnormal intermediate; the
   // others take the loop that flushes them like the reference's FTZ arithmetic does.
   const int32_t *pdf_support;
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
// SL[m] / SR[m]: total length of the m leftmost / rightmost INNER segments (m = 0..nseg-2).
template <class SegPtr>
__device__ __forceinline__ int effective_len(SegPtr s, const int *SL, const int *SR, int nseg,
                                             uint32_t imask, int nimp, int inner, int fl, int rl)
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
   // >= 5 segments, :476-515.  The reference walks the start position i = 1..s[0] of the left
   // mate and counts the positions whose two mates cover exactly the non-implicit inner
   // segments: bp_last = fl - i - inner bases fall in the last segment (positions with
   // bp_last outside [1, s_last] are skipped by its int-vs-uint compare), the right mate then
   // covers kR(i) inner segments from the right (as long as rl - bp_last exceeds their running
   // length) and the left mate kL(i) from the left (as long as rl - i does).  kR is
   // non-decreasing and kL non-increasing in i, so every condition is an interval of i and the
   // count is interval arithmetic on the prefix sums SL / SR -- identical to the scan
   // (checked against the reference on 180 000 cases), O(segments) instead of O(s_last*segments).
   const int ni = nseg - 2;
   if ((imask & 1u) || ((imask >> (nseg - 1)) & 1u)) return 0; // ends are never implicit
   const int B = fl - inner;
   int i_lo = B - (int)s[nseg - 1];
   if (i_lo < 1) i_lo = 1;
   int i_hi = B - 1;
   if (i_hi > (int)s[0]) i_hi = (int)s[0];
   if (i_lo > i_hi) return 0;
   const int BIG = 1 << 30;
   // i with kL(i) == c:  [rl - SL[c], rl - SL[c-1] - 1]   (open-ended at c == ni / c == 0)
   // i with kR(i) >= m:  i >= SR[m-1] - rl + B + 1         (m >= 1)
   if (imask != 0u) {
      const int a = __ffs(imask) - 1;          // first implicit inner segment
      const int b = 31 - __clz(imask);         // last one
      const uint32_t run = imask >> a;
      if (run & (run + 1u)) return 0;          // not one contiguous block: no position matches
      const int cL = a - 1, cR = ni - b;       // required kL and kR
      int lo = i_lo, hi = i_hi;
      if (cL >= 1) hi = min(hi, rl - SL[cL - 1] - 1);
      if (cL < ni) lo = max(lo, rl - SL[cL]);
      if (cR >= 1) lo = max(lo, SR[cR - 1] - rl + B + 1);
      if (cR < ni) hi = min(hi, SR[cR] - rl + B);
      return max(0, hi - lo + 1);
   }
   // nothing implicit: the prefix and the suffix must cover all inner segments, kL + kR >= ni
   int total = 0;
   for (int c = 0; c <= ni; ++c) {
      int lo = i_lo, hi = i_hi;
      if (c >= 1) hi = min(hi, rl - SL[c - 1] - 1);
      if (c < ni) lo = max(lo, rl - SL[c]);
      const int m = ni - c;
      if (m >= 1) lo = max(lo, SR[m - 1] - rl + B + 1);
      total += max(0, hi - lo + 1);
   }
   (void)BIG;
   return total;
}

// (pdf * eff) / (L - fl + 1): the denominator is a small positive integer, so the v_rcp_f64
// seed + two Newton steps + one residual correction give the correctly rounded quotient in all
// but a vanishing fraction of cases (<= 1 ulp), without the IEEE div_scale / div_fixup tail
// that costs more than the whole effective-length evaluation.  A non-positive denominator
// (inconsistent input) falls back to the IEEE division so that inf / NaN come out as they would.
__device__ __forceinline__ double bw_div(double n, double d)
{
   if (!(d > 0.0)) return n / d;
   double r = __builtin_amdgcn_rcp(d);
   double e = __builtin_fma(-d, r, 1.0);
   r = __builtin_fma(r, e, r);
   e = __builtin_fma(-d, r, 1.0);
   r = __builtin_fma(r, e, r);
   const double q = n * r;
   return __builtin_fma(__builtin_fma(-d, q, n), r, q);
}

// A density at or above this cannot lead to a subnormal product or quotient: pdf * eff >= pdf (eff >= 1, or the term is an
// exact 0), and the quotient by L - fl + 1 < 2^31 stays above 1e-290.
constexpr double kBwTiny = 1e-280;
// The reference is built -Ofast: FTZ / DAZ.  Its term `pdf * le_eff / (L - fl + 1)` (estimate.cpp:225) is flushed to 0 where
// the product or the quotient comes out subnormal; the sum of the (normal or zero) terms is subnormal only through a cancellation of two of them that the tests have never seen.
// Only densities below 1e-290 -- a fragment length some thirty standard deviations out -- get there, but a weight that is
// exactly 0 in the reference must not be 1e-310 here (the -f table prints it).
__device__ __forceinline__ double bw_flush(double x) { return __builtin_fabs(x) < 2.2250738585072014e-308 ? 0.0 : x; } // (an effective length can be negative: isoform.h:105-129)

// first and last index of a non-zero density, and of a density >= kBwTiny (one workgroup); an all-zero table gives the
// empty ranges [1, 0]
__global__ __launch_bounds__(256) void pdf_support_kernel(const double *pdf, int n, int32_t *out)
{
   __shared__ int lo, hi, blo, bhi, inner;
   if (threadIdx.x == 0) lo = blo = 0x7fffffff, hi = bhi = -1, inner = 0;
   __syncthreads();
   int mylo = 0x7fffffff, myhi = -1, myblo = 0x7fffffff, mybhi = -1;
   for (int i = threadIdx.x; i < n; i += 256) {
      const double v = pdf[i];
      if (v != 0.0) {
         mylo = min(mylo, i);
         myhi = max(myhi, i);
      }
      if (v >= kBwTiny) {
         myblo = min(myblo, i);
         mybhi = max(mybhi, i);
      }
   }
   if (myhi >= 0) {
      atomicMin(&lo, mylo);
      atomicMax(&hi, myhi);
   }
   if (mybhi >= 0) {
      atomicMin(&blo, myblo);
      atomicMax(&bhi, mybhi);
   }
   __syncthreads();
   int tiny_inside = 0;
   for (int i = threadIdx.x; i < n; i += 256) {
      const double v = pdf[i];
      if (v != 0.0 && v < kBwTiny && i > blo && i < bhi) tiny_inside = 1;
   }
   if (tiny_inside) atomicOr(&inner, 1);
   __syncthreads();
   if (threadIdx.x == 0) {
      out[0] = hi >= 0 ? lo : 1;
      out[1] = hi >= 0 ? hi : 0;
      out[2] = bhi >= 0 ? blo : 1;
      out[3] = bhi >= 0 ? bhi : 0;
      out[4] = inner;
   }
}

constexpr int kBinWeightMaxSeg = 32; // the reference's `1u << idx` masks stop at 32 segments too

__device__ __forceinline__ int bw_bcast(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
__device__ __forceinline__ int64_t bw_bcast(int64_t v, int lane)
{
   return (int64_t)(((uint64_t)(uint32_t)bw_bcast((int)(v >> 32), lane) << 32) | (uint32_t)bw_bcast((int)v, lane));
}

__global__ __launch_bounds__(64) void binweight_kernel(BinWeightArgs a)
{
   __shared__ uint32_t s_seg[kBinWeightMaxSeg];
   __shared__ int s_SL[kBinWeightMaxSeg], s_SR[kBinWeightMaxSeg];
   const int lane = threadIdx.x;
   // IEEE mode on purpose: the fp64 division below needs denormal support to be exact.
   // The reference's FTZ arithmetic is mirrored where it is observable: in the pdf table
   // (subnormal densities are 0, sbgpu_insert_pdf_table) and, for the pairs whose fragment
   // lengths reach the table's tiny tail, term by term (bw_flush).
   // One wave per pair, lanes over the fragment lengths -- but a wave takes a BATCH of 64 consecutive pairs: lane i
   // reads pair i's description (offsets, isoform length, implicit mask, target, its first four segment lengths) with
   // coalesced loads, two round trips for the 64 pairs, and the pairs are then served one after the other from those
   // registers (v_readlane).  Read per pair through the scalar cache, the description was three dependent round trips
   // in front of every pair's arithmetic and the kernel sat 60 % of its cycles waiting.  Results collect in the lanes and
   // leave with one store.
   const int64_t n_batches = (a.n_pairs + 63) / 64;
   for (int64_t bt = blockIdx.x; bt < n_batches; bt += gridDim.x) {
      const int64_t pm = bt * 64 + lane;
      const bool in = pm < a.n_pairs;
      const int64_t my_off = in ? a.seg_off[pm] : 0;
      const int my_nseg = in ? (int)(a.seg_off[pm + 1] - my_off) : 0;
      const int my_L = in ? a.iso_len[pm] : 1;
      const int my_imask = in ? (int)a.implicit_mask[pm] : 0;
      int my_s[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) my_s[k] = (in && !a.long_read && my_nseg <= 4 && k < my_nseg) ? (int)a.seg_lens[my_off + k] : 0;
      double my_acc = 0.0;
      const int npb = (int)(a.n_pairs - bt * 64 < 64 ? a.n_pairs - bt * 64 : 64);
      for (int q = 0; q < npb; ++q) {
         const int nseg = bw_bcast(my_nseg, q), L = bw_bcast(my_L, q);
         double acc = 0.0;
         if (a.long_read) { // estimate.cpp:236-247
            acc = 1.0 / (double)L;
         } else if (nseg <= 4) {
            // closed forms (1-4 segments, nearly all pairs): the segment lengths are wave-uniform scalars, no LDS,
            // no barrier
            const uint32_t s4[4] = {(uint32_t)bw_bcast(my_s[0], q), (uint32_t)bw_bcast(my_s[1], q), (uint32_t)bw_bcast(my_s[2], q),
                                    (uint32_t)bw_bcast(my_s[3], q)};
            const uint32_t imask = (uint32_t)bw_bcast(my_imask, q);
            const int nimp = __popc(imask);
            const int lmax = (int)(s4[0] + s4[1] + s4[2] + s4[3]);
            const int inner = nseg > 2 ? lmax - (int)s4[0] - (int)s4[nseg > 0 ? nseg - 1 : 0] : 0;
            int lmin = a.lmin_base;                // estimate.cpp:214-219
            if (nseg > 2) lmin = max(lmin, inner); // :220-221
            const int f0 = max(lmin, a.pdf_support[0]), f1 = min(lmax, a.pdf_support[1]);
            if (f0 >= a.pdf_support[2] && f1 <= a.pdf_support[3] && !a.pdf_support[4]) {
               for (int fl = f0 + lane; fl <= f1; fl += 64) { // :223-227, lanes over fl
                  const int e = effective_len(s4, (const int *)nullptr, (const int *)nullptr, nseg, imask, nimp, inner, fl, a.read_len);
                  acc += bw_div(a.pdf[fl] * (double)e, (double)(L - fl + 1));
               }
            } else { // fragment lengths in the table's tiny tail: subnormal intermediates are flushed like the reference's
               for (int fl = f0 + lane; fl <= f1; fl += 64) {
                  const int e = effective_len(s4, (const int *)nullptr, (const int *)nullptr, nseg, imask, nimp, inner, fl, a.read_len);
                  acc += bw_flush(bw_div(bw_flush(a.pdf[fl] * (double)e), (double)(L - fl + 1)));
               }
            }
            acc = wave_group_sum<64>(acc);
         } else {
            const int64_t off = bw_bcast(my_off, q);
            __syncthreads(); // the previous pair's readers are done with s_seg
            if (lane < nseg) s_seg[lane] = a.seg_lens[off + lane];
            __syncthreads();
            if (lane < nseg - 1) {
               // prefix sums over the inner segments 1 .. nseg-2, from the left and from the right
               int sl = 0, sr = 0;
               for (int m = 1; m <= lane; ++m) {
                  sl += (int)s_seg[m];
                  sr += (int)s_seg[nseg - 1 - m];
               }
               s_SL[lane] = sl;
               s_SR[lane] = sr;
            }
            __syncthreads();
            const uint32_t imask = (uint32_t)bw_bcast(my_imask, q);
            const int nimp = __popc(imask);
            int lmax = 0, inner = 0;
            for (int k = 0; k < nseg; ++k) {
               lmax += (int)s_seg[k];
               if (k >= 1 && k < nseg - 1) inner += (int)s_seg[k];
            }
            int lmin = a.lmin_base;                // estimate.cpp:214-219
            if (nseg > 2) lmin = max(lmin, inner); // :220-221
            const int f0 = max(lmin, a.pdf_support[0]), f1 = min(lmax, a.pdf_support[1]);
            if (f0 >= a.pdf_support[2] && f1 <= a.pdf_support[3] && !a.pdf_support[4]) {
               for (int fl = f0 + lane; fl <= f1; fl += 64) { // :223-227, lanes over fl
                  const int e = effective_len((const uint32_t *)s_seg, s_SL, s_SR, nseg, imask, nimp, inner, fl, a.read_len);
                  acc += bw_div(a.pdf[fl] * (double)e, (double)(L - fl + 1));
               }
            } else { // (see above)
               for (int fl = f0 + lane; fl <= f1; fl += 64) {
                  const int e = effective_len((const uint32_t *)s_seg, s_SL, s_SR, nseg, imask, nimp, inner, fl, a.read_len);
                  acc += bw_flush(bw_div(bw_flush(a.pdf[fl] * (double)e), (double)(L - fl + 1)));
               }
            }
            // wave sum (order differs from the reference's sequential loop by rounding only)
            acc = wave_group_sum<64>(acc);
         }
         my_acc = lane == q ? acc : my_acc;
      }
      if (in) a.out[a.out_index ? a.out_index[pm] : pm] = my_acc;
   }
}

} // namespace sb
