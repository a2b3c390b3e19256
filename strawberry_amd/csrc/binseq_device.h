// strawberry_amd/csrc/binseq_device.h -- the per-bin sequence statistics of the `-f` table
// (SURVEY 8(a) A8): GC ratio, hexamer entropy and the four "high GC stretch" flags of a bin's
// sequence, i.e. of its segments' bases concatenated.
//   /root/reference/src/alignments.cpp:1622-1636   which statistics, in which order
//   /root/reference/include/isoform.h:173-182      the bin's sequence
//   /root/reference/include/kmer.h:14-135          SortedKmer / Entropy / GCRatio / HighGCStrech
//
// One wave (a 64-lane workgroup) per bin.  Every base is read from HBM once: the wave walks the
// concatenated sequence 64 positions at a time, each lane fetches one byte, and three ballots turn
// the 64 bytes into three 64-bit words in LDS -- the low and the high bit of the 2-bit base code
// (kmer.h:106-124; every byte that is not ACGTacgt codes as A) and the "is G or C" bit (kmer.h:91-104).
// Everything else works on those bit planes:
//   GC count         popcount of the GC plane;
//   hexamer at p     6 bits of each code plane starting at bit p -> a 12-bit index.  Entropy only
//                    needs the multiset of counts, so the index does not have to be the reference's
//                    packing: any one-to-one function of the hexamer does (the reference sorts the
//                    hexamers only to count equal ones, kmer.h:42-64);
//   counts           4096 u32 counters in LDS, ds_add_u32 per position;
//   windows          the 20 / 40 bits of the GC plane starting at bit p, popcount against the cutoff.
// The planes hold kChunk + 64 positions (the 64 extra ones are the look-ahead of the last hexamers and
// windows of a chunk); longer bins go chunk by chunk, the counters persist.
//
// Entropy = -sum over distinct hexamers of p log p, p = count / total (kmer.h:50-64).  A bin of one
// chunk (<= 4096 bases) sums it per POSITION -- every occurrence of a hexamer with count c contributes 1/c of its
// term: -(1/total) log(c / total) -- so the cost follows the bin's length, not the 4096 counters;
// a longer bin scans the counters.  Both sum positive terms only.  fp64 with the device's log():
// agrees with the reference to ~1e-15 relative, which the tests state as a tolerance (the reference
// prints six decimals).
//
// HighGCStrech compares (double)gc / w with 0.8 / 0.9 (kmer.h:72,85): for w = 20, 40 those are
// gc > 16, 18, 32, 36 (16/20 and 32/40 round to the double 0.8 itself, 18/20 and 36/40 to 0.9).
#pragma once

#include "device_common.h"

namespace sb {

constexpr int kBinSeqChunk = 4096;               // positions per chunk (a multiple of 64)
constexpr int kBinSeqWords = kBinSeqChunk / 64;  // plane words per chunk, + 1 look-ahead word
constexpr int kBinSeqErrRange = 1;               // a segment lies outside the genome window
constexpr int kBinSeqErrLong = 2;                // a bin longer than 2^31 - 1 bases

struct BinSeqArgs {
   const uint8_t *genome;  // genome[0] is base `genome_start` (1-based) of the chromosome
   int64_t genome_start, genome_len;
   int64_t n_bins;
   const int64_t *seg_off; // [n_bins + 1]
   const uint32_t *seg_left, *seg_right;
   double *gc, *entropy;
   uint8_t *flags;
   int32_t *error;         // or-ed kBinSeqErr*
};

__device__ __forceinline__ uint64_t plane_bits(const uint64_t *plane, int word, int lane)
{
   const uint64_t a = plane[word], b = plane[word + 1];
   return lane ? (a >> lane) | (b << (64 - lane)) : a;
}

__global__ __launch_bounds__(64) void binseq_kernel(BinSeqArgs a)
{
   __shared__ uint32_t hist[4096];
   __shared__ uint64_t plane_lo[kBinSeqWords + 1], plane_hi[kBinSeqWords + 1], plane_gc[kBinSeqWords + 1];
   const int lane = threadIdx.x;
   const int64_t bin = blockIdx.x;
   const int64_t s0 = a.seg_off[bin], s1 = a.seg_off[bin + 1];

   // ---- the bin's length, and its segments checked against the genome window
   int64_t len = 0;
   bool bad = false;
   for (int64_t s = s0 + lane; s < s1; s += 64) {
      const int64_t l = a.seg_left[s], r = a.seg_right[s];
      bad |= (r < l) | (l < a.genome_start) | (r - a.genome_start >= a.genome_len);
      len += r - l + 1;
   }
   for (int m = 1; m < 64; m <<= 1) len += __shfl_xor(len, m);
   if (__ballot(bad) != 0 || len > 0x7fffffff) {
      if (lane == 0) {
         atomicOr(a.error, __ballot(bad) != 0 ? kBinSeqErrRange : kBinSeqErrLong);
         a.gc[bin] = 0.0;
         a.entropy[bin] = 0.0;
         a.flags[bin] = 0;
      }
      return;
   }
   const int L = (int)len;
   const int total = L - 5; // hexamers (kmer.h:19-41); <= 0: none

   for (int i = lane; i < 4096 / 4; i += 64) ((uint4 *)hist)[i] = make_uint4(0, 0, 0, 0);

   // per-lane cursor into the segment list: the segment holding this lane's next position
   int64_t seg = s0;
   int seg_begin = 0, seg_end = 0; // concatenated positions [seg_begin, seg_end) are segment `seg`
   int64_t seg_base = 0;           // index into genome[] of the segment's first base
   if (s1 > s0) {
      seg_end = (int)(a.seg_right[s0] - a.seg_left[s0] + 1);
      seg_base = (int64_t)a.seg_left[s0] - a.genome_start;
   }

   int gc_count = 0;
   uint32_t flag_bits = 0;
   double acc = 0.0;
   const bool one_chunk = L <= kBinSeqChunk;

   for (int p0 = 0; p0 < L; p0 += kBinSeqChunk) {
      __syncthreads(); // the previous chunk's readers are done with the planes
      // ---- fetch: positions p0 .. p0 + kChunk + 63, one byte per lane, three ballots per 64.  The look-ahead
      // word of the previous chunk is this chunk's word 0: every base is fetched once and a lane's cursor only
      // ever moves forward.
      const int n_words = min(kBinSeqWords + 1, (L - p0 + 63) / 64); // words holding a position, look-ahead included
      if (p0 > 0 && lane == 0) {
         plane_lo[0] = plane_lo[kBinSeqWords];
         plane_hi[0] = plane_hi[kBinSeqWords];
         plane_gc[0] = plane_gc[kBinSeqWords];
      }
      for (int w = p0 > 0 ? 1 : 0; w < min(n_words + 1, kBinSeqWords + 1); ++w) {
         uint64_t lo = 0, hi = 0, g = 0;
         if (w < n_words) {
            const int p = p0 + 64 * w + lane;
            uint32_t c = 0;
            if (p < L) {
               while (p >= seg_end) {
                  ++seg;
                  seg_begin = seg_end;
                  seg_end += (int)(a.seg_right[seg] - a.seg_left[seg] + 1);
                  seg_base = (int64_t)a.seg_left[seg] - a.genome_start;
               }
               c = a.genome[seg_base + (p - seg_begin)];
            }
            const uint32_t u = c & 0xDFu; // upper case
            const bool isC = u == 'C', isG = u == 'G', isT = u == 'T';
            lo = __ballot((isC | isT) & (p < L));
            hi = __ballot((isG | isT) & (p < L));
            g = __ballot((isC | isG | (c == 1u) | (c == 2u)) & (p < L));
         }
         if (lane == 0) {
            plane_lo[w] = lo;
            plane_hi[w] = hi;
            plane_gc[w] = g;
         }
      }
      __syncthreads();
      // ---- count
      const int own_words = min(kBinSeqWords, n_words);
      for (int w = 0; w < own_words; ++w) {
         const int p = p0 + 64 * w + lane;
         if (lane == 0) gc_count += __popcll(plane_gc[w]);
         if (p < total) {
            const uint32_t idx = (uint32_t)(plane_bits(plane_lo, w, lane) & 63) | ((uint32_t)(plane_bits(plane_hi, w, lane) & 63) << 6);
            atomicAdd(&hist[idx], 1u);
         }
         const uint64_t g = plane_bits(plane_gc, w, lane);
         const int g20 = __popcll(g & 0xFFFFFull), g40 = __popcll(g & 0xFFFFFFFFFFull);
         const bool w20 = p + 20 <= L, w40 = p + 40 <= L;
         flag_bits |= (uint32_t)(w20 & (g20 > 16)) | ((uint32_t)(w20 & (g20 > 18)) << 1) | ((uint32_t)(w40 & (g40 > 32)) << 2) |
                      ((uint32_t)(w40 & (g40 > 36)) << 3);
      }
      if (one_chunk) {
         // ---- entropy per position while the planes are still here (one chunk: every count is final)
         __syncthreads();
         for (int w = 0; w < own_words; ++w) {
            const int p = 64 * w + lane;
            if (p < total) {
               const uint32_t idx = (uint32_t)(plane_bits(plane_lo, w, lane) & 63) | ((uint32_t)(plane_bits(plane_hi, w, lane) & 63) << 6);
               acc -= log((double)hist[idx] / (double)total);
            }
         }
      }
   }
   if (!one_chunk) {
      __syncthreads();
      for (int i = lane; i < 4096; i += 64) {
         const uint32_t c = hist[i];
         if (c) {
            const double p = (double)c / (double)total;
            acc -= p * log(p);
         }
      }
   }
   for (int m = 1; m < 64; m <<= 1) {
      acc += __shfl_xor(acc, m);
      flag_bits |= (uint32_t)__shfl_xor((int)flag_bits, m);
   }
   if (lane == 0) {
      a.gc[bin] = (double)gc_count / (double)L; // 0/0 = NaN for an empty bin (the reference asserts there)
      a.entropy[bin] = total > 0 ? (one_chunk ? acc / (double)total : acc) : 0.0;
      a.flags[bin] = (uint8_t)flag_bits;
   }
}

} // namespace sb
