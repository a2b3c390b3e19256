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
//   counts           4096 counters in 8 KB of LDS (16-bit pairs; see kBinSeqPackedMax), ds_add_u32 per position;
//   windows          the 20 / 40 bits of the GC plane starting at bit p, popcount against the cutoff.
// The planes hold kChunk + 64 positions (the 64 extra ones are the look-ahead of the last hexamers and
// windows of a chunk); longer bins go chunk by chunk, the counters persist.
//
// Entropy = -sum over distinct hexamers of p log p, p = count / total (kmer.h:50-64).  A bin of one
// chunk (<= 4096 bases) sums it per POSITION -- every occurrence of a hexamer with count c contributes 1/c of its
// term: (log total - log c) / total, log c from a 64-entry table -- so the cost follows the bin's length, not
// the 4096 counters;
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

// 32 bits of a plane starting at bit `lane` of 64-position word `word` (+ 32 * `more`): one v_alignbit_b32 over two
// neighbouring 32-bit halves.  The planes are read as uint32_t; a word's low half comes first (little endian).
__device__ __forceinline__ uint32_t plane_bits32(const uint32_t *plane, int word, int lane, int more = 0)
{
   const int i = 2 * word + (lane >> 5) + more;
   return __builtin_amdgcn_alignbit(plane[i + 1], plane[i], (uint32_t)lane & 31u);
}

// One batch of the fetch: kBinSeqBatch words (64 positions each) starting at word w0 of the chunk at p0.  All the
// byte loads of a batch are issued before the first one is used (the ballots come in a second loop), so a lane has
// up to kBinSeqBatch loads in flight -- the kernel is otherwise bound by the latency of one dependent load per word.
// TABLE: the bin's segment list lives in LDS (tbl_end / tbl_shift, <= 64 segments) and a lane's cursor walks it
// there; otherwise the cursor walks the list in global memory (any number of segments).
constexpr int kBinSeqBatch = 8;

struct SegCursor {
   int64_t seg;    // index of the segment (into the bin's list when TABLE, global otherwise)
   int end;        // the segment ends before concatenated position `end`
   uint32_t shift; // index into genome[] of concatenated position p while in this segment: p + shift (mod 2^32)
};

template <bool TABLE>
__device__ __forceinline__ void fetch_batch(const BinSeqArgs &a, SegCursor &cur, const int *tbl_end, const uint32_t *tbl_shift,
                                            const uint8_t *lut, int p0, int w0, int w1, int L, int lane, uint64_t *plane_lo,
                                            uint64_t *plane_hi, uint64_t *plane_gc)
{
   uint32_t c[kBinSeqBatch];
#pragma unroll
   for (int i = 0; i < kBinSeqBatch; ++i) {
      const int p = p0 + 64 * (w0 + i) + lane;
      c[i] = 0;
      if (w0 + i < w1 && p < L) {
         while (p >= cur.end) {
            ++cur.seg;
            const int begin = cur.end;
            if (TABLE) {
               cur.end = tbl_end[cur.seg];
               cur.shift = tbl_shift[cur.seg];
            } else {
               cur.end += (int)(a.seg_right[cur.seg] - a.seg_left[cur.seg] + 1);
               cur.shift = (uint32_t)(a.seg_left[cur.seg] - (uint32_t)a.genome_start) - (uint32_t)begin;
            }
         }
         c[i] = a.genome[(uint32_t)p + cur.shift]; // genome_len < 2^32 (checked by the launcher): a 32-bit offset
      }
   }
#pragma unroll
   for (int i = 0; i < kBinSeqBatch; ++i) {
      if (w0 + i < w1) { // wave-uniform
         // lut[byte]: bit 0 / 1 = the base code's low / high bit, bit 2 = counts as GC (kmer.h:91-124).  A position
         // past the bin's end holds byte 0: no bit in any plane.
         const uint32_t t = lut[c[i]];
         const uint64_t lo = __ballot(t & 1u), hi = __ballot(t & 2u), g = __ballot(t & 4u);
         if (lane == 0) {
            plane_lo[w0 + i] = lo;
            plane_hi[w0 + i] = hi;
            plane_gc[w0 + i] = g;
         }
      }
   }
}

// The 4096 hexamer counters share 8 KB of LDS (twice as many waves per CU as with 16 KB):
//   total <= 65535   packed: two 16-bit counters per word, counter idx in half (idx & 1) of word idx >> 1; a plain
//                    ds_add_u32 of 1 << 16 * (idx & 1) cannot carry from one half into the other;
//   larger bins      two passes over the bin, pass q counting the hexamers with idx >> 11 == q in 2048 32-bit counters.
constexpr int kBinSeqPackedMax = 65535;

// log(n) for n = 0 .. 4096 (entry 0 unused), filled once per device by the host with its own libm -- the one the
// reference calls (binseq_api.hip).  A one-chunk bin takes log(total) and the log of every count from here.
__device__ double g_binseq_log_n[kBinSeqChunk + 1];
// byte -> bit 0 / 1: low / high bit of the base code (C = 1, G = 2, T = 3, anything else 0; either case),
// bit 2: counts as GC (C c G g and the bytes 1, 2).  Filled with the log table.
__device__ uint8_t g_binseq_lut[256];

__global__ __launch_bounds__(64) void binseq_kernel(BinSeqArgs a)
{
   __shared__ uint32_t hist[2048];
   // one spare 32-bit half after the look-ahead word: plane_bits32(.., more = 1) reads it (and masks it away)
   __shared__ uint64_t plane_lo[kBinSeqWords + 2], plane_hi[kBinSeqWords + 2], plane_gc[kBinSeqWords + 2];
   __shared__ int tbl_end[64];
   __shared__ uint32_t tbl_shift[64];
   __shared__ uint8_t lut[256];
   __shared__ double log_tbl[64]; // log(1 .. 64)
   const uint32_t *lo32 = (const uint32_t *)plane_lo, *hi32 = (const uint32_t *)plane_hi, *gc32 = (const uint32_t *)plane_gc;
   const int lane = threadIdx.x;
   const int64_t bin = blockIdx.x;
   const int64_t s0 = a.seg_off[bin], s1 = a.seg_off[bin + 1];
   const bool table = s1 - s0 <= 64;

   // ---- the bin's length, and its segments checked against the genome window
   int64_t len = 0;
   bool bad = false;
   int my_len = 0;
   uint32_t my_base = 0; // lane k: segment k (table form)
   for (int64_t s = s0 + lane; s < s1; s += 64) {
      const int64_t l = a.seg_left[s], r = a.seg_right[s];
      bad |= (r < l) | (l < a.genome_start) | (r - a.genome_start >= a.genome_len);
      len += r - l + 1;
      my_len = (int)(r - l + 1);
      my_base = (uint32_t)(l - a.genome_start);
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
   const int L = __builtin_amdgcn_readfirstlane((int)len); // wave-uniform: loop counts stay in SGPRs
   const int total = L - 5; // hexamers (kmer.h:19-41); <= 0: none
   if (table) {
      int scan = my_len; // inclusive prefix sum over lanes: the end position of lane k's segment
      for (int m = 1; m < 64; m <<= 1) {
         const int up = __shfl_up(scan, m);
         if (lane >= m) scan += up;
      }
      tbl_end[lane] = scan;
      tbl_shift[lane] = my_base - (uint32_t)(scan - my_len);
   }
   log_tbl[lane] = g_binseq_log_n[lane + 1];
   ((uint32_t *)lut)[lane] = ((const uint32_t *)g_binseq_lut)[lane];
   for (int i = lane; i < 2048 / 4; i += 64) ((uint4 *)hist)[i] = make_uint4(0, 0, 0, 0);
   if (lane == 0) plane_lo[kBinSeqWords + 1] = plane_hi[kBinSeqWords + 1] = plane_gc[kBinSeqWords + 1] = 0;

   int gc_count = 0; // per lane, summed at the end
   int max20 = 0, max40 = 0;
   double acc = 0.0;
   const bool one_chunk = L <= kBinSeqChunk;
   const bool packed = total <= kBinSeqPackedMax;
   const double log_total = (one_chunk && total > 0) ? g_binseq_log_n[total] : 0.0; // only the one-chunk form uses it

   for (int pass = 0; pass < (packed ? 1 : 2); ++pass) {
      // per-lane cursor into the segment list: the segment holding this lane's next position
      SegCursor cur;
      cur.seg = table ? 0 : s0;
      cur.end = 0;
      cur.shift = 0;
      if (s1 > s0) {
         cur.end = (int)(a.seg_right[s0] - a.seg_left[s0] + 1);
         cur.shift = (uint32_t)(a.seg_left[s0] - (uint32_t)a.genome_start);
      }
      for (int p0 = 0; p0 < L; p0 += kBinSeqChunk) {
         __syncthreads(); // the previous chunk's readers are done with the planes (first chunk: the tables are written)
         // ---- fetch: positions p0 .. p0 + kChunk + 63, one byte per lane, three ballots per 64.  The look-ahead
         // word of the previous chunk is this chunk's word 0: every base is fetched once per pass and a lane's
         // cursor only ever moves forward.
         const int n_words = min(kBinSeqWords + 1, (L - p0 + 63) / 64); // words holding a position, look-ahead included
         if (p0 > 0 && lane == 0) {
            plane_lo[0] = plane_lo[kBinSeqWords];
            plane_hi[0] = plane_hi[kBinSeqWords];
            plane_gc[0] = plane_gc[kBinSeqWords];
         }
         for (int w0 = p0 > 0 ? 1 : 0; w0 < n_words; w0 += kBinSeqBatch) {
            if (table)
               fetch_batch<true>(a, cur, tbl_end, tbl_shift, lut, p0, w0, n_words, L, lane, plane_lo, plane_hi, plane_gc);
            else
               fetch_batch<false>(a, cur, tbl_end, tbl_shift, lut, p0, w0, n_words, L, lane, plane_lo, plane_hi, plane_gc);
         }
         if (n_words <= kBinSeqWords && lane == 0) { // the word after the last one is read as look-ahead: zero
            plane_lo[n_words] = 0;
            plane_hi[n_words] = 0;
            plane_gc[n_words] = 0;
         }
         __syncthreads();
         // ---- count.  Window flags: a lane keeps the largest GC count of the 20- and 40-base windows starting at its
         // positions.  A window running past the bin's end only sees zero bits there, and the last full window
         // contains whatever it does see -- so it cannot raise the maximum above a full window's unless the bin
         // is shorter than the window, which the final test excludes.
         const int own_words = min(kBinSeqWords, n_words);
         if (packed) {
            for (int w = 0; w < own_words; ++w) {
               const uint32_t lo = plane_bits32(lo32, w, lane), hi = plane_bits32(hi32, w, lane);
               // counter (lo6 | hi6 << 6): 16-bit half (lo & 1) of the 32-bit word (lo6 >> 1) | hi6 << 5
               if (p0 + 64 * w + lane < total) atomicAdd(&hist[((lo & 62u) >> 1) | ((hi & 63u) << 5)], (lo & 1u) ? 0x10000u : 1u);
            }
         } else {
            for (int w = 0; w < own_words; ++w) {
               const uint32_t idx = (plane_bits32(lo32, w, lane) & 63u) | ((plane_bits32(hi32, w, lane) & 63u) << 6);
               if (p0 + 64 * w + lane < total && (int)(idx >> 11) == pass) atomicAdd(&hist[idx & 2047u], 1u);
            }
         }
         if (pass == 0) {
            for (int w = 0; w < own_words; ++w) {
               const uint32_t g0 = plane_bits32(gc32, w, lane), g1 = plane_bits32(gc32, w, lane, 1);
               max20 = max(max20, (int)__popc(g0 & 0xFFFFFu));
               max40 = max(max40, (int)(__popc(g0) + __popc(g1 & 0xFFu)));
            }
            if (lane < own_words) gc_count += __popcll(plane_gc[lane]); // lane w: word w's GC bits
         }
         if (one_chunk) {
            // ---- entropy per position while the planes are still here (one chunk: every count is final).  A position
            // whose hexamer occurs c times adds (log total - log c) / total; log c from the table for c <= 64.
            __syncthreads();
            for (int w = 0; w < own_words; ++w) {
               const int p = 64 * w + lane;
               if (p < total) {
                  const uint32_t lo = plane_bits32(lo32, w, lane), hi = plane_bits32(hi32, w, lane);
                  const uint32_t v = hist[((lo & 62u) >> 1) | ((hi & 63u) << 5)];
                  const uint32_t c = (lo & 1u) ? v >> 16 : v & 0xFFFFu;
                  acc += log_total - (c <= 64 ? log_tbl[c - 1] : g_binseq_log_n[c]);
               }
            }
         }
      }
      if (!one_chunk) { // scan the counters; clear them for the second pass
         __syncthreads();
         for (int i = lane; i < 2048; i += 64) {
            const uint32_t v = hist[i];
            hist[i] = 0;
            const uint32_t c0 = packed ? (v & 0xFFFFu) : v, c1 = packed ? (v >> 16) : 0u;
            if (c0) {
               const double q = (double)c0 / (double)total;
               acc -= q * log(q);
            }
            if (c1) {
               const double q = (double)c1 / (double)total;
               acc -= q * log(q);
            }
         }
      }
   }
   for (int m = 1; m < 64; m <<= 1) {
      acc += __shfl_xor(acc, m);
      gc_count += __shfl_xor(gc_count, m);
   }
   const uint32_t m20 = wave_max_u32((uint32_t)max20), m40 = wave_max_u32((uint32_t)max40);
   if (lane == 0) {
      a.gc[bin] = (double)gc_count / (double)L; // 0/0 = NaN for an empty bin (the reference asserts there)
      a.entropy[bin] = total > 0 ? (one_chunk ? acc / (double)total : acc) : 0.0;
      a.flags[bin] = (uint8_t)((L >= 20 ? (uint32_t)(m20 > 16) | ((uint32_t)(m20 > 18) << 1) : 0u) |
                               (L >= 40 ? ((uint32_t)(m40 > 32) << 2) | ((uint32_t)(m40 > 36) << 3) : 0u));
   }
}

} // namespace sb
