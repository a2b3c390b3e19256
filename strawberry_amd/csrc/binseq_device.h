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
// TABLE: the bin's segment list lives in LDS (tbl_end / tbl_base, <= 64 segments) and a lane's cursor walks it
// there; otherwise the cursor walks the list in global memory (any number of segments).
constexpr int kBinSeqBatch = 16;

struct SegCursor {
   int64_t seg;   // index of the segment (into the bin's list when TABLE, global otherwise)
   int begin, end; // concatenated positions [begin, end) are that segment
   int64_t base;  // index into genome[] of the segment's first base
};

template <bool TABLE>
__device__ __forceinline__ void fetch_batch(const BinSeqArgs &a, SegCursor &cur, const int *tbl_end, const int64_t *tbl_base,
                                            int p0, int w0, int w1, int L, int lane, uint64_t *plane_lo, uint64_t *plane_hi,
                                            uint64_t *plane_gc)
{
   uint32_t c[kBinSeqBatch];
#pragma unroll
   for (int i = 0; i < kBinSeqBatch; ++i) {
      const int p = p0 + 64 * (w0 + i) + lane;
      c[i] = 0;
      if (w0 + i < w1 && p < L) {
         while (p >= cur.end) {
            ++cur.seg;
            cur.begin = cur.end;
            if (TABLE) {
               cur.end = tbl_end[cur.seg];
               cur.base = tbl_base[cur.seg];
            } else {
               cur.end += (int)(a.seg_right[cur.seg] - a.seg_left[cur.seg] + 1);
               cur.base = (int64_t)a.seg_left[cur.seg] - a.genome_start;
            }
         }
         c[i] = a.genome[cur.base + (p - cur.begin)];
      }
   }
#pragma unroll
   for (int i = 0; i < kBinSeqBatch; ++i) {
      if (w0 + i < w1) { // wave-uniform
         // a position past the bin's end holds c = 0: no bit in any plane
         const uint32_t u = c[i] & 0xDFu; // upper case
         const bool isC = u == 'C', isG = u == 'G', isT = u == 'T';
         const uint64_t lo = __ballot(isC | isT), hi = __ballot(isG | isT), g = __ballot(isC | isG | (c[i] - 1u < 2u));
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

__global__ __launch_bounds__(64) void binseq_kernel(BinSeqArgs a)
{
   __shared__ uint32_t hist[2048];
   // one spare 32-bit half after the look-ahead word: plane_bits32(.., more = 1) reads it (and masks it away)
   __shared__ uint64_t plane_lo[kBinSeqWords + 2], plane_hi[kBinSeqWords + 2], plane_gc[kBinSeqWords + 2];
   __shared__ int tbl_end[64];
   __shared__ int64_t tbl_base[64];
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
   int64_t my_base = 0; // lane k: segment k (table form)
   for (int64_t s = s0 + lane; s < s1; s += 64) {
      const int64_t l = a.seg_left[s], r = a.seg_right[s];
      bad |= (r < l) | (l < a.genome_start) | (r - a.genome_start >= a.genome_len);
      len += r - l + 1;
      my_len = (int)(r - l + 1);
      my_base = l - a.genome_start;
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
   if (table) {
      int scan = my_len; // inclusive prefix sum over lanes: the end position of lane k's segment
      for (int m = 1; m < 64; m <<= 1) {
         const int up = __shfl_up(scan, m);
         if (lane >= m) scan += up;
      }
      tbl_end[lane] = scan;
      tbl_base[lane] = my_base;
   }
   log_tbl[lane] = log((double)(lane + 1));
   for (int i = lane; i < 2048 / 4; i += 64) ((uint4 *)hist)[i] = make_uint4(0, 0, 0, 0);
   if (lane == 0) plane_lo[kBinSeqWords + 1] = plane_hi[kBinSeqWords + 1] = plane_gc[kBinSeqWords + 1] = 0;

   int gc_count = 0;
   uint32_t flag_bits = 0;
   double acc = 0.0;
   const bool one_chunk = L <= kBinSeqChunk;
   const bool packed = total <= kBinSeqPackedMax;
   const double log_total = total > 0 ? log((double)total) : 0.0;

   for (int pass = 0; pass < (packed ? 1 : 2); ++pass) {
      // per-lane cursor into the segment list: the segment holding this lane's next position
      SegCursor cur;
      cur.seg = table ? 0 : s0;
      cur.begin = 0;
      cur.end = 0;
      cur.base = 0;
      if (s1 > s0) {
         cur.end = (int)(a.seg_right[s0] - a.seg_left[s0] + 1);
         cur.base = (int64_t)a.seg_left[s0] - a.genome_start;
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
               fetch_batch<true>(a, cur, tbl_end, tbl_base, p0, w0, n_words, L, lane, plane_lo, plane_hi, plane_gc);
            else
               fetch_batch<false>(a, cur, tbl_end, tbl_base, p0, w0, n_words, L, lane, plane_lo, plane_hi, plane_gc);
         }
         if (n_words <= kBinSeqWords && lane == 0) { // the word after the last one is read as look-ahead: zero
            plane_lo[n_words] = 0;
            plane_hi[n_words] = 0;
            plane_gc[n_words] = 0;
         }
         __syncthreads();
         // ---- count
         const int own_words = min(kBinSeqWords, n_words);
         for (int w = 0; w < own_words; ++w) {
            const int p = p0 + 64 * w + lane;
            if (p < total) {
               const uint32_t idx = (plane_bits32(lo32, w, lane) & 63u) | ((plane_bits32(hi32, w, lane) & 63u) << 6);
               if (packed)
                  atomicAdd(&hist[idx >> 1], 1u << ((idx & 1u) * 16u));
               else if ((int)(idx >> 11) == pass)
                  atomicAdd(&hist[idx & 2047u], 1u);
            }
            if (pass == 0) {
               if (lane == 0) gc_count += __popcll(plane_gc[w]);
               const uint32_t g0 = plane_bits32(gc32, w, lane), g1 = plane_bits32(gc32, w, lane, 1);
               const int g20 = __popc(g0 & 0xFFFFFu), g40 = __popc(g0) + __popc(g1 & 0xFFu);
               const bool w20 = p + 20 <= L, w40 = p + 40 <= L;
               flag_bits |= (uint32_t)(w20 & (g20 > 16)) | ((uint32_t)(w20 & (g20 > 18)) << 1) | ((uint32_t)(w40 & (g40 > 32)) << 2) |
                            ((uint32_t)(w40 & (g40 > 36)) << 3);
            }
         }
         if (one_chunk) {
            // ---- entropy per position while the planes are still here (one chunk: every count is final).  A position
            // whose hexamer occurs c times adds (log total - log c) / total; log c from the table for c <= 64.
            __syncthreads();
            for (int w = 0; w < own_words; ++w) {
               const int p = 64 * w + lane;
               if (p < total) {
                  const uint32_t idx = (plane_bits32(lo32, w, lane) & 63u) | ((plane_bits32(hi32, w, lane) & 63u) << 6);
                  const uint32_t c = (hist[idx >> 1] >> ((idx & 1u) * 16u)) & 0xFFFFu;
                  acc += log_total - (c <= 64 ? log_tbl[c - 1] : log((double)c));
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
      flag_bits |= (uint32_t)__shfl_xor((int)flag_bits, m);
   }
   if (lane == 0) {
      a.gc[bin] = (double)gc_count / (double)L; // 0/0 = NaN for an empty bin (the reference asserts there)
      a.entropy[bin] = total > 0 ? (one_chunk ? acc / (double)total : acc) : 0.0;
      a.flags[bin] = (uint8_t)flag_bits;
   }
}

} // namespace sb
