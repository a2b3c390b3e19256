// strawberry_amd/csrc/matepair_device.h -- what the mate-pairing kernels share (HitCluster::addOpenHit / addHit on the GPU,
// /root/reference/src/alignments.cpp:423-655; the rules are restated at sbgpu_pair_mates_host in
// include/sbgpu.h): the records' arrays, a mate's features (readhit_2_genomicFeats), and the kernels of the cluster streaming
// (sbgpu_assign_reads_device).  The pairing itself: matepair_flat.h (all clusters of a call at once; round 3's
// one-workgroup-per-cluster kernels, which nothing had reached since round 4, are gone).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_common.h"

namespace sb {

constexpr int kMaxFragSpanDev = 1000000; // src/common.cpp:17

struct MateArgs {
   int64_t n_loci;
   const int64_t *locus_read_off; // [n_loci + 1]
   const uint64_t *read_id;
   const int64_t *block_off;
   const uint32_t *block_left, *block_right;
   const uint32_t *partner_pos;
   const uint8_t *flags;
   const int32_t *nh;
   // the pairs (written by the fill)
   double *pair_mass;
   int64_t *left_off, *right_off; // [n_pairs + 1] (the last entry is written by the host)
   uint8_t *left_code, *right_code;
   uint32_t *left_left, *left_right, *right_left, *right_right;
};

// The features a record's aligned blocks make (readhit_2_genomicFeats, src/contig.cpp:12-53): a MATCH per block, an INTRON
// between two blocks -- none where they touch: an insertion in the read (M I M) leaves two MATCH features side by side
__device__ __forceinline__ int mate_feature_count(const MateArgs &a, int64_t r)
{
   const int64_t b0 = a.block_off[r], b1 = a.block_off[r + 1];
   if (b1 <= b0) return 0;
   int n = 1;
   for (int64_t b = b0 + 1; b < b1; ++b) n += (a.block_left[b] == a.block_right[b - 1] + 1u) ? 1 : 2;
   return n;
}

__device__ __forceinline__ void write_mate(const MateArgs &a, int64_t r, uint8_t *code, uint32_t *left, uint32_t *right, int64_t at)
{
   const int64_t b0 = a.block_off[r], b1 = a.block_off[r + 1];
   for (int64_t b = b0; b < b1; ++b) {
      if (b > b0 && a.block_left[b] != a.block_right[b - 1] + 1u) { // (touching blocks: an insertion, no intron)
         code[at] = 1;
         left[at] = a.block_right[b - 1] + 1;
         right[at] = a.block_left[b] - 1;
         ++at;
      }
      code[at] = 0;
      left[at] = a.block_left[b];
      right[at] = a.block_right[b];
      ++at;
   }
}

struct AssignArgs {
   int64_t n_clusters, n_reads;
   const int32_t *c_ref;
   const uint32_t *c_left, *c_right;
   const uint8_t *c_strand;
   const int32_t *r_ref;
   const uint32_t *r_left, *r_right;
   uint8_t *r_flags; // may be null
   int64_t *ub;            // [n_clusters] first record behind the cluster's end
   const int64_t *pos;     // [n_clusters + 1] where every cluster's pass begins (prefix maximum of ub)
   int32_t *read_cluster;
};
__device__ __forceinline__ unsigned long long ref_pos_key(int32_t ref, uint32_t pos) { return ((unsigned long long)(uint32_t)ref << 32) | pos; }

__global__ __launch_bounds__(256) void cluster_bounds_kernel(AssignArgs a)
{
   const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
   if (k >= a.n_clusters) return;
   const unsigned long long end = ref_pos_key(a.c_ref[k], a.c_right[k]);
   int64_t lo = 0, hi = a.n_reads; // first record with (ref, left) > (cluster ref, cluster right): hit_gt_cluster
   while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if (ref_pos_key(a.r_ref[mid], a.r_left[mid]) <= end) lo = mid + 1;
      else hi = mid;
   }
   a.ub[k] = lo;
}

__global__ __launch_bounds__(256) void assign_reads_kernel(AssignArgs a)
{
   const int64_t n_in = min(a.n_reads, a.pos[a.n_clusters]); // the records inside some cluster's pass
   const int lane = (int)(threadIdx.x & 63u);
   // A wave takes a CONTIGUOUS stretch of records, 64 per trip, and carries the cluster of the last record it placed with that
   // cluster's range of records: the next trip's 64 lie inside it as a rule (a cluster of the chain sample holds 6 500
   // records) and are placed without a load; only a trip that leaves the range searches (wave_range_of: three dependent
   // round trips -- with a search per trip, as before, they were most of the kernel: 4.3 ms for 6.6 GB).
   const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6, wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
   const int64_t per = ((a.n_reads + n_waves * 64 - 1) / (n_waves * 64)) * 64; // records per wave, whole trips
   const int64_t begin = wave * per, end = min(begin + per, a.n_reads);
   int64_t k_c = -1, lo_c = 0, hi_c = 0; // the carried cluster and its records [lo_c, hi_c)
   for (int64_t i0 = begin; i0 < end; i0 += 64) { // (uniform)
      const int64_t i = i0 + lane;
      // the last cluster whose pass begins at or before record i and is not empty there: pos[k] <= i < pos[k + 1]
      int64_t kw = 0;
      if (i0 < n_in) {
         const int64_t last = min(i0 + 63, n_in - 1);
         if (k_c >= 0 && i0 >= lo_c && last < hi_c) {
            kw = k_c;
         } else {
            kw = wave_range_of(a.pos, a.n_clusters, i0, n_in);
            k_c = __shfl(kw, 63); // (the lane of the trip's last record, or of the last record inside a pass: the search clamps)
            lo_c = a.pos[k_c], hi_c = a.pos[k_c + 1];
         }
      }
      if (i >= a.n_reads) continue;
      int32_t c = -1;
      if (i < n_in) {
         const int64_t k = kw;
         const bool lt = a.r_ref[i] < a.c_ref[k] || (a.r_ref[i] == a.c_ref[k] && a.r_right[i] < a.c_left[k]); // hit_lt_cluster
         const int xs = a.r_flags ? (a.r_flags[i] >> 2) & 3 : 0;
         const bool strand_off = xs != 0 && xs != (int)a.c_strand[k]; // alignments.cpp:1168
         if (!lt && !strand_off) c = (int32_t)k;
      }
      a.read_cluster[i] = c;
      if (a.r_flags && c < 0) a.r_flags[i] |= 16u;
   }
}

} // namespace sb
