// strawberry_amd/csrc/matepair_device.h -- HitCluster::addOpenHit / addHit on the GPU: the
// alignment records of a cluster -> its read pairs (/root/reference/src/alignments.cpp:423-655; the rules are
// restated at sbgpu_pair_mates_host in include/sbgpu.h).
//
// The reference walks the records in arrival order and keeps, per read id, a list of mates that wait for their
// partner.  Only records of ONE read id ever interact, so: one workgroup per cluster,
//   1. (read id, arrival index) of every record go to LDS and are sorted (bitonic): a read id's records become
//      neighbours, in arrival order;
//   2. the first record of every read id walks its group with the reference's list logic (groups are two records
//      long, a handful for multi-mapped reads) and marks, per record, what became of it: the record that COMPLETES a
//      pair (or is a single read) knows its partner and which mate it is;
//   3. the cluster's hits are ordered by completion (addHit is called when the second mate arrives): a pair's rank is
//      the number of completing records before its own -- one prefix count over the arrival order;
//   4. after the host has turned the per-cluster counts into offsets, a second kernel writes the pairs as
//      sbgpu_pairs_t arrays (mates as MATCH / INTRON feature lists), which sbgpu_collapse_pairs_device reads.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "bitonic_big.h"
#include "device_common.h"

namespace sb {

constexpr int kMateMaxReads = 8192;  // records of one cluster (LDS sort)
constexpr int kMateThreads = 256;
constexpr int kMateOpenMax = 8;      // mates of one read id waiting at a time
constexpr int kMaxFragSpanDev = 1000000; // src/common.cpp:17
enum : int32_t { kMateTooMany = 1, kMateOpenOverflow = 2 };
// what became of a record (per record, arrival order)
enum : int8_t { kRecRefused = 0, kRecOrphan = 1, kRecFirst = 2, kRecCompletesAsRight = 3, kRecCompletesAsLeft = 4, kRecSingleLeft = 5, kRecSingleRight = 6 };

struct MateArgs {
   int64_t n_loci;
   const int64_t *locus_read_off; // [n_loci + 1]
   const uint64_t *read_id;
   const int64_t *block_off;
   const uint32_t *block_left, *block_right;
   const uint32_t *partner_pos;
   const uint8_t *flags;
   const int32_t *nh;
   // per record scratch
   int8_t *fate;      // kRec*
   int32_t *partner;  // for a completing record: arrival index (inside the cluster) of the mate that waited
   int32_t *rank;     // for a completing record or a single read: its pair's rank inside the cluster
   // per cluster
   int32_t *n_pairs, *n_complete, *n_single, *n_refused, *n_orphan, *n_lfeat, *n_rfeat;
   int32_t *flags_out;
   // pass 2
   const int64_t *pair_off, *lfeat_base, *rfeat_base; // [n_loci + 1]
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

__device__ __forceinline__ bool mate_key_less(unsigned long long ka, int ia, unsigned long long kb, int ib)
{
   return ka != kb ? ka < kb : ia < ib;
}

// exclusive prefix sums of v[0 .. n), in place, by the whole workgroup of THREADS threads
template <int THREADS>
__device__ inline void block_exclusive_scan(int *v, int n, int *partial /*[THREADS]*/)
{
   const int tid = threadIdx.x;
   const int chunk = (n + THREADS - 1) / THREADS;
   const int lo = min(n, tid * chunk), hi = min(n, lo + chunk);
   int s = 0;
   for (int i = lo; i < hi; ++i) s += v[i];
   partial[tid] = s;
   __syncthreads();
   if (tid == 0) {
      int run = 0;
      for (int t = 0; t < THREADS; ++t) {
         const int x = partial[t];
         partial[t] = run;
         run += x;
      }
   }
   __syncthreads();
   int run = partial[tid];
   for (int i = lo; i < hi; ++i) {
      const int x = v[i];
      v[i] = run;
      run += x;
   }
   __syncthreads();
}

// One cluster by one workgroup of THREADS threads.  key / idx [pow2ceil(n)]: LDS for clusters of up to kMateMaxReads
// records, global scratch for bigger ones (matepair_big_kernel) -- the same steps either way.
template <int THREADS>
__device__ __forceinline__ void matepair_one_locus(const MateArgs &a, int64_t l, int n, unsigned long long *key, int *idx, int *partial, int *counts,
                                                   unsigned long long *stage_k = nullptr, int *stage_i = nullptr) // (LDS chunk buffers: the global-memory form's sort)
{
   const int tid = threadIdx.x;
   const int64_t q0 = a.locus_read_off[l];
   int npad = 1;
   while (npad < n) npad <<= 1;
   for (int i = tid; i < npad; i += THREADS) {
      key[i] = i < n ? a.read_id[q0 + i] : ~0ull;
      idx[i] = i < n ? i : 0x7fffffff;
   }
   __syncthreads();
   // ---- bitonic sort on (read id, arrival index)
   if (stage_k) bitonic_sort_global<THREADS, 4096>(key, idx, npad, stage_k, stage_i);
   else
   for (int k = 2; k <= npad; k <<= 1)
      for (int j = k >> 1; j > 0; j >>= 1) {
         for (int i = tid; i < npad; i += THREADS) {
            const int p = i ^ j;
            if (p > i) {
               const bool up = (i & k) == 0;
               const unsigned long long ki = key[i], kp = key[p];
               const int ii = idx[i], ip = idx[p];
               const bool swap = up ? mate_key_less(kp, ip, ki, ii) : mate_key_less(ki, ii, kp, ip);
               if (swap) {
                  key[i] = kp, key[p] = ki;
                  idx[i] = ip, idx[p] = ii;
               }
            }
         }
         __syncthreads();
      }
   // ---- every read id's records, in arrival order, with the reference's open-mate list (alignments.cpp:535-641)
   int my_refused = 0, my_orphan = 0, my_single = 0, my_complete = 0, my_bad = 0;
   for (int s = tid; s < n; s += THREADS) {
      if (s > 0 && key[s - 1] == key[s]) continue; // not the first of its read id
      int open[kMateOpenMax];
      int n_open = 0;
      for (int t = s; t < n && key[t] == key[s]; ++t) {
         const int i = idx[t];
         const int64_t r = q0 + i;
         const int64_t b0 = a.block_off[r], b1 = a.block_off[r + 1];
         const uint32_t left = b1 > b0 ? a.block_left[b0] : 0u, right = b1 > b0 ? a.block_right[b1 - 1] : 0u;
         const uint32_t ppos = a.partner_pos[r];
         const uint8_t fl = a.flags[r];
         if (fl & 16u) { // not this cluster's record (sbgpu_assign_reads_*): never offered to addOpenHit
            a.fate[r] = kRecRefused;
            continue;
         }
         if (b1 <= b0 || (int64_t)right - (int64_t)left > kMaxFragSpanDev) { // :512-518
            a.fate[r] = kRecRefused;
            ++my_refused;
            continue;
         }
         if (ppos == 0 || (fl & 2u)) { // a single read (:535-545)
            a.fate[r] = (fl & 1u) ? kRecSingleRight : kRecSingleLeft;
            a.partner[r] = -1;
            ++my_single;
            continue;
         }
         const int strand = (fl >> 2) & 3;
         int hit = -1;
         for (int o = 0; o < n_open && hit < 0; ++o) { // :590-623, oldest first
            const int64_t w = q0 + open[o];
            const int wstrand = (a.flags[w] >> 2) & 3;
            const bool strand_agree = wstrand == strand || strand == 0 || wstrand == 0;
            if (a.block_left[a.block_off[w]] == ppos && strand_agree && a.partner_pos[w] == left) hit = o;
         }
         if (hit >= 0) {
            const int64_t w = q0 + open[hit];
            // the waiting mate is the left one when its partner lies behind it (:559-585)
            const bool waiting_is_left = a.partner_pos[w] > a.block_left[a.block_off[w]];
            a.fate[r] = waiting_is_left ? kRecCompletesAsRight : kRecCompletesAsLeft;
            a.partner[r] = open[hit];
            a.fate[w] = kRecFirst;
            for (int o = hit; o + 1 < n_open; ++o) open[o] = open[o + 1];
            --n_open;
            ++my_complete;
         } else if (ppos == left) { // :585, :640: partner and read start at the same position
            a.fate[r] = kRecRefused;
            ++my_refused;
         } else if (n_open < kMateOpenMax) {
            a.fate[r] = kRecOrphan; // until its partner comes
            open[n_open++] = i;
         } else {
            a.fate[r] = kRecOrphan;
            my_bad |= kMateOpenOverflow;
         }
      }
      my_orphan += n_open;
   }
   if (my_refused) atomicAdd(&counts[0], my_refused);
   if (my_orphan) atomicAdd(&counts[1], my_orphan);
   if (my_single) atomicAdd(&counts[2], my_single);
   if (my_complete) atomicAdd(&counts[3], my_complete);
   if (my_bad) atomicOr(a.flags_out, my_bad);
   __syncthreads();
   // ---- ranks in completion order: prefix count of the completing records over the arrival order
   int *flag = idx; // (the sort's indices are no longer needed)
   for (int i = tid; i < n; i += THREADS) flag[i] = a.fate[q0 + i] >= kRecCompletesAsRight ? 1 : 0;
   __syncthreads();
   block_exclusive_scan<THREADS>(flag, n, partial);
   int lf = 0, rf = 0;
   for (int i = tid; i < n; i += THREADS) {
      const int64_t r = q0 + i;
      const int8_t f = a.fate[r];
      if (f < kRecCompletesAsRight) continue;
      a.rank[r] = flag[i];
      const int nf_me = mate_feature_count(a, r);
      const int nf_w = a.partner[r] >= 0 ? mate_feature_count(a, q0 + a.partner[r]) : 0;
      const bool me_right = f == kRecCompletesAsRight || f == kRecSingleRight;
      lf += me_right ? nf_w : nf_me;
      rf += me_right ? nf_me : nf_w;
   }
   if (lf) atomicAdd(&counts[4], lf);
   if (rf) atomicAdd(&counts[5], rf);
   __syncthreads();
   if (tid == 0) {
      a.n_refused[l] = counts[0];
      a.n_orphan[l] = counts[1];
      a.n_single[l] = counts[2];
      a.n_complete[l] = counts[3];
      a.n_pairs[l] = counts[2] + counts[3];
      a.n_lfeat[l] = counts[4];
      a.n_rfeat[l] = counts[5];
   }
   __syncthreads();
}

// CAP: the LDS arrays' capacity.  Three instantiations: clusters of up to kMateSmallReads records (12 KB of LDS: many
// workgroups per CU -- with the full-size arrays a CU holds ONE workgroup, and 20 000 clusters of a thousand records
// take their turns 78 deep), up to kMateMidReads (48 KB) and up to kMateMaxReads; each serves the clusters in (LO, CAP].
constexpr int kMateSmallReads = 1024, kMateMidReads = 4096;
template <int CAP, int LO>
__global__ __launch_bounds__(kMateThreads) void matepair_locus_kernel(MateArgs a)
{
   __shared__ unsigned long long key[CAP];
   __shared__ int idx[CAP];
   __shared__ int partial[kMateThreads];
   __shared__ int counts[8];
   const int tid = threadIdx.x;
   for (int64_t l = blockIdx.x; l < a.n_loci; l += gridDim.x) {
      const int64_t nl = a.locus_read_off[l + 1] - a.locus_read_off[l];
      if (nl > CAP || nl <= LO) continue; // another instantiation's, or matepair_big_kernel's
      if (tid < 8) counts[tid] = 0;
      __syncthreads();
      matepair_one_locus<kMateThreads>(a, l, (int)nl, key, idx, partial, counts);
   }
}

// Clusters of more than kMateMaxReads records (a highly expressed gene): the same steps with 1024 threads and the sort's
// arrays in global scratch, one workgroup per such cluster.
constexpr int kMateBigThreads = 1024;
struct MateBigArgs {
   int32_t n_big;
   const int32_t *loci;     // [n_big]
   const int64_t *big_off;  // [n_big + 1] first scratch element of each (pow2ceil of its records)
   unsigned long long *key; // [big_off[n_big]]
   int *idx, *cl, *cr;      // idx: the sort; cl / cr: the fill kernel's feature counts
};

__global__ __launch_bounds__(kMateBigThreads) void matepair_big_kernel(MateArgs a, MateBigArgs b)
{
   __shared__ int partial[kMateBigThreads];
   __shared__ int counts[8];
   __shared__ unsigned long long stage_k[4096];
   __shared__ int stage_i[4096];
   const int tid = threadIdx.x;
   for (int i = blockIdx.x; i < b.n_big; i += gridDim.x) {
      const int64_t l = b.loci[i], o = b.big_off[i];
      if (tid < 8) counts[tid] = 0;
      __syncthreads();
      matepair_one_locus<kMateBigThreads>(a, l, (int)(a.locus_read_off[l + 1] - a.locus_read_off[l]), b.key + o, b.idx + o, partial, counts, stage_k, stage_i);
   }
}

// a record's aligned blocks as MATCH / INTRON features (readhit_2_genomicFeats, src/contig.cpp:12-53)
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

template <int THREADS>
__device__ __forceinline__ void matepair_fill_one(const MateArgs &a, int64_t l, int *cl, int *cr, int *partial)
{
   const int tid = threadIdx.x;
   const int64_t q0 = a.locus_read_off[l];
   const int n = (int)(a.locus_read_off[l + 1] - q0);
   const int64_t p0 = a.pair_off[l];
   const int np = (int)(a.pair_off[l + 1] - p0);
   if (np == 0) return;
   for (int i = tid; i < n; i += THREADS) {
      const int64_t r = q0 + i;
      const int8_t f = a.fate[r];
      if (f < kRecCompletesAsRight) continue;
      const int k = a.rank[r];
      const int nf_me = mate_feature_count(a, r);
      const int64_t w = a.partner[r] >= 0 ? q0 + a.partner[r] : -1;
      const int nf_w = w >= 0 ? mate_feature_count(a, w) : 0;
      const bool me_right = f == kRecCompletesAsRight || f == kRecSingleRight;
      cl[k] = me_right ? nf_w : nf_me;
      cr[k] = me_right ? nf_me : nf_w;
   }
   __syncthreads();
   block_exclusive_scan<THREADS>(cl, np, partial);
   block_exclusive_scan<THREADS>(cr, np, partial);
   for (int i = tid; i < n; i += THREADS) {
      const int64_t r = q0 + i;
      const int8_t f = a.fate[r];
      if (f < kRecCompletesAsRight) continue;
      const int k = a.rank[r];
      const int64_t w = a.partner[r] >= 0 ? q0 + a.partner[r] : -1;
      const bool me_right = f == kRecCompletesAsRight || f == kRecSingleRight;
      const int64_t rl = me_right ? w : r, rr = me_right ? r : w; // the left / right mate's record (-1: none)
      const int64_t lo = a.lfeat_base[l] + cl[k], ro = a.rfeat_base[l] + cr[k];
      a.left_off[p0 + k] = lo;
      a.right_off[p0 + k] = ro;
      if (rl >= 0) write_mate(a, rl, a.left_code, a.left_left, a.left_right, lo);
      if (rr >= 0) write_mate(a, rr, a.right_code, a.right_left, a.right_right, ro);
      // the reads' masses (src/read.cpp:49-53, 734-741)
      double m = 0.0;
      if (w >= 0) m = 0.5 / (double)a.nh[rl] + 0.5 / (double)a.nh[rr];
      else m = 1.0 / (double)a.nh[r];
      a.pair_mass[p0 + k] = m;
   }
   __syncthreads();
}

template <int CAP, int LO>
__global__ __launch_bounds__(kMateThreads) void matepair_fill_kernel(MateArgs a)
{
   __shared__ int cl[CAP], cr[CAP]; // per pair (rank order): feature counts, then offsets
   __shared__ int partial[kMateThreads];
   for (int64_t l = blockIdx.x; l < a.n_loci; l += gridDim.x) {
      const int64_t nl = a.locus_read_off[l + 1] - a.locus_read_off[l];
      if (nl > CAP || nl <= LO) continue; // another instantiation's, or matepair_big_fill_kernel's
      matepair_fill_one<kMateThreads>(a, l, cl, cr, partial);
   }
}

__global__ __launch_bounds__(kMateBigThreads) void matepair_big_fill_kernel(MateArgs a, MateBigArgs b)
{
   __shared__ int partial[kMateBigThreads];
   for (int i = blockIdx.x; i < b.n_big; i += gridDim.x) matepair_fill_one<kMateBigThreads>(a, b.loci[i], b.cl + b.big_off[i], b.cr + b.big_off[i], partial);
}

// ------------------------------------------------------------------ cluster streaming (sbgpu_assign_reads_device)
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
   const int64_t stride = (int64_t)gridDim.x * blockDim.x;
   const int64_t n_in = min(a.n_reads, a.pos[a.n_clusters]); // the records inside some cluster's pass
   const int lane = (int)(threadIdx.x & 63u);
   for (int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x - lane; i0 < a.n_reads; i0 += stride) { // (a wave's records: uniform)
      const int64_t i = i0 + lane;
      // the last cluster whose pass begins at or before record i and is not empty there: pos[k] <= i < pos[k + 1] (the
      // wave's search, device_common.h)
      const int64_t kw = i0 < n_in ? wave_range_of(a.pos, a.n_clusters, i0, n_in) : 0;
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
