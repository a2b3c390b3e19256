// strawberry_amd/csrc/collapse_device.h -- HitCluster::collapseAndFilterHits on the GPU.
//
// /root/reference/src/alignments.cpp:656-703: the read pairs of a cluster are sorted by (left end, right end),
// pairs with a mate whose reference span is an outlier are skipped, the others add their raw mass to the
// cluster's mass and are collapsed with the previous unique hit when both mates are equal; every unique hit
// then becomes a Contig (src/contig.cpp:216-267).  One workgroup per locus:
//   1. sort keys (left << 32 | right) and the pairs' input indices go to LDS; a bitonic sort on (key, index)
//      gives the reference's order with ties in input order (what sbgpu_collapse_pairs_host's stable sort gives);
//   2. the mates' spans are integers, so their sum -- and the mean -- are exact in any order; the sum of squared
//      deviations is not, so one thread adds it in input order, as std::inner_product does (common.h:100-110);
//   3. the filter, "equal to the previous kept pair" and Contig(PairedHit)'s feature count are per-pair work;
//      the cluster's mass is one sequential double sum in sorted order, a unique hit's mass one per group;
//   4. after the host has turned the per-locus counts into offsets, a second kernel writes the unique hits
//      (sbgpu_hits_t layout) where the exon-bin kernel reads them.
// Loci of more than 4096 pairs take the same steps with their arrays in global memory (collapse_big_kernel).
// Limit, reported through a flag (the caller then uses sbgpu_collapse_pairs_host): at most 24 features per mate.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "bitonic_big.h"

namespace sb {

constexpr int kCollapseMax = 4096;   // pairs of one locus (LDS sort)
constexpr int kCollapseThreads = 256;
constexpr int kMateFeatMax = 24;     // features of one mate the main kernels handle (the merged list in LDS up to four per mate -- collapse_flat.h -- else in private memory)
constexpr int kMateFeatLong = 512;   // ... and the flat form's kernels for long mates (long reads: the list in private memory)
enum : int32_t { kCollapseTooMany = 1, kCollapseLongMate = 2, kCollapseNoMates = 4 };

struct CollapseArgs {
   int64_t n_loci;
   const int64_t *locus_pair_off; // [n_loci + 1] pairs are grouped by locus
   const double *pair_mass;
   const int64_t *left_off, *right_off;
   const uint8_t *left_code, *right_code;
   const uint32_t *left_left, *left_right, *right_left, *right_right;
   // per sorted position of a locus (index locus_pair_off[l] + i)
   int32_t *order;   // input index (inside the locus) of the pair at sorted position i
   int32_t *nfeat;   // > 0: a unique hit with that many features starts here; 0: none
   float *mass;      // its collapse mass
   // per locus
   double *cluster_mass;
   int32_t *n_hits, *n_feats, *n_filtered, *n_rejected;
   int32_t *flags;
   // pass 2
   const int64_t *hit_off, *feat_base; // [n_loci + 1]: first unique hit / first feature of each locus
   int32_t *hit_locus;
   int64_t *feat_off; // [n_hits + 1]
   uint8_t *feat_code;
   uint32_t *feat_left, *feat_right;
   float *hit_mass;
};

struct MateRef {
   const uint8_t *c;
   const uint32_t *l, *r;
   int n;
};
__device__ __forceinline__ MateRef left_mate(const CollapseArgs &a, int64_t p)
{
   const int64_t o = a.left_off[p];
   return MateRef{a.left_code + o, a.left_left + o, a.left_right + o, (int)(a.left_off[p + 1] - o)};
}
__device__ __forceinline__ MateRef right_mate(const CollapseArgs &a, int64_t p)
{
   const int64_t o = a.right_off[p];
   return MateRef{a.right_code + o, a.right_left + o, a.right_right + o, (int)(a.right_off[p + 1] - o)};
}
// ReadHit::operator== (src/read.cpp:196-207): same start, same CIGAR
__device__ __forceinline__ bool mate_equal(const MateRef &x, const MateRef &y)
{
   if (x.n != y.n) return false;
   for (int i = 0; i < x.n; ++i)
      if (x.c[i] != y.c[i] || x.l[i] != y.l[i] || x.r[i] != y.r[i]) return false;
   return true;
}
// include/common.h:112-134
__device__ __forceinline__ double ref_phi_dev(double x)
{
   const double a1 = 0.254829592, a2 = -0.284496736, a3 = 1.421413741, a4 = -1.453152027, a5 = 1.061405429, p = 0.3275911;
   const int sign = x < 0 ? -1 : 1;
   x = fabs(x) / sqrt(2.0);
   const double t = 1.0 / (1.0 + p * x);
   const double y = 1.0 - (((((a5 * t + a4) * t) + a3) * t + a2) * t + a1) * t * exp(-x * x);
   return 0.5 * (1.0 + sign * y);
}

// Contig(PairedHit), src/contig.cpp:216-267, as sbgpu_hit_features (locus_bins.cpp) does it: a GAP between
// mates that are apart; else the features of both sorted by (offset, length) and merged (contig.h:111-137):
// equal introns fuse, overlapping blocks fuse, two different introns or blocks that merely abut reject the pair.
// Returns the number of features (0: rejected); writes them when `out_*` are given.
struct Feat {
   uint32_t l, r;
   uint8_t c;
};
__device__ __forceinline__ bool feat_less(const Feat &x, const Feat &y) { return x.l != y.l ? x.l < y.l : (x.r - x.l) < (y.r - y.l); }
// Where the merged list lives while it is sorted and fused.  Private memory (scratch) serves any length; for the short
// mates of short reads the list sits in the workgroup's LDS, a column per thread: through scratch every pair of the
// sorted-order kernels cost a round trip of its lines to HBM (profiles/r05_c3front_pmc_summary.json: flat_heads wrote
// 17 GB for 1.8 GB of results, flat_fill 33 GB for 8 GB).
template <int N>
struct FeatsPrivate {
   Feat g[N];
   __device__ __forceinline__ Feat get(int i) const { return g[i]; }
   __device__ __forceinline__ void set(int i, const Feat &f) { g[i] = f; }
};
struct FeatsLds { // entry i of this thread: lr[i * stride], c[i * stride] (the pointers carry the thread's own offset)
   uint2 *lr;
   uint8_t *c;
   int stride;
   __device__ __forceinline__ Feat get(int i) const
   {
      const uint2 v = lr[i * stride];
      return Feat{v.x, v.y, c[i * stride]};
   }
   __device__ __forceinline__ void set(int i, const Feat &f) const
   {
      lr[i * stride] = uint2{f.l, f.r};
      c[i * stride] = f.c;
   }
};
template <class G>
__device__ __forceinline__ int hit_features_in(G &g, const MateRef &a, const MateRef &b, uint8_t *out_c, uint32_t *out_l, uint32_t *out_r)
{
   int n = 0;
   auto push_sorted = [&](const Feat &f) { // insertion keeps g sorted by (offset, length)
      int pos = n++;
      while (pos > 0) {
         const Feat q = g.get(pos - 1);
         if (!feat_less(f, q)) break;
         g.set(pos, q);
         --pos;
      }
      g.set(pos, f);
   };
   for (int i = 0; i < a.n; ++i) push_sorted(Feat{a.l[i], a.r[i], a.c[i]});
   for (int i = 0; i < b.n; ++i) push_sorted(Feat{b.l[i], b.r[i], b.c[i]});
   if (a.n && b.n) {
      const int64_t gap = (int64_t)b.l[0] - (int64_t)a.r[a.n - 1] - 1; // :234
      if (gap > 0) {
         push_sorted(Feat{a.r[a.n - 1] + 1, (uint32_t)(a.r[a.n - 1] + gap), 2});
      } else {
         int m = 0;
         for (int i = 0; i < n; ++i) {
            Feat f = g.get(i);
            while (i + 1 < n) {
               const Feat q = g.get(i + 1);
               if (f.c != q.c) break;
               if (f.c == 1) {
                  if (!(f.l == q.l && f.r == q.r)) return 0; // two different introns
               } else {
                  if (f.r < q.l) return 0; // blocks that do not overlap (abutting included)
                  f.r = max(f.r, q.r);
               }
               ++i;
            }
            g.set(m++, f);
         }
         n = m;
         // merged blocks keep their offsets, so the (offset, length) order can only change among equal offsets
         for (int i = 1; i < n; ++i) {
            const Feat f = g.get(i);
            int pos = i;
            while (pos > 0) {
               const Feat q = g.get(pos - 1);
               if (!feat_less(f, q)) break;
               g.set(pos, q);
               --pos;
            }
            g.set(pos, f);
         }
      }
   }
   if (out_c)
      for (int i = 0; i < n; ++i) {
         const Feat f = g.get(i);
         out_c[i] = f.c;
         out_l[i] = f.l;
         out_r[i] = f.r;
      }
   return n;
}
// CAP: the most features a mate may have (the merged list in private memory: 2 CAP + 1 entries)
template <int CAP>
__device__ inline int hit_features_cap(const MateRef &a, const MateRef &b, uint8_t *out_c, uint32_t *out_l, uint32_t *out_r)
{
   FeatsPrivate<2 * CAP + 1> g;
   return hit_features_in(g, a, b, out_c, out_l, out_r);
}

__device__ inline int hit_features_dev(const MateRef &a, const MateRef &b, uint8_t *out_c, uint32_t *out_l, uint32_t *out_r)
{
   return hit_features_cap<kMateFeatMax>(a, b, out_c, out_l, out_r);
}

__device__ __forceinline__ uint32_t pair_left_pos(const MateRef &a, const MateRef &b)
{
   if (a.n && b.n) return min(a.l[0], b.l[0]); // PairedHit::left_pos, src/read.cpp:797-807
   return a.n ? a.l[0] : b.l[0];
}
__device__ __forceinline__ uint32_t pair_right_pos(const MateRef &a, const MateRef &b)
{
   if (a.n && b.n) return max(a.r[a.n - 1], b.r[b.n - 1]);
   return b.n ? b.r[b.n - 1] : a.r[a.n - 1];
}

// What a workgroup shares while it serves one locus (LDS in both forms)
struct CollapseShared {
   double mean, sd5;
   int nmates, hits, feats, filt, rej, bad;
};

// One locus by one workgroup of THREADS threads.  key / idx [n2 = pow2ceil(np)], span_l / span_r / skip [np]: LDS for
// loci of up to kCollapseMax pairs, global scratch for bigger ones (collapse_big_kernel) -- the same steps either way.
// stage_k / stage_i (CH elements each, LDS): given by the global-memory form only -- the sort's chunk buffers, and where the
// two one-thread sums read their operands from (a dependent global load per step costs a microsecond; 10^5 of them, 0.1 s)
constexpr int kCollapseStage = 4096;
template <int THREADS>
__device__ __forceinline__ void collapse_one_locus(const CollapseArgs &a, int64_t l, int np, unsigned long long *key, int *idx, int *span_l,
                                                   int *span_r, unsigned char *skip, double *red, CollapseShared &sh,
                                                   unsigned long long *stage_k = nullptr, int *stage_i = nullptr)
{
   const int tid = threadIdx.x;
   const int64_t q0 = a.locus_pair_off[l];
   int n2 = 1;
   while (n2 < np) n2 <<= 1;
   // ---- keys, spans
   int bad = 0;
   double span_sum = 0.0;
   int n_mates = 0;
   for (int i = tid; i < n2; i += THREADS) {
      unsigned long long k = ~0ull;
      if (i < np) {
         const MateRef x = left_mate(a, q0 + i), y = right_mate(a, q0 + i);
         if (x.n > kMateFeatMax || y.n > kMateFeatMax) bad |= kCollapseLongMate;
         if (x.n == 0 && y.n == 0) {
            bad |= kCollapseNoMates;
         } else {
            k = ((unsigned long long)pair_left_pos(x, y) << 32) | pair_right_pos(x, y);
         }
         span_l[i] = x.n ? (int)(x.r[x.n - 1] - x.l[0] + 1) : -1;
         span_r[i] = y.n ? (int)(y.r[y.n - 1] - y.l[0] + 1) : -1;
         if (x.n) span_sum += (double)span_l[i], ++n_mates;
         if (y.n) span_sum += (double)span_r[i], ++n_mates;
      }
      key[i] = k;
      idx[i] = i;
   }
   if (bad) atomicOr(&sh.bad, bad);
   // the spans are whole numbers: their sum is exact whatever the order
   red[tid] = span_sum;
   __syncthreads();
   for (int w = THREADS / 2; w > 0; w >>= 1) {
      if (tid < w) red[tid] += red[tid + w];
      __syncthreads();
   }
   const double total_span = red[0];
   __syncthreads();
   red[tid] = (double)n_mates;
   __syncthreads();
   for (int w = THREADS / 2; w > 0; w >>= 1) {
      if (tid < w) red[tid] += red[tid + w];
      __syncthreads();
   }
   if (tid == 0) {
      sh.nmates = (int)red[0];
      sh.mean = total_span / red[0];
   }
   __syncthreads();
   if (sh.bad) {
      if (tid == 0) atomicOr(a.flags, sh.bad);
      __syncthreads();
      return;
   }
   // ---- bitonic sort of (key, input index)
   if (stage_k) bitonic_sort_global<THREADS, kCollapseStage>(key, idx, n2, stage_k, stage_i);
   else
   for (int k2 = 2; k2 <= n2; k2 <<= 1)
      for (int j = k2 >> 1; j > 0; j >>= 1) {
         for (int i = tid; i < n2; i += THREADS) {
            const int p = i ^ j;
            if (p > i) {
               const bool up = (i & k2) == 0;
               const unsigned long long ki = key[i], kp = key[p];
               const int ii = idx[i], ip = idx[p];
               const bool greater = ki > kp || (ki == kp && ii > ip);
               if (greater == up) {
                  key[i] = kp, key[p] = ki;
                  idx[i] = ip, idx[p] = ii;
               }
            }
         }
         __syncthreads();
      }
   // ---- sd: the squared deviations added in INPUT order (left mate, then right mate of each pair), by one thread
   if (!stage_k) {
      if (tid == 0) {
         const double mean = sh.mean;
         double sq = 0.0;
         for (int i = 0; i < np; ++i) {
            if (span_l[i] >= 0) {
               const double d = (double)span_l[i] - mean;
               sq += d * d;
            }
            if (span_r[i] >= 0) {
               const double d = (double)span_r[i] - mean;
               sq += d * d;
            }
         }
         sh.sd5 = sqrt(sq / (double)sh.nmates) * 5;
      }
   } else {
      // the same sum, the spans staged through LDS a chunk at a time (the order of the additions is unchanged)
      int *sl = stage_i, *sr = (int *)stage_k;
      double sq = 0.0;
      for (int c0 = 0; c0 < np; c0 += kCollapseStage) {
         const int m = min(kCollapseStage, np - c0);
         for (int t = tid; t < m; t += THREADS) sl[t] = span_l[c0 + t], sr[t] = span_r[c0 + t];
         __syncthreads();
         if (tid == 0) {
            const double mean = sh.mean;
            for (int i = 0; i < m; ++i) {
               if (sl[i] >= 0) {
                  const double d = (double)sl[i] - mean;
                  sq += d * d;
               }
               if (sr[i] >= 0) {
                  const double d = (double)sr[i] - mean;
                  sq += d * d;
               }
            }
         }
         __syncthreads();
      }
      if (tid == 0) sh.sd5 = sqrt(sq / (double)sh.nmates) * 5;
   }
   __syncthreads();
   // ---- the span filter (:670-682), and the masses in sorted order
   double *pmass = (double *)key;
   for (int i = tid; i < np; i += THREADS) {
      const int p = idx[i];
      a.order[q0 + i] = p;
      bool sk = false;
      if (span_l[p] >= 0 && ref_phi_dev(((double)(uint32_t)span_l[p] - sh.mean) / sh.sd5) > 0.999) sk = true;
      if (span_r[p] >= 0 && ref_phi_dev(((double)(uint32_t)span_r[p] - sh.mean) / sh.sd5) > 0.999) sk = true;
      skip[i] = sk ? 1 : 0;
   }
   __syncthreads(); // everybody is done with key[] as keys
   for (int i = tid; i < np; i += THREADS) pmass[i] = a.pair_mass[q0 + idx[i]];
   __syncthreads();
   // ---- the cluster's mass: kept pairs in sorted order, one running double (:683-684)
   if (!stage_k) {
      if (tid == 0) {
         double m = 0.0;
         for (int i = 0; i < np; ++i)
            if (!skip[i]) m += pmass[i];
         a.cluster_mass[l] = m;
      }
   } else {
      double *pm = (double *)stage_k; // (staged like the spans above; skipped pairs go in as -1)
      double m = 0.0;
      for (int c0 = 0; c0 < np; c0 += kCollapseStage) {
         const int mm = min(kCollapseStage, np - c0);
         __syncthreads();
         for (int t = tid; t < mm; t += THREADS) pm[t] = skip[c0 + t] ? -1.0 : pmass[c0 + t];
         __syncthreads();
         if (tid == 0)
            for (int i = 0; i < mm; ++i)
               if (pm[i] >= 0.0) m += pm[i];
      }
      if (tid == 0) a.cluster_mass[l] = m;
      __syncthreads();
   }
   // ---- unique hits: a kept pair that differs from the previous kept pair (:685-697)
   int my_hits = 0, my_feats = 0, my_filt = 0, my_rej = 0;
   for (int i = tid; i < np; i += THREADS) {
      a.nfeat[q0 + i] = 0;
      a.mass[q0 + i] = 0.0f;
      if (skip[i]) {
         ++my_filt;
         continue;
      }
      const MateRef x = left_mate(a, q0 + idx[i]), y = right_mate(a, q0 + idx[i]);
      int prev = i - 1;
      while (prev >= 0 && skip[prev]) --prev;
      if (prev >= 0 && mate_equal(left_mate(a, q0 + idx[prev]), x) && mate_equal(right_mate(a, q0 + idx[prev]), y)) continue;
      // head of a group: its mass = the members' masses added in order, in double; stored as float (Contig::mass())
      double m = pmass[i];
      for (int k = i + 1; k < np; ++k) {
         if (skip[k]) continue;
         if (!(mate_equal(left_mate(a, q0 + idx[k]), x) && mate_equal(right_mate(a, q0 + idx[k]), y))) break;
         m += pmass[k];
      }
      const int nf = hit_features_dev(x, y, nullptr, nullptr, nullptr);
      if (nf <= 0) {
         ++my_rej; // Contig(PairedHit) rejects the pair: no hit, its mass stays in the cluster's
         continue;
      }
      a.nfeat[q0 + i] = nf;
      a.mass[q0 + i] = (float)m;
      ++my_hits;
      my_feats += nf;
   }
   if (my_hits) atomicAdd(&sh.hits, my_hits);
   if (my_feats) atomicAdd(&sh.feats, my_feats);
   if (my_filt) atomicAdd(&sh.filt, my_filt);
   if (my_rej) atomicAdd(&sh.rej, my_rej);
   __syncthreads();
   if (tid == 0) {
      a.n_hits[l] = sh.hits;
      a.n_feats[l] = sh.feats;
      a.n_filtered[l] = sh.filt;
      a.n_rejected[l] = sh.rej;
   }
   __syncthreads();
}

// CAP: the LDS arrays' capacity.  Two instantiations: loci of up to kCollapseSmall pairs (23 KB of LDS: seven workgroups
// per CU; with the full-size arrays a CU holds one) and loci of up to kCollapseMax; each serves the loci in (LO, CAP].
constexpr int kCollapseSmall = 1024;
template <int CAP, int LO>
__global__ __launch_bounds__(kCollapseThreads) void collapse_locus_kernel(CollapseArgs a)
{
   __shared__ unsigned long long key[CAP]; // sort keys; afterwards the pairs' masses (as doubles)
   __shared__ int idx[CAP];
   __shared__ int span_l[CAP], span_r[CAP]; // by input index; -1: no such mate
   __shared__ unsigned char skip[CAP];      // by sorted position
   __shared__ double red[kCollapseThreads];
   __shared__ CollapseShared sh;
   const int tid = threadIdx.x;
   for (int64_t l = blockIdx.x; l < a.n_loci; l += gridDim.x) {
      const int64_t q0 = a.locus_pair_off[l];
      const int64_t npl = a.locus_pair_off[l + 1] - q0;
      if (npl > CAP || npl <= LO) continue; // another instantiation's, or collapse_big_kernel's
      if (tid == 0) {
         a.cluster_mass[l] = 0.0;
         a.n_hits[l] = a.n_feats[l] = a.n_filtered[l] = a.n_rejected[l] = 0;
         sh.hits = sh.feats = sh.filt = sh.rej = sh.bad = 0;
      }
      __syncthreads();
      if (npl == 0) continue;
      collapse_one_locus<kCollapseThreads>(a, l, (int)npl, key, idx, span_l, span_r, skip, red, sh);
   }
}

// Loci of more than kCollapseMax pairs (any highly expressed gene): the same steps with the arrays in global scratch
// and 1024 threads -- the bitonic sort then runs through the caches (a locus of 10^5 pairs: 153 passes of 64
// compare-exchanges per thread), the two order-bound sums stay one thread's loops.  One workgroup per such locus.
constexpr int kCollapseBigThreads = 1024;
struct CollapseBigArgs {
   int32_t n_big;
   const int32_t *loci;      // [n_big] their locus numbers
   const int64_t *big_off;   // [n_big + 1] first scratch element of each (in units of its n2: pow2ceil of its pairs)
   unsigned long long *key;  // [big_off[n_big]]
   int *idx, *span_l, *span_r;
   unsigned char *skip;
};

__global__ __launch_bounds__(kCollapseBigThreads) void collapse_big_kernel(CollapseArgs a, CollapseBigArgs b)
{
   __shared__ double red[kCollapseBigThreads];
   __shared__ CollapseShared sh;
   __shared__ unsigned long long stage_k[kCollapseStage];
   __shared__ int stage_i[kCollapseStage];
   const int tid = threadIdx.x;
   for (int i = blockIdx.x; i < b.n_big; i += gridDim.x) {
      const int64_t l = b.loci[i], o = b.big_off[i];
      const int64_t npl = a.locus_pair_off[l + 1] - a.locus_pair_off[l];
      if (tid == 0) {
         a.cluster_mass[l] = 0.0;
         a.n_hits[l] = a.n_feats[l] = a.n_filtered[l] = a.n_rejected[l] = 0;
         sh.hits = sh.feats = sh.filt = sh.rej = sh.bad = 0;
      }
      __syncthreads();
      collapse_one_locus<kCollapseBigThreads>(a, l, (int)npl, b.key + o, b.idx + o, b.span_l + o, b.span_r + o, b.skip + o, red, sh, stage_k, stage_i);
   }
}

// pass 2: the unique hits of every locus at their final places
__global__ __launch_bounds__(kCollapseThreads) void collapse_fill_kernel(CollapseArgs a)
{
   __shared__ int cnt_h[kCollapseThreads], cnt_f[kCollapseThreads];
   const int tid = threadIdx.x;
   for (int64_t l = blockIdx.x; l < a.n_loci; l += gridDim.x) {
      const int64_t q0 = a.locus_pair_off[l];
      const int np = (int)(a.locus_pair_off[l + 1] - q0);
      const int64_t h0 = a.hit_off[l], f0 = a.feat_base[l];
      // thread t owns the sorted positions [t * per, t * per + per): hits come out in sorted order
      const int per = (np + kCollapseThreads - 1) / kCollapseThreads;
      int nh = 0, nf = 0;
      for (int i = tid * per; i < min(np, tid * per + per); ++i)
         if (a.nfeat[q0 + i] > 0) {
            ++nh;
            nf += a.nfeat[q0 + i];
         }
      cnt_h[tid] = nh;
      cnt_f[tid] = nf;
      __syncthreads();
      if (tid == 0) {
         int sh = 0, sf = 0;
         for (int t = 0; t < kCollapseThreads; ++t) {
            const int ch = cnt_h[t], cf = cnt_f[t];
            cnt_h[t] = sh, cnt_f[t] = sf;
            sh += ch, sf += cf;
         }
      }
      __syncthreads();
      int64_t h = h0 + cnt_h[tid], f = f0 + cnt_f[tid];
      for (int i = tid * per; i < min(np, tid * per + per); ++i) {
         const int n = a.nfeat[q0 + i];
         if (n <= 0) continue;
         const int64_t p = q0 + a.order[q0 + i];
         a.hit_locus[h] = (int32_t)l;
         a.feat_off[h] = f;
         a.hit_mass[h] = a.mass[q0 + i];
         hit_features_dev(left_mate(a, p), right_mate(a, p), a.feat_code + f, a.feat_left + f, a.feat_right + f);
         ++h;
         f += n;
      }
      __syncthreads();
   }
}

} // namespace sb
