// strawberry_amd/csrc/collapse_flat.h -- HitCluster::collapseAndFilterHits for ALL clusters of a call at once
// (/root/reference/src/alignments.cpp:656-703; round 4: replaces one-workgroup-per-cluster, collapse_device.h).
//
// The per-cluster form sorted each cluster's pairs with a bitonic network in LDS (55-105 barrier passes for a cluster of a
// few thousand pairs) or through global memory (a highly expressed gene: one workgroup of 1024 threads, milliseconds), and a
// sample's time was its biggest clusters'.  Here nothing but two order-bound sums is per cluster:
//
//   keys      a thread per pair: the sort keys, the mates' spans, the clusters' integer span sums (exact in any order);
//   sort      the order is std::sort's on (left, right) with ties in input order, which is what the host form's stable sort and
//             the per-cluster kernel's (key, index) network give (src/read.cpp:917-923).  ONE stable device-wide radix sort
//             (rocPRIM onesweep) where its key fits 64 bits -- a pair's left end relative to its cluster's leftmost, the
//             clusters' ranges laid end to end (ascending with (cluster, left)), the pair's length in the bits below; the host
//             reads the two widths back after the keys kernel.  Else round 4's TWO sorts, least significant key first: by the
//             right end, then by (cluster << 32 | left end);
//   sd        the reference adds the squared deviations of the spans in INPUT order, one running double (std::inner_product,
//             common.h:100-110) -- sequential by definition -- but the sum is only ever used in a PREDICATE, the span filter
//             phi((span - mean) / (5 sd)) > 0.999 (:666-682).  The spans are integers, so the exact sum of squared deviations
//             is (n S2 - S1^2) / n with S1 = sum of spans, S2 = sum of squares -- integers, exact in any order (128-bit).
//             The filter is decided with the sd from that; a decision that an error of n ulps in the sd could flip (phi
//             within 1e-9 + 1e-15 n (1 + |x|) of 0.999: not observed) marks its cluster, and marked clusters get the
//             reference's running sum after all (a WAVE per cluster: the values handed round with v_readlane, every lane
//             carrying the same sum) and their flags again;
//   flags     a thread per sorted position: the span filter;
//   heads     "differs from the previous KEPT pair" (:685-697): the previous kept position is a device-wide running
//             maximum (scan), the comparison and Contig(PairedHit)'s feature count are per-position work;
//   ranks     three device-wide scans give every unique hit its group, its place and its first feature's place -- the
//             per-cluster offsets are those scans read at the clusters' first positions (no host prefix sums in between);
//   masses    the cluster's mass and every group's are running doubles in SORTED order (:683-684, :688-696).  Where every
//             mass of a cluster is a multiple of 2^-20 below 2^31 -- always, unless multi-mapped reads are let in with NH 3, 5,
//             ... -- every partial sum is exact and the order cannot matter: a thread per position, the lanes of a wave that
//             share a cluster / a group add up in registers, one hardware fp64 atomic per run.  The other clusters get the
//             running sums (a wave per cluster, as for the sd);
//   fill      a thread per unique hit writes it where the exon-bin kernel reads it.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "collapse_device.h"
#include "device_common.h"

namespace sb {

struct FlatCollapseArgs {
   CollapseArgs a; // the pairs (inputs) and the unique hits' arrays (outputs of the fill)
   int64_t n_pairs;
   // per pair, input order
   uint32_t *key_right;          // sort 1
   unsigned long long *key_hi;   // (cluster << 32) | left end: sort 2, gathered through sort 1's permutation
   int32_t *span_l, *span_r;     // -1: no such mate
   // per cluster
   unsigned long long *span_sum; // S1: integer sum of the mates' spans
   unsigned long long *span_sq;  // S2: integer sum of their squares
   int32_t *n_mates;
   int32_t *cl_flags;            // kNeedSeqMass / kNeedSeqSd: this cluster's sums must run in the reference's order
   double *mean, *sd5;
   // per sorted position (the clusters keep their ranges: the cluster is the key's high part)
   const int32_t *perm1;              // after sort 1
   unsigned long long *key2;          // key_hi gathered by perm1 (sort 2's input)
   const unsigned long long *key2s;   // sort 2's keys out: cluster of a sorted position = key2s[s] >> 32 (two-sort form)
   // the one-sort form (round 5): a cluster's pairs lie in [min left, max left]; those ranges laid end to end give every
   // pair a "concatenated coordinate" X = base[cluster] + left - min left that ascends with (cluster, left), and
   // (X << span_bits) | (right - left) is ONE key for the whole order wherever it fits 64 bits
   uint32_t *cl_minl, *cl_maxl;       // per cluster (min starts at 0xFFFFFFFF, max at 0)
   unsigned long long *cl_span;       // per cluster (+ one entry: 0): max left - min left + 1, 0 for a cluster without pairs
   const unsigned long long *cl_base; // its exclusive scan (one entry beyond the end: the total)
   unsigned long long *order_stats;   // [0] the longest pair (right - left) of the call, [1] != 0: a pair with right < left (two sorts then)
   int span_bits;
   int32_t *cluster_of;               // per sorted position: its cluster (both forms; what the later kernels read)
   const int32_t *order;              // input pair (global index) at sorted position s
   uint8_t *skip, *head;
   int32_t *kept_pos;                 // s where kept, -1 where skipped; scanned (running max) into last_kept
   const int32_t *last_kept;
   int32_t *nfeat;                    // > 0: a unique hit with that many features starts here
   int32_t *is_hit;                   // nfeat > 0 (scan input)
   const int32_t *gid;                // inclusive scan of head: the group of a kept position (1-based)
   const int64_t *hit_rank;           // exclusive scan of is_hit (one entry beyond the end: the total)
   const int64_t *feat_base;          // exclusive scan of nfeat  (likewise)
   double *gmass;                     // [groups + 1]
   int64_t *locus_hit_off;            // [n_loci + 1]
   unsigned long long *counts;        // [0] filtered, [1] rejected, [2] != 0: a pair with a long mate was met
};

enum : int32_t { kNeedSeqMass = 1, kNeedSeqSd = 2 };

// The merged feature list of a pair of SHORT mates (at most kShortFeat features each: every short read) in the
// workgroup's LDS; longer mates' in private memory.  `lr` / `c`: (2 kShortFeat + 1) x 256 entries of the calling workgroup.
constexpr int kShortFeat = 4;
constexpr int kShortSlots = 2 * kShortFeat + 1;
__device__ __forceinline__ int flat_hit_features(const MateRef &x, const MateRef &y, uint2 *lr, uint8_t *c, uint8_t *out_c, uint32_t *out_l,
                                                 uint32_t *out_r)
{
   // Mates that lie apart (a GAP between them: the usual paired-end fragment), or a single mate: the list is the left mate's
   // features, the GAP, the right mate's -- already in (offset, length) order when each mate's own features ascend, which
   // is checked while they are read (a caller's pairs need not; sbgpu_pair_mates_* write them so).  Nothing to sort or fuse.
   {
      const bool both = x.n > 0 && y.n > 0;
      bool plain = !both || (int64_t)y.l[0] - (int64_t)x.r[x.n - 1] - 1 > 0;
      for (int i = 1; plain && i < x.n; ++i) plain = x.l[i] > x.l[i - 1];
      for (int i = 1; plain && i < y.n; ++i) plain = y.l[i] > y.l[i - 1];
      if (plain) {
         if (out_c) {
            int n = 0;
            for (int i = 0; i < x.n; ++i, ++n) out_c[n] = x.c[i], out_l[n] = x.l[i], out_r[n] = x.r[i];
            if (both) out_c[n] = 2, out_l[n] = x.r[x.n - 1] + 1u, out_r[n] = y.l[0] - 1u, ++n;
            for (int i = 0; i < y.n; ++i, ++n) out_c[n] = y.c[i], out_l[n] = y.l[i], out_r[n] = y.r[i];
         }
         return x.n + y.n + (both ? 1 : 0);
      }
   }
   if (x.n <= kShortFeat && y.n <= kShortFeat) {
      FeatsLds g = {lr + threadIdx.x, c + threadIdx.x, 256};
      return hit_features_in(g, x, y, out_c, out_l, out_r);
   }
   return hit_features_dev(x, y, out_c, out_l, out_r);
}

// ---- keys: a thread per pair
__global__ __launch_bounds__(256) void flat_keys_kernel(FlatCollapseArgs f)
{
   const CollapseArgs &a = f.a;
   const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
   int bad = 0, need = 0;
   int32_t l = -1;
   unsigned long long sum = 0, sum2 = 0;
   int nm = 0;
   uint32_t my_lp = 0xffffffffu, my_rp = 0u; // the pair's ends (have_ends: it has mates)
   bool have_ends = false;
   const int64_t p0 = p - (int64_t)(threadIdx.x & 63u); // (the wave's first pair; the search is the wave's: device_common.h)
   const int32_t lw = p0 < f.n_pairs ? (int32_t)wave_range_of(a.locus_pair_off, a.n_loci, p0, f.n_pairs) : -1;
   if (p < f.n_pairs) {
      l = lw;
      const MateRef x = left_mate(a, p), y = right_mate(a, p);
      if (x.n > kMateFeatLong || y.n > kMateFeatLong) bad |= kCollapseLongMate;
      uint32_t lp = 0xffffffffu, rp = 0xffffffffu;
      if (x.n == 0 && y.n == 0) bad |= kCollapseNoMates;
      else lp = pair_left_pos(x, y), rp = pair_right_pos(x, y), my_lp = lp, my_rp = rp, have_ends = true;
      f.key_right[p] = rp;
      f.key_hi[p] = ((unsigned long long)(uint32_t)l << 32) | lp;
      const int sl = x.n ? (int)(x.r[x.n - 1] - x.l[0] + 1) : -1, sr = y.n ? (int)(y.r[y.n - 1] - y.l[0] + 1) : -1;
      f.span_l[p] = sl;
      f.span_r[p] = sr;
      if (sl >= 0) sum += (unsigned long long)sl, sum2 += (unsigned long long)sl * (unsigned long long)sl, ++nm;
      if (sr >= 0) sum += (unsigned long long)sr, sum2 += (unsigned long long)sr * (unsigned long long)sr, ++nm;
      if (sl >= (1 << 24) || sr >= (1 << 24)) need |= kNeedSeqSd; // (S2 could leave 64 bits)
      // a mass that is a multiple of 2^-20 and at most 2 (the reference's are 1 / NH and 0.5 / NH): sums of up to 2^31 of
      // them stay below 2^32 in units of 2^-20 -- 52 bits -- so every partial sum is exact whatever the order.  Anything
      // else (a caller's own masses) takes the running sums in the reference's order
      const double m = a.pair_mass[p], m20 = m * 1048576.0;
      if (!(m >= 0.0 && m <= 2.0 && m20 == (double)(unsigned long long)m20)) need |= kNeedSeqMass;
      if (need) atomicOr(&f.cl_flags[l], need);
   }
   if (bad) atomicOr(a.flags, bad);
   // the one-sort form's numbers: the cluster's leftmost and rightmost left end, the call's longest pair
   {
      const bool have = have_ends;
      const uint32_t klo = have ? my_lp : 0xffffffffu, khi = have ? my_lp : 0u, krp = have ? my_rp : 0u;
      const bool backwards = have && krp < klo;
      uint32_t span = have && !backwards ? krp - klo : 0u;
      const int32_t lf = __shfl(l, __ffsll((long long)__ballot(l >= 0)) - 1);
      const bool wf = have && l == lf;
      uint32_t mn = wf ? klo : 0xffffffffu, mx = wf ? khi : 0u;
      for (int o = 32; o > 0; o >>= 1) {
         mn = min(mn, (uint32_t)__shfl_xor((int)mn, o));
         mx = max(mx, (uint32_t)__shfl_xor((int)mx, o));
         span = max(span, (uint32_t)__shfl_xor((int)span, o));
      }
      const unsigned long long anyb = __ballot(backwards), anyw = __ballot(wf);
      if (have && !wf) {
         atomicMin(&f.cl_minl[l], klo);
         atomicMax(&f.cl_maxl[l], khi);
      } else if (wf && (int)(threadIdx.x & 63) == __ffsll((long long)anyw) - 1) {
         atomicMin(&f.cl_minl[l], mn);
         atomicMax(&f.cl_maxl[l], mx);
      }
      if ((threadIdx.x & 63) == 0) {
         // (a plain read first: three million waves' atomics on ONE word queue at the L2, ~10 ns each -- 25 ms; only a
         // wave that would raise the maximum goes there)
         if ((unsigned long long)span > __hip_atomic_load(&f.order_stats[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            atomicMax(&f.order_stats[0], (unsigned long long)span);
         if (anyb) f.order_stats[1] = 1ull; // (any writer, same value)
      }
   }
   // the spans are whole numbers: their sum is exact in any order.  The lanes of a wave mostly share their cluster:
   // the lanes whose cluster is the first active lane's add up in registers, one atomic for them; the others on their own
   const int32_t l0 = __shfl(l, __ffsll((long long)__ballot(l >= 0)) - 1);
   const bool with_first = l >= 0 && l == l0;
   unsigned long long s2 = with_first ? sum : 0, q2 = with_first ? sum2 : 0;
   int n2 = with_first ? nm : 0;
   for (int o = 32; o > 0; o >>= 1) {
      s2 += __shfl_xor(s2, o);
      q2 += __shfl_xor(q2, o);
      n2 += __shfl_xor(n2, o);
   }
   if (l >= 0) {
      if (!with_first) {
         atomicAdd(&f.span_sum[l], sum);
         atomicAdd(&f.span_sq[l], sum2);
         atomicAdd(&f.n_mates[l], nm);
      } else if ((int)(threadIdx.x & 63) == __ffsll((long long)__ballot(with_first)) - 1) {
         atomicAdd(&f.span_sum[l], s2);
         atomicAdd(&f.span_sq[l], q2);
         atomicAdd(&f.n_mates[l], n2);
      }
   }
}

__global__ __launch_bounds__(256) void flat_gather_kernel(FlatCollapseArgs f)
{
   const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
   if (s < f.n_pairs) f.key2[s] = f.key_hi[f.perm1[s]];
}

// ---- the one-sort form: the clusters' spans (a thread per cluster), then every pair's key (a thread per pair, input order)
__global__ __launch_bounds__(256) void flat_cluster_spans_kernel(FlatCollapseArgs f)
{
   const int64_t l = (int64_t)blockIdx.x * 256 + threadIdx.x;
   if (l > f.a.n_loci) return;
   const bool any = l < f.a.n_loci && f.cl_maxl[l] >= f.cl_minl[l];
   f.cl_span[l] = any ? (unsigned long long)(f.cl_maxl[l] - f.cl_minl[l]) + 1ull : 0ull;
}
__global__ __launch_bounds__(256) void flat_rekey_kernel(FlatCollapseArgs f)
{
   const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
   if (p >= f.n_pairs) return;
   const unsigned long long kh = f.key_hi[p];
   const int32_t l = (int32_t)(kh >> 32);
   const uint32_t lp = (uint32_t)kh;
   const unsigned long long x = f.cl_base[l] + (unsigned long long)(lp - f.cl_minl[l]);
   f.key2[p] = (x << f.span_bits) | (unsigned long long)(f.key_right[p] - lp);
}
// the cluster of every sorted position: the clusters keep their ranges under either order
template <bool FROM_KEYS>
__global__ __launch_bounds__(256) void flat_cluster_of_kernel(FlatCollapseArgs f)
{
   const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
   if (FROM_KEYS) {
      if (s < f.n_pairs) f.cluster_of[s] = (int32_t)(f.key2s[s] >> 32);
   } else {
      const int64_t s0 = s - (int64_t)(threadIdx.x & 63u);
      if (s0 >= f.n_pairs) return; // (the whole wave)
      const int32_t l = (int32_t)wave_range_of(f.a.locus_pair_off, f.a.n_loci, s0, f.n_pairs);
      if (s < f.n_pairs) f.cluster_of[s] = l;
   }
}

// a double held by lane k of the wave, as a wave-uniform value
__device__ __forceinline__ double flat_bcast(double v, int k)
{
   return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), k), __builtin_amdgcn_readlane(__double2loint(v), k));
}

// ---- sd: a wave per cluster; the squared deviations in INPUT order (left mate, then right mate of each pair).
// The lanes square their own element's deviations side by side; only the additions are sequential: the values come round as
// wave-uniform scalars (v_readlane) and every lane carries the same running sum.  A mate that is not there adds 0.0, which
// leaves a non-negative sum as it is (the reference skips it).  The next 64 elements are loaded while these are added.
__global__ __launch_bounds__(64) void flat_sd_kernel(FlatCollapseArgs f)
{
   const CollapseArgs &a = f.a;
   const int lane = threadIdx.x;
   for (int64_t l = blockIdx.x; l < a.n_loci; l += gridDim.x) {
      const int64_t q0 = a.locus_pair_off[l], q1 = a.locus_pair_off[l + 1];
      const int nm = f.n_mates[l];
      if (q1 == q0 || nm == 0 || !(f.cl_flags[l] & kNeedSeqSd)) continue; // (the usual cluster: decided from the integer moments)
      const double mean = (double)f.span_sum[l] / (double)nm;
      double sq = 0.0;
      int sl = q0 + lane < q1 ? f.span_l[q0 + lane] : -1, sr = q0 + lane < q1 ? f.span_r[q0 + lane] : -1;
      for (int64_t c0 = q0; c0 < q1; c0 += 64) {
         const int m = (int)min((int64_t)64, q1 - c0);
         const double el = (double)sl - mean, er = (double)sr - mean;
         const double dl = sl >= 0 ? el * el : 0.0, dr = sr >= 0 ? er * er : 0.0;
         const int64_t nx = c0 + 64 + lane; // the next chunk's loads, in flight during the additions below
         sl = nx < q1 ? f.span_l[nx] : -1;
         sr = nx < q1 ? f.span_r[nx] : -1;
         if (m == 64) {
#pragma unroll
            for (int k = 0; k < 64; ++k) {
               sq += flat_bcast(dl, k);
               sq += flat_bcast(dr, k);
            }
         } else {
            for (int k = 0; k < m; ++k) {
               sq += flat_bcast(dl, k);
               sq += flat_bcast(dr, k);
            }
         }
      }
      if (lane == 0) {
         f.mean[l] = mean;
         f.sd5[l] = sqrt(sq / (double)nm) * 5;
      }
   }
}

// ---- the span filter (:666-682): a thread per sorted position.  PASS 1 decides from the integer moments and marks the
// clusters where that is not safe; PASS 2 (after flat_sd_kernel has given those the reference's running sum) redoes theirs.
template <int PASS>
__global__ __launch_bounds__(256) void flat_flags_kernel(FlatCollapseArgs f)
{
   const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
   if (s >= f.n_pairs) return;
   const int32_t l = f.cluster_of[s];
   const int need = f.cl_flags[l] & kNeedSeqSd;
   if (PASS == 2 && !need) return;
   const int32_t p = f.order[s];
   const int sl = f.span_l[p], sr = f.span_r[p];
   bool sk = false;
   if (PASS == 2) {
      const double mean = f.mean[l], sd5 = f.sd5[l];
      if (sl >= 0 && ref_phi_dev(((double)(uint32_t)sl - mean) / sd5) > 0.999) sk = true;
      if (sr >= 0 && ref_phi_dev(((double)(uint32_t)sr - mean) / sd5) > 0.999) sk = true;
   } else if (!need) {
      const int nm = f.n_mates[l];
      const unsigned long long S1 = f.span_sum[l], S2 = f.span_sq[l];
      const double mean = (double)S1 / (double)nm; // (the reference's mean, to the bit: an exact integer over a count)
      const unsigned __int128 num = (unsigned __int128)S2 * (unsigned)nm - (unsigned __int128)S1 * S1; // n^2 x the variance: exact
      const double numd = (double)(unsigned long long)(num >> 64) * 18446744073709551616.0 + (double)(unsigned long long)num;
      const double sq = numd / (double)nm;         // the sum of squared deviations, to a few ulps
      const double sd5 = sqrt(sq / (double)nm) * 5;
      const double tol0 = 1e-9 + 1e-15 * (double)nm;
      // all spans equal (num == 0: every read unspliced and of one length): the reference's running sum is 0.0 too -- every
      // deviation is exactly 0 -- so its sd is this one to the bit (the filter then sees 0 / 0 and keeps the pair)
      bool unsure = false;
      const bool exact = num == 0;
      if (sl >= 0) {
         const double x = ((double)(uint32_t)sl - mean) / sd5, ph = ref_phi_dev(x);
         sk |= ph > 0.999;
         unsure |= !exact && !(fabs(ph - 0.999) > tol0 * (1.0 + fabs(x)));
      }
      if (sr >= 0) {
         const double x = ((double)(uint32_t)sr - mean) / sd5, ph = ref_phi_dev(x);
         sk |= ph > 0.999;
         unsure |= !exact && !(fabs(ph - 0.999) > tol0 * (1.0 + fabs(x)));
      }
      if (unsure) atomicOr(&f.cl_flags[l], kNeedSeqSd); // (a NaN that is not the all-equal case included)
   }
   f.skip[s] = sk ? 1 : 0;
   f.kept_pos[s] = sk ? -1 : (int32_t)s;
}

// A pair whose mates have at most kShortFeat features each, in registers: every load of the pair goes out together (a
// level of the kernels below), none waits for another.  Slots beyond a mate's features hold zeros, so two pairs are equal
// (ReadHit::operator== on both mates) exactly when every word is.
struct ShortPair {
   int nl, nr; // -1: not held here (a mate with more features): the arrays serve
   uint32_t l[2 * kShortFeat], r[2 * kShortFeat];
   uint32_t c[2]; // the codes, a byte per slot: [0] the left mate's, [1] the right mate's
};
struct PairOffsets {
   int64_t lo, ro;
   int nl, nr;
};
__device__ __forceinline__ PairOffsets flat_pair_offsets(const CollapseArgs &a, int64_t p, bool live)
{
   PairOffsets o = {0, 0, 0, 0};
   if (live) { // (four loads in flight together)
      const int64_t l0 = a.left_off[p], l1 = a.left_off[p + 1], r0 = a.right_off[p], r1 = a.right_off[p + 1];
      o.lo = l0, o.ro = r0, o.nl = (int)(l1 - l0), o.nr = (int)(r1 - r0);
   }
   return o;
}
// A mate's (up to) four features come with ONE 16-byte load per coordinate array and one 4-byte load of the codes, from its
// first feature on -- what lies behind its last feature (the next mates') is masked off --, where four slots x three arrays
// cost twelve loads and twelve 64-bit addresses per mate.  Reading four slots from the mate's first must stay inside the
// arrays: `nlf` / `nrf`, the arrays' lengths (left_off / right_off at n_pairs); a mate too close to the end is read slot
// by slot.
struct __attribute__((packed, aligned(4))) CollapseU32x4 {
   uint32_t v[4];
};
struct __attribute__((packed, aligned(1))) CollapseU32 {
   uint32_t v;
};
static_assert(kShortFeat == 4, "the wide feature loads are written for four slots");
__device__ __forceinline__ void flat_short_mate(const uint8_t *code, const uint32_t *left, const uint32_t *right, int64_t o, int n, bool held,
                                                int64_t total, uint32_t *l, uint32_t *r, uint32_t &c)
{
   c = 0u;
   if (held && n > 0 && o + kShortFeat <= total) {
      const CollapseU32x4 vl = *reinterpret_cast<const CollapseU32x4 *>(left + o), vr = *reinterpret_cast<const CollapseU32x4 *>(right + o);
      const uint32_t vc = reinterpret_cast<const CollapseU32 *>(code + o)->v;
#pragma unroll
      for (int i = 0; i < kShortFeat; ++i) l[i] = i < n ? vl.v[i] : 0u, r[i] = i < n ? vr.v[i] : 0u;
      c = n >= 4 ? vc : (vc & ((1u << (8 * n)) - 1u));
   } else {
#pragma unroll
      for (int i = 0; i < kShortFeat; ++i) {
         const bool in = held && i < n;
         l[i] = in ? left[o + i] : 0u;
         r[i] = in ? right[o + i] : 0u;
         c |= in ? (uint32_t)code[o + i] << (8 * i) : 0u;
      }
   }
}
__device__ __forceinline__ ShortPair flat_short_pair(const CollapseArgs &a, const PairOffsets &o, bool live, int64_t nlf, int64_t nrf)
{
   ShortPair q;
   const bool held = live && o.nl <= kShortFeat && o.nr <= kShortFeat;
   q.nl = held ? o.nl : -1, q.nr = held ? o.nr : -1;
   flat_short_mate(a.left_code, a.left_left, a.left_right, o.lo, o.nl, held, nlf, q.l, q.r, q.c[0]);
   flat_short_mate(a.right_code, a.right_left, a.right_right, o.ro, o.nr, held, nrf, q.l + kShortFeat, q.r + kShortFeat, q.c[1]);
   return q;
}
__device__ __forceinline__ bool short_pair_equal(const ShortPair &x, const ShortPair &y)
{
   bool e = x.nl == y.nl && x.nr == y.nr && x.c[0] == y.c[0] && x.c[1] == y.c[1];
#pragma unroll
   for (int i = 0; i < 2 * kShortFeat; ++i) e = e && x.l[i] == y.l[i] && x.r[i] == y.r[i];
   return e;
}
// the mates lie apart (or there is one mate) and each mate's features ascend: the hit's features are the left mate's, the
// GAP, the right mate's, as they are (flat_hit_features' first case, asked of the registers)
__device__ __forceinline__ bool short_pair_plain(const ShortPair &q)
{
   const bool both = q.nl > 0 && q.nr > 0;
   bool plain = true;
#pragma unroll
   for (int i = 1; i < kShortFeat; ++i) {
      plain = plain && (i >= q.nl || q.l[i] > q.l[i - 1]);
      plain = plain && (i >= q.nr || q.l[kShortFeat + i] > q.l[kShortFeat + i - 1]);
   }
   uint32_t last_r = 0;
#pragma unroll
   for (int i = 0; i < kShortFeat; ++i) last_r = i == q.nl - 1 ? q.r[i] : last_r;
   return plain && (!both || (int64_t)q.l[kShortFeat] - (int64_t)last_r - 1 > 0);
}

// ---- unique hits: a kept pair that differs from the previous kept pair of its cluster (:685-697)
__global__ __launch_bounds__(256) void flat_heads_kernel(FlatCollapseArgs f)
{
   __shared__ uint2 s_lr[kShortSlots * 256];
   __shared__ uint8_t s_c[kShortSlots * 256];
   const CollapseArgs &a = f.a;
   const int64_t s = xcd_tile() * 256 + threadIdx.x; // (XCD-aware tile order: device_common.h)
   int filt = 0, rej = 0;
   if (s < f.n_pairs) {
      int32_t nf = 0;
      uint8_t hd = 0;
      // The loads go out level by level, each level's together (round 5: as written before -- a position's pair, its
      // offsets, its features one by one inside the comparison's loop, then the same for the previous kept pair -- a wave
      // made ten DEPENDENT round trips and took 22 us for 14 GB of traffic in all):
      //   1  skip, cluster, pair, the previous kept position      2  the cluster's first position, that position's pair,
      //   this pair's offsets      3  this pair's features, the previous pair's offsets      4  the previous pair's features
      const int64_t nlf = a.left_off[f.n_pairs], nrf = a.right_off[f.n_pairs]; // (uniform: the feature arrays' lengths)
      const bool sk = f.skip[s] != 0;
      const int32_t l = f.cluster_of[s];
      const int64_t p = f.order[s];
      const int64_t prev = s > 0 ? (int64_t)f.last_kept[s - 1] : -1;
      const int64_t q0 = a.locus_pair_off[l];
      const bool has_prev = !sk && prev >= 0; // (prev >= q0 is asked below: q0 arrives with this level)
      const int64_t pp = has_prev ? (int64_t)f.order[prev] : 0;
      const PairOffsets op = flat_pair_offsets(a, p, !sk);
      const bool cmp = has_prev && prev >= q0;
      const PairOffsets opp = flat_pair_offsets(a, pp, cmp);
      const ShortPair me = flat_short_pair(a, op, !sk, nlf, nrf);
      const ShortPair pr = flat_short_pair(a, opp, cmp, nlf, nrf);
      if (sk) {
         filt = 1;
      } else {
         const MateRef x = MateRef{a.left_code + op.lo, a.left_left + op.lo, a.left_right + op.lo, op.nl};
         const MateRef y = MateRef{a.right_code + op.ro, a.right_left + op.ro, a.right_right + op.ro, op.nr};
         bool same = false;
         if (cmp) {
            if (me.nl >= 0 && pr.nl >= 0) {
               same = short_pair_equal(me, pr);
            } else if (op.nl == opp.nl && op.nr == opp.nr) { // (a mate of more than four features: the arrays)
               same = mate_equal(MateRef{a.left_code + opp.lo, a.left_left + opp.lo, a.left_right + opp.lo, opp.nl}, x) &&
                      mate_equal(MateRef{a.right_code + opp.ro, a.right_left + opp.ro, a.right_right + opp.ro, opp.nr}, y);
            }
         }
         if (!same) {
            hd = 1;
            if (x.n > kMateFeatMax || y.n > kMateFeatMax) { // a long mate (long reads): flat_heads_long_kernel counts its features
               nf = -1;
               f.counts[2] = 1; // (any writer, same value)
            } else {
               if (me.nl >= 0 && short_pair_plain(me)) nf = me.nl + me.nr + ((me.nl > 0 && me.nr > 0) ? 1 : 0);
               else nf = flat_hit_features(x, y, s_lr, s_c, nullptr, nullptr, nullptr);
               if (nf <= 0) nf = 0, rej = 1; // Contig(PairedHit) rejects the pair: no hit, its mass stays in the cluster's
            }
         }
      }
      f.head[s] = hd;
      f.nfeat[s] = nf;
      f.is_hit[s] = nf > 0 ? 1 : 0;
   }
   const unsigned long long bf = __ballot(filt), br = __ballot(rej);
   if ((threadIdx.x & 63) == 0) {
      if (bf) atomicAdd(&f.counts[0], (unsigned long long)__popcll(bf));
      if (br) atomicAdd(&f.counts[1], (unsigned long long)__popcll(br));
   }
}

// ---- pairs with a mate of more than kMateFeatMax features (long reads: dozens of exons): the same two steps -- the
// features' count for the heads, the features themselves for the fill -- with the merged list in private memory
// (hit_features_cap<kMateFeatLong>: 12 KB per lane).  Launched with a small grid after the main kernels; it returns at once
// when they met no such pair (counts[2]).
__global__ __launch_bounds__(64) void flat_heads_long_kernel(FlatCollapseArgs f)
{
   if (!f.counts[2]) return;
   const CollapseArgs &a = f.a;
   for (int64_t s = (int64_t)blockIdx.x * 64 + threadIdx.x; s < f.n_pairs; s += (int64_t)gridDim.x * 64) {
      if (f.nfeat[s] != -1) continue;
      const int64_t p = f.order[s];
      int nf = hit_features_cap<kMateFeatLong>(left_mate(a, p), right_mate(a, p), nullptr, nullptr, nullptr);
      if (nf <= 0) {
         nf = 0;
         atomicAdd(&f.counts[1], 1ull);
      }
      f.nfeat[s] = nf;
      f.is_hit[s] = nf > 0 ? 1 : 0;
   }
}
__global__ __launch_bounds__(64) void flat_fill_long_kernel(FlatCollapseArgs f)
{
   if (!f.counts[2]) return;
   const CollapseArgs &a = f.a;
   for (int64_t s = (int64_t)blockIdx.x * 64 + threadIdx.x; s < f.n_pairs; s += (int64_t)gridDim.x * 64) {
      if (f.nfeat[s] <= 0) continue;
      const int64_t p = f.order[s];
      const MateRef x = left_mate(a, p), y = right_mate(a, p);
      if (x.n <= kMateFeatMax && y.n <= kMateFeatMax) continue; // flat_fill_kernel's
      const int64_t fb = f.feat_base[s];
      hit_features_cap<kMateFeatLong>(x, y, a.feat_code + fb, a.feat_left + fb, a.feat_right + fb);
   }
}

// ---- masses: a wave per cluster; the cluster's mass and each group's as running doubles over the KEPT pairs in sorted
// order (:683-684, :688-696) -- same pattern as the sd: the lanes fetch their element's mass side by side (a skipped pair
// brings 0.0: adding it changes nothing), the additions go round as wave-uniform scalars, the next 64 elements are
// fetched meanwhile.  Also the cluster's first unique hit (the scan read at its first position).
struct FlatMassElem {
   double pm;
   int code, gi; // code: 0 skipped, 1 kept, 3 kept and the head of a group
};
__device__ __forceinline__ FlatMassElem flat_mass_load(const FlatCollapseArgs &f, int64_t s, int64_t q1)
{
   FlatMassElem e = {0.0, 0, 0};
   if (s < q1 && !f.skip[s]) {
      e.code = f.head[s] ? 3 : 1;
      e.pm = f.a.pair_mass[f.order[s]];
      e.gi = f.gid[s];
   }
   return e;
}
__global__ __launch_bounds__(64) void flat_mass_kernel(FlatCollapseArgs f)
{
   const CollapseArgs &a = f.a;
   const int lane = threadIdx.x;
   for (int64_t l = blockIdx.x; l < a.n_loci; l += gridDim.x) {
      if (!(f.cl_flags[l] & kNeedSeqMass)) continue; // (the usual cluster: flat_mass_any_order_kernel)
      const int64_t q0 = a.locus_pair_off[l], q1 = a.locus_pair_off[l + 1];
      double m = 0.0, g = 0.0;
      int gcur = 0; // the group being summed (0: none yet)
      FlatMassElem nxt = flat_mass_load(f, q0 + lane, q1);
      for (int64_t c0 = q0; c0 < q1; c0 += 64) {
         const int n = (int)min((int64_t)64, q1 - c0);
         const FlatMassElem e = nxt;
         nxt = flat_mass_load(f, c0 + 64 + lane, q1);
         // lanes whose element starts a group, as a mask: the additions run from head to head without looking at the codes
         unsigned long long heads = __ballot(e.code == 3);
         int k = 0;
         while (k < n) {
            const int stop = heads ? min(n, (int)__ffsll((long long)heads) - 1) : n; // the next head at or behind k (heads below k are cleared)
            for (; k < stop; ++k) {
               const double v = flat_bcast(e.pm, k);
               g += v;
               m += v;
            }
            if (k < n) { // k is a head: the group before it is complete
               if (gcur && lane == 0) f.gmass[gcur] = g;
               gcur = __builtin_amdgcn_readlane(e.gi, k);
               const double v = flat_bcast(e.pm, k);
               g = v;
               m += v;
               heads &= heads - 1;
               ++k;
            }
         }
      }
      if (lane == 0) {
         if (gcur) f.gmass[gcur] = g;
         a.cluster_mass[l] = m;
      }
   }
}

// sum of v over the lanes that share `key` (keys do not decrease along the wave); true in the LAST lane of each run, which
// then holds the run's sum
__device__ __forceinline__ bool flat_run_sum(int key, double &v)
{
   const int lane = threadIdx.x & 63;
   for (int o = 1; o < 64; o <<= 1) {
      const double w = __shfl_up(v, o);
      const int k = __shfl_up(key, o);
      if (lane >= o && k == key) v += w;
   }
   const int kn = __shfl_down(key, 1);
   return lane == 63 || kn != key;
}

// ---- masses where the order cannot matter (every mass of the cluster a multiple of 2^-20: see the head of the file):
// a thread per sorted position; runs of one cluster / one group inside a wave add up in registers, one atomic per run.
// Also every cluster's first unique hit (the scan read at its first position).
__global__ __launch_bounds__(256) void flat_mass_any_order_kernel(FlatCollapseArgs f)
{
   const CollapseArgs &a = f.a;
   const int64_t s = xcd_tile() * 256 + threadIdx.x; // (XCD-aware tile order: device_common.h)
   if (s <= a.n_loci) f.locus_hit_off[s] = f.hit_rank[s < a.n_loci ? a.locus_pair_off[s] : f.n_pairs];
   int l = -1, gk = 0x7fffffff; // (beyond the last position: keys that keep the runs apart)
   double v = 0.0;
   if (s < f.n_pairs) {
      l = f.cluster_of[s];
      gk = f.gid[s]; // the group of the last head at or before s (a skipped position adds 0.0 to it: nothing)
      if (f.cl_flags[l] & kNeedSeqMass) l = -1; // flat_mass_kernel's
      else if (!f.skip[s]) v = a.pair_mass[f.order[s]];
   }
   double vc = v, vg = v;
   const bool last_c = flat_run_sum(l, vc), last_g = flat_run_sum(gk, vg);
   if (l >= 0 && last_c && vc != 0.0) unsafeAtomicAdd(&a.cluster_mass[l], vc);
   if (last_g && vg != 0.0) unsafeAtomicAdd(&f.gmass[gk], vg);
}

// ---- fill: a thread per sorted position that starts a unique hit
__global__ __launch_bounds__(256) void flat_fill_kernel(FlatCollapseArgs f)
{
   __shared__ uint2 s_lr[kShortSlots * 256];
   __shared__ uint8_t s_c[kShortSlots * 256];
   const CollapseArgs &a = f.a;
   const int64_t s = xcd_tile() * 256 + threadIdx.x; // (XCD-aware tile order: device_common.h)
   if (s >= f.n_pairs) return;
   // (levels as in flat_heads_kernel: 1 the position's own numbers; 2 the pair's offsets, the group's mass; 3 the features)
   const int n = f.nfeat[s];
   const int64_t h = f.hit_rank[s], fb = f.feat_base[s], p = f.order[s];
   const int32_t loc = f.cluster_of[s], g = f.gid[s];
   const bool live = n > 0;
   const PairOffsets op = flat_pair_offsets(a, p, live);
   const double gm = live ? f.gmass[g] : 0.0;
   const ShortPair me = flat_short_pair(a, op, live, a.left_off[f.n_pairs], a.right_off[f.n_pairs]);
   if (!live) return;
   a.hit_locus[h] = loc;
   a.feat_off[h] = fb;
   a.hit_mass[h] = (float)gm; // stored as float (Contig::mass())
   if (me.nl >= 0 && short_pair_plain(me)) { // the left mate's features, the GAP, the right mate's: from the registers
      const bool both = me.nl > 0 && me.nr > 0;
      uint32_t last_r = 0;
#pragma unroll
      for (int i = 0; i < kShortFeat; ++i) {
         if (i < me.nl) {
            a.feat_code[fb + i] = (uint8_t)(me.c[0] >> (8 * i));
            a.feat_left[fb + i] = me.l[i];
            a.feat_right[fb + i] = me.r[i];
            last_r = me.r[i];
         }
      }
      int64_t at = fb + me.nl;
      if (both) {
         a.feat_code[at] = 2;
         a.feat_left[at] = last_r + 1u;
         a.feat_right[at] = me.l[kShortFeat] - 1u;
         ++at;
      }
#pragma unroll
      for (int i = 0; i < kShortFeat; ++i) {
         if (i < me.nr) {
            a.feat_code[at + i] = (uint8_t)(me.c[1] >> (8 * i));
            a.feat_left[at + i] = me.l[kShortFeat + i];
            a.feat_right[at + i] = me.r[kShortFeat + i];
         }
      }
      return;
   }
   const MateRef x = MateRef{a.left_code + op.lo, a.left_left + op.lo, a.left_right + op.lo, op.nl};
   const MateRef y = MateRef{a.right_code + op.ro, a.right_left + op.ro, a.right_right + op.ro, op.nr};
   if (x.n <= kMateFeatMax && y.n <= kMateFeatMax) flat_hit_features(x, y, s_lr, s_c, a.feat_code + fb, a.feat_left + fb, a.feat_right + fb);
   // (else: flat_fill_long_kernel writes the features)
}

} // namespace sb
