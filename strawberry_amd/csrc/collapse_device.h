// strawberry_amd/csrc/collapse_device.h -- what the duplicate-collapse kernels share (HitCluster::collapseAndFilterHits on the
// GPU; /root/reference/src/alignments.cpp:656-703, src/contig.cpp:216-267): the pairs' arrays, a mate as a
// feature list, "two mates are equal", the reference's phi(), and Contig(PairedHit)'s merged feature list.  The collapse
// itself: collapse_flat.h (all clusters of a call at once; round 3's one-workgroup-per-cluster kernels, which nothing had
// reached since round 4, are gone).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>


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

} // namespace sb
