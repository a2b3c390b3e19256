// strawberry_amd/csrc/locus_bins.cpp -- host bookkeeping around the exon-bin kernel
// (SURVEY 8(a) A5): what the reference does per locus, single-threaded and map-based, between
// the interval tests (on the GPU here: exonbin_device.h) and the bin-weight / EM inputs.
//
//   sbgpu_segments_host   IRanges::disjoint over a locus' unique exons
//                         (/root/reference/include/estimate.hpp:80-91, include/interval.hpp:150-191)
//   sbgpu_hit_features    Contig::Contig(const PairedHit&), src/contig.cpp:216-267
//   sbgpu_bins_create     LocusContext::assign_exon_bin + set_maps (src/estimate.cpp:135-198,
//                         include/estimate.hpp:29-52), ExonBin::read_count (include/isoform.h:285-296),
//                         ExonBin::bin_under_iso (include/isoform.h:363-411) for every (bin, isoform)
//                         pair LocusContext::set_theory_bin_weight visits (src/estimate.cpp:201-213)
//
// Plain host C++: no HIP calls.  All arithmetic on the results happens in the kernels.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <map>
#include <new>
#include <string>
#include <vector>

#include "../../include/sbgpu.h"
#include "api_internal.h"

using sb::api_fail;

struct sbgpu_bins {
   int64_t n_loci = 0, n_iso = 0, n_bins = 0, n_elem = 0, n_pairs = 0, n_pair_segs = 0, n_hits_used = 0;
   int32_t key_words = 0, compat_words = 0;
   std::vector<int64_t> row_off, iso_off, f_off;
   sb::PodVec<int32_t> count;
   sb::PodVec<uint32_t> bin_key, bin_compat;
   std::vector<int32_t> iso_len;
   std::vector<int64_t> hit_bin; // global bin of every hit, -1 when it has no compatible isoform
   // ... or, from the device grouping of sbgpu_quantify_host, still in HBM (8 bytes per hit: it crosses PCIe when an export
   // asks for it, not before): an arena of its own (sb::dev_take), n_dev_hit_bin entries
   char *dev_hit_bin = nullptr;
   size_t dev_hit_bin_cap = 0;
   int64_t n_dev_hit_bin = 0;
   std::vector<int64_t> pair_seg_off, pair_out_index;
   std::vector<uint32_t> pair_seg_lens, pair_mask;
   std::vector<int32_t> pair_iso_len;
   std::vector<double> F; // the EM batch's weights, when the handle comes from sbgpu_quantify_host
   sb::DevicePairs dev;   // pairs made on the device: downloaded into the vectors above on first export
   bool pairs_on_device = false, pairs_downloaded = false;
   sb::DeviceBinArrays dev_bins; // device grouping: count / key / compat stay on the device until an export asks for them
   bool bins_downloaded = false;
   // how the hits were grouped (sbgpu_bins_grouping): on the device, or by the host code -- and then why the device form
   // was not used (empty: nobody asked it, e.g. sbgpu_bins_create called directly)
   bool grouped_on_device = false;
   std::string host_grouping_reason;
   ~sbgpu_bins()
   {
      sb::dev_give(dev.arena, dev.capacity);
      sb::dev_give(dev_bins.arena, dev_bins.capacity);
      sb::dev_give(dev_hit_bin, dev_hit_bin_cap);
   }
};

namespace {

struct Iv {
   uint32_t l, r;
};

// IRanges<GenomicFeature,false>::disjoint (interval.hpp:150-191) on sorted unique closed exons:
// cut at every left and every right+1 ("bars"); a piece between consecutive bars is kept when
// it is covered (the walk only opens a piece at a covered base, :173-183).
void disjoint(std::vector<Iv> ex, std::vector<Iv> *out)
{
   out->clear();
   if (ex.empty()) return;
   std::vector<uint64_t> bars;
   for (const Iv &e : ex) {
      bars.push_back(e.l);
      bars.push_back((uint64_t)e.r + 1);
   }
   std::sort(bars.begin(), bars.end());
   bars.erase(std::unique(bars.begin(), bars.end()), bars.end());
   std::sort(ex.begin(), ex.end(), [](const Iv &a, const Iv &b) { return a.l != b.l ? a.l < b.l : a.r < b.r; });
   // covered(p): some exon holds base p.  Bars ascend, so sweep the exons once.
   size_t k = 0;
   uint64_t reach = 0; // max right+1 over exons with left <= current bar
   for (size_t b = 0; b + 1 < bars.size(); ++b) {
      while (k < ex.size() && ex[k].l <= bars[b]) {
         reach = std::max<uint64_t>(reach, (uint64_t)ex[k].r + 1);
         ++k;
      }
      if (reach > bars[b]) out->push_back({(uint32_t)bars[b], (uint32_t)(bars[b + 1] - 1)});
   }
}

} // namespace

extern "C" {

int64_t sbgpu_segments_host(int64_t n_loci, const int64_t *iso_off, const int64_t *exon_off,
                            const uint32_t *exon_left, const uint32_t *exon_right, int64_t *seg_off,
                            uint32_t *seg_left, uint32_t *seg_right, int64_t cap)
{
   if (n_loci < 0 || (n_loci && (!iso_off || !exon_off))) return api_fail(SBGPU_EINVAL, "sbgpu_segments_host: null argument");
   int64_t total = 0;
   std::vector<Iv> ex, segs;
   if (seg_off) seg_off[0] = 0;
   for (int64_t l = 0; l < n_loci; ++l) {
      ex.clear();
      for (int64_t i = iso_off[l]; i < iso_off[l + 1]; ++i)
         for (int64_t e = exon_off[i]; e < exon_off[i + 1]; ++e) {
            if (exon_right[e] < exon_left[e]) return api_fail(SBGPU_EINVAL, "sbgpu_segments_host: exon with right < left");
            ex.push_back({exon_left[e], exon_right[e]});
         }
      disjoint(ex, &segs);
      for (const Iv &s : segs) {
         if (total < cap && seg_left && seg_right) {
            seg_left[total] = s.l;
            seg_right[total] = s.r;
         }
         ++total;
      }
      if (seg_off) seg_off[l + 1] = total;
   }
   return total;
}

int sbgpu_hit_features(int n_left, const uint8_t *lcode, const uint32_t *lleft, const uint32_t *lright,
                       int n_right, const uint8_t *rcode, const uint32_t *rleft, const uint32_t *rright,
                       uint8_t *code_out, uint32_t *left_out, uint32_t *right_out)
{
   if (n_left < 0 || n_right < 0 || (n_left && (!lcode || !lleft || !lright)) || (n_right && (!rcode || !rleft || !rright)) ||
       !code_out || !left_out || !right_out)
      return api_fail(SBGPU_EINVAL, "sbgpu_hit_features: bad argument");
   struct Ft {
      uint8_t c;
      uint32_t l, r;
   };
   // GenomicFeature::operator<, src/contig.cpp:186-193: by offset, then length
   auto less = [](const Ft &a, const Ft &b) { return a.l != b.l ? a.l < b.l : (a.r - a.l) < (b.r - b.l); };
   std::vector<Ft> g;
   if (n_left && n_right) {
      for (int i = 0; i < n_left; ++i) g.push_back({lcode[i], lleft[i], lright[i]});
      for (int i = 0; i < n_right; ++i) g.push_back({rcode[i], rleft[i], rright[i]});
      // ReadHit::left()/right() are the mate's alignment ends
      const int64_t gap = (int64_t)rleft[0] - (int64_t)lright[n_left - 1] - 1; // :234
      if (gap > 0) {
         g.push_back({2, lright[n_left - 1] + 1, (uint32_t)(lright[n_left - 1] + gap)});
      } else {
         std::sort(g.begin(), g.end(), less);
         // merge_genomicFeats, include/contig.h:111-137
         std::vector<Ft> res;
         for (size_t i = 0; i < g.size(); ++i) {
            res.push_back(g[i]);
            Ft &f = res.back();
            while (i + 1 < g.size() && f.c == g[i + 1].c) {
               if (f.c == 1) {
                  if (!(f.l == g[i + 1].l && f.r == g[i + 1].r)) return 0; // two different introns
               } else {
                  if (f.r < g[i + 1].l) return 0; // blocks that do not overlap (abutting included)
                  f.r = std::max(f.r, g[i + 1].r);
               }
               ++i;
            }
         }
         g.swap(res);
      }
   } else {
      for (int i = 0; i < n_right; ++i) g.push_back({rcode[i], rleft[i], rright[i]});
      for (int i = 0; i < n_left; ++i) g.push_back({lcode[i], lleft[i], lright[i]});
   }
   std::sort(g.begin(), g.end(), less); // :257
   for (size_t i = 0; i < g.size(); ++i) {
      code_out[i] = g[i].c;
      left_out[i] = g[i].l;
      right_out[i] = g[i].r;
   }
   return (int)g.size();
}

int64_t sbgpu_frag_lens_host(const sbgpu_annotation_t *an, const sbgpu_hits_t *hits, int32_t compat_words,
                             const uint32_t *compat, int32_t *frag_len_out)
{
   if (!an || !hits || (hits->n_hits && (!compat || !frag_len_out || !hits->hit_locus || !hits->feat_off)) || compat_words < 1)
      return api_fail(SBGPU_EINVAL, "sbgpu_frag_lens_host: bad argument");
   int64_t n = 0;
   for (int64_t h = 0; h < hits->n_hits; ++h) {
      frag_len_out[h] = -1;
      const int32_t loc = hits->hit_locus[h];
      if (loc < 0 || loc >= an->n_loci) return api_fail(SBGPU_EINVAL, "sbgpu_frag_lens_host: hit_locus out of range");
      const int64_t f0 = hits->feat_off[h], f1 = hits->feat_off[h + 1];
      if (f1 <= f0) continue;
      // compatible with exactly one transcript (alignments.cpp:1383-1392)
      int counter = 0;
      int64_t mark = 0;
      const int64_t niso = an->iso_off[loc + 1] - an->iso_off[loc];
      for (int64_t j = 0; j < niso && j < 32 * (int64_t)compat_words; ++j)
         if ((compat[h * compat_words + (j >> 5)] >> (j & 31)) & 1u) {
            ++counter;
            mark = j;
         }
      if (counter != 1) continue;
      // Contig::exonic_overlaps_len(transcript, hit.left(), hit.right()), src/contig.cpp:412-426
      const uint32_t left = hits->feat_left[f0], right = hits->feat_right[f1 - 1];
      const int64_t iso = an->iso_off[loc] + mark;
      int64_t len = 0;
      for (int64_t e = an->exon_off[iso]; e < an->exon_off[iso + 1]; ++e) {
         const uint32_t xl = an->exon_left[e], xr = an->exon_right[e];
         if (xl <= right && left <= xr) len += (int64_t)std::min(xr, right) - (int64_t)std::max(xl, left) + 1; // :118-125
      }
      frag_len_out[h] = (int32_t)len;
      ++n;
   }
   return n;
}

} // extern "C"

namespace {
// Bins that were already grouped (on the device, bins_device.h): per locus their number, and per bin
// the count, key words and compat union, in first-appearance order.  Only the pairs are left to do.
struct PreGrouped {
   const int64_t *row_off;
   const int32_t *count;
   const uint32_t *key, *compat;
   int64_t n_hits_used;
   const sb::DevicePairs *dev; // the pairs too were made on the device
   // with `dev`: count / key / compat are device pointers into an arena the handle takes over; the isoforms' lengths
   const sb::DeviceBinArrays *dev_bins;
   const std::vector<int32_t> *iso_len;
};

int bins_create_impl(const sbgpu_annotation_t *an, const sbgpu_hits_t *hits, const float *hit_mass,
                     int32_t compat_words, int32_t key_words, const uint32_t *compat, const uint32_t *key,
                     const PreGrouped *pre, sbgpu_bins_t **out)
{
   static const sbgpu_hits_t no_hits = {0, nullptr, nullptr, nullptr, nullptr, nullptr};
   if (pre) hits = &no_hits;
   if (!an || !hits || !out) return api_fail(SBGPU_EINVAL, "sbgpu_bins_create: null argument");
   *out = nullptr;
   const int64_t nl = an->n_loci, nh = hits->n_hits;
   if (nl < 0 || nh < 0 || compat_words < 0 || key_words < 0) return api_fail(SBGPU_EINVAL, "sbgpu_bins_create: bad counts");
   if (nl && (!an->iso_off || !an->exon_off || !an->seg_off)) return api_fail(SBGPU_EINVAL, "sbgpu_bins_create: null annotation");
   if (nh && (!hits->hit_locus || !hits->feat_off || !hit_mass || !compat || !key))
      return api_fail(SBGPU_EINVAL, "sbgpu_bins_create: null hits");
   for (int64_t h = 0; h < nh; ++h)
      if (hits->hit_locus[h] < 0 || hits->hit_locus[h] >= nl) return api_fail(SBGPU_EINVAL, "sbgpu_bins_create: hit_locus out of range");
   for (int64_t l = 0; l < nl; ++l)
      if (an->iso_off[l + 1] - an->iso_off[l] > 32 * (int64_t)compat_words || an->seg_off[l + 1] - an->seg_off[l] > 32 * (int64_t)key_words)
         return api_fail(SBGPU_ESHAPE, "sbgpu_bins_create: word counts do not cover a locus");
   sbgpu_bins *B = new (std::nothrow) sbgpu_bins();
   if (!B) return api_fail(SBGPU_ENOMEM, "sbgpu_bins_create: out of memory");
   B->grouped_on_device = pre != nullptr; // a handle made from the device grouping's arrays
   auto bail = [&](int code, const char *msg) {
      delete B;
      return api_fail(code, msg);
   };
   const bool timing = std::getenv("SBGPU_HOST_TIMING") != nullptr; // diagnostic: stage times on stderr
   auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
   double t_stage = now();
   auto stage = [&](const char *name) {
      if (timing) {
         const double t = now();
         std::fprintf(stderr, "sbgpu_bins_create: %-10s %.1f ms\n", name, (t - t_stage) * 1e3);
         t_stage = t;
      }
   };
   try {
      B->n_loci = nl;
      B->key_words = key_words;
      B->compat_words = compat_words;
      const int64_t n_iso = nl ? an->iso_off[nl] : 0;
      B->n_iso = n_iso;
      if (nl) B->iso_off.assign(an->iso_off, an->iso_off + nl + 1);
      else B->iso_off.assign(1, 0);
      if (pre && pre->iso_len && (int64_t)pre->iso_len->size() == n_iso) {
         B->iso_len = *pre->iso_len; // (sb::iso_segments made them already)
      } else {
         B->iso_len.resize((size_t)n_iso);
         for (int64_t i = 0; i < n_iso; ++i) { // Contig::exonic_length, src/contig.cpp:436-445
            int64_t len = 0;
            for (int64_t e = an->exon_off[i]; e < an->exon_off[i + 1]; ++e) len += (int64_t)an->exon_right[e] - an->exon_left[e] + 1;
            B->iso_len[(size_t)i] = (int32_t)len;
         }
      }
      if (pre && pre->dev) {
         // bins and pairs both come from the device: the handle only takes the arrays over
         B->row_off.assign(pre->row_off, pre->row_off + nl + 1);
         B->n_bins = B->row_off[(size_t)nl];
         B->f_off.assign((size_t)nl + 1, 0);
         for (int64_t l = 0; l < nl; ++l)
            B->f_off[(size_t)l + 1] = B->f_off[(size_t)l] + (B->row_off[(size_t)l + 1] - B->row_off[(size_t)l]) * (an->iso_off[l + 1] - an->iso_off[l]);
         B->n_elem = B->f_off[(size_t)nl];
         if (pre->dev_bins && pre->dev_bins->arena) {
            B->dev_bins = *pre->dev_bins;
         } else {
            B->count.assign(pre->count, pre->count + B->n_bins);
            B->bin_key.assign(pre->key, pre->key + B->n_bins * key_words);
            B->bin_compat.assign(pre->compat, pre->compat + B->n_bins * compat_words);
         }
         B->n_hits_used = pre->n_hits_used;
         B->dev = *pre->dev;
         B->pairs_on_device = true;
         B->n_pairs = pre->dev->n_pairs;
         B->n_pair_segs = pre->dev->n_pair_segs;
         stage("take over");
         *out = B;
         return SBGPU_OK;
      }
      // hits of each locus, in input order
      std::vector<int64_t> loc_start((size_t)nl + 1, 0), order((size_t)nh);
      for (int64_t h = 0; h < nh; ++h) ++loc_start[(size_t)hits->hit_locus[h] + 1];
      for (int64_t l = 0; l < nl; ++l) loc_start[(size_t)l + 1] += loc_start[(size_t)l];
      {
         std::vector<int64_t> fill(loc_start.begin(), loc_start.end() - 1);
         for (int64_t h = 0; h < nh; ++h) order[(size_t)fill[(size_t)hits->hit_locus[h]]++] = h;
      }
      B->hit_bin.assign((size_t)nh, -1);
      stage("setup");

      // ---- per locus, independent of every other locus: worker threads pull loci from a counter
      struct LocusOut {
         int64_t nb = 0, used = 0;
         std::vector<int32_t> count;
         std::vector<uint32_t> key, compat;          // nb * kw, nb * cw
         std::vector<uint32_t> pair_seg_lens, pair_mask;
         std::vector<int32_t> pair_nseg, pair_iso;    // per pair: segments, isoform (local index)
         std::vector<int64_t> pair_bin;               // per pair: local bin
         int err = 0;
         const char *msg = nullptr;
      };
      std::vector<LocusOut> res((size_t)nl);
      const uint32_t *fl = hits->feat_left, *fr = hits->feat_right;
      const int64_t *fo = hits->feat_off;
      // Contig::operator< (contig.cpp:342-347): features compared by (offset, length), lexicographic,
      // a proper prefix first; equal sequences are one std::set element and the first inserted stays
      auto frag_cmp = [&](int64_t x, int64_t y) {
         const int64_t nx = fo[x + 1] - fo[x], ny = fo[y + 1] - fo[y], n = nx < ny ? nx : ny;
         for (int64_t i = 0; i < n; ++i) {
            const uint32_t lx = fl[fo[x] + i], ly = fl[fo[y] + i];
            if (lx != ly) return lx < ly ? -1 : 1;
            const uint32_t wx = fr[fo[x] + i] - lx, wy = fr[fo[y] + i] - ly;
            if (wx != wy) return wx < wy ? -1 : 1;
         }
         return nx == ny ? 0 : (nx < ny ? -1 : 1);
      };
      auto do_locus = [&](int64_t l) {
         LocusOut &R = res[(size_t)l];
         const int64_t i0 = an->iso_off[l], niso = an->iso_off[l + 1] - i0;
         const int64_t s0 = an->seg_off[l], nseg = an->seg_off[l + 1] - s0;
         const int64_t q0 = loc_start[(size_t)l], nq = loc_start[(size_t)l + 1] - q0;
         const int cw = compat_words, kw = key_words;
         if (pre) { // grouped on the device: take the bins as they are
            const int64_t b0 = pre->row_off[l], pnb = pre->row_off[l + 1] - b0;
            R.nb = pnb;
            R.count.assign(pre->count + b0, pre->count + b0 + pnb);
            R.key.assign(pre->key + b0 * kw, pre->key + (b0 + pnb) * kw);
            R.compat.assign(pre->compat + b0 * cw, pre->compat + (b0 + pnb) * cw);
         } else {
         // bins keyed by the key words, numbered in order of first appearance (UniqPushAndReturnIdx):
         // open-addressing table of bin ids, a bin's key = the words of its first hit
         size_t cap = 16;
         while (cap < (size_t)nq * 2) cap <<= 1;
         std::vector<int32_t> table(cap, -1);
         std::vector<int64_t> first_hit;      // per bin
         std::vector<int32_t> local((size_t)nq, -1), n_in_bin;
         for (int64_t q = 0; q < nq; ++q) {
            const int64_t h = order[(size_t)(q0 + q)];
            const uint32_t *cwp = compat + h * cw, *kwp = key + h * kw;
            uint32_t any_c = 0, any_k = 0;
            uint64_t hash = 1469598103934665603ull;
            for (int w = 0; w < cw; ++w) any_c |= cwp[w];
            for (int w = 0; w < kw; ++w) {
               any_k |= kwp[w];
               hash = (hash ^ kwp[w]) * 1099511628211ull;
            }
            if (!any_c || !any_k) continue; // no compatible isoform / set_maps: coords.empty()
            size_t slot = (size_t)(hash ^ (hash >> 29)) & (cap - 1);
            int32_t b;
            for (;;) {
               b = table[slot];
               if (b < 0 || std::memcmp(key + first_hit[(size_t)b] * kw, kwp, (size_t)kw * 4) == 0) break;
               slot = (slot + 1) & (cap - 1);
            }
            if (b < 0) {
               b = (int32_t)first_hit.size();
               table[slot] = b;
               first_hit.push_back(h);
               n_in_bin.push_back(0);
               R.compat.insert(R.compat.end(), (size_t)cw, 0u);
            }
            for (int w = 0; w < cw; ++w) R.compat[(size_t)b * cw + w] |= cwp[w];
            local[(size_t)q] = b;
            ++n_in_bin[(size_t)b];
            ++R.used;
         }
         const int64_t nb = R.nb = (int64_t)first_hit.size();
         R.key.resize((size_t)nb * kw);
         for (int64_t b = 0; b < nb; ++b) std::memcpy(&R.key[(size_t)b * kw], key + first_hit[(size_t)b] * kw, (size_t)kw * 4);
         // members of each bin, then ExonBin::read_count (isoform.h:285-296): distinct fragments in the
         // std::set's order, float accumulation, truncated to int at estimate.cpp:288
         std::vector<int64_t> start((size_t)nb + 1, 0), members((size_t)R.used);
         for (int64_t b = 0; b < nb; ++b) start[(size_t)b + 1] = start[(size_t)b] + n_in_bin[(size_t)b];
         {
            std::vector<int64_t> fill(start.begin(), start.end() - 1);
            for (int64_t q = 0; q < nq; ++q)
               if (local[(size_t)q] >= 0) {
                  const int64_t h = order[(size_t)(q0 + q)];
                  members[(size_t)fill[(size_t)local[(size_t)q]]++] = h;
                  B->hit_bin[(size_t)h] = local[(size_t)q]; // local id for now; made global below
               }
         }
         R.count.resize((size_t)nb);
         for (int64_t b = 0; b < nb; ++b) {
            int64_t *m0 = members.data() + start[(size_t)b], *m1 = members.data() + start[(size_t)b + 1];
            // input order breaks ties, so of equal fragments the first inserted comes first
            std::stable_sort(m0, m1, [&](int64_t x, int64_t y) { return frag_cmp(x, y) < 0; });
            float sum = 0.0f;
            for (int64_t *m = m0; m < m1; ++m)
               if (m == m0 || frag_cmp(m[-1], m[0]) != 0) sum += hit_mass[*m];
            R.count[(size_t)b] = (int32_t)sum;
         }
         } // host grouping
         if (pre && pre->dev) return; // ... and so were the pairs
         const int64_t nb = R.nb;
         // (bin, isoform) pairs of set_theory_bin_weight with ExonBin::bin_under_iso.
         // Isoform::_exon_segs (isoform.h:59-71): the locus' segments inside one of its exons
         std::vector<std::vector<int32_t>> iso_segs((size_t)niso);
         for (int64_t j = 0; j < niso; ++j) {
            const int64_t iso = i0 + j, e0 = an->exon_off[iso], ne = an->exon_off[iso + 1] - e0;
            int64_t e = 0;
            for (int64_t sidx = 0; sidx < nseg; ++sidx) {
               const uint32_t sl = an->seg_left[s0 + sidx], sr = an->seg_right[s0 + sidx];
               while (e < ne && an->exon_right[e0 + e] < sl) ++e; // contig.cpp:615-634; segments ascend
               if (e < ne && an->exon_left[e0 + e] <= sl && an->exon_right[e0 + e] >= sr) iso_segs[(size_t)j].push_back((int32_t)sidx);
            }
         }
         std::vector<int32_t> bin_segs;
         for (int64_t j = 0; j < niso; ++j) {
            const std::vector<int32_t> &is = iso_segs[(size_t)j];
            for (int64_t b = 0; b < nb; ++b) {
               if (!((R.compat[(size_t)b * cw + (size_t)(j >> 5)] >> (j & 31)) & 1u)) continue;
               bin_segs.clear();
               const uint32_t *k = &R.key[(size_t)b * kw];
               for (int w = 0; w < kw; ++w)
                  for (uint32_t bits = k[w]; bits; bits &= bits - 1) bin_segs.push_back(32 * w + __builtin_ctz(bits));
               // isoform.h:381-391: the isoform's segments from the bin's first to its last.  Segment
               // indices ascend with their left ends, so lower_bound on start positions = on indices
               const size_t low = std::lower_bound(is.begin(), is.end(), bin_segs.front()) - is.begin();
               const size_t up = std::lower_bound(is.begin(), is.end(), bin_segs.back()) - is.begin();
               if (low >= is.size() || up >= is.size() || up < low) {
                  R.err = SBGPU_ESHAPE;
                  R.msg = "sbgpu_bins_create: a bin is not under an isoform it is compatible with";
                  return;
               }
               size_t n = up - low + 1;
               if (n > 32) {
                  // the bin-weight model's masks stop at 32 segments (as the reference's `1u << idx` do); the
                  // pair is still a pair -- the long-read workflow weighs it 1/L without looking at segments --
                  // so it is emitted without segments and sbgpu_binweight_* refuses it in the short-read model
                  R.pair_nseg.push_back(0);
                  R.pair_mask.push_back(0);
                  R.pair_iso.push_back((int32_t)j);
                  R.pair_bin.push_back(b);
                  continue;
               }
               uint32_t mask = 0;
               size_t c = 1, i = 1; // :393-409
               while (i + 1 < n) {
                  if (c >= bin_segs.size() || is[low + i] < bin_segs[c]) {
                     mask |= 1u << i;
                     ++i;
                  } else if (is[low + i] == bin_segs[c]) {
                     ++i;
                     ++c;
                  } else {
                     R.err = SBGPU_ESHAPE;
                     R.msg = "sbgpu_bins_create: a bin holds a segment its isoform lacks";
                     return;
                  }
               }
               for (size_t q = 0; q < n; ++q) {
                  const int32_t sidx = is[low + q];
                  R.pair_seg_lens.push_back(an->seg_right[s0 + sidx] - an->seg_left[s0 + sidx] + 1);
               }
               R.pair_nseg.push_back((int32_t)n);
               R.pair_mask.push_back(mask);
               R.pair_iso.push_back((int32_t)j);
               R.pair_bin.push_back(b);
            }
         }
      };
      unsigned nt = std::thread::hardware_concurrency();
      if (const char *e = std::getenv("SBGPU_HOST_THREADS")) nt = (unsigned)std::atoi(e);
      if (nt < 1) nt = 1;
      if (nt > 64) nt = 64;
      if ((int64_t)nt > nl) nt = (unsigned)(nl > 0 ? nl : 1);
      std::atomic<bool> oom(false);
      // run f(l) for every locus on nt threads, 16 loci at a time
      auto for_each_locus = [&](auto f) {
         std::atomic<int64_t> next(0);
         auto worker = [&]() {
            try {
               for (;;) {
                  const int64_t l0 = next.fetch_add(16);
                  if (l0 >= nl) break;
                  for (int64_t l = l0; l < nl && l < l0 + 16; ++l) f(l);
               }
            } catch (const std::bad_alloc &) {
               oom = true;
            }
         };
         if (nt == 1) {
            worker();
         } else {
            std::vector<std::thread> pool;
            for (unsigned t = 0; t < nt; ++t) pool.emplace_back(worker);
            for (auto &t : pool) t.join();
         }
      };
      for_each_locus(do_locus);
      if (oom) return bail(SBGPU_ENOMEM, "sbgpu_bins_create: out of memory");
      stage("loci");
      // ---- stitch the loci together, in locus order: offsets first, then every locus copies its part
      B->row_off.assign((size_t)nl + 1, 0);
      B->f_off.assign((size_t)nl + 1, 0);
      std::vector<int64_t> pair0((size_t)nl + 1, 0), pseg0((size_t)nl + 1, 0);
      for (int64_t l = 0; l < nl; ++l) {
         const LocusOut &R = res[(size_t)l];
         if (R.err) return bail(R.err, R.msg);
         const int64_t niso = an->iso_off[l + 1] - an->iso_off[l];
         B->row_off[(size_t)l + 1] = B->row_off[(size_t)l] + R.nb;
         B->f_off[(size_t)l + 1] = B->f_off[(size_t)l] + R.nb * niso;
         pair0[(size_t)l + 1] = pair0[(size_t)l] + (int64_t)R.pair_mask.size();
         pseg0[(size_t)l + 1] = pseg0[(size_t)l] + (int64_t)R.pair_seg_lens.size();
         B->n_hits_used += R.used;
      }
      B->n_bins = B->row_off[(size_t)nl];
      const int64_t n_pairs = pair0[(size_t)nl];
      B->count.resize((size_t)B->n_bins);
      B->bin_key.resize((size_t)B->n_bins * key_words);
      B->bin_compat.resize((size_t)B->n_bins * compat_words);
      B->pair_seg_lens.resize((size_t)pseg0[(size_t)nl]);
      B->pair_mask.resize((size_t)n_pairs);
      B->pair_iso_len.resize((size_t)n_pairs);
      B->pair_out_index.resize((size_t)n_pairs);
      B->pair_seg_off.assign((size_t)n_pairs + 1, 0);
      for_each_locus([&](int64_t l) {
         LocusOut &R = res[(size_t)l];
         const int64_t niso = an->iso_off[l + 1] - an->iso_off[l], f0 = B->f_off[(size_t)l], b0 = B->row_off[(size_t)l];
         std::copy(R.count.begin(), R.count.end(), B->count.begin() + b0);
         std::copy(R.key.begin(), R.key.end(), B->bin_key.begin() + b0 * key_words);
         std::copy(R.compat.begin(), R.compat.end(), B->bin_compat.begin() + b0 * compat_words);
         std::copy(R.pair_seg_lens.begin(), R.pair_seg_lens.end(), B->pair_seg_lens.begin() + pseg0[(size_t)l]);
         std::copy(R.pair_mask.begin(), R.pair_mask.end(), B->pair_mask.begin() + pair0[(size_t)l]);
         int64_t so = pseg0[(size_t)l];
         for (size_t p = 0; p < R.pair_mask.size(); ++p) {
            const size_t g = (size_t)pair0[(size_t)l] + p;
            so += R.pair_nseg[p];
            B->pair_seg_off[g + 1] = so;
            B->pair_iso_len[g] = B->iso_len[(size_t)(an->iso_off[l] + R.pair_iso[p])];
            B->pair_out_index[g] = f0 + R.pair_bin[p] * niso + R.pair_iso[p];
         }
         for (int64_t q = loc_start[(size_t)l]; q < loc_start[(size_t)l + 1]; ++q) {
            int64_t &hb = B->hit_bin[(size_t)order[(size_t)q]];
            if (hb >= 0) hb += b0;
         }
         R = LocusOut(); // release
      });
      if (oom) return bail(SBGPU_ENOMEM, "sbgpu_bins_create: out of memory");
      stage("stitch");
      B->n_elem = B->f_off.back();
      B->n_pairs = (int64_t)B->pair_mask.size();
      B->n_pair_segs = (int64_t)B->pair_seg_lens.size();
   } catch (const std::bad_alloc &) {
      return bail(SBGPU_ENOMEM, "sbgpu_bins_create: out of memory");
   }
   if (pre) B->n_hits_used = pre->n_hits_used;
   if (pre && pre->dev) {
      B->dev = *pre->dev;
      B->pairs_on_device = true;
      B->n_pairs = pre->dev->n_pairs;
      B->n_pair_segs = pre->dev->n_pair_segs;
   }
   *out = B;
   return SBGPU_OK;
}

} // namespace

namespace sb {
void bins_set_weights(sbgpu_bins_t *b, std::vector<double> &&F) { b->F = std::move(F); }
void bins_set_grouping(sbgpu_bins_t *b, bool on_device, const std::string &why_host)
{
   b->grouped_on_device = on_device;
   b->host_grouping_reason = why_host;
}
void bins_set_hit_bin(sbgpu_bins_t *b, std::vector<int64_t> &&hb) { b->hit_bin = std::move(hb); }
void bins_set_device_hit_bin(sbgpu_bins_t *b, char *arena, size_t capacity, int64_t n)
{
   b->dev_hit_bin = arena;
   b->dev_hit_bin_cap = capacity;
   b->n_dev_hit_bin = n;
}
const double *bins_weights_tail(const sbgpu_bins_t *b, size_t at) { return b->F.data() + at; }
const DevicePairs *bins_device_pairs(const sbgpu_bins_t *b) { return b && b->pairs_on_device ? &b->dev : nullptr; }
int bins_from_groups(const sbgpu_annotation_t *an, int32_t compat_words, int32_t key_words, const int64_t *row_off,
                     const DeviceBinArrays &arrays, int64_t n_hits_used, const DevicePairs *pairs, const std::vector<int32_t> *iso_len,
                     sbgpu_bins_t **out)
{
   const PreGrouped pre = {row_off, nullptr, nullptr, nullptr, n_hits_used, pairs, &arrays, iso_len};
   return bins_create_impl(an, nullptr, nullptr, compat_words, key_words, nullptr, nullptr, &pre, out);
}
} // namespace sb

extern "C" {

int sbgpu_bins_create(const sbgpu_annotation_t *an, const sbgpu_hits_t *hits, const float *hit_mass,
                      int32_t compat_words, int32_t key_words, const uint32_t *compat, const uint32_t *key,
                      sbgpu_bins_t **out)
{
   return bins_create_impl(an, hits, hit_mass, compat_words, key_words, compat, key, nullptr, out);
}

void sbgpu_bins_destroy(sbgpu_bins_t *b) { delete b; }

int sbgpu_bins_export_weights(const sbgpu_bins_t *b, double *F_out)
{
   if (!b || !F_out) return api_fail(SBGPU_EINVAL, "sbgpu_bins_export_weights: null argument");
   if ((int64_t)b->F.size() < b->n_elem || (b->F.empty() && b->n_elem == 0 && b->n_bins > 0))
      return api_fail(SBGPU_EINVAL, "sbgpu_bins_export_weights: this handle holds no weights");
   if (b->n_elem) std::memcpy(F_out, b->F.data(), (size_t)b->n_elem * sizeof(double));
   return SBGPU_OK;
}

int sbgpu_bins_grouping(const sbgpu_bins_t *b, int32_t *on_device, const char **why_host)
{
   if (!b || !on_device) return api_fail(SBGPU_EINVAL, "sbgpu_bins_grouping: null argument");
   *on_device = b->grouped_on_device ? 1 : 0;
   if (why_host) *why_host = b->host_grouping_reason.c_str();
   return SBGPU_OK;
}

int sbgpu_bins_info(const sbgpu_bins_t *b, int64_t info[8])
{
   if (!b || !info) return api_fail(SBGPU_EINVAL, "sbgpu_bins_info: null argument");
   info[0] = b->n_loci;
   info[1] = b->n_iso;
   info[2] = b->n_bins;
   info[3] = b->n_elem;
   info[4] = b->n_pairs;
   info[5] = b->n_pair_segs;
   info[6] = b->n_hits_used;
   info[7] = b->key_words;
   return SBGPU_OK;
}

int sbgpu_bins_export(const sbgpu_bins_t *b, int64_t *row_off, int64_t *iso_off, int64_t *f_off, int32_t *count,
                      int32_t *iso_len, uint32_t *bin_key, uint32_t *bin_compat, int64_t *hit_bin,
                      int64_t *pair_seg_off, uint32_t *pair_seg_lens, uint32_t *pair_implicit_mask,
                      int32_t *pair_iso_len, int64_t *pair_out_index)
{
   if (!b) return api_fail(SBGPU_EINVAL, "sbgpu_bins_export: null argument");
   if (b->pairs_on_device && !b->pairs_downloaded &&
       (pair_seg_off || pair_seg_lens || pair_implicit_mask || pair_iso_len || pair_out_index)) {
      // pairs made on the device: bring them over once, now that somebody wants them on the host
      sbgpu_bins *m = const_cast<sbgpu_bins *>(b);
      const sb::DevicePairs &d = b->dev;
      m->pair_seg_off.assign((size_t)d.n_pairs + 1, 0);
      m->pair_seg_lens.assign((size_t)d.n_pair_segs, 0);
      m->pair_mask.assign((size_t)d.n_pairs, 0);
      m->pair_iso_len.assign((size_t)d.n_pairs, 0);
      m->pair_out_index.assign((size_t)d.n_pairs, 0);
      hipError_t e = hipMemcpy(m->pair_seg_off.data(), d.seg_off(), ((size_t)d.n_pairs + 1) * 8, hipMemcpyDeviceToHost);
      if (e == hipSuccess && d.n_pair_segs) e = hipMemcpy(m->pair_seg_lens.data(), d.seg_lens(), (size_t)d.n_pair_segs * 4, hipMemcpyDeviceToHost);
      if (e == hipSuccess && d.n_pairs) e = hipMemcpy(m->pair_mask.data(), d.mask(), (size_t)d.n_pairs * 4, hipMemcpyDeviceToHost);
      if (e == hipSuccess && d.n_pairs) e = hipMemcpy(m->pair_iso_len.data(), d.iso_len(), (size_t)d.n_pairs * 4, hipMemcpyDeviceToHost);
      if (e == hipSuccess && d.n_pairs) e = hipMemcpy(m->pair_out_index.data(), d.out_index(), (size_t)d.n_pairs * 8, hipMemcpyDeviceToHost);
      if (e != hipSuccess) return api_fail(SBGPU_EHIP, std::string("sbgpu_bins_export: download of the pairs: ") + hipGetErrorString(e));
      m->pairs_downloaded = true;
   }
#define SB_COPY(dst, vec)                                                              \
   if (dst && !(vec).empty()) std::memcpy(dst, (vec).data(), (vec).size() * sizeof((vec)[0]))
   if (b->dev_bins.arena && !b->bins_downloaded && (count || bin_key || bin_compat)) {
      // the per-bin arrays of a device grouping: downloaded now, once
      sbgpu_bins *m = const_cast<sbgpu_bins *>(b);
      const sb::DeviceBinArrays &d = b->dev_bins;
      m->count.resize((size_t)b->n_bins);
      m->bin_key.resize((size_t)b->n_bins * b->key_words);
      m->bin_compat.resize((size_t)b->n_bins * b->compat_words);
      hipError_t e = hipSuccess;
      if (b->n_bins) e = hipMemcpy(m->count.data(), d.arena + d.o_count, (size_t)b->n_bins * 4, hipMemcpyDeviceToHost);
      if (e == hipSuccess && b->n_bins) e = hipMemcpy(m->bin_key.data(), d.arena + d.o_key, (size_t)b->n_bins * 4 * b->key_words, hipMemcpyDeviceToHost);
      if (e == hipSuccess && b->n_bins) e = hipMemcpy(m->bin_compat.data(), d.arena + d.o_compat, (size_t)b->n_bins * 4 * b->compat_words, hipMemcpyDeviceToHost);
      if (e != hipSuccess) return api_fail(SBGPU_EHIP, std::string("sbgpu_bins_export: download of the bins: ") + hipGetErrorString(e));
      m->bins_downloaded = true;
   }
   SB_COPY(count, b->count);
   SB_COPY(bin_key, b->bin_key);
   SB_COPY(bin_compat, b->bin_compat);
   SB_COPY(row_off, b->row_off);
   SB_COPY(iso_off, b->iso_off);
   SB_COPY(f_off, b->f_off);
   SB_COPY(iso_len, b->iso_len);
   SB_COPY(hit_bin, b->hit_bin);
   if (hit_bin && b->hit_bin.empty() && b->dev_hit_bin && b->n_dev_hit_bin) { // straight into the caller's array
      const hipError_t e = hipMemcpy(hit_bin, b->dev_hit_bin, (size_t)b->n_dev_hit_bin * 8, hipMemcpyDeviceToHost);
      if (e != hipSuccess) return api_fail(SBGPU_EHIP, std::string("sbgpu_bins_export: download of hit -> bin: ") + hipGetErrorString(e));
   }
   SB_COPY(pair_seg_off, b->pair_seg_off);
   SB_COPY(pair_seg_lens, b->pair_seg_lens);
   SB_COPY(pair_implicit_mask, b->pair_mask);
   SB_COPY(pair_iso_len, b->pair_iso_len);
   SB_COPY(pair_out_index, b->pair_out_index);
#undef SB_COPY
   return SBGPU_OK;
}

// ---------------------------------------------------------------- pairs -> unique hits
} // extern "C"

struct sbgpu_uniq {
   int64_t n_loci = 0, n_filtered = 0, n_rejected = 0, total_mapped = 0;
   std::vector<int32_t> hit_locus;
   std::vector<int64_t> feat_off{0};
   std::vector<uint8_t> feat_code;
   std::vector<uint32_t> feat_left, feat_right;
   std::vector<float> hit_mass;
   std::vector<double> cluster_mass;
};

namespace {
// include/common.h:112-134
double ref_phi(double x)
{
   const double a1 = 0.254829592, a2 = -0.284496736, a3 = 1.421413741, a4 = -1.453152027, a5 = 1.061405429, p = 0.3275911;
   int sign = 1;
   if (x < 0) sign = -1;
   x = std::fabs(x) / std::sqrt(2.0);
   const double t = 1.0 / (1.0 + p * x);
   const double y = 1.0 - (((((a5 * t + a4) * t) + a3) * t + a2) * t + a1) * t * std::exp(-x * x);
   return 0.5 * (1.0 + sign * y);
}
} // namespace

extern "C" {

int sbgpu_collapse_pairs_host(int64_t n_loci, const sbgpu_pairs_t *pr, sbgpu_uniq_t **out)
{
   if (!pr || !out || n_loci < 0) return api_fail(SBGPU_EINVAL, "sbgpu_collapse_pairs_host: bad argument");
   *out = nullptr;
   const int64_t np = pr->n_pairs;
   if (np < 0 || (np && (!pr->pair_locus || !pr->pair_mass || !pr->left_off || !pr->right_off)))
      return api_fail(SBGPU_EINVAL, "sbgpu_collapse_pairs_host: null array");
   sbgpu_uniq *U = new (std::nothrow) sbgpu_uniq();
   if (!U) return api_fail(SBGPU_ENOMEM, "sbgpu_collapse_pairs_host: out of memory");
   try {
      U->n_loci = n_loci;
      U->cluster_mass.assign((size_t)n_loci, 0.0);
      // pairs of each locus, input order
      std::vector<int64_t> start((size_t)n_loci + 1, 0), order((size_t)np);
      for (int64_t p = 0; p < np; ++p) {
         if (pr->pair_locus[p] < 0 || pr->pair_locus[p] >= n_loci) {
            delete U;
            return api_fail(SBGPU_EINVAL, "sbgpu_collapse_pairs_host: pair_locus out of range");
         }
         if (pr->left_off[p + 1] == pr->left_off[p] && pr->right_off[p + 1] == pr->right_off[p]) {
            delete U;
            return api_fail(SBGPU_EINVAL, "sbgpu_collapse_pairs_host: a pair without mates");
         }
         ++start[(size_t)pr->pair_locus[p] + 1];
      }
      for (int64_t l = 0; l < n_loci; ++l) start[(size_t)l + 1] += start[(size_t)l];
      {
         std::vector<int64_t> fill(start.begin(), start.end() - 1);
         for (int64_t p = 0; p < np; ++p) order[(size_t)fill[(size_t)pr->pair_locus[p]]++] = p;
      }
      struct Mate {
         const uint8_t *c;
         const uint32_t *l, *r;
         int64_t n;
      };
      auto left_mate = [&](int64_t p) { const int64_t o = pr->left_off[p]; return Mate{pr->left_code + o, pr->left_left + o, pr->left_right + o, pr->left_off[p + 1] - o}; };
      auto right_mate = [&](int64_t p) { const int64_t o = pr->right_off[p]; return Mate{pr->right_code + o, pr->right_left + o, pr->right_right + o, pr->right_off[p + 1] - o}; };
      auto mate_equal = [](const Mate &a, const Mate &b) { // ReadHit::operator==: same start, same CIGAR
         if (a.n != b.n) return false;
         for (int64_t i = 0; i < a.n; ++i)
            if (a.c[i] != b.c[i] || a.l[i] != b.l[i] || a.r[i] != b.r[i]) return false;
         return true;
      };
      auto left_pos = [&](int64_t p) { // PairedHit::left_pos / right_pos, src/read.cpp:797-819
         const Mate a = left_mate(p), b = right_mate(p);
         if (a.n && b.n) return std::min(a.l[0], b.l[0]);
         return a.n ? a.l[0] : b.l[0];
      };
      auto right_pos = [&](int64_t p) {
         const Mate a = left_mate(p), b = right_mate(p);
         if (a.n && b.n) return std::max(a.r[a.n - 1], b.r[b.n - 1]);
         return b.n ? b.r[b.n - 1] : a.r[a.n - 1];
      };
      std::vector<uint8_t> oc;
      std::vector<uint32_t> ol, orr;
      for (int64_t l = 0; l < n_loci; ++l) {
         int64_t *q0 = order.data() + start[(size_t)l], *q1 = order.data() + start[(size_t)l + 1];
         if (q0 == q1) continue;
         // spans of all mates of the cluster (HitCluster::_read_ref_span), getMeanAndSd (common.h:100-110)
         double sum = 0.0;
         int64_t n_mates = 0;
         for (int64_t *q = q0; q < q1; ++q)
            for (const Mate &m : {left_mate(*q), right_mate(*q)})
               if (m.n) {
                  sum += (double)(int)(m.r[m.n - 1] - m.l[0] + 1);
                  ++n_mates;
               }
         const double mean = sum / (double)n_mates;
         double sq = 0.0;
         for (int64_t *q = q0; q < q1; ++q)
            for (const Mate &m : {left_mate(*q), right_mate(*q)})
               if (m.n) {
                  const double d = (double)(int)(m.r[m.n - 1] - m.l[0] + 1) - mean;
                  sq += d * d;
               }
         const double sd = std::sqrt(sq / (double)n_mates) * 5;
         std::stable_sort(q0, q1, [&](int64_t x, int64_t y) {
            const uint32_t lx = left_pos(x), ly = left_pos(y);
            return lx != ly ? lx < ly : right_pos(x) < right_pos(y);
         });
         int64_t last = -1; // the pair the latest unique hit was made of
         double mass = 0.0;
         auto flush = [&]() {
            if (last < 0) return;
            const Mate a = left_mate(last), b = right_mate(last);
            oc.resize((size_t)(a.n + b.n + 1));
            ol.resize(oc.size());
            orr.resize(oc.size());
            const int n = sbgpu_hit_features((int)a.n, a.c, a.l, a.r, (int)b.n, b.c, b.l, b.r, oc.data(), ol.data(), orr.data());
            if (n <= 0) {
               ++U->n_rejected;
               return;
            }
            U->hit_locus.push_back((int32_t)l);
            U->feat_code.insert(U->feat_code.end(), oc.begin(), oc.begin() + n);
            U->feat_left.insert(U->feat_left.end(), ol.begin(), ol.begin() + n);
            U->feat_right.insert(U->feat_right.end(), orr.begin(), orr.begin() + n);
            U->feat_off.push_back((int64_t)U->feat_code.size());
            U->hit_mass.push_back((float)mass); // Contig::mass() returns float
         };
         for (int64_t *q = q0; q < q1; ++q) {
            const Mate a = left_mate(*q), b = right_mate(*q);
            bool skip = false;
            for (const Mate &m : {a, b})
               if (m.n && ref_phi(((double)(uint32_t)(m.r[m.n - 1] - m.l[0] + 1) - mean) / sd) > 0.999) skip = true;
            if (skip) {
               ++U->n_filtered;
               continue;
            }
            U->cluster_mass[(size_t)l] += pr->pair_mass[*q];
            const bool same = last >= 0 && mate_equal(left_mate(last), a) && mate_equal(right_mate(last), b);
            if (same) {
               mass += pr->pair_mass[*q];
            } else {
               flush();
               last = *q;
               mass = pr->pair_mass[*q];
            }
         }
         flush();
         U->total_mapped += (int64_t)(int)U->cluster_mass[(size_t)l];
      }
   } catch (const std::bad_alloc &) {
      delete U;
      return api_fail(SBGPU_ENOMEM, "sbgpu_collapse_pairs_host: out of memory");
   }
   *out = U;
   return SBGPU_OK;
}

void sbgpu_uniq_destroy(sbgpu_uniq_t *u) { delete u; }

int sbgpu_uniq_info(const sbgpu_uniq_t *u, int64_t info[8])
{
   if (!u || !info) return api_fail(SBGPU_EINVAL, "sbgpu_uniq_info: null argument");
   info[0] = (int64_t)u->hit_locus.size();
   info[1] = (int64_t)u->feat_code.size();
   info[2] = u->n_filtered;
   info[3] = u->n_rejected;
   info[4] = u->total_mapped;
   info[5] = u->n_loci;
   info[6] = info[7] = 0;
   return SBGPU_OK;
}

int sbgpu_uniq_export(const sbgpu_uniq_t *u, int32_t *hit_locus, int64_t *feat_off, uint8_t *feat_code, uint32_t *feat_left,
                      uint32_t *feat_right, float *hit_mass, double *cluster_mass)
{
   if (!u) return api_fail(SBGPU_EINVAL, "sbgpu_uniq_export: null argument");
#define SB_COPY(dst, vec)                                                              \
   if (dst && !(vec).empty()) std::memcpy(dst, (vec).data(), (vec).size() * sizeof((vec)[0]))
   SB_COPY(hit_locus, u->hit_locus);
   SB_COPY(feat_off, u->feat_off);
   SB_COPY(feat_code, u->feat_code);
   SB_COPY(feat_left, u->feat_left);
   SB_COPY(feat_right, u->feat_right);
   SB_COPY(hit_mass, u->hit_mass);
   SB_COPY(cluster_mass, u->cluster_mass);
#undef SB_COPY
   return SBGPU_OK;
}

} // extern "C"
