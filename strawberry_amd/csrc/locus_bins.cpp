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
   std::vector<int32_t> count;
   std::vector<uint32_t> bin_key, bin_compat;
   std::vector<int32_t> iso_len;
   std::vector<int64_t> hit_bin; // global bin of every hit, -1 when it has no compatible isoform
   std::vector<int64_t> pair_seg_off, pair_out_index;
   std::vector<uint32_t> pair_seg_lens, pair_mask;
   std::vector<int32_t> pair_iso_len;
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

int sbgpu_bins_create(const sbgpu_annotation_t *an, const sbgpu_hits_t *hits, const float *hit_mass,
                      int32_t compat_words, int32_t key_words, const uint32_t *compat, const uint32_t *key,
                      sbgpu_bins_t **out)
{
   if (!an || !hits || !out) return api_fail(SBGPU_EINVAL, "sbgpu_bins_create: null argument");
   *out = nullptr;
   const int64_t nl = an->n_loci, nh = hits->n_hits;
   if (nl < 0 || nh < 0 || compat_words < 0 || key_words < 0) return api_fail(SBGPU_EINVAL, "sbgpu_bins_create: bad counts");
   if (nl && (!an->iso_off || !an->exon_off || !an->seg_off)) return api_fail(SBGPU_EINVAL, "sbgpu_bins_create: null annotation");
   if (nh && (!hits->hit_locus || !hits->feat_off || !hit_mass || !compat || !key))
      return api_fail(SBGPU_EINVAL, "sbgpu_bins_create: null hits");
   sbgpu_bins *B = new (std::nothrow) sbgpu_bins();
   if (!B) return api_fail(SBGPU_ENOMEM, "sbgpu_bins_create: out of memory");
   auto bail = [&](int code, const char *msg) {
      delete B;
      return api_fail(code, msg);
   };
   try {
      B->n_loci = nl;
      B->key_words = key_words;
      B->compat_words = compat_words;
      const int64_t n_iso = nl ? an->iso_off[nl] : 0;
      B->n_iso = n_iso;
      B->iso_off.assign(an->iso_off, an->iso_off + nl + 1);
      if (nl == 0) B->iso_off.assign(1, 0);
      B->iso_len.resize((size_t)n_iso);
      for (int64_t i = 0; i < n_iso; ++i) { // Contig::exonic_length, src/contig.cpp:436-445
         int64_t len = 0;
         for (int64_t e = an->exon_off[i]; e < an->exon_off[i + 1]; ++e) len += (int64_t)an->exon_right[e] - an->exon_left[e] + 1;
         B->iso_len[(size_t)i] = (int32_t)len;
      }
      // hits of each locus, in input order
      std::vector<int64_t> loc_start((size_t)nl + 1, 0), order((size_t)nh);
      for (int64_t h = 0; h < nh; ++h) {
         if (hits->hit_locus[h] < 0 || hits->hit_locus[h] >= nl) return bail(SBGPU_EINVAL, "sbgpu_bins_create: hit_locus out of range");
         ++loc_start[(size_t)hits->hit_locus[h] + 1];
      }
      for (int64_t l = 0; l < nl; ++l) loc_start[(size_t)l + 1] += loc_start[(size_t)l];
      {
         std::vector<int64_t> fill(loc_start.begin(), loc_start.end() - 1);
         for (int64_t h = 0; h < nh; ++h) order[(size_t)fill[(size_t)hits->hit_locus[h]]++] = h;
      }
      B->hit_bin.assign((size_t)nh, -1);
      B->row_off.assign(1, 0);
      B->f_off.assign(1, 0);
      B->pair_seg_off.assign(1, 0);

      typedef std::vector<uint32_t> Key;
      typedef std::vector<uint64_t> Frag; // (offset << 32 | length) per feature: Contig::operator<, contig.cpp:342-347
      struct Bin {
         std::map<Frag, float> frags; // ExonBin::_frags (a std::set<Contig>): first insertion wins
         Key compat;
      };
      std::vector<Iv> iso_segs, bin_segs;
      for (int64_t l = 0; l < nl; ++l) {
         const int64_t i0 = an->iso_off[l], niso = an->iso_off[l + 1] - i0;
         const int64_t s0 = an->seg_off[l], nseg = an->seg_off[l + 1] - s0;
         if (niso > 32 * (int64_t)compat_words || nseg > 32 * (int64_t)key_words)
            return bail(SBGPU_ESHAPE, "sbgpu_bins_create: word counts do not cover a locus");
         std::map<Key, int> index; // UniqPushAndReturnIdx: bins in order of first appearance
         std::vector<Bin> bins;
         std::vector<const Key *> bin_keys;
         for (int64_t q = loc_start[(size_t)l]; q < loc_start[(size_t)l + 1]; ++q) {
            const int64_t h = order[(size_t)q];
            const uint32_t *cw = compat + h * compat_words, *kw = key + h * key_words;
            bool any_c = false, any_k = false;
            for (int w = 0; w < compat_words; ++w) any_c |= cw[w] != 0;
            for (int w = 0; w < key_words; ++w) any_k |= kw[w] != 0;
            if (!any_c || !any_k) continue; // no compatible isoform / set_maps: coords.empty()
            Key k(kw, kw + key_words);
            auto ins = index.emplace(k, (int)bins.size());
            if (ins.second) {
               bins.emplace_back();
               bins.back().compat.assign((size_t)compat_words, 0);
               bin_keys.push_back(&ins.first->first);
            }
            Bin &b = bins[(size_t)ins.first->second];
            for (int w = 0; w < compat_words; ++w) b.compat[(size_t)w] |= cw[w];
            Frag fr;
            for (int64_t f = hits->feat_off[h]; f < hits->feat_off[h + 1]; ++f)
               fr.push_back(((uint64_t)hits->feat_left[f] << 32) | (uint64_t)(hits->feat_right[f] - hits->feat_left[f] + 1));
            b.frags.emplace(std::move(fr), hit_mass[h]);
            B->hit_bin[(size_t)h] = B->n_bins + ins.first->second;
            ++B->n_hits_used;
         }
         const int64_t nb = (int64_t)bins.size();
         for (int64_t b = 0; b < nb; ++b) {
            float sum = 0.0f; // ExonBin::read_count, isoform.h:285-296: float accumulation in set order
            for (const auto &kv : bins[(size_t)b].frags) sum += kv.second;
            B->count.push_back((int32_t)sum); // n[i] = bin.read_count(), estimate.cpp:288
            B->bin_key.insert(B->bin_key.end(), bin_keys[(size_t)b]->begin(), bin_keys[(size_t)b]->end());
            B->bin_compat.insert(B->bin_compat.end(), bins[(size_t)b].compat.begin(), bins[(size_t)b].compat.end());
         }
         // (bin, isoform) pairs of set_theory_bin_weight with ExonBin::bin_under_iso
         const int64_t f0 = B->f_off.back();
         for (int64_t j = 0; j < niso; ++j) {
            const int64_t iso = i0 + j, e0 = an->exon_off[iso], ne = an->exon_off[iso + 1] - e0;
            // Isoform::_exon_segs (isoform.h:59-71): the locus' segments inside one of its exons
            iso_segs.clear();
            for (int64_t s = 0; s < nseg; ++s) {
               const uint32_t sl = an->seg_left[s0 + s], sr = an->seg_right[s0 + s];
               int64_t e = 0;
               while (e < ne && an->exon_right[e0 + e] < sl) ++e; // contig.cpp:615-634
               if (e < ne && an->exon_left[e0 + e] <= sl && an->exon_right[e0 + e] >= sr) iso_segs.push_back({sl, sr});
            }
            for (int64_t b = 0; b < nb; ++b) {
               if (!((bins[(size_t)b].compat[(size_t)(j >> 5)] >> (j & 31)) & 1u)) continue;
               bin_segs.clear();
               const Key &k = *bin_keys[(size_t)b];
               for (int64_t s = 0; s < nseg; ++s)
                  if ((k[(size_t)(s >> 5)] >> (s & 31)) & 1u) bin_segs.push_back({an->seg_left[s0 + s], an->seg_right[s0 + s]});
               // isoform.h:381-391: isoform segments from the bin's first to its last
               auto lb = [&](uint32_t v) {
                  size_t p = 0;
                  while (p < iso_segs.size() && iso_segs[p].l < v) ++p;
                  return p;
               };
               const size_t low = lb(bin_segs.front().l), up = lb(bin_segs.back().l);
               if (low >= iso_segs.size() || up >= iso_segs.size() || up < low)
                  return bail(SBGPU_ESHAPE, "sbgpu_bins_create: a bin is not under an isoform it is compatible with");
               const size_t n = up - low + 1;
               if (n > 32) return bail(SBGPU_ESHAPE, "sbgpu_bins_create: a bin spans more than 32 isoform segments");
               uint32_t mask = 0;
               size_t c = 1, i = 1; // :393-409
               while (i + 1 < n) {
                  if (c >= bin_segs.size() || iso_segs[low + i].l < bin_segs[c].l) {
                     mask |= 1u << i;
                     ++i;
                  } else if (iso_segs[low + i].l == bin_segs[c].l) {
                     ++i;
                     ++c;
                  } else {
                     return bail(SBGPU_ESHAPE, "sbgpu_bins_create: a bin holds a segment its isoform lacks");
                  }
               }
               for (size_t q = 0; q < n; ++q) B->pair_seg_lens.push_back(iso_segs[low + q].r - iso_segs[low + q].l + 1);
               B->pair_seg_off.push_back((int64_t)B->pair_seg_lens.size());
               B->pair_mask.push_back(mask);
               B->pair_iso_len.push_back(B->iso_len[(size_t)iso]);
               B->pair_out_index.push_back(f0 + b * niso + j);
            }
         }
         B->n_bins += nb;
         B->row_off.push_back(B->n_bins);
         B->f_off.push_back(f0 + nb * niso);
      }
      B->n_elem = B->f_off.back();
      B->n_pairs = (int64_t)B->pair_mask.size();
      B->n_pair_segs = (int64_t)B->pair_seg_lens.size();
   } catch (const std::bad_alloc &) {
      return bail(SBGPU_ENOMEM, "sbgpu_bins_create: out of memory");
   }
   *out = B;
   return SBGPU_OK;
}

void sbgpu_bins_destroy(sbgpu_bins_t *b) { delete b; }

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
#define SB_COPY(dst, vec)                                                              \
   if (dst && !(vec).empty()) std::memcpy(dst, (vec).data(), (vec).size() * sizeof((vec)[0]))
   SB_COPY(row_off, b->row_off);
   SB_COPY(iso_off, b->iso_off);
   SB_COPY(f_off, b->f_off);
   SB_COPY(count, b->count);
   SB_COPY(iso_len, b->iso_len);
   SB_COPY(bin_key, b->bin_key);
   SB_COPY(bin_compat, b->bin_compat);
   SB_COPY(hit_bin, b->hit_bin);
   SB_COPY(pair_seg_off, b->pair_seg_off);
   SB_COPY(pair_seg_lens, b->pair_seg_lens);
   SB_COPY(pair_implicit_mask, b->pair_mask);
   SB_COPY(pair_iso_len, b->pair_iso_len);
   SB_COPY(pair_out_index, b->pair_out_index);
#undef SB_COPY
   return SBGPU_OK;
}

} // extern "C"
