// strawberry_amd/csrc/exonbin_api.hip -- sbgpu_exonbin_device / sbgpu_exonbin_host
// (include/sbgpu.h): launch of the per-hit compatibility + bin-key kernel.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <thread>
#include <vector>

#include "../../include/sbgpu.h"
#include "api_internal.h"
#include "bins_device.h"
#include "exonbin_device.h"

using sb::api_fail;

namespace sb {
// sbgpu_exonbin_device that also leaves every hit's span and sequence hash (exonbin_device.h) when asked to
static size_t up256b(size_t b) { return (b + 255) & ~(size_t)255; }
size_t seg_basis_bytes(int64_t n_loci, int64_t n_iso)
{
   const size_t ni1 = (size_t)(n_iso > 0 ? n_iso : 1);
   return 4 * up256b(ni1 * 8) + 2 * up256b((size_t)n_loci * 8) + up256b((size_t)n_loci * 4);
}
int make_seg_basis(sbgpu_ctx_t *c, const sbgpu_annotation_t *an, int64_t n_iso, char *dm, void *stream, DeviceSegBasis *out)
{
   const size_t ni1 = (size_t)(n_iso > 0 ? n_iso : 1);
   const size_t isz = up256b(ni1 * 8), lsz = up256b((size_t)an->n_loci * 8);
   const size_t o_mem = 0, o_sta = isz, o_memh = 2 * isz, o_stah = 3 * isz, o_adj = 4 * isz, o_adjh = o_adj + lsz, o_ok = o_adjh + lsz;
   sb::ExonBinArgs a = {};
   a.iso_off = an->iso_off;
   a.exon_off = an->exon_off;
   a.exon_left = an->exon_left;
   a.exon_right = an->exon_right;
   a.seg_off = an->seg_off;
   a.seg_left = an->seg_left;
   a.seg_right = an->seg_right;
   const int64_t cap = (int64_t)sb::ctx_cu_count(c) * 8, lb = (an->n_loci + 255) / 256, ib = ((int64_t)ni1 + 255) / 256;
   hipLaunchKernelGGL(sb::iso_masks_locus_kernel, dim3((unsigned)(lb < cap ? lb : cap)), dim3(256), 0, (hipStream_t)stream, a, an->n_loci,
                      (uint32_t *)(dm + o_ok), (uint64_t *)(dm + o_adj), (uint64_t *)(dm + o_adjh));
   hipLaunchKernelGGL(sb::iso_masks_kernel, dim3((unsigned)(ib < cap * 4 ? ib : cap * 4)), dim3(256), 0, (hipStream_t)stream, a, an->n_loci, n_iso,
                      (uint64_t *)(dm + o_mem), (uint64_t *)(dm + o_sta), (uint32_t *)(dm + o_ok), (uint64_t *)(dm + o_memh), (uint64_t *)(dm + o_stah));
   const hipError_t e = hipGetLastError();
   if (e != hipSuccess) return api_fail(SBGPU_EHIP, std::string("iso_masks_kernel: ") + hipGetErrorString(e));
   out->member = (const uint64_t *)(dm + o_mem);
   out->start = (const uint64_t *)(dm + o_sta);
   out->adj = (const uint64_t *)(dm + o_adj);
   out->member_hi = (const uint64_t *)(dm + o_memh);
   out->start_hi = (const uint64_t *)(dm + o_stah);
   out->adj_hi = (const uint64_t *)(dm + o_adjh);
   out->ok = (const uint32_t *)(dm + o_ok);
   return SBGPU_OK;
}

int exonbin_device_impl(sbgpu_ctx_t *c, const sbgpu_annotation_t *an, const sbgpu_hits_t *hits, int32_t compat_words,
                        int32_t key_words, uint32_t *d_compat, uint32_t *d_key, uint64_t *d_span, uint32_t *d_fhash, void *stream,
                        int64_t n_iso, const DeviceSegBasis *seg_pre)
{
   if (!c || !an || !hits) return api_fail(SBGPU_EINVAL, "sbgpu_exonbin_device: null argument");
   if (hits->n_hits == 0) return SBGPU_OK;
   if (hits->n_hits < 0 || an->n_loci < 1 || compat_words < 0 || key_words < 0)
      return api_fail(SBGPU_EINVAL, "sbgpu_exonbin_device: bad counts");
   if (!an->iso_off || !an->exon_off || !an->seg_off || !hits->hit_locus || !hits->feat_off ||
       (compat_words && !d_compat) || (key_words && !d_key))
      return api_fail(SBGPU_EINVAL, "sbgpu_exonbin_device: null device pointer");
   sb::ExonBinArgs a;
   a.iso_off = an->iso_off;
   a.exon_off = an->exon_off;
   a.exon_left = an->exon_left;
   a.exon_right = an->exon_right;
   a.seg_off = an->seg_off;
   a.seg_left = an->seg_left;
   a.seg_right = an->seg_right;
   a.n_hits = hits->n_hits;
   a.hit_locus = hits->hit_locus;
   a.feat_off = hits->feat_off;
   a.feat_code = hits->feat_code;
   a.feat_left = hits->feat_left;
   a.feat_right = hits->feat_right;
   a.compat_words = compat_words;
   a.key_words = key_words;
   a.compat = d_compat;
   a.key = d_key;
   a.span = d_span;
   a.fhash = d_span ? d_fhash : nullptr;
   a.iso_member = a.iso_start = nullptr;
   a.locus_seg_ok = nullptr;
   a.locus_adj = nullptr;
   a.iso_member_hi = a.iso_start_hi = a.locus_adj_hi = nullptr;
   // the isoforms in the segment basis (exonbin_device.h): loci of up to 64 segments (key_words <= 2 covers them all)
   static const bool seg_basis = !(sb::exp_env("SBGPU_EXONBIN_SEGBASIS") && std::atoi(sb::exp_env("SBGPU_EXONBIN_SEGBASIS")) == 0);
   if (seg_basis && key_words >= 1 && compat_words >= 1) {
      DeviceSegBasis sbz;
      if (seg_pre) {
         sbz = *seg_pre; // made once for a resident annotation
      } else {
         if (n_iso < 0) { // a caller with a device annotation only: one 8-byte read
            hipError_t ec = hipMemcpyAsync(&n_iso, an->iso_off + an->n_loci, 8, hipMemcpyDeviceToHost, (hipStream_t)stream);
            if (ec == hipSuccess) ec = hipStreamSynchronize((hipStream_t)stream);
            if (ec != hipSuccess) return api_fail(SBGPU_EHIP, std::string("sbgpu_exonbin_device: ") + hipGetErrorString(ec));
         }
         char *dm = nullptr;
         hipError_t em = sb::ctx_scratch(c, 3, seg_basis_bytes(an->n_loci, n_iso), &dm);
         if (em != hipSuccess) return api_fail(em == hipErrorOutOfMemory ? SBGPU_ENOMEM : SBGPU_EHIP, std::string("hipMalloc: ") + hipGetErrorString(em));
         const int rc = make_seg_basis(c, an, n_iso, dm, stream, &sbz);
         if (rc != SBGPU_OK) return rc;
      }
      a.iso_member = sbz.member;
      a.iso_start = sbz.start;
      a.locus_seg_ok = sbz.ok;
      a.locus_adj = sbz.adj;
      a.iso_member_hi = sbz.member_hi;
      a.iso_start_hi = sbz.start_hi;
      a.locus_adj_hi = sbz.adj_hi;
   }
   const int64_t blocks_wanted = (hits->n_hits + 255) / 256;
   if (blocks_wanted > 0x7fffffff) return api_fail(SBGPU_ESHAPE, "sbgpu_exonbin_device: more than 2^39 hits in one call");
   static const bool lane_form = sb::exp_env("SBGPU_EXONBIN_LANE") && std::atoi(sb::exp_env("SBGPU_EXONBIN_LANE")) != 0;
   if (lane_form) { // per-lane kernel only (A/B measurements)
      const int64_t cap = (int64_t)sb::ctx_cu_count(c) * 32;
      hipLaunchKernelGGL(sb::exonbin_lane_kernel, dim3((unsigned)(blocks_wanted < cap ? blocks_wanted : cap)), dim3(256), 0,
                         (hipStream_t)stream, a);
      if (d_span)
         hipLaunchKernelGGL(sb::hit_signature_kernel, dim3((unsigned)(blocks_wanted < cap ? blocks_wanted : cap)), dim3(256), 0,
                            (hipStream_t)stream, a.n_hits, a.feat_off, a.feat_left, a.feat_right, d_span, d_fhash);
   } else {
      hipLaunchKernelGGL(sb::exonbin_kernel, dim3((unsigned)blocks_wanted), dim3(256), 0, (hipStream_t)stream, a);
      // loci of 65-128 segments or isoforms (form 3 of iso_masks_kernel): their regular hits in the 128-bit segment basis.
      // Only an annotation with words beyond two can have such a locus; everywhere else the kernel is not launched.
      if (a.locus_seg_ok && (key_words > 2 || compat_words > 2))
         hipLaunchKernelGGL(sb::exonbin_seg128_kernel, dim3((unsigned)blocks_wanted), dim3(256), 0, (hipStream_t)stream, a);
   }
   hipError_t e = hipGetLastError();
   if (e != hipSuccess) return api_fail(SBGPU_EHIP, std::string("exonbin_kernel: ") + hipGetErrorString(e));
   return SBGPU_OK;
}
} // namespace sb

extern "C" {

int sbgpu_exonbin_device(sbgpu_ctx_t *c, const sbgpu_annotation_t *an, const sbgpu_hits_t *hits,
                         int32_t compat_words, int32_t key_words, uint32_t *d_compat, uint32_t *d_key,
                         void *stream)
{
   return sb::exonbin_device_impl(c, an, hits, compat_words, key_words, d_compat, d_key, nullptr, nullptr, stream, -1);
}

int sbgpu_exonbin_host(sbgpu_ctx_t *c, const sbgpu_annotation_t *an, const sbgpu_hits_t *hits,
                       int32_t compat_words, int32_t key_words, uint32_t *compat_out, uint32_t *key_out)
{
   if (!c || !an || !hits) return api_fail(SBGPU_EINVAL, "sbgpu_exonbin_host: null argument");
   const int64_t nh = hits->n_hits, nl = an->n_loci;
   if (nh == 0) return SBGPU_OK;
   if (nh < 0 || nl < 1 || compat_words < 0 || key_words < 0)
      return api_fail(SBGPU_EINVAL, "sbgpu_exonbin_host: bad counts");
   if (!an->iso_off || !an->exon_off || !an->seg_off || !hits->hit_locus || !hits->feat_off ||
       (compat_words && !compat_out) || (key_words && !key_out))
      return api_fail(SBGPU_EINVAL, "sbgpu_exonbin_host: null argument");
   // ---- validate the CSR structure (the device form trusts it)
   if (an->iso_off[0] != 0 || an->exon_off[0] != 0 || an->seg_off[0] != 0 || hits->feat_off[0] != 0)
      return api_fail(SBGPU_EINVAL, "sbgpu_exonbin_host: offsets must start at 0");
   for (int64_t l = 0; l < nl; ++l) {
      const int64_t ni = an->iso_off[l + 1] - an->iso_off[l], ns = an->seg_off[l + 1] - an->seg_off[l];
      if (ni < 0 || ns < 0) return api_fail(SBGPU_EINVAL, "sbgpu_exonbin_host: decreasing offsets");
      if (ni > 32 * (int64_t)compat_words)
         return api_fail(SBGPU_ESHAPE, "sbgpu_exonbin_host: compat_words does not cover a locus' isoforms");
      if (ns > 32 * (int64_t)key_words)
         return api_fail(SBGPU_ESHAPE, "sbgpu_exonbin_host: key_words does not cover a locus' segments");
   }
   const int64_t n_iso = an->iso_off[nl], n_seg = an->seg_off[nl];
   for (int64_t i = 0; i < n_iso; ++i)
      if (an->exon_off[i + 1] < an->exon_off[i]) return api_fail(SBGPU_EINVAL, "sbgpu_exonbin_host: decreasing exon_off");
   const int64_t n_exon = an->exon_off[n_iso];
   if ((n_exon && (!an->exon_left || !an->exon_right)) || (n_seg && (!an->seg_left || !an->seg_right)))
      return api_fail(SBGPU_EINVAL, "sbgpu_exonbin_host: null coordinate array");
   for (int64_t h = 0; h < nh; ++h) {
      if (hits->hit_locus[h] < 0 || hits->hit_locus[h] >= nl)
         return api_fail(SBGPU_EINVAL, "sbgpu_exonbin_host: hit_locus out of range");
      if (hits->feat_off[h + 1] < hits->feat_off[h]) return api_fail(SBGPU_EINVAL, "sbgpu_exonbin_host: decreasing feat_off");
   }
   const int64_t n_feat = hits->feat_off[nh];
   if (n_feat && (!hits->feat_code || !hits->feat_left || !hits->feat_right))
      return api_fail(SBGPU_EINVAL, "sbgpu_exonbin_host: null feature array");

   // ---- one device arena, one upload per array
   struct Part {
      const void *src;
      size_t bytes, off;
   };
   Part parts[] = {
      {an->iso_off, (size_t)(nl + 1) * 8, 0},     {an->exon_off, (size_t)(n_iso + 1) * 8, 0},
      {an->seg_off, (size_t)(nl + 1) * 8, 0},     {hits->feat_off, (size_t)(nh + 1) * 8, 0},
      {an->exon_left, (size_t)n_exon * 4, 0},     {an->exon_right, (size_t)n_exon * 4, 0},
      {an->seg_left, (size_t)n_seg * 4, 0},       {an->seg_right, (size_t)n_seg * 4, 0},
      {hits->hit_locus, (size_t)nh * 4, 0},       {hits->feat_left, (size_t)n_feat * 4, 0},
      {hits->feat_right, (size_t)n_feat * 4, 0},  {hits->feat_code, (size_t)n_feat, 0},
   };
   size_t total = 0;
   for (Part &p : parts) {
      p.off = total;
      total += (p.bytes + 255) & ~(size_t)255;
   }
   const size_t off_compat = total;
   total += (((size_t)nh * compat_words * 4) + 255) & ~(size_t)255;
   const size_t off_key = total;
   total += (((size_t)nh * key_words * 4) + 255) & ~(size_t)255;
   char *d = nullptr;
   hipError_t e = hipMalloc(&d, total ? total : 256);
   if (e != hipSuccess)
      return api_fail(e == hipErrorOutOfMemory ? SBGPU_ENOMEM : SBGPU_EHIP, std::string("hipMalloc: ") + hipGetErrorString(e));
   hipStream_t s = sb::ctx_stream(c);
   auto bail = [&](hipError_t err, const char *what) {
      (void)hipFree(d);
      return api_fail(SBGPU_EHIP, std::string(what) + ": " + hipGetErrorString(err));
   };
   for (Part &p : parts)
      if (p.bytes && (e = hipMemcpyAsync(d + p.off, p.src, p.bytes, hipMemcpyHostToDevice, s)) != hipSuccess)
         return bail(e, "hipMemcpyAsync(H2D)");
   sbgpu_annotation_t dan = *an;
   dan.iso_off = (const int64_t *)(d + parts[0].off);
   dan.exon_off = (const int64_t *)(d + parts[1].off);
   dan.seg_off = (const int64_t *)(d + parts[2].off);
   dan.exon_left = (const uint32_t *)(d + parts[4].off);
   dan.exon_right = (const uint32_t *)(d + parts[5].off);
   dan.seg_left = (const uint32_t *)(d + parts[6].off);
   dan.seg_right = (const uint32_t *)(d + parts[7].off);
   sbgpu_hits_t dh = *hits;
   dh.feat_off = (const int64_t *)(d + parts[3].off);
   dh.hit_locus = (const int32_t *)(d + parts[8].off);
   dh.feat_left = (const uint32_t *)(d + parts[9].off);
   dh.feat_right = (const uint32_t *)(d + parts[10].off);
   dh.feat_code = (const uint8_t *)(d + parts[11].off);
   int rc = sbgpu_exonbin_device(c, &dan, &dh, compat_words, key_words, (uint32_t *)(d + off_compat),
                                 (uint32_t *)(d + off_key), s);
   if (rc != SBGPU_OK) {
      (void)hipFree(d);
      return rc;
   }
   if (compat_words &&
       (e = hipMemcpyAsync(compat_out, d + off_compat, (size_t)nh * compat_words * 4, hipMemcpyDeviceToHost, s)) != hipSuccess)
      return bail(e, "hipMemcpyAsync(D2H compat)");
   if (key_words && (e = hipMemcpyAsync(key_out, d + off_key, (size_t)nh * key_words * 4, hipMemcpyDeviceToHost, s)) != hipSuccess)
      return bail(e, "hipMemcpyAsync(D2H key)");
   if ((e = hipStreamSynchronize(s)) != hipSuccess) return bail(e, "hipStreamSynchronize");
   (void)hipFree(d);
   return SBGPU_OK;
}

} // extern "C"

namespace sb {
// The isoforms' segment lists (Isoform::_exon_segs, include/isoform.h:59-71; contig.cpp:615-634), their loci and
// exonic lengths, from the host annotation: two passes over the loci on host threads (count, then write).
void iso_segments(const sbgpu_annotation_t *an, IsoSegments *out)
{
   const int64_t nl = an->n_loci, n_iso = an->iso_off[nl];
   std::vector<int64_t> &iso_seg_off = out->seg_off;
   std::vector<int32_t> &iso_seg_idx = out->seg_idx, &iso_locus = out->locus, &iso_len = out->len;
   iso_seg_off.assign((size_t)n_iso + 1, 0);
   iso_locus.assign((size_t)(n_iso > 0 ? n_iso : 1), 0);
   iso_len.assign((size_t)(n_iso > 0 ? n_iso : 1), 0);
   iso_seg_idx.clear();
   {
      // two passes over the loci on host threads: count the segments of every isoform, then write them
      unsigned nt = std::thread::hardware_concurrency();
      if (nt > 16) nt = 16;
      if (const char *ev = std::getenv("SBGPU_HOST_THREADS")) nt = (unsigned)std::atoi(ev);
      if (nt < 1) nt = 1;
      if ((int64_t)nt * 2048 > nl) nt = (unsigned)std::max<int64_t>(1, nl / 2048);
      auto walk = [&](int64_t l, bool fill) {
         const int64_t s0 = an->seg_off[l], nseg = an->seg_off[l + 1] - s0;
         for (int64_t iso = an->iso_off[l]; iso < an->iso_off[l + 1]; ++iso) {
            const int64_t e0 = an->exon_off[iso], ne = an->exon_off[iso + 1] - e0;
            int64_t e = 0, n = 0;
            int32_t *dst = fill ? iso_seg_idx.data() + iso_seg_off[(size_t)iso] : nullptr;
            for (int64_t sidx = 0; sidx < nseg; ++sidx) {
               const uint32_t sl = an->seg_left[s0 + sidx], sr = an->seg_right[s0 + sidx];
               while (e < ne && an->exon_right[e0 + e] < sl) ++e; // contig.cpp:615-634; segments ascend
               if (e < ne && an->exon_left[e0 + e] <= sl && an->exon_right[e0 + e] >= sr) {
                  if (fill) dst[n] = (int32_t)sidx;
                  ++n;
               }
            }
            if (!fill) {
               int64_t len = 0;
               for (int64_t x = 0; x < ne; ++x) len += (int64_t)an->exon_right[e0 + x] - an->exon_left[e0 + x] + 1;
               iso_seg_off[(size_t)iso + 1] = n; // a count for now
               iso_locus[(size_t)iso] = (int32_t)l;
               iso_len[(size_t)iso] = (int32_t)len;
            }
         }
      };
      auto run = [&](bool fill) {
         if (nt <= 1) {
            for (int64_t l = 0; l < nl; ++l) walk(l, fill);
            return;
         }
         std::vector<std::thread> pool;
         for (unsigned t = 0; t < nt; ++t)
            pool.emplace_back([&, t]() {
               for (int64_t l = nl * t / nt; l < nl * (t + 1) / nt; ++l) walk(l, fill);
            });
         for (auto &th : pool) th.join();
      };
      run(false);
      for (int64_t i = 0; i < n_iso; ++i) iso_seg_off[(size_t)i + 1] += iso_seg_off[(size_t)i];
      iso_seg_idx.resize((size_t)iso_seg_off[(size_t)n_iso]);
      run(true);
   }
}

int bins_create_device_impl(sbgpu_ctx_t *c, const sbgpu_annotation_t *an, const sbgpu_hits_t *dh, const float *d_mass,
                            const int64_t *locus_hit_off, int32_t compat_words, int32_t key_words, const uint32_t *d_compat,
                            const uint32_t *d_key, int64_t *d_hit_bin, void *stream, const IsoSegments *iso_pre, sbgpu_bins_t **out,
                            const uint64_t *d_span, const uint32_t *d_fhash, const GroupingHooks *hooks)
{
   if (!c || !an || !dh || !locus_hit_off || !out) return api_fail(SBGPU_EINVAL, "sbgpu_bins_create_device: null argument");
   *out = nullptr;
   const int64_t nl = an->n_loci, nh = dh->n_hits;
   if (nl < 1 || nh < 0 || compat_words < 1 || key_words < 1) return api_fail(SBGPU_EINVAL, "sbgpu_bins_create_device: bad counts");
   if (nh > 0x7fffffff) return api_fail(SBGPU_ESHAPE, "sbgpu_bins_create_device: more than 2^31 hits in one call");
   if (nh && (!dh->feat_off || !dh->feat_left || !dh->feat_right || !d_mass || !d_compat || !d_key))
      return api_fail(SBGPU_EINVAL, "sbgpu_bins_create_device: null device pointer");
   if (locus_hit_off[0] != 0 || locus_hit_off[nl] != nh) return api_fail(SBGPU_EINVAL, "sbgpu_bins_create_device: locus_hit_off does not span the hits");
   for (int64_t l = 0; l < nl; ++l) {
      if (locus_hit_off[l + 1] < locus_hit_off[l]) return api_fail(SBGPU_EINVAL, "sbgpu_bins_create_device: decreasing locus_hit_off");
      if (an->iso_off[l + 1] - an->iso_off[l] > 32 * (int64_t)compat_words || an->seg_off[l + 1] - an->seg_off[l] > 32 * (int64_t)key_words)
         return api_fail(SBGPU_ESHAPE, "sbgpu_bins_create_device: word counts do not cover a locus");
   }
   hipStream_t s = (hipStream_t)stream;
   const char *timing_env = std::getenv("SBGPU_HOST_TIMING");
   const bool timing = timing_env != nullptr, timing_sync = timing && std::atoi(timing_env) != 2; // diagnostic: stage times on stderr; =2: host clock only, no synchronisation
   auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
   double t_stage = now();
   auto stage = [&](const char *name) {
      if (timing) {
         if (timing_sync) (void)hipStreamSynchronize(s);
         const double t = now();
         std::fprintf(stderr, "  bins_create_device: %-16s %.2f ms\n", name, (t - t_stage) * 1e3);
         t_stage = t;
      }
   };
   // one arena for the scratch: [hit_bin_local | bin_rep | bin_count | bin_compat | n_bins | n_used | flags | offsets x2]
   const size_t nh1 = (size_t)(nh > 0 ? nh : 1);
   auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
   size_t off = 0;
   const size_t o_local = off; off += up(nh1 * 4);
   const size_t o_rep = off; off += up(nh1 * 4);
   const size_t o_zero = off; // bin_count + bin_compat are zeroed together
   const size_t o_cnt = off; off += up(nh1 * 4);
   const size_t o_cmp = off; off += up(nh1 * 4 * (size_t)compat_words);
   const size_t zero_bytes = off - o_zero;
   const size_t o_nb = off; off += up((size_t)nl * 4);
   const size_t o_nu = off; off += up((size_t)nl * 4);
   const size_t o_flag = off; off += 256; // [0]: the grouping kernels' flags, [1]: the middle's (pack, pairs)
   const size_t o_hoff = off; off += up((size_t)(nl + 1) * 8);
   const size_t o_roff = off; off += up((size_t)(nl + 1) * 8);
   const size_t o_order = off; off += up((size_t)nl * 4);
   const size_t o_dup = off; off += up(nh1);
   const size_t o_span = off; off += d_span ? 0 : up(nh1 * 8);   // spans and hashes: the exon-bin kernel's, or made here
   const size_t o_fhash = off; off += d_span ? 0 : up(nh1 * 4);
   // the scratch arenas live with the context (sb::ctx_scratch): nothing to free here (d2 is not scratch: below)
   char *d = nullptr, *d2 = nullptr;
   hipError_t e = sb::ctx_scratch(c, 2, off, &d);
   if (e != hipSuccess) return api_fail(e == hipErrorOutOfMemory ? SBGPU_ENOMEM : SBGPU_EHIP, std::string("hipMalloc: ") + hipGetErrorString(e));
   auto bail = [&](int code, const std::string &msg) { return api_fail(code, msg); };
#define SB_TRY(expr)                                                                        \
   do {                                                                                     \
      hipError_t e_ = (expr);                                                               \
      if (e_ != hipSuccess) return bail(SBGPU_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
   } while (0)
   sb::BinsArgs a;
   a.n_loci = nl;
   a.locus_hit_off = (const int64_t *)(d + o_hoff);
   a.feat_off = dh->feat_off;
   a.feat_left = dh->feat_left;
   a.feat_right = dh->feat_right;
   a.mass = d_mass;
   a.compat_words = compat_words;
   a.key_words = key_words;
   a.compat = d_compat;
   a.key = d_key;
   if (!d_span && nh) {
      const int64_t blocks = (nh + 255) / 256, capb = (int64_t)sb::ctx_cu_count(c) * 32;
      hipLaunchKernelGGL(sb::hit_signature_kernel, dim3((unsigned)(blocks < capb ? blocks : capb)), dim3(256), 0, s, nh,
                         dh->feat_off, dh->feat_left, dh->feat_right, (uint64_t *)(d + o_span), (uint32_t *)(d + o_fhash));
      SB_TRY(hipGetLastError());
   }
   a.span = d_span ? d_span : (const uint64_t *)(d + o_span);
   a.fhash = d_span ? d_fhash : (const uint32_t *)(d + o_fhash);
   a.hit_bin_local = (int32_t *)(d + o_local);
   a.bin_rep = (int32_t *)(d + o_rep);
   a.bin_count = (int32_t *)(d + o_cnt);
   a.bin_compat = (uint32_t *)(d + o_cmp);
   a.dup = (uint8_t *)(d + o_dup);
   a.n_bins = (int32_t *)(d + o_nb);
   a.n_used = (int32_t *)(d + o_nu);
   a.flags = (int32_t *)(d + o_flag);
   const int64_t cap = (int64_t)sb::ctx_cu_count(c) * 8;
   const int64_t n_iso = an->iso_off[nl], n_seg = an->seg_off[nl];
   const sbgpu_annotation_t *d_an = hooks ? hooks->d_annot : nullptr;
   // ---- the rest of the device memory, all sized before anything runs (bins <= hits), so that the kernels below
   // follow each other in the stream: the packed per-bin arrays, and what the pairs' kernels read
   // (an allocation of its own, not context scratch: the handle takes it over, see DeviceBinArrays)
   size_t off2 = 0;
   const size_t p_cnt = off2; off2 += up(nh1 * 4);
   const size_t p_key = off2; off2 += up(nh1 * 4 * (size_t)key_words);
   const size_t p_cmp = off2; off2 += up(nh1 * 4 * (size_t)compat_words);
   size_t d2_cap = 0;
   e = sb::dev_take(off2, &d2, &d2_cap);
   if (e != hipSuccess) return bail(e == hipErrorOutOfMemory ? SBGPU_ENOMEM : SBGPU_EHIP, std::string("hipMalloc: ") + hipGetErrorString(e));
   struct ArenaGuard { // given back on every early return; released once the handle owns it
      char *&p;
      size_t &cap;
      hipStream_t s;
      ~ArenaGuard()
      {
         if (p) {
            (void)hipStreamSynchronize(s);
            sb::dev_give(p, cap);
         }
      }
   } d2_guard = {d2, d2_cap, s};
   sb::IsoSegments own_segments;
   if (!iso_pre) { // a caller that knows the annotation earlier makes them while something else runs (chain_api.hip)
      sb::iso_segments(an, &own_segments);
      iso_pre = &own_segments;
   }
   const std::vector<int64_t> &iso_seg_off = iso_pre->seg_off;
   const std::vector<int32_t> &iso_seg_idx = iso_pre->seg_idx, &iso_locus = iso_pre->locus, &iso_len = iso_pre->len;
   const size_t ni1 = (size_t)(n_iso > 0 ? n_iso : 1), nsi1 = iso_seg_idx.empty() ? 1 : iso_seg_idx.size();
   size_t off3 = 0;
   const size_t r_isoff = off3; off3 += up((size_t)(nl + 1) * 8);
   const size_t r_foff = off3; off3 += up((size_t)(nl + 1) * 8);
   const size_t r_segoff = off3; off3 += up((size_t)(nl + 1) * 8);
   const size_t r_segl = off3; off3 += up((size_t)(n_seg + 1) * 4);
   const size_t r_segr = off3; off3 += up((size_t)(n_seg + 1) * 4);
   const size_t r_isegoff = off3; off3 += up((ni1 + 1) * 8);
   const size_t r_isegidx = off3; off3 += up(nsi1 * 4);
   const size_t r_isoloc = off3; off3 += up(ni1 * 4);
   const size_t r_isolen = off3; off3 += up(ni1 * 4);
   const size_t r_pcnt = off3; off3 += up(ni1 * 4);
   const size_t r_scnt = off3; off3 += up(ni1 * 4);
   const size_t r_poff = off3; off3 += up((ni1 + 1) * 8);
   const size_t r_soff = off3; off3 += up((ni1 + 1) * 8);
   const size_t r_tot = off3; off3 += 256; // totals of the two scans: [n_bins, n_elem, n_pairs, n_pair_segs]
   const int64_t row_tiles = (nl + sb::kScanTile - 1) / sb::kScanTile, iso_tiles = std::max<int64_t>(1, (n_iso + sb::kScanTile - 1) / sb::kScanTile);
   const size_t r_part = off3; off3 += up((size_t)(2 * std::max(row_tiles, iso_tiles) + 2) * 8); // the scans' tile sums
   char *d3 = nullptr;
   e = sb::ctx_scratch(c, 4, off3, &d3);
   if (e != hipSuccess) return bail(e == hipErrorOutOfMemory ? SBGPU_ENOMEM : SBGPU_EHIP, std::string("hipMalloc: ") + hipGetErrorString(e));
   // pinned host words for what comes back between the kernels (copies into pageable memory would hold the host
   // until the stream has reached them): [bins per locus | hits used per locus | flags | totals]
   const size_t h_nb = 0, h_nu = up((size_t)nl * 4), h_flag = h_nu + up((size_t)nl * 4), h_tot = h_flag + 256;
   char *pin = nullptr;
   e = sb::ctx_pinned(c, 0, h_tot + 256, &pin);
   if (e != hipSuccess) return bail(SBGPU_ENOMEM, std::string("hipHostMalloc: ") + hipGetErrorString(e));
   int32_t *nb = (int32_t *)(pin + h_nb), *nu = (int32_t *)(pin + h_nu);
   volatile int32_t *flags_h = (volatile int32_t *)(pin + h_flag);
   volatile int64_t *totals_h = (volatile int64_t *)(pin + h_tot);
   hipStream_t cs = nullptr;
   hipEvent_t ev_up = nullptr, ev_rows = nullptr, ev_in = nullptr, ev_acc = nullptr;
   SB_TRY(sb::ctx_copy_stream(c, &cs));
   SB_TRY(sb::ctx_event(c, 0, &ev_up));
   SB_TRY(sb::ctx_event(c, 1, &ev_rows));
   SB_TRY(sb::ctx_event(c, 2, &ev_in));
   SB_TRY(sb::ctx_event(c, 4, &ev_acc));
   // ---- uploads on the copy stream, beside whatever the main stream still runs (the caller's exon-bin kernel): a copy
   // in the main stream between two kernels costs the hand-over to the DMA engine and back, 0.3 ms each way measured.
   // First what the grouping kernels read (hit offsets, the loci's order), then the annotation's part of the pairs' inputs.
   // (Streams share hardware queues -- the copy stream may sit in the main stream's queue and then runs in issue order
   // with it -- so nothing here counts on the overlap.)
   SB_TRY(hipMemcpyAsync(d + o_hoff, locus_hit_off, (size_t)(nl + 1) * 8, hipMemcpyHostToDevice, cs));
   // The bins' counts and compat words are [n_hits]-sized (a locus' bins sit at its hit offset): 1.3 GB for 1.6e8 hits,
   // 0.18 ms to zero at the memory's rate.  Only bins_locus_kernel's global atomics need the zeros -- the two-pass form
   // and the big table; the single-pass kernels write every entry they own -- so they are made where those run.
   auto zero_bins = [&](hipStream_t zs) -> hipError_t {
      const int64_t n16 = (int64_t)(zero_bytes / 16); // (the arena's parts are 256-byte multiples)
      hipLaunchKernelGGL(sb::bins_zero_kernel, dim3((unsigned)std::min<int64_t>((n16 + 255) / 256, cap * 4)), dim3(256), 0, zs, (uint4 *)(d + o_zero), n16);
      return hipGetLastError();
   };
   hipLaunchKernelGGL(sb::bins_zero_kernel, dim3(1), dim3(256), 0, cs, (uint4 *)(d + o_flag), (int64_t)16);
   SB_TRY(hipGetLastError());
   const int64_t *d_iso_off = d_an ? d_an->iso_off : (const int64_t *)(d3 + r_isoff);
   // small and big loci in two launches (two LDS table sizes)
   std::vector<int32_t> order((size_t)nl);
   int64_t n_small = 0;
   for (int64_t l = 0; l < nl; ++l)
      if (locus_hit_off[l + 1] - locus_hit_off[l] <= sb::kBinsSmallHits) order[(size_t)n_small++] = (int32_t)l;
   int64_t n_big = 0;
   for (int64_t l = 0; l < nl; ++l)
      if (locus_hit_off[l + 1] - locus_hit_off[l] > sb::kBinsSmallHits) order[(size_t)(n_small + n_big++)] = (int32_t)l;
   // the very big ones first, largest to smallest: a locus is one workgroup's work, and a locus of 10^5 hits that starts
   // last would be the kernel's tail (only those few are sorted: a full sort of 55 000 loci costs 2.6 ms of host time).
   // They also get a launch of their own with 1024 threads; the loci between -- a few thousand hits, two or three per
   // thread of such a workgroup -- are bound by the per-locus fixed cost (table set-up, six workgroup barriers, the
   // ranking), which 256-thread workgroups, several per CU, pay side by side: 2.62 -> 1.83 ms on the chain sample
   // (SBGPU_BINS_MID_THREADS=1024|512|256, SBGPU_BINS_HEAVY_HITS=n for A/B; 8192 was the best of 2000 ... 100 000).
   int64_t n_heavy = 0;
   static const int64_t heavy_env = sb::exp_env("SBGPU_BINS_HEAVY_HITS") ? std::atoll(sb::exp_env("SBGPU_BINS_HEAVY_HITS")) : 0;
   {
      const int64_t heavy = heavy_env > 0 ? heavy_env : std::max<int64_t>(8192, 3 * (nh / std::max<int64_t>(nl, 1)));
      auto hits_of = [&](int32_t l) { return locus_hit_off[l + 1] - locus_hit_off[l]; };
      auto first_big = order.begin() + n_small, last_big = first_big + n_big;
      auto mid = std::stable_partition(first_big, last_big, [&](int32_t l) { return hits_of(l) >= heavy; });
      std::sort(first_big, mid, [&](int32_t x, int32_t y) { return hits_of(x) != hits_of(y) ? hits_of(x) > hits_of(y) : x < y; });
      n_heavy = mid - first_big;
   }
   // the single-pass kernels (bins_device.h) where a bin's compat union fits two words; SBGPU_BINS_TWO_PASS=1: the older form (A/B)
   static const bool two_pass_env = sb::exp_env("SBGPU_BINS_TWO_PASS") && std::atoi(sb::exp_env("SBGPU_BINS_TWO_PASS")) != 0;
   const bool single_pass = compat_words <= 2 && key_words <= 2 && !two_pass_env;
   if (!single_pass) SB_TRY(zero_bins(cs));
   SB_TRY(hipMemcpyAsync(d + o_order, order.data(), (size_t)nl * 4, hipMemcpyHostToDevice, cs));
   SB_TRY(hipEventRecord(ev_in, cs));
   SB_TRY(hipStreamWaitEvent(s, ev_in, 0));
   sb::ctx_stage_begin(c, single_pass ? "bins_accum_kernel" : "bins_locus_kernel", s);
   if (n_small) {
      a.n_loci = n_small;
      a.loci = (const int32_t *)(d + o_order);
      const dim3 grid((unsigned)std::min<int64_t>(n_small, cap * 4));
      if (single_pass && compat_words == 1) hipLaunchKernelGGL((sb::bins_accum_kernel<sb::kBinsSlotsSmall, sb::kBinsMaxSmall, sb::kBinsThreads, 1>), grid, dim3(sb::kBinsThreads), 0, s, a);
      else if (single_pass) hipLaunchKernelGGL((sb::bins_accum_kernel<sb::kBinsSlotsSmall, sb::kBinsMaxSmall, sb::kBinsThreads, 2>), grid, dim3(sb::kBinsThreads), 0, s, a);
      else hipLaunchKernelGGL((sb::bins_locus_kernel<sb::kBinsSlotsSmall, sb::kBinsMaxSmall, sb::kBinsThreads>), grid, dim3(sb::kBinsThreads), 0, s, a);
      SB_TRY(hipGetLastError());
   }
   if (n_big) {
      // first the middle table; whatever has more bins than it holds is redone below
      a.n_loci = n_big;
      a.loci = (const int32_t *)(d + o_order) + n_small;
      const dim3 grid((unsigned)std::min<int64_t>(n_big, cap * 4));
      static const int mid_threads = sb::exp_env("SBGPU_BINS_MID_THREADS") ? std::atoi(sb::exp_env("SBGPU_BINS_MID_THREADS")) : 256;
      if (single_pass && mid_threads != 1024 && n_heavy < n_big) {
         // the heavy loci on 1024 threads, the others on fewer (more workgroups per CU, cheaper barriers)
         if (n_heavy) {
            a.n_loci = n_heavy;
            const dim3 gh((unsigned)std::min<int64_t>(n_heavy, cap * 4));
            if (compat_words == 1) hipLaunchKernelGGL((sb::bins_accum_kernel<sb::kBinsSlotsMid, sb::kBinsMaxMid, sb::kBinsThreadsAccum, 1, true>), gh, dim3(sb::kBinsThreadsAccum), 0, s, a);
            else hipLaunchKernelGGL((sb::bins_accum_kernel<sb::kBinsSlotsMid, sb::kBinsMaxMid, sb::kBinsThreadsAccum, 2, true>), gh, dim3(sb::kBinsThreadsAccum), 0, s, a);
         }
         a.n_loci = n_big - n_heavy;
         a.loci = (const int32_t *)(d + o_order) + n_small + n_heavy;
         const dim3 gm((unsigned)std::min<int64_t>(a.n_loci, cap * 8));
         // ... and on a table of 1024 slots / 512 bins (24 KB of LDS: six workgroups per CU instead of three; a locus of
         // more bins is redone with the big table like any other overflow): 1.83 -> 1.41 ms (SBGPU_BINS_MID_SLOTS=2048 for A/B)
         static const int mid_slots = sb::exp_env("SBGPU_BINS_MID_SLOTS") ? std::atoi(sb::exp_env("SBGPU_BINS_MID_SLOTS")) : 1024;
         if (mid_threads == 256 && mid_slots == 1024 && compat_words == 1) hipLaunchKernelGGL((sb::bins_accum_kernel<1024, 512, 256, 1, true>), gm, dim3(256), 0, s, a);
         else if (mid_threads == 256 && mid_slots == 1024) hipLaunchKernelGGL((sb::bins_accum_kernel<1024, 512, 256, 2, true>), gm, dim3(256), 0, s, a);
         else if (mid_threads == 256 && compat_words == 1) hipLaunchKernelGGL((sb::bins_accum_kernel<sb::kBinsSlotsMid, sb::kBinsMaxMid, 256, 1, true>), gm, dim3(256), 0, s, a);
         else if (mid_threads == 256) hipLaunchKernelGGL((sb::bins_accum_kernel<sb::kBinsSlotsMid, sb::kBinsMaxMid, 256, 2, true>), gm, dim3(256), 0, s, a);
         else if (compat_words == 1) hipLaunchKernelGGL((sb::bins_accum_kernel<sb::kBinsSlotsMid, sb::kBinsMaxMid, 512, 1, true>), gm, dim3(512), 0, s, a);
         else hipLaunchKernelGGL((sb::bins_accum_kernel<sb::kBinsSlotsMid, sb::kBinsMaxMid, 512, 2, true>), gm, dim3(512), 0, s, a);
      } else
      if (single_pass && compat_words == 1) hipLaunchKernelGGL((sb::bins_accum_kernel<sb::kBinsSlotsMid, sb::kBinsMaxMid, sb::kBinsThreadsAccum, 1, true>), grid, dim3(sb::kBinsThreadsAccum), 0, s, a);
      else if (single_pass) hipLaunchKernelGGL((sb::bins_accum_kernel<sb::kBinsSlotsMid, sb::kBinsMaxMid, sb::kBinsThreadsAccum, 2, true>), grid, dim3(sb::kBinsThreadsAccum), 0, s, a);
      else hipLaunchKernelGGL((sb::bins_locus_kernel<sb::kBinsSlotsMid, sb::kBinsMaxMid, sb::kBinsThreadsMid, true>), grid, dim3(sb::kBinsThreadsMid), 0, s, a);
   }
   SB_TRY(hipGetLastError());
   sb::ctx_stage_end(c, s);
   // bins / hits used per locus and the grouping kernels' flags, to the pinned words -- on the copy stream, behind an
   // event of the main one, which goes straight on to the middle (a copy in the main stream holds the next kernel back
   // by the hand-over to the DMA engine: 0.25 ms measured here)
   auto fetch_rows = [&]() -> hipError_t {
      hipError_t x = hipEventRecord(ev_acc, s);
      if (x == hipSuccess) x = hipStreamWaitEvent(cs, ev_acc, 0);
      if (x == hipSuccess) x = hipMemcpyAsync(nb, d + o_nb, (size_t)nl * 4, hipMemcpyDeviceToHost, cs);
      if (x == hipSuccess) x = hipMemcpyAsync(nu, d + o_nu, (size_t)nl * 4, hipMemcpyDeviceToHost, cs);
      if (x == hipSuccess) x = hipMemcpyAsync((void *)flags_h, d + o_flag, 8, hipMemcpyDeviceToHost, cs);
      if (x == hipSuccess) x = hipEventRecord(ev_rows, cs);
      return x;
   };
   // (the annotation's part goes up while the grouping kernels run)
   if (!d_an) {
      SB_TRY(hipMemcpyAsync(d3 + r_isoff, an->iso_off, (size_t)(nl + 1) * 8, hipMemcpyHostToDevice, cs));
      SB_TRY(hipMemcpyAsync(d3 + r_segoff, an->seg_off, (size_t)(nl + 1) * 8, hipMemcpyHostToDevice, cs));
      if (n_seg) {
         SB_TRY(hipMemcpyAsync(d3 + r_segl, an->seg_left, (size_t)n_seg * 4, hipMemcpyHostToDevice, cs));
         SB_TRY(hipMemcpyAsync(d3 + r_segr, an->seg_right, (size_t)n_seg * 4, hipMemcpyHostToDevice, cs));
      }
   }
   const sb::DeviceIsoSegments *d_iso = hooks ? hooks->d_iso : nullptr;
   if (!d_iso) {
      SB_TRY(hipMemcpyAsync(d3 + r_isegoff, iso_seg_off.data(), (size_t)(n_iso + 1) * 8, hipMemcpyHostToDevice, cs));
      if (!iso_seg_idx.empty()) SB_TRY(hipMemcpyAsync(d3 + r_isegidx, iso_seg_idx.data(), iso_seg_idx.size() * 4, hipMemcpyHostToDevice, cs));
      if (n_iso) {
         SB_TRY(hipMemcpyAsync(d3 + r_isoloc, iso_locus.data(), (size_t)n_iso * 4, hipMemcpyHostToDevice, cs));
         SB_TRY(hipMemcpyAsync(d3 + r_isolen, iso_len.data(), (size_t)n_iso * 4, hipMemcpyHostToDevice, cs));
      }
   }
   SB_TRY(hipEventRecord(ev_up, cs));
   SB_TRY(fetch_rows());
   // ---- the middle: rows scanned, bins packed, pairs counted and scanned -- behind the grouping kernels without a
   // host round trip.  Its flags are a word of their own (o_flag + 4), cleared per launch: the middle may run a second
   // time (below).
   sb::BinsPackArgs pk;
   pk.n_loci = nl;
   pk.locus_hit_off = a.locus_hit_off;
   pk.row_off = (const int64_t *)(d + o_roff);
   pk.compat_words = compat_words;
   pk.key_words = key_words;
   pk.key = d_key;
   pk.hit_bin_local = a.hit_bin_local;
   pk.bin_rep = a.bin_rep;
   pk.bin_count_in = a.bin_count;
   pk.bin_compat_in = a.bin_compat;
   pk.count = (int32_t *)(d2 + p_cnt);
   pk.bin_key = (uint32_t *)(d2 + p_key);
   pk.bin_compat = (uint32_t *)(d2 + p_cmp);
   pk.hit_bin = d_hit_bin;
   pk.flags = a.flags + 1;
   sb::PairsArgs pa;
   pa.n_iso = n_iso;
   pa.iso_locus = d_iso ? d_iso->locus : (const int32_t *)(d3 + r_isoloc);
   pa.iso_off = d_iso_off;
   pa.row_off = pk.row_off;
   pa.f_off = (const int64_t *)(d3 + r_foff);
   pa.seg_off = d_an ? d_an->seg_off : (const int64_t *)(d3 + r_segoff);
   pa.seg_left = d_an ? d_an->seg_left : (const uint32_t *)(d3 + r_segl);
   pa.seg_right = d_an ? d_an->seg_right : (const uint32_t *)(d3 + r_segr);
   pa.iso_seg_off = d_iso ? d_iso->seg_off : (const int64_t *)(d3 + r_isegoff);
   pa.iso_seg_idx = d_iso ? d_iso->seg_idx : (const int32_t *)(d3 + r_isegidx);
   pa.iso_len = d_iso ? d_iso->len : (const int32_t *)(d3 + r_isolen);
   pa.compat_words = compat_words;
   pa.key_words = key_words;
   pa.bin_key = pk.bin_key;
   pa.bin_compat = pk.bin_compat;
   pa.pair_cnt = (int32_t *)(d3 + r_pcnt);
   pa.seg_cnt = (int32_t *)(d3 + r_scnt);
   pa.pair_off = (const int64_t *)(d3 + r_poff);
   pa.pseg_off = (const int64_t *)(d3 + r_soff);
   pa.pair_seg_off = nullptr;
   pa.pair_seg_lens = pa.pair_mask = nullptr;
   pa.pair_iso_len = nullptr;
   pa.pair_out_index = nullptr;
   pa.flags = a.flags + 1;
   // one wave per locus (bins_pairs_locus_kernel) for loci of up to 64 isoforms, the thread-per-isoform kernel for the
   // others; SBGPU_PAIRS_BY_ISOFORM=1: the latter everywhere (A/B)
   static const bool pairs_by_iso = sb::exp_env("SBGPU_PAIRS_BY_ISOFORM") && std::atoi(sb::exp_env("SBGPU_PAIRS_BY_ISOFORM")) != 0;
   bool any_wide_locus = false;
   for (int64_t l = 0; l < nl && !any_wide_locus; ++l) any_wide_locus = an->iso_off[l + 1] - an->iso_off[l] > 64;
   pa.only_wide_loci = pairs_by_iso ? 0 : 1;
   const unsigned lgrid = (unsigned)std::max<int64_t>(1, std::min<int64_t>((nl + 3) / 4, cap * 8));
   const unsigned pgrid = (unsigned)std::max<int64_t>(1, std::min<int64_t>((n_iso + 255) / 256, cap * 4));
   auto launch_middle = [&]() -> hipError_t {
      hipError_t x = hipMemsetAsync(d + o_flag + 4, 0, 4, s);
      if (x == hipSuccess) x = hipStreamWaitEvent(s, ev_up, 0);
      if (x != hipSuccess) return x;
      sb::ctx_stage_begin(c, "bins_scan_*_kernel<rows> + bins_pack_kernel", s);
      int64_t *part_a = (int64_t *)(d3 + r_part), *part_b = part_a + std::max(row_tiles, iso_tiles) + 1;
      sb::ScanArgs sr = {nl, a.n_bins, nullptr, d_iso_off, (int64_t *)(d + o_roff), (int64_t *)(d3 + r_foff), part_a, part_b, (int64_t *)(d3 + r_tot)};
      hipLaunchKernelGGL(sb::bins_scan_tiles_kernel<0>, dim3((unsigned)row_tiles), dim3(256), 0, s, sr);
      hipLaunchKernelGGL(sb::bins_scan_parts_kernel, dim3(1), dim3(256), 0, s, sr, row_tiles);
      hipLaunchKernelGGL(sb::bins_scan_apply_kernel<0>, dim3((unsigned)row_tiles), dim3(256), 0, s, sr);
      hipLaunchKernelGGL(sb::bins_pack_kernel, dim3((unsigned)(nl < cap ? nl : cap)), dim3(256), 0, s, pk);
      sb::ctx_stage_end(c, s);
      sb::ctx_stage_begin(c, "bins_pairs_kernel<count> + bins_scan_*_kernel<pairs>", s);
      if (pairs_by_iso || any_wide_locus) hipLaunchKernelGGL(sb::bins_pairs_kernel<false>, dim3(pgrid), dim3(256), 0, s, pa);
      if (!pairs_by_iso) hipLaunchKernelGGL(sb::bins_pairs_locus_kernel<false>, dim3(lgrid), dim3(256), 0, s, pa, nl);
      sb::ScanArgs sp = {n_iso, pa.pair_cnt, pa.seg_cnt, nullptr, (int64_t *)(d3 + r_poff), (int64_t *)(d3 + r_soff), part_a, part_b, (int64_t *)(d3 + r_tot) + 2};
      hipLaunchKernelGGL(sb::bins_scan_tiles_kernel<1>, dim3((unsigned)iso_tiles), dim3(256), 0, s, sp);
      hipLaunchKernelGGL(sb::bins_scan_parts_kernel, dim3(1), dim3(256), 0, s, sp, n_iso ? iso_tiles : 0);
      hipLaunchKernelGGL(sb::bins_scan_apply_kernel<1>, dim3((unsigned)iso_tiles), dim3(256), 0, s, sp);
      sb::ctx_stage_end(c, s);
      if ((x = hipGetLastError()) != hipSuccess) return x;
      x = hipMemcpyAsync((void *)totals_h, d3 + r_tot, 32, hipMemcpyDeviceToHost, s);
      if (x == hipSuccess) x = hipMemcpyAsync((void *)flags_h, d + o_flag, 8, hipMemcpyDeviceToHost, s);
      return x;
   };
   // Where nothing on the host has to come between (the usual case: single-pass kernels, hit -> bin not asked for), the
   // middle is launched before the bin counts are back; should the counts then call for a fix-up -- a locus for the big
   // table, fractional masses -- the fix-up runs and the middle is launched again over its result.
   const bool speculative = single_pass && !d_hit_bin;
   if (speculative) SB_TRY(launch_middle());
   SB_TRY(hipEventSynchronize(ev_rows));
   int32_t flags = flags_h[0];
   bool fixed_up = false;
   {
      // loci the middle table could not hold: again, with the big one
      std::vector<int32_t> redo;
      for (int64_t l = 0; l < nl; ++l)
         if (nb[(size_t)l] < 0) redo.push_back((int32_t)l);
      if (!redo.empty()) {
         fixed_up = true;
         SB_TRY(hipStreamSynchronize(s));
         SB_TRY(hipMemcpyAsync(d + o_order, redo.data(), redo.size() * 4, hipMemcpyHostToDevice, s));
         a.n_loci = (int64_t)redo.size();
         a.loci = (const int32_t *)(d + o_order);
         if (single_pass) { // these loci's entries were never zeroed (the other loci's stand as they are)
            hipLaunchKernelGGL(sb::bins_zero_loci_kernel, dim3((unsigned)std::min<int64_t>((int64_t)redo.size(), cap)), dim3(256), 0, s, a);
            SB_TRY(hipGetLastError());
         }
         hipLaunchKernelGGL((sb::bins_locus_kernel<sb::kBinsSlotsBig, sb::kBinsMaxBig, sb::kBinsThreadsBig>),
                            dim3((unsigned)std::min<int64_t>((int64_t)redo.size(), cap)), dim3(sb::kBinsThreadsBig), 0, s, a);
         SB_TRY(hipGetLastError());
         SB_TRY(fetch_rows());
         SB_TRY(hipEventSynchronize(ev_rows));
         flags = flags_h[0];
      }
   }
   // hit -> bin: the single-pass kernels leave it out; it is made where somebody reads it (the caller's hit_bin, the
   // ordered masses below).  Loci the big table served have theirs already.
   if (single_pass && (d_hit_bin || flags == sb::kBinsFractional) && nh) {
      std::vector<int32_t> todo;
      for (int64_t l = 0; l < nl; ++l)
         if (nb[(size_t)l] <= sb::kBinsMaxMid) todo.push_back((int32_t)l);
      if (!todo.empty()) {
         SB_TRY(hipStreamSynchronize(s));
         SB_TRY(hipMemcpyAsync(d + o_order, todo.data(), todo.size() * 4, hipMemcpyHostToDevice, s));
         a.n_loci = (int64_t)todo.size();
         a.loci = (const int32_t *)(d + o_order);
         hipLaunchKernelGGL((sb::bins_assign_kernel<sb::kBinsSlotsMid, sb::kBinsThreadsMid>),
                            dim3((unsigned)std::min<int64_t>((int64_t)todo.size(), cap * 4)), dim3(sb::kBinsThreadsMid), 0, s, a);
         SB_TRY(hipGetLastError());
         SB_TRY(hipStreamSynchronize(s)); // (`todo` leaves scope)
      }
   }
   if (flags == sb::kBinsFractional) {
      // fractional masses (multi-mapped reads): the bins stand, their masses are summed again in float, in the
      // order of the reference's std::set (bins_device.h)
      fixed_up = true;
      std::vector<int32_t> all((size_t)nl);
      for (int64_t l = 0; l < nl; ++l) all[(size_t)l] = (int32_t)l;
      SB_TRY(hipStreamSynchronize(s));
      SB_TRY(hipMemcpyAsync(d + o_order, all.data(), (size_t)nl * 4, hipMemcpyHostToDevice, s));
      SB_TRY(hipMemsetAsync(d + o_flag, 0, 4, s));
      a.n_loci = nl;
      a.loci = (const int32_t *)(d + o_order);
      hipLaunchKernelGGL(sb::bins_ordered_mass_kernel, dim3((unsigned)std::min<int64_t>(nl, cap * 4)), dim3(256), 0, s, a);
      SB_TRY(hipGetLastError());
      SB_TRY(hipMemcpyAsync((void *)flags_h, d + o_flag, 8, hipMemcpyDeviceToHost, s));
      SB_TRY(hipStreamSynchronize(s));
      flags = flags_h[0];
   }
   if (flags) {
      (void)hipStreamSynchronize(s);
      std::string why = "sbgpu_bins_create_device: not covered by the device form:";
      if (flags & sb::kBinsUnsorted) why += " hits of a locus are not sorted by (left, right);";
      if (flags & sb::kBinsFractional) why += " fractional hit masses next to another obstacle;";
      if (flags & sb::kBinsTableFull) why += " a locus has more bins than the LDS table holds;";
      if (flags & sb::kBinsRunTooLong) why += " fractional masses and more than 48 fragments of one bin starting at one position;";
      return bail(SBGPU_EUNSUPPORTED, why + " use sbgpu_bins_create");
   }
   stage("group kernel");
   // ---- host: the same prefix sums (the handle, the caller's plan), while the middle runs
   std::vector<int64_t> row_off((size_t)nl + 1, 0), f_off((size_t)nl + 1, 0);
   int64_t used = 0;
   for (int64_t l = 0; l < nl; ++l) {
      row_off[(size_t)l + 1] = row_off[(size_t)l] + nb[(size_t)l];
      f_off[(size_t)l + 1] = f_off[(size_t)l] + (int64_t)nb[(size_t)l] * (an->iso_off[l + 1] - an->iso_off[l]);
      used += nu[(size_t)l];
   }
   const int64_t n_bins = row_off[(size_t)nl];
   if (hooks && hooks->rows_known) hooks->rows_known(row_off.data(), f_off.data());
   // the pairs' arena, allocated by the size of the last call's while the middle runs (none yet, or too small: below)
   sb::DevicePairs dp;
   size_t dp_bytes = sb::ctx_pairs_hint(c);
   if (dp_bytes && sb::dev_take(dp_bytes, &dp.arena, &dp.capacity) != hipSuccess) dp.arena = nullptr, dp.capacity = 0;
   dp_bytes = dp.capacity;
   auto bail3 = [&](int code, const std::string &msg) {
      (void)hipStreamSynchronize(s);
      sb::dev_give(dp.arena, dp.capacity);
      return bail(code, msg);
   };
#define SB_TRY3(expr)                                                                          \
   do {                                                                                        \
      hipError_t e_ = (expr);                                                                  \
      if (e_ != hipSuccess) return bail3(SBGPU_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
   } while (0)
   if (!speculative || fixed_up) SB_TRY3(launch_middle());
   SB_TRY3(hipStreamSynchronize(s));
   flags = flags_h[1];
   if (totals_h[0] != n_bins || totals_h[1] != f_off[(size_t)nl]) return bail3(SBGPU_EHIP, "sbgpu_bins_create_device: device and host prefix sums differ");
   if (flags & sb::kBinsMassOverflow) return bail3(SBGPU_EUNSUPPORTED, "sbgpu_bins_create_device: a bin's mass reaches 2^24; use sbgpu_bins_create");
   if (flags & (sb::kPairsNotUnder | sb::kPairsForeignSegment))
      return bail3(SBGPU_EUNSUPPORTED, "sbgpu_bins_create_device: a bin does not sit under an isoform it is compatible with; sbgpu_bins_create reports the details");
   stage("pack + pairs count");
   dp.any_wide = (flags & sb::kPairsWide) != 0;
   dp.n_pairs = totals_h[2];
   dp.n_pair_segs = totals_h[3];
   size_t off4 = 0;
   dp.o_seg_off = off4; off4 += up(((size_t)dp.n_pairs + 1) * 8);
   dp.o_out_index = off4; off4 += up(((size_t)dp.n_pairs + 1) * 8);
   dp.o_seg_lens = off4; off4 += up(((size_t)dp.n_pair_segs + 1) * 4);
   dp.o_mask = off4; off4 += up(((size_t)dp.n_pairs + 1) * 4);
   dp.o_iso_len = off4; off4 += up(((size_t)dp.n_pairs + 1) * 4);
   if (off4 > dp_bytes) {
      sb::dev_give(dp.arena, dp.capacity);
      dp.arena = nullptr;
      e = sb::dev_take(off4, &dp.arena, &dp.capacity);
      if (e != hipSuccess) return bail3(e == hipErrorOutOfMemory ? SBGPU_ENOMEM : SBGPU_EHIP, std::string("hipMalloc: ") + hipGetErrorString(e));
   }
   sb::ctx_set_pairs_hint(c, off4);
   pa.pair_seg_off = (int64_t *)(dp.arena + dp.o_seg_off);
   pa.pair_seg_lens = (uint32_t *)(dp.arena + dp.o_seg_lens);
   pa.pair_mask = (uint32_t *)(dp.arena + dp.o_mask);
   pa.pair_iso_len = (int32_t *)(dp.arena + dp.o_iso_len);
   pa.pair_out_index = (int64_t *)(dp.arena + dp.o_out_index);
   sb::ctx_stage_begin(c, "bins_pairs_kernel<fill>", s);
   if (pairs_by_iso || any_wide_locus) hipLaunchKernelGGL(sb::bins_pairs_kernel<true>, dim3(pgrid), dim3(256), 0, s, pa);
   if (!pairs_by_iso) hipLaunchKernelGGL(sb::bins_pairs_locus_kernel<true>, dim3(lgrid), dim3(256), 0, s, pa, nl);
   sb::ctx_stage_end(c, s);
   SB_TRY3(hipGetLastError());
   stage("pairs fill launch");
   // the caller's kernels behind the grouping (bin weights, EM) go into the stream now
   if (hooks && hooks->after_pairs) {
      const sb::DeviceGrouping g = {row_off.data(), f_off.data(), n_bins, f_off[(size_t)nl], (const int32_t *)(d2 + p_cnt), &dp};
      const int rc_after = hooks->after_pairs(g);
      if (rc_after != SBGPU_OK) {
         (void)hipStreamSynchronize(s);
         sb::dev_give(dp.arena, dp.capacity);
         return rc_after;
      }
   }
#undef SB_TRY3
#undef SB_TRY
   const sb::DeviceBinArrays dba = {d2, d2_cap, p_cnt, p_key, p_cmp};
   const int rc = sb::bins_from_groups(an, compat_words, key_words, row_off.data(), dba, used, &dp, &iso_pre->len, out);
   if (rc != SBGPU_OK) {
      (void)hipStreamSynchronize(s);
      sb::dev_give(dp.arena, dp.capacity);
   } else {
      d2 = nullptr; // the handle's now (d2_guard lets go)
   }
   stage("handle");
   return rc;
}
} // namespace sb

extern "C" {

int sbgpu_bins_create_device(sbgpu_ctx_t *c, const sbgpu_annotation_t *an, const sbgpu_hits_t *dh, const float *d_mass,
                             const int64_t *locus_hit_off, int32_t compat_words, int32_t key_words, const uint32_t *d_compat,
                             const uint32_t *d_key, int64_t *d_hit_bin, void *stream, sbgpu_bins_t **out)
{
   return sb::bins_create_device_impl(c, an, dh, d_mass, locus_hit_off, compat_words, key_words, d_compat, d_key, d_hit_bin, stream,
                                      nullptr, out, nullptr, nullptr);
}

} // extern "C"
