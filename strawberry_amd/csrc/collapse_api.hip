// strawberry_amd/csrc/collapse_api.hip -- sbgpu_collapse_pairs_device (include/sbgpu.h): read pairs resident in
// HBM -> unique hits resident in HBM (HitCluster::collapseAndFilterHits + Contig(PairedHit),
// /root/reference/src/alignments.cpp:656-703, src/contig.cpp:216-267).  Kernels: collapse_device.h.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include <rocprim/rocprim.hpp>

#include "../../include/sbgpu.h"
#include "api_internal.h"
#include "collapse_device.h"
#include "collapse_flat.h"

using sb::api_fail;

struct sbgpu_uniq_dev {
   int device = 0;
   char *arena = nullptr; // the unique hits (sbgpu_hits_t layout) + their masses (sb::dev_take'n)
   size_t arena_cap = 0;
   int64_t n_loci = 0, n_hits = 0, n_feat = 0, n_filtered = 0, n_rejected = 0, total_mapped = 0;
   int32_t *d_hit_locus = nullptr;
   int64_t *d_feat_off = nullptr;
   uint8_t *d_feat_code = nullptr;
   uint32_t *d_feat_left = nullptr, *d_feat_right = nullptr;
   float *d_mass = nullptr;
   std::vector<int64_t> locus_hit_off; // host
   std::vector<double> cluster_mass;   // host
};

namespace {
size_t up256(size_t b) { return (b + 255) & ~(size_t)255; }
} // namespace

// ---- the flat form (collapse_flat.h): every cluster of the call at once
static int collapse_flat(sbgpu_ctx_t *c, int64_t n_loci, const sbgpu_pairs_t *dp, const int64_t *locus_pair_off, hipStream_t s,
                         sbgpu_uniq_dev *U)
{
   const int64_t np = dp->n_pairs;
   if (np >= ((int64_t)1 << 31) - 2) return api_fail(SBGPU_EUNSUPPORTED, "sbgpu_collapse_pairs_device: more than 2^31 read pairs in one call; split the call or use sbgpu_collapse_pairs_host");
   char *w = nullptr;
   size_t w_cap = 0;
   auto bail = [&](int code, const std::string &msg) {
      (void)hipStreamSynchronize(s);
      sb::dev_give(w, w_cap);
      return api_fail(code, msg);
   };
#define SB_TRY(expr)                                                                                     \
   do {                                                                                                  \
      hipError_t e_ = (expr);                                                                            \
      if (e_ != hipSuccess) return bail(e_ == hipErrorOutOfMemory ? SBGPU_ENOMEM : SBGPU_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
   } while (0)
   SB_TRY(hipSetDevice(U->device));
   const size_t n = (size_t)np, n1 = n + 1, nl1 = (size_t)n_loci + 1;
   unsigned locus_bits = 1;
   while (((int64_t)1 << locus_bits) < n_loci) ++locus_bits;
   // temporary storage of the library calls: the largest of them
   size_t tmp_bytes = 0;
   {
      size_t b = 0;
      (void)rocprim::radix_sort_pairs(nullptr, b, (const uint32_t *)nullptr, (uint32_t *)nullptr, rocprim::counting_iterator<int32_t>(0), (int32_t *)nullptr, n, 0, 32, s);
      tmp_bytes = std::max(tmp_bytes, b);
      (void)rocprim::radix_sort_pairs(nullptr, b, (const unsigned long long *)nullptr, (unsigned long long *)nullptr, (const int32_t *)nullptr, (int32_t *)nullptr, n, 0, 32 + locus_bits, s);
      tmp_bytes = std::max(tmp_bytes, b);
      (void)rocprim::radix_sort_pairs(nullptr, b, (const unsigned long long *)nullptr, (unsigned long long *)nullptr, rocprim::counting_iterator<int32_t>(0), (int32_t *)nullptr, n, 0, 64, s);
      tmp_bytes = std::max(tmp_bytes, b);
      (void)rocprim::exclusive_scan(nullptr, b, (const unsigned long long *)nullptr, (unsigned long long *)nullptr, 0ull, nl1, rocprim::plus<unsigned long long>(), s);
      tmp_bytes = std::max(tmp_bytes, b);
      (void)rocprim::inclusive_scan(nullptr, b, (const int32_t *)nullptr, (int32_t *)nullptr, n, rocprim::maximum<int32_t>(), s);
      tmp_bytes = std::max(tmp_bytes, b);
      (void)rocprim::inclusive_scan(nullptr, b, rocprim::make_transform_iterator((const uint8_t *)nullptr, [] __device__(uint8_t v) { return (int32_t)v; }), (int32_t *)nullptr, n, rocprim::plus<int32_t>(), s);
      tmp_bytes = std::max(tmp_bytes, b);
      (void)rocprim::exclusive_scan(nullptr, b, rocprim::make_transform_iterator((const int32_t *)nullptr, [] __device__(int32_t v) { return (int64_t)v; }), (int64_t *)nullptr, (int64_t)0, n1, rocprim::plus<int64_t>(), s);
      tmp_bytes = std::max(tmp_bytes, b);
   }
   size_t off = 0;
   auto take = [&](size_t bytes) {
      const size_t o = off;
      off += up256(bytes ? bytes : 8);
      return o;
   };
   const size_t o_poff = take(nl1 * 8), o_kr = take(n * 4), o_kr2 = take(n * 4), o_khi = take(n * 8), o_sl = take(n * 4), o_sr = take(n * 4);
   const size_t o_sum = take(nl1 * 8), o_sq = take(nl1 * 8), o_nm = take(nl1 * 4), o_clf = take(nl1 * 4), o_cm = take(nl1 * 8), o_mean = take(nl1 * 8), o_sd = take(nl1 * 8);
   const size_t o_perm1 = take(n * 4), o_key2 = take(n * 8), o_key2s = take(n * 8), o_order = take(n * 4);
   const size_t o_skip = take(n), o_head = take(n), o_kept = take(n * 4), o_last = take(n * 4), o_nfeat = take(n1 * 4), o_ishit = take(n1 * 4);
   const size_t o_gid = take(n * 4), o_hrank = take(n1 * 8), o_fbase = take(n1 * 8), o_gmass = take(n1 * 8), o_hoff = take(nl1 * 8);
   const size_t o_minl = take(nl1 * 4), o_maxl = take(nl1 * 4), o_cspan = take(nl1 * 8), o_cbase = take(nl1 * 8), o_clof = take(n * 4), o_ostat = take(64);
   const size_t o_counts = take(64), o_flag = take(64), o_tmp = take(tmp_bytes);
   SB_TRY(sb::dev_take(off, &w, &w_cap));
   // what must start at zero: the clusters' sums, the counters and flags, the entry beyond the last position of the two scan inputs
   SB_TRY(hipMemsetAsync(w + o_sum, 0, o_mean - o_sum, s)); // S1, S2, counts, flags, the clusters' masses
   SB_TRY(hipMemsetAsync(w + o_gmass, 0, n1 * 8, s));
   // SBGPU_COLLAPSE_FORCE_SEQ=1 (tests): every cluster takes the running sums in the reference's order, as if it had been marked
   if (const char *fs = std::getenv("SBGPU_COLLAPSE_FORCE_SEQ"); fs && std::atoi(fs) != 0)
      SB_TRY(hipMemsetD32Async((hipDeviceptr_t)(w + o_clf), sb::kNeedSeqMass | sb::kNeedSeqSd, nl1, s));
   SB_TRY(hipMemsetAsync(w + o_counts, 0, o_tmp - o_counts, s));
   SB_TRY(hipMemsetAsync(w + o_minl, 0xFF, nl1 * 4, s)); // (the clusters' leftmost left end starts at the top)
   SB_TRY(hipMemsetAsync(w + o_maxl, 0, nl1 * 4, s));
   SB_TRY(hipMemsetAsync(w + o_ostat, 0, 64, s));
   SB_TRY(hipMemsetAsync(w + o_nfeat + n * 4, 0, 4, s));
   SB_TRY(hipMemsetAsync(w + o_ishit + n * 4, 0, 4, s));
   SB_TRY(hipMemcpyAsync(w + o_poff, locus_pair_off, nl1 * 8, hipMemcpyHostToDevice, s));
   sb::FlatCollapseArgs f = {};
   sb::CollapseArgs &a = f.a;
   a.n_loci = n_loci;
   a.locus_pair_off = (const int64_t *)(w + o_poff);
   a.pair_mass = dp->pair_mass;
   a.left_off = dp->left_off, a.right_off = dp->right_off;
   a.left_code = dp->left_code, a.right_code = dp->right_code;
   a.left_left = dp->left_left, a.left_right = dp->left_right;
   a.right_left = dp->right_left, a.right_right = dp->right_right;
   a.cluster_mass = (double *)(w + o_cm);
   a.flags = (int32_t *)(w + o_flag);
   f.n_pairs = np;
   f.key_right = (uint32_t *)(w + o_kr);
   f.key_hi = (unsigned long long *)(w + o_khi);
   f.span_l = (int32_t *)(w + o_sl), f.span_r = (int32_t *)(w + o_sr);
   f.span_sum = (unsigned long long *)(w + o_sum);
   f.span_sq = (unsigned long long *)(w + o_sq);
   f.n_mates = (int32_t *)(w + o_nm);
   f.cl_flags = (int32_t *)(w + o_clf);
   f.mean = (double *)(w + o_mean), f.sd5 = (double *)(w + o_sd);
   f.perm1 = (const int32_t *)(w + o_perm1);
   f.key2 = (unsigned long long *)(w + o_key2);
   f.key2s = (const unsigned long long *)(w + o_key2s);
   f.order = (const int32_t *)(w + o_order);
   f.skip = (uint8_t *)(w + o_skip), f.head = (uint8_t *)(w + o_head);
   f.kept_pos = (int32_t *)(w + o_kept);
   f.last_kept = (const int32_t *)(w + o_last);
   f.nfeat = (int32_t *)(w + o_nfeat), f.is_hit = (int32_t *)(w + o_ishit);
   f.gid = (const int32_t *)(w + o_gid);
   f.hit_rank = (const int64_t *)(w + o_hrank), f.feat_base = (const int64_t *)(w + o_fbase);
   f.gmass = (double *)(w + o_gmass);
   f.locus_hit_off = (int64_t *)(w + o_hoff);
   f.counts = (unsigned long long *)(w + o_counts);
   f.cl_minl = (uint32_t *)(w + o_minl), f.cl_maxl = (uint32_t *)(w + o_maxl);
   f.cl_span = (unsigned long long *)(w + o_cspan), f.cl_base = (const unsigned long long *)(w + o_cbase);
   f.order_stats = (unsigned long long *)(w + o_ostat);
   f.cluster_of = (int32_t *)(w + o_clof);
   const unsigned gp = (unsigned)((n + 255) / 256);
   const unsigned gw = (unsigned)std::min<int64_t>(n_loci + 1, (int64_t)sb::ctx_cu_count(c) * 64);
   void *tmp = w + o_tmp;
   size_t tb = tmp_bytes;
   hipLaunchKernelGGL(sb::flat_keys_kernel, dim3(gp), dim3(256), 0, s, f);
   SB_TRY(hipGetLastError());
   // The order of std::sort on (left end, right end), ties in input order.  ONE stable sort where the key fits 64 bits (it does
   // unless the clusters' ranges of left ends add up to more than 2^(64 - bits of the longest pair)): a cluster's left ends
   // relative to its leftmost, the clusters' ranges laid end to end, the pair's length below (collapse_flat.h).  What
   // the key's width is, the host must know: one 24-byte read-back after the keys kernel.  Otherwise (or with
   // SBGPU_COLLAPSE_TWO_SORTS=1: tests) round 4's two sorts, least significant key first: by the right end, then by
   // (cluster << 32 | left end).
   hipLaunchKernelGGL(sb::flat_cluster_spans_kernel, dim3((unsigned)((nl1 + 255) / 256)), dim3(256), 0, s, f);
   SB_TRY(rocprim::exclusive_scan(tmp, tb, (const unsigned long long *)f.cl_span, (unsigned long long *)(w + o_cbase), 0ull, nl1, rocprim::plus<unsigned long long>(), s));
   unsigned long long x_total = 0, ostat[2] = {0, 0};
   SB_TRY(hipMemcpyAsync(&x_total, w + o_cbase + (size_t)n_loci * 8, 8, hipMemcpyDeviceToHost, s));
   SB_TRY(hipMemcpyAsync(ostat, w + o_ostat, 16, hipMemcpyDeviceToHost, s));
   SB_TRY(hipStreamSynchronize(s));
   auto bits_of = [](unsigned long long v) {
      unsigned b = 0;
      while (b < 64 && (v >> b)) ++b;
      return b;
   };
   const unsigned span_bits = bits_of(ostat[0]), x_bits = bits_of(x_total);
   const bool two_sorts_env = std::getenv("SBGPU_COLLAPSE_TWO_SORTS") && std::atoi(std::getenv("SBGPU_COLLAPSE_TWO_SORTS")) != 0;
   const bool one_sort = !two_sorts_env && !ostat[1] && span_bits + x_bits <= 64;
   f.span_bits = (int)span_bits;
   tb = tmp_bytes;
   if (one_sort) {
      hipLaunchKernelGGL(sb::flat_rekey_kernel, dim3(gp), dim3(256), 0, s, f);
      SB_TRY(rocprim::radix_sort_pairs(tmp, tb, (const unsigned long long *)f.key2, (unsigned long long *)(w + o_key2s), rocprim::counting_iterator<int32_t>(0), (int32_t *)(w + o_order), n, 0, std::max(1u, span_bits + x_bits), s));
      hipLaunchKernelGGL(sb::flat_cluster_of_kernel<false>, dim3(gp), dim3(256), 0, s, f);
   } else {
      SB_TRY(rocprim::radix_sort_pairs(tmp, tb, (const uint32_t *)f.key_right, (uint32_t *)(w + o_kr2), rocprim::counting_iterator<int32_t>(0), (int32_t *)(w + o_perm1), n, 0, 32, s));
      hipLaunchKernelGGL(sb::flat_gather_kernel, dim3(gp), dim3(256), 0, s, f);
      tb = tmp_bytes;
      SB_TRY(rocprim::radix_sort_pairs(tmp, tb, (const unsigned long long *)f.key2, (unsigned long long *)(w + o_key2s), f.perm1, (int32_t *)(w + o_order), n, 0, 32 + locus_bits, s));
      hipLaunchKernelGGL(sb::flat_cluster_of_kernel<true>, dim3(gp), dim3(256), 0, s, f);
   }
   SB_TRY(hipGetLastError());
   // the span filter from the integer moments; the clusters it marks (none, as a rule) get the running sum and their flags again
   hipLaunchKernelGGL(sb::flat_flags_kernel<1>, dim3(gp), dim3(256), 0, s, f);
   hipLaunchKernelGGL(sb::flat_sd_kernel, dim3(gw), dim3(64), 0, s, f);
   hipLaunchKernelGGL(sb::flat_flags_kernel<2>, dim3(gp), dim3(256), 0, s, f);
   SB_TRY(hipGetLastError());
   tb = tmp_bytes;
   SB_TRY(rocprim::inclusive_scan(tmp, tb, (const int32_t *)f.kept_pos, (int32_t *)(w + o_last), n, rocprim::maximum<int32_t>(), s));
   hipLaunchKernelGGL(sb::flat_heads_kernel, dim3(sb::xcd_grid(gp)), dim3(256), 0, s, f);
   hipLaunchKernelGGL(sb::flat_heads_long_kernel, dim3(256), dim3(64), 0, s, f); // (returns at once unless a mate has more than 24 features)
   SB_TRY(hipGetLastError());
   tb = tmp_bytes;
   SB_TRY(rocprim::inclusive_scan(tmp, tb, rocprim::make_transform_iterator((const uint8_t *)f.head, [] __device__(uint8_t v) { return (int32_t)v; }), (int32_t *)(w + o_gid), n, rocprim::plus<int32_t>(), s));
   tb = tmp_bytes;
   SB_TRY(rocprim::exclusive_scan(tmp, tb, rocprim::make_transform_iterator((const int32_t *)f.is_hit, [] __device__(int32_t v) { return (int64_t)v; }), (int64_t *)(w + o_hrank), (int64_t)0, n1, rocprim::plus<int64_t>(), s));
   tb = tmp_bytes;
   SB_TRY(rocprim::exclusive_scan(tmp, tb, rocprim::make_transform_iterator((const int32_t *)f.nfeat, [] __device__(int32_t v) { return (int64_t)v; }), (int64_t *)(w + o_fbase), (int64_t)0, n1, rocprim::plus<int64_t>(), s));
   hipLaunchKernelGGL(sb::flat_mass_any_order_kernel, dim3(sb::xcd_grid((int64_t)((std::max(n, nl1) + 255) / 256))), dim3(256), 0, s, f);
   hipLaunchKernelGGL(sb::flat_mass_kernel, dim3(gw), dim3(64), 0, s, f); // the clusters with a mass that is no multiple of 2^-20
   SB_TRY(hipGetLastError());
   // ---- what the host needs: the flags, the totals, the clusters' first hits and masses
   int32_t flags = 0;
   unsigned long long counts[2] = {0, 0};
   int64_t n_feat = 0;
   SB_TRY(hipMemcpyAsync(&flags, w + o_flag, 4, hipMemcpyDeviceToHost, s));
   SB_TRY(hipMemcpyAsync(counts, w + o_counts, 16, hipMemcpyDeviceToHost, s));
   SB_TRY(hipMemcpyAsync(&n_feat, w + o_fbase + n * 8, 8, hipMemcpyDeviceToHost, s));
   SB_TRY(hipMemcpyAsync(U->locus_hit_off.data(), w + o_hoff, nl1 * 8, hipMemcpyDeviceToHost, s));
   SB_TRY(hipMemcpyAsync(U->cluster_mass.data(), w + o_cm, (size_t)n_loci * 8, hipMemcpyDeviceToHost, s));
   SB_TRY(hipStreamSynchronize(s));
   if (flags) {
      if (flags & sb::kCollapseNoMates) return bail(SBGPU_EINVAL, "sbgpu_collapse_pairs_device: a pair without mates");
      return bail(SBGPU_EUNSUPPORTED, "sbgpu_collapse_pairs_device: not covered by the device form: a mate has more than 512 features; use sbgpu_collapse_pairs_host");
   }
   U->n_filtered = (int64_t)counts[0];
   U->n_rejected = (int64_t)counts[1];
   for (int64_t l = 0; l < n_loci; ++l) U->total_mapped += (int64_t)(int)U->cluster_mass[(size_t)l]; // src/alignments.cpp:1372
   U->n_hits = U->locus_hit_off[(size_t)n_loci];
   U->n_feat = n_feat;
   // ---- the unique hits' own arena
   const size_t nh1 = (size_t)U->n_hits + 1, nfe1 = (size_t)U->n_feat + 1;
   size_t t = 0;
   const size_t u_off = t; t += up256(nh1 * 8);
   const size_t u_loc = t; t += up256(nh1 * 4);
   const size_t u_mass = t; t += up256(nh1 * 4);
   const size_t u_left = t; t += up256(nfe1 * 4);
   const size_t u_right = t; t += up256(nfe1 * 4);
   const size_t u_code = t; t += up256(nfe1);
   SB_TRY(sb::dev_take(t, &U->arena, &U->arena_cap));
   U->d_feat_off = (int64_t *)(U->arena + u_off);
   U->d_hit_locus = (int32_t *)(U->arena + u_loc);
   U->d_mass = (float *)(U->arena + u_mass);
   U->d_feat_left = (uint32_t *)(U->arena + u_left);
   U->d_feat_right = (uint32_t *)(U->arena + u_right);
   U->d_feat_code = (uint8_t *)(U->arena + u_code);
   SB_TRY(hipMemcpyAsync(U->d_feat_off + U->n_hits, &U->n_feat, 8, hipMemcpyHostToDevice, s));
   a.hit_locus = U->d_hit_locus;
   a.feat_off = U->d_feat_off;
   a.feat_code = U->d_feat_code;
   a.feat_left = U->d_feat_left;
   a.feat_right = U->d_feat_right;
   a.hit_mass = U->d_mass;
   hipLaunchKernelGGL(sb::flat_fill_kernel, dim3(sb::xcd_grid(gp)), dim3(256), 0, s, f);
   hipLaunchKernelGGL(sb::flat_fill_long_kernel, dim3(256), dim3(64), 0, s, f);
   SB_TRY(hipGetLastError());
   SB_TRY(hipStreamSynchronize(s)); // the scratch goes back to the pool
#undef SB_TRY
   sb::dev_give(w, w_cap);
   return SBGPU_OK;
}

extern "C" {

void sbgpu_uniq_dev_destroy(sbgpu_uniq_dev_t *u)
{
   if (!u) return;
   sb::dev_give(u->arena, u->arena_cap); // (waits for the device the arena lives on, as hipFree did)
   delete u;
}

int sbgpu_collapse_pairs_device(sbgpu_ctx_t *c, int64_t n_loci, const sbgpu_pairs_t *dp, const int64_t *locus_pair_off,
                                void *stream, sbgpu_uniq_dev_t **out)
{
   if (!c || !dp || !out || !locus_pair_off || n_loci < 0) return api_fail(SBGPU_EINVAL, "sbgpu_collapse_pairs_device: bad argument");
   *out = nullptr;
   const int64_t np = dp->n_pairs;
   if (np < 0 || locus_pair_off[0] != 0 || locus_pair_off[n_loci] != np)
      return api_fail(SBGPU_EINVAL, "sbgpu_collapse_pairs_device: locus_pair_off does not cover the pairs");
   if (np && (!dp->pair_mass || !dp->left_off || !dp->right_off)) return api_fail(SBGPU_EINVAL, "sbgpu_collapse_pairs_device: null array");
   for (int64_t l = 0; l < n_loci; ++l)
      if (locus_pair_off[l + 1] < locus_pair_off[l]) return api_fail(SBGPU_EINVAL, "sbgpu_collapse_pairs_device: locus_pair_off must ascend");
   hipStream_t s = (hipStream_t)stream;
   sbgpu_uniq_dev *U = new (std::nothrow) sbgpu_uniq_dev();
   if (!U) return api_fail(SBGPU_ENOMEM, "sbgpu_collapse_pairs_device: out of host memory");
   U->device = sb::ctx_device(c);
   U->n_loci = n_loci;
   U->locus_hit_off.assign((size_t)n_loci + 1, 0);
   U->cluster_mass.assign((size_t)n_loci, 0.0);
   if (np == 0 || n_loci == 0) {
      *out = U;
      return SBGPU_OK;
   }
   // every cluster of the call at once (collapse_flat.h)
   const int rc = collapse_flat(c, n_loci, dp, locus_pair_off, s, U);
   if (rc != SBGPU_OK) {
      sbgpu_uniq_dev_destroy(U);
      return rc;
   }
   *out = U;
   return SBGPU_OK;
}

int sbgpu_uniq_dev_info(const sbgpu_uniq_dev_t *u, int64_t info[8])
{
   if (!u || !info) return api_fail(SBGPU_EINVAL, "sbgpu_uniq_dev_info: null argument");
   info[0] = u->n_hits;
   info[1] = u->n_feat;
   info[2] = u->n_filtered;
   info[3] = u->n_rejected;
   info[4] = u->total_mapped;
   info[5] = u->n_loci;
   info[6] = info[7] = 0;
   return SBGPU_OK;
}

int sbgpu_uniq_dev_hits(const sbgpu_uniq_dev_t *u, sbgpu_hits_t *d_hits, const float **d_hit_mass, const int64_t **locus_hit_off)
{
   if (!u || !d_hits) return api_fail(SBGPU_EINVAL, "sbgpu_uniq_dev_hits: null argument");
   d_hits->n_hits = u->n_hits;
   d_hits->hit_locus = u->d_hit_locus;
   d_hits->feat_off = u->d_feat_off;
   d_hits->feat_code = u->d_feat_code;
   d_hits->feat_left = u->d_feat_left;
   d_hits->feat_right = u->d_feat_right;
   if (d_hit_mass) *d_hit_mass = u->d_mass;
   if (locus_hit_off) *locus_hit_off = u->locus_hit_off.data();
   return SBGPU_OK;
}

int sbgpu_uniq_dev_export(const sbgpu_uniq_dev_t *u, int32_t *hit_locus, int64_t *feat_off, uint8_t *feat_code, uint32_t *feat_left,
                          uint32_t *feat_right, float *hit_mass, double *cluster_mass)
{
   if (!u) return api_fail(SBGPU_EINVAL, "sbgpu_uniq_dev_export: null argument");
   hipError_t e = hipSetDevice(u->device);
   auto get = [&](void *dst, const void *src, size_t bytes) {
      if (e == hipSuccess && dst && bytes) e = hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost);
   };
   if (u->arena) {
      get(hit_locus, u->d_hit_locus, (size_t)u->n_hits * 4);
      get(feat_off, u->d_feat_off, (size_t)(u->n_hits + 1) * 8);
      get(feat_code, u->d_feat_code, (size_t)u->n_feat);
      get(feat_left, u->d_feat_left, (size_t)u->n_feat * 4);
      get(feat_right, u->d_feat_right, (size_t)u->n_feat * 4);
      get(hit_mass, u->d_mass, (size_t)u->n_hits * 4);
   } else if (feat_off) {
      feat_off[0] = 0;
   }
   if (cluster_mass && !u->cluster_mass.empty()) std::memcpy(cluster_mass, u->cluster_mass.data(), u->cluster_mass.size() * 8);
   if (e != hipSuccess) return api_fail(SBGPU_EHIP, std::string("sbgpu_uniq_dev_export: ") + hipGetErrorString(e));
   return SBGPU_OK;
}

} // extern "C"
