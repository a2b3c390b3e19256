// strawberry_amd/csrc/chain_api.hip -- sbgpu_quantify_host (include/sbgpu.h): the path's entry points
// chained on one stream with the intermediate results resident in HBM.  No torch, no Python: this is
// what a C / C++ driver calls (include/sbgpu_host.hpp wraps it).
#include <hip/hip_runtime.h>

#include <climits>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <functional>
#include <new>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

#include "../../include/sbgpu.h"
#include "api_internal.h"
#include "fraglen_device.h"

using sb::api_fail;

namespace {

// the empirical insert-size law of a handle: kept alive with the handle through its weights' tail
struct DeviceBuf {
   char *p = nullptr; // the context's scratch (sb::ctx_scratch): not freed here
};

size_t up256(size_t b) { return (b + 255) & ~(size_t)255; }

// what sbgpu_quantify_resident adds to the chain: the epilogue behind the EM, with the path's collectives inside the call
struct ResidentOpts {
   int64_t mapped_reads;                   // this rank's part of Sample::total_mapped_reads()
   const sbgpu_abundance_params_t *params; // total_mapped_reads and insert_mean are filled in here
   sbgpu_comm_t *comm;                     // nullptr: a world of one
   sbgpu_abundances_t *out;
};

} // namespace

// `dev_hit_off` != nullptr: the hits (hits->..., hit_mass) are DEVICE arrays already, grouped by locus as
// dev_hit_off[n_loci + 1] (host) says -- sbgpu_quantify_device; else host arrays -- sbgpu_quantify_host.
static int quantify_impl(sbgpu_ctx_t *c, const sbgpu_annotation_t *an, const sbgpu_hits_t *hits, const float *hit_mass,
                         const int64_t *dev_hit_off, const sbgpu_insert_t *insert, int32_t read_len, int32_t long_read,
                         double *theta_out, int32_t *status_out, int32_t *iters_out, uint32_t *compat_out,
                         sbgpu_insert_t *insert_used, sbgpu_bins_t **bins_out, const ResidentOpts *ro = nullptr)
{
   const bool on_dev = dev_hit_off != nullptr;
   if (!c || !an || !hits || !bins_out || (!ro && (!theta_out || !status_out || !iters_out)))
      return api_fail(SBGPU_EINVAL, "sbgpu_quantify_host: null argument");
   *bins_out = nullptr;
   const int64_t nl = an->n_loci, nh = hits->n_hits;
   if (nl < 1 || nh < 0) return api_fail(SBGPU_EINVAL, "sbgpu_quantify_host: bad counts");
   if (!an->iso_off || !an->exon_off || !an->seg_off || (nh && (!hits->hit_locus || !hits->feat_off || !hit_mass)))
      return api_fail(SBGPU_EINVAL, "sbgpu_quantify_host: null array");
   if (!insert && !insert_used) return api_fail(SBGPU_EINVAL, "sbgpu_quantify_host: insert_used is needed when no insert-size law is given");
   if (on_dev && compat_out) return api_fail(SBGPU_EINVAL, "sbgpu_quantify_device: returns no compat words");
   const int64_t n_iso = an->iso_off[nl], n_exon = an->exon_off[n_iso], n_seg = an->seg_off[nl];
   int64_t n_feat = 0;
   if (nh && !on_dev) n_feat = hits->feat_off[nh];
   // an annotation kept resident (sbgpu_annotation_pin) and given again: its device copies and tables are used as they are
   const sb::ResidentAnnotation *res = sb::ctx_resident_annotation(c);
   if (res && !res->matches(an)) res = nullptr;
   int64_t max_iso = 1, max_seg = 1;
   if (res) {
      max_iso = res->max_iso, max_seg = res->max_seg;
   } else {
      for (int64_t l = 0; l < nl; ++l) {
         max_iso = std::max(max_iso, an->iso_off[l + 1] - an->iso_off[l]);
         max_seg = std::max(max_seg, an->seg_off[l + 1] - an->seg_off[l]);
      }
   }
   const int32_t cw = (int32_t)((max_iso + 31) / 32), kw = (int32_t)((max_seg + 31) / 32);
   // hits grouped by locus?  (the device grouping needs it; the host one does not)
   bool grouped = true;
   std::vector<int64_t> locus_hit_off((size_t)nl + 1, 0);
   if (on_dev) {
      locus_hit_off.assign(dev_hit_off, dev_hit_off + nl + 1);
      if (locus_hit_off[0] != 0 || locus_hit_off[(size_t)nl] != nh) return api_fail(SBGPU_EINVAL, "sbgpu_quantify_device: locus_hit_off does not cover the hits");
   }
   hipStream_t s = sb::ctx_stream(c);
   const char *timing_env = std::getenv("SBGPU_HOST_TIMING");
   const bool timing = timing_env != nullptr, timing_sync = timing && std::atoi(timing_env) != 2; // diagnostic: stage times on stderr; =2: host clock only, no synchronisation
   auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
   double t_stage = now();
   auto stage = [&](const char *name) {
      if (timing) {
         if (timing_sync) (void)hipStreamSynchronize(s);
         const double t = now();
         std::fprintf(stderr, "sbgpu_quantify_host: %-18s %.2f ms\n", name, (t - t_stage) * 1e3);
         t_stage = t;
      }
   };
   if (!on_dev && nh) {
      // every hit's locus in range, and the hits grouped by locus?  A pass over 4 bytes per hit (0.7 GB at 1.8e8 hits): split
      // over a few host threads; the loci's offsets are then binary searches (grouped) or a counting pass (not grouped)
      const int32_t *hl = hits->hit_locus;
      const int T = (int)std::max<int64_t>(1, std::min<int64_t>({(int64_t)16, (int64_t)std::thread::hardware_concurrency(), nh / (1 << 20) + 1}));
      std::vector<int> bad((size_t)T, 0), unsorted((size_t)T, 0);
      auto scan = [&](int t) {
         const int64_t h0 = nh * t / T, h1 = nh * (t + 1) / T;
         int b = 0, u = 0;
         int32_t prev = h0 ? hl[h0 - 1] : -1;
         for (int64_t h = h0; h < h1; ++h) {
            const int32_t l = hl[h];
            b |= (l < 0) | (l >= nl);
            u |= l < prev;
            // grouped hits: locus k's hits begin where the first locus >= k appears (the thread that sees the step writes the
            // entries; every step is seen by exactly one thread).  Garbage when the hits turn out not to be grouped: redone below.
            if (l > prev && !b && prev >= -1)
               for (int64_t k = (int64_t)prev + 1; k <= l; ++k) locus_hit_off[(size_t)k] = h;
            prev = l;
         }
         bad[(size_t)t] = b, unsorted[(size_t)t] = u;
      };
      {
         std::vector<std::thread> pool;
         try {
            for (int t = 1; t < T; ++t) pool.emplace_back(scan, t);
         } catch (const std::system_error &) { // fewer threads than hoped: the rest is done here
         }
         for (int t = (int)pool.size() + 1; t < T; ++t) scan(t);
         scan(0);
         for (std::thread &th : pool) th.join();
      }
      for (int t = 0; t < T; ++t) {
         if (bad[(size_t)t]) return api_fail(SBGPU_EINVAL, "sbgpu_quantify_host: hit_locus out of range");
         if (unsorted[(size_t)t]) grouped = false;
      }
      if (grouped) {
         for (int64_t l = (int64_t)hl[nh - 1] + 1; l <= nl; ++l) locus_hit_off[(size_t)l] = nh; // loci behind the last hit
      } else {
         std::fill(locus_hit_off.begin(), locus_hit_off.end(), 0);
         for (int64_t h = 0; h < nh; ++h) ++locus_hit_off[(size_t)hl[h] + 1];
         for (int64_t l = 0; l < nl; ++l) locus_hit_off[(size_t)l + 1] += locus_hit_off[(size_t)l];
      }
   }

   stage("validate hits");
   // ---- inputs: one arena, one copy per array
   struct Part {
      const void *src;
      size_t bytes, off;
   };
   const size_t nh1 = (size_t)std::max<int64_t>(nh, 1);
   Part parts[] = {
      {an->iso_off, (size_t)(nl + 1) * 8, 0},    {an->exon_off, (size_t)(n_iso + 1) * 8, 0},
      {an->seg_off, (size_t)(nl + 1) * 8, 0},    {hits->feat_off, nh ? (size_t)(nh + 1) * 8 : 0, 0},
      {an->exon_left, (size_t)n_exon * 4, 0},    {an->exon_right, (size_t)n_exon * 4, 0},
      {an->seg_left, (size_t)n_seg * 4, 0},      {an->seg_right, (size_t)n_seg * 4, 0},
      {hits->hit_locus, (size_t)nh * 4, 0},      {hits->feat_left, (size_t)n_feat * 4, 0},
      {hits->feat_right, (size_t)n_feat * 4, 0}, {hits->feat_code, (size_t)n_feat, 0},
      {hit_mass, (size_t)nh * 4, 0},
   };
   if (on_dev)
      for (int k : {3, 8, 9, 10, 11, 12}) parts[k].bytes = 0; // the hits are in HBM already
   if (res)
      for (int k : {0, 1, 2, 4, 5, 6, 7}) parts[k].bytes = 0; // so is the annotation
   size_t total = 0;
   for (Part &p : parts) {
      p.off = total;
      total += up256(p.bytes ? p.bytes : 8);
   }
   const size_t o_compat = total; total += up256(nh1 * 4 * (size_t)cw);
   const size_t o_key = total; total += up256(nh1 * 4 * (size_t)kw);
   const size_t o_hbin = total; total += up256(8); // (hit -> bin has an arena of its own where it is made at all)
   const size_t o_span = total; total += up256(nh1 * 8);
   const size_t o_fhash = total; total += up256(nh1 * 4);
   DeviceBuf in;
   hipError_t e = sb::ctx_scratch(c, 0, total, &in.p);
   if (e != hipSuccess) return api_fail(e == hipErrorOutOfMemory ? SBGPU_ENOMEM : SBGPU_EHIP, std::string("hipMalloc: ") + hipGetErrorString(e));
#define SB_TRY(expr)                                                                                          \
   do {                                                                                                       \
      hipError_t e_ = (expr);                                                                                 \
      if (e_ != hipSuccess) return api_fail(SBGPU_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_));   \
   } while (0)
#define SB_RC(expr)                      \
   do {                                  \
      const int rc_ = (expr);            \
      if (rc_ != SBGPU_OK) return rc_;   \
   } while (0)
   // The isoforms' segment lists depend on the annotation only: a helper thread makes them (1 ms of host work for
   // 20 000 loci) while this one feeds the uploads.
   sb::IsoSegments iso_pre; // declared BEFORE the worker: on an early return the thread is joined while its output still lives
   struct Joiner {
      std::thread t;
      ~Joiner()
      {
         if (t.joinable()) t.join();
      }
   } iso_worker;
   if (grouped && nh && !res) {
      try {
         iso_worker.t = std::thread([&]() { sb::iso_segments(an, &iso_pre); });
      } catch (const std::system_error &) { // no thread to be had: make them here
         sb::iso_segments(an, &iso_pre);
      }
   }
   for (Part &p : parts)
      if (p.bytes) SB_TRY(hipMemcpyAsync(in.p + p.off, p.src, p.bytes, hipMemcpyHostToDevice, s));
   sbgpu_annotation_t dan = *an;
   if (res) dan = res->dev;
   else
      dan.iso_off = (const int64_t *)(in.p + parts[0].off);
   if (!res) {
      dan.exon_off = (const int64_t *)(in.p + parts[1].off);
      dan.seg_off = (const int64_t *)(in.p + parts[2].off);
      dan.exon_left = (const uint32_t *)(in.p + parts[4].off);
      dan.exon_right = (const uint32_t *)(in.p + parts[5].off);
      dan.seg_left = (const uint32_t *)(in.p + parts[6].off);
      dan.seg_right = (const uint32_t *)(in.p + parts[7].off);
   }
   sbgpu_hits_t dh = *hits;
   const float *d_mass = hit_mass;
   if (!on_dev) {
      dh.feat_off = (const int64_t *)(in.p + parts[3].off);
      dh.hit_locus = (const int32_t *)(in.p + parts[8].off);
      dh.feat_left = (const uint32_t *)(in.p + parts[9].off);
      dh.feat_right = (const uint32_t *)(in.p + parts[10].off);
      dh.feat_code = (const uint8_t *)(in.p + parts[11].off);
      d_mass = (const float *)(in.p + parts[12].off);
   }
   uint32_t *d_compat = (uint32_t *)(in.p + o_compat), *d_key = (uint32_t *)(in.p + o_key);
   // hit -> bin of the host entry: the handle keeps it in HBM (an arena of its own) and brings it over when an export asks
   struct ArenaGuard {
      char *p = nullptr;
      size_t cap = 0;
      ~ArenaGuard() { sb::dev_give(p, cap); }
   } hb_arena;
   if (!on_dev && nh) {
      const hipError_t eh = sb::dev_take((size_t)nh * 8, &hb_arena.p, &hb_arena.cap);
      if (eh != hipSuccess) return api_fail(eh == hipErrorOutOfMemory ? SBGPU_ENOMEM : SBGPU_EHIP, std::string("hipMalloc(hit -> bin): ") + hipGetErrorString(eh));
   }
   int64_t *d_hit_bin = hb_arena.p ? (int64_t *)hb_arena.p : (int64_t *)(in.p + o_hbin);
   uint64_t *d_span = (uint64_t *)(in.p + o_span);
   uint32_t *d_fhash = (uint32_t *)(in.p + o_fhash);

   stage("upload");
   sb::ctx_stage_reset(c);
   // ---- A5: the interval tests
   sb::ctx_stage_begin(c, "exonbin_kernel", s);
   if (nh) SB_RC(sb::exonbin_device_impl(c, &dan, &dh, cw, kw, d_compat, d_key, d_span, d_fhash, s, n_iso, res ? &res->seg_basis : nullptr));
   sb::ctx_stage_end(c, s);
   stage("exonbin kernel");
   std::vector<uint32_t> compat_h, key_h;
   auto need_compat = [&]() -> int {
      if (compat_h.empty() && nh) {
         compat_h.resize((size_t)nh * cw);
         SB_TRY(hipMemcpyAsync(compat_h.data(), d_compat, compat_h.size() * 4, hipMemcpyDeviceToHost, s));
         SB_TRY(hipStreamSynchronize(s));
      }
      return SBGPU_OK;
   };
   // the insert-size table reaches as far as the longest locus' segments together (no (bin, isoform) pair spans more)
   int64_t max_l = 1;
   if (!long_read && res) max_l = res->max_locus_span;
   else if (!long_read)
      for (int64_t l = 0; l < nl; ++l) {
         int64_t tot = 0;
         for (int64_t k = an->seg_off[l]; k < an->seg_off[l + 1]; ++k) tot += (int64_t)an->seg_right[k] - an->seg_left[k] + 1;
         max_l = std::max(max_l, tot);
      }
   if (max_l > (1 << 26)) return api_fail(SBGPU_ESHAPE, "sbgpu_quantify_host: segment lengths out of range");
   const int32_t pdf_len = (int32_t)max_l + 1;
   char *d_pdf = nullptr;
   if (hipError_t ep = sb::ctx_scratch(c, 6, (size_t)pdf_len * 8, &d_pdf); ep != hipSuccess)
      return api_fail(ep == hipErrorOutOfMemory ? SBGPU_ENOMEM : SBGPU_EHIP, std::string("hipMalloc: ") + hipGetErrorString(ep));
   std::vector<double> pdf((size_t)pdf_len, 0.0);
   hipStream_t cs = nullptr;
   hipEvent_t ev_pdf = nullptr, ev_hist = nullptr;
   SB_TRY(sb::ctx_copy_stream(c, &cs));
   SB_TRY(sb::ctx_event(c, 3, &ev_pdf));
   // ---- the insert-size law: given, or the empirical one from the hits (pass 1 of the reference: Sample::fragLenDist,
   // src/alignments.cpp:1363-1410, on the device -- fraglen_device.h)
   sbgpu_insert_t ins;
   std::vector<double> emp_hist;
   int64_t n_frag_lens = 0, hist_len = 0;
   const unsigned long long *h_hist = nullptr; // pinned: the histogram as the device (and the other ranks) made it
   int64_t mapped_total = ro ? ro->mapped_reads : 0;
   if (insert) {
      ins = *insert;
      ins.read_len = read_len;
      ins.long_read = long_read;
      // the table of the law, on the copy stream beside the kernels (the bin-weight launch waits for its event)
      SB_RC(sbgpu_insert_pdf_table(&ins, pdf_len, pdf.data()));
      SB_TRY(hipMemcpyAsync(d_pdf, pdf.data(), (size_t)pdf_len * 8, hipMemcpyHostToDevice, cs));
      SB_TRY(hipEventRecord(ev_pdf, cs));
      if (ro && ro->comm) { // Sample::total_mapped_reads() over all ranks (alignments.cpp:1372)
         SB_RC(sbgpu_allreduce_sum_i64_host(ro->comm, &mapped_total, 1));
      }
   } else {
      // every rank's histogram has the same length: the longest locus of any of them
      hist_len = pdf_len;
      if (ro && ro->comm) SB_RC(sbgpu_allreduce_max_i64_host(ro->comm, &hist_len, 1));
      char *d_hist = nullptr, *pin = nullptr;
      const size_t hist_bytes = (size_t)(hist_len + 2) * 8; // [hist_len]: lengths beyond the table; [hist_len + 1]: the mapped-read total
      if (hipError_t eh = sb::ctx_scratch(c, 7, hist_bytes, &d_hist); eh != hipSuccess)
         return api_fail(eh == hipErrorOutOfMemory ? SBGPU_ENOMEM : SBGPU_EHIP, std::string("hipMalloc: ") + hipGetErrorString(eh));
      SB_TRY(sb::ctx_pinned(c, 1, hist_bytes, &pin));
      SB_TRY(sb::ctx_event(c, 5, &ev_hist));
      SB_TRY(hipMemsetAsync(d_hist, 0, hist_bytes, s));
      if (ro) SB_TRY(hipMemcpyAsync(d_hist + (size_t)(hist_len + 1) * 8, &ro->mapped_reads, 8, hipMemcpyHostToDevice, s));
      sb::ctx_stage_begin(c, "fraglen_hist_kernel", s);
      if (nh) {
         sb::FragLenArgs fa;
         fa.n_hits = nh, fa.compat_words = cw;
         fa.hit_locus = dh.hit_locus, fa.compat = d_compat, fa.span = d_span;
         fa.iso_off = dan.iso_off, fa.exon_off = dan.exon_off, fa.exon_left = dan.exon_left, fa.exon_right = dan.exon_right;
         fa.hist_len = hist_len, fa.hist = (unsigned long long *)d_hist;
         const int64_t want = std::min<int64_t>((nh + sb::kFragLenThreads - 1) / sb::kFragLenThreads, (int64_t)sb::ctx_cu_count(c) * 2);
         hipLaunchKernelGGL(sb::fraglen_hist_kernel, dim3((unsigned)want), dim3(sb::kFragLenThreads), 0, s, fa);
         SB_TRY(hipGetLastError());
      }
      sb::ctx_stage_end(c, s);
      // the law is the WHOLE sample's: one all-reduce(sum) of the histogram (exact integers) with the mapped-read total behind it
      if (ro && ro->comm) SB_RC(sbgpu_allreduce_sum_i64(ro->comm, (int64_t *)d_hist, hist_len + 2, s));
      SB_TRY(hipMemcpyAsync(pin, d_hist, hist_bytes, hipMemcpyDeviceToHost, s));
      SB_TRY(hipEventRecord(ev_hist, s));
      h_hist = (const unsigned long long *)pin;
   }
   // called before the bin weights are launched: by then the histogram has long arrived (the grouping's kernels ran behind it)
   bool law_ready = insert != nullptr;
   auto finish_law = [&]() -> int {
      if (law_ready) return SBGPU_OK;
      SB_TRY(hipEventSynchronize(ev_hist));
      if (h_hist[hist_len]) return api_fail(SBGPU_ESHAPE, "sbgpu_quantify_host: a fragment length beyond the longest locus");
      if (ro) mapped_total = (int64_t)h_hist[hist_len + 1];
      // InsertSize(const vector<int> frag_lens), src/read.cpp:238-262; mean_and_sd_insert_size :14-20 -- size, sum, sum of
      // squares, extremes and histogram of the sample, all read off the histogram (integers: exact in any order; the
      // reference's running doubles are the same numbers as long as they stay below 2^53)
      int64_t lo = -1, hi = -1;
      unsigned long long n = 0;
      unsigned __int128 sum = 0, sq = 0;
      for (int64_t l = 0; l < hist_len; ++l)
         if (const unsigned long long k = h_hist[l]) {
            if (lo < 0) lo = l;
            hi = l;
            n += k;
            sum += (unsigned __int128)k * (unsigned long long)l;
            sq += (unsigned __int128)k * (unsigned long long)l * (unsigned long long)l;
         }
      if (n < 1) return api_fail(SBGPU_EINVAL, "sbgpu_quantify_host: no hit fits exactly one transcript: no empirical insert-size law (\"Not enough reads\")");
      if (n > (unsigned long long)INT32_MAX) return api_fail(SBGPU_ESHAPE, "sbgpu_quantify_host: more than 2^31 fragment lengths");
      emp_hist.assign((size_t)(hi - lo + 1), 0.0);
      for (int64_t l = lo; l <= hi; ++l) emp_hist[(size_t)(l - lo)] = (double)h_hist[l];
      n_frag_lens = (int64_t)n;
      ins.mean = (double)sum / (double)n;
      ins.sd = std::sqrt((double)sq / (double)n - ins.mean * ins.mean);
      ins.use_emp = 1;
      ins.start_offset = (int32_t)lo;
      ins.end_offset = (int32_t)hi;
      ins.total_reads = (int32_t)n;
      ins.emp_hist = emp_hist.data();
      ins.read_len = read_len;
      ins.long_read = long_read;
      SB_RC(sbgpu_insert_pdf_table(&ins, pdf_len, pdf.data()));
      SB_TRY(hipMemcpyAsync(d_pdf, pdf.data(), (size_t)pdf_len * 8, hipMemcpyHostToDevice, cs));
      SB_TRY(hipEventRecord(ev_pdf, cs));
      law_ready = true;
      stage("insert size");
      return SBGPU_OK;
   };
   stage("pdf table");
   // ---- everything behind the bins: weights straight into the EM batch's F (A4), plan + EM (A1/A2), the downloads.
   // With the device grouping the weights are launched from inside bins_create_device_impl -- as soon as the pairs' fill
   // kernel is in the stream -- so that the handle's bookkeeping runs beside that kernel; the EM goes in once the plan's
   // thread is through, still before the weights are done.
   sbgpu_plan_t *plan = nullptr;
   struct PlanGuard {
      sbgpu_plan_t *&p;
      ~PlanGuard() { if (p) sbgpu_plan_destroy(p); }
   } plan_guard = {plan};
   DeviceBuf w;
   std::vector<double> F;
   // the EM plan is host work of a millisecond or two (size classes, one upload): with the device grouping a helper
   // thread makes it as soon as the loci's bin counts are known, beside the pairs' kernels
   struct PlanJob {
      std::thread t;
      std::vector<int64_t> row_off, f_off;
      int rc = SBGPU_OK;
      std::string err;
      bool started = false;
      ~PlanJob()
      {
         if (t.joinable()) t.join();
      }
   } plan_job; // (declared after plan_guard: joined before the plan is destroyed)
   size_t q_theta = 0, q_st = 0, q_it = 0, q_fpkm = 0, q_frac = 0, q_tpm = 0, q_keep = 0, q_sum = 0, q_ilen = 0;
   hipError_t e1 = hipSuccess, e2 = hipSuccess, e3 = hipSuccess, e4 = hipSuccess;
   int64_t n_bins = 0, n_elem = 0, n_pairs = 0, n_psegs = 0;
   size_t q_F = 0;
   // host pair arrays (host grouping only)
   std::vector<int64_t> row_off_h, iso_off_h, f_off_h, pair_seg_off, pair_out;
   std::vector<int32_t> count_h, pair_len;
   std::vector<uint32_t> pair_segs, pair_mask;
   size_t q_cnt = 0;
   // the bin weights, straight into the EM batch's F
   auto launch_weights = [&](const sb::DevicePairs *dpairs) -> int {
      if (!long_read) {
         if (dpairs) {
            if (dpairs->any_wide) return api_fail(SBGPU_ESHAPE, "sbgpu_quantify_host: a bin spans more than 32 isoform segments");
         } else {
            for (int64_t p = 0; p < n_pairs; ++p)
               if (pair_seg_off[(size_t)p + 1] == pair_seg_off[(size_t)p])
                  return api_fail(SBGPU_ESHAPE, "sbgpu_quantify_host: a bin spans more than 32 isoform segments");
         }
      }
      const size_t np1 = (size_t)std::max<int64_t>(dpairs ? 1 : n_pairs, 1), ne1 = (size_t)std::max<int64_t>(n_elem, 1),
                   nb1 = (size_t)std::max<int64_t>(n_bins, 1), ns1 = (size_t)(dpairs ? 1 : n_psegs + 1);
      size_t t2 = 0;
      const size_t q_off = t2; t2 += up256((np1 + 1) * 8);
      const size_t q_idx = t2; t2 += up256(np1 * 8);
      const size_t q_seg = t2; t2 += up256(ns1 * 4);
      const size_t q_mask = t2; t2 += up256(np1 * 4);
      const size_t q_len = t2; t2 += up256(np1 * 4);
      q_cnt = t2; t2 += up256(nb1 * 4);
      q_F = t2; t2 += up256(ne1 * 8);
      q_theta = t2; t2 += up256((size_t)(n_iso + 1) * 8);
      q_st = t2; t2 += up256((size_t)(nl + 1) * 4);
      q_it = t2; t2 += up256((size_t)(nl + 1) * 4);
      if (ro) { // the epilogue's arrays (sbgpu_quantify_resident)
         q_fpkm = t2; t2 += up256((size_t)(n_iso + 1) * 8);
         q_frac = t2; t2 += up256((size_t)(n_iso + 1) * 8);
         q_tpm = t2; t2 += up256((size_t)(n_iso + 1) * 8);
         q_keep = t2; t2 += up256((size_t)(n_iso + 1) * 4);
         q_ilen = t2; t2 += up256((size_t)(n_iso + 1) * 4);
         q_sum = t2; t2 += up256(8);
      }
      hipError_t ew = sb::ctx_scratch(c, 1, t2, &w.p);
      if (ew != hipSuccess) return api_fail(ew == hipErrorOutOfMemory ? SBGPU_ENOMEM : SBGPU_EHIP, std::string("hipMalloc: ") + hipGetErrorString(ew));
      SB_TRY(hipMemsetAsync(w.p + q_F, 0, ne1 * 8, s));
      SB_RC(finish_law());
      SB_TRY(hipStreamWaitEvent(s, ev_pdf, 0));
      if (n_pairs) {
         const int64_t *d_off = dpairs ? dpairs->seg_off() : (const int64_t *)(w.p + q_off);
         const uint32_t *d_seg = dpairs ? dpairs->seg_lens() : (const uint32_t *)(w.p + q_seg);
         const uint32_t *d_msk = dpairs ? dpairs->mask() : (const uint32_t *)(w.p + q_mask);
         const int32_t *d_len = dpairs ? dpairs->iso_len() : (const int32_t *)(w.p + q_len);
         const int64_t *d_idx = dpairs ? dpairs->out_index() : (const int64_t *)(w.p + q_idx);
         if (!dpairs) {
            SB_TRY(hipMemcpyAsync(w.p + q_off, pair_seg_off.data(), (size_t)(n_pairs + 1) * 8, hipMemcpyHostToDevice, s));
            SB_TRY(hipMemcpyAsync(w.p + q_idx, pair_out.data(), (size_t)n_pairs * 8, hipMemcpyHostToDevice, s));
            if (n_psegs) SB_TRY(hipMemcpyAsync(w.p + q_seg, pair_segs.data(), (size_t)n_psegs * 4, hipMemcpyHostToDevice, s));
            SB_TRY(hipMemcpyAsync(w.p + q_mask, pair_mask.data(), (size_t)n_pairs * 4, hipMemcpyHostToDevice, s));
            SB_TRY(hipMemcpyAsync(w.p + q_len, pair_len.data(), (size_t)n_pairs * 4, hipMemcpyHostToDevice, s));
         }
         const int32_t lmin_base = ins.use_emp ? ins.start_offset : ins.read_len;
         sb::ctx_stage_begin(c, "binweight_kernel", s);
         SB_RC(sbgpu_binweight_device(c, n_pairs, d_off, d_seg, d_msk, d_len, d_idx, (const double *)d_pdf, pdf_len,
                                      ins.read_len, lmin_base, ins.long_read, (double *)(w.p + q_F), s));
         sb::ctx_stage_end(c, s);
      }
      stage("bin weights");
      return SBGPU_OK;
   };
   // ---- A1/A2: the EM behind them (the counts are on the device already when the grouping ran there)
   auto launch_em = [&](const int64_t *row_off, const int64_t *f_off, const int32_t *d_count, const int32_t *h_count) -> int {
      if (!d_count) {
         if (n_bins) SB_TRY(hipMemcpyAsync(w.p + q_cnt, h_count, (size_t)n_bins * 4, hipMemcpyHostToDevice, s));
         d_count = (const int32_t *)(w.p + q_cnt);
      }
      if (plan_job.started) {
         plan_job.t.join();
         plan_job.started = false;
         if (plan_job.rc != SBGPU_OK) return api_fail(plan_job.rc, plan_job.err);
      } else {
         SB_RC(sbgpu_plan_create(c, nl, row_off, an->iso_off, f_off, &plan));
      }
      stage("plan");
      sb::ctx_stage_begin(c, "em kernels", s);
      const int rce = sbgpu_em_run_device(c, plan, d_count, (const double *)(w.p + q_F), (double *)(w.p + q_theta),
                                          (int32_t *)(w.p + q_st), (int32_t *)(w.p + q_it), s);
      sb::ctx_stage_end(c, s);
      return rce;
   };
   // ---- A3 / A7 behind the EM (sbgpu_quantify_resident): theta -> FPKM / Frac / keep (estimate.cpp:314-355), the FPKM total of
   // ALL ranks (alignments.cpp:1821-1824: the path's one collective per step), TPM (:1825-1829) -- theta never leaves the device
   auto launch_epilogue = [&]() -> int {
      if (!ro) return SBGPU_OK;
      if (mapped_total < 1 || mapped_total > (int64_t)INT32_MAX)
         return api_fail(SBGPU_EINVAL, "sbgpu_quantify_resident: the mapped-read total must be in [1, 2^31) (Sample::total_mapped_reads() is an int)");
      const int32_t *d_len = nullptr;
      if (res) {
         d_len = res->d_iso.len;
      } else {
         if ((int64_t)iso_pre.len.size() != n_iso) sb::iso_segments(an, &iso_pre);
         SB_TRY(hipMemcpyAsync(w.p + q_ilen, iso_pre.len.data(), (size_t)n_iso * 4, hipMemcpyHostToDevice, s));
         d_len = (const int32_t *)(w.p + q_ilen);
      }
      sbgpu_abundance_params_t par = *ro->params;
      par.total_mapped_reads = (int32_t)mapped_total;
      par.insert_mean = ins.mean; // _sample._insert_size_dist->_mean (estimate.cpp:318)
      sb::ctx_stage_begin(c, "abundance + tpm", s);
      SB_RC(sbgpu_abundance_device(c, plan, (const double *)(w.p + q_theta), (const int32_t *)(w.p + q_st), d_len, &par, (double *)(w.p + q_fpkm),
                                   (double *)(w.p + q_frac), (int32_t *)(w.p + q_keep), (double *)(w.p + q_sum), s));
      if (ro->comm) SB_RC(sbgpu_allreduce_sum_f64(ro->comm, (double *)(w.p + q_sum), 1, s));
      SB_RC(sbgpu_tpm_device(c, n_iso, (const double *)(w.p + q_fpkm), (const int32_t *)(w.p + q_keep), (const double *)(w.p + q_sum),
                             (double *)(w.p + q_tpm), s));
      sb::ctx_stage_end(c, s);
      return SBGPU_OK;
   };
   // the results come down last: a copy into the caller's pageable memory holds the host until the EM is done, and
   // the handle's host work is to run beside the kernels, not behind them
   hipError_t e7 = hipSuccess;
   auto download = [&]() {
      F.assign(on_dev ? (size_t)0 : (size_t)n_elem, 0.0); // (device entry: the weights are not brought back)
      double *h_theta = ro ? ro->out->theta : theta_out;
      int32_t *h_st = ro ? ro->out->status : status_out, *h_it = ro ? ro->out->iters : iters_out;
      if (h_theta) e1 = hipMemcpyAsync(h_theta, w.p + q_theta, (size_t)n_iso * 8, hipMemcpyDeviceToHost, s);
      if (h_st) e2 = hipMemcpyAsync(h_st, w.p + q_st, (size_t)nl * 4, hipMemcpyDeviceToHost, s);
      if (h_it) e3 = hipMemcpyAsync(h_it, w.p + q_it, (size_t)nl * 4, hipMemcpyDeviceToHost, s);
      e4 = (n_elem && !on_dev) ? hipMemcpyAsync(F.data(), w.p + q_F, (size_t)n_elem * 8, hipMemcpyDeviceToHost, s) : hipSuccess;
      if (ro) {
         sbgpu_abundances_t *o = ro->out;
         auto get = [&](void *dst, size_t off, size_t bytes) {
            if (dst && e7 == hipSuccess) e7 = hipMemcpyAsync(dst, w.p + off, bytes, hipMemcpyDeviceToHost, s);
         };
         get(o->fpkm, q_fpkm, (size_t)n_iso * 8), get(o->frac, q_frac, (size_t)n_iso * 8), get(o->tpm, q_tpm, (size_t)n_iso * 8);
         get(o->keep, q_keep, (size_t)n_iso * 4), get(&o->total_fpkm, q_sum, 8);
         o->d_theta = (const double *)(w.p + q_theta), o->d_fpkm = (const double *)(w.p + q_fpkm), o->d_frac = (const double *)(w.p + q_frac);
         o->d_tpm = (const double *)(w.p + q_tpm), o->d_keep = (const int32_t *)(w.p + q_keep);
         o->d_status = (const int32_t *)(w.p + q_st), o->d_iters = (const int32_t *)(w.p + q_it);
         o->total_mapped_reads = mapped_total, o->n_frag_lens = n_frag_lens;
      }
   };
   // ---- A5: bins (device; host when the device form declines)
   sbgpu_bins_t *bins = nullptr;
   int rc = SBGPU_EUNSUPPORTED;
   bool rest_launched = false;
   if (grouped && nh) {
      if (iso_worker.t.joinable()) iso_worker.t.join();
      sb::GroupingHooks hooks;
      hooks.d_annot = &dan;
      if (res) hooks.d_iso = &res->d_iso;
      hooks.rows_known = [&](const int64_t *row_off, const int64_t *f_off) {
         plan_job.row_off.assign(row_off, row_off + nl + 1); // (the caller's arrays do not outlive its frame)
         plan_job.f_off.assign(f_off, f_off + nl + 1);
         try {
            plan_job.t = std::thread([&]() {
               plan_job.rc = sbgpu_plan_create(c, nl, plan_job.row_off.data(), an->iso_off, plan_job.f_off.data(), &plan);
               if (plan_job.rc != SBGPU_OK) plan_job.err = sbgpu_last_error(); // (the error slot is per thread)
            });
            plan_job.started = true;
         } catch (const std::system_error &) { // no thread to be had: launch_em makes the plan
         }
      };
      const int32_t *d_count_dev = nullptr;
      hooks.after_pairs = [&](const sb::DeviceGrouping &g) -> int {
         n_bins = g.n_bins, n_elem = g.n_elem, n_pairs = g.pairs->n_pairs, n_psegs = g.pairs->n_pair_segs;
         d_count_dev = g.d_count; // (the handle's own arena, sb::dev_take'n by the grouping: valid as long as the handle lives)
         rest_launched = true;
         return launch_weights(g.pairs);
      };
      // (hits given on the device: the caller did not ask for hit -> bin, so it is not made)
      rc = sb::bins_create_device_impl(c, an, &dh, d_mass, locus_hit_off.data(), cw, kw, d_compat, d_key, on_dev ? nullptr : d_hit_bin, s,
                                       res ? &res->iso : &iso_pre, &bins, d_span, d_fhash, &hooks);
      if (rc == SBGPU_OK && rest_launched) rc = launch_em(plan_job.row_off.data(), plan_job.f_off.data(), d_count_dev, nullptr);
      if (rc == SBGPU_OK && rest_launched) rc = launch_epilogue();
      if (rc != SBGPU_OK) {
         // a grouping that failed after the plan's thread was started: the thread is over before anything else happens
         if (plan_job.t.joinable()) plan_job.t.join();
         plan_job.started = false;
         if (rest_launched) (void)hipStreamSynchronize(s);
         if (plan) {
            sbgpu_plan_destroy(plan);
            plan = nullptr;
         }
         if (rest_launched) { // the handle exists already
            sbgpu_bins_destroy(bins);
            bins = nullptr;
            return rc;
         }
      }
   }
   const bool on_device = rc == SBGPU_OK;
   if (rc == SBGPU_EUNSUPPORTED && on_dev && nh) // (no hits at all: the host code makes the empty handle)
      return api_fail(rc, "sbgpu_quantify_device: the device grouping does not cover these hits (unsorted, fractional masses or a locus of thousands of bins): use sbgpu_quantify_host");
   std::string why_host;
   if (rc == SBGPU_EUNSUPPORTED || !(grouped && nh)) {
      // the host code groups (same bins, slower): the handle says so and why (sbgpu_bins_grouping)
      why_host = !nh ? "no hits" : !grouped ? "the hits are not grouped by locus" : sbgpu_last_error();
      if (timing && nh) std::fprintf(stderr, "sbgpu_quantify_host: the device grouping declined: %s\n", why_host.c_str());
   }
   if (rc == SBGPU_EUNSUPPORTED) {
      SB_RC(need_compat());
      key_h.resize(nh1 * (size_t)kw);
      if (nh) {
         SB_TRY(hipMemcpyAsync(key_h.data(), d_key, (size_t)nh * kw * 4, hipMemcpyDeviceToHost, s));
         SB_TRY(hipStreamSynchronize(s));
      }
      if (compat_h.empty()) compat_h.resize((size_t)cw);
      rc = sbgpu_bins_create(an, hits, hit_mass, cw, kw, compat_h.data(), key_h.data(), &bins);
   }
   if (rc != SBGPU_OK) return rc;
   if (!on_device) sb::bins_set_grouping(bins, false, why_host);
   struct BinsGuard {
      sbgpu_bins_t *b;
      ~BinsGuard() { sbgpu_bins_destroy(b); }
   } guard = {bins};
   stage("bins + pairs");
   if (!rest_launched) { // host grouping (or no hits at all): the pairs come from the handle
      int64_t info[8];
      SB_RC(sbgpu_bins_info(bins, info));
      n_bins = info[2], n_elem = info[3], n_pairs = info[4], n_psegs = info[5];
      const sb::DevicePairs *dpairs = sb::bins_device_pairs(bins);
      row_off_h.resize((size_t)nl + 1), iso_off_h.resize((size_t)nl + 1), f_off_h.resize((size_t)nl + 1), count_h.resize((size_t)n_bins + 1);
      if (!dpairs) {
         pair_seg_off.resize((size_t)n_pairs + 1);
         pair_out.resize((size_t)n_pairs + 1);
         pair_len.resize((size_t)n_pairs + 1);
         pair_segs.resize((size_t)n_psegs + 1);
         pair_mask.resize((size_t)n_pairs + 1);
      }
      SB_RC(sbgpu_bins_export(bins, row_off_h.data(), iso_off_h.data(), f_off_h.data(), count_h.data(), nullptr, nullptr, nullptr, nullptr,
                              dpairs ? nullptr : pair_seg_off.data(), dpairs ? nullptr : pair_segs.data(),
                              dpairs ? nullptr : pair_mask.data(), dpairs ? nullptr : pair_len.data(),
                              dpairs ? nullptr : pair_out.data()));
      stage("export");
      SB_RC(launch_weights(dpairs));
      SB_RC(launch_em(row_off_h.data(), f_off_h.data(), nullptr, count_h.data()));
      SB_RC(launch_epilogue());
   }
   download();
   hipError_t e5 = hipSuccess;
   hipError_t e6 = hipStreamSynchronize(s);
   stage("plan + EM + download");
   for (hipError_t x : {e1, e2, e3, e4, e5, e6, e7})
      if (x != hipSuccess) return api_fail(SBGPU_EHIP, std::string("sbgpu_quantify_host: download: ") + hipGetErrorString(x));
   // a wide-locus barrier that timed out leaves its loci unsolved (status SBGPU_EM_UNSOLVED): that is a failed call
   if (sb::ctx_take_wide_error(c))
      return api_fail(SBGPU_EHIP, "sbgpu_quantify_host: a barrier of the wide-locus EM kernel timed out: the loci it served have no result");
   if (compat_out) {
      SB_RC(need_compat());
      if (nh) std::memcpy(compat_out, compat_h.data(), (size_t)nh * cw * 4);
   }
   if (on_device && nh && !on_dev) { // hit -> bin stays in HBM with the handle (8 bytes per hit cross PCIe on request only)
      sb::bins_set_device_hit_bin(bins, hb_arena.p, hb_arena.cap, nh);
      hb_arena.p = nullptr, hb_arena.cap = 0;
   }
   if (insert_used) {
      *insert_used = ins;
      if (!insert) {
         // the histogram lives on with the handle: behind the weights
         const size_t at = F.size();
         F.insert(F.end(), emp_hist.begin(), emp_hist.end());
         sb::bins_set_weights(bins, std::move(F));
         insert_used->emp_hist = sb::bins_weights_tail(bins, at);
      } else {
         sb::bins_set_weights(bins, std::move(F));
      }
   } else {
      sb::bins_set_weights(bins, std::move(F));
   }
#undef SB_TRY
#undef SB_RC
   guard.b = nullptr;
   *bins_out = bins;
   stage("results + handle");
   return SBGPU_OK;
}

extern "C" {

int sbgpu_quantify_host(sbgpu_ctx_t *c, const sbgpu_annotation_t *an, const sbgpu_hits_t *hits, const float *hit_mass,
                        const sbgpu_insert_t *insert, int32_t read_len, int32_t long_read, double *theta_out,
                        int32_t *status_out, int32_t *iters_out, uint32_t *compat_out, sbgpu_insert_t *insert_used,
                        sbgpu_bins_t **bins_out)
{
   return quantify_impl(c, an, hits, hit_mass, nullptr, insert, read_len, long_read, theta_out, status_out, iters_out, compat_out,
                        insert_used, bins_out);
}

int sbgpu_annotation_pin(sbgpu_ctx_t *c, const sbgpu_annotation_t *an)
{
   if (!c || !an) return api_fail(SBGPU_EINVAL, "sbgpu_annotation_pin: null argument");
   const int64_t nl = an->n_loci;
   if (nl < 1 || !an->iso_off || !an->exon_off || !an->seg_off) return api_fail(SBGPU_EINVAL, "sbgpu_annotation_pin: bad annotation");
   const int64_t n_iso = an->iso_off[nl], n_exon = an->exon_off[n_iso], n_seg = an->seg_off[nl];
   if ((n_exon && (!an->exon_left || !an->exon_right)) || (n_seg && (!an->seg_left || !an->seg_right)))
      return api_fail(SBGPU_EINVAL, "sbgpu_annotation_pin: null array");
   sb::ResidentAnnotation *r = new (std::nothrow) sb::ResidentAnnotation();
   if (!r) return api_fail(SBGPU_ENOMEM, "sbgpu_annotation_pin: out of host memory");
   r->key = *an;
   r->print = sb::ResidentAnnotation::fingerprint(an);
   for (int64_t l = 0; l < nl; ++l) {
      r->max_iso = std::max(r->max_iso, an->iso_off[l + 1] - an->iso_off[l]);
      r->max_seg = std::max(r->max_seg, an->seg_off[l + 1] - an->seg_off[l]);
      int64_t tot = 0;
      for (int64_t k = an->seg_off[l]; k < an->seg_off[l + 1]; ++k) tot += (int64_t)an->seg_right[k] - an->seg_left[k] + 1;
      r->max_locus_span = std::max(r->max_locus_span, tot);
   }
   sb::iso_segments(an, &r->iso);
   struct Part {
      const void *src;
      size_t bytes, off;
   } parts[] = {
      {an->iso_off, (size_t)(nl + 1) * 8, 0},    {an->exon_off, (size_t)(n_iso + 1) * 8, 0}, {an->seg_off, (size_t)(nl + 1) * 8, 0},
      {an->exon_left, (size_t)n_exon * 4, 0},    {an->exon_right, (size_t)n_exon * 4, 0},    {an->seg_left, (size_t)n_seg * 4, 0},
      {an->seg_right, (size_t)n_seg * 4, 0},     {r->iso.seg_off.data(), r->iso.seg_off.size() * 8, 0},
      {r->iso.seg_idx.data(), r->iso.seg_idx.size() * 4, 0}, {r->iso.locus.data(), r->iso.locus.size() * 4, 0},
      {r->iso.len.data(), r->iso.len.size() * 4, 0},
   };
   size_t total = 0;
   for (Part &p : parts) {
      p.off = total;
      total += up256(p.bytes ? p.bytes : 8);
   }
   const size_t o_segbasis = total;
   total += sb::seg_basis_bytes(nl, n_iso);
   hipError_t e = hipSetDevice(sb::ctx_device(c));
   if (e == hipSuccess) e = sb::dev_take(total, &r->arena, &r->capacity);
   for (Part &p : parts)
      if (e == hipSuccess && p.bytes) e = hipMemcpy(r->arena + p.off, p.src, p.bytes, hipMemcpyHostToDevice);
   if (e != hipSuccess) {
      sb::dev_give(r->arena, r->capacity);
      delete r;
      return api_fail(e == hipErrorOutOfMemory ? SBGPU_ENOMEM : SBGPU_EHIP, std::string("sbgpu_annotation_pin: ") + hipGetErrorString(e));
   }
   r->dev = *an;
   r->dev.iso_off = (const int64_t *)(r->arena + parts[0].off);
   r->dev.exon_off = (const int64_t *)(r->arena + parts[1].off);
   r->dev.seg_off = (const int64_t *)(r->arena + parts[2].off);
   r->dev.exon_left = (const uint32_t *)(r->arena + parts[3].off);
   r->dev.exon_right = (const uint32_t *)(r->arena + parts[4].off);
   r->dev.seg_left = (const uint32_t *)(r->arena + parts[5].off);
   r->dev.seg_right = (const uint32_t *)(r->arena + parts[6].off);
   r->d_iso.seg_off = (const int64_t *)(r->arena + parts[7].off);
   r->d_iso.seg_idx = (const int32_t *)(r->arena + parts[8].off);
   r->d_iso.locus = (const int32_t *)(r->arena + parts[9].off);
   r->d_iso.len = (const int32_t *)(r->arena + parts[10].off);
   // the isoforms in the segment basis (the exon-bin kernel's masks): once, here
   int rc_sb = sb::make_seg_basis(c, &r->dev, n_iso, r->arena + o_segbasis, sb::ctx_stream(c), &r->seg_basis);
   if (rc_sb == SBGPU_OK && hipStreamSynchronize(sb::ctx_stream(c)) != hipSuccess) rc_sb = api_fail(SBGPU_EHIP, "sbgpu_annotation_pin: iso_masks_kernel failed");
   if (rc_sb != SBGPU_OK) {
      sb::dev_give(r->arena, r->capacity);
      delete r;
      return rc_sb;
   }
   sb::ctx_set_resident_annotation(c, r);
   return SBGPU_OK;
}

int sbgpu_annotation_unpin(sbgpu_ctx_t *c)
{
   if (!c) return api_fail(SBGPU_EINVAL, "sbgpu_annotation_unpin: null context");
   sb::ctx_set_resident_annotation(c, nullptr);
   return SBGPU_OK;
}

int sbgpu_annotation_unpin_matching(sbgpu_ctx_t *c, const sbgpu_annotation_t *an, int32_t *released)
{
   if (!c || !an) return api_fail(SBGPU_EINVAL, "sbgpu_annotation_unpin_matching: null argument");
   const sb::ResidentAnnotation *res = sb::ctx_resident_annotation(c);
   const bool mine = res && res->same_arrays(an);
   if (mine) sb::ctx_set_resident_annotation(c, nullptr);
   if (released) *released = mine ? 1 : 0;
   return SBGPU_OK;
}

int sbgpu_quantify_device(sbgpu_ctx_t *c, const sbgpu_annotation_t *an, const sbgpu_hits_t *d_hits, const float *d_hit_mass,
                          const int64_t *locus_hit_off, const sbgpu_insert_t *insert, int32_t read_len, int32_t long_read,
                          double *theta_out, int32_t *status_out, int32_t *iters_out, sbgpu_bins_t **bins_out)
{
   if (!locus_hit_off) return api_fail(SBGPU_EINVAL, "sbgpu_quantify_device: null locus_hit_off");
   sbgpu_insert_t used;
   return quantify_impl(c, an, d_hits, d_hit_mass, locus_hit_off, insert, read_len, long_read, theta_out, status_out, iters_out,
                        nullptr, &used, bins_out);
}

int sbgpu_quantify_resident(sbgpu_ctx_t *c, const sbgpu_annotation_t *an, const sbgpu_hits_t *d_hits, const float *d_hit_mass,
                            const int64_t *locus_hit_off, const sbgpu_insert_t *insert, int32_t read_len, int32_t long_read,
                            int64_t mapped_reads, const sbgpu_abundance_params_t *params, sbgpu_comm_t *comm,
                            sbgpu_insert_t *insert_used, sbgpu_abundances_t *out, sbgpu_bins_t **bins_out)
{
   if (!locus_hit_off || !params || !out) return api_fail(SBGPU_EINVAL, "sbgpu_quantify_resident: null argument");
   if (mapped_reads < 0) return api_fail(SBGPU_EINVAL, "sbgpu_quantify_resident: negative mapped-read count");
   sbgpu_insert_t used;
   const ResidentOpts ro = {mapped_reads, params, comm, out};
   return quantify_impl(c, an, d_hits, d_hit_mass, locus_hit_off, insert, read_len, long_read, nullptr, nullptr, nullptr, nullptr,
                        insert_used ? insert_used : &used, bins_out, &ro);
}

} // extern "C"
