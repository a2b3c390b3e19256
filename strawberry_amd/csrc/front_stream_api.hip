// strawberry_amd/csrc/front_stream_api.hip -- sbgpu_front_stream_* (include/sbgpu.h): alignment records -> abundances for a
// caller that holds the records in HOST memory and hands them over chunk by chunk, with a bounded footprint on the device.
//
// The reference streams too: Sample::nextClusterRefDemand / procSample hold ONE cluster's reads at a time
// (/root/reference/src/alignments.cpp:1145-1187, 1736-1811).  The device entries of the front end take a whole sample's records
// resident (196 GB of arenas at 3.9e8 records); here the unit is a chunk of the inflated record stream (BGZF inflate is zlib on
// host threads and stays the caller's):
//
//   push(chunk i)   the chunk's bytes start their way to the device (copy stream; two device buffers alternate), then chunk
//                   i - 1 -- uploaded during the push before -- is computed: sbgpu_bam_decode_device ->
//                   sbgpu_assign_reads_device over the clusters not yet finished -> the clusters that are COMPLETE (a record
//                   behind their end has been seen) -> sbgpu_pair_mates_device -> sbgpu_collapse_pairs_device -> their unique
//                   hits appended to the stream's store on the device.  The records of the first incomplete cluster onward
//                   are carried: their bytes are copied in front of the next chunk's (device to device) and decoded again
//                   with it -- a cluster's records are never split;
//   end()           the last chunk (every remaining cluster is complete now), then ONE sbgpu_quantify_resident over the
//                   store: pass 1 (the empirical insert-size law is the WHOLE sample's: it cannot be known earlier), bins,
//                   weights, EM, FPKM, the all-reduce, TPM.  The unique hits are a seventh of the records' bytes.
//
// So upload and compute overlap chunk by chunk, the arenas are a chunk's, and what grows with the sample is the store.
// Results: those of the resident entries on the whole sample, bit for bit (same clusters, same order, same kernels).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/sbgpu.h"
#include "api_internal.h"

using sb::api_fail;

namespace {
size_t up256(size_t b) { return (b + 255) & ~(size_t)255; }

// a device array that grows (geometric; the old block goes back to the pool)
struct Grow {
   char *p = nullptr;
   size_t cap = 0, used = 0;
   hipError_t reserve(size_t want, hipStream_t s)
   {
      if (want <= cap) return hipSuccess;
      size_t ncap = std::max(want, cap + cap / 2);
      char *np = nullptr;
      size_t got = 0;
      hipError_t e = sb::dev_take(ncap, &np, &got);
      if (e != hipSuccess) return e;
      if (used) e = hipMemcpyAsync(np, p, used, hipMemcpyDeviceToDevice, s);
      if (e == hipSuccess) e = hipStreamSynchronize(s);
      sb::dev_give(p, cap);
      p = np, cap = got;
      return e;
   }
   void release()
   {
      sb::dev_give(p, cap);
      p = nullptr, cap = used = 0;
   }
};

__global__ void add_i32_kernel(int32_t *x, int64_t n, int32_t add)
{
   const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
   if (i < n) x[i] += add;
}
__global__ void add_i64_kernel(int64_t *x, int64_t n, int64_t add)
{
   const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
   if (i < n) x[i] += add;
}
// a window's record offsets: the carried records' (the tail of the window before, re-based) and the chunk's (behind the carry)
__global__ void window_off_kernel(int64_t *dst, const int64_t *prev_tail, int64_t n_carry, int64_t c0, const int64_t *chunk_off, int64_t n_new,
                                  int64_t carry_bytes)
{
   const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
   if (i < n_carry) dst[i] = prev_tail[i] - c0;
   else if (i <= n_carry + n_new) dst[i] = carry_bytes + (n_new ? chunk_off[i - n_carry] : 0);
}
} // namespace

struct sbgpu_front_stream {
   sbgpu_ctx_t *ctx = nullptr;
   int device = 0;
   // the clusters (host copies), and how far the stream has come
   std::vector<int32_t> c_ref;
   std::vector<uint32_t> c_left, c_right;
   std::vector<uint8_t> c_strand;
   int64_t n_clusters = 0, k0 = 0; // clusters [0, k0) are done
   sbgpu_bam_opts_t opts = {};
   // two device buffers: [carry room | chunk]; the chunk in flight (uploaded, not yet computed)
   int64_t chunk_cap = 0;
   char *buf[2] = {nullptr, nullptr};
   size_t buf_cap[2] = {0, 0};
   int cur = 0;                    // the buffer that holds the pending chunk
   bool pending = false;
   int64_t pend_bytes = 0, pend_records = 0;
   Grow d_pend[2];                 // the pending chunk's record offsets (relative to the chunk's first byte) [n + 1], uploaded with the chunk
   hipEvent_t ev_up[2] = {nullptr, nullptr};
   hipStream_t up_stream = nullptr; // the uploads' own stream, at the LOWEST priority: HIP streams of one priority share a few hardware
                                    // queues (the context alone has ten), and an upload that lands in the compute stream's queue
                                    // serialises with the kernels it is meant to run beside; another priority is another queue
   // the carry: bytes (at the tail of the buffer just computed, copied in front of the next chunk) and their offsets -- the
   // tail of the window's offsets (device), from record carry_rec on, first byte carry_c0
   int64_t carry_bytes = 0, carry_n = 0, carry_rec = 0, carry_c0 = 0;
   Grow d_win[2];                  // the windows' record offsets on the device (this window's, the one before's)
   int win = 0;
   // the store: the unique hits of the finished clusters (sbgpu_hits_t layout)
   Grow s_locus, s_foff, s_code, s_left, s_right, s_mass;
   int64_t n_hits = 0, n_feat = 0;
   std::vector<int64_t> locus_hit_off; // [n_clusters + 1] once finished
   // totals
   int64_t n_records = 0, n_decoded = 0, n_accepted = 0, n_pairs = 0, n_filtered = 0, mapped_reads = 0, n_chunks = 0, carry_max = 0, redecoded = 0;
   size_t free_at_begin = 0, min_free = 0;
   bool ended = false;
   double t_wait = 0, t_compute = 0, t_enqueue = 0, t_last = 0; // seconds: waiting for uploads, computing chunks, enqueuing uploads, the last stage
   void note_memory()
   {
      size_t fr = 0, tot = 0;
      if (hipMemGetInfo(&fr, &tot) == hipSuccess) min_free = std::min(min_free, fr);
   }
};

namespace {

#define SB_TRY(expr)                                                                                                 \
   do {                                                                                                              \
      hipError_t e_ = (expr);                                                                                        \
      if (e_ != hipSuccess)                                                                                          \
         return api_fail(e_ == hipErrorOutOfMemory ? SBGPU_ENOMEM : SBGPU_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
   } while (0)
#define SB_RC(expr)                    \
   do {                                \
      const int rc_ = (expr);          \
      if (rc_ != SBGPU_OK) return rc_; \
   } while (0)

// the chunk in buffer `b` (uploaded), with the carry in front of it: decode .. unique hits of the clusters it completes
int compute_pending(sbgpu_front_stream *F, bool last)
{
   sbgpu_ctx_t *c = F->ctx;
   hipStream_t s = sb::ctx_stream(c);
   // every stage below synchronises the stream it ran on before it returns, so the arenas its handles give back are idle: they
   // must not wait for the DEVICE -- the next chunk's upload is running on it
   sb::DevGiveStreamSynced no_device_wait;
   const int b = F->cur;
   auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
   const double t_in = now();
   SB_TRY(hipEventSynchronize(F->ev_up[b])); // the chunk has arrived
   const double t_arrived = now();
   F->t_wait += t_arrived - t_in;
   struct Clock {
      double &acc, t0;
      ~Clock() { acc += std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count() - t0; }
   } clock = {F->t_compute, t_arrived};
   // the window: [carry | chunk], its record offsets
   const int64_t n_carry = F->carry_n;
   const int64_t n_new = F->pending ? F->pend_records : 0;
   const int64_t n_rec = n_carry + n_new, w_bytes = F->carry_bytes + (F->pending ? F->pend_bytes : 0);
   char *w0 = F->buf[b] + F->chunk_cap - F->carry_bytes; // (the carry was copied to end where the chunk begins)
   F->n_chunks += F->pending ? 1 : 0;
   F->redecoded += n_carry;
   if (n_rec == 0) {
      F->pending = false;
      return SBGPU_OK;
   }
   // (made on the device from the window before and the chunk's own offsets: 8 bytes per record never pass through host loops)
   Grow &W = F->d_win[F->win], &P = F->d_win[F->win ^ 1];
   W.used = 0;
   SB_TRY(W.reserve((size_t)(n_rec + 1) * 8, s));
   hipLaunchKernelGGL(window_off_kernel, dim3((unsigned)((n_rec + 1 + 255) / 256)), dim3(256), 0, s, (int64_t *)W.p,
                      (const int64_t *)P.p + F->carry_rec, n_carry, F->carry_c0, (const int64_t *)F->d_pend[b].p, n_new, F->carry_bytes);
   SB_TRY(hipGetLastError());
   sbgpu_bamreads_t *R = nullptr;
   SB_RC(sbgpu_bam_decode_device(c, (const uint8_t *)w0, w_bytes, (const int64_t *)W.p, n_rec, &F->opts, s, &R));
   struct ReadsGuard {
      sbgpu_bamreads_t *r;
      ~ReadsGuard() { sbgpu_bamreads_destroy(r); }
   } rg = {R};
   int64_t binfo[16];
   SB_RC(sbgpu_bamreads_info(R, binfo));
   sbgpu_reads_t reads;
   const int32_t *d_ref = nullptr;
   const uint32_t *d_left = nullptr, *d_right = nullptr;
   SB_RC(sbgpu_bamreads_reads(R, &reads, &d_ref, &d_left, &d_right));
   const int64_t n_reads = reads.n_reads;
   // which of the remaining clusters every read is offered to
   const int64_t nc = F->n_clusters - F->k0;
   std::vector<int64_t> roff((size_t)nc + 1, 0);
   char *d_cluster = nullptr;
   size_t d_cluster_cap = 0;
   SB_TRY(sb::dev_take((size_t)std::max<int64_t>(n_reads, 1) * 4, &d_cluster, &d_cluster_cap));
   struct Give {
      char *p;
      size_t cap;
      ~Give() { sb::dev_give(p, cap); }
   } gc = {d_cluster, d_cluster_cap};
   sbgpu_clusters_t cl = {nc, F->c_ref.data() + F->k0, F->c_left.data() + F->k0, F->c_right.data() + F->k0, F->c_strand.data() + F->k0};
   if (nc > 0)
      SB_RC(sbgpu_assign_reads_device(c, &cl, n_reads, d_ref, d_left, d_right, (uint8_t *)reads.flags, (int32_t *)d_cluster, roff.data(), s));
   // complete: a read behind the cluster's end has been seen (or nothing more will come)
   int64_t n_done = 0;
   if (last) n_done = nc;
   else
      while (n_done < nc && roff[(size_t)n_done + 1] < n_reads) ++n_done;
   const int64_t reads_done = nc > 0 ? roff[(size_t)n_done] : n_reads;
   // ---- the carry for the next window: from the record of the first read that was not consumed
   int64_t carry_rec = n_rec, c0 = w_bytes; // (window record index; its first byte in the window)
   if (!last && reads_done < n_reads) {
      const int64_t *d_record = sb::bamreads_device_record(R);
      SB_TRY(hipMemcpyAsync(&carry_rec, d_record + reads_done, 8, hipMemcpyDeviceToHost, s));
      SB_TRY(hipStreamSynchronize(s));
      SB_TRY(hipMemcpyAsync(&c0, (const int64_t *)W.p + carry_rec, 8, hipMemcpyDeviceToHost, s));
      SB_TRY(hipStreamSynchronize(s));
   }
   if (n_done > 0 && reads_done > 0) {
      sbgpu_reads_t part = reads;
      part.n_reads = reads_done;
      sbgpu_matepairs_t *M = nullptr;
      SB_RC(sbgpu_pair_mates_device(c, n_done, &part, roff.data(), s, &M));
      struct MG {
         sbgpu_matepairs_t *m;
         ~MG() { sbgpu_matepairs_destroy(m); }
      } mg = {M};
      sbgpu_pairs_t pairs;
      const int64_t *poff = nullptr;
      SB_RC(sbgpu_matepairs_pairs(M, &pairs, &poff));
      F->n_pairs += pairs.n_pairs;
      sbgpu_uniq_dev_t *U = nullptr;
      SB_RC(sbgpu_collapse_pairs_device(c, n_done, &pairs, poff, s, &U));
      struct UG {
         sbgpu_uniq_dev_t *u;
         ~UG() { sbgpu_uniq_dev_destroy(u); }
      } ug = {U};
      int64_t ui[8];
      SB_RC(sbgpu_uniq_dev_info(U, ui));
      sbgpu_hits_t h;
      const float *d_mass = nullptr;
      const int64_t *hoff = nullptr;
      SB_RC(sbgpu_uniq_dev_hits(U, &h, &d_mass, &hoff));
      const int64_t nh = ui[0], nf = ui[1];
      F->n_filtered += ui[2], F->mapped_reads += ui[4];
      // append to the store: the cluster numbers and the feature offsets shifted to the stream's
      SB_TRY(F->s_locus.reserve((size_t)(F->n_hits + nh) * 4, s));
      SB_TRY(F->s_foff.reserve((size_t)(F->n_hits + nh + 1) * 8, s));
      SB_TRY(F->s_mass.reserve((size_t)(F->n_hits + nh) * 4, s));
      SB_TRY(F->s_code.reserve((size_t)(F->n_feat + nf), s));
      SB_TRY(F->s_left.reserve((size_t)(F->n_feat + nf) * 4, s));
      SB_TRY(F->s_right.reserve((size_t)(F->n_feat + nf) * 4, s));
      if (nh) {
         int32_t *dl = (int32_t *)F->s_locus.p + F->n_hits;
         int64_t *df = (int64_t *)F->s_foff.p + F->n_hits;
         SB_TRY(hipMemcpyAsync(dl, h.hit_locus, (size_t)nh * 4, hipMemcpyDeviceToDevice, s));
         SB_TRY(hipMemcpyAsync(df, h.feat_off, (size_t)(nh + 1) * 8, hipMemcpyDeviceToDevice, s));
         SB_TRY(hipMemcpyAsync((float *)F->s_mass.p + F->n_hits, d_mass, (size_t)nh * 4, hipMemcpyDeviceToDevice, s));
         if (nf) {
            SB_TRY(hipMemcpyAsync(F->s_code.p + F->n_feat, h.feat_code, (size_t)nf, hipMemcpyDeviceToDevice, s));
            SB_TRY(hipMemcpyAsync((uint32_t *)F->s_left.p + F->n_feat, h.feat_left, (size_t)nf * 4, hipMemcpyDeviceToDevice, s));
            SB_TRY(hipMemcpyAsync((uint32_t *)F->s_right.p + F->n_feat, h.feat_right, (size_t)nf * 4, hipMemcpyDeviceToDevice, s));
         }
         if (F->k0) hipLaunchKernelGGL(add_i32_kernel, dim3((unsigned)((nh + 255) / 256)), dim3(256), 0, s, dl, nh, (int32_t)F->k0);
         if (F->n_feat) hipLaunchKernelGGL(add_i64_kernel, dim3((unsigned)((nh + 1 + 255) / 256)), dim3(256), 0, s, df, nh + 1, F->n_feat);
         SB_TRY(hipGetLastError());
      }
      for (int64_t k = 0; k < n_done; ++k) F->locus_hit_off[(size_t)(F->k0 + k + 1)] = F->n_hits + hoff[(size_t)k + 1];
      F->n_hits += nh, F->n_feat += nf;
      F->s_locus.used = (size_t)F->n_hits * 4, F->s_foff.used = (size_t)(F->n_hits + 1) * 8, F->s_mass.used = (size_t)F->n_hits * 4;
      F->s_code.used = (size_t)F->n_feat, F->s_left.used = F->s_right.used = (size_t)F->n_feat * 4;
      SB_TRY(hipStreamSynchronize(s)); // (the handles go away with this scope)
   } else {
      for (int64_t k = 0; k < n_done; ++k) F->locus_hit_off[(size_t)(F->k0 + k + 1)] = F->n_hits;
   }
   F->n_records += n_new, F->n_decoded += n_rec, F->n_accepted += nc > 0 ? reads_done : 0;
   F->k0 += n_done;
   // ---- carry: the window's records from carry_rec on, in front of where the next chunk lands in the OTHER buffer
   const int64_t cb = w_bytes - c0;
   if (cb > F->chunk_cap)
      return api_fail(SBGPU_ESHAPE, "sbgpu_front_stream_push: the records of one cluster exceed a chunk's capacity (" + std::to_string(cb) + " bytes): begin the stream with larger chunks");
   F->carry_n = n_rec - carry_rec, F->carry_rec = carry_rec, F->carry_c0 = c0;
   F->win ^= 1; // (this window's offsets are the next one's "window before")
   F->carry_bytes = cb;
   F->carry_max = std::max(F->carry_max, cb);
   if (cb) {
      SB_TRY(hipMemcpyAsync(F->buf[b ^ 1] + F->chunk_cap - cb, w0 + c0, (size_t)cb, hipMemcpyDeviceToDevice, s));
      SB_TRY(hipStreamSynchronize(s));
   }
   F->pending = false;
   F->note_memory();
   return SBGPU_OK;
}

} // namespace

extern "C" {

int sbgpu_front_stream_begin(sbgpu_ctx_t *c, const sbgpu_clusters_t *cl, const sbgpu_bam_opts_t *opts, int64_t chunk_bytes,
                             sbgpu_front_stream_t **out)
{
   if (!c || !cl || !opts || !out) return api_fail(SBGPU_EINVAL, "sbgpu_front_stream_begin: null argument");
   *out = nullptr;
   if (cl->n_clusters < 0 || (cl->n_clusters && (!cl->ref || !cl->left || !cl->right || !cl->strand)))
      return api_fail(SBGPU_EINVAL, "sbgpu_front_stream_begin: bad clusters");
   if (chunk_bytes < (1 << 16) || chunk_bytes > ((int64_t)1 << 36)) return api_fail(SBGPU_EINVAL, "sbgpu_front_stream_begin: a chunk holds 64 KB .. 64 GB");
   sbgpu_front_stream *F = new (std::nothrow) sbgpu_front_stream();
   if (!F) return api_fail(SBGPU_ENOMEM, "sbgpu_front_stream_begin: out of host memory");
   F->ctx = c, F->device = sb::ctx_device(c);
   F->n_clusters = cl->n_clusters;
   F->c_ref.assign(cl->ref, cl->ref + cl->n_clusters);
   F->c_left.assign(cl->left, cl->left + cl->n_clusters);
   F->c_right.assign(cl->right, cl->right + cl->n_clusters);
   F->c_strand.assign(cl->strand, cl->strand + cl->n_clusters);
   F->opts = *opts;
   F->chunk_cap = (int64_t)up256((size_t)chunk_bytes);
   F->locus_hit_off.assign((size_t)cl->n_clusters + 1, 0);
   hipError_t e = hipSetDevice(F->device);
   size_t tot = 0;
   if (e == hipSuccess) e = hipMemGetInfo(&F->free_at_begin, &tot);
   F->min_free = F->free_at_begin;
   if (e == hipSuccess) {
      int least = 0, greatest = 0;
      e = hipDeviceGetStreamPriorityRange(&least, &greatest);
      if (e == hipSuccess) e = hipStreamCreateWithPriority(&F->up_stream, hipStreamNonBlocking, least);
   }
   for (int b = 0; b < 2 && e == hipSuccess; ++b) {
      e = sb::dev_take((size_t)F->chunk_cap * 2, &F->buf[b], &F->buf_cap[b]); // [carry room | chunk]
      if (e == hipSuccess) e = hipEventCreateWithFlags(&F->ev_up[b], hipEventDisableTiming);
   }
   if (e != hipSuccess) {
      sbgpu_front_stream_destroy(F);
      return api_fail(e == hipErrorOutOfMemory ? SBGPU_ENOMEM : SBGPU_EHIP, std::string("sbgpu_front_stream_begin: ") + hipGetErrorString(e));
   }
   *out = F;
   return SBGPU_OK;
}

int sbgpu_front_stream_push(sbgpu_front_stream_t *F, const uint8_t *bytes, int64_t n_bytes, const int64_t *rec_off, int64_t n_records)
{
   if (!F || F->ended) return api_fail(SBGPU_EINVAL, "sbgpu_front_stream_push: no stream (or it has ended)");
   if (n_bytes < 0 || (n_bytes && !bytes) || n_bytes > F->chunk_cap) return api_fail(SBGPU_EINVAL, "sbgpu_front_stream_push: a chunk holds at most the bytes the stream was begun with");
   SB_TRY(hipSetDevice(F->device));
   // the chunk's record offsets: the caller's (noted while inflating), or found here
   std::vector<int64_t> found;
   if (rec_off) {
      if (n_records < 0 || rec_off[0] != 0 || rec_off[n_records] != n_bytes) return api_fail(SBGPU_EINVAL, "sbgpu_front_stream_push: rec_off must run from 0 to n_bytes (whole records only)");
   } else {
      found.resize((size_t)(n_bytes / 36 + 2));
      n_records = sbgpu_bam_index_host(bytes, n_bytes, found.data(), (int64_t)found.size() - 1);
      if (n_records < 0) return SBGPU_ESHAPE;
      rec_off = found.data();
   }
   // this chunk starts its way to the device (into the buffer the pending chunk does NOT use), its offsets in front of it ...
   const double t_push = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
   const int nb = F->pending ? F->cur ^ 1 : F->cur;
   hipStream_t cs = F->up_stream;
   F->d_pend[nb].used = 0;
   SB_TRY(F->d_pend[nb].reserve((size_t)(n_records + 1) * 8, cs));
   SB_TRY(hipMemcpyAsync(F->d_pend[nb].p, rec_off, (size_t)(n_records + 1) * 8, hipMemcpyHostToDevice, cs));
   if (!found.empty()) SB_TRY(hipStreamSynchronize(cs)); // (the offsets found here live in this frame)
   if (n_bytes) SB_TRY(hipMemcpyAsync(F->buf[nb] + F->chunk_cap, bytes, (size_t)n_bytes, hipMemcpyHostToDevice, cs));
   SB_TRY(hipEventRecord(F->ev_up[nb], cs));
   F->t_enqueue += std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count() - t_push;
   // ... while the pending one is computed (its carry lands in front of this chunk, in buffer nb)
   if (F->pending) SB_RC(compute_pending(F, false));
   else if (F->carry_bytes) return api_fail(SBGPU_EHIP, "sbgpu_front_stream_push: internal: a carry without a pending chunk");
   F->cur = nb;
   F->pending = true;
   F->pend_bytes = n_bytes;
   F->pend_records = n_records;
   return SBGPU_OK;
}

int sbgpu_front_stream_end(sbgpu_front_stream_t *F, const sbgpu_annotation_t *an, const sbgpu_insert_t *insert, int32_t read_len, int32_t long_read,
                           const sbgpu_abundance_params_t *params, sbgpu_comm_t *comm, sbgpu_insert_t *insert_used, sbgpu_abundances_t *out,
                           sbgpu_bins_t **bins_out)
{
   if (!F || F->ended) return api_fail(SBGPU_EINVAL, "sbgpu_front_stream_end: no stream (or it has ended)");
   if (!an || !params || !out || !bins_out) return api_fail(SBGPU_EINVAL, "sbgpu_front_stream_end: null argument");
   if (an->n_loci != F->n_clusters) return api_fail(SBGPU_EINVAL, "sbgpu_front_stream_end: the annotation's loci are the stream's clusters, one to one");
   SB_TRY(hipSetDevice(F->device));
   if (!F->pending) { // nothing was pushed (or an empty last chunk): the carry alone, if any
      SB_TRY(hipEventRecord(F->ev_up[F->cur], sb::ctx_stream(F->ctx)));
   }
   SB_RC(compute_pending(F, true));
   F->ended = true;
   // the chunk buffers are not needed any more: the last stage gets their room
   for (int b = 0; b < 2; ++b) {
      sb::dev_give(F->buf[b], F->buf_cap[b]);
      F->buf[b] = nullptr, F->buf_cap[b] = 0;
   }
   for (int b = 0; b < 2; ++b) F->d_win[b].release(), F->d_pend[b].release();
   (void)sbgpu_release_idle_memory();
   hipStream_t s = sb::ctx_stream(F->ctx);
   SB_TRY(F->s_foff.reserve(8, s));
   if (F->n_hits == 0) SB_TRY(hipMemsetAsync(F->s_foff.p, 0, 8, s));
   sbgpu_hits_t h = {F->n_hits, (const int32_t *)F->s_locus.p, (const int64_t *)F->s_foff.p, (const uint8_t *)F->s_code.p,
                     (const uint32_t *)F->s_left.p, (const uint32_t *)F->s_right.p};
   const double t_q = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
   const int rc = sbgpu_quantify_resident(F->ctx, an, &h, (const float *)F->s_mass.p, F->locus_hit_off.data(), insert, read_len, long_read,
                                          F->mapped_reads, params, comm, insert_used, out, bins_out);
   F->t_last = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count() - t_q;
   F->note_memory();
   if (std::getenv("SBGPU_HOST_TIMING")) // diagnostic: where the pass' time went, on stderr
      std::fprintf(stderr, "sbgpu_front_stream: %lld chunks | waiting for uploads %.1f ms | computing chunks %.1f ms | enqueuing uploads %.1f ms | last stage %.1f ms\n",
                   (long long)F->n_chunks, F->t_wait * 1e3, F->t_compute * 1e3, F->t_enqueue * 1e3, F->t_last * 1e3);
   return rc;
}

int sbgpu_front_stream_info(const sbgpu_front_stream_t *F, int64_t info[16])
{
   if (!F || !info) return api_fail(SBGPU_EINVAL, "sbgpu_front_stream_info: null argument");
   const int64_t v[16] = {F->n_records, F->n_accepted, F->n_pairs, F->n_hits, F->n_feat, F->n_filtered, F->mapped_reads, F->n_chunks,
                          F->k0, F->carry_max, F->redecoded, (int64_t)F->min_free, F->chunk_cap, F->ended ? 1 : 0, (int64_t)F->free_at_begin, 0};
   std::memcpy(info, v, sizeof(v));
   return SBGPU_OK;
}

int sbgpu_front_stream_hits(const sbgpu_front_stream_t *F, sbgpu_hits_t *d_hits, const float **d_hit_mass, const int64_t **locus_hit_off)
{
   if (!F) return api_fail(SBGPU_EINVAL, "sbgpu_front_stream_hits: null stream");
   if (d_hits)
      *d_hits = {F->n_hits, (const int32_t *)F->s_locus.p, (const int64_t *)F->s_foff.p, (const uint8_t *)F->s_code.p, (const uint32_t *)F->s_left.p,
                 (const uint32_t *)F->s_right.p};
   if (d_hit_mass) *d_hit_mass = (const float *)F->s_mass.p;
   if (locus_hit_off) *locus_hit_off = F->locus_hit_off.data();
   return SBGPU_OK;
}

void sbgpu_front_stream_destroy(sbgpu_front_stream_t *F)
{
   if (!F) return;
   (void)hipSetDevice(F->device);
   (void)hipDeviceSynchronize(); // (an upload may still be in flight)
   for (int b = 0; b < 2; ++b) {
      sb::dev_give(F->buf[b], F->buf_cap[b]);
      if (F->ev_up[b]) (void)hipEventDestroy(F->ev_up[b]);
   }
   if (F->up_stream) (void)hipStreamDestroy(F->up_stream);
   for (int b = 0; b < 2; ++b) F->d_win[b].release(), F->d_pend[b].release();
   for (Grow *g : {&F->s_locus, &F->s_foff, &F->s_code, &F->s_left, &F->s_right, &F->s_mass}) g->release();
   delete F;
}

} // extern "C"
