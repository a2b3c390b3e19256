// strawberry_amd/csrc/bamdecode_api.hip -- sbgpu_bam_index_host / sbgpu_bam_decode_host / _device (include/sbgpu.h): BAM
// alignment records -> the read stream, BAMHitFactory::getHitFromBuf (/root/reference/src/read.cpp:480-715).  The decoder
// body and the two kernels: bamdecode_device.h.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

#include <rocprim/rocprim.hpp>

#include "../../include/sbgpu.h"
#include "api_internal.h"
#include "bamdecode_device.h"

using sb::api_fail;

struct sbgpu_bamreads {
   bool on_device = false;
   int device = 0;
   int64_t n_records = 0, n_reads = 0, n_blocks = 0, any_paired = 0;
   int64_t by_status[11] = {};
   std::vector<char> host; // host form: one buffer
   char *arena = nullptr;  // device form: one arena (sb::dev_take'n)
   size_t arena_cap = 0;
   // into the buffer / arena
   uint8_t *status = nullptr;   // [n_records]
   int64_t *record = nullptr;   // [n_reads] ...
   uint64_t *read_id = nullptr;
   int32_t *ref = nullptr, *nh = nullptr, *nm = nullptr, *read_len = nullptr;
   uint32_t *left = nullptr, *right = nullptr, *partner_pos = nullptr, *sam_flag = nullptr;
   uint8_t *flags = nullptr;
   int64_t *block_off = nullptr; // [n_reads + 1]
   uint32_t *block_left = nullptr, *block_right = nullptr; // [n_blocks]
};

namespace sb {
const int64_t *bamreads_device_record(const sbgpu_bamreads_t *b) { return b && b->on_device ? b->record : nullptr; }
} // namespace sb

namespace {
size_t up256(size_t b) { return (b + 255) & ~(size_t)255; }

// the handle's arrays inside one block of memory; returns its size
size_t lay_out(sbgpu_bamreads *B, char *base)
{
   size_t off = 0;
   auto take = [&](size_t bytes) {
      char *p = base ? base + off : nullptr;
      off += up256(bytes ? bytes : 8);
      return p;
   };
   const size_t n = (size_t)B->n_records, m = (size_t)B->n_reads, nb = (size_t)B->n_blocks;
   B->status = (uint8_t *)take(n);
   B->record = (int64_t *)take(m * 8);
   B->read_id = (uint64_t *)take(m * 8);
   B->ref = (int32_t *)take(m * 4), B->nh = (int32_t *)take(m * 4), B->nm = (int32_t *)take(m * 4), B->read_len = (int32_t *)take(m * 4);
   B->left = (uint32_t *)take(m * 4), B->right = (uint32_t *)take(m * 4), B->partner_pos = (uint32_t *)take(m * 4);
   B->sam_flag = (uint32_t *)take(m * 4);
   B->flags = (uint8_t *)take(m);
   B->block_off = (int64_t *)take((m + 1) * 8);
   B->block_left = (uint32_t *)take(nb * 4), B->block_right = (uint32_t *)take(nb * 4);
   return off;
}

int check_opts(const sbgpu_bam_opts_t *o, const char *who)
{
   if (!o) return api_fail(SBGPU_EINVAL, std::string(who) + ": null options");
   if (o->library < 0 || o->library > 2) return api_fail(SBGPU_EINVAL, std::string(who) + ": library must be 0 (unstranded), 1 (fr) or 2 (rf)");
   return SBGPU_OK;
}
} // namespace

extern "C" {

int64_t sbgpu_bam_index_host(const uint8_t *bytes, int64_t n_bytes, int64_t *rec_off, int64_t cap)
{
   if ((!bytes && n_bytes) || !rec_off || n_bytes < 0 || cap < 0) {
      api_fail(SBGPU_EINVAL, "sbgpu_bam_index_host: bad argument");
      return -1;
   }
   int64_t n = 0, p = 0;
   while (p < n_bytes) {
      if (p + 4 > n_bytes) {
         api_fail(SBGPU_ESHAPE, "sbgpu_bam_index_host: the stream ends inside a record's size word");
         return -1;
      }
      const int32_t bs = sb::bam_i32(bytes + p);
      if (bs < 0 || p + 4 + (int64_t)bs > n_bytes) {
         api_fail(SBGPU_ESHAPE, "sbgpu_bam_index_host: record " + std::to_string(n) + " runs past the end of the stream");
         return -1;
      }
      if (n >= cap) {
         api_fail(SBGPU_ESHAPE, "sbgpu_bam_index_host: more than `cap` records");
         return -1;
      }
      rec_off[n++] = p;
      p += 4 + (int64_t)bs;
   }
   rec_off[n] = p;
   return n;
}

void sbgpu_bamreads_destroy(sbgpu_bamreads_t *b)
{
   if (!b) return;
   sb::dev_give(b->arena, b->arena_cap);
   delete b;
}

int sbgpu_bam_decode_host(const uint8_t *bytes, int64_t n_bytes, const int64_t *rec_off, int64_t n, const sbgpu_bam_opts_t *opts,
                          sbgpu_bamreads_t **out)
{
   if (!out || n < 0 || n_bytes < 0 || (n && (!bytes || !rec_off))) return api_fail(SBGPU_EINVAL, "sbgpu_bam_decode_host: bad argument");
   *out = nullptr;
   if (const int rc = check_opts(opts, "sbgpu_bam_decode_host")) return rc;
   for (int64_t r = 0; r < n; ++r)
      if (rec_off[r] < 0 || rec_off[r + 1] < rec_off[r] || rec_off[r + 1] > n_bytes)
         return api_fail(SBGPU_EINVAL, "sbgpu_bam_decode_host: rec_off is not an ascending list of offsets inside the stream");
   sbgpu_bamreads *B = new (std::nothrow) sbgpu_bamreads();
   if (!B) return api_fail(SBGPU_ENOMEM, "sbgpu_bam_decode_host: out of memory");
   try {
      // Two passes over the records on host threads (SBGPU_HOST_THREADS, default min(16, hardware threads)), each thread a
      // contiguous range: (A) decide and count, (B) decode again and write -- nothing is kept per record between them.
      B->n_records = n;
      unsigned nt = std::thread::hardware_concurrency();
      nt = nt ? std::min(nt, 16u) : 4u;
      if (const char *e = std::getenv("SBGPU_HOST_THREADS")) nt = (unsigned)std::atoi(e);
      nt = (unsigned)std::max<int64_t>(1, std::min<int64_t>((int64_t)nt, (n + 4095) / 4096));
      struct Part {
         int64_t r0 = 0, r1 = 0, reads = 0, blocks = 0, paired = 0, by_status[11] = {};
      };
      std::vector<Part> part(nt);
      for (unsigned t = 0; t < nt; ++t) part[t].r0 = n * t / nt, part[t].r1 = n * (t + 1) / nt;
      // (a thread that cannot be started -- std::system_error -- must not leave joinable threads behind in a vector
      // that is being destroyed: its part and the ones after it run on the calling thread, what was started is joined)
      auto run = [&](auto &&fn) {
         std::vector<std::thread> pool;
         unsigned started = 1;
         try {
            for (; started < nt; ++started) pool.emplace_back(fn, started);
         } catch (const std::system_error &) {
         }
         fn(0u);
         for (unsigned t = started; t < nt; ++t) fn(t);
         for (std::thread &th : pool) th.join();
      };
      run([&](unsigned t) {
         Part &p = part[t];
         sb::BamRead x;
         for (int64_t r = p.r0; r < p.r1; ++r) {
            sb::bam_decode_record(bytes + rec_off[r], rec_off[r + 1] - rec_off[r], *opts, x);
            ++p.by_status[x.status <= SBGPU_BAM_TRUNCATED ? x.status : SBGPU_BAM_TRUNCATED];
            p.paired |= x.paired;
            if (x.status == SBGPU_BAM_OK) ++p.reads, p.blocks += x.n_blocks;
         }
      });
      std::vector<int64_t> read_base(nt + 1, 0), block_base(nt + 1, 0);
      for (unsigned t = 0; t < nt; ++t) {
         read_base[t + 1] = read_base[t] + part[t].reads;
         block_base[t + 1] = block_base[t] + part[t].blocks;
         B->any_paired |= part[t].paired;
         for (int k = 0; k <= SBGPU_BAM_TRUNCATED; ++k) B->by_status[k] += part[t].by_status[k];
      }
      B->n_reads = read_base[nt], B->n_blocks = block_base[nt];
      B->host.resize(lay_out(B, nullptr));
      lay_out(B, B->host.data());
      run([&](unsigned t) {
         const Part &p = part[t];
         int64_t k = read_base[t], b = block_base[t];
         sb::BamRead x;
         for (int64_t r = p.r0; r < p.r1; ++r) {
            sb::bam_decode_record(bytes + rec_off[r], rec_off[r + 1] - rec_off[r], *opts, x);
            B->status[r] = x.status;
            if (x.status != SBGPU_BAM_OK) continue;
            B->record[k] = r, B->read_id[k] = x.read_id, B->ref[k] = x.ref, B->nh[k] = x.nh, B->nm[k] = x.nm, B->read_len[k] = x.read_len;
            B->left[k] = x.left, B->right[k] = x.right, B->partner_pos[k] = x.partner_pos, B->sam_flag[k] = x.sam_flag, B->flags[k] = x.flags;
            B->block_off[k] = b;
            sb::bam_record_blocks(bytes + rec_off[r], B->block_left + b, B->block_right + b);
            b += x.n_blocks;
            ++k;
         }
      });
      B->block_off[B->n_reads] = B->n_blocks;
   } catch (const std::bad_alloc &) {
      delete B;
      return api_fail(SBGPU_ENOMEM, "sbgpu_bam_decode_host: out of memory");
   } catch (const std::system_error &e) {
      delete B;
      return api_fail(SBGPU_ENOMEM, std::string("sbgpu_bam_decode_host: ") + e.what());
   }
   *out = B;
   return SBGPU_OK;
}

int sbgpu_bam_decode_device(sbgpu_ctx_t *c, const uint8_t *d_bytes, int64_t n_bytes, const int64_t *d_rec_off, int64_t n,
                            const sbgpu_bam_opts_t *opts, void *stream, sbgpu_bamreads_t **out)
{
   if (!c || !out || n < 0 || n_bytes < 0 || (n && (!d_bytes || !d_rec_off))) return api_fail(SBGPU_EINVAL, "sbgpu_bam_decode_device: bad argument");
   *out = nullptr;
   if (const int rc = check_opts(opts, "sbgpu_bam_decode_device")) return rc;
   if (n >= ((int64_t)1 << 31) - 2) return api_fail(SBGPU_EUNSUPPORTED, "sbgpu_bam_decode_device: more than 2^31 records in one call; split the stream");
   sbgpu_bamreads *B = new (std::nothrow) sbgpu_bamreads();
   if (!B) return api_fail(SBGPU_ENOMEM, "sbgpu_bam_decode_device: out of memory");
   B->on_device = true;
   B->device = sb::ctx_device(c);
   B->n_records = n;
   hipStream_t s = stream ? (hipStream_t)stream : sb::ctx_stream(c);
   char *w = nullptr;
   size_t w_cap = 0;
   auto bail = [&](int code, const std::string &msg) {
      (void)hipStreamSynchronize(s);
      sb::dev_give(w, w_cap);
      sbgpu_bamreads_destroy(B);
      return api_fail(code, msg);
   };
#define SB_TRY(expr)                                                                                     \
   do {                                                                                                  \
      hipError_t e_ = (expr);                                                                            \
      if (e_ != hipSuccess) return bail(e_ == hipErrorOutOfMemory ? SBGPU_ENOMEM : SBGPU_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
   } while (0)
   SB_TRY(hipSetDevice(B->device));
   if (n == 0) {
      *out = B;
      return SBGPU_OK;
   }
   // what a workgroup may ask for on this device
   int lds_max = 0;
   if (hipDeviceGetAttribute(&lds_max, hipDeviceAttributeMaxSharedMemoryPerBlock, sb::ctx_device(c)) != hipSuccess || lds_max <= 0) lds_max = 64 * 1024;
   // ---- one pass (bam_onepass_kernel): the handle's arrays get room for EVERY record and two blocks per record; a sample
   // whose accepted records hold more blocks than that (or more than 2^30 records, or a look-back that ran out) is decoded
   // by the two kernels below instead
   const bool two_pass_only = std::getenv("SBGPU_BAM_TWO_PASS") && std::atoi(std::getenv("SBGPU_BAM_TWO_PASS")) != 0; // (the tests' hook for the fallback)
   if (!two_pass_only && n <= sb::kOneMaxRecords) {
      const int64_t block_cap = 2 * n + 1024;
      B->n_reads = n, B->n_blocks = block_cap;
      SB_TRY(sb::dev_take(lay_out(B, nullptr), &B->arena, &B->arena_cap));
      lay_out(B, B->arena);
      const size_t tiles = ((size_t)n + sb::kOneTile - 1) / sb::kOneTile;
      const size_t o_state = 0, o_cnt1 = up256(tiles * 8), o_ticket = o_cnt1 + up256(16 * 8), w_bytes = o_ticket + 256;
      SB_TRY(sb::dev_take(w_bytes, &w, &w_cap));
      SB_TRY(hipMemsetAsync(w, 0, w_bytes, s));
      sb::BamOneArgs a = {};
      a.bytes = d_bytes, a.n_bytes = n_bytes, a.rec_off = d_rec_off, a.n = n, a.opts = *opts;
      a.status = B->status;
      a.o_record = B->record, a.o_read_id = B->read_id, a.o_ref = B->ref, a.o_nh = B->nh, a.o_nm = B->nm, a.o_read_len = B->read_len;
      a.o_left = B->left, a.o_right = B->right, a.o_partner_pos = B->partner_pos, a.o_sam_flag = B->sam_flag, a.o_flags = B->flags;
      a.o_block_off = B->block_off, a.o_block_left = B->block_left, a.o_block_right = B->block_right, a.block_cap = block_cap;
      a.counts = (unsigned long long *)(w + o_cnt1), a.tile_state = (unsigned long long *)(w + o_state), a.ticket = (unsigned int *)(w + o_ticket);
      // a wave's part of the staging buffer: 64 average records + 4 % + 512 bytes, in steps of 256 (a pass whose records do
      // not fit walks them in global memory: slower, same results) -- three workgroups of four waves then fit a CU's 160 KB
      // up to ~195 bytes per record
      int stage_bytes = (int)std::min<int64_t>(std::max<int64_t>(4096, ((64 * (n_bytes / n + 1)) * 26 / 25 + 512 + 255) & ~(int64_t)255), 32 * 1024);
      stage_bytes = std::max(0, std::min(stage_bytes, (lds_max - 512) / sb::kOneWaves)) & ~255;
      const int resident = std::max(1, std::min(8, (int)(160 * 1024 / ((size_t)stage_bytes * sb::kOneWaves + 512))));
      const int64_t grid = std::min<int64_t>((int64_t)tiles, (int64_t)sb::ctx_cu_count(c) * resident);
      hipLaunchKernelGGL(sb::bam_onepass_kernel, dim3((unsigned)grid), dim3(64 * sb::kOneWaves), (size_t)stage_bytes * sb::kOneWaves, s, a, stage_bytes);
      // (a launch that is refused -- the staging buffer on a device with less LDS than it was told -- counts as a failure too)
      unsigned long long counts[16] = {};
      counts[12] = 1;
      if (hipGetLastError() == hipSuccess) {
         SB_TRY(hipMemcpyAsync(counts, w + o_cnt1, sizeof(counts), hipMemcpyDeviceToHost, s));
         SB_TRY(hipStreamSynchronize(s));
      }
      sb::dev_give(w, w_cap);
      w = nullptr, w_cap = 0;
      if (counts[12] == 0) {
         B->n_reads = (int64_t)counts[13], B->n_blocks = (int64_t)counts[14];
         for (int k = 0; k <= SBGPU_BAM_TRUNCATED; ++k) B->by_status[k] = (int64_t)counts[k];
         B->any_paired = counts[11] ? 1 : 0;
         *out = B;
         return SBGPU_OK;
      }
      sb::dev_give(B->arena, B->arena_cap);
      B->arena = nullptr, B->arena_cap = 0;
   }
   // ---- two passes: decide and count (bam_scan_kernel), scans over the waves' totals, compact (bam_fill_kernel)
   const size_t nn = (size_t)n, n1 = nn + 1, nt = (nn + 63) / 64, nt1 = nt + 1; // (tiles of 64 records: a wave's pass)
   auto as_i64_i32 = [] __device__(int32_t v) { return (int64_t)v; };
   size_t tmp_bytes = 0;
   {
      size_t b = 0;
      (void)rocprim::exclusive_scan(nullptr, b, rocprim::make_transform_iterator((const int32_t *)nullptr, as_i64_i32), (int64_t *)nullptr, (int64_t)0, nt1, rocprim::plus<int64_t>(), s);
      tmp_bytes = std::max(tmp_bytes, b);
   }
   size_t off = 0;
   auto take = [&](size_t bytes) {
      const size_t o = off;
      off += up256(bytes ? bytes : 8);
      return o;
   };
   const size_t o_status = take(nn), o_acc = take(n1), o_nb = take(n1 * 4), o_rid = take(nn * 8), o_ref = take(nn * 4), o_nh = take(nn * 4),
                o_nm = take(nn * 4), o_rl = take(nn * 4), o_left = take(nn * 4), o_right = take(nn * 4), o_pp = take(nn * 4), o_sf = take(nn * 4),
                o_fl = take(nn), o_ib = take(nn * 4 * 2 * sb::kBamInlineBlocks), o_tr = take(nt1 * 4), o_tb = take(nt1 * 4), o_rat = take(nt1 * 8), o_bat = take(nt1 * 8),
                o_cnt = take(16 * 8), o_tmp = take(tmp_bytes);
   SB_TRY(sb::dev_take(off, &w, &w_cap));
   SB_TRY(hipMemsetAsync(w + o_cnt, 0, 16 * 8, s));
   SB_TRY(hipMemsetAsync(w + o_tr + nt * 4, 0, 4, s)); // (the scans' entry beyond the last tile)
   SB_TRY(hipMemsetAsync(w + o_tb + nt * 4, 0, 4, s));
   sb::BamScanArgs a = {};
   a.bytes = d_bytes, a.n_bytes = n_bytes, a.rec_off = d_rec_off, a.n = n, a.opts = *opts;
   a.status = (uint8_t *)(w + o_status), a.n_blocks = (int32_t *)(w + o_nb), a.accepted = (uint8_t *)(w + o_acc);
   a.read_id = (uint64_t *)(w + o_rid), a.ref = (int32_t *)(w + o_ref), a.nh = (int32_t *)(w + o_nh), a.nm = (int32_t *)(w + o_nm);
   a.read_len = (int32_t *)(w + o_rl), a.left = (uint32_t *)(w + o_left), a.right = (uint32_t *)(w + o_right);
   a.partner_pos = (uint32_t *)(w + o_pp), a.sam_flag = (uint32_t *)(w + o_sf), a.flags = (uint8_t *)(w + o_fl);
   a.inline_blocks = (uint32_t *)(w + o_ib);
   a.counts = (unsigned long long *)(w + o_cnt);
   a.tile_reads = (int32_t *)(w + o_tr), a.tile_blocks = (int32_t *)(w + o_tb);
   const int64_t cap = (int64_t)sb::ctx_cu_count(c) * 32, blocks = std::min<int64_t>((n + 255) / 256, cap);
   // the staging buffer of a workgroup (one wave, 64 records): 1.1 x 64 average records, 8-64 KB
   int stage_bytes = 8 * 1024;
   while (stage_bytes < 64 * 1024 && (int64_t)stage_bytes < 70 * (n_bytes / n + 1)) stage_bytes += 2 * 1024;
   // (what a workgroup may ask for, less the kernel's static words; a launch that is refused all the same is repeated
   // without the buffer: the kernel then walks every record in global memory: slower, same results)
   stage_bytes = std::max(0, std::min(stage_bytes, lds_max - 256));
   const int resident = std::max(1, std::min(16, (int)(160 * 1024 / (stage_bytes + 256))));
   const int64_t scan_blocks = std::min<int64_t>((n + 63) / 64, (int64_t)sb::ctx_cu_count(c) * resident * 4);
   hipLaunchKernelGGL(sb::bam_scan_kernel, dim3((unsigned)scan_blocks), dim3(64), (size_t)stage_bytes, s, a, stage_bytes);
   if (hipGetLastError() != hipSuccess && stage_bytes > 0) {
      stage_bytes = 0;
      hipLaunchKernelGGL(sb::bam_scan_kernel, dim3((unsigned)scan_blocks), dim3(64), 0, s, a, 0);
   }
   SB_TRY(hipGetLastError());
   size_t tb = tmp_bytes;
   SB_TRY(rocprim::exclusive_scan(w + o_tmp, tb, rocprim::make_transform_iterator((const int32_t *)a.tile_reads, as_i64_i32), (int64_t *)(w + o_rat), (int64_t)0, nt1, rocprim::plus<int64_t>(), s));
   tb = tmp_bytes;
   SB_TRY(rocprim::exclusive_scan(w + o_tmp, tb, rocprim::make_transform_iterator((const int32_t *)a.tile_blocks, as_i64_i32), (int64_t *)(w + o_bat), (int64_t)0, nt1, rocprim::plus<int64_t>(), s));
   // the totals: the output arena's size depends on them
   int64_t totals[2] = {0, 0};
   unsigned long long counts[16];
   SB_TRY(hipMemcpyAsync(&totals[0], w + o_rat + nt * 8, 8, hipMemcpyDeviceToHost, s));
   SB_TRY(hipMemcpyAsync(&totals[1], w + o_bat + nt * 8, 8, hipMemcpyDeviceToHost, s));
   SB_TRY(hipMemcpyAsync(counts, w + o_cnt, sizeof(counts), hipMemcpyDeviceToHost, s));
   SB_TRY(hipStreamSynchronize(s));
   B->n_reads = totals[0], B->n_blocks = totals[1];
   for (int k = 0; k <= SBGPU_BAM_TRUNCATED; ++k) B->by_status[k] = (int64_t)counts[k];
   B->any_paired = counts[11] ? 1 : 0;
   SB_TRY(sb::dev_take(lay_out(B, nullptr), &B->arena, &B->arena_cap));
   lay_out(B, B->arena);
   SB_TRY(hipMemcpyAsync(B->status, w + o_status, nn, hipMemcpyDeviceToDevice, s));
   sb::BamFillArgs f = {};
   f.bytes = d_bytes, f.rec_off = d_rec_off, f.n = n;
   f.accepted = a.accepted, f.tile_read_at = (const int64_t *)(w + o_rat), f.tile_block_at = (const int64_t *)(w + o_bat);
   f.read_id = a.read_id, f.ref = a.ref, f.nh = a.nh, f.nm = a.nm, f.read_len = a.read_len;
   f.left = a.left, f.right = a.right, f.partner_pos = a.partner_pos, f.sam_flag = a.sam_flag, f.flags = a.flags;
   f.n_blocks = a.n_blocks, f.inline_blocks = a.inline_blocks;
   f.o_record = B->record, f.o_read_id = B->read_id, f.o_ref = B->ref, f.o_nh = B->nh, f.o_nm = B->nm, f.o_read_len = B->read_len;
   f.o_left = B->left, f.o_right = B->right, f.o_partner_pos = B->partner_pos, f.o_sam_flag = B->sam_flag, f.o_flags = B->flags;
   f.o_block_off = B->block_off, f.o_block_left = B->block_left, f.o_block_right = B->block_right;
   hipLaunchKernelGGL(sb::bam_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, s, f);
   SB_TRY(hipGetLastError());
   SB_TRY(hipStreamSynchronize(s)); // (the scratch goes back to the pool; the handle's arrays are complete when the call returns)
   sb::dev_give(w, w_cap);
#undef SB_TRY
   *out = B;
   return SBGPU_OK;
}

int sbgpu_bamreads_info(const sbgpu_bamreads_t *b, int64_t info[16])
{
   if (!b || !info) return api_fail(SBGPU_EINVAL, "sbgpu_bamreads_info: null argument");
   info[0] = b->n_records, info[1] = b->n_reads, info[2] = b->n_blocks, info[3] = b->any_paired, info[4] = b->on_device ? 1 : 0;
   for (int k = 0; k <= SBGPU_BAM_TRUNCATED; ++k) info[5 + k] = b->by_status[k];
   return SBGPU_OK;
}

int sbgpu_bamreads_reads(const sbgpu_bamreads_t *b, sbgpu_reads_t *reads, const int32_t **read_ref, const uint32_t **read_left,
                         const uint32_t **read_right)
{
   if (!b || !reads) return api_fail(SBGPU_EINVAL, "sbgpu_bamreads_reads: null argument");
   reads->n_reads = b->n_reads;
   reads->read_id = b->read_id;
   reads->block_off = b->block_off;
   reads->block_left = b->block_left, reads->block_right = b->block_right;
   reads->partner_pos = b->partner_pos;
   reads->flags = b->flags;
   reads->nh = b->nh;
   if (read_ref) *read_ref = b->ref;
   if (read_left) *read_left = b->left;
   if (read_right) *read_right = b->right;
   return SBGPU_OK;
}

int sbgpu_bamreads_export(const sbgpu_bamreads_t *b, uint8_t *status, int64_t *record, uint64_t *read_id, int32_t *ref, uint32_t *left,
                          uint32_t *right, uint32_t *partner_pos, uint8_t *flags, int32_t *nh, int32_t *nm, int32_t *read_len,
                          uint32_t *sam_flag, int64_t *block_off, uint32_t *block_left, uint32_t *block_right)
{
   if (!b) return api_fail(SBGPU_EINVAL, "sbgpu_bamreads_export: null handle");
   if (b->on_device) {
      const hipError_t e0 = hipSetDevice(b->device);
      if (e0 != hipSuccess) return api_fail(SBGPU_EHIP, std::string("sbgpu_bamreads_export: ") + hipGetErrorString(e0));
   }
   if (b->n_records == 0) {
      if (block_off) block_off[0] = 0;
      return SBGPU_OK;
   }
   auto put = [&](void *dst, const void *src, size_t bytes) -> hipError_t {
      if (!dst || !bytes) return hipSuccess;
      if (b->on_device) return hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost);
      std::memcpy(dst, src, bytes);
      return hipSuccess;
   };
   const size_t n = (size_t)b->n_records, m = (size_t)b->n_reads, nb = (size_t)b->n_blocks;
   hipError_t e = put(status, b->status, n);
   if (e == hipSuccess) e = put(record, b->record, m * 8);
   if (e == hipSuccess) e = put(read_id, b->read_id, m * 8);
   if (e == hipSuccess) e = put(ref, b->ref, m * 4);
   if (e == hipSuccess) e = put(left, b->left, m * 4);
   if (e == hipSuccess) e = put(right, b->right, m * 4);
   if (e == hipSuccess) e = put(partner_pos, b->partner_pos, m * 4);
   if (e == hipSuccess) e = put(flags, b->flags, m);
   if (e == hipSuccess) e = put(nh, b->nh, m * 4);
   if (e == hipSuccess) e = put(nm, b->nm, m * 4);
   if (e == hipSuccess) e = put(read_len, b->read_len, m * 4);
   if (e == hipSuccess) e = put(sam_flag, b->sam_flag, m * 4);
   if (e == hipSuccess) e = put(block_off, b->block_off, (m + 1) * 8);
   if (e == hipSuccess) e = put(block_left, b->block_left, nb * 4);
   if (e == hipSuccess) e = put(block_right, b->block_right, nb * 4);
   if (e != hipSuccess) return api_fail(SBGPU_EHIP, std::string("sbgpu_bamreads_export: ") + hipGetErrorString(e));
   return SBGPU_OK;
}

} // extern "C"
