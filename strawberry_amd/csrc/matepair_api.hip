// strawberry_amd/csrc/matepair_api.hip -- sbgpu_pair_mates_host / _device (include/sbgpu.h): alignment records ->
// read pairs, HitCluster::addOpenHit + addHit (/root/reference/src/alignments.cpp:423-655).  Kernels: matepair_device.h.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <unordered_map>
#include <vector>

#include <rocprim/rocprim.hpp>

#include "../../include/sbgpu.h"
#include "api_internal.h"
#include "matepair_device.h"
#include "matepair_flat.h"

using sb::api_fail;

struct sbgpu_matepairs {
   bool on_device = false;
   int device = 0;
   int64_t n_loci = 0, n_pairs = 0, n_complete = 0, n_single = 0, n_refused = 0, n_orphan = 0, n_lfeat = 0, n_rfeat = 0;
   std::vector<int64_t> locus_pair_off; // host
   // host form
   std::vector<int32_t> pair_locus;
   std::vector<double> pair_mass;
   std::vector<int64_t> left_off, right_off;
   std::vector<uint8_t> left_code, right_code;
   std::vector<uint32_t> left_left, left_right, right_left, right_right;
   // device form: one arena (sb::dev_take'n)
   char *arena = nullptr;
   size_t arena_cap = 0;
   double *d_mass = nullptr;
   int64_t *d_left_off = nullptr, *d_right_off = nullptr;
   uint8_t *d_left_code = nullptr, *d_right_code = nullptr;
   uint32_t *d_left_left = nullptr, *d_left_right = nullptr, *d_right_left = nullptr, *d_right_right = nullptr;
   bool positional = false;  // device form: the positional matching served the call (no sort)
   uint32_t why_sorted = 0;  // else: sb::kPosUnsorted | kPosConflict (0: forced)
};

namespace {
size_t up256(size_t b) { return (b + 255) & ~(size_t)255; }
constexpr int kMaxFragSpanHost = 1000000; // src/common.cpp:17
} // namespace

// ---- the flat form (matepair_flat.h): every cluster of the call at once
static int pair_mates_flat(sbgpu_ctx_t *c, int64_t n_loci, const sbgpu_reads_t *dr, const int64_t *locus_read_off, hipStream_t s,
                           sbgpu_matepairs *M)
{
   const int64_t nr = dr->n_reads;
   if (nr >= ((int64_t)1 << 31) - 2) return api_fail(SBGPU_EUNSUPPORTED, "sbgpu_pair_mates_device: more than 2^31 records in one call; split the call or use sbgpu_pair_mates_host");
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
   SB_TRY(hipSetDevice(M->device));
   const size_t n = (size_t)nr, n1 = n + 1, nl1 = (size_t)n_loci + 1;
   if (n_loci > ((int64_t)1 << 23)) return api_fail(SBGPU_EUNSUPPORTED, "sbgpu_pair_mates_device: more than 2^23 clusters in one call; split the call");
   // the first sort's key (flat_mate_keys_kernel): 32 bits = the number of a group of 2^g neighbouring clusters, then a hash
   // of (cluster, read id) of two bits more than the logarithm of the biggest group's record count; the smallest g that fits
   const size_t nt = (n1 + 63) / 64, nt1 = nt + 1; // tiles of 64 pairs (flat_mate_count_kernel's waves)
   unsigned locus_bits = 1;
   while (((int64_t)1 << locus_bits) < n_loci) ++locus_bits;
   unsigned hash_bits = 32, group_shift = locus_bits, sort_bits = 32;
   for (unsigned g = 0; g <= locus_bits; ++g) {
      int64_t biggest = 1;
      for (int64_t l = 0; l < n_loci; l += (int64_t)1 << g) biggest = std::max(biggest, locus_read_off[std::min<int64_t>(n_loci, l + ((int64_t)1 << g))] - locus_read_off[l]);
      unsigned hb = 2;
      while (((int64_t)1 << (hb - 2)) < biggest) ++hb;
      if (locus_bits - g + hb <= 32 || g == locus_bits) {
         hash_bits = std::min(32u, hb), group_shift = g;
         sort_bits = std::min(32u, locus_bits - g + hash_bits);
         break;
      }
   }
   size_t tmp_bytes = 0, sort_tmp_bytes = 0;
   {
      size_t b = 0;
      (void)rocprim::radix_sort_pairs(nullptr, b, (const uint32_t *)nullptr, (uint32_t *)nullptr, rocprim::counting_iterator<int32_t>(0), (int32_t *)nullptr, n, 0, sort_bits, s);
      sort_tmp_bytes = b;
      (void)rocprim::exclusive_scan(nullptr, b, rocprim::make_transform_iterator((const int32_t *)nullptr, [] __device__(int32_t v) { return (int64_t)v; }), (int64_t *)nullptr, (int64_t)0, nt1, rocprim::plus<int64_t>(), s);
      tmp_bytes = std::max(tmp_bytes, b);
   }
   size_t off = 0;
   auto take = [&](size_t bytes) {
      const size_t o = off;
      off += up256(bytes ? bytes : 8);
      return o;
   };
   // what both forms use
   const size_t o_roff = take(nl1 * 8), o_rarr = take(n * sizeof(sb::FlatRec)), o_lefts = take(n * 4), o_claim = take(n * 4), o_slot = take(n * 8), o_runlen = take(n * 4);
   const size_t o_done = take(n * 8), o_td = take(nt1 * 4), o_tda = take(nt1 * 8), o_prec = take(n * 4), o_pval = take(n * 4);
   const size_t o_lf = take(n1 * 4), o_rf = take(n1 * 4), o_tl = take(nt1 * 4), o_tr = take(nt1 * 4), o_ls = take(nt1 * 8), o_rs = take(nt1 * 8), o_poff = take(nl1 * 8), o_counts = take(64 * 64 + 256), o_tmp = take(tmp_bytes);
   SB_TRY(sb::dev_take(off, &w, &w_cap));
   SB_TRY(hipMemsetAsync(w + o_counts, 0, 64 * 64 + 256, s));
   SB_TRY(hipMemsetAsync(w + o_td + ((n + 63) / 64) * 4, 0, 4, s)); // (the scans' entry beyond the last tile)
   SB_TRY(hipMemsetAsync(w + o_tl + nt * 4, 0, 4, s));
   SB_TRY(hipMemsetAsync(w + o_tr + nt * 4, 0, 4, s));
   SB_TRY(hipMemcpyAsync(w + o_roff, locus_read_off, nl1 * 8, hipMemcpyHostToDevice, s));
   sb::FlatMateArgs f = {};
   sb::MateArgs &a = f.a;
   a.n_loci = n_loci;
   a.locus_read_off = (const int64_t *)(w + o_roff);
   a.read_id = dr->read_id;
   a.block_off = dr->block_off;
   a.block_left = dr->block_left, a.block_right = dr->block_right;
   a.partner_pos = dr->partner_pos;
   a.flags = dr->flags;
   a.nh = dr->nh;
   f.n_reads = nr;
   f.hash_bits = (int)hash_bits, f.group_shift = (int)group_shift;
   f.rec_arr = (sb::FlatRec *)(w + o_rarr);
   f.lefts = (uint32_t *)(w + o_lefts), f.claim = (uint32_t *)(w + o_claim);
   f.slot = (uint32_t *)(w + o_slot), f.runlen = (uint32_t *)(w + o_runlen);
   f.done = (unsigned long long *)(w + o_done);
   f.tile_done = (int32_t *)(w + o_td), f.tile_done_at = (const int64_t *)(w + o_tda);
   f.pair_rec_w = (uint32_t *)(w + o_prec), f.pair_val_w = (uint32_t *)(w + o_pval);
   f.pair_rec = (const uint32_t *)(w + o_prec), f.pair_val = (const uint32_t *)(w + o_pval);
   f.lfeat = (int32_t *)(w + o_lf), f.rfeat = (int32_t *)(w + o_rf);
   f.tile_l = (int32_t *)(w + o_tl), f.tile_r = (int32_t *)(w + o_tr);
   f.lscan = (const int64_t *)(w + o_ls), f.rscan = (const int64_t *)(w + o_rs);
   f.locus_pair_off = (int64_t *)(w + o_poff);
   f.counts = (unsigned long long *)(w + o_counts);
   f.trouble = (uint32_t *)(w + o_counts + 64 * 64);
   const unsigned gr = (unsigned)((n + 255) / 256), gr1 = (unsigned)((std::max(n1, nl1) + 255) / 256);
   void *tmp = w + o_tmp;
   size_t tb = tmp_bytes;
   unsigned long long slots[64 * 8], counts[4] = {0, 0, 0, 0};
   int64_t totals[2] = {0, 0};
   uint32_t trouble = 0;
   // what follows the matching, whichever form made it: the pairs in the order their completing records arrive, the mates'
   // feature counts, where every pair's features go; the totals come back with one synchronisation
   auto rank_and_count = [&]() -> int {
      hipLaunchKernelGGL(sb::flat_mate_done_tiles_kernel, dim3(gr), dim3(256), 0, s, f);
      tb = tmp_bytes;
      SB_TRY(rocprim::exclusive_scan(tmp, tb, rocprim::make_transform_iterator((const int32_t *)f.tile_done, [] __device__(int32_t v) { return (int64_t)v; }), (int64_t *)(w + o_tda), (int64_t)0, (n + 63) / 64 + 1, rocprim::plus<int64_t>(), s));
      hipLaunchKernelGGL(sb::flat_mate_order_kernel, dim3(gr), dim3(256), 0, s, f);
      hipLaunchKernelGGL(sb::flat_mate_count_kernel, dim3(sb::xcd_grid(gr1)), dim3(256), 0, s, f);
      SB_TRY(hipGetLastError());
      tb = tmp_bytes;
      SB_TRY(rocprim::exclusive_scan(tmp, tb, rocprim::make_transform_iterator((const int32_t *)f.tile_l, [] __device__(int32_t v) { return (int64_t)v; }), (int64_t *)(w + o_ls), (int64_t)0, nt1, rocprim::plus<int64_t>(), s));
      tb = tmp_bytes;
      SB_TRY(rocprim::exclusive_scan(tmp, tb, rocprim::make_transform_iterator((const int32_t *)f.tile_r, [] __device__(int32_t v) { return (int64_t)v; }), (int64_t *)(w + o_rs), (int64_t)0, nt1, rocprim::plus<int64_t>(), s));
      SB_TRY(hipMemcpyAsync(slots, w + o_counts, sizeof(slots), hipMemcpyDeviceToHost, s));
      SB_TRY(hipMemcpyAsync(&trouble, w + o_counts + 64 * 64, 4, hipMemcpyDeviceToHost, s));
      SB_TRY(hipMemcpyAsync(&totals[0], w + o_ls + nt * 8, 8, hipMemcpyDeviceToHost, s));
      SB_TRY(hipMemcpyAsync(&totals[1], w + o_rs + nt * 8, 8, hipMemcpyDeviceToHost, s));
      SB_TRY(hipMemcpyAsync(M->locus_pair_off.data(), w + o_poff, nl1 * 8, hipMemcpyDeviceToHost, s));
      SB_TRY(hipStreamSynchronize(s));
      return SBGPU_OK;
   };
   // ---- the positional form (matepair_flat.h): no sort.  Its kernels say when they cannot serve the call (records that do
   // not ascend by position inside a cluster, a read id with several fitting mates); the ranking behind them has run by then -- it costs the call one more pass, in the rare case.
   // SBGPU_PAIR_FORCE_SORT=1 (tests): every call takes the sorted form.
   const char *force_env = std::getenv("SBGPU_PAIR_FORCE_SORT");
   bool sorted_form = force_env && std::atoi(force_env) != 0;
   if (!sorted_form) {
      SB_TRY(hipMemsetAsync(w + o_slot, 0, n * 8, s));
      hipLaunchKernelGGL(sb::flat_mate_rec_kernel, dim3(sb::xcd_grid(gr)), dim3(256), 0, s, f);
      hipLaunchKernelGGL(sb::flat_mate_match_kernel, dim3(sb::xcd_grid(gr)), dim3(256), 0, s, f);
      SB_TRY(hipGetLastError());
      if (const int rc = rank_and_count(); rc != SBGPU_OK) return rc;
      sorted_form = trouble != 0;
   }
   M->positional = !sorted_form;
   M->why_sorted = trouble;
   if (sorted_form) {
      // ---- the sorted form: ONE stable radix sort brings the records of a read id together in arrival order, a walk pairs them
      // with the reference's chain rules (any number of waiting mates per read id)
      char *w2 = nullptr;
      size_t w2_cap = 0, off2 = 0;
      auto take2 = [&](size_t bytes) {
         const size_t o = off2;
         off2 += up256(bytes ? bytes : 8);
         return o;
      };
      const size_t o_key = take2(n * 4), o_skey = take2(n * 4), o_order = take2(n * 4), o_rec = take2(n * sizeof(sb::FlatRec)), o_state = take2(n), o_stmp = take2(sort_tmp_bytes);
      SB_TRY(sb::dev_take(off2, &w2, &w2_cap));
      struct Give {
         char *&p;
         size_t &cap;
         hipStream_t s;
         ~Give()
         {
            (void)hipStreamSynchronize(s);
            sb::dev_give(p, cap);
         }
      } give2 = {w2, w2_cap, s};
      f.key = (uint32_t *)(w2 + o_key);
      f.skey = (const uint32_t *)(w2 + o_skey);
      f.order = (const int32_t *)(w2 + o_order);
      f.rec = (sb::FlatRec *)(w2 + o_rec);
      f.state = (uint8_t *)(w2 + o_state);
      f.claim = nullptr;
      SB_TRY(hipMemsetAsync(w + o_counts, 0, 64 * 64 + 256, s));
      hipLaunchKernelGGL(sb::flat_mate_keys_kernel, dim3(gr), dim3(256), 0, s, f); // (also: the records as the rules see them, `done` zeroed)
      SB_TRY(hipGetLastError());
      size_t stb = sort_tmp_bytes;
      SB_TRY(rocprim::radix_sort_pairs(w2 + o_stmp, stb, (const uint32_t *)f.key, (uint32_t *)(w2 + o_skey), rocprim::counting_iterator<int32_t>(0), (int32_t *)(w2 + o_order), n, 0, sort_bits, s));
      hipLaunchKernelGGL(sb::flat_mate_pack_kernel, dim3(gr), dim3(256), 0, s, f);
      hipLaunchKernelGGL(sb::flat_mate_walk_kernel, dim3(gr), dim3(256), 0, s, f);
      SB_TRY(hipGetLastError());
      if (const int rc = rank_and_count(); rc != SBGPU_OK) return rc;
   }
   for (int k = 0; k < 64; ++k)
      for (int i = 0; i < 4; ++i) counts[i] += slots[k * 8 + i]; // (wrapping sums: a slot's orphan count may be "negative")
   M->n_refused = (int64_t)counts[0], M->n_orphan = (int64_t)counts[1], M->n_single = (int64_t)counts[2], M->n_complete = (int64_t)counts[3];
   M->n_pairs = M->locus_pair_off[(size_t)n_loci];
   M->n_lfeat = totals[0], M->n_rfeat = totals[1];
   if (M->n_pairs != M->n_single + M->n_complete) return bail(SBGPU_EHIP, "sbgpu_pair_mates_device: the pairs' count and the walk's counters disagree");
   // ---- the pairs' own arena
   const size_t np1 = (size_t)M->n_pairs + 1, nlf = (size_t)M->n_lfeat + 1, nrf = (size_t)M->n_rfeat + 1;
   size_t t = 0;
   const size_t u_mass = t; t += up256(np1 * 8);
   const size_t u_loff = t; t += up256(np1 * 8);
   const size_t u_roff = t; t += up256(np1 * 8);
   const size_t u_ll = t; t += up256(nlf * 4);
   const size_t u_lr = t; t += up256(nlf * 4);
   const size_t u_rl = t; t += up256(nrf * 4);
   const size_t u_rr = t; t += up256(nrf * 4);
   const size_t u_lc = t; t += up256(nlf);
   const size_t u_rc = t; t += up256(nrf);
   SB_TRY(sb::dev_take(t, &M->arena, &M->arena_cap));
   M->d_mass = (double *)(M->arena + u_mass);
   M->d_left_off = (int64_t *)(M->arena + u_loff);
   M->d_right_off = (int64_t *)(M->arena + u_roff);
   M->d_left_left = (uint32_t *)(M->arena + u_ll), M->d_left_right = (uint32_t *)(M->arena + u_lr);
   M->d_right_left = (uint32_t *)(M->arena + u_rl), M->d_right_right = (uint32_t *)(M->arena + u_rr);
   M->d_left_code = (uint8_t *)(M->arena + u_lc), M->d_right_code = (uint8_t *)(M->arena + u_rc);
   a.pair_mass = M->d_mass;
   a.left_off = M->d_left_off, a.right_off = M->d_right_off;
   a.left_code = M->d_left_code, a.right_code = M->d_right_code;
   a.left_left = M->d_left_left, a.left_right = M->d_left_right;
   a.right_left = M->d_right_left, a.right_right = M->d_right_right;
   hipLaunchKernelGGL(sb::flat_mate_fill_kernel, dim3(sb::xcd_grid((int64_t)((np1 + 255) / 256))), dim3(256), 0, s, f, M->n_pairs);
   SB_TRY(hipGetLastError());
   SB_TRY(hipStreamSynchronize(s)); // the scratch goes back to the pool
#undef SB_TRY
   sb::dev_give(w, w_cap);
   return SBGPU_OK;
}

extern "C" {

void sbgpu_matepairs_destroy(sbgpu_matepairs_t *m)
{
   if (!m) return;
   sb::dev_give(m->arena, m->arena_cap); // (waits for the device the arena lives on, as hipFree did)
   delete m;
}

int sbgpu_pair_mates_host(int64_t n_loci, const sbgpu_reads_t *rd, const int64_t *locus_read_off, sbgpu_matepairs_t **out)
{
   if (!rd || !out || !locus_read_off || n_loci < 0) return api_fail(SBGPU_EINVAL, "sbgpu_pair_mates_host: bad argument");
   *out = nullptr;
   const int64_t nr = rd->n_reads;
   if (nr < 0 || locus_read_off[0] != 0 || locus_read_off[n_loci] != nr) return api_fail(SBGPU_EINVAL, "sbgpu_pair_mates_host: locus_read_off does not cover the reads");
   if (nr && (!rd->read_id || !rd->block_off || !rd->partner_pos || !rd->flags || !rd->nh)) return api_fail(SBGPU_EINVAL, "sbgpu_pair_mates_host: null array");
   sbgpu_matepairs *M = new (std::nothrow) sbgpu_matepairs();
   if (!M) return api_fail(SBGPU_ENOMEM, "sbgpu_pair_mates_host: out of memory");
   try {
      M->n_loci = n_loci;
      M->locus_pair_off.assign((size_t)n_loci + 1, 0);
      M->left_off.push_back(0);
      M->right_off.push_back(0);
      auto left_of = [&](int64_t r) { return rd->block_left[rd->block_off[r]]; };
      auto push_mate = [&](int64_t r, std::vector<uint8_t> &c, std::vector<uint32_t> &l, std::vector<uint32_t> &rr) {
         // readhit_2_genomicFeats, src/contig.cpp:12-53: the blocks with the introns between them (none between blocks
         // that touch: an insertion in the read)
         for (int64_t b = rd->block_off[r]; b < rd->block_off[r + 1]; ++b) {
            if (b > rd->block_off[r] && rd->block_left[b] != rd->block_right[b - 1] + 1u) {
               c.push_back(1);
               l.push_back(rd->block_right[b - 1] + 1);
               rr.push_back(rd->block_left[b] - 1);
            }
            c.push_back(0);
            l.push_back(rd->block_left[b]);
            rr.push_back(rd->block_right[b]);
         }
      };
      auto add_hit = [&](int64_t l, int64_t left_rec, int64_t right_rec) { // HitCluster::addHit, in completion order
         if (left_rec >= 0) push_mate(left_rec, M->left_code, M->left_left, M->left_right);
         if (right_rec >= 0) push_mate(right_rec, M->right_code, M->right_left, M->right_right);
         M->left_off.push_back((int64_t)M->left_code.size());
         M->right_off.push_back((int64_t)M->right_code.size());
         double m = 0.0; // PairedHit::init_raw_mass, src/read.cpp:734-741 over ReadHit masses (:49-53)
         if (left_rec >= 0 && right_rec >= 0) m = 0.5 / rd->nh[left_rec] + 0.5 / rd->nh[right_rec];
         else m = 1.0 / rd->nh[left_rec >= 0 ? left_rec : right_rec];
         M->pair_mass.push_back(m);
         M->pair_locus.push_back((int32_t)l);
      };
      for (int64_t l = 0; l < n_loci; ++l) {
         if (locus_read_off[l + 1] < locus_read_off[l]) {
            delete M;
            return api_fail(SBGPU_EINVAL, "sbgpu_pair_mates_host: locus_read_off must ascend");
         }
         std::unordered_map<uint64_t, std::vector<int64_t>> open; // HitCluster::_open_mates: read id -> waiting records, oldest first
         for (int64_t r = locus_read_off[l]; r < locus_read_off[l + 1]; ++r) {
            const int64_t b0 = rd->block_off[r], b1 = rd->block_off[r + 1];
            if (rd->flags[r] & SBGPU_READ_SKIP) continue; // not this cluster's record (sbgpu_assign_reads_*)
            if (b1 <= b0) {
               ++M->n_refused;
               continue;
            }
            const uint32_t left = rd->block_left[b0], right = rd->block_right[b1 - 1];
            if ((int64_t)right - (int64_t)left > kMaxFragSpanHost) { // alignments.cpp:512-518
               ++M->n_refused;
               continue;
            }
            const uint32_t ppos = rd->partner_pos[r];
            const uint8_t fl = rd->flags[r];
            if (ppos == 0 || (fl & SBGPU_READ_PARTNER_ELSEWHERE)) { // :535-545
               if (fl & SBGPU_READ_REVERSE) add_hit(l, -1, r);
               else add_hit(l, r, -1);
               ++M->n_single;
               continue;
            }
            const int strand = (fl >> 2) & 3;
            std::vector<int64_t> &chain = open[rd->read_id[r]];
            bool done = false;
            for (size_t o = 0; o < chain.size(); ++o) { // :590-623
               const int64_t w = chain[o];
               const int wstrand = (rd->flags[w] >> 2) & 3;
               const bool strand_agree = wstrand == strand || strand == 0 || wstrand == 0;
               if (left_of(w) == ppos && strand_agree && rd->partner_pos[w] == left) {
                  const bool waiting_is_left = rd->partner_pos[w] > left_of(w);
                  if (waiting_is_left) add_hit(l, w, r);
                  else add_hit(l, r, w);
                  chain.erase(chain.begin() + (std::ptrdiff_t)o);
                  ++M->n_complete;
                  done = true;
                  break;
               }
            }
            if (done) continue;
            if (ppos == left) { // :585, :640
               ++M->n_refused;
               continue;
            }
            chain.push_back(r);
         }
         for (const auto &kv : open) M->n_orphan += (int64_t)kv.second.size(); // clearOpenMates (:653)
         M->locus_pair_off[(size_t)l + 1] = (int64_t)M->pair_mass.size();
      }
      M->n_pairs = (int64_t)M->pair_mass.size();
      M->n_lfeat = (int64_t)M->left_code.size();
      M->n_rfeat = (int64_t)M->right_code.size();
   } catch (const std::bad_alloc &) {
      delete M;
      return api_fail(SBGPU_ENOMEM, "sbgpu_pair_mates_host: out of memory");
   }
   *out = M;
   return SBGPU_OK;
}

int sbgpu_pair_mates_device(sbgpu_ctx_t *c, int64_t n_loci, const sbgpu_reads_t *dr, const int64_t *locus_read_off, void *stream,
                            sbgpu_matepairs_t **out)
{
   if (!c || !dr || !out || !locus_read_off || n_loci < 0) return api_fail(SBGPU_EINVAL, "sbgpu_pair_mates_device: bad argument");
   *out = nullptr;
   const int64_t nr = dr->n_reads;
   if (nr < 0 || locus_read_off[0] != 0 || locus_read_off[n_loci] != nr) return api_fail(SBGPU_EINVAL, "sbgpu_pair_mates_device: locus_read_off does not cover the reads");
   if (nr && (!dr->read_id || !dr->block_off || !dr->block_left || !dr->block_right || !dr->partner_pos || !dr->flags || !dr->nh))
      return api_fail(SBGPU_EINVAL, "sbgpu_pair_mates_device: null device pointer");
   for (int64_t l = 0; l < n_loci; ++l)
      if (locus_read_off[l + 1] < locus_read_off[l]) return api_fail(SBGPU_EINVAL, "sbgpu_pair_mates_device: locus_read_off must ascend");
   hipStream_t s = (hipStream_t)stream;
   sbgpu_matepairs *M = new (std::nothrow) sbgpu_matepairs();
   if (!M) return api_fail(SBGPU_ENOMEM, "sbgpu_pair_mates_device: out of host memory");
   M->on_device = true;
   M->device = sb::ctx_device(c);
   M->n_loci = n_loci;
   M->locus_pair_off.assign((size_t)n_loci + 1, 0);
   if (nr == 0 || n_loci == 0) {
      *out = M;
      return SBGPU_OK;
   }
   // every cluster of the call at once (matepair_flat.h)
   const int rc = pair_mates_flat(c, n_loci, dr, locus_read_off, s, M);
   if (rc != SBGPU_OK) {
      sbgpu_matepairs_destroy(M);
      return rc;
   }
   *out = M;
   return SBGPU_OK;
}

// ---- cluster streaming
static int check_clusters(const sbgpu_clusters_t *cl, const char *who)
{
   if (!cl || cl->n_clusters < 0 || (cl->n_clusters && (!cl->ref || !cl->left || !cl->right || !cl->strand)))
      return api_fail(SBGPU_EINVAL, std::string(who) + ": bad clusters");
   for (int64_t k = 0; k + 1 < cl->n_clusters; ++k)
      if (cl->ref[k] > cl->ref[k + 1] || (cl->ref[k] == cl->ref[k + 1] && cl->left[k] > cl->left[k + 1]))
         return api_fail(SBGPU_EINVAL, std::string(who) + ": clusters must come sorted by (reference, left)");
   return SBGPU_OK;
}

int sbgpu_assign_reads_host(const sbgpu_clusters_t *cl, int64_t n_reads, const int32_t *read_ref, const uint32_t *read_left,
                            const uint32_t *read_right, uint8_t *flags, int32_t *read_cluster, int64_t *off)
{
   const int rc = check_clusters(cl, "sbgpu_assign_reads_host");
   if (rc != SBGPU_OK) return rc;
   if (n_reads < 0 || !off || (n_reads && (!read_ref || !read_left || !read_right || !read_cluster)))
      return api_fail(SBGPU_EINVAL, "sbgpu_assign_reads_host: null argument");
   auto key = [](int32_t ref, uint32_t pos) { return ((uint64_t)(uint32_t)ref << 32) | pos; };
   // where every cluster's pass begins: the prefix maximum of "first record behind the cluster's end"
   off[0] = 0;
   for (int64_t k = 0; k < cl->n_clusters; ++k) {
      const uint64_t end = key(cl->ref[k], cl->right[k]);
      int64_t lo = 0, hi = n_reads;
      while (lo < hi) {
         const int64_t mid = (lo + hi) >> 1;
         if (key(read_ref[mid], read_left[mid]) <= end) lo = mid + 1;
         else hi = mid;
      }
      off[k + 1] = std::max(off[k], lo);
   }
   for (int64_t k = 0; k < cl->n_clusters; ++k)
      for (int64_t i = off[k]; i < off[k + 1]; ++i) {
         const bool lt = read_ref[i] < cl->ref[k] || (read_ref[i] == cl->ref[k] && read_right[i] < cl->left[k]);
         const int xs = flags ? (flags[i] >> 2) & 3 : 0;
         const bool strand_off = xs != 0 && xs != (int)cl->strand[k];
         read_cluster[i] = (lt || strand_off) ? -1 : (int32_t)k;
      }
   for (int64_t i = cl->n_clusters ? off[cl->n_clusters] : 0; i < n_reads; ++i) read_cluster[i] = -1;
   if (flags)
      for (int64_t i = 0; i < n_reads; ++i)
         if (read_cluster[i] < 0) flags[i] |= SBGPU_READ_SKIP;
   return SBGPU_OK;
}

int sbgpu_assign_reads_device(sbgpu_ctx_t *c, const sbgpu_clusters_t *cl, int64_t n_reads, const int32_t *d_ref, const uint32_t *d_left,
                              const uint32_t *d_right, uint8_t *d_flags, int32_t *d_read_cluster, int64_t *off, void *stream)
{
   const int rc = check_clusters(cl, "sbgpu_assign_reads_device");
   if (rc != SBGPU_OK) return rc;
   if (!c || n_reads < 0 || !off || (n_reads && (!d_ref || !d_left || !d_right || !d_read_cluster)))
      return api_fail(SBGPU_EINVAL, "sbgpu_assign_reads_device: null argument");
   const int64_t nc = cl->n_clusters;
   off[0] = 0;
   if (nc == 0 || n_reads == 0) {
      for (int64_t k = 0; k < nc; ++k) off[k + 1] = 0;
      if (n_reads) {
         hipError_t e = hipMemsetAsync(d_read_cluster, 0xFF, (size_t)n_reads * 4, (hipStream_t)stream);
         if (e != hipSuccess) return api_fail(SBGPU_EHIP, std::string("hipMemsetAsync: ") + hipGetErrorString(e));
      }
      return SBGPU_OK;
   }
   hipStream_t s = (hipStream_t)stream;
   char *w = nullptr;
   const size_t nc1 = (size_t)nc + 1;
   size_t t = 0;
   const size_t o_ref = t; t += up256(nc1 * 4);
   const size_t o_left = t; t += up256(nc1 * 4);
   const size_t o_right = t; t += up256(nc1 * 4);
   const size_t o_strand = t; t += up256(nc1);
   const size_t o_ub = t; t += up256(nc1 * 8);
   const size_t o_pos = t; t += up256(nc1 * 8);
   hipError_t e = sb::ctx_scratch(c, 5, t, &w);
   if (e != hipSuccess) return api_fail(e == hipErrorOutOfMemory ? SBGPU_ENOMEM : SBGPU_EHIP, std::string("hipMalloc: ") + hipGetErrorString(e));
#define SB_TRY(expr)                                                                                        \
   do {                                                                                                     \
      hipError_t e_ = (expr);                                                                               \
      if (e_ != hipSuccess) return api_fail(SBGPU_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
   } while (0)
   SB_TRY(hipMemcpyAsync(w + o_ref, cl->ref, (size_t)nc * 4, hipMemcpyHostToDevice, s));
   SB_TRY(hipMemcpyAsync(w + o_left, cl->left, (size_t)nc * 4, hipMemcpyHostToDevice, s));
   SB_TRY(hipMemcpyAsync(w + o_right, cl->right, (size_t)nc * 4, hipMemcpyHostToDevice, s));
   SB_TRY(hipMemcpyAsync(w + o_strand, cl->strand, (size_t)nc, hipMemcpyHostToDevice, s));
   sb::AssignArgs a = {};
   a.n_clusters = nc, a.n_reads = n_reads;
   a.c_ref = (const int32_t *)(w + o_ref);
   a.c_left = (const uint32_t *)(w + o_left), a.c_right = (const uint32_t *)(w + o_right);
   a.c_strand = (const uint8_t *)(w + o_strand);
   a.r_ref = d_ref, a.r_left = d_left, a.r_right = d_right, a.r_flags = d_flags;
   a.ub = (int64_t *)(w + o_ub);
   a.pos = (const int64_t *)(w + o_pos);
   a.read_cluster = d_read_cluster;
   hipLaunchKernelGGL(sb::cluster_bounds_kernel, dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, s, a);
   SB_TRY(hipGetLastError());
   std::vector<int64_t> ub((size_t)nc);
   SB_TRY(hipMemcpyAsync(ub.data(), w + o_ub, (size_t)nc * 8, hipMemcpyDeviceToHost, s));
   SB_TRY(hipStreamSynchronize(s));
   for (int64_t k = 0; k < nc; ++k) off[k + 1] = std::max(off[k], ub[(size_t)k]); // the pass never goes back
   SB_TRY(hipMemcpyAsync(w + o_pos, off, nc1 * 8, hipMemcpyHostToDevice, s));
   const int64_t blocks = std::min<int64_t>((n_reads + 255) / 256, (int64_t)sb::ctx_cu_count(c) * 32);
   hipLaunchKernelGGL(sb::assign_reads_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a);
   SB_TRY(hipGetLastError());
   SB_TRY(hipStreamSynchronize(s)); // (`off` was read from host memory)
#undef SB_TRY
   return SBGPU_OK;
}

int sbgpu_matepairs_info(const sbgpu_matepairs_t *m, int64_t info[8])
{
   if (!m || !info) return api_fail(SBGPU_EINVAL, "sbgpu_matepairs_info: null argument");
   info[0] = m->n_pairs;
   info[1] = m->n_complete;
   info[2] = m->n_single;
   info[3] = m->n_refused;
   info[4] = m->n_orphan;
   info[5] = m->n_lfeat;
   info[6] = m->n_rfeat;
   info[7] = m->on_device ? (1 | (m->positional ? 2 : 0) | ((int64_t)m->why_sorted << 2)) : 0;
   return SBGPU_OK;
}

int sbgpu_matepairs_pairs(const sbgpu_matepairs_t *m, sbgpu_pairs_t *p, const int64_t **locus_pair_off)
{
   if (!m || !p) return api_fail(SBGPU_EINVAL, "sbgpu_matepairs_pairs: null argument");
   p->n_pairs = m->n_pairs;
   if (m->on_device) {
      p->pair_locus = nullptr;
      p->pair_mass = m->d_mass;
      p->left_off = m->d_left_off, p->left_code = m->d_left_code, p->left_left = m->d_left_left, p->left_right = m->d_left_right;
      p->right_off = m->d_right_off, p->right_code = m->d_right_code, p->right_left = m->d_right_left, p->right_right = m->d_right_right;
   } else {
      p->pair_locus = m->pair_locus.data();
      p->pair_mass = m->pair_mass.data();
      p->left_off = m->left_off.data(), p->left_code = m->left_code.data(), p->left_left = m->left_left.data(), p->left_right = m->left_right.data();
      p->right_off = m->right_off.data(), p->right_code = m->right_code.data(), p->right_left = m->right_left.data(), p->right_right = m->right_right.data();
   }
   if (locus_pair_off) *locus_pair_off = m->locus_pair_off.data();
   return SBGPU_OK;
}

int sbgpu_matepairs_export(const sbgpu_matepairs_t *m, double *pair_mass, int64_t *left_off, uint8_t *left_code, uint32_t *left_left,
                           uint32_t *left_right, int64_t *right_off, uint8_t *right_code, uint32_t *right_left, uint32_t *right_right)
{
   if (!m) return api_fail(SBGPU_EINVAL, "sbgpu_matepairs_export: null argument");
   const size_t np = (size_t)m->n_pairs, nl = (size_t)m->n_lfeat, nr = (size_t)m->n_rfeat;
   if (m->on_device) {
      hipError_t e = hipSetDevice(m->device);
      auto get = [&](void *dst, const void *src, size_t bytes) {
         if (e == hipSuccess && dst && bytes && src) e = hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost);
      };
      if (m->arena) {
         get(pair_mass, m->d_mass, np * 8);
         get(left_off, m->d_left_off, (np + 1) * 8);
         get(right_off, m->d_right_off, (np + 1) * 8);
         get(left_code, m->d_left_code, nl), get(left_left, m->d_left_left, nl * 4), get(left_right, m->d_left_right, nl * 4);
         get(right_code, m->d_right_code, nr), get(right_left, m->d_right_left, nr * 4), get(right_right, m->d_right_right, nr * 4);
      } else {
         if (left_off) left_off[0] = 0;
         if (right_off) right_off[0] = 0;
      }
      if (e != hipSuccess) return api_fail(SBGPU_EHIP, std::string("sbgpu_matepairs_export: ") + hipGetErrorString(e));
      return SBGPU_OK;
   }
   auto cp = [](void *dst, const void *src, size_t bytes) {
      if (dst && bytes) std::memcpy(dst, src, bytes);
   };
   cp(pair_mass, m->pair_mass.data(), np * 8);
   cp(left_off, m->left_off.data(), (np + 1) * 8);
   cp(right_off, m->right_off.data(), (np + 1) * 8);
   cp(left_code, m->left_code.data(), nl), cp(left_left, m->left_left.data(), nl * 4), cp(left_right, m->left_right.data(), nl * 4);
   cp(right_code, m->right_code.data(), nr), cp(right_left, m->right_left.data(), nr * 4), cp(right_right, m->right_right.data(), nr * 4);
   return SBGPU_OK;
}

} // extern "C"
