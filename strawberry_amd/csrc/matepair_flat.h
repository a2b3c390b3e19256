// strawberry_amd/csrc/matepair_flat.h -- HitCluster::addOpenHit / addHit for ALL clusters of a call at once
// (/root/reference/src/alignments.cpp:423-655; round 4: replaces one-workgroup-per-cluster, matepair_device.h).
//
// Only records of ONE read id of ONE cluster ever interact.  The per-cluster form brought them together with a bitonic sort
// per cluster (LDS, or global memory for a highly expressed gene) and a sample's time was its biggest clusters'.  Here:
//
//   keys    a thread per record: a 32-bit key -- the number of a group of neighbouring clusters, then a hash of (cluster,
//           read id) -- and the 24 bytes of the record the rules look at, both in arrival order (coalesced);
//   sort    ONE stable device-wide radix sort (rocPRIM onesweep) on that key, the arrival index as value: the records of a
//           read id become neighbours in ARRIVAL order.  Records of other read ids or clusters that share the key share
//           the run; the walk below takes the records of ITS read id and cluster out of it (ids and clusters themselves
//           are compared), so a collision costs a few extra comparisons and nothing else;
//   pack    the records' 24 bytes moved into sorted order once (read in sequence afterwards);
//   walk    the first record of every read id walks its group with the reference's open-mate rules (:535-641): a waiting
//           mate is a state byte per record ("open" until a partner takes it, oldest first), so any number of mates of one
//           read id may wait -- the per-cluster form's list of 8 is gone.  What a record completes is left at the
//           record's ARRIVAL index;
//   rank    addHit is called when the SECOND mate arrives: the pairs are ordered by their completing record's arrival index
//           -- a compaction of the completing records in arrival order (counts per tile of 64 positions, a scan over the
//           tiles, a pass that writes every completing record to its rank; round 4 sorted a second time), for all
//           clusters at once, because the records come cluster by cluster; where a cluster's pairs begin is a binary
//           search in the list;
//   count   the mates' feature counts per pair and their sums per tile of 64 pairs; two scans over the tiles;
//   fill    a thread per pair writes its mates as MATCH / INTRON features (contig.cpp:12-53) and its mass.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "matepair_device.h"
#include "device_common.h"

namespace sb {

struct FlatRec { // a record as the rules see it
   uint32_t rid_lo, rid_hi;
   uint32_t left, right; // first block's left end, last block's right end
   uint32_t ppos;        // the mate's position
   uint32_t misc;        // bits 0-7: the record's flags; bit 8: it has blocks; bits 9-31: its cluster
};

struct FlatMateArgs {
   MateArgs a;       // the records (inputs) and the pairs' arrays (outputs of the fill)
   int64_t n_reads;
   uint32_t *key;                   // per record, arrival order: (cluster >> group_shift) << hash_bits | the top hash_bits of a hash of (cluster, read id)
   const uint32_t *skey;            // sorted
   int hash_bits, group_shift;
   const int32_t *order;            // record (arrival index) at sorted position s
   FlatRec *rec_arr;                // the records as the rules see them, ARRIVAL order (made beside the keys, with coalesced loads)
   FlatRec *rec;                    // sorted order
   uint8_t *state;                  // sorted order: 0 nothing, 1 waits for its mate, 2 taken
   unsigned long long *done;        // ARRIVAL order, zeroed by the keys kernel: what record r completes -- 0 nothing, else bit 32 set and
                                    // in the low word: bit 31 the completing record is the RIGHT mate, bits 0-30 the waiting mate's
                                    // arrival index (0x7FFFFFFF: a single read)
   int32_t *tile_done;              // completing records per tile of 64 arrival positions (one entry beyond the end: 0)
   const int64_t *tile_done_at;     // ... their exclusive scan
   uint32_t *pair_rec_w, *pair_val_w;   // (pair_rec / pair_val, writable)
   const uint32_t *pair_rec, *pair_val; // pair k of the call: the completing records in arrival order (their arrival index; 0xFFFFFFFF
                                        // from the last pair on) and their low words of `done`
   int32_t *lfeat, *rfeat;          // per pair (one entry beyond the end: 0)
   int32_t *tile_l, *tile_r;        // their sums over tiles of 64 pairs (a wave of the count kernel; one entry beyond the end: 0)
   const int64_t *lscan, *rscan;    // exclusive scans of the TILES' sums: the device-wide scans run over 1/64 of the pairs, the
                                    // fill adds the part inside its wave
   int64_t *locus_pair_off;         // [n_loci + 1]
   unsigned long long *counts;      // 64 slots of 8 words (a cache line each): [0] refused [1] orphan [2] single [3] complete
};

__device__ __forceinline__ uint32_t flat_hash32(uint64_t x)
{
   x ^= x >> 33;
   x *= 0xff51afd7ed558ccdull;
   x ^= x >> 33;
   x *= 0xc4ceb9fe1a85ec53ull;
   x ^= x >> 33;
   return (uint32_t)x;
}

__global__ __launch_bounds__(256) void flat_mate_keys_kernel(FlatMateArgs f)
{
   const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
   const int64_t r0 = r - (int64_t)(threadIdx.x & 63u);
   if (r0 >= f.n_reads) return; // (the whole wave)
   const int64_t lo = wave_range_of(f.a.locus_read_off, f.a.n_loci, r0, f.n_reads); // last cluster whose records begin at or before r
   if (r >= f.n_reads) return;
   const MateArgs &a = f.a;
   const uint64_t rid = a.read_id[r];
   // The sort only has to bring the records of a (cluster, read id) together, in arrival order (it is stable); the pairs
   // are put in order by the second sort.  So the key is a hash of the two -- as many bits as the records of a GROUP of
   // neighbouring clusters ask for (two more than their count's logarithm: a run holds half a foreign record on average;
   // the walk tells them apart by read id and cluster) -- behind the group's number: 32 bits in all, four passes over
   // 8-byte elements where (cluster << 32 | hash) took six over 12-byte ones.  The group in front keeps the sorted order
   // local: the pack kernel's gather of a sorted position's record then stays inside the group's records (a few MB, in
   // cache), where a key that is all hash sends every lane to a line of its own anywhere in the call's 9 GB.
   const uint32_t h = f.hash_bits ? flat_hash32(rid ^ ((uint64_t)lo * 0x9E3779B97F4A7C15ull)) >> (32 - f.hash_bits) : 0u;
   f.key[r] = (f.hash_bits < 32 ? (uint32_t)(lo >> f.group_shift) << f.hash_bits : 0u) | h;
   // the record as the rules see it, here where the lanes' records are neighbours in every array (round 5: the pack kernel
   // used to collect these fields in SORTED order -- seven scattered reads per record, 94 GB fetched for 15 GB wanted at
   // 3.9e8 records; it now moves one 24-byte struct per record)
   const int64_t b0 = a.block_off[r], b1 = a.block_off[r + 1];
   FlatRec q;
   q.rid_lo = (uint32_t)rid, q.rid_hi = (uint32_t)(rid >> 32);
   q.left = b1 > b0 ? a.block_left[b0] : 0u;
   q.right = b1 > b0 ? a.block_right[b1 - 1] : 0u;
   q.ppos = a.partner_pos[r];
   q.misc = (uint32_t)a.flags[r] | (b1 > b0 ? 256u : 0u) | ((uint32_t)lo << 9);
   f.rec_arr[r] = q;
   f.done[r] = 0ull;
}

__global__ __launch_bounds__(256) void flat_mate_pack_kernel(FlatMateArgs f)
{
   const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
   if (s >= f.n_reads) return;
   f.rec[s] = f.rec_arr[f.order[s]];
   f.state[s] = 0;
}

// the first record of every read id walks its group (arrival order) with the reference's open-mate rules
__global__ __launch_bounds__(256) void flat_mate_walk_kernel(FlatMateArgs f)
{
   const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
   int refused = 0, orphan = 0, single = 0, complete = 0;
   if (s < f.n_reads) {
      const uint32_t k = f.skey[s];
      const FlatRec me = f.rec[s];
      auto mine = [&](const FlatRec &q) { return q.rid_lo == me.rid_lo && q.rid_hi == me.rid_hi && (q.misc >> 9) == (me.misc >> 9); };
      bool first = true; // no earlier record of this read id (and cluster) in the key's run
      for (int64_t t = s - 1; t >= 0 && f.skey[t] == k; --t)
         if (mine(f.rec[t])) {
            first = false;
            break;
         }
      if (first) {
         // which records of the run wait for their mate: the first 64 positions behind s as a bit mask in registers (every
         // group but a read with dozens of alignments in one cluster ends there), the rest as state bytes in memory
         unsigned long long open_mask = 0;
         bool far_open = false;
         for (int64_t t = s; t < f.n_reads && f.skey[t] == k; ++t) {
            const FlatRec q = t == s ? me : f.rec[t];
            if (!mine(q)) continue; // another read id (or cluster) with the same hash
            const uint32_t fl = q.misc & 255u;
            if (fl & 16u) continue; // not this cluster's record (sbgpu_assign_reads_*): never offered to addOpenHit
            if (!(q.misc & 256u) || (int64_t)q.right - (int64_t)q.left > kMaxFragSpanDev) { // :512-518
               ++refused;
               continue;
            }
            const uint32_t r_t = (uint32_t)f.order[t];
            if (q.ppos == 0 || (fl & 2u)) { // a single read (:535-545)
               f.done[r_t] = (1ull << 32) | ((fl & 1u) ? 0x80000000u : 0u) | 0x7FFFFFFFu;
               ++single;
               continue;
            }
            const int strand = (fl >> 2) & 3;
            auto fits = [&](const FlatRec &w) { // :590-623
               const int wstrand = (w.misc >> 2) & 3;
               const bool strand_agree = wstrand == strand || strand == 0 || wstrand == 0;
               return w.left == q.ppos && strand_agree && w.ppos == q.left;
            };
            int64_t hit = -1;
            for (unsigned long long m = open_mask; m && hit < 0; m &= m - 1) { // oldest first
               const int64_t o = s + (__ffsll((long long)m) - 1);
               if (fits(o == s ? me : f.rec[o])) hit = o;
            }
            if (hit < 0 && far_open)
               for (int64_t o = s + 64; o < t && hit < 0; ++o) {
                  if (f.state[o] != 1) continue;
                  const FlatRec w = f.rec[o];
                  if (!mine(w)) continue; // (another read id of the run, waiting for ITS mate)
                  if (fits(w)) hit = o;
               }
            if (hit >= 0) {
               const FlatRec w = hit == s ? me : f.rec[hit];
               const bool waiting_is_left = w.ppos > w.left; // the waiting mate is the left one when its partner lies behind it (:559-585)
               f.done[r_t] = (1ull << 32) | (waiting_is_left ? 0x80000000u : 0u) | (uint32_t)f.order[hit];
               if (hit - s < 64) open_mask &= ~(1ull << (hit - s));
               else f.state[hit] = 2;
               ++complete;
               --orphan;
            } else if (q.ppos == q.left) { // :585, :640: partner and read start at the same position
               ++refused;
            } else { // waits for its partner
               if (t - s < 64) open_mask |= 1ull << (t - s);
               else f.state[t] = 1, far_open = true;
               ++orphan;
            }
         }
      }
   }
   // the call's totals: a wave adds up in registers, the workgroup in LDS, and one lane adds the workgroup's four numbers to
   // one of 64 slots (a slot per cache line): 3e5 waves adding to ONE word take their turns at the L2 -- 3.7 of this
   // kernel's first 3.8 ms were that queue
   for (int o = 32; o > 0; o >>= 1) {
      refused += __shfl_xor(refused, o);
      orphan += __shfl_xor(orphan, o);
      single += __shfl_xor(single, o);
      complete += __shfl_xor(complete, o);
   }
   __shared__ int part[4][4];
   const int wave = threadIdx.x >> 6;
   if ((threadIdx.x & 63) == 0) part[wave][0] = refused, part[wave][1] = orphan, part[wave][2] = single, part[wave][3] = complete;
   __syncthreads();
   if (threadIdx.x < 4) {
      const long long v = (long long)part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
      if (v) atomicAdd(&f.counts[(size_t)(blockIdx.x & 63) * 8 + threadIdx.x], (unsigned long long)v);
   }
}

// ---- the pairs' order: addHit is called when the second mate arrives, so pair k of the call is the k-th COMPLETING record in
// arrival order.  Round 4 sorted (completing record's arrival index, value) once more (four radix passes over 3.9e8
// elements, half of them "completes nothing"); the walk now leaves the value AT the completing record's arrival index
// (`done`; a scatter inside the group's records), and the order is a compaction: completing records per tile of 64
// arrival positions, a scan over the tiles, a pass that writes every completing record to its rank.
__global__ __launch_bounds__(256) void flat_mate_done_tiles_kernel(FlatMateArgs f)
{
   const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
   const unsigned long long m = __ballot(r < f.n_reads && f.done[r] != 0ull);
   if ((threadIdx.x & 63u) == 0 && r < f.n_reads) f.tile_done[r >> 6] = (int32_t)__popcll(m);
}
__global__ __launch_bounds__(256) void flat_mate_order_kernel(FlatMateArgs f)
{
   const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
   const int lane = (int)(threadIdx.x & 63u);
   const unsigned long long d = r < f.n_reads ? f.done[r] : 0ull;
   const unsigned long long m = __ballot(d != 0ull);
   if (r >= f.n_reads) return;
   const int64_t n_pairs = f.tile_done_at[(f.n_reads + 63) >> 6];
   if (d != 0ull) {
      const int64_t k = f.tile_done_at[r >> 6] + __popcll(m & ((1ull << lane) - 1ull));
      f.pair_rec_w[k] = (uint32_t)r;
      f.pair_val_w[k] = (uint32_t)d;
   }
   if (r >= n_pairs) f.pair_rec_w[r] = 0xFFFFFFFFu; // (position r of the list, not record r: from the last pair on)
}

// pair k (the k-th completing record in arrival order): its mates' feature counts; and where every cluster's pairs begin
__global__ __launch_bounds__(256) void flat_mate_count_kernel(FlatMateArgs f)
{
   const MateArgs &a = f.a;
   const int64_t k = xcd_tile() * 256 + threadIdx.x; // (XCD-aware tile order: device_common.h)
   if (k <= a.n_loci) { // the first pair whose completing record is this cluster's: the keys ascend, cluster by cluster
      const uint32_t want = (uint32_t)a.locus_read_off[k];
      int64_t lo = 0, hi = f.n_reads;
      while (lo < hi) {
         const int64_t mid = (lo + hi) >> 1;
         if (f.pair_rec[mid] < want) lo = mid + 1;
         else hi = mid;
      }
      f.locus_pair_off[k] = k == a.n_loci ? [&] { // (all pairs: the first key that completes nothing)
         int64_t l2 = 0, h2 = f.n_reads;
         while (l2 < h2) {
            const int64_t mid = (l2 + h2) >> 1;
            if (f.pair_rec[mid] != 0xFFFFFFFFu) l2 = mid + 1;
            else h2 = mid;
         }
         return l2;
      }() : lo;
   }
   int lf = 0, rf = 0; // (lanes beyond the end stay for the wave's sums)
   if (k < f.n_reads && f.pair_rec[k] != 0xFFFFFFFFu) {
      const int64_t r = f.pair_rec[k];
      const uint32_t v = f.pair_val[k];
      const bool me_right = (v >> 31) != 0;
      const int64_t w = (v & 0x7FFFFFFFu) == 0x7FFFFFFFu ? -1 : (int64_t)(v & 0x7FFFFFFFu);
      const int nf_me = mate_feature_count(a, r);
      const int nf_w = w >= 0 ? mate_feature_count(a, w) : 0;
      lf = me_right ? nf_w : nf_me;
      rf = me_right ? nf_me : nf_w;
   }
   if (k <= f.n_reads) f.lfeat[k] = lf, f.rfeat[k] = rf;
   int sl = lf, sr = rf; // (a wave's 64 k are one tile)
#pragma unroll
   for (int o = 32; o > 0; o >>= 1) {
      sl += __shfl_xor(sl, o);
      sr += __shfl_xor(sr, o);
   }
   if ((threadIdx.x & 63u) == 0 && k <= f.n_reads) f.tile_l[k >> 6] = sl, f.tile_r[k >> 6] = sr;
}

__global__ __launch_bounds__(256) void flat_mate_fill_kernel(FlatMateArgs f, int64_t n_pairs)
{
   const MateArgs &a = f.a;
   const int64_t k = xcd_tile() * 256 + threadIdx.x; // (XCD-aware tile order: device_common.h)
   // the pair's first features: its tile's (the scans over the tiles) and the features of the pairs in front of it in the wave
   const int lane = (int)(threadIdx.x & 63u);
   const int lf = k <= f.n_reads ? f.lfeat[k] : 0, rf = k <= f.n_reads ? f.rfeat[k] : 0;
   int il = lf, ir = rf;
#pragma unroll
   for (int o = 1; o < 64; o <<= 1) {
      const int tl = __shfl_up(il, o), tr = __shfl_up(ir, o);
      if (lane >= o) il += tl, ir += tr;
   }
   if (k > n_pairs) return;
   const int64_t lo = f.lscan[k >> 6] + (il - lf), ro = f.rscan[k >> 6] + (ir - rf);
   a.left_off[k] = lo;
   a.right_off[k] = ro;
   if (k == n_pairs) return; // (the entry beyond the last pair: the totals)
   const int64_t r = f.pair_rec[k];
   const uint32_t v = f.pair_val[k];
   const bool me_right = (v >> 31) != 0;
   const int64_t w = (v & 0x7FFFFFFFu) == 0x7FFFFFFFu ? -1 : (int64_t)(v & 0x7FFFFFFFu);
   const int64_t rl = me_right ? w : r, rr = me_right ? r : w; // the left / right mate's record (-1: none)
   if (rl >= 0) write_mate(a, rl, a.left_code, a.left_left, a.left_right, lo);
   if (rr >= 0) write_mate(a, rr, a.right_code, a.right_left, a.right_right, ro);
   // the reads' masses (src/read.cpp:49-53, 734-741)
   double m;
   if (w >= 0) m = 0.5 / (double)a.nh[rl] + 0.5 / (double)a.nh[rr];
   else m = 1.0 / (double)a.nh[r];
   a.pair_mass[k] = m;
}

} // namespace sb
