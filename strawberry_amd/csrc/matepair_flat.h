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
   // the positional form (flat_mate_match_kernel): no sort at all where the records come sorted by position
   uint32_t *lefts;                 // per record, arrival order: its first block's left end (what the search looks at)
   uint32_t *claim;                 // per record: 1 + the arrival index of the record that completes it (nullptr in the sorted form)
   uint32_t *slot;                  // TWO per record, zeroed: the runs as hash tables of their openers (1 + arrival index)
   uint32_t *runlen;                // per record, written at the runs' first records: the run's length
   uint32_t *trouble;               // [1], zeroed: kPosUnsorted | kPosConflict -- the call is then served by the sorted form
};
enum : uint32_t { kPosUnsorted = 1u, kPosConflict = 2u };

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

// ---- The positional form (round 6).  The reference's match rule is positional (src/alignments.cpp:612-615: a waiting mate fits
// when it.left_pos() == hit_partner_pos && expected_pos == hit->left(), same read id, strands agree), the records of a cluster
// arrive sorted by position, and a record's mate position says where its partner lies: a record whose mate lies BEFORE it
// (partner_pos < left: the closing mate of a properly oriented pair) finds the records that start at partner_pos by a search
// in its own cluster's records and looks for its read id among them.  No record has to be moved: the sort that brought the
// records of a read id together (one radix sort over all records, a gather into sorted order, a walk: 26 of the stage's 47 ms at
// 3.9e8 records) is not needed.
//
// Exactness.  Call x and y RELATED when they are records of one cluster and one read id, eligible for pairing, x before y in
// arrival order, x.left == y.partner_pos, x.partner_pos == y.left and their strands agree -- the reference's test.  addOpenHit
// pairs y with the OLDEST waiting record related to it (:593-641); when every record is related to at most one other, the
// relation IS the reference's result, whatever else the read id's chain holds: x had nobody to complete when it arrived
// (its only relation arrives later), so it waits; y finds it, first fit or not, because nothing else in the chain fits.  A
// record with two relations -- a closing mate that finds two fitting openers, an opener two closers fit: reads aligned more
// than once in a cluster, --allow-multimapped-hits -- makes the outcome depend on the chain's order; the kernel then raises
// kPosConflict and the whole call is served by the sorted form below, which walks the chains in arrival order.  So is a call
// whose records do not ascend by left end inside a cluster (kPosUnsorted: the search would not be a search).  A record whose
// mate position is its own left end never waits and never completes (:585, :640 -- neither does anything related to it).
__device__ __forceinline__ bool flat_mate_eligible(const FlatRec &q) // offered to addOpenHit, not refused, not a single read
{
   const uint32_t fl = q.misc & 255u;
   return !(fl & 16u) && (q.misc & 256u) && (int64_t)q.right - (int64_t)q.left <= kMaxFragSpanDev && q.ppos != 0 && !(fl & 2u);
}

// The records that start at one position of one cluster are a RUN of the arrival order [start, end).  Looking through a run for
// a read id costs its length, and a sample's deep positions hold thousands of records (every closing mate looking through all
// of them: 58 ms at 3.9e8 records, the sort it was to replace took 26).  So the run doubles as a hash table: `slot[2 start + k]`,
// k = hash(read id) mod twice the run's length (two slots per record: a run of openers alone -- thousands of copies of one
// fragment -- is then half full), linear probing inside the run -- an opener (a record that waits: its partner lies behind it)
// puts its arrival index there, a closing mate probes from the same k until it meets an empty slot.  Everything a thread touches
// lies inside the run.
//
// Where a run begins and ends: a record is a HEAD when it is its cluster's first or starts elsewhere than the record before it;
// a wave's ballot of heads answers for the runs that begin / end inside the wave's 64 records, a search for the others (in the
// call's own arrays: `lefts` is being written by this very launch).
__device__ __forceinline__ uint32_t flat_left_of(const MateArgs &a, int64_t r)
{
   const int64_t b = a.block_off[r];
   return b < a.block_off[r + 1] ? a.block_left[b] : 0u; // (a record without blocks: such a call is kPosUnsorted anyway)
}

// ONE pass over the records in arrival order: the record as the rules see it, its left end for the searches, the heads, the
// openers into their runs' tables (`slot` zeroed by the caller), the runs' lengths at their heads; kPosUnsorted where the
// left ends of a cluster do not ascend or a record has no blocks.
__global__ __launch_bounds__(256) void flat_mate_rec_kernel(FlatMateArgs f)
{
   const int64_t r = xcd_tile() * 256 + threadIdx.x; // (XCD-aware tile order: a run's slots stay in ONE L2, the closing mates' too)
   const int lane = (int)(threadIdx.x & 63u);
   const int64_t r0 = r - lane;
   if (r0 >= f.n_reads) return; // (the whole wave)
   const int64_t lo = wave_range_of(f.a.locus_read_off, f.a.n_loci, r0, f.n_reads); // last cluster whose records begin at or before r
   const MateArgs &a = f.a;
   const bool in = r < f.n_reads;
   bool blocks = false;
   FlatRec q = {};
   if (in) {
      const uint64_t rid = a.read_id[r];
      const int64_t b0 = a.block_off[r], b1 = a.block_off[r + 1];
      blocks = b1 > b0;
      q.rid_lo = (uint32_t)rid, q.rid_hi = (uint32_t)(rid >> 32);
      q.left = blocks ? a.block_left[b0] : 0u;
      q.right = blocks ? a.block_right[b1 - 1] : 0u;
      q.ppos = a.partner_pos[r];
      q.misc = (uint32_t)a.flags[r] | (blocks ? 256u : 0u) | ((uint32_t)lo << 9);
      f.rec_arr[r] = q;
      f.lefts[r] = q.left;
   }
   // against the record before (the wave's first lane: a load): heads, and the order the searches need
   uint32_t prev_left = (uint32_t)__shfl_up((int)q.left, 1);
   int64_t prev_lo = __shfl_up(lo, 1);
   bool prev_blocks = __shfl_up((int)blocks, 1) != 0;
   if (lane == 0 && r > 0) {
      const int64_t p0 = a.block_off[r - 1], p1 = a.block_off[r];
      prev_blocks = p1 > p0;
      prev_left = prev_blocks ? a.block_left[p0] : 0u;
      prev_lo = a.locus_read_off[lo] == r ? lo - 1 : lo; // (the record before a cluster's first belongs to another cluster)
   }
   const bool bad = in && (!blocks || (r > 0 && prev_lo == lo && (!prev_blocks || prev_left > q.left)));
   if (__ballot(bad) && lane == 0) atomicOr(f.trouble, kPosUnsorted);
   const bool head = in && (r == 0 || prev_lo != lo || prev_left != q.left);
   const unsigned long long heads = __ballot(head);
   const bool opener = in && flat_mate_eligible(q) && q.ppos > q.left;
   if (!head && !opener) return;
   const int64_t c0 = a.locus_read_off[lo], c1 = a.locus_read_off[lo + 1];
   // the run's end: the next head behind this record -- in the wave, or where the left ends first exceed this one
   int64_t end;
   const unsigned long long above = lane < 63 ? heads & (~0ull << (lane + 1)) : 0ull;
   if (above) {
      end = r0 + (__ffsll((long long)above) - 1);
   } else {
      int64_t at = min(r0 + 63, f.n_reads - 1); // the wave's last record: of this run (no head behind this lane)
      if (at + 1 >= c1) end = c1;
      else {
         int64_t step = 64;
         while (at + step < c1 && flat_left_of(a, at + step) == q.left) at += step, step <<= 1;
         int64_t hi = min(at + step, c1); // not of the run (or the cluster's end)
         while (hi - at > 1) {
            const int64_t mid = (at + hi) >> 1;
            if (flat_left_of(a, mid) == q.left) at = mid;
            else hi = mid;
         }
         end = hi;
      }
   }
   if (head) f.runlen[r] = (uint32_t)(end - r);
   if (!opener) return;
   // the run's start: the last head at or before this record
   int64_t start;
   const unsigned long long upto = heads & (lane < 63 ? ((1ull << (lane + 1)) - 1ull) : ~0ull);
   if (upto) {
      start = r0 + (63 - __clzll((long long)upto));
   } else {
      int64_t hi = r0, step = 64; // r0 is of the run (no head up to this lane); the first such index >= c0
      int64_t lo2 = hi - step;
      while (lo2 >= c0 && flat_left_of(a, lo2) == q.left) hi = lo2, step <<= 1, lo2 = hi - step;
      if (lo2 < c0) lo2 = c0 - 1; // (a virtual record before the cluster's first: not of the run)
      while (hi - lo2 > 1) {      // lo2: not of the run; hi: of the run
         const int64_t mid = (lo2 + hi) >> 1;
         if (flat_left_of(a, mid) == q.left) hi = mid;
         else lo2 = mid;
      }
      start = hi;
   }
   const uint32_t len = 2u * (uint32_t)(end - start);
   const uint64_t rid = ((uint64_t)q.rid_hi << 32) | q.rid_lo;
   int64_t at = 2 * start + (int64_t)(flat_hash32(rid) % len);
   for (uint32_t tries = 0; tries < len; ++tries) {
      if (atomicCAS(&f.slot[at], 0u, (uint32_t)r + 1u) == 0u) break;
      if (++at == 2 * end) at = 2 * start;
   }
}

// lower_bound of `want` in lefts[lo0 .. hi0) (ascending), a lane by itself: gallop back from hi0, then bisect
__device__ __forceinline__ int64_t flat_lower_bound_back(const uint32_t *__restrict__ lefts, int64_t lo0, int64_t hi0, uint32_t want)
{
   int64_t hi = hi0, step = 64; // lefts[hi] >= want (or hi == hi0)
   int64_t lo = hi0 - step;
   while (lo > lo0 && lefts[lo] >= want) {
      hi = lo;
      step <<= 1;
      lo = hi0 - step;
   }
   if (lo < lo0) lo = lo0;
   if (lo >= hi) return hi;
   if (lefts[lo] >= want) return lo; // (the range's first element)
   while (hi - lo > 1) {             // lefts[lo] < want <= lefts[hi]
      const int64_t mid = (lo + hi) >> 1;
      if (lefts[mid] < want) lo = mid;
      else hi = mid;
   }
   return hi;
}

__global__ __launch_bounds__(256) void flat_mate_match_kernel(FlatMateArgs f)
{
   const int64_t r = xcd_tile() * 256 + threadIdx.x; // (XCD-aware tile order, as the openers': the probes hit the L2 that holds the run)
   const int lane = (int)(threadIdx.x & 63u);
   const int64_t r0 = r - lane;
   int refused = 0, orphan = 0, single = 0, complete = 0;
   bool conflict = false;
   FlatRec q = {};
   int kind = 0; // 1: a closing mate (its partner lies before it): it searches
   unsigned long long done = 0ull; // what this record completes (written for EVERY record: nobody zeroed the array)
   if (r < f.n_reads) {
      q = f.rec_arr[r];
      const uint32_t fl = q.misc & 255u;
      if (fl & 16u) { // not this cluster's record (sbgpu_assign_reads_*): never offered to addOpenHit
      } else if (!(q.misc & 256u) || (int64_t)q.right - (int64_t)q.left > kMaxFragSpanDev) { // :512-518
         ++refused;
      } else if (q.ppos == 0 || (fl & 2u)) { // a single read (:535-545)
         done = (1ull << 32) | ((fl & 1u) ? 0x80000000u : 0u) | 0x7FFFFFFFu;
         ++single;
      } else if (q.ppos > q.left) { // its partner lies behind it: it waits (whoever completes it takes it off the orphans' count)
         ++orphan;
      } else if (q.ppos == q.left) { // :585, :640
         ++refused;
      } else {
         kind = 1;
      }
   }
   // ---- where the records that start at partner_pos begin.  A wave's closing mates are neighbours: their partners lie in ONE
   // stretch of the cluster's records, a few hundred to a few thousand records back.  The wave finds that stretch's beginning
   // 64 ways at a time (two or three coalesced loads), every lane then bisects inside it -- ten steps that stay in the
   // vector cache -- where a lane by itself galloped and bisected through sixteen dependent loads from further away.
   const unsigned long long closers = __ballot(kind == 1);
   if (closers) {
      const uint32_t cl = q.misc >> 9;
      const uint32_t cl0 = (uint32_t)__builtin_amdgcn_readlane((int)cl, __ffsll((long long)closers) - 1);
      const bool together = __ballot(kind == 1 && cl != cl0) == 0ull; // (the closing mates of the wave share a cluster)
      int64_t hi = -1; // lower_bound of q.ppos in this cluster's records before r
      if (together) {
         const int64_t c0 = f.a.locus_read_off[cl0];
         const uint32_t pmin = wave_min_u32(kind == 1 ? q.ppos : 0xFFFFFFFFu);
         // "below": a record that starts before pmin (the record before the cluster's first: virtually).  a_lo = the last record
         // before the wave that is below; every closing mate's answer lies behind it.
         // coarse: probes at strides of 64, 4096, ... back from the wave's first record, 64 at a time
         int64_t a_lo, a_hi; // a_lo is below; a_hi is not (or is the wave's first record): the last below lies in [a_lo, a_hi)
         {
            int64_t top = r0, stride = 64;
            for (;;) {
               const int64_t idx = top - (int64_t)(lane + 1) * stride;
               const bool below = idx < c0 || f.lefts[idx] < pmin;
               const unsigned long long m = __ballot(below); // (from some lane on: the lanes go back in time, `lefts` ascends)
               if (m) {
                  const int first = __ffsll((long long)m) - 1;
                  a_lo = top - (int64_t)(first + 1) * stride;
                  a_hi = top - (int64_t)first * stride;
                  break;
               }
               top -= 64 * stride;
               stride *= 64;
            }
            if (a_lo < c0) a_lo = c0 - 1;
         }
         // fine: 64 probes across what is left, until one record is
         while (a_hi - a_lo > 1) {
            const int64_t st = (a_hi - a_lo - 1 + 63) >> 6; // candidates a_lo + 1 .. a_hi - 1
            const int64_t idx = a_lo + 1 + (int64_t)lane * st;
            const bool below = idx < a_hi && f.lefts[idx] < pmin;
            const int nb = __popcll(__ballot(below)); // (the first nb lanes)
            if (nb == 0) {
               a_hi = a_lo + 1;
            } else {
               const int64_t next = a_lo + 1 + (int64_t)nb * st;
               a_lo = a_lo + 1 + (int64_t)(nb - 1) * st;
               if (next < a_hi) a_hi = next;
            }
         }
         if (kind == 1) { // lefts[a_lo] < pmin <= partner_pos: the answer lies in (a_lo, r]
            int64_t lo = a_lo;
            hi = r;
            while (hi - lo > 1) {
               const int64_t mid = (lo + hi) >> 1;
               if (f.lefts[mid] < q.ppos) lo = mid;
               else hi = mid;
            }
         }
      } else if (kind == 1) {
         hi = flat_lower_bound_back(f.lefts, f.a.locus_read_off[cl], r, q.ppos);
      }
      if (kind == 1) {
         const uint32_t fl = q.misc & 255u;
         const int strand = (int)(fl >> 2) & 3;
         int64_t hit = -1;
         int n_fit = 0;
         if (hi < r && f.lefts[hi] == q.ppos) { // a run starts there: probe it for this read id
            const int64_t start = hi, end = hi + (int64_t)f.runlen[hi];
            const uint32_t len = 2u * (uint32_t)(end - start);
            const uint64_t rid = ((uint64_t)q.rid_hi << 32) | q.rid_lo;
            int64_t at = 2 * start + (int64_t)(flat_hash32(rid) % len);
            for (uint32_t tries = 0; tries < len; ++tries) {
               const uint32_t v = f.slot[at];
               if (v == 0u) break;
               const int64_t t = (int64_t)v - 1;
               const FlatRec w = f.rec_arr[t]; // (an opener: eligible, its partner behind it)
               const int wstrand = (int)((w.misc & 255u) >> 2) & 3;
               if (w.rid_lo == q.rid_lo && w.rid_hi == q.rid_hi && w.ppos == q.left && (wstrand == strand || strand == 0 || wstrand == 0))
                  if (n_fit++ == 0) hit = t;
               if (++at == 2 * end) at = 2 * start;
            }
         }
         if (n_fit == 1) {
            // an opener that two closing mates fit is a conflict: each writes its own index here (a plain store: no atomic round
            // trip per pair), and flat_mate_count_kernel, which visits every pair, checks that the opener still names ITS closing mate
            f.claim[hit] = (uint32_t)r + 1u;
            done = (1ull << 32) | 0x80000000u | (uint32_t)hit; // (the waiting mate is the left one, :559-585)
            ++complete;
            --orphan;
         } else if (n_fit == 0) {
            ++orphan; // waits for a partner that never comes
         } else {
            conflict = true; // two openers fit: which one is the oldest WAITING one is the chain's business
         }
      }
   }
   if (r < f.n_reads) f.done[r] = done;
   if (__ballot(conflict) && (threadIdx.x & 63u) == 0) atomicOr(f.trouble, kPosConflict);
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
      // (the positional form: two closing mates that took the same opener -- the second store won)
      if (f.claim && w >= 0 && f.claim[w] != (uint32_t)r + 1u) atomicOr(f.trouble, kPosConflict);
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
