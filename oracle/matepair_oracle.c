/* oracle/matepair_oracle.c -- TEST INFRASTRUCTURE ONLY: a plain-C restatement of the reference's mate pairing,
 * HitCluster::addOpenHit + addHit (/root/reference/src/alignments.cpp:423-461, 490-650), for one cluster:
 *   :512-518  a record spanning more than kMaxFragSpan (src/common.cpp:17) is refused
 *   :535-545  a single read (no partner, ReadHit::is_singleton src/read.cpp:170-178, or partner on another
 *             reference) is a hit of its own: PairedHit(nullptr, hit) on the reverse strand, else (hit, nullptr)
 *   :547-585  no open mate of this read id: the record waits -- as the left mate when its partner lies behind it,
 *             as the right one when before it; refused when both start at one position
 *   :586-628  else the read id's open mates are tried oldest first: same start as this record's partner, strands
 *             agreeing or unknown, expecting its partner where this record starts -> the pair is complete: addHit
 *   :629-641  no match: the record waits as well
 *   clearOpenMates (:653): what still waits at the end is dropped
 * Pair mass: the reads' masses, 0.5 / NH each, 1 / NH for a single read (src/read.cpp:49-53, 734-741).
 * Pinned by tests/test_matepair_oracle.py against the reference's own HitCluster (oracle/ref_shim.cpp:
 * ref_cluster_from_records), through the unique hits its collapse leaves.  Only tests/ may use this file. */
#include "matepair_oracle.h"

#include <stdlib.h>

#define SBO_MAX_FRAG_SPAN 1000000

int sbo_pair_mates(int n_reads, const uint64_t *read_id, const int64_t *block_off, const uint32_t *block_left,
                   const uint32_t *block_right, const uint32_t *partner_pos, const uint8_t *flags, const int32_t *nh,
                   int32_t *left_rec, int32_t *right_rec, double *mass, int32_t counts[4])
{
   int n_pairs = 0;
   counts[0] = counts[1] = counts[2] = counts[3] = 0;
   int *open = (int *)malloc((size_t)(n_reads > 0 ? n_reads : 1) * sizeof(int)); /* waiting records, oldest first */
   int n_open = 0;
   for (int r = 0; r < n_reads; ++r) {
      const int64_t b0 = block_off[r], b1 = block_off[r + 1];
      if (b1 <= b0) {
         ++counts[2];
         continue;
      }
      const uint32_t left = block_left[b0], right = block_right[b1 - 1];
      if ((long long)right - (long long)left > SBO_MAX_FRAG_SPAN) { /* :512 */
         ++counts[2];
         continue;
      }
      const uint32_t ppos = partner_pos[r];
      if (ppos == 0 || (flags[r] & 2u)) { /* :535 */
         if (flags[r] & 1u) left_rec[n_pairs] = -1, right_rec[n_pairs] = r;
         else left_rec[n_pairs] = r, right_rec[n_pairs] = -1;
         mass[n_pairs] = 1.0 / nh[r];
         ++n_pairs;
         ++counts[1];
         continue;
      }
      const int strand = (flags[r] >> 2) & 3;
      int hit = -1;
      for (int o = 0; o < n_open && hit < 0; ++o) { /* the chain of this read id, in insertion order */
         const int w = open[o];
         if (read_id[w] != read_id[r]) continue;
         const int ws = (flags[w] >> 2) & 3;
         const int strand_agree = ws == strand || strand == 0 || ws == 0;
         if (block_left[block_off[w]] == ppos && strand_agree && partner_pos[w] == left) hit = o; /* :603-606 */
      }
      if (hit >= 0) {
         const int w = open[hit];
         const int waiting_is_left = partner_pos[w] > block_left[block_off[w]]; /* how it was opened (:559-585) */
         const int l = waiting_is_left ? w : r, rr = waiting_is_left ? r : w;
         left_rec[n_pairs] = l, right_rec[n_pairs] = rr;
         mass[n_pairs] = 0.5 / nh[l] + 0.5 / nh[rr];
         ++n_pairs;
         ++counts[0];
         for (int o = hit; o + 1 < n_open; ++o) open[o] = open[o + 1];
         --n_open;
         continue;
      }
      if (ppos == left) { /* :585, :640 */
         ++counts[2];
         continue;
      }
      open[n_open++] = r;
   }
   counts[3] = n_open;
   free(open);
   return n_pairs;
}

/* The read stream of quant mode (Sample::nextClusterRefDemand, src/alignments.cpp:1145-1187), as the loop it is:
 * `cur` is the hit factory's position; a cluster takes records until one lies behind it (that one is "rewound":
 * looked at again by the next cluster). */
void sbo_assign_reads(int n_clusters, const int32_t *c_ref, const uint32_t *c_left, const uint32_t *c_right, const uint8_t *c_strand,
                      int64_t n_reads, const int32_t *r_ref, const uint32_t *r_left, const uint32_t *r_right, const uint8_t *r_xs,
                      int32_t *read_cluster, int64_t *off)
{
   for (int64_t i = 0; i < n_reads; ++i) read_cluster[i] = -1;
   int64_t cur = 0;
   for (int k = 0; k < n_clusters; ++k) {
      off[k] = cur;
      while (cur < n_reads) { /* recordsRemain() */
         const int64_t i = cur++;
         /* hit_lt_cluster(hit, cluster, 0), alignments.cpp:32-37 */
         const int lt = r_ref[i] != c_ref[k] ? r_ref[i] < c_ref[k] : r_right[i] < c_left[k];
         /* hit_gt_cluster(hit, cluster, 0), :39-49 */
         const int gt = r_ref[i] != c_ref[k] ? r_ref[i] > c_ref[k] : r_left[i] > c_right[k];
         if (lt) {
            /* the hit lies before this region: passed over */
         } else if (gt) {
            --cur; /* rewindHit() */
            break;
         } else if (r_xs[i] != 0 && r_xs[i] != c_strand[k]) {
            /* :1168: a known strand other than the cluster's */
         } else {
            read_cluster[i] = k; /* clusterOut.addOpenHit(new_hit, false, false) */
         }
      }
   }
   off[n_clusters] = cur;
}
