/* oracle/bamdecode_oracle.c -- TEST INFRASTRUCTURE ONLY: a plain-C restatement of how the reference turns one BAM
 * alignment record into a ReadHit, for checking sbgpu_bam_decode_* (strawberry_amd/csrc/bamdecode_api.hip).  Nothing
 * in the product links or calls this file.
 *
 * Follows, line by line:
 *   BAMHitFactory::getHitFromBuf           /root/reference/src/read.cpp:480-715
 *   ReadTable::get_id / hashString (FNV-1) /root/reference/src/read.cpp:214-219, include/read.hpp:164-173
 *   ReadHit::ReadHit, mass, is_singleton   /root/reference/src/read.cpp:24-54, 170-178
 *   ReadHit::read_len                      /root/reference/src/read.cpp:61-80
 *   readhit_2_genomicFeats                 /root/reference/src/contig.cpp:12-53
 *   bam_aux_get, __skip_tag, bam_aux2i, bam_aux2A, bam_aux_type2size
 *                                          /root/reference/external/samtools-0.1.19/bam_aux.c:28-47,163-201, bam.h:772-778
 *   the record layout (bam1_core_t, bam1_cigar / bam1_aux)  external/samtools-0.1.19/bam.h:225-269, bam.c (bam_read1)
 *
 * Pinned against the reference itself: oracle/ref_shim.cpp's ref_bam_decode runs the reference's own BAMHitFactory
 * over the same file (tests/test_bamdecode.py, tools/make_bamdecode_golden.py -> tests/golden/bamdecode_cases.npz).
 *
 * Quirks kept on purpose:
 *   - an insertion or deletion is refused unless it is at least the THIRD kept operation (`i-1 <= 0`, :594): "10M2I10M"
 *     is refused, "3S10M2I10M" is kept;
 *   - hard clips and pads are not part of the kept CIGAR, so they do not count in that position test;
 *   - the ZF tag is read and never used (the constructor recomputes the mass, read.cpp:49-53);
 *   - NM goes through an unsigned char (:617,:649);
 *   - samtools 0.1.19 skips a `d` (double) tag as if it had no payload and parses the payload bytes as tags.
 * Every read stays inside the record: where the reference would run past it (a `d` tag at the very end), the scan
 * stops -- the bytes beyond a record are not the record's.                                                         */
#include "bamdecode_oracle.h"

#include <string.h>

static int32_t rd_i32(const uint8_t *p)
{
   uint32_t v = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
   return (int32_t)v;
}
static uint32_t rd_u32(const uint8_t *p) { return (uint32_t)rd_i32(p); }

int64_t sbo_bam_index(const uint8_t *bytes, int64_t n_bytes, int64_t *rec_off, int64_t cap)
{
   int64_t n = 0, p = 0;
   while (p < n_bytes) {
      if (p + 4 > n_bytes) return -1;
      const int32_t bs = rd_i32(bytes + p);
      if (bs < 0 || p + 4 + (int64_t)bs > n_bytes) return -1;
      if (n >= cap) return -1;
      rec_off[n++] = p;
      p += 4 + (int64_t)bs;
   }
   if (n > cap) return -1;
   rec_off[n] = p;
   return n;
}

static int aux_type2size(int x) /* bam.h:772-778 */
{
   if (x == 'C' || x == 'c' || x == 'A') return 1;
   if (x == 'S' || x == 's') return 2;
   if (x == 'I' || x == 'i' || x == 'f' || x == 'F') return 4;
   return 0;
}
static int up(int c) { return (c >= 'a' && c <= 'z') ? c - 32 : c; }

/* bam_aux_get: the position of the tag's type byte, or NULL */
static const uint8_t *aux_get(const uint8_t *s, const uint8_t *end, char t0, char t1)
{
   const int y = ((int)(uint8_t)t0 << 8) | (uint8_t)t1;
   while (s < end) {
      if (s + 1 >= end) return 0; /* (the reference reads s[1] regardless) */
      const int x = ((int)s[0] << 8) | s[1];
      s += 2;
      if (x == y) return s < end ? s : 0;
      if (s >= end) return 0;
      const int type = up(*s);
      ++s;
      if (type == 'Z' || type == 'H') {
         while (s < end && *s) ++s;
         ++s;
      } else if (type == 'B') {
         if (s + 5 > end) return 0;
         const int64_t count = rd_i32(s + 1);
         if (count < 0) return 0; /* (the reference would step backwards) */
         s += 5 + (int64_t)aux_type2size(*s) * count;
      } else {
         s += aux_type2size(type);
      }
   }
   return 0;
}
static int32_t aux2i(const uint8_t *s, const uint8_t *end) /* bam_aux.c:163-174 */
{
   if (!s) return 0;
   const int type = *s++;
   if (type == 'c') return s + 1 <= end ? (int32_t)(int8_t)s[0] : 0;
   if (type == 'C') return s + 1 <= end ? (int32_t)s[0] : 0;
   if (type == 's') return s + 2 <= end ? (int32_t)(int16_t)((uint16_t)s[0] | ((uint16_t)s[1] << 8)) : 0;
   if (type == 'S') return s + 2 <= end ? (int32_t)((uint16_t)s[0] | ((uint16_t)s[1] << 8)) : 0;
   if (type == 'i' || type == 'I') return s + 4 <= end ? rd_i32(s) : 0;
   return 0;
}

void sbo_bam_decode(const uint8_t *bytes, const int64_t *rec_off, int64_t n, const sbo_bam_opts *o, uint8_t *status,
                    uint64_t *read_id, int32_t *ref, uint32_t *left, uint32_t *right, uint8_t *strand, uint8_t *partner_same_ref,
                    uint32_t *partner_pos, int32_t *nm, int32_t *nh, uint32_t *sam_flag, uint8_t *singleton, double *mass,
                    int32_t *read_len, int64_t *cig_off, uint8_t *cig_type, uint32_t *cig_len, int64_t *feat_off,
                    uint8_t *feat_code, uint32_t *feat_left, uint32_t *feat_right, int32_t *any_paired)
{
   int64_t nc = 0, nf = 0;
   *any_paired = 0;
   cig_off[0] = feat_off[0] = 0;
   for (int64_t r = 0; r < n; ++r) {
      const uint8_t *rec = bytes + rec_off[r];
      const int32_t block_size = rd_i32(rec);
      const uint8_t *core = rec + 4, *data = rec + 36, *end = rec + 4 + block_size;
      int st = SBO_BAM_OK;
      read_id[r] = 0, ref[r] = -1, left[r] = right[r] = 0, strand[r] = 0, partner_same_ref[r] = 0, partner_pos[r] = 0;
      nm[r] = 0, nh[r] = 1, sam_flag[r] = 0, singleton[r] = 0, mass[r] = 0.0, read_len[r] = 0;
      const int64_t c0 = nc, f0 = nf;
      do {
         if (block_size < 32) {
            st = SBO_BAM_TRUNCATED;
            break;
         }
         const int32_t tid = rd_i32(core), pos0 = rd_i32(core + 4);
         const uint32_t bin_mq_nl = rd_u32(core + 8), flag_nc = rd_u32(core + 12);
         const int32_t l_qseq = rd_i32(core + 16), mtid = rd_i32(core + 20), mpos0 = rd_i32(core + 24);
         const int l_qname = (int)(bin_mq_nl & 0xff), n_cigar = (int)(flag_nc & 0xffff);
         const uint32_t flag = flag_nc >> 16;
         sam_flag[r] = flag;
         if (l_qseq < 0 || 32 + (int64_t)l_qname + 4 * (int64_t)n_cigar + ((int64_t)l_qseq + 1) / 2 + (int64_t)l_qseq > (int64_t)block_size) {
            st = SBO_BAM_TRUNCATED;
            break;
         }
         /* :504 the read id: FNV-1 over the name up to its NUL (`hash ^= *s` with a signed char) */
         {
            uint64_t h = 0xcbf29ce484222325ull;
            for (int k = 0; k < l_qname && data[k]; ++k) {
               h *= 1099511628211ull;
               h ^= (uint64_t)(int64_t)(int8_t)data[k];
            }
            read_id[r] = h;
         }
         if ((flag & 0x4) || tid < 0) { /* :508 */
            st = SBO_BAM_UNMAPPED;
            break;
         }
         if (o->n_ref > 0 && tid >= o->n_ref) { /* :531 */
            st = SBO_BAM_BAD_REF;
            break;
         }
         /* :536-587 the CIGAR, operation by operation; H and P are dropped from what the ReadHit keeps */
         const uint8_t *cig = data + l_qname;
         int64_t rlen = 0, eff = 0;
         for (int i = 0; i < n_cigar && st == SBO_BAM_OK; ++i) {
            const uint32_t w = rd_u32(cig + 4 * i);
            const int32_t length = (int32_t)(w >> 4);
            if (length <= 0) {
               st = SBO_BAM_ZERO_OP;
               break;
            }
            switch (w & 0xf) {
            case 0: /* BAM_CMATCH */
               rlen += length, eff += length;
               cig_type[nc] = 0, cig_len[nc] = (uint32_t)length, ++nc;
               break;
            case 1: /* BAM_CINS */
               cig_type[nc] = 1, cig_len[nc] = (uint32_t)length, ++nc;
               break;
            case 2: /* BAM_CDEL */
               rlen += length;
               cig_type[nc] = 2, cig_len[nc] = (uint32_t)length, ++nc;
               break;
            case 4: /* BAM_CSOFT_CLIP */
               cig_type[nc] = 4, cig_len[nc] = (uint32_t)length, ++nc;
               break;
            case 5: /* BAM_CHARD_CLIP */
            case 6: /* BAM_CPAD */
               break;
            case 3: /* BAM_CREF_SKIP */
               rlen += length;
               cig_type[nc] = 3, cig_len[nc] = (uint32_t)length, ++nc;
               if (length > o->max_intron) st = SBO_BAM_INTRON_LONG;
               else if (length < o->min_intron) st = SBO_BAM_INTRON_SHORT;
               break;
            default:
               st = SBO_BAM_OP;
               break;
            }
         }
         if (st != SBO_BAM_OK) break;
         /* :592-599 insertions and deletions between two matches, and not among the first two kept operations */
         const int64_t kept = nc - c0;
         for (int64_t i = 0; i < kept; ++i) {
            const int t = cig_type[c0 + i];
            if (t == 1 || t == 2) {
               if (i - 1 <= 0 || i + 1 >= kept) st = SBO_BAM_INDEL;
               else if (cig_type[c0 + i - 1] != 0 || cig_type[c0 + i + 1] != 0) st = SBO_BAM_INDEL;
               if (st != SBO_BAM_OK) break;
            }
         }
         if (st != SBO_BAM_OK) break;
         if (eff <= 1) { /* :601 */
            st = SBO_BAM_SHORT;
            break;
         }
         if (flag & 0x1) *any_paired = 1; /* :605-607 SINGLE_END_EXP = false */
         /* :619-634 the transcription strand: the XS tag ... */
         const uint8_t *aux = data + l_qname + 4 * (int64_t)n_cigar + l_qseq + (l_qseq + 1) / 2;
         int sd = 0;
         {
            const uint8_t *p = aux_get(aux, end, 'X', 'S');
            if (p && p + 1 < end && p[0] == 'A') { /* bam_aux2A: any other type gives 0 */
               if (p[1] == '+') sd = 1;
               else if (p[1] == '-') sd = 2;
            }
         }
         /* :636-651 ... else the library type */
         const int rev = (flag & 0x10) != 0, fr = o->library == 1, rf = o->library == 2;
         if (sd == 0 && (fr || rf)) {
            const int toward = (rf && rev) || (fr && !rev);
            if (flag & 0x40) sd = toward ? 1 : 2;
            else sd = toward ? 2 : 1;
         }
         {
            const uint8_t *p = aux_get(aux, end, 'N', 'M'); /* :653-656, through an unsigned char */
            if (p) nm[r] = (int32_t)(uint8_t)aux2i(p, end);
            p = aux_get(aux, end, 'N', 'H'); /* :658-661 */
            if (p) nh[r] = aux2i(p, end);
         }
         if (o->unique_only && (nh[r] > 1 || (flag & 0x100))) { /* :670 */
            st = SBO_BAM_MULTI;
            break;
         }
         /* :692-704 the ReadHit */
         const uint32_t pos = (uint32_t)pos0 + 1u, mate_pos = (uint32_t)mpos0 + 1u;
         ref[r] = tid;
         left[r] = pos;
         right[r] = pos + (uint32_t)rlen - 1u;
         strand[r] = (uint8_t)sd;
         const int32_t partner_ref = mtid < 0 ? -1 : mtid; /* "*" is not in the table */
         partner_same_ref[r] = partner_ref == tid;
         partner_pos[r] = mate_pos;
         singleton[r] = (mate_pos == 0 || partner_ref == -1 || partner_ref != tid); /* read.cpp:170-178 */
         mass[r] = singleton[r] ? 1.0 / nh[r] : 0.5 / nh[r];                        /* read.cpp:49-53 */
         /* ReadHit::read_len and readhit_2_genomicFeats over the kept CIGAR */
         uint32_t offset = pos;
         int32_t qlen = 0;
         for (int64_t i = 0; i < kept; ++i) {
            const int t = cig_type[c0 + i];
            const uint32_t len = cig_len[c0 + i];
            if (t == 0 || t == 4 || t == 1) qlen += (int32_t)len;
            if (t == 0 || t == 3) {
               feat_code[nf] = t == 0 ? 0 : 1;
               feat_left[nf] = offset;
               feat_right[nf] = offset + len - 1u;
               ++nf;
               offset += len;
            } else if (t == 2) {
               feat_right[nf - 1] += len; /* feats.back()._len += length (contig.cpp:33) */
               offset += len;
            }
         }
         read_len[r] = qlen;
      } while (0);
      if (st != SBO_BAM_OK) nc = c0, nf = f0; /* nothing of a refused record stays */
      status[r] = (uint8_t)st;
      cig_off[r + 1] = nc;
      feat_off[r + 1] = nf;
   }
}
