// strawberry_amd/csrc/bamdecode_device.h -- BAM alignment records -> the read stream (the part of it
// that sits in front of the clusters: BAMHitFactory::getHitFromBuf, /root/reference/src/read.cpp:480-715).
//
// The reference reads a BAM record (samtools 0.1.19's bam_read1), then decides -- on the flag word, the CIGAR, three
// auxiliary tags and four option globals -- whether the record becomes a ReadHit and with which interval, strand, mate
// position, NH and read id.  Integer and byte work on ~150-400 bytes per record, every record by itself: one lane per
// record here, two passes (decide + count the aligned blocks; compact the accepted records and write their blocks), two
// scans over the waves' totals (tiles of 64 records) between them.  The same decoder body serves the host entry (sbgpu_bam_decode_host).
//
// What is decoded is the UNCOMPRESSED record stream (after BGZF inflate, which stays with the caller -- zlib on host
// cores) behind the BAM header: int32 block_size, the 32-byte core, read name, CIGAR, sequence, qualities, tags.
//
// Kept from the reference, quirks included (oracle/bamdecode_oracle.c states them line by line):
//   - refused: unmapped (flag 0x4 or no reference); a CIGAR operation of length 0; an operation other than M I D N S H P;
//     an N outside [min_intron, max_intron]; an I or D that is not between two M or that is among the first two KEPT
//     operations (H and P are not kept); at most one aligned base; NH > 1 or a secondary alignment while unique_only;
//   - the interval is [pos + 1, pos + M + D + N lengths]; the aligned blocks are the M runs, a D extends the block in front
//     of it, an I leaves two blocks that touch (sbgpu_pair_mates_* write no INTRON between touching blocks);
//   - strand: the XS:A tag ('+' / '-'; any other type or value: unknown), else the library type (fr / rf) with the
//     first-in-pair and reverse bits (:636-651);
//   - read id: FNV-1 of the name with the bytes as signed chars (include/read.hpp:164-173);
//   - tags are found the way samtools 0.1.19 finds them (bam_aux.c:28-47): types upper-cased, sizes 1 / 2 / 4, Z and H to
//     their NUL, B by its count -- and a `d` as if it had no payload.  No read leaves the record.
#pragma once

#include <stdint.h>

#include "../../include/sbgpu.h"

#ifdef __HIPCC__
#define SB_HD __host__ __device__ __forceinline__
#else
#define SB_HD inline
#endif

namespace sb {

constexpr int kBamInlineBlocks = 3; // a record's first aligned blocks travel with its scalars (more: the CIGAR is walked again)
struct BamRead {
   uint64_t read_id;
   int32_t ref;
   uint32_t left, right, partner_pos, sam_flag;
   int32_t nh, nm, read_len, n_blocks;
   uint32_t bl[kBamInlineBlocks], br[kBamInlineBlocks];
   uint8_t status, flags, paired;
};

SB_HD uint32_t bam_u32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
SB_HD int32_t bam_i32(const uint8_t *p) { return (int32_t)bam_u32(p); }
SB_HD int bam_aux_size(int x) // bam_aux_type2size, samtools-0.1.19/bam.h:772-778
{
   if (x == 'C' || x == 'c' || x == 'A') return 1;
   if (x == 'S' || x == 's') return 2;
   if (x == 'I' || x == 'i' || x == 'f' || x == 'F') return 4;
   return 0;
}

// How the decoder reads the record's bytes: EIGHT AT A TIME, at the places where every record asks for them.  peek8(p, end)
// returns bytes p .. p + 7 as a little-endian word.  ByteLoads: byte loads, nothing at or beyond `end` (host, and records
// walked in global memory).  ByteWindow: the two aligned 8-byte words around p, shifted together -- for records inside
// the kernels' staging buffer, where up to 15 bytes beyond a record's end are the buffer's own.
// (Round 5's window kept the aligned word around the LAST byte asked for and reloaded it under a branch when the walk left
// it: with 64 lanes at 64 alignments some lane reloads at nearly every byte, so the wave waited an LDS round trip per BYTE
// of names, CIGARs and tags -- 6.4 of the 10.7 ms of the one-pass kernel at 10^8 records.  Now the loads sit at uniform
// places -- four for the core, one per eight name bytes, one per CIGAR operation, one per tag -- and the bytes between them
// come out of registers.)
struct ByteLoads {
   SB_HD uint64_t peek8(const uint8_t *p, const uint8_t *end) const
   {
      uint64_t w = 0;
      for (int i = 0; i < 8; ++i)
         if (p + i < end) w |= (uint64_t)p[i] << (8 * i);
      return w;
   }
};
struct ByteWindow {
   SB_HD uint64_t peek8(const uint8_t *p, const uint8_t *) const
   {
      const uintptr_t q = (uintptr_t)p;
#if defined(__HIP_DEVICE_COMPILE__)
      // (the buffer is LDS and the pointer says so: ds_read2_b64 -- through the generic pointer the decoder's loads were
      // flat_load_dwordx2, which look the address space up per lane and count against both wait counters)
      const __attribute__((address_space(3))) uint64_t *b = (const __attribute__((address_space(3))) uint64_t *)(uint32_t)(q & ~(uintptr_t)7);
#else
      const uint64_t *b = (const uint64_t *)(q & ~(uintptr_t)7);
#endif
      const uint64_t lo = b[0], hi = b[1];
      const unsigned sh = 8u * (unsigned)(q & 7u);
      return sh ? (lo >> sh) | (hi << (64u - sh)) : lo;
   }
};
// bam_aux2i on a tag whose type byte and value are bytes 0 and 1.. of `w`; `s` is where the type byte sits
SB_HD int32_t aux_int_of(uint64_t w, const uint8_t *s, const uint8_t *end)
{
   const int type = (int)(w & 0xffu);
   const uint64_t v = w >> 8;
   ++s;
   if (type == 'c') return s + 1 <= end ? (int32_t)(int8_t)(v & 0xffu) : 0;
   if (type == 'C') return s + 1 <= end ? (int32_t)(v & 0xffu) : 0;
   if (type == 's') return s + 2 <= end ? (int32_t)(int16_t)(uint16_t)(v & 0xffffu) : 0;
   if (type == 'S') return s + 2 <= end ? (int32_t)(v & 0xffffu) : 0;
   if (type == 'i' || type == 'I') return s + 4 <= end ? (int32_t)(uint32_t)(v & 0xffffffffu) : 0;
   return 0;
}

// One record: `rec` points at its block_size word, `avail` bytes are the record's (to the next record's start).
template <class RD>
SB_HD void bam_decode_record(RD &rd, const uint8_t *rec, int64_t avail, const sbgpu_bam_opts_t &o, BamRead &out)
{
   out.read_id = 0, out.ref = -1, out.left = out.right = out.partner_pos = out.sam_flag = 0;
   out.nh = 1, out.nm = 0, out.read_len = 0, out.n_blocks = 0, out.flags = 0, out.paired = 0;
   out.status = SBGPU_BAM_TRUNCATED;
   if (avail < 36) return;
   // the size word and the 32-byte core: four loads, all on their way together
   const uint8_t *stop = rec + avail; // (ByteLoads: nothing at or beyond it is read)
   const uint64_t c0 = rd.peek8(rec, stop), c1 = rd.peek8(rec + 8, stop), c2 = rd.peek8(rec + 16, stop), c3 = rd.peek8(rec + 24, stop);
   const int32_t block_size = (int32_t)(uint32_t)c0;
   if (block_size < 32 || (int64_t)block_size + 4 > avail) return;
   const uint8_t *data = rec + 36, *end = rec + 4 + block_size;
   const int32_t tid = (int32_t)(uint32_t)(c0 >> 32), pos0 = (int32_t)(uint32_t)c1;
   const uint32_t bin_mq_nl = (uint32_t)(c1 >> 32), flag_nc = (uint32_t)c2;
   const int32_t l_qseq = (int32_t)(uint32_t)(c2 >> 32), mtid = (int32_t)(uint32_t)c3, mpos0 = (int32_t)(uint32_t)(c3 >> 32);
   const int l_qname = (int)(bin_mq_nl & 0xffu), n_cigar = (int)(flag_nc & 0xffffu);
   const uint32_t flag = flag_nc >> 16;
   out.sam_flag = flag;
   if (l_qseq < 0 || 32 + (int64_t)l_qname + 4 * (int64_t)n_cigar + ((int64_t)l_qseq + 1) / 2 + (int64_t)l_qseq > (int64_t)block_size) return;
   { // :504 ReadTable::get_id: FNV-1 over the name up to its NUL, the bytes as signed chars -- eight bytes per load
      uint64_t h = 0xcbf29ce484222325ull;
      bool open = true;
      for (int k = 0; k < l_qname && open; k += 8) {
         const uint64_t w = rd.peek8(data + k, end);
         for (int i = 0; i < 8; ++i) {
            const uint32_t c = (uint32_t)(w >> (8 * i)) & 0xffu;
            open = open && k + i < l_qname && c != 0u;
            h = open ? (h * 1099511628211ull) ^ (uint64_t)(int64_t)(int8_t)c : h;
         }
      }
      out.read_id = h;
   }
   if ((flag & 0x4u) || tid < 0) { // :508
      out.status = SBGPU_BAM_UNMAPPED;
      return;
   }
   if (o.n_ref > 0 && tid >= o.n_ref) { // :531
      out.status = SBGPU_BAM_BAD_REF;
      return;
   }
   // :536-599 the CIGAR.  The reference first walks every operation (length, kind, intron limits), then tests the kept
   // list for misplaced insertions / deletions: one walk here, the first complaint of the first kind wins over the second.
   const uint8_t *cig = data + l_qname;
   int64_t rlen = 0, eff = 0;
   int32_t qlen = 0, blocks = 0;
   int kept = 0, prev = -1;       // kept operations so far; the kind of the last one
   const uint32_t pos = (uint32_t)pos0 + 1u;
#pragma unroll
   for (int k = 0; k < kBamInlineBlocks; ++k) out.bl[k] = out.br[k] = 0u;
   bool indel_open = false, indel_bad = false;
   int walk = SBGPU_BAM_OK;
   for (int i = 0; i < n_cigar; ++i) {
      const uint32_t w = (uint32_t)rd.peek8(cig + 4 * i, end);
      const int32_t length = (int32_t)(w >> 4);
      if (length <= 0) {
         walk = SBGPU_BAM_ZERO_OP;
         break;
      }
      const int op = (int)(w & 0xfu);
      if (op == 5 || op == 6) continue; // H, P: not kept
      if (op > 6) {
         walk = SBGPU_BAM_OP;
         break;
      }
      if (indel_open) { // the operation behind an I / D must be an M
         indel_bad |= op != 0;
         indel_open = false;
      }
      if (op == 0) {
         // a block [offset, offset + length - 1] (readhit_2_genomicFeats); the first few are kept here (static indices: registers)
#pragma unroll
         for (int k = 0; k < kBamInlineBlocks; ++k)
            if (blocks == k) out.bl[k] = pos + (uint32_t)rlen, out.br[k] = pos + (uint32_t)rlen + (uint32_t)length - 1u;
         rlen += length, eff += length, qlen += length, ++blocks;
      } else if (op == 1 || op == 2) {
         indel_bad |= (kept - 1 <= 0) | (prev != 0); // `i-1 <= 0`: among the first two kept operations (:594)
         indel_open = true;
         if (op == 1) {
            qlen += length;
         } else { // a deletion extends the block in front of it
#pragma unroll
            for (int k = 0; k < kBamInlineBlocks; ++k)
               if (blocks == k + 1) out.br[k] += (uint32_t)length;
            rlen += length;
         }
      } else if (op == 4) {
         qlen += length;
      } else { // N
         rlen += length;
         if (length > o.max_intron) walk = SBGPU_BAM_INTRON_LONG;
         else if (length < o.min_intron) walk = SBGPU_BAM_INTRON_SHORT;
         if (walk != SBGPU_BAM_OK) break;
      }
      prev = op;
      ++kept;
   }
   if (walk != SBGPU_BAM_OK) {
      out.status = (uint8_t)walk;
      return;
   }
   if (indel_bad || indel_open) { // (open at the end: no operation behind it, :594)
      out.status = SBGPU_BAM_INDEL;
      return;
   }
   if (eff <= 1) { // :601
      out.status = SBGPU_BAM_SHORT;
      return;
   }
   out.paired = (flag & 0x1u) ? 1 : 0; // :605-607 (SINGLE_END_EXP = false), whatever the tests below say
   // the tags: XS (:619-634), NM (:653-656), NH (:658-661); the first of each, found by ONE scan of the list
   int sd = 0;
   {
      const uint8_t *s = data + l_qname + 4 * (int64_t)n_cigar + l_qseq + (l_qseq + 1) / 2;
      bool got_xs = false, got_nm = false, got_nh = false;
      while (s + 1 < end && !(got_xs && got_nm && got_nh)) {
         // a tag's two name bytes, its type byte and up to five bytes of its value: one load
         const uint64_t w = rd.peek8(s, end);
         const int x = (int)(((w & 0xffu) << 8) | ((w >> 8) & 0xffu));
         s += 2;
         if (s >= end) break;
         const uint64_t tv = w >> 16; // the type byte, then the value
         if (x == (('X' << 8) | 'S') && !got_xs) {
            got_xs = true;
            if (s + 1 < end && (tv & 0xffu) == 'A') {
               const uint32_t v = (uint32_t)(tv >> 8) & 0xffu;
               sd = v == '+' ? 1 : (v == '-' ? 2 : 0);
            }
         } else if (x == (('N' << 8) | 'M') && !got_nm) {
            got_nm = true;
            out.nm = (int32_t)(uint8_t)aux_int_of(tv, s, end); // through an unsigned char (:617)
         } else if (x == (('N' << 8) | 'H') && !got_nh) {
            got_nh = true;
            out.nh = aux_int_of(tv, s, end);
         }
         int type = (int)(tv & 0xffu);
         ++s;
         if (type >= 'a' && type <= 'z') type -= 32; // toupper (__skip_tag)
         if (type == 'Z' || type == 'H') {
            // to its NUL (or the record's end), eight bytes per load
            bool found = false;
            while (s < end && !found) {
               const uint64_t z = rd.peek8(s, end);
               int adv = 8;
               for (int i = 7; i >= 0; --i)
                  if (((z >> (8 * i)) & 0xffu) == 0u && s + i < end) adv = i, found = true;
               s += adv;
            }
            if (s > end) s = end;
            ++s;
         } else if (type == 'B') {
            if (s + 5 > end) break;
            const int64_t count = (int32_t)(uint32_t)((tv >> 16) & 0xffffffffull); // (tv: type, subtype, the count's four bytes)
            if (count < 0) break;
            s += 5 + (int64_t)bam_aux_size((int)((tv >> 8) & 0xffu)) * count;
         } else {
            s += bam_aux_size(type);
         }
      }
   }
   const bool rev = (flag & 0x10u) != 0, fr = o.library == 1, rf = o.library == 2;
   if (sd == 0 && (fr || rf)) { // :636-651
      const bool toward = (rf && rev) || (fr && !rev);
      sd = (flag & 0x40u) ? (toward ? 1 : 2) : (toward ? 2 : 1);
   }
   if (o.unique_only && (out.nh > 1 || (flag & 0x100u))) { // :670
      out.status = SBGPU_BAM_MULTI;
      return;
   }
   out.status = SBGPU_BAM_OK;
   out.ref = tid;
   out.left = pos;
   out.right = pos + (uint32_t)rlen - 1u;
   out.partner_pos = (uint32_t)mpos0 + 1u;
   out.flags = (uint8_t)((rev ? SBGPU_READ_REVERSE : 0u) | (mtid != tid ? SBGPU_READ_PARTNER_ELSEWHERE : 0u) | ((uint32_t)sd << 2));
   out.read_len = qlen;
   out.n_blocks = blocks;
}
SB_HD void bam_decode_record(const uint8_t *rec, int64_t avail, const sbgpu_bam_opts_t &o, BamRead &out)
{
   ByteLoads rd;
   bam_decode_record(rd, rec, avail, o, out);
}

// The aligned blocks of an ACCEPTED record (readhit_2_genomicFeats' MATCH features, src/contig.cpp:12-53): every M
// starts a block at the running offset, a D extends the block in front of it, N and D move the offset, I / S / H / P do not.
SB_HD void bam_record_blocks(const uint8_t *rec, uint32_t *block_left, uint32_t *block_right)
{
   const uint8_t *core = rec + 4, *data = rec + 36;
   const int l_qname = (int)(bam_u32(core + 8) & 0xffu), n_cigar = (int)(bam_u32(core + 12) & 0xffffu);
   const uint8_t *cig = data + l_qname;
   uint32_t offset = (uint32_t)bam_i32(core + 4) + 1u;
   int b = -1;
   for (int i = 0; i < n_cigar; ++i) {
      const uint32_t w = bam_u32(cig + 4 * i), len = w >> 4;
      const int op = (int)(w & 0xfu);
      if (op == 0) {
         ++b;
         block_left[b] = offset;
         block_right[b] = offset + len - 1u;
         offset += len;
      } else if (op == 2) {
         block_right[b] += len;
         offset += len;
      } else if (op == 3) {
         offset += len;
      }
   }
}

#ifdef __HIPCC__
struct BamScanArgs {
   const uint8_t *bytes;
   int64_t n_bytes;
   const int64_t *rec_off; // [n + 1]
   int64_t n;
   sbgpu_bam_opts_t opts;
   // per record
   uint8_t *status;
   int32_t *n_blocks; // 0 for a refused record
   uint8_t *accepted; // 0: refused; 1: accepted; 2: accepted, and its ONE block is [left, right] (an unspliced read: nearly every
                      // record) -- its inline blocks are then neither written here nor read by the fill
   uint64_t *read_id;
   int32_t *ref, *nh, *nm, *read_len;
   uint32_t *left, *right, *partner_pos, *sam_flag;
   uint8_t *flags;
   uint32_t *inline_blocks; // [n][2 * kBamInlineBlocks]: left ends, right ends of the record's first blocks
   unsigned long long *counts; // [16]: 0-10 by status, 11: records with the paired flag that got as far as :605
   int32_t *tile_reads, *tile_blocks; // per tile of 64 records (a wave's pass): accepted records, their blocks
};

// One wave per workgroup, 64 records per pass.  A lane that walks its record straight from global memory touches two or
// three cache lines of its own per record, byte by byte: with every lane of every wave doing so the vector L1 holds nothing
// and every byte load becomes an L2 request -- 10 GB of them for a 0.9 GB stream (1.1 ms per 4 * 10^6 records, 0.8 TB/s).
// The 64 records of a wave are CONTIGUOUS in the stream, so the wave copies their whole range into LDS with 16-byte loads,
// eight in flight per lane (every cache line of the stream is read once, by one instruction), and the lanes walk their
// records there through ByteWindow.  The buffer's size comes with the launch (dynamic LDS: ~1.1 x 64 average records, so
// that several workgroups fit a CU); a range beyond it (long reads) is walked in global memory as before.
__global__ __launch_bounds__(64) void bam_scan_kernel(BamScanArgs a, int stage_bytes)
{
   extern __shared__ __attribute__((aligned(16))) uint8_t stage[];
   __shared__ unsigned int cnt[16];
   const int lane = threadIdx.x;
   if (lane < 16) cnt[lane] = 0;
   __syncthreads();
   const int64_t n_tiles = (a.n + 63) >> 6;
   for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
      const int64_t r0 = tile << 6, r1 = r0 + 64 < a.n ? r0 + 64 : a.n;
      const int64_t s0 = a.rec_off[r0], s1 = a.rec_off[r1];
      const uint8_t *g = a.bytes + s0;
      const int shift = (int)((uintptr_t)g & 15u);
      // (offsets that do not ascend inside the stream: nothing is staged, and the lanes below refuse their records)
      const bool staged = s0 >= 0 && s1 >= s0 && s1 <= a.n_bytes && s1 - s0 <= (int64_t)stage_bytes - 16;
      if (staged) {
         const int len = (int)(s1 - s0);
         const int head = min((16 - shift) & 15, len), nbody = (len - head) >> 4, tail = len - head - (nbody << 4);
         if (lane < head) stage[shift + lane] = g[lane];
         const uint4 *src = reinterpret_cast<const uint4 *>(g + head);
         uint4 *dst = reinterpret_cast<uint4 *>(stage + shift + head);
         for (int c0 = 0; c0 < nbody; c0 += 8 * 64) {
            uint4 v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
               const int c = c0 + k * 64 + lane;
               v[k] = c < nbody ? src[c] : uint4{0, 0, 0, 0};
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
               const int c = c0 + k * 64 + lane;
               if (c < nbody) dst[c] = v[k];
            }
         }
         if (lane < tail) stage[shift + head + (nbody << 4) + lane] = g[head + (nbody << 4) + lane];
      }
      __syncthreads();
      const int64_t r = r0 + lane;
      bool my_ok = false;
      int my_nb = 0;
      if (r < r1) {
         const int64_t o0 = a.rec_off[r], o1 = a.rec_off[r + 1];
         const bool inside = o0 >= s0 && o1 >= o0 && o1 <= s1 && o0 >= 0 && o1 <= a.n_bytes; // (the caller's offsets are device data)
         BamRead x;
         if (!inside) {
            bam_decode_record(a.bytes, 0, a.opts, x); // TRUNCATED, nothing read
         } else if (staged) {
            ByteWindow rd;
            bam_decode_record(rd, stage + shift + (o0 - s0), o1 - o0, a.opts, x);
         } else {
            bam_decode_record(a.bytes + o0, o1 - o0, a.opts, x);
         }
         a.status[r] = x.status;
         const bool one_plain = x.status == SBGPU_BAM_OK && x.n_blocks == 1 && x.bl[0] == x.left && x.br[0] == x.right;
         a.accepted[r] = x.status == SBGPU_BAM_OK ? (one_plain ? 2 : 1) : 0;
         a.n_blocks[r] = x.status == SBGPU_BAM_OK ? x.n_blocks : 0;
         a.read_id[r] = x.read_id;
         a.ref[r] = x.ref, a.nh[r] = x.nh, a.nm[r] = x.nm, a.read_len[r] = x.read_len;
         a.left[r] = x.left, a.right[r] = x.right, a.partner_pos[r] = x.partner_pos, a.sam_flag[r] = x.sam_flag;
         a.flags[r] = x.flags;
         if (x.status == SBGPU_BAM_OK && !one_plain) {
            uint32_t *ib = a.inline_blocks + r * (2 * kBamInlineBlocks);
#pragma unroll
            for (int k = 0; k < kBamInlineBlocks; ++k) ib[k] = x.bl[k], ib[kBamInlineBlocks + k] = x.br[k];
         }
         atomicAdd(&cnt[x.status < 11 ? x.status : 10], 1u);
         if (x.paired) atomicAdd(&cnt[11], 1u);
         my_ok = x.status == SBGPU_BAM_OK;
         my_nb = my_ok ? x.n_blocks : 0;
      }
      // the tile's counts: the device-wide scans run over TILES (1/64 of the records), the fill finds a record's place from
      // its tile's and its neighbours' inside the wave
      const unsigned long long okm = __ballot(my_ok);
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) my_nb += __shfl_xor(my_nb, o);
      if (lane == 0) a.tile_reads[tile] = (int32_t)__popcll(okm), a.tile_blocks[tile] = my_nb;
      __syncthreads(); // (the next pass overwrites the buffer)
   }
   if (lane < 16 && cnt[lane]) atomicAdd(&a.counts[lane], (unsigned long long)cnt[lane]);
}

struct BamFillArgs {
   const uint8_t *bytes;
   const int64_t *rec_off;
   int64_t n;
   const uint8_t *accepted;
   const int64_t *tile_read_at;  // [tiles + 1] exclusive scan of the tiles' accepted records (a tile: 64 records)
   const int64_t *tile_block_at; // [tiles + 1] ... of their blocks
   // per record (the scan's)
   const uint64_t *read_id;
   const int32_t *ref, *nh, *nm, *read_len;
   const uint32_t *left, *right, *partner_pos, *sam_flag;
   const uint8_t *flags;
   const int32_t *n_blocks;
   const uint32_t *inline_blocks;
   // per accepted record
   int64_t *o_record;
   uint64_t *o_read_id;
   int32_t *o_ref, *o_nh, *o_nm, *o_read_len;
   uint32_t *o_left, *o_right, *o_partner_pos, *o_sam_flag;
   uint8_t *o_flags;
   int64_t *o_block_off; // [accepted + 1]
   uint32_t *o_block_left, *o_block_right;
};

__global__ __launch_bounds__(256) void bam_fill_kernel(BamFillArgs a)
{
   const int lane = (int)(threadIdx.x & 63u);
   const int64_t n_tiles = (a.n + 63) >> 6, wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
   for (int64_t tile = wave; tile < n_tiles; tile += n_waves) { // a wave per tile of 64 records
      const int64_t r = (tile << 6) + lane;
      const int acc_kind = r < a.n ? (int)a.accepted[r] : 0;
      const bool acc = acc_kind != 0;
      const int nb = acc ? a.n_blocks[r] : 0;
      // the record's place: its tile's (the scans over the tiles) and the accepted records / blocks of the lanes in front
      const unsigned long long okm = __ballot(acc);
      int incl = nb;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
         const int t = __shfl_up(incl, o);
         if (lane >= o) incl += t;
      }
      if (r == a.n - 1) a.o_block_off[a.tile_read_at[n_tiles]] = a.tile_block_at[n_tiles];
      if (!acc) continue;
      const int64_t k = a.tile_read_at[tile] + __popcll(okm & ((1ull << lane) - 1ull)), b = a.tile_block_at[tile] + (incl - nb);
      a.o_record[k] = r;
      a.o_read_id[k] = a.read_id[r];
      a.o_ref[k] = a.ref[r], a.o_nh[k] = a.nh[r], a.o_nm[k] = a.nm[r], a.o_read_len[k] = a.read_len[r];
      const uint32_t left = a.left[r], right = a.right[r];
      a.o_left[k] = left, a.o_right[k] = right, a.o_partner_pos[k] = a.partner_pos[r], a.o_sam_flag[k] = a.sam_flag[r];
      a.o_flags[k] = a.flags[r];
      a.o_block_off[k] = b;
      if (acc_kind == 2) { // one block, the read's own interval
         a.o_block_left[b] = left, a.o_block_right[b] = right;
      } else if (nb <= kBamInlineBlocks) { // (nearly every record: nothing of the stream is read again)
         const uint32_t *ib = a.inline_blocks + r * (2 * kBamInlineBlocks);
         for (int q = 0; q < nb; ++q) a.o_block_left[b + q] = ib[q], a.o_block_right[b + q] = ib[kBamInlineBlocks + q];
      } else {
         bam_record_blocks(a.bytes + a.rec_off[r], a.o_block_left + b, a.o_block_right + b);
      }
   }
}

// ------------------------------------------------------------------ one pass (round 6)
// The two kernels above move every accepted record's scalars twice (the scan writes them per RECORD, the fill reads them
// and writes them per accepted READ: 62 GB beside the stream's 68 GB at 3.9 * 10^8 records).  Here a record's place among
// the accepted ones is known while its scalars are still in registers: a workgroup of four waves decodes a TILE of 1 024
// consecutive records (every wave 256 of them, in four passes of 64 through its own part of the staging buffer), holds the
// results (12 registers per record), and gets the tile's base -- accepted records and aligned blocks in front of it -- by a
// DECOUPLED LOOK-BACK over the tiles before it: a tile publishes its own counts as soon as it has them, then reads the 64
// tiles in front of it at once (one per lane), adds counts up to the nearest tile that already knows its inclusive prefix,
// and publishes its own.  One 64-bit word per tile: flag (2 bits: nothing yet / own counts / inclusive prefix), accepted
// records (30 bits), blocks (32 bits) -- written and read with one relaxed device-scope atomic each, nothing else crosses
// between workgroups.  Tiles are dealt by a ticket counter, so a tile only ever waits for tiles whose workgroups are
// already running; a wait is bounded all the same (kOneMaxPolls), and a look-back that runs out, like a block count that
// leaves 32 bits or the caller's capacity, is counted in counts[12]: the caller then decodes with the two kernels above.
// A tile of 64 records (a wave by itself) was measured in round 5: 6 * 10^6 tiles make the chain of look-backs the whole
// kernel, 17 x slower; 1 024 records per tile: 3.8 * 10^5 tiles, the chain runs beside the decoding.
struct BamOneArgs {
   const uint8_t *bytes;
   int64_t n_bytes;
   const int64_t *rec_off; // [n + 1]
   int64_t n;
   sbgpu_bam_opts_t opts;
   uint8_t *status; // per record
   // per accepted record (room for n of them)
   int64_t *o_record;
   uint64_t *o_read_id;
   int32_t *o_ref, *o_nh, *o_nm, *o_read_len;
   uint32_t *o_left, *o_right, *o_partner_pos, *o_sam_flag;
   uint8_t *o_flags;
   int64_t *o_block_off; // [accepted + 1]
   uint32_t *o_block_left, *o_block_right;
   int64_t block_cap;          // room in o_block_*
   unsigned long long *counts; // [16]: 0-10 by status, 11: paired records, 12: failures (see above), 13: accepted records, 14: their blocks
   unsigned long long *tile_state; // [tiles], zeroed
   unsigned int *ticket;           // zeroed
};

constexpr int kOneWaves = 4, kOnePasses = 4, kOneTile = 64 * kOneWaves * kOnePasses;
constexpr unsigned long long kOneFlagMask = 3ull << 62, kOneCounts = 1ull << 62, kOnePrefix = 2ull << 62;
constexpr int kOneMaxPolls = 1 << 20;
constexpr int64_t kOneMaxRecords = ((int64_t)1 << 30) - 1; // (30 bits of accepted records in a tile's word)

__device__ __forceinline__ unsigned long long one_word(unsigned long long flag, uint32_t reads, uint32_t blocks)
{
   return flag | ((unsigned long long)(reads & 0x3fffffffu) << 32) | blocks;
}
__device__ __forceinline__ unsigned long long one_wave_sum(unsigned long long v)
{
#pragma unroll
   for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
   return v;
}

struct BamHeld { // what stays of a decoded record until its place is known
   uint64_t read_id;
   int32_t ref, nh, nm, read_len;
   uint32_t left, right, partner_pos, sam_flag;
   uint32_t mid0, mid1; // two blocks: [left, mid0] [mid1, right]
   uint32_t meta;       // bits 0-7: flags; 8-9: 0 refused, 1 accepted (the CIGAR is walked again for its blocks), 2 one block = [left, right],
                        // 3 two blocks as above; 16-31: blocks
};

__global__ __launch_bounds__(64 * kOneWaves) void bam_onepass_kernel(BamOneArgs a, int stage_bytes)
{
   extern __shared__ __attribute__((aligned(16))) uint8_t stage_all[];
   __shared__ unsigned int cnt[16];
   __shared__ unsigned int s_tile, s_wr[kOneWaves], s_wb[kOneWaves], s_base_r;
   __shared__ unsigned long long s_base_b;
   const int tid = (int)threadIdx.x, lane = tid & 63, wv = tid >> 6;
   uint8_t *stage = stage_all + (size_t)wv * (size_t)stage_bytes;
   if (tid < 16) cnt[tid] = 0;
   const int64_t n_tiles = (a.n + kOneTile - 1) / kOneTile;
   const unsigned long long lanes_before = (1ull << lane) - 1ull;
   for (;;) {
      __syncthreads(); // (the tile before: everyone has read s_tile and the bases)
      if (tid == 0) s_tile = atomicAdd(a.ticket, 1u);
      __syncthreads();
      const int64_t tile = (int64_t)s_tile;
      if (tile >= n_tiles) break;
      BamHeld h[kOnePasses];
      unsigned long long okm[kOnePasses];
#pragma unroll
      for (int q = 0; q < kOnePasses; ++q) h[q] = BamHeld{}, okm[q] = 0ull;
      // (ONE copy of the decoder: the passes are a loop, and a pass' results go to their registers by selects -- unrolled, the
      // four copies of the three decoder forms cost 690 spilled scalar registers and a third of the occupancy)
#pragma unroll 1
      for (int j = 0; j < kOnePasses; ++j) {
         const int64_t r0 = tile * kOneTile + (int64_t)(wv * kOnePasses + j) * 64, r1 = r0 + 64 < a.n ? r0 + 64 : a.n;
         if (r0 >= a.n) break; // (wave-uniform)
         const int64_t s0 = a.rec_off[r0], s1 = a.rec_off[r1];
         const uint8_t *g = a.bytes + s0;
         const int shift = (int)((uintptr_t)g & 15u);
         // (offsets that do not ascend inside the stream: nothing is staged, and the lanes below refuse their records)
         const bool staged = s0 >= 0 && s1 >= s0 && s1 <= a.n_bytes && s1 - s0 <= (int64_t)stage_bytes - 16;
         if (staged) {
            const int len = (int)(s1 - s0);
            const int head = min((16 - shift) & 15, len), nbody = (len - head) >> 4, tail = len - head - (nbody << 4);
            if (lane < head) stage[shift + lane] = g[lane];
            const uint4 *src = reinterpret_cast<const uint4 *>(g + head);
            uint4 *dst = reinterpret_cast<uint4 *>(stage + shift + head);
            for (int c0 = 0; c0 < nbody; c0 += 8 * 64) {
               uint4 v[8];
#pragma unroll
               for (int k = 0; k < 8; ++k) {
                  const int c = c0 + k * 64 + lane;
                  v[k] = c < nbody ? src[c] : uint4{0, 0, 0, 0};
               }
#pragma unroll
               for (int k = 0; k < 8; ++k) {
                  const int c = c0 + k * 64 + lane;
                  if (c < nbody) dst[c] = v[k];
               }
            }
            if (lane < tail) stage[shift + head + (nbody << 4) + lane] = g[head + (nbody << 4) + lane];
         }
         // (the buffer is the WAVE's own: its LDS instructions execute in program order, so the lanes' reads below see the
         // writes above and the next pass' writes come after them -- the fence is for the compiler)
         __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
         __builtin_amdgcn_wave_barrier();
         const int64_t r = r0 + lane;
         bool my_ok = false;
         BamHeld t = {};
         if (r < r1) {
            const int64_t o0 = a.rec_off[r], o1 = a.rec_off[r + 1];
            const bool inside = o0 >= s0 && o1 >= o0 && o1 <= s1 && o0 >= 0 && o1 <= a.n_bytes; // (the caller's offsets are device data)
            BamRead x;
            if (!inside) {
               bam_decode_record(a.bytes, 0, a.opts, x); // TRUNCATED, nothing read
            } else if (staged) {
               ByteWindow rd;
               bam_decode_record(rd, stage + shift + (o0 - s0), o1 - o0, a.opts, x);
            } else {
               bam_decode_record(a.bytes + o0, o1 - o0, a.opts, x);
            }
            a.status[r] = x.status;
            my_ok = x.status == SBGPU_BAM_OK;
            const bool from_left = x.bl[0] == x.left;
            const bool one = my_ok && x.n_blocks == 1 && from_left && x.br[0] == x.right;
            const bool two = my_ok && x.n_blocks == 2 && from_left && x.br[1] == x.right;
            t.read_id = x.read_id;
            t.ref = x.ref, t.nh = x.nh, t.nm = x.nm, t.read_len = x.read_len;
            t.left = x.left, t.right = x.right, t.partner_pos = x.partner_pos, t.sam_flag = x.sam_flag;
            t.mid0 = x.br[0], t.mid1 = x.bl[1];
            t.meta = my_ok ? ((uint32_t)x.flags | ((one ? 2u : (two ? 3u : 1u)) << 8) | ((uint32_t)x.n_blocks << 16)) : 0u;
            atomicAdd(&cnt[x.status < 11 ? x.status : 10], 1u);
            if (x.paired) atomicAdd(&cnt[11], 1u);
         }
         const unsigned long long okj = __ballot(my_ok);
#pragma unroll
         for (int q = 0; q < kOnePasses; ++q) {
            const bool here = q == j; // (uniform)
            h[q].read_id = here ? t.read_id : h[q].read_id;
            h[q].ref = here ? t.ref : h[q].ref, h[q].nh = here ? t.nh : h[q].nh, h[q].nm = here ? t.nm : h[q].nm;
            h[q].read_len = here ? t.read_len : h[q].read_len, h[q].left = here ? t.left : h[q].left, h[q].right = here ? t.right : h[q].right;
            h[q].partner_pos = here ? t.partner_pos : h[q].partner_pos, h[q].sam_flag = here ? t.sam_flag : h[q].sam_flag;
            h[q].mid0 = here ? t.mid0 : h[q].mid0, h[q].mid1 = here ? t.mid1 : h[q].mid1, h[q].meta = here ? t.meta : h[q].meta;
            okm[q] = here ? okj : okm[q];
         }
         __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
         __builtin_amdgcn_wave_barrier();
      }
      // the wave's counts, and every record's blocks in front of it inside the wave
      uint32_t wave_reads = 0, wave_blocks = 0, excl[kOnePasses], pass_blocks[kOnePasses];
#pragma unroll
      for (int j = 0; j < kOnePasses; ++j) {
         const uint32_t nb = h[j].meta >> 16;
         uint32_t incl = nb;
#pragma unroll
         for (int o = 1; o < 64; o <<= 1) {
            const uint32_t t = __shfl_up(incl, o);
            if (lane >= o) incl += t;
         }
         excl[j] = incl - nb;
         pass_blocks[j] = __shfl(incl, 63);
         wave_reads += (uint32_t)__popcll(okm[j]);
         wave_blocks += pass_blocks[j];
      }
      if (lane == 0) s_wr[wv] = wave_reads, s_wb[wv] = wave_blocks;
      __syncthreads();
      if (wv == 0) {
         uint32_t R = 0, Bk = 0;
#pragma unroll
         for (int w = 0; w < kOneWaves; ++w) R += s_wr[w], Bk += s_wb[w];
         unsigned long long sum_r = 0, sum_b = 0;
         bool failed = false;
         if (tile > 0) {
            if (lane == 0) __hip_atomic_store(a.tile_state + tile, one_word(kOneCounts, R, Bk), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int64_t hi = tile;; hi -= 64) {
               const int64_t idx = hi - 1 - lane; // lane 0: the tile just in front
               unsigned long long v = idx >= 0 ? __hip_atomic_load(a.tile_state + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : kOnePrefix;
               // (wanted: every tile up to the nearest one that knows its prefix -- what lies beyond that one is not waited for)
               unsigned long long pm = 0ull;
               for (int polls = 0;; ++polls) {
                  pm = __ballot((v & kOneFlagMask) == kOnePrefix);
                  const unsigned long long wanted = pm ? ((pm & (0ull - pm)) - 1ull) : ~0ull; // the lanes in front of the nearest prefix
                  if ((__ballot((v & kOneFlagMask) == 0ull) & wanted) == 0ull) break;
                  if (polls >= kOneMaxPolls) {
                     failed = true;
                     break;
                  }
                  __builtin_amdgcn_s_sleep(1);
                  if ((v & kOneFlagMask) == 0ull) v = __hip_atomic_load(a.tile_state + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
               }
               if (failed) break;
               const int first = pm ? (int)__ffsll((long long)pm) - 1 : 64; // the nearest tile that knows its prefix
               const bool in = lane <= first;
               sum_r += one_wave_sum(in ? ((v >> 32) & 0x3fffffffull) : 0ull);
               sum_b += one_wave_sum(in ? (v & 0xffffffffull) : 0ull);
               if (pm) break;
            }
         }
         const unsigned long long incl_r = sum_r + R, incl_b = sum_b + Bk;
         failed = failed || incl_b > 0xffffffffull || incl_r > 0x3fffffffull || (int64_t)incl_b > a.block_cap;
         if (lane == 0) {
            __hip_atomic_store(a.tile_state + tile, one_word(kOnePrefix, (uint32_t)incl_r, (uint32_t)incl_b), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_base_r = (uint32_t)sum_r;
            s_base_b = sum_b;
            if (failed) atomicAdd(&a.counts[12], 1ull);
            if (tile == n_tiles - 1) {
               a.counts[13] = incl_r, a.counts[14] = incl_b;
               if ((int64_t)incl_r <= a.n) a.o_block_off[incl_r] = (int64_t)incl_b;
            }
         }
      }
      __syncthreads();
      int64_t run_r = (int64_t)s_base_r, run_b = (int64_t)s_base_b;
#pragma unroll
      for (int w = 0; w < kOneWaves; ++w)
         if (w < wv) run_r += s_wr[w], run_b += s_wb[w];
#pragma unroll
      for (int j = 0; j < kOnePasses; ++j) {
         const uint32_t kind = (h[j].meta >> 8) & 3u, nb = h[j].meta >> 16;
         const int64_t k = run_r + __popcll(okm[j] & lanes_before), b = run_b + excl[j];
         if (kind != 0u && k < a.n) {
            const int64_t r = tile * kOneTile + (int64_t)(wv * kOnePasses + j) * 64 + lane;
            a.o_record[k] = r;
            a.o_read_id[k] = h[j].read_id;
            a.o_ref[k] = h[j].ref, a.o_nh[k] = h[j].nh, a.o_nm[k] = h[j].nm, a.o_read_len[k] = h[j].read_len;
            a.o_left[k] = h[j].left, a.o_right[k] = h[j].right, a.o_partner_pos[k] = h[j].partner_pos, a.o_sam_flag[k] = h[j].sam_flag;
            a.o_flags[k] = (uint8_t)(h[j].meta & 0xffu);
            a.o_block_off[k] = b;
            if (b + (int64_t)nb <= a.block_cap) {
               if (kind == 2u) {
                  a.o_block_left[b] = h[j].left, a.o_block_right[b] = h[j].right;
               } else if (kind == 3u) {
                  a.o_block_left[b] = h[j].left, a.o_block_right[b] = h[j].mid0;
                  a.o_block_left[b + 1] = h[j].mid1, a.o_block_right[b + 1] = h[j].right;
               } else {
                  bam_record_blocks(a.bytes + a.rec_off[r], a.o_block_left + b, a.o_block_right + b);
               }
            }
         }
         run_r += __popcll(okm[j]);
         run_b += pass_blocks[j];
      }
   }
   __syncthreads();
   if (tid < 16 && cnt[tid]) atomicAdd(&a.counts[tid], (unsigned long long)cnt[tid]);
}
#endif

} // namespace sb
