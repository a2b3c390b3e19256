// examples/bam_reads.cpp -- the read stream of a BAM file from C++14 over the C ABI: what the reference's
// Sample::next_valid_alignment loop (src/alignments.cpp:976-1009: BAMHitFactory::nextRecord + getHitFromBuf,
// src/read.cpp:455-715) hands to its clusters, for the whole file at once.
//
//   g++ -std=c++14 -Iinclude examples/bam_reads.cpp -Lstrawberry_amd/lib -lsbgpu -lz -Wl,-rpath,$PWD/strawberry_amd/lib
//   ./a.out in.bam [-j min_intron] [-J max_intron] [--allow-multimapped-hits] [--fr | --rf] > reads.tsv
//
// The program inflates the BGZF blocks with zlib (gzread walks the chain of gzip members), skips the BAM header, lets
// sbgpu_bam_index_host find the records and sbgpu_bam_decode_host decide about them (sbgpu_bam_decode_device takes the same
// bytes in device memory and leaves the arrays there for sbgpu_assign_reads_device / sbgpu_pair_mates_device), and prints
// one line per accepted record: record index, read id, reference name, first and last aligned base, strand, mate position,
// flags, NH, NM, read length, aligned blocks.  Counts by reason of refusal go to stderr.
#include <zlib.h>

#include <cinttypes>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "sbgpu.h"

static int32_t le32(const uint8_t *p) { return (int32_t)((uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24)); }

int main(int argc, char **argv)
{
   if (argc < 2) {
      std::fprintf(stderr, "usage: %s in.bam [-j min_intron] [-J max_intron] [--allow-multimapped-hits] [--fr | --rf]\n", argv[0]);
      return 2;
   }
   sbgpu_bam_opts_t opts = {20, 300000, 1, 0, 0}; // the reference's defaults (src/common.cpp:19-21,67-69)
   for (int i = 2; i < argc; ++i) {
      const std::string a = argv[i];
      if (a == "-j" && i + 1 < argc) opts.min_intron = std::atoi(argv[++i]);
      else if (a == "-J" && i + 1 < argc) opts.max_intron = std::atoi(argv[++i]);
      else if (a == "--allow-multimapped-hits") opts.unique_only = 0;
      else if (a == "--fr") opts.library = 1;
      else if (a == "--rf") opts.library = 2;
      else {
         std::fprintf(stderr, "unknown option %s\n", a.c_str());
         return 2;
      }
   }
   // BGZF: a chain of gzip members
   gzFile f = gzopen(argv[1], "rb");
   if (!f) {
      std::fprintf(stderr, "cannot open %s\n", argv[1]);
      return 1;
   }
   std::vector<uint8_t> raw;
   std::vector<uint8_t> chunk(1 << 20);
   for (;;) {
      const int n = gzread(f, chunk.data(), (unsigned)chunk.size());
      if (n < 0) {
         std::fprintf(stderr, "%s: inflate failed\n", argv[1]);
         return 1;
      }
      if (n == 0) break;
      raw.insert(raw.end(), chunk.begin(), chunk.begin() + n);
   }
   gzclose(f);
   // the header: magic, text, references
   if (raw.size() < 12 || std::memcmp(raw.data(), "BAM\1", 4) != 0) {
      std::fprintf(stderr, "%s: not a BAM file\n", argv[1]);
      return 1;
   }
   size_t p = 8 + (size_t)le32(raw.data() + 4);
   const int32_t n_ref = le32(raw.data() + p);
   p += 4;
   std::vector<std::string> ref_name;
   for (int32_t r = 0; r < n_ref; ++r) {
      const int32_t l_name = le32(raw.data() + p);
      ref_name.emplace_back((const char *)raw.data() + p + 4);
      p += 8 + (size_t)l_name;
   }
   opts.n_ref = n_ref;
   const uint8_t *records = raw.data() + p;
   const int64_t n_bytes = (int64_t)(raw.size() - p);
   std::vector<int64_t> rec_off((size_t)(n_bytes / 36 + 2));
   const int64_t n = sbgpu_bam_index_host(records, n_bytes, rec_off.data(), (int64_t)rec_off.size() - 1);
   if (n < 0) {
      std::fprintf(stderr, "%s\n", sbgpu_last_error());
      return 1;
   }
   sbgpu_bamreads_t *rd = nullptr;
   if (sbgpu_bam_decode_host(records, n_bytes, rec_off.data(), n, &opts, &rd) != SBGPU_OK) {
      std::fprintf(stderr, "%s\n", sbgpu_last_error());
      return 1;
   }
   int64_t info[16];
   sbgpu_bamreads_info(rd, info);
   static const char *why[] = {"accepted", "unmapped", "reference id not in the header", "CIGAR operation of length 0", "unsupported CIGAR operation",
                               "intron too long", "intron too short", "misplaced insertion / deletion", "at most one aligned base",
                               "multiple hits", "truncated record"};
   std::fprintf(stderr, "%" PRId64 " records, %s library\n", info[0], info[3] ? "paired-end" : "single-end");
   for (int s = 0; s <= SBGPU_BAM_TRUNCATED; ++s)
      if (info[5 + s]) std::fprintf(stderr, "  %-32s %" PRId64 "\n", why[s], info[5 + s]);
   const size_t m = (size_t)info[1], nb = (size_t)info[2];
   std::vector<int64_t> record(m), block_off(m + 1);
   std::vector<uint64_t> read_id(m);
   std::vector<int32_t> ref(m), nh(m), nm(m), read_len(m);
   std::vector<uint32_t> left(m), right(m), partner(m), block_left(nb), block_right(nb);
   std::vector<uint8_t> flags(m);
   if (sbgpu_bamreads_export(rd, nullptr, record.data(), read_id.data(), ref.data(), left.data(), right.data(), partner.data(), flags.data(),
                             nh.data(), nm.data(), read_len.data(), nullptr, block_off.data(), block_left.data(), block_right.data()) != SBGPU_OK) {
      std::fprintf(stderr, "%s\n", sbgpu_last_error());
      return 1;
   }
   sbgpu_bamreads_destroy(rd);
   for (size_t k = 0; k < m; ++k) {
      const int xs = (flags[k] >> 2) & 3;
      std::printf("%" PRId64 "\t%016" PRIx64 "\t%s\t%u\t%u\t%c\t%u\t%s%s\t%d\t%d\t%d\t", record[k], read_id[k], ref_name[(size_t)ref[k]].c_str(), left[k],
                  right[k], xs == 1 ? '+' : (xs == 2 ? '-' : '.'), partner[k], (flags[k] & SBGPU_READ_REVERSE) ? "r" : "f",
                  (flags[k] & SBGPU_READ_PARTNER_ELSEWHERE) ? "e" : "", nh[k], nm[k], read_len[k]);
      for (int64_t b = block_off[k]; b < block_off[k + 1]; ++b) std::printf("%s%u-%u", b > block_off[k] ? "," : "", block_left[(size_t)b], block_right[(size_t)b]);
      std::printf("\n");
   }
   return 0;
}
