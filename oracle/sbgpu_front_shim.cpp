// oracle/sbgpu_front_shim.cpp -- TEST INFRASTRUCTURE ONLY (nothing under strawberry_amd/ refers to it).
//
// The drop-in at its deepest: `make -C oracle ref` links oracle/_ref/strawberry_sbgpu_front from the reference's UNMODIFIED
// objects -- its main, option parsing, GTF reader, Contig::print2gtf -- with THREE functions replaced (weakened in a copy
// of alignments.o): Sample::inspect_read_len, Sample::preProcess and Sample::procSample
// (/root/reference/src/alignments.cpp:957-974, 1189-1232, 1736-1834), i.e. all three passes the reference makes over the
// BAM file.  The replacement reads the file ONCE (zlib on host threads: BGZF inflate stays on the host) and does the rest through libsbgpu:
//   sbgpu_bam_decode_device        BAMHitFactory::getHitFromBuf for every record            (read.cpp:480-715)
//   sbgpu_assign_reads_device      Sample::nextClusterRefDemand's pass                       (alignments.cpp:1145-1187)
//   sbgpu_pair_mates_device        HitCluster::addOpenHit / addHit                           (alignments.cpp:423-655)
//   sbgpu_collapse_pairs_device    HitCluster::collapseAndFilterHits + Contig(PairedHit)     (alignments.cpp:656-703)
//   sbgpu_quantify_host            LocusContext's constructor + estimate_abundances          (estimate.hpp:60-109, estimate.cpp:135-355)
//                                  -- with -f (the table needs every hit's bin and the weights on the host); WITHOUT -f the
//                                  program stays RESIDENT (round 6): the unique hits never leave the device, preProcess ends in
//   sbgpu_quantify_resident        pass 1 on the device (the empirical insert-size law when no -i was given), bins, weights, EM,
//                                  FPKM / Frac / keep / TPM; procSample only prints
// The clusters (which transcripts form a locus, in which order) come from the reference's own Sample::addRef2Cluster; the GTF
// is written by the reference's own Contig::print2gtf, the -f table by sbgpu_format_context_row.
//
// tests/test_reference_driver_gpu.py: its files equal the reference binary's byte for byte; tools/dropin_timing.py times it.
// Not covered (the program says so and stops): -b; assembly mode (no -g: the clusters would come from the reads).
#include "alignments.h" // the reference's: /root/reference/include/alignments.h:178-290 (Sample)
#include "estimate.hpp"

#include <zlib.h>

#include <chrono>
#include <climits>
#include <cstdlib>
#include <cstring>
#include <map>
#include <atomic>
#include <memory>
#include <thread>

#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>

#include "sbgpu_host.hpp"

namespace {
using clk = std::chrono::steady_clock;
double secs(clk::time_point a, clk::time_point b) { return std::chrono::duration<double>(b - a).count(); }

const sbgpu::Context &device_context()
{
   static const sbgpu::Context ctx(0); // throws (no CPU fallback) when there is no gfx950 device
   return ctx;
}
void hip_check(hipError_t e, const char *what)
{
   if (e != hipSuccess) {
      std::fprintf(stderr, "strawberry_sbgpu_front: %s: %s\n", what, hipGetErrorString(e));
      std::exit(1);
   }
}
struct Locus {
   RefID ref_id;
   uint left, right;
   std::vector<Contig> transcripts; // cluster->ref_mRNAs(): the reference's isoform order
};
// what the three replaced functions share
struct Front {
   std::vector<uint8_t> raw; // the inflated file
   size_t rec_begin = 0;
   int64_t n_bytes = 0, n_records = 0;
   void *d_bytes = nullptr, *d_rec_off = nullptr;
   sbgpu_bamreads_t *reads = nullptr; // device arrays
   int64_t info[16] = {};
   std::vector<Locus> loci;
   sbgpu::LocusBatch batch; // annotation + unique hits (host copies)
   double t_inflate = 0, t_decode = 0, t_front = 0;
   // resident mode (no -f): what sbgpu_quantify_resident left on the host
   bool resident = false;
   std::vector<double> r_theta, r_fpkm, r_frac, r_tpm;
   std::vector<int32_t> r_keep, r_status;
};
Front &front()
{
   static Front f;
   return f;
}
int32_t le32(const uint8_t *p) { return (int32_t)((uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24)); }
// BGZF: a chain of gzip members of at most 64 KB each, every one carrying its own compressed size (the BC subfield) and
// its inflated size (the member's last word): the members are found by a walk over the headers and inflated by host
// threads, each into its place.  (Inflate is host work and stays the caller's; this is the caller.)
void inflate_bgzf(const std::string &path, std::vector<uint8_t> &out)
{
   FILE *f = std::fopen(path.c_str(), "rb");
   if (!f) {
      std::fprintf(stderr, "strawberry_sbgpu_front: cannot open %s\n", path.c_str());
      std::exit(1);
   }
   std::fseek(f, 0, SEEK_END);
   const long sz = std::ftell(f);
   std::fseek(f, 0, SEEK_SET);
   std::vector<uint8_t> comp((size_t)sz);
   if (sz && std::fread(comp.data(), 1, (size_t)sz, f) != (size_t)sz) {
      std::fprintf(stderr, "strawberry_sbgpu_front: cannot read %s\n", path.c_str());
      std::exit(1);
   }
   std::fclose(f);
   struct Member {
      size_t in, in_len, out, out_len;
   };
   std::vector<Member> members;
   size_t p = 0, total = 0;
   while (p + 18 <= comp.size()) {
      const uint8_t *h = comp.data() + p;
      if (!(h[0] == 0x1f && h[1] == 0x8b && h[2] == 8 && (h[3] & 4))) break;
      const size_t xlen = (size_t)h[10] | ((size_t)h[11] << 8);
      size_t bsize = 0;
      for (size_t q = 12; q + 4 <= 12 + xlen;) {
         const size_t slen = (size_t)h[q + 2] | ((size_t)h[q + 3] << 8);
         if (h[q] == 'B' && h[q + 1] == 'C' && slen == 2) bsize = ((size_t)h[q + 4] | ((size_t)h[q + 5] << 8)) + 1;
         q += 4 + slen;
      }
      if (!bsize || p + bsize > comp.size()) break;
      const size_t isize = (size_t)(uint32_t)le32(h + bsize - 4);
      members.push_back({p + 12 + xlen, bsize - 12 - xlen - 8, total, isize});
      total += isize;
      p += bsize;
   }
   if (p != comp.size()) {
      std::fprintf(stderr, "strawberry_sbgpu_front: %s is not a BGZF file\n", path.c_str());
      std::exit(1);
   }
   out.resize(total);
   const unsigned hw = std::thread::hardware_concurrency();
   const size_t n_threads = std::max<size_t>(1, std::min<size_t>(hw ? hw : 4, 16));
   std::vector<std::thread> pool;
   std::atomic<size_t> next(0);
   std::atomic<int> failed(0);
   for (size_t t = 0; t < n_threads; ++t)
      pool.emplace_back([&] {
         for (;;) {
            const size_t first = next.fetch_add(64);
            if (first >= members.size()) break;
            for (size_t k = first; k < std::min(first + 64, members.size()); ++k) {
               const Member &m = members[k];
               if (!m.out_len) continue;
               z_stream zs;
               std::memset(&zs, 0, sizeof(zs));
               if (inflateInit2(&zs, -15) != Z_OK) {
                  failed = 1;
                  continue;
               }
               zs.next_in = comp.data() + m.in;
               zs.avail_in = (uInt)m.in_len;
               zs.next_out = out.data() + m.out;
               zs.avail_out = (uInt)m.out_len;
               if (inflate(&zs, Z_FINISH) != Z_STREAM_END || zs.avail_out != 0) failed = 1;
               inflateEnd(&zs);
            }
         }
      });
   for (std::thread &t : pool) t.join();
   if (failed) {
      std::fprintf(stderr, "strawberry_sbgpu_front: inflate failed\n");
      std::exit(1);
   }
}
} // namespace


// replaces /root/reference/src/alignments.cpp:957-974 -- and reads the file, for all three passes
void Sample::inspect_read_len()
{
   Front &F = front();
   if (!no_assembly) {
      std::fprintf(stderr, "strawberry_sbgpu_front: assembly mode is not covered by this driver (give -g and -r, or use strawberry_sbgpu_chain)\n");
      std::exit(2);
   }
   const clk::time_point t0 = clk::now();
   inflate_bgzf(sample_path(), F.raw);
   size_t p = 8 + (size_t)le32(F.raw.data() + 4);
   const int32_t n_ref = le32(F.raw.data() + p);
   p += 4;
   for (int32_t r = 0; r < n_ref; ++r) p += 8 + (size_t)le32(F.raw.data() + p);
   F.rec_begin = p;
   F.n_bytes = (int64_t)(F.raw.size() - p);
   const uint8_t *records = F.raw.data() + p;
   std::vector<int64_t> rec_off((size_t)(F.n_bytes / 36 + 2));
   F.n_records = sbgpu_bam_index_host(records, F.n_bytes, rec_off.data(), (int64_t)rec_off.size() - 1);
   sbgpu::check((int)(F.n_records < 0 ? SBGPU_ESHAPE : 0), "sbgpu_bam_index_host");
   const clk::time_point t1 = clk::now();
   F.t_inflate = secs(t0, t1);
   // every record through getHitFromBuf's rules, on the device; the option globals as the command line left them
   const sbgpu::Context &ctx = device_context();
   hip_check(hipMalloc(&F.d_bytes, (size_t)F.n_bytes + 16), "hipMalloc");
   hip_check(hipMalloc(&F.d_rec_off, (size_t)(F.n_records + 1) * 8), "hipMalloc");
   hip_check(hipMemcpy(F.d_bytes, records, (size_t)F.n_bytes, hipMemcpyHostToDevice), "hipMemcpy");
   hip_check(hipMemcpy(F.d_rec_off, rec_off.data(), (size_t)(F.n_records + 1) * 8, hipMemcpyHostToDevice), "hipMemcpy");
   sbgpu_bam_opts_t o = {kMinIntronLength, kMaxIntronLength, use_only_unique_hits ? 1 : 0, fr_strand ? 1 : (rf_strand ? 2 : 0), n_ref};
   sbgpu::check(sbgpu_bam_decode_device(ctx.get(), (const uint8_t *)F.d_bytes, F.n_bytes, (const int64_t *)F.d_rec_off, F.n_records, &o, nullptr,
                                        &F.reads),
                "sbgpu_bam_decode_device");
   sbgpu::check(sbgpu_bamreads_info(F.reads, F.info), "sbgpu_bamreads_info");
   if (F.info[3]) SINGLE_END_EXP = false; // read.cpp:605-607
   // the read-length histogram of the first kMaxReadNum4RL accepted records (alignments.cpp:957-974)
   std::vector<int32_t> read_len((size_t)F.info[1]);
   sbgpu::check(sbgpu_bamreads_export(F.reads, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                                      read_len.data(), nullptr, nullptr, nullptr, nullptr),
                "sbgpu_bamreads_export");
   const size_t lim = std::min<size_t>(read_len.size(), (size_t)std::max(0, kMaxReadNum4RL));
   for (size_t k = 0; k < lim; ++k) _hit_factory->_reads_table._read_len_abs[(uint)read_len[k]]++;
   F.t_decode = secs(t1, clk::now());
}

// replaces /root/reference/src/alignments.cpp:1189-1232: clusters, mate pairing, duplicate collapse, the mapped-read total and
// (without -i) the sample of the empirical insert-size distribution -- for the whole file at once
void Sample::preProcess(FILE *log)
{
   Front &F = front();
   if (BIAS_CORRECTION) {
      std::fprintf(stderr, "strawberry_sbgpu_front: -b is not covered by this driver (use strawberry_sbgpu_batched)\n");
      std::exit(2);
   }
   const clk::time_point t0 = clk::now();
   const sbgpu::Context &ctx = device_context();
   // the clusters: the reference's own grouping of the annotation's transcripts (addRef2Cluster, alignments.cpp:1012-1090)
   reset_refmRNAs();
   std::vector<int32_t> c_ref;
   std::vector<uint32_t> c_left, c_right;
   std::vector<uint8_t> c_strand;
   while (true) {
      HitCluster c;
      if (addRef2Cluster(c) == 0) break;
      F.loci.push_back({c.ref_id(), c.left(), c.right(), c.ref_mRNAs()});
      c_ref.push_back((int32_t)c.ref_id());
      c_left.push_back(c.left());
      c_right.push_back(c.right());
      c_strand.push_back(c.ref_strand() == Strand_t::StrandPlus ? 1 : (c.ref_strand() == Strand_t::StrandMinus ? 2 : 0));
   }
   const int64_t n_all = (int64_t)F.loci.size(), n_reads = F.info[1];
   // which cluster every record is offered to (nextClusterRefDemand's pass), flags in place
   sbgpu_reads_t reads;
   const int32_t *d_ref;
   const uint32_t *d_left, *d_right;
   sbgpu::check(sbgpu_bamreads_reads(F.reads, &reads, &d_ref, &d_left, &d_right), "sbgpu_bamreads_reads");
   sbgpu_clusters_t cl = {n_all, c_ref.data(), c_left.data(), c_right.data(), c_strand.data()};
   void *d_cluster = nullptr;
   hip_check(hipMalloc(&d_cluster, (size_t)(n_reads + 1) * 4), "hipMalloc");
   std::vector<int64_t> off((size_t)n_all + 1, 0);
   sbgpu::check(sbgpu_assign_reads_device(ctx.get(), &cl, n_reads, d_ref, d_left, d_right, (uint8_t *)reads.flags, (int32_t *)d_cluster, off.data(),
                                          nullptr),
                "sbgpu_assign_reads_device");
   // The reference stops making clusters once the file is exhausted (nextClusterRefDemand returns -1 when no record
   // remains, alignments.cpp:1152): the cluster during whose pass the last record is consumed is the last one.
   int64_t n_loci = n_all ? 1 : 0;
   while (n_loci < n_all && off[(size_t)n_loci] < n_reads) ++n_loci;
   F.loci.resize((size_t)n_loci);
   off.resize((size_t)n_loci + 1);
   _num_cluster = (int)n_loci;
   // pairs, then unique hits
   sbgpu_matepairs_t *mp = nullptr;
   sbgpu::check(sbgpu_pair_mates_device(ctx.get(), n_loci, &reads, off.data(), nullptr, &mp), "sbgpu_pair_mates_device");
   sbgpu_pairs_t pairs;
   const int64_t *pair_off = nullptr;
   sbgpu::check(sbgpu_matepairs_pairs(mp, &pairs, &pair_off), "sbgpu_matepairs_pairs");
   sbgpu_uniq_dev_t *uq = nullptr;
   sbgpu::check(sbgpu_collapse_pairs_device(ctx.get(), n_loci, &pairs, pair_off, nullptr, &uq), "sbgpu_collapse_pairs_device");
   int64_t ui[8];
   sbgpu::check(sbgpu_uniq_dev_info(uq, ui), "sbgpu_uniq_dev_info");
   // the batch: annotation from the clusters' transcripts, hits from the collapse
   sbgpu::LocusBatch &B = F.batch;
   for (const Locus &lc : F.loci) {
      std::vector<std::vector<std::pair<uint32_t, uint32_t>>> tx;
      for (const Contig &t : lc.transcripts) { // the S_MATCH features are the exons (estimate.hpp:80-84)
         tx.emplace_back();
         for (const auto &f : t._genomic_feats)
            if (f._match_op._code == Match_t::S_MATCH) tx.back().emplace_back(f.left(), f.right());
      }
      B.add_locus(tx);
   }
   // ---- without -f nothing but the abundances is wanted: the unique hits stay where they are (round 6)
   if (!print_frag_context && ui[0] > 0) {
      const int64_t nl = n_loci;
      B.seg_off.assign((size_t)nl + 1, 0);
      int64_t ns = sbgpu_segments_host(nl, B.iso_off.data(), B.exon_off.data(), B.exon_left.data(), B.exon_right.data(), B.seg_off.data(), nullptr,
                                       nullptr, 0);
      sbgpu::check((int)(ns < 0 ? ns : 0), "sbgpu_segments_host");
      B.seg_left.assign((size_t)ns + 1, 0);
      B.seg_right.assign((size_t)ns + 1, 0);
      sbgpu_segments_host(nl, B.iso_off.data(), B.exon_off.data(), B.exon_left.data(), B.exon_right.data(), B.seg_off.data(), B.seg_left.data(),
                          B.seg_right.data(), ns);
      sbgpu_annotation_t an = B.annotation();
      sbgpu_hits_t dh;
      const float *d_mass = nullptr;
      const int64_t *hoff = nullptr;
      sbgpu::check(sbgpu_uniq_dev_hits(uq, &dh, &d_mass, &hoff), "sbgpu_uniq_dev_hits");
      // the law main will make AFTER this function (Strawberry.cpp:329-356): N(200, 80) for a single-end library, -i's, or --
      // the default -- the empirical one of pass 1, which the device builds (insert == NULL)
      sbgpu_insert_t given = {};
      const sbgpu_insert_t *ins = nullptr;
      if (long_read_sample || SINGLE_END_EXP) given.mean = 200.0, given.sd = 80.0, ins = &given;
      else if (kInsertSizeMean != 0 && kInsertSizeSD != 0) given.mean = kInsertSizeMean, given.sd = kInsertSizeSD, ins = &given;
      sbgpu_abundance_params_t par = {};
      par.filter_by_expression = filter_by_expression ? 1 : 0;
      par.min_isoform_frac = kMinIsoformFrac;
      par.effective_len_norm = effective_len_norm ? 1 : 0;
      const size_t n_iso = (size_t)B.iso_off.back();
      F.r_theta.assign(n_iso + 1, 0.0), F.r_fpkm.assign(n_iso + 1, 0.0), F.r_frac.assign(n_iso + 1, 0.0), F.r_tpm.assign(n_iso + 1, 0.0);
      F.r_keep.assign(n_iso + 1, 0), F.r_status.assign((size_t)nl + 1, 0);
      sbgpu_abundances_t res = {};
      res.theta = F.r_theta.data(), res.fpkm = F.r_fpkm.data(), res.frac = F.r_frac.data(), res.tpm = F.r_tpm.data();
      res.keep = F.r_keep.data(), res.status = F.r_status.data();
      sbgpu_insert_t used;
      sbgpu_bins_t *bins = nullptr;
      const int rc = sbgpu_quantify_resident(ctx.get(), &an, &dh, d_mass, hoff, ins, _hit_factory->_reads_table.read_len_mode(), long_read_sample ? 1 : 0,
                                             ui[4], &par, nullptr, &used, &res, &bins);
      if (rc == SBGPU_OK) {
         F.resident = true;
         _total_mapped_reads = (int)ui[4]; // fragLenDist's total (alignments.cpp:1372): the sum over the clusters of (int) weighted_mass()
         if (!ins) // main makes InsertSize(_frag_dist) next: the same sample, length by length (the law's own histogram)
            for (int32_t l = used.start_offset; l <= used.end_offset; ++l)
               for (int64_t k = 0; k < (int64_t)used.emp_hist[l - used.start_offset]; ++k) _hit_factory->_reads_table._frag_dist.push_back(l);
         sbgpu_bins_destroy(bins);
      } else if (rc != SBGPU_EUNSUPPORTED) { // (declined -- fractional masses, ... --: the host route below serves)
         sbgpu::check(rc, "sbgpu_quantify_resident");
      }
   }
   if (F.resident) {
      sbgpu_uniq_dev_destroy(uq);
      sbgpu_matepairs_destroy(mp);
      sbgpu_bamreads_destroy(F.reads);
      F.reads = nullptr;
      hip_check(hipFree(d_cluster), "hipFree");
      hip_check(hipFree(F.d_bytes), "hipFree");
      hip_check(hipFree(F.d_rec_off), "hipFree");
      F.d_bytes = F.d_rec_off = nullptr;
      std::vector<uint8_t>().swap(F.raw);
      for (int64_t l = 0; l < n_loci; ++l) {
         const Locus &lc = F.loci[(size_t)l];
         std::fprintf(log, "Finish inspecting locus: %s:%d-%d\n", _hit_factory->_ref_table.ref_real_name(lc.ref_id).c_str(), lc.left, lc.right);
         std::fprintf(log, "Found %d of ref mRNAs from the reference gtf file.\n", (int)lc.transcripts.size());
      }
      F.t_front = secs(t0, clk::now());
      return;
   }
   B.hit_locus.assign((size_t)ui[0], 0);
   B.feat_off.assign((size_t)ui[0] + 1, 0);
   B.feat_code.assign((size_t)ui[1], 0);
   B.feat_left.assign((size_t)ui[1], 0);
   B.feat_right.assign((size_t)ui[1], 0);
   B.hit_mass.assign((size_t)ui[0], 0.0f);
   std::vector<double> cluster_mass((size_t)n_loci + 1, 0.0);
   sbgpu::check(sbgpu_uniq_dev_export(uq, B.hit_locus.data(), B.feat_off.data(), B.feat_code.data(), B.feat_left.data(), B.feat_right.data(),
                                      B.hit_mass.data(), cluster_mass.data()),
                "sbgpu_uniq_dev_export");
   sbgpu_uniq_dev_destroy(uq);
   sbgpu_matepairs_destroy(mp);
   sbgpu_bamreads_destroy(F.reads);
   F.reads = nullptr;
   hip_check(hipFree(d_cluster), "hipFree");
   hip_check(hipFree(F.d_bytes), "hipFree");
   hip_check(hipFree(F.d_rec_off), "hipFree");
   F.d_bytes = F.d_rec_off = nullptr;
   std::vector<uint8_t>().swap(F.raw);
   // fragLenDist (alignments.cpp:1363-1430): the mapped-read total ...
   int total = 0;
   for (int64_t l = 0; l < n_loci; ++l) total += (int)cluster_mass[(size_t)l];
   _total_mapped_reads = total;
   // ... and, when no -i was given, the fragment lengths of the hits that fit exactly one transcript
   if (!(kInsertSizeMean != 0 && kInsertSizeSD != 0) && !SINGLE_END_EXP && ui[0] > 0) {
      const int64_t nl = n_loci;
      B.seg_off.assign((size_t)nl + 1, 0);
      int64_t ns = sbgpu_segments_host(nl, B.iso_off.data(), B.exon_off.data(), B.exon_left.data(), B.exon_right.data(), B.seg_off.data(), nullptr,
                                       nullptr, 0);
      sbgpu::check((int)(ns < 0 ? ns : 0), "sbgpu_segments_host");
      B.seg_left.assign((size_t)ns + 1, 0);
      B.seg_right.assign((size_t)ns + 1, 0);
      sbgpu_segments_host(nl, B.iso_off.data(), B.exon_off.data(), B.exon_left.data(), B.exon_right.data(), B.seg_off.data(), B.seg_left.data(),
                          B.seg_right.data(), ns);
      int64_t max_iso = 1, max_seg = 1;
      for (int64_t l = 0; l < nl; ++l) {
         max_iso = std::max(max_iso, B.iso_off[(size_t)l + 1] - B.iso_off[(size_t)l]);
         max_seg = std::max(max_seg, B.seg_off[(size_t)l + 1] - B.seg_off[(size_t)l]);
      }
      const int32_t cw = (int32_t)((max_iso + 31) / 32), kw = (int32_t)((max_seg + 31) / 32);
      std::vector<uint32_t> compat((size_t)ui[0] * cw + 1, 0), key((size_t)ui[0] * kw + 1, 0);
      sbgpu_annotation_t an = B.annotation();
      sbgpu_hits_t ht = B.hits();
      sbgpu::check(sbgpu_exonbin_host(ctx.get(), &an, &ht, cw, kw, compat.data(), key.data()), "sbgpu_exonbin_host");
      std::vector<int32_t> fl((size_t)ui[0], -1);
      sbgpu::check((int)std::min<int64_t>(0, sbgpu_frag_lens_host(&an, &ht, cw, compat.data(), fl.data())), "sbgpu_frag_lens_host");
      for (int32_t v : fl)
         if (v >= 0) _hit_factory->_reads_table._frag_dist.push_back(v);
   }
   for (int64_t l = 0; l < n_loci; ++l) {
      const Locus &lc = F.loci[(size_t)l];
      std::fprintf(log, "Finish inspecting locus: %s:%d-%d\n", _hit_factory->_ref_table.ref_real_name(lc.ref_id).c_str(), lc.left, lc.right);
      std::fprintf(log, "Found %d of ref mRNAs from the reference gtf file.\n", (int)lc.transcripts.size());
   }
   F.t_front = secs(t0, clk::now());
}

// replaces /root/reference/src/alignments.cpp:1736-1834
void Sample::procSample(FILE *pfile, FILE *plogfile, FILE *fragfile)
{
   Front &F = front();
   const clk::time_point t_begin = clk::now();
   const RefSeqTable &ref_t = _hit_factory->_ref_table;
   if (fragfile != NULL) { // the -f table's header (alignments.cpp:1746-1752)
      std::vector<std::string> header = {"sample", "sample_frag_count", "gene_id", "gene_frag_count", "transcripts", "FPKMs",
                                         "conditional_probabilities", "class_probabilities", "path_symbol", "path_count",
                                         "path_gc_content", "path_hexmer_entropy", "gc_stretch_0.8_20", "gc_stretch_0.9_20",
                                         "gc_stretch_0.8_40", "gc_stretch_0.9_40"};
      pretty_print(fragfile, header, "\t");
   }
   sbgpu::LocusBatch &batch = F.batch;
   const std::vector<Locus> &loci = F.loci;
   if (F.resident) {
      // everything was computed in preProcess, on the device: the theta log (estimate.cpp:310-313), quantifyCluster's notice
      // (alignments.cpp:1531-1532), the isoforms that survive the filter (estimate.cpp:346-355) with the reference's own print2gtf
      for (int64_t l = 0; l < batch.n_loci(); ++l) {
         if (F.r_status[(size_t)l] == SBGPU_EM_INIT_EMPTY) continue; // estimate_abundances() false: the locus is omitted
         const Locus &lc = loci[(size_t)l];
         const int64_t j0 = batch.iso_off[(size_t)l], niso = batch.iso_off[(size_t)l + 1] - j0;
         for (int64_t j = 0; j < niso; ++j)
            std::fprintf(plogfile, "isoform %d has %f raw read count.\n", (int)j + 1, F.r_theta[(size_t)(j0 + j)]);
         std::cerr << ref_t.ref_real_name(lc.ref_id) << "\t" << lc.left << "\t" << lc.right << " finishes abundances estimation" << std::endl;
      }
      for (int64_t l = 0; l < batch.n_loci(); ++l) {
         if (F.r_status[(size_t)l] == SBGPU_EM_INIT_EMPTY) continue;
         const Locus &lc = loci[(size_t)l];
         const int64_t j0 = batch.iso_off[(size_t)l], niso = batch.iso_off[(size_t)l + 1] - j0;
         for (int64_t j = 0; j < niso; ++j) {
            const size_t g = (size_t)(j0 + j);
            if (!F.r_keep[g]) continue;
            const bool na = F.r_keep[g] == 2; // effective length below zero: "NA" (estimate.cpp:319-322, :338-341)
            const Contig &t = lc.transcripts[(size_t)j];
            t.print2gtf(pfile, _hit_factory->_ref_table, na ? std::string("NA") : std::to_string(F.r_fpkm[g]), na ? std::string("NA") : std::to_string(F.r_frac[g]),
                        std::to_string(F.r_tpm[g]), t.parent_id(), t.annotated_trans_id(), t.ref_gene_id(), t.ref_gene_name());
         }
      }
      const char *timing = std::getenv("SBGPU_DROPIN_TIMING");
      if (timing && timing[0] == '1')
         std::fprintf(stderr, "sbgpu_front (resident): inflate + index %.3f s (%lld records) | upload + sbgpu_bam_decode_device + read lengths %.3f s | clusters, "
                              "stream, pairs, unique hits, pass 1, bins, weights, EM, FPKM, TPM (device) %.3f s | output %.3f s\n",
                      F.t_inflate, (long long)F.n_records, F.t_decode, F.t_front, secs(t_begin, clk::now()));
      return;
   }
   // ---- solve: bins, weights and the EM of all loci in ONE call; the reference's epilogue arithmetic (LocusBatch::quantify)
   sbgpu::InsertSize ins;
   // (a long-read sample has no insert-size law -- main never makes one, Strawberry.cpp:338-353 -- and needs none: its
   // bin weights are 1 / L_j, estimate.cpp:236-247; the library is handed a placeholder it does not read)
   if (_insert_size_dist) {
      ins.mean = _insert_size_dist->_mean;
      ins.sd = _insert_size_dist->_sd;
      ins.use_emp = _insert_size_dist->_use_emp;
      ins.start_offset = _insert_size_dist->_start_offset;
      ins.end_offset = _insert_size_dist->_end_offset;
      ins.total_reads = _insert_size_dist->_total_reads;
      ins.emp_dist = _insert_size_dist->_emp_dist;
   } else {
      ins.mean = 200.0, ins.sd = 80.0;
   }
   sbgpu_abundance_params_t par = {};
   par.total_mapped_reads = total_mapped_reads();
   par.filter_by_expression = filter_by_expression ? 1 : 0;
   par.min_isoform_frac = kMinIsoformFrac;
   par.effective_len_norm = effective_len_norm ? 1 : 0;
   par.insert_mean = _insert_size_dist ? _insert_size_dist->_mean : 0.0;
   if (batch.n_loci() > 0)
      batch.quantify(device_context(), &ins, _hit_factory->_reads_table.read_len_mode(), par, long_read_sample);
   const clk::time_point t_solved = clk::now();

   // ---- epilogue, locus by locus (as oracle/sbgpu_chain_shim.cpp): the theta log (estimate.cpp:310-313), quantifyCluster's
   // notice (alignments.cpp:1531-1532), the -f table (printContext), the isoforms that survive the filter (estimate.cpp:346-355)
   struct Out {
      const Contig *t;
      const sbgpu::Isoform *iso;
   };
   std::vector<Out> isoforms;
   const std::string sample = sample_name();
   std::vector<char> buf(1 << 20);
   const int64_t n_bins = batch.n_loci() ? batch.row_off.back() : 0;
   std::vector<int64_t> last_hit((size_t)n_bins, -1), n_in_bin((size_t)n_bins, 0);
   if (fragfile != NULL)
      for (int64_t h = 0; h < batch.n_hits(); ++h) {
         const int64_t b = batch.hit_bin[(size_t)h];
         if (b < 0) continue;
         const int64_t l = batch.hit_locus[(size_t)h], j0 = batch.iso_off[(size_t)l], niso = batch.iso_off[(size_t)l + 1] - j0;
         bool any = false;
         for (int64_t j = 0; j < niso; ++j)
            any |= ((batch.compat[(size_t)(h * batch.compat_words + (j >> 5))] >> (j & 31)) & 1u) && batch.isoforms[(size_t)(j0 + j)].kept;
         if (!any) continue;
         last_hit[(size_t)b] = h;
         ++n_in_bin[(size_t)b];
      }
   for (int64_t l = 0; l < batch.n_loci(); ++l) {
      const Locus &lc = loci[(size_t)l];
      if (batch.status[(size_t)l] == SBGPU_EM_INIT_EMPTY) continue; // estimate_abundances() false: the locus is omitted
      const int64_t j0 = batch.iso_off[(size_t)l], niso = batch.iso_off[(size_t)l + 1] - j0;
      for (int64_t j = 0; j < niso; ++j)
         std::fprintf(plogfile, "isoform %d has %f raw read count.\n", (int)j + 1, batch.theta[(size_t)(j0 + j)]);
      std::vector<int64_t> kept;
      for (int64_t j = 0; j < niso; ++j)
         if (batch.isoforms[(size_t)(j0 + j)].kept) {
            kept.push_back(j);
            isoforms.push_back({&lc.transcripts[(size_t)j], &batch.isoforms[(size_t)(j0 + j)]});
         }
      std::cerr << ref_t.ref_real_name(lc.ref_id) << "\t" << lc.left << "\t" << lc.right << " finishes abundances estimation" << std::endl;
      if (fragfile == NULL || kept.empty()) continue;
      const int64_t b0 = batch.row_off[(size_t)l], b1 = batch.row_off[(size_t)l + 1];
      const int64_t s0 = batch.seg_off[(size_t)l], nseg = batch.seg_off[(size_t)l + 1] - s0;
      uint32_t gene_frags = 0;
      std::map<std::vector<std::pair<uint32_t, uint32_t>>, int64_t> by_coords; // the std::map order of printContext
      for (int64_t b = b0; b < b1; ++b) {
         if (n_in_bin[(size_t)b] == 0) continue;
         std::vector<std::pair<uint32_t, uint32_t>> coords;
         for (int64_t s = 0; s < nseg; ++s)
            if ((batch.bin_key[(size_t)(b * batch.key_words + (s >> 5))] >> (s & 31)) & 1u)
               coords.emplace_back(batch.seg_left[(size_t)(s0 + s)], batch.seg_right[(size_t)(s0 + s)]);
         by_coords[coords] = b;
         gene_frags += (uint32_t)n_in_bin[(size_t)b];
      }
      std::vector<std::string> name_s;
      std::vector<const char *> names;
      std::vector<double> fpkm, frac;
      for (int64_t j : kept) name_s.push_back(lc.transcripts[(size_t)j].annotated_trans_id());
      for (size_t k = 0; k < kept.size(); ++k) {
         names.push_back(name_s[k].c_str());
         fpkm.push_back(batch.isoforms[(size_t)(j0 + kept[k])].FPKM);
         frac.push_back(batch.isoforms[(size_t)(j0 + kept[k])].frac);
      }
      const std::string gene = lc.transcripts[(size_t)kept[0]].parent_id();
      for (const auto &kv : by_coords) {
         const int64_t b = kv.second, h = last_hit[(size_t)b];
         std::vector<double> prob;
         std::vector<uint32_t> sl, sr;
         for (int64_t j : kept)
            prob.push_back(((batch.compat[(size_t)(h * batch.compat_words + (j >> 5))] >> (j & 31)) & 1u)
                              ? batch.F[(size_t)(batch.f_off[(size_t)l] + (b - b0) * niso + j)]
                              : 0.0);
         for (const auto &c : kv.first) {
            sl.push_back(c.first);
            sr.push_back(c.second);
         }
         const int n = sbgpu_format_context_row(buf.data(), (int)buf.size(), sample.c_str(), total_mapped_reads(), gene.c_str(), gene_frags,
                                                (int)kept.size(), names.data(), fpkm.data(), prob.data(), frac.data(), (int)sl.size(),
                                                sl.data(), sr.data(), (uint32_t)n_in_bin[(size_t)b]);
         sbgpu::check(n, "sbgpu_format_context_row");
         std::fwrite(buf.data(), 1, (size_t)n, fragfile);
      }
   }
   // alignments.cpp:1821-1834: TPM over the surviving isoforms, then the reference's own print2gtf
   double total_fpkm = 0.0;
   for (const Out &o : isoforms) total_fpkm += o.iso->FPKM;
   for (const Out &o : isoforms) {
      const double tpm = 1e6 * o.iso->FPKM / total_fpkm;
      o.t->print2gtf(pfile, _hit_factory->_ref_table, o.iso->FPKM_s, o.iso->frac_s, std::to_string(tpm), o.t->parent_id(),
                     o.t->annotated_trans_id(), o.t->ref_gene_id(), o.t->ref_gene_name());
   }
   const char *timing = std::getenv("SBGPU_DROPIN_TIMING");
   if (timing && timing[0] == '1') {
      const clk::time_point t_end = clk::now();
      std::fprintf(stderr, "sbgpu_front: inflate + index %.3f s (%lld records) | upload + sbgpu_bam_decode_device + read lengths %.3f s | clusters, "
                           "stream, pairs, unique hits (device) + their download %.3f s | ONE sbgpu_quantify_host call (%lld loci, %lld unique "
                           "hits) and the epilogue arithmetic %.3f s | output %.3f s\n",
                   F.t_inflate, (long long)F.n_records, F.t_decode, F.t_front, (long long)batch.n_loci(), (long long)batch.n_hits(),
                   secs(t_begin, t_solved), secs(t_solved, t_end));
   }
}
