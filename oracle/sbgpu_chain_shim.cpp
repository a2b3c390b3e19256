// oracle/sbgpu_chain_shim.cpp -- TEST INFRASTRUCTURE ONLY (nothing under strawberry_amd/ refers to it).
//
// The drop-in one level up ("one level up": Sample::quantifyCluster): `make -C oracle ref` links
// oracle/_ref/strawberry_sbgpu_chain from the reference's UNMODIFIED objects with ONE function replaced, Sample::procSample
// (/root/reference/src/alignments.cpp:1736-1834, weakened in a copy of alignments.o).  The replacement walks the clusters
// with the reference's own classes -- BAM decode, nextClusterRefDemand, finalizeCluster (mate pairing, duplicate collapse),
// Contig(PairedHit) -- and hands every locus' transcripts and unique hits to ONE sbgpu::LocusBatch; a single
// sbgpu_quantify_host call then does, for the whole sample on the device, what LocusContext's constructor and
// estimate_abundances do per locus on the host (exon bins, bin weights, EM: estimate.hpp:60-109, estimate.cpp:135-355);
// the output is written by the reference's own Contig::print2gtf (with the numbers the device produced) and by the
// library's formatter for the -f table (sbgpu_format_context_row: Sample::printContext, alignments.cpp:1549-1639).
// No LocusContext is ever built: bins and weights are 45 % of the reference's quantification time.
//
// tests/test_reference_driver_gpu.py: its files equal the reference binary's byte for byte; tools/dropin_timing.py times it.
// -b genome.fa (round 5): the chromosomes' bases come from the reference's own FaSeqGetter (Sample::load_chrom_fasta,
// alignments.cpp:811-819, as its loop does at :1763-1779); the six sequence columns of the -f table (alignments.cpp:1622-1636)
// are the library's -- sbgpu_binseq_host for all bins of a chromosome at once, sbgpu_format_context_row_seq per row.
#include "alignments.h" // the reference's: /root/reference/include/alignments.h:178-290 (Sample)
#include "estimate.hpp"

#include <chrono>
#include <climits>
#include <cstdlib>
#include <map>
#include <memory>

#include "sbgpu_host.hpp"

namespace {
const sbgpu::Context &device_context()
{
   static const sbgpu::Context ctx(0); // throws (no CPU fallback) when there is no gfx950 device
   return ctx;
}
struct Locus {
   RefID ref_id;
   uint left, right;
   std::vector<Contig> transcripts; // cluster->ref_mRNAs(): the reference's isoform order
};
} // namespace

// replaces /root/reference/src/alignments.cpp:1736-1834
void Sample::procSample(FILE *pfile, FILE *plogfile, FILE *fragfile)
{
   using clk = std::chrono::steady_clock;
   auto secs = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double>(b - a).count(); };
   const clk::time_point t_begin = clk::now();
   _hit_factory->reset();
   reset_refmRNAs();
   const RefSeqTable &ref_t = _hit_factory->_ref_table;
   if (fragfile != NULL) { // the -f table's header (alignments.cpp:1746-1752)
      std::vector<std::string> header = {"sample", "sample_frag_count", "gene_id", "gene_frag_count", "transcripts", "FPKMs",
                                         "conditional_probabilities", "class_probabilities", "path_symbol", "path_count",
                                         "path_gc_content", "path_hexmer_entropy", "gc_stretch_0.8_20", "gc_stretch_0.9_20",
                                         "gc_stretch_0.8_40", "gc_stretch_0.9_40"};
      pretty_print(fragfile, header, "\t");
   }
   // ---- collect: transcripts and unique hits of every locus, in cluster order
   sbgpu::LocusBatch batch;
   std::vector<Locus> loci;
   std::vector<uint8_t> code;
   std::vector<uint32_t> fl, fr;
   // -b: every chromosome's bases, loaded when the clusters reach it (the reference's loop: alignments.cpp:1763-1779)
   std::map<RefID, std::string> chrom_seq;
   while (true) {
      std::shared_ptr<HitCluster> cluster(new HitCluster());
      if (-1 == nextClusterRefDemand(*cluster)) break;
      if (cluster->ref_id() == -1) continue;
      if (BIAS_CORRECTION && !chrom_seq.count(cluster->ref_id())) {
         load_chrom_fasta(cluster->ref_id());
         const uint n = _fasta_getter->loadSeq(); // (the number of bases it holds now)
         std::string &seq = chrom_seq[cluster->ref_id()];
         for (uint at = 1; at <= n; at += 1u << 20) seq += _fasta_getter->fetchSeq(at, std::min<uint>(1u << 20, n - at + 1)); // (fetchSeq builds its answer on the stack)
      }
      finalizeCluster(cluster, true);
      Locus lc = {cluster->ref_id(), cluster->left(), cluster->right(), cluster->ref_mRNAs()};
      std::vector<std::vector<std::pair<uint32_t, uint32_t>>> tx;
      for (const Contig &t : lc.transcripts) { // the S_MATCH features are the exons (estimate.hpp:80-84)
         tx.emplace_back();
         for (const auto &f : t._genomic_feats)
            if (f._match_op._code == Match_t::S_MATCH) tx.back().emplace_back(f.left(), f.right());
      }
      const int32_t l = (int32_t)batch.add_locus(tx);
      for (auto r = cluster->uniq_hits().cbegin(); r != cluster->uniq_hits().cend(); ++r) { // estimate.hpp:68-78
         Contig hit(*r);
         if (hit.ref_id() == -1) {
            std::fprintf(plogfile, "paired reads %s and %s are not compatible\n", r->left_read_obj().read_name().c_str(),
                         r->_right_read->read_name().c_str());
            continue;
         }
         code.clear(), fl.clear(), fr.clear();
         for (const auto &f : hit._genomic_feats) {
            code.push_back((uint8_t)f._match_op._code);
            fl.push_back(f.left());
            fr.push_back(f.right());
         }
         batch.add_hit(l, (int)code.size(), code.data(), fl.data(), fr.data(), hit.mass());
      }
      loci.push_back(std::move(lc));
   }
   const clk::time_point t_collected = clk::now();

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

   // ---- epilogue, locus by locus: the theta log (estimate.cpp:310-313), quantifyCluster's notice (alignments.cpp:1531-1532),
   // the -f table (printContext), and the isoforms that survive the filter (estimate.cpp:346-355)
   struct Out {
      const Contig *t;
      const sbgpu::Isoform *iso;
   };
   std::vector<Out> isoforms;
   const std::string sample = sample_name();
   std::vector<char> buf(1 << 20);
   // hits per bin, counted over the surviving isoforms only, and each bin's LAST such hit (printContext runs after the
   // filter: get_frag_info, estimate.hpp:173-196)
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
   // -b: GC ratio, hexamer entropy and the four high-GC-stretch flags of every bin's sequence (its segments' bases in
   // order: ExonBin::bin_dnaseq, isoform.h:173-182), a chromosome's bins in one sbgpu_binseq_host call
   std::vector<double> bin_gc((size_t)n_bins, 0.0), bin_entropy((size_t)n_bins, 0.0);
   std::vector<uint8_t> bin_flags((size_t)n_bins, 0);
   if (BIAS_CORRECTION && fragfile != NULL) {
      for (int64_t l0 = 0; l0 < batch.n_loci();) {
         int64_t l1 = l0;
         while (l1 < batch.n_loci() && loci[(size_t)l1].ref_id == loci[(size_t)l0].ref_id) ++l1;
         const std::string &seq = chrom_seq[loci[(size_t)l0].ref_id];
         const int64_t b0 = batch.row_off[(size_t)l0], b1 = batch.row_off[(size_t)l1];
         std::vector<int64_t> off(1, 0);
         std::vector<uint32_t> sl, sr;
         for (int64_t l = l0; l < l1; ++l) {
            const int64_t s0 = batch.seg_off[(size_t)l], nseg = batch.seg_off[(size_t)l + 1] - s0;
            for (int64_t b = batch.row_off[(size_t)l]; b < batch.row_off[(size_t)l + 1]; ++b) {
               for (int64_t s2 = 0; s2 < nseg; ++s2)
                  if ((batch.bin_key[(size_t)(b * batch.key_words + (s2 >> 5))] >> (s2 & 31)) & 1u) {
                     sl.push_back(batch.seg_left[(size_t)(s0 + s2)]);
                     sr.push_back(batch.seg_right[(size_t)(s0 + s2)]);
                  }
               off.push_back((int64_t)sl.size());
            }
         }
         if (b1 > b0)
            sbgpu::check(sbgpu_binseq_host(device_context().get(), (const uint8_t *)seq.data(), 1, (int64_t)seq.size(), b1 - b0, off.data(),
                                           sl.data(), sr.data(), bin_gc.data() + b0, bin_entropy.data() + b0, bin_flags.data() + b0),
                         "sbgpu_binseq_host");
         l0 = l1;
      }
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
         const int n = BIAS_CORRECTION
                          ? sbgpu_format_context_row_seq(buf.data(), (int)buf.size(), sample.c_str(), total_mapped_reads(), gene.c_str(), gene_frags,
                                                         (int)kept.size(), names.data(), fpkm.data(), prob.data(), frac.data(), (int)sl.size(),
                                                         sl.data(), sr.data(), (uint32_t)n_in_bin[(size_t)b], bin_gc[(size_t)b],
                                                         bin_entropy[(size_t)b], bin_flags[(size_t)b])
                          : sbgpu_format_context_row(buf.data(), (int)buf.size(), sample.c_str(), total_mapped_reads(), gene.c_str(), gene_frags,
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
      std::fprintf(stderr, "sbgpu_chain procSample: total %.3f s = collect (BAM pass 2, clustering, pairing, collapse, Contig(PairedHit)) %.3f s + "
                           "ONE sbgpu_quantify_host call (%lld loci, %lld unique hits: bins, weights, EM; upload included) and the epilogue arithmetic %.3f s + "
                           "output %.3f s\n",
                   secs(t_begin, t_end), secs(t_begin, t_collected), (long long)batch.n_loci(), (long long)batch.n_hits(),
                   secs(t_collected, t_solved), secs(t_solved, t_end));
   }
}
