// examples/quantify_fragments.cpp -- a C++14 driver over include/sbgpu_host.hpp, the way
// Strawberry's own Sample::procSample would use libsbgpu.so: read an annotation
// and read pairs, quantify every locus in one batch on the GPU, write the reference's two output
// files (GTF and the -f context table).  Used by tests/test_cpp_driver.py, which checks the files
// byte for byte against what the reference binary wrote for the same input.
//
//   g++ -std=c++14 -O2 -Iinclude examples/quantify_fragments.cpp -Lstrawberry_amd/lib -lsbgpu
//       -Wl,-rpath,$PWD/strawberry_amd/lib -o quantify_fragments
//   ./quantify_fragments input.txt out.gtf ctx.tsv [genome.fa] [--rank R --world W --comm-id FILE [--comm-nonce N]]
//         (genome.fa: the reference's `-b` option -- six sequence columns per bin in the -f table,
//          src/alignments.cpp:1622-1636)
//   Several GPUs: one process per GPU, rank R of W on GPU R; locus l belongs to rank l mod W.  Every rank reads
//   the whole input, keeps its loci and their pairs, and writes out.gtf / ctx.tsv with the suffix ".rank<R>";
//   the two sums over all loci -- mapped reads before the EM, FPKM after it (src/alignments.cpp:1372, :1821-1824)
//   -- are all-reduced over RCCL (sbgpu::Comm; rank 0 leaves the RCCL id in FILE for the others).
//
// Input (plain text, whitespace separated):
//   sample <name>  chrom <name>  strand <+|->  insert <mean> <sd>  read_len <n>  min_isoform_frac <x>  long_read <0|1>
//         (insert 0 0: no -i, empirical distribution; long_read: the reference's long-read workflow,
//          Strawberry.cpp:292-303, decided by the caller from the read lengths)
//   loci <L>
//     locus <gene_id> <n_isoforms> <+|-> <chrom>                (the gene's strand and chromosome)
//       iso <transcript_id> <n_exons> <left> <right> ...          (the reference's isoform order)
//   pairs <P>
//     pair <locus> <mass> <n_left_blocks> <l> <r> ... <n_right_blocks> <l> <r> ...
//         (every sequenced read pair, in any order: aligned blocks of each mate; mass = the pair's raw
//          mass, 1 / NH -- duplicates are collapsed by the library like the reference does)
#include <algorithm>
#include <cstdio>
#include <fstream>
#include <iostream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

#include "sbgpu_host.hpp"

namespace {

struct Pair {
   int locus;
   double mass;
   std::vector<std::pair<uint32_t, uint32_t>> left, right;
};

// readhit_2_genomicFeats for an M/N CIGAR (src/contig.cpp:12-53): blocks with the introns between them
void mate_features(const std::vector<std::pair<uint32_t, uint32_t>> &blocks, std::vector<uint8_t> &c, std::vector<uint32_t> &l,
                   std::vector<uint32_t> &r)
{
   for (size_t k = 0; k < blocks.size(); ++k) {
      if (k) {
         c.push_back(1);
         l.push_back(blocks[k - 1].second + 1);
         r.push_back(blocks[k].first - 1);
      }
      c.push_back(0);
      l.push_back(blocks[k].first);
      r.push_back(blocks[k].second);
   }
}

} // namespace

int main(int argc, char **argv)
{
   int rank = 0, world = 1;
   std::string comm_id_file;
   unsigned long long comm_nonce = 0; // the same value on every rank of ONE launch: an older launch's id file is not ours
   {
      // trailing options; what is left are the positional arguments
      int n = argc;
      for (int i = 1; i + 1 < n;) {
         const std::string a = argv[i];
         if (a == "--rank" || a == "--world" || a == "--comm-id" || a == "--comm-nonce") {
            if (a == "--rank") rank = std::atoi(argv[i + 1]);
            else if (a == "--world") world = std::atoi(argv[i + 1]);
            else if (a == "--comm-nonce") comm_nonce = std::strtoull(argv[i + 1], nullptr, 10);
            else comm_id_file = argv[i + 1];
            for (int k = i; k + 2 < n; ++k) argv[k] = argv[k + 2];
            n -= 2;
         } else {
            ++i;
         }
      }
      argc = n;
   }
   if ((argc != 4 && argc != 5) || world < 1 || rank < 0 || rank >= world) {
      std::fprintf(stderr, "usage: %s input.txt out.gtf ctx.tsv [genome.fa] [--rank R --world W --comm-id FILE [--comm-nonce N]]\n", argv[0]);
      return 2;
   }
   const std::string part = world > 1 ? ".rank" + std::to_string(rank) : std::string();
   std::ifstream in(argv[1]);
   std::string tok, sample, chrom, strand;
   double ins_mean = 0, ins_sd = 0, min_frac = 0;
   int read_len = 0;
   int long_read = 0;
   in >> tok >> sample >> tok >> chrom >> tok >> strand >> tok >> ins_mean >> ins_sd >> tok >> read_len >> tok >> min_frac >> tok >>
      long_read;
   int64_t L = 0, P = 0;
   in >> tok >> L;
   sbgpu::LocusBatch batch;
   std::vector<std::string> gene_id, gene_strand, gene_chrom;
   std::vector<std::vector<std::string>> tx_id;
   std::vector<std::vector<std::vector<std::pair<uint32_t, uint32_t>>>> tx_exons;
   for (int64_t l = 0; l < L; ++l) {
      std::string g, gs, gc;
      int niso;
      in >> tok >> g >> niso >> gs >> gc;
      gene_id.push_back(g);
      gene_strand.push_back(gs);
      gene_chrom.push_back(gc);
      tx_id.emplace_back();
      tx_exons.emplace_back();
      for (int j = 0; j < niso; ++j) {
         std::string t;
         int ne;
         in >> tok >> t >> ne;
         std::vector<std::pair<uint32_t, uint32_t>> ex((size_t)ne);
         for (auto &e : ex) in >> e.first >> e.second;
         tx_id.back().push_back(t);
         tx_exons.back().push_back(ex);
      }
      batch.add_locus(tx_exons.back());
   }
   in >> tok >> P;
   std::vector<Pair> pairs((size_t)P);
   for (auto &p : pairs) {
      int nl, nr;
      in >> tok >> p.locus >> p.mass >> nl;
      p.left.resize((size_t)nl);
      for (auto &b : p.left) in >> b.first >> b.second;
      in >> nr;
      p.right.resize((size_t)nr);
      for (auto &b : p.right) in >> b.first >> b.second;
   }
   if (!in) {
      std::fprintf(stderr, "malformed input\n");
      return 2;
   }
   // every sequenced copy goes in as it is; the library does HitCluster::collapseAndFilterHits
   // (sort by the pair's ends, span filter, collapse of equal pairs) and Contig(PairedHit)
   std::vector<int32_t> p_locus;
   std::vector<double> p_mass;
   std::vector<int64_t> lo{0}, ro{0};
   std::vector<uint8_t> lc, rc;
   std::vector<uint32_t> ll, lr, rl, rr;
   for (const Pair &p : pairs) {
      if (p.locus % world != rank) continue; // another rank's locus
      p_locus.push_back(p.locus);
      p_mass.push_back(p.mass);
      mate_features(p.left, lc, ll, lr);
      mate_features(p.right, rc, rl, rr);
      lo.push_back((int64_t)lc.size());
      ro.push_back((int64_t)rc.size());
   }
   const sbgpu_pairs_t raw = {(int64_t)p_locus.size(), p_locus.data(), p_mass.data(), lo.data(), lc.data(), ll.data(), lr.data(),
                              ro.data(), rc.data(), rl.data(), rr.data()};
   int total_mapped = 0;
   try {
      total_mapped = batch.set_hits_from_pairs(raw);
   } catch (const std::exception &e) {
      std::fprintf(stderr, "error: %s\n", e.what());
      return 1;
   }

   try {
      sbgpu::Context ctx(rank % std::max(1, sbgpu_device_count()));
      sbgpu::Comm comm(ctx, rank, world, comm_id_file, comm_nonce);
      // Sample::_total_mapped_reads counts the whole sample (src/alignments.cpp:1372): sum over the ranks
      total_mapped = (int)comm.allreduce_sum((int64_t)total_mapped);
      sbgpu::InsertSize ins(ins_mean, ins_sd); // `insert 0 0`: no -i, build the empirical distribution
      sbgpu_abundance_params_t par = {};
      par.total_mapped_reads = total_mapped;
      par.filter_by_expression = 1;
      par.min_isoform_frac = min_frac;
      batch.quantify(ctx, (ins_mean != 0 && ins_sd != 0) ? &ins : nullptr, read_len, par, long_read != 0); // Strawberry.cpp:339-356
      // loci of other ranks hold no hits here: they are not this rank's to report
      for (int64_t l = 0; l < L; ++l)
         if (l % world != rank)
            for (int64_t j = batch.iso_off[(size_t)l]; j < batch.iso_off[(size_t)l + 1]; ++j) batch.isoforms[(size_t)j].kept = false;
      // the FPKM total runs over every isoform of the sample (src/alignments.cpp:1821-1824): sum over the ranks
      sbgpu::finalize_tpm(batch.isoforms, comm.allreduce_sum(sbgpu::sum_fpkm(batch.isoforms)));
   } catch (const std::exception &e) {
      std::fprintf(stderr, "error: %s\n", e.what());
      return 1;
   }

   // ---- GTF: Contig::print2gtf for every kept isoform, locus by locus (src/alignments.cpp:1831-1834)
   std::ofstream gtf(std::string(argv[2]) + part);
   std::vector<char> buf(1 << 20);
   for (int64_t l = 0; l < L; ++l) {
      for (size_t j = 0; j < tx_id[(size_t)l].size(); ++j) {
         const sbgpu::Isoform &t = batch.isoforms[(size_t)batch.iso_off[(size_t)l] + j];
         if (!t.kept) continue;
         std::vector<int32_t> el, er;
         for (const auto &e : tx_exons[(size_t)l][j]) {
            el.push_back((int32_t)e.first);
            er.push_back((int32_t)e.second);
         }
         const int n = sbgpu_format_gtf_transcript(buf.data(), (int)buf.size(), gene_chrom[(size_t)l].c_str(), gene_strand[(size_t)l][0], gene_id[(size_t)l].c_str(),
                                                   tx_id[(size_t)l][j].c_str(), gene_id[(size_t)l].c_str(), gene_id[(size_t)l].c_str(),
                                                   (int)el.size(), el.data(), er.data(), t.FPKM, t.frac, t.TPM,
                                                   t.FPKM_s == "NA" ? 2 : 1);
         sbgpu::check(n, "sbgpu_format_gtf_transcript");
         gtf.write(buf.data(), n);
      }
   }
   // ---- the -f table: Sample::printContext (src/alignments.cpp:1549-1639)
   std::ofstream ctxf(std::string(argv[3]) + part);
   ctxf << "sample\tsample_frag_count\tgene_id\tgene_frag_count\ttranscripts\tFPKMs\tconditional_probabilities\t"
           "class_probabilities\tpath_symbol\tpath_count\tpath_gc_content\tpath_hexmer_entropy\tgc_stretch_0.8_20\t"
           "gc_stretch_0.9_20\tgc_stretch_0.8_40\tgc_stretch_0.9_40\n";
   // printContext runs after the expression filter, over the surviving isoforms only: a hit counts when
   // it is compatible with one of them (get_frag_info, include/estimate.hpp:173-196)
   const int64_t n_bins = batch.row_off.back();
   std::vector<int64_t> last_hit((size_t)n_bins, -1), n_in_bin((size_t)n_bins, 0);
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
   // -b genome.fa: GC ratio, hexamer entropy and high-GC-stretch flags of every bin's sequence, one kernel launch
   std::vector<double> bin_gc, bin_entropy;
   std::vector<uint8_t> bin_flags;
   if (argc == 5) {
      for (const std::string &gc : gene_chrom)
         if (gc != chrom) {
            std::fprintf(stderr, "genome.fa: this driver takes one chromosome per run (header `chrom %s`, a locus on %s)\n", chrom.c_str(), gc.c_str());
            return 2;
         }
      std::ifstream fa(argv[4]);
      if (!fa) {
         std::fprintf(stderr, "cannot open %s\n", argv[4]);
         return 2;
      }
      std::string genome, line;
      bool mine = false;
      while (std::getline(fa, line)) {
         if (!line.empty() && line[0] == '>')
            mine = line.substr(1, line.find_first_of(" \t") == std::string::npos ? std::string::npos : line.find_first_of(" \t") - 1) == chrom;
         else if (mine)
            genome += line;
      }
      try {
         sbgpu::Context ctx(rank % std::max(1, sbgpu_device_count()));
         sbgpu::BinSequenceStats st = sbgpu::bin_sequence_stats(ctx, batch, genome);
         bin_gc.swap(st.gc);
         bin_entropy.swap(st.entropy);
         bin_flags.swap(st.flags);
      } catch (const std::exception &e) {
         std::fprintf(stderr, "error: %s\n", e.what());
         return 1;
      }
   }
   for (int64_t l = 0; l < L; ++l) {
      const int64_t b0 = batch.row_off[(size_t)l], b1 = batch.row_off[(size_t)l + 1];
      const int64_t j0 = batch.iso_off[(size_t)l], niso = batch.iso_off[(size_t)l + 1] - j0;
      const int64_t s0 = batch.seg_off[(size_t)l], nseg = batch.seg_off[(size_t)l + 1] - s0;
      uint32_t gene_frags = 0;
      // bins in std::map order of their coordinate sets (:1552-1563)
      std::map<std::vector<std::pair<uint32_t, uint32_t>>, int64_t> by_coords;
      for (int64_t b = b0; b < b1; ++b) {
         if (n_in_bin[(size_t)b] == 0) continue;
         std::vector<std::pair<uint32_t, uint32_t>> coords;
         for (int64_t s = 0; s < nseg; ++s)
            if ((batch.bin_key[(size_t)(b * batch.key_words + (s >> 5))] >> (s & 31)) & 1u)
               coords.emplace_back(batch.seg_left[(size_t)(s0 + s)], batch.seg_right[(size_t)(s0 + s)]);
         by_coords[coords] = b;
         gene_frags += (uint32_t)n_in_bin[(size_t)b];
      }
      std::vector<int64_t> kept;
      std::vector<const char *> names;
      std::vector<double> fpkm, frac;
      for (int64_t j = 0; j < niso; ++j) {
         if (!batch.isoforms[(size_t)(j0 + j)].kept) continue;
         kept.push_back(j);
         names.push_back(tx_id[(size_t)l][(size_t)j].c_str());
         fpkm.push_back(batch.isoforms[(size_t)(j0 + j)].FPKM);
         frac.push_back(batch.isoforms[(size_t)(j0 + j)].frac);
      }
      if (kept.empty()) continue;
      for (const auto &kv : by_coords) {
         const int64_t b = kv.second, h = last_hit[(size_t)b];
         std::vector<double> prob;
         std::vector<uint32_t> sl, sr;
         for (int64_t j : kept) // the weights of the isoforms the bin's LAST fragment fits (:1556-1563)
            prob.push_back(((batch.compat[(size_t)(h * batch.compat_words + (j >> 5))] >> (j & 31)) & 1u)
                              ? batch.F[(size_t)(batch.f_off[(size_t)l] + (b - b0) * niso + j)]
                              : 0.0);
         for (const auto &c : kv.first) {
            sl.push_back(c.first);
            sr.push_back(c.second);
         }
         const int n =
            bin_gc.empty()
               ? sbgpu_format_context_row(buf.data(), (int)buf.size(), sample.c_str(), total_mapped, gene_id[(size_t)l].c_str(),
                                          gene_frags, (int)kept.size(), names.data(), fpkm.data(), prob.data(), frac.data(),
                                          (int)sl.size(), sl.data(), sr.data(), (uint32_t)n_in_bin[(size_t)b])
               : sbgpu_format_context_row_seq(buf.data(), (int)buf.size(), sample.c_str(), total_mapped, gene_id[(size_t)l].c_str(),
                                              gene_frags, (int)kept.size(), names.data(), fpkm.data(), prob.data(), frac.data(),
                                              (int)sl.size(), sl.data(), sr.data(), (uint32_t)n_in_bin[(size_t)b], bin_gc[(size_t)b],
                                              bin_entropy[(size_t)b], bin_flags[(size_t)b]);
         sbgpu::check(n, "sbgpu_format_context_row");
         ctxf.write(buf.data(), n);
      }
   }
   std::fprintf(stderr, "%lld loci, %lld hits, %lld bins, %d mapped reads\n", (long long)L, (long long)batch.n_hits(), (long long)n_bins,
                total_mapped);
   return 0;
}
