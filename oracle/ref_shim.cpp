/*
 * oracle/ref_shim.cpp -- TEST INFRASTRUCTURE ONLY.
 *
 * Our own thin extern-"C" wrapper around the REFERENCE's unmodified classes.
 * It contains no reference code: it #includes the reference's headers from
 * /root/reference/include at build time and is linked (oracle/Makefile, target
 * `ref`) against objects compiled straight from /root/reference/src/ *.cpp.
 * The result, oracle/_ref/libstrawberry_ref.so, is git-ignored; it is used to
 *   - pin oracle/em_oracle.c (tests/test_oracle_vs_ref.py),
 *   - generate tests/golden/ vectors (tools/make_goldens.py),
 *   - serve as bench.py's cpu_baseline of kind "reference".
 */
#include <limits>
#include <vector>
#include <cstdint>
#include <numeric>
#include <set>

#include "estimate.hpp" /* /root/reference/include/estimate.hpp:230-257 (EmSolver) */
#include <cassert>
#include <string>
#include "kmer.h"     /* /root/reference/include/kmer.h:14-135 (Kmer<string>) */

extern "C" {

/* EmSolver em; em.init(niso, n, alpha); if (ok) em.run();  -- exactly the call
 * sequence of src/estimate.cpp:305-308.
 * returns: bit0 = init() result, bit1 = run() result.                        */
int ref_em_locus(int nrow, int niso, const int32_t *count, const double *F,
                 double *theta_out)
{
   std::vector<int> n(count, count + nrow);
   std::vector<std::vector<double>> alpha((size_t)nrow, std::vector<double>((size_t)niso));
   for (int i = 0; i < nrow; ++i)
      for (int j = 0; j < niso; ++j) alpha[i][j] = F[(size_t)i * niso + j];
   EmSolver em;
   bool ok = em.init(niso, n, alpha);
   bool ran = false;
   if (ok) ran = em.run();
   for (int j = 0; j < niso; ++j) theta_out[j] = em._theta[j];
   return (ok ? 1 : 0) | (ran ? 2 : 0);
}

/* Batch form over the CSR-of-loci layout (include/sbgpu.h), loci [lo,hi). */
void ref_em_batch(int64_t lo, int64_t hi, const int64_t *row_off,
                  const int64_t *iso_off, const int64_t *f_off,
                  const int32_t *count, const double *F, double *theta_out,
                  int32_t *flags_out)
{
   for (int64_t l = lo; l < hi; ++l) {
      int nrow = (int)(row_off[l + 1] - row_off[l]);
      int niso = (int)(iso_off[l + 1] - iso_off[l]);
      int fl = ref_em_locus(nrow, niso, count + row_off[l], F + f_off[l],
                            theta_out + iso_off[l]);
      if (flags_out) flags_out[l] = fl;
   }
}

/* ExonBin::effective_len, include/isoform.h:419-516 (state-independent: it only
 * reads its arguments), called on a dummy one-segment bin.                   */
int ref_effective_len(int nseg, const uint32_t *seg_lens, int nimp,
                      const uint32_t *implicit_idx, int fl, int rl)
{
   std::set<std::pair<uint, uint>> coords;
   coords.insert(std::make_pair(1u, 2u));
   ExonBin eb(coords);
   std::vector<uint> sl(seg_lens, seg_lens + nseg);
   std::vector<uint> imp(implicit_idx, implicit_idx + nimp);
   return eb.effective_len(sl, imp, fl, rl);
}

/* InsertSize::emp_dist_pdf, src/read.cpp:274-297.  n_frag_lens==0 -> the
 * Gaussian object InsertSize(mean, sd); else the empirical object built from
 * the fragment-length sample (src/read.cpp:241-272).                        */
static InsertSize *make_insert(double mean, double sd, int n, const int32_t *frag_lens)
{
   if (n <= 0) return new InsertSize(mean, sd);
   return new InsertSize(std::vector<int>(frag_lens, frag_lens + n));
}

void ref_insert_pdf(double mean, double sd, int n_frag_lens, const int32_t *frag_lens,
                    int fl_lo, int fl_hi, double *pdf_out, double *mean_sd_off_out)
{
   InsertSize *is = make_insert(mean, sd, n_frag_lens, frag_lens);
   for (int fl = fl_lo; fl <= fl_hi; ++fl) pdf_out[fl - fl_lo] = is->emp_dist_pdf((uint)fl);
   if (mean_sd_off_out) {
      mean_sd_off_out[0] = is->_mean;
      mean_sd_off_out[1] = is->_sd;
      mean_sd_off_out[2] = is->_use_emp ? is->_start_offset : 0;
      mean_sd_off_out[3] = is->_use_emp ? is->_end_offset : 0;
   }
   delete is;
}

/* One (bin, isoform) weight: the loop of LocusContext::set_theory_bin_weight,
 * src/estimate.cpp:209-230, driven with the reference's own effective_len and
 * emp_dist_pdf (that member function is private, so its 10-line loop is
 * re-driven here; the end-to-end goldens of tools/make_goldens.py pin the
 * real one through the -f context TSV).                                      */
double ref_bin_weight(int nseg, const uint32_t *seg_lens, int nimp,
                      const uint32_t *implicit_idx, int iso_len, int rl,
                      double mean, double sd, int n_frag_lens, const int32_t *frag_lens)
{
   InsertSize *is = make_insert(mean, sd, n_frag_lens, frag_lens);
   std::set<std::pair<uint, uint>> coords;
   coords.insert(std::make_pair(1u, 2u));
   ExonBin eb(coords);
   std::vector<uint> sl(seg_lens, seg_lens + nseg);
   std::vector<uint> imp(implicit_idx, implicit_idx + nimp);
   int lmax = std::accumulate(sl.begin(), sl.end(), 0);
   int lmin = is->_use_emp ? is->_start_offset : rl;
   if (sl.size() > 2) lmin = std::max(lmin, std::accumulate(sl.begin() + 1, sl.end() - 1, 0));
   double weight = 0.0;
   for (int fl = lmin; fl <= lmax; ++fl) {
      double le_eff = eb.effective_len(sl, imp, fl, rl);
      weight += is->emp_dist_pdf(fl) * le_eff / (iso_len - fl + 1);
   }
   delete is;
   return weight;
}

/* ---- exon-bin assignment, integer part ---------------------------
 * Contig::is_compatible(read, isoform), src/contig.cpp:547-599, on Contigs built with the
 * reference's own public constructor (include/contig.h:165-182) from flat feature lists
 * (code 0 MATCH / 1 INTRON / 2 GAP, left, length).                                       */
static Contig make_contig(int n, const int32_t *code, const uint32_t *left, const int32_t *len, bool is_ref)
{
   std::vector<GenomicFeature> feats;
   for (int k = 0; k < n; ++k) feats.push_back(GenomicFeature((Match_t)code[k], left[k], len[k]));
   return Contig(0, 1, Strand_t::StrandPlus, 1.0, feats, is_ref);
}

int ref_is_compatible(int n_read, const int32_t *rcode, const uint32_t *rleft, const int32_t *rlen,
                      int n_iso, const int32_t *icode, const uint32_t *ileft, const int32_t *ilen)
{
   Contig read = make_contig(n_read, rcode, rleft, rlen, false);
   Contig iso = make_contig(n_iso, icode, ileft, ilen, true);
   return Contig::is_compatible(read, iso) ? 1 : 0;
}

/* Bin key of a read: which disjoint exon segments any of its MATCH blocks overlaps --
 * the double loop of LocusContext::overlap_exons (src/estimate.cpp:115-131, a member that
 * needs a whole LocusContext, so its loop is re-driven here) over the reference's
 * GenomicFeature::overlaps (src/contig.cpp:98-102).  key_out[k] = 1 if segment k is in. */
void ref_overlap_key(int n_read, const int32_t *rcode, const uint32_t *rleft, const int32_t *rlen,
                     int n_seg, const uint32_t *sleft, const uint32_t *sright, uint8_t *key_out)
{
   for (int k = 0; k < n_seg; ++k) {
      GenomicFeature seg(Match_t::S_MATCH, sleft[k], (int)(sright[k] - sleft[k] + 1));
      key_out[k] = 0;
      for (int f = 0; f < n_read; ++f) {
         if (rcode[f] != 0) continue;
         GenomicFeature rf(Match_t::S_MATCH, rleft[f], rlen[f]);
         if (GenomicFeature::overlaps(rf, seg)) key_out[k] = 1;
      }
   }
}

/* Contig::Contig(const PairedHit&), src/contig.cpp:216-267, on ReadHits built with the
 * reference's public constructor (include/read.hpp:89-100) from aligned blocks (an M/N CIGAR).
 * Returns the number of features (0: the reference marks the pair incompatible, ref_id -1). */
static ReadHitPtr make_readhit(int n, const uint32_t *bl, const uint32_t *br, uint32_t flag)
{
   std::vector<CigarOp> cig;
   for (int k = 0; k < n; ++k) {
      if (k && bl[k] != br[k - 1] + 1) cig.push_back(CigarOp(REF_SKIP, bl[k] - br[k - 1] - 1));
      else if (k) cig.push_back(CigarOp(INS, 1)); /* blocks that touch: an insertion in the read (M I M) */
      cig.push_back(CigarOp(MATCH, br[k] - bl[k] + 1));
   }
   GenomicInterval iv(0, bl[0], br[n - 1], Strand_t::StrandPlus);
   return ReadHitPtr(new ReadHit(1, "r", iv, cig, 0, 0, 0, 1, flag, 1.0, NULL));
}

int ref_pairedhit_features(int n_left, const uint32_t *ll, const uint32_t *lr, int n_right, const uint32_t *rl,
                           const uint32_t *rr, int32_t *code_out, uint32_t *left_out, uint32_t *right_out)
{
   ReadHitPtr L = n_left ? make_readhit(n_left, ll, lr, 99) : ReadHitPtr();
   ReadHitPtr R = n_right ? make_readhit(n_right, rl, rr, 147) : ReadHitPtr();
   PairedHit ph(L, R);
   Contig c(ph);
   if (c.ref_id() == -1) return 0;
   int n = 0;
   for (const auto &f : c._genomic_feats) {
      code_out[n] = (int32_t)f._match_op._code;
      left_out[n] = f.left();
      right_out[n] = f.right();
      ++n;
   }
   return n;
}

/* HitCluster::collapseAndFilterHits (src/alignments.cpp:658-703) on the reference's own HitCluster, filled the way
 * the reference fills it: every read goes through HitCluster::addOpenHit (:490-650: mate pairing, the cluster's
 * read spans), in coordinate order (left ends ascending, ties in input order), then the reference's own collapse
 * runs.  Pair p: left mate blocks [lo[p], lo[p+1]) of (ll, lr), right mate [ro[p], ro[p+1]) of (rl, rr) (one of
 * them may be empty: a singleton), NH tag nh[p].  Out, per unique hit in the reference's order: the input pair it
 * was made of (read id), its collapse mass; *cluster_mass = HitCluster::_weighted_mass.  Returns the number of
 * unique hits, -1 when a read was refused.                                                                      */
int ref_collapse_cluster(int n_pairs, const int64_t *lo, const uint32_t *ll, const uint32_t *lr, const int64_t *ro,
                         const uint32_t *rl, const uint32_t *rr, const int32_t *nh, int32_t *uniq_pair_out,
                         double *uniq_mass_out, double *cluster_mass_out)
{
   struct Rd {
      uint32_t pos;
      int pair, side;
   };
   std::vector<Rd> reads;
   for (int p = 0; p < n_pairs; ++p) {
      if (lo[p + 1] > lo[p]) reads.push_back({ll[lo[p]], p, 0});
      if (ro[p + 1] > ro[p]) reads.push_back({rl[ro[p]], p, 1});
   }
   std::stable_sort(reads.begin(), reads.end(), [](const Rd &a, const Rd &b) { return a.pos < b.pos; });
   auto make = [&](int p, int side) {
      const int64_t o = side ? ro[p] : lo[p], n = (side ? ro[p + 1] : lo[p + 1]) - o;
      const uint32_t *bl = (side ? rl : ll) + o, *br = (side ? rr : lr) + o;
      std::vector<CigarOp> cig;
      for (int64_t k = 0; k < n; ++k) {
         if (k) cig.push_back(CigarOp(REF_SKIP, bl[k] - br[k - 1] - 1));
         cig.push_back(CigarOp(MATCH, br[k] - bl[k] + 1));
      }
      const bool has_mate = side ? lo[p + 1] > lo[p] : ro[p + 1] > ro[p];
      const int partner_pos = has_mate ? (int)(side ? ll[lo[p]] : rl[ro[p]]) : 0;
      GenomicInterval iv(0, bl[0], br[n - 1], Strand_t::StrandPlus);
      return ReadHitPtr(new ReadHit((ReadID)(p + 1), "r", iv, cig, 0, partner_pos, 0, nh[p], side ? 147u : 99u, 1.0, NULL));
   };
   HitCluster hc;
   for (const Rd &r : reads)
      if (!hc.addOpenHit(make(r.pair, r.side), true, true)) return -1;
   const int n = hc.collapseAndFilterHits();
   int k = 0;
   for (const PairedHit &u : hc.uniq_hits()) {
      uniq_pair_out[k] = (int32_t)((u._left_read ? u.left_read_obj().read_id() : u.right_read_obj().read_id()) - 1);
      uniq_mass_out[k] = u.collapse_mass();
      ++k;
   }
   *cluster_mass_out = hc._weighted_mass;
   return n;
}

/* Mate pairing AND collapse on the reference's own HitCluster: the records go through HitCluster::addOpenHit
 * (src/alignments.cpp:490-650) exactly as given -- order, read ids, partner positions, strands, NH -- then
 * collapseAndFilterHits runs.  A record's name is its index, so the unique hits can name their mates.
 * flags: bit 0 reverse strand, bit 1 partner on another reference, bits 2-3 XS strand (0 unknown, 1 +, 2 -).
 * Out, per unique hit: left_rec / right_rec (record indices, -1: none), collapse mass; *cluster_mass; *n_hits = the
 * number of PairedHits before the collapse (HitCluster::size()).  Returns the number of unique hits.              */
int ref_cluster_from_records(int n_reads, const uint64_t *read_id, const int64_t *block_off, const uint32_t *bl, const uint32_t *br,
                             const uint32_t *partner_pos, const uint8_t *flags, const int32_t *nh, int32_t *left_rec,
                             int32_t *right_rec, double *uniq_mass, double *cluster_mass, int32_t *n_hits)
{
   HitCluster hc;
   for (int r = 0; r < n_reads; ++r) {
      const int64_t o = block_off[r], n = block_off[r + 1] - o;
      if (n <= 0) continue;
      std::vector<CigarOp> cig;
      for (int64_t k = 0; k < n; ++k) {
         if (k && bl[o + k] != br[o + k - 1] + 1) cig.push_back(CigarOp(REF_SKIP, bl[o + k] - br[o + k - 1] - 1));
         else if (k) cig.push_back(CigarOp(INS, 1)); /* blocks that touch: an insertion in the read */
         cig.push_back(CigarOp(MATCH, br[o + k] - bl[o + k] + 1));
      }
      const int xs = (flags[r] >> 2) & 3;
      const Strand_t strand = xs == 1 ? Strand_t::StrandPlus : (xs == 2 ? Strand_t::StrandMinus : Strand_t::StrandUnknown);
      GenomicInterval iv(0, bl[o], br[o + n - 1], strand);
      const RefID partner_ref = (flags[r] & 2u) ? 1 : 0;
      ReadHitPtr hit(new ReadHit((ReadID)read_id[r], std::to_string(r), iv, cig, partner_ref, (int)partner_pos[r], 0, nh[r],
                                 (flags[r] & 1u) ? 16u : 0u, 1.0, NULL));
      hc.addOpenHit(hit, true, true);
   }
   *n_hits = hc.size();
   if (hc.size() == 0) {
      *cluster_mass = 0.0;
      return 0;
   }
   const int n = hc.collapseAndFilterHits();
   int k = 0;
   for (const PairedHit &u : hc.uniq_hits()) {
      left_rec[k] = u._left_read ? std::stoi(u.left_read_obj().read_name()) : -1;
      right_rec[k] = u._right_read ? std::stoi(u.right_read_obj().read_name()) : -1;
      uniq_mass[k] = u.collapse_mass();
      ++k;
   }
   *cluster_mass = hc._weighted_mass;
   return n;
}

/* The six per-bin sequence statistics of the `-f` table, by the reference's own templates
 * exactly as src/alignments.cpp:1623-1629 calls them.  out6 = gc, entropy, 4 flags (0/1).
 * The caller keeps to len > 40: the reference's live asserts abort below that.           */
void ref_kmer_stats(const char *seq, int len, double *out6)
{
   std::string s(seq, seq + len);
   out6[0] = Kmer<std::string>::GCRatio(s.begin(), s.end());
   out6[1] = Kmer<std::string>::Entropy(s, 6);
   out6[2] = Kmer<std::string>::HighGCStrech(s.begin(), s.end(), 20, 0.8);
   out6[3] = Kmer<std::string>::HighGCStrech(s.begin(), s.end(), 20, 0.9);
   out6[4] = Kmer<std::string>::HighGCStrech(s.begin(), s.end(), 40, 0.8);
   out6[5] = Kmer<std::string>::HighGCStrech(s.begin(), s.end(), 40, 0.9);
}

/* BAMHitFactory::getHitFromBuf (src/read.cpp:480-715) on every record of a BAM file, through the reference's own
 * BAMHitFactory (samopen + bam_read1 of the vendored samtools 0.1.19), with the reference's option globals set as the
 * command line would set them (-j / -J, --allow-multimapped-hits, --fr / --rf; src/Strawberry.cpp:129-169).  Per record, in file
 * order: accepted (getHitFromBuf's return value), and for the accepted ones the ReadHit's fields (sam_flag: the bits the
 * class shows -- 16 reverse, 64 first, 128 second -- and bit 31 = is_singleton()).  The CIGAR the
 * ReadHit keeps (H and P ops are not in it) is flattened into cig_type / cig_len at cig_off[r]; the features
 * readhit_2_genomicFeats (src/contig.cpp:12-53) makes of it into feat_* at feat_off[r].  Returns the number of
 * records read, or -1 when an array is too small.  *single_end_out = the global SINGLE_END_EXP afterwards.      */
int ref_bam_decode(const char *bam_path, int min_intron, int max_intron, int unique_only, int library, int64_t cap_records,
                   int64_t cap_ops, uint8_t *accepted, uint64_t *read_id, int32_t *ref_id, uint32_t *left, uint32_t *right,
                   uint8_t *strand, uint8_t *partner_same_ref, uint32_t *partner_pos, int32_t *num_mismatch, int32_t *num_hits,
                   uint32_t *sam_flag, double *mass, int32_t *read_len, int64_t *cig_off, uint8_t *cig_type, uint32_t *cig_len,
                   int64_t *feat_off, uint8_t *feat_code, uint32_t *feat_left, uint32_t *feat_right, int32_t *single_end_out)
{
   kMinIntronLength = min_intron;
   kMaxIntronLength = max_intron;
   use_only_unique_hits = unique_only != 0;
   fr_strand = library == 1;
   rf_strand = library == 2;
   SINGLE_END_EXP = true;
   ReadTable rt;
   RefSeqTable st(true);
   BAMHitFactory hf(bam_path, rt, st);
   hf.inspect_header(); /* fills the reference-name table from the @SQ lines, as Sample's constructor does */
   int64_t n = 0, nc = 0, nf = 0;
   const char *buf = NULL;
   size_t buf_size = 0;
   cig_off[0] = feat_off[0] = 0;
   while (hf.nextRecord(buf, buf_size)) {
      if (n >= cap_records) return -1;
      ReadHit rh;
      const bool ok = hf.getHitFromBuf(buf, rh);
      accepted[n] = ok ? 1 : 0;
      if (ok) {
         read_id[n] = (uint64_t)rh.read_id();
         ref_id[n] = (int32_t)rh.ref_id();
         left[n] = rh.left();
         right[n] = rh.right();
         strand[n] = rh.strand() == Strand_t::StrandPlus ? 1 : (rh.strand() == Strand_t::StrandMinus ? 2 : 0);
         partner_same_ref[n] = rh.partner_ref_id() == rh.ref_id() ? 1 : 0;
         partner_pos[n] = (uint32_t)rh.partner_pos();
         num_mismatch[n] = rh.num_mismatch();
         num_hits[n] = rh.numHits();
         sam_flag[n] = (rh.reverseCompl() ? 16u : 0u) | (rh.is_first() ? 64u : 0u) | (rh.is_second() ? 128u : 0u) | (rh.is_singleton() ? 1u << 31 : 0u);
         mass[n] = rh.mass();
         read_len[n] = (int32_t)rh.read_len();
         for (const CigarOp &c : rh.cigar()) {
            if (nc >= cap_ops) return -1;
            cig_type[nc] = (uint8_t)c._type;
            cig_len[nc] = c._length;
            ++nc;
         }
         std::vector<GenomicFeature> feats;
         readhit_2_genomicFeats(rh, feats);
         for (const GenomicFeature &f : feats) {
            if (nf >= cap_ops) return -1;
            feat_code[nf] = (uint8_t)f._match_op._code;
            feat_left[nf] = f.left();
            feat_right[nf] = f.right();
            ++nf;
         }
      }
      ++n;
      cig_off[n] = nc;
      feat_off[n] = nf;
   }
   *single_end_out = SINGLE_END_EXP ? 1 : 0;
   return (int)n;
}

} /* extern "C" */
