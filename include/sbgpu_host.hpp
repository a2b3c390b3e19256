/*
 * include/sbgpu_host.hpp -- the host side above the C ABI, in the reference's own language.
 *
 * ruolin/strawberry is C++14; this header (C++14, header-only, depends only on sbgpu.h) is what
 * its driver would include.  It mirrors the reference's call surface for the path:
 *
 *   sbgpu::EmSolver            include/estimate.hpp:230-257  (init / run / _theta), one locus
 *   sbgpu::EmBatch             the same for many loci in ONE device call: collect -> solve
 *   sbgpu::LocusBatch          LocusContext's constructor + estimate_abundances for many loci
 *                              (include/estimate.hpp:60-103, src/estimate.cpp:279-355):
 *                              fragments -> exon bins -> bin weights -> EM -> FPKM / Frac
 *   sbgpu::finalize_tpm        Sample::procSample's tail, src/alignments.cpp:1821-1829
 *
 * Every number comes from libsbgpu.so (HIP kernels); the only arithmetic here is the reference's
 * own host epilogue (theta -> FPKM / Frac / TPM, a dozen flops per isoform), kept as host code on
 * purpose: INTEGRATION.md section 2.  Errors are exceptions carrying sbgpu_last_error().
 */
#ifndef SBGPU_HOST_HPP_
#define SBGPU_HOST_HPP_

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <ctime>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "sbgpu.h"

namespace sbgpu {

struct Error : std::runtime_error {
   int code;
   Error(int c, const std::string &what) : std::runtime_error(what + ": " + sbgpu_last_error()), code(c) {}
};
inline void check(int rc, const char *what)
{
   if (rc < 0) throw Error(rc, what);
}

/* One per process and GPU. */
class Context {
   sbgpu_ctx_t *h_ = nullptr;

public:
   explicit Context(int device = 0) { check(sbgpu_init(device, &h_), "sbgpu_init"); }
   ~Context()
   {
      if (h_) sbgpu_finalize(h_);
   }
   Context(const Context &) = delete;
   Context &operator=(const Context &) = delete;
   sbgpu_ctx_t *get() const { return h_; }
};

/* Many loci, one solve.  add() takes exactly EmSolver::init's arguments. */
class EmBatch {
public:
   std::vector<int64_t> row_off{0}, iso_off{0}, f_off{0};
   std::vector<int32_t> count;
   std::vector<double> F;
   std::vector<double> theta;   /* after solve(): em._theta of every locus, concatenated */
   std::vector<int32_t> status; /* SBGPU_EM_* per locus                                   */
   std::vector<int32_t> iters;

   int64_t size() const { return (int64_t)row_off.size() - 1; }

   int64_t add(int num_iso, const std::vector<int> &n, const std::vector<std::vector<double>> &alpha)
   {
      for (size_t i = 0; i < n.size(); ++i) {
         count.push_back(n[i]);
         for (int j = 0; j < num_iso; ++j) F.push_back(alpha[i][(size_t)j]);
      }
      row_off.push_back((int64_t)count.size());
      iso_off.push_back(iso_off.back() + num_iso);
      f_off.push_back((int64_t)F.size());
      return size() - 1;
   }

   void solve(const Context &ctx)
   {
      const int64_t n = size();
      theta.assign((size_t)iso_off.back() + 1, 0.0);
      status.assign((size_t)n + 1, 0);
      iters.assign((size_t)n + 1, 0);
      sbgpu_batch_t b = {n, row_off.data(), iso_off.data(), f_off.data(), count.data(), F.data()};
      check(sbgpu_em_batch(ctx.get(), &b, theta.data(), status.data(), iters.data()), "sbgpu_em_batch");
   }
   /* the two bools of the reference, src/estimate.cpp:307-308 */
   bool init_ok(int64_t l) const { return status[(size_t)l] != SBGPU_EM_INIT_EMPTY; }
   bool run_ok(int64_t l) const { return status[(size_t)l] == SBGPU_EM_OK || status[(size_t)l] == SBGPU_EM_MAXITER; }
};

/* include/estimate.hpp:230-257 for a single locus (tests; a caller with one locus at hand).
 * The device call happens in init(): it returns the reference's bool, _theta holds theta_0 until
 * run() -- which returns the reference's second bool -- installs the solution.               */
class EmSolver {
   const Context &ctx_;
   std::vector<double> solved_;
   int32_t status_ = SBGPU_EM_INIT_EMPTY;

public:
   std::vector<double> _theta;
   explicit EmSolver(const Context &ctx) : ctx_(ctx) {}
   bool init(const int num_iso, const std::vector<int> &count, const std::vector<std::vector<double>> &model)
   {
      EmBatch b;
      b.add(num_iso, count, model);
      b.solve(ctx_);
      status_ = b.status[0];
      solved_.assign(b.theta.begin(), b.theta.begin() + num_iso);
      double total = 0.0;
      for (int c : count) total += (double)c;
      _theta.assign((size_t)num_iso, total / num_iso); /* src/estimate.cpp:374-375 */
      return status_ != SBGPU_EM_INIT_EMPTY;
   }
   bool run()
   {
      if (status_ == SBGPU_EM_INIT_EMPTY || status_ == SBGPU_EM_DENOM_ZERO) return false; /* theta_0 stays */
      _theta = solved_;
      return true;
   }
};

struct InsertSize { /* include/read.hpp:176-192 */
   double mean = 200.0, sd = 80.0;
   bool use_emp = false;
   int start_offset = 0, end_offset = 0, total_reads = 0;
   std::vector<double> emp_dist;
   InsertSize() = default;
   InsertSize(double m, double s) : mean(m), sd(s) {}
   /* InsertSize(const vector<int> frag_lens), src/read.cpp:238-262 (+ mean_and_sd_insert_size, :14-20) */
   explicit InsertSize(const std::vector<int> &frag_lens) : use_emp(true)
   {
      total_reads = (int)frag_lens.size();
      if (total_reads < 1) throw std::runtime_error("Not enough reads");
      double sum = 0.0, sq = 0.0;
      int lo = frag_lens[0], hi = frag_lens[0];
      for (int v : frag_lens) {
         sum += v;
         lo = v < lo ? v : lo;
         hi = v > hi ? v : hi;
      }
      for (int v : frag_lens) sq += (double)v * v;
      mean = sum / frag_lens.size();
      sd = std::sqrt(sq / frag_lens.size() - mean * mean);
      start_offset = lo;
      end_offset = hi;
      emp_dist.assign((size_t)(hi - lo + 1), 0.0);
      for (int v : frag_lens) emp_dist[(size_t)(v - lo)] += 1.0;
   }
};

struct Isoform { /* what the epilogue fills: include/isoform.h:40-58 */
   int length = 0;
   double theta = 0.0, FPKM = 0.0, frac = 0.0, TPM = 0.0;
   std::string FPKM_s = "nan", frac_s = "nan", TPM_s = "nan";
   bool kept = true;
};

/* Fragments of many loci -> abundances.  Fill with add_locus / add_hit in the reference's order
 * (isoforms as LocusContext::_transcripts, hits as HitCluster::uniq_hits()), then quantify().   */
class LocusBatch {
public:
   /* annotation */
   std::vector<int64_t> iso_off{0}, exon_off{0}, seg_off;
   std::vector<uint32_t> exon_left, exon_right, seg_left, seg_right;
   /* hits */
   std::vector<int32_t> hit_locus;
   std::vector<int64_t> feat_off{0};
   std::vector<uint8_t> feat_code;
   std::vector<uint32_t> feat_left, feat_right;
   std::vector<float> hit_mass;
   /* results */
   std::vector<int64_t> row_off, f_off, hit_bin;
   std::vector<int32_t> count, status, iters;
   std::vector<uint32_t> compat, key, bin_key;
   std::vector<double> F, theta;
   std::vector<Isoform> isoforms; /* all loci, concatenated like iso_off */
   int32_t compat_words = 1, key_words = 1;

   int64_t n_loci() const { return (int64_t)iso_off.size() - 1; }
   int64_t n_hits() const { return (int64_t)hit_locus.size(); }

   /* exons: the S_MATCH features of each transcript's Contig::_genomic_feats, closed coordinates */
   int64_t add_locus(const std::vector<std::vector<std::pair<uint32_t, uint32_t>>> &transcripts)
   {
      for (const auto &t : transcripts) {
         for (const auto &e : t) {
            exon_left.push_back(e.first);
            exon_right.push_back(e.second);
         }
         exon_off.push_back((int64_t)exon_left.size());
      }
      iso_off.push_back((int64_t)exon_off.size() - 1);
      return n_loci() - 1;
   }
   /* one unique hit: its Contig features (code 0 MATCH / 1 INTRON / 2 GAP) and (float) collapse mass */
   void add_hit(int32_t locus, int n_feat, const uint8_t *code, const uint32_t *left, const uint32_t *right, float mass)
   {
      hit_locus.push_back(locus);
      feat_code.insert(feat_code.end(), code, code + n_feat);
      feat_left.insert(feat_left.end(), left, left + n_feat);
      feat_right.insert(feat_right.end(), right, right + n_feat);
      feat_off.push_back((int64_t)feat_code.size());
      hit_mass.push_back(mass);
   }
   /* a read pair given by its mates' features: Contig(PairedHit), src/contig.cpp:216-267.
    * Returns false when the reference rejects the pair (it then only counts towards the mapped total). */
   bool add_pair(int32_t locus, const std::vector<uint8_t> &lc, const std::vector<uint32_t> &ll, const std::vector<uint32_t> &lr,
                 const std::vector<uint8_t> &rc, const std::vector<uint32_t> &rl, const std::vector<uint32_t> &rr, float mass)
   {
      const size_t cap = lc.size() + rc.size() + 1;
      std::vector<uint8_t> c(cap);
      std::vector<uint32_t> l(cap), r(cap);
      const int n = sbgpu_hit_features((int)lc.size(), lc.data(), ll.data(), lr.data(), (int)rc.size(), rc.data(), rl.data(),
                                       rr.data(), c.data(), l.data(), r.data());
      check(n, "sbgpu_hit_features");
      if (n == 0) return false;
      add_hit(locus, n, c.data(), l.data(), r.data(), mass);
      return true;
   }

   /* All aligned read pairs of the batch at once: HitCluster::collapseAndFilterHits + Contig(PairedHit)
    * (sbgpu_collapse_pairs_host).  Replaces the hits added so far; returns the reference's
    * _total_mapped_reads (sum over loci of the int-truncated cluster mass, src/alignments.cpp:1372).   */
   int set_hits_from_pairs(const sbgpu_pairs_t &pairs)
   {
      sbgpu_uniq_t *u = nullptr;
      check(sbgpu_collapse_pairs_host(n_loci(), &pairs, &u), "sbgpu_collapse_pairs_host");
      int64_t info[8];
      sbgpu_uniq_info(u, info);
      hit_locus.assign((size_t)info[0], 0);
      feat_off.assign((size_t)info[0] + 1, 0);
      feat_code.assign((size_t)info[1], 0);
      feat_left.assign((size_t)info[1], 0);
      feat_right.assign((size_t)info[1], 0);
      hit_mass.assign((size_t)info[0], 0.0f);
      const int rc = sbgpu_uniq_export(u, hit_locus.data(), feat_off.data(), feat_code.data(), feat_left.data(), feat_right.data(),
                                       hit_mass.data(), nullptr);
      sbgpu_uniq_destroy(u);
      check(rc, "sbgpu_uniq_export");
      return (int)info[4];
   }

   sbgpu_annotation_t annotation() const
   {
      sbgpu_annotation_t a = {n_loci(), iso_off.data(), exon_off.data(), exon_left.data(), exon_right.data(),
                              seg_off.data(), seg_left.data(), seg_right.data()};
      return a;
   }
   sbgpu_hits_t hits() const
   {
      sbgpu_hits_t h = {n_hits(), hit_locus.data(), feat_off.data(), feat_code.data(), feat_left.data(), feat_right.data()};
      return h;
   }

   InsertSize insert; /* the distribution quantify() used (the empirical one when it was asked to build it) */

   /* LocusContext ctor + estimate_abundances for every locus.  total_mapped_reads, min_isoform_frac
    * etc. are the globals the reference reads (sbgpu_abundance_params_t).  ins_in == nullptr: no -i was
    * given, the empirical insert-size distribution is built from the hits first (Sample::fragLenDist,
    * src/alignments.cpp:1363-1407 + Strawberry.cpp:345-355).                                       */
   void quantify(const Context &ctx, const InsertSize *ins_in, int read_len, const sbgpu_abundance_params_t &par,
                 bool long_read = false)
   {
      const int64_t nl = n_loci();
      /* _exon_segs, include/estimate.hpp:80-91 */
      seg_off.assign((size_t)nl + 1, 0);
      int64_t ns = sbgpu_segments_host(nl, iso_off.data(), exon_off.data(), exon_left.data(), exon_right.data(), seg_off.data(),
                                       nullptr, nullptr, 0);
      check((int)(ns < 0 ? ns : 0), "sbgpu_segments_host");
      seg_left.assign((size_t)ns + 1, 0);
      seg_right.assign((size_t)ns + 1, 0);
      sbgpu_segments_host(nl, iso_off.data(), exon_off.data(), exon_left.data(), exon_right.data(), seg_off.data(),
                          seg_left.data(), seg_right.data(), ns);
      int64_t max_iso = 1, max_seg = 1;
      for (int64_t l = 0; l < nl; ++l) {
         max_iso = std::max(max_iso, iso_off[(size_t)l + 1] - iso_off[(size_t)l]);
         max_seg = std::max(max_seg, seg_off[(size_t)l + 1] - seg_off[(size_t)l]);
      }
      compat_words = (int32_t)((max_iso + 31) / 32);
      key_words = (int32_t)((max_seg + 31) / 32);
      /* assign_exon_bin + set_theory_bin_weight + EmSolver for all loci: one call, everything between the
       * hits and theta stays on the device (sbgpu_quantify_host) */
      const int64_t nh = n_hits();
      const int64_t n_iso = iso_off[(size_t)nl];
      compat.assign((size_t)nh * compat_words + 1, 0);
      theta.assign((size_t)n_iso + 1, 0.0);
      status.assign((size_t)nl + 1, 0);
      iters.assign((size_t)nl + 1, 0);
      sbgpu_annotation_t an = annotation();
      sbgpu_hits_t ht = hits();
      sbgpu_insert_t si, used;
      if (ins_in) {
         insert = *ins_in;
         si.mean = insert.mean;
         si.sd = insert.sd;
         si.use_emp = insert.use_emp;
         si.start_offset = insert.start_offset;
         si.end_offset = insert.end_offset;
         si.total_reads = insert.total_reads;
         si.emp_hist = insert.emp_dist.empty() ? nullptr : insert.emp_dist.data();
         si.read_len = read_len;
         si.long_read = long_read;
      }
      sbgpu_bins_t *bins = nullptr;
      check(sbgpu_quantify_host(ctx.get(), &an, &ht, hit_mass.data(), ins_in ? &si : nullptr, read_len, long_read, theta.data(),
                                status.data(), iters.data(), compat.data(), &used, &bins),
            "sbgpu_quantify_host");
      if (!ins_in) { /* the empirical law the library built (Sample::fragLenDist + InsertSize(frag_lens)) */
         insert = InsertSize();
         insert.mean = used.mean;
         insert.sd = used.sd;
         insert.use_emp = true;
         insert.start_offset = used.start_offset;
         insert.end_offset = used.end_offset;
         insert.total_reads = used.total_reads;
         insert.emp_dist.assign(used.emp_hist, used.emp_hist + (used.end_offset - used.start_offset + 1));
      }
      int64_t info[8];
      sbgpu_bins_info(bins, info);
      const int64_t n_bins = info[2], n_elem = info[3];
      row_off.assign((size_t)nl + 1, 0);
      f_off.assign((size_t)nl + 1, 0);
      count.assign((size_t)n_bins + 1, 0);
      std::vector<int32_t> iso_len((size_t)n_iso + 1, 0);
      bin_key.assign((size_t)n_bins * key_words + 1, 0);
      hit_bin.assign((size_t)nh + 1, -1);
      F.assign((size_t)n_elem + 1, 0.0);
      int rc = sbgpu_bins_export(bins, row_off.data(), nullptr, f_off.data(), count.data(), iso_len.data(), bin_key.data(), nullptr,
                                 hit_bin.data(), nullptr, nullptr, nullptr, nullptr, nullptr);
      if (rc == SBGPU_OK) rc = sbgpu_bins_export_weights(bins, F.data());
      sbgpu_bins_destroy(bins);
      check(rc, "sbgpu_bins_export");
      /* the reference's own epilogue, src/estimate.cpp:310-355 */
      isoforms.assign((size_t)n_iso, Isoform());
      for (int64_t l = 0; l < nl; ++l) {
         const int64_t j0 = iso_off[(size_t)l], j1 = iso_off[(size_t)l + 1];
         for (int64_t j = j0; j < j1; ++j) isoforms[(size_t)j].length = iso_len[(size_t)j];
         if (status[(size_t)l] == SBGPU_EM_INIT_EMPTY) { /* estimate_abundances() == false: the locus is dropped */
            for (int64_t j = j0; j < j1; ++j) isoforms[(size_t)j].kept = false;
            continue;
         }
         double sum_fpkm = 0.0;
         for (int64_t j = j0; j < j1; ++j) {
            Isoform &t = isoforms[(size_t)j];
            t.theta = theta[(size_t)j];
            double kb;
            if (par.effective_len_norm) {
               kb = t.length - par.insert_mean;
               if (kb < 0) {
                  t.FPKM_s = "NA";
                  continue;
               }
               kb = 1e3 / kb;
            } else {
               kb = 1e3 / t.length;
            }
            const double rpm = 1e6 / par.total_mapped_reads;
            t.FPKM = t.theta * rpm * kb;
            sum_fpkm += t.FPKM;
            t.FPKM_s = std::to_string(t.FPKM);
         }
         for (int64_t j = j0; j < j1; ++j) {
            Isoform &t = isoforms[(size_t)j];
            if (t.FPKM_s == "NA") {
               t.frac_s = "NA";
               continue;
            }
            t.frac = t.FPKM / sum_fpkm;
            t.frac_s = std::to_string(t.frac);
         }
         if (par.filter_by_expression)
            for (int64_t j = j0; j < j1; ++j)
               if (isoforms[(size_t)j].frac < par.min_isoform_frac) isoforms[(size_t)j].kept = false;
      }
   }
};

/* The collective of a multi-GPU run (one process per GPU, loci sharded over the ranks): the two cross-locus
 * sums of the reference, `_total_mapped_reads` (src/alignments.cpp:1372) and the FPKM total (:1821-1824).
 * Rank 0 makes the RCCL id and leaves it in `id_file` (written under a temporary name and renamed, so a reader
 * never sees half of it); the other ranks wait for the file.  world == 1 needs no file and no RCCL.
 * A file left behind by an earlier launch must not be taken for this one's: the file starts with `nonce` -- a value
 * the launcher gives to ALL ranks of one launch (job id, start time; `--comm-nonce` of examples/quantify_fragments) --
 * and readers wait for a file that carries theirs; rank 0 also removes a stale file before it writes and removes its
 * own once every rank has joined (sbgpu_comm_init is collective: it returns when all have read the id).  With the
 * default nonce 0 the path must not exist when the launch begins.                                              */
class Comm {
 public:
   Comm(const Context &ctx, int rank, int world, const std::string &id_file = std::string(), uint64_t nonce = 0) : h_(nullptr)
   {
      uint8_t id[SBGPU_COMM_ID_BYTES] = {0};
      if (world > 1) {
         if (id_file.empty()) throw Error(SBGPU_EINVAL, "sbgpu::Comm: a world of several ranks needs an id file");
         if (rank == 0) {
            std::remove(id_file.c_str()); /* whatever an earlier launch left */
            check(sbgpu_comm_unique_id(id), "sbgpu_comm_unique_id");
            const std::string tmp = id_file + ".tmp";
            FILE *f = std::fopen(tmp.c_str(), "wb");
            if (!f || std::fwrite(&nonce, 1, sizeof(nonce), f) != sizeof(nonce) || std::fwrite(id, 1, sizeof(id), f) != sizeof(id) ||
                std::fclose(f) != 0 || std::rename(tmp.c_str(), id_file.c_str()) != 0)
               throw Error(SBGPU_EINVAL, "sbgpu::Comm: cannot write " + id_file);
         } else {
            for (int tries = 0;; ++tries) {
               FILE *f = std::fopen(id_file.c_str(), "rb");
               if (f) {
                  uint64_t theirs = 0;
                  const size_t n0 = std::fread(&theirs, 1, sizeof(theirs), f);
                  const size_t n = std::fread(id, 1, sizeof(id), f);
                  std::fclose(f);
                  if (n0 == sizeof(theirs) && n == sizeof(id) && theirs == nonce) break; /* (another launch's file: keep waiting) */
               }
               if (tries > 6000) throw Error(SBGPU_ERCCL, "sbgpu::Comm: no id of this launch in " + id_file + " after 60 s");
               struct timespec ts = {0, 10 * 1000 * 1000};
               nanosleep(&ts, nullptr);
            }
         }
      }
      check(sbgpu_comm_init(ctx.get(), rank, world, world > 1 ? id : nullptr, &h_), "sbgpu_comm_init");
      if (world > 1 && rank == 0) std::remove(id_file.c_str()); /* every rank has joined: nothing is left for the next launch to trip over */
   }
   ~Comm() { sbgpu_comm_destroy(h_); }
   Comm(const Comm &) = delete;
   Comm &operator=(const Comm &) = delete;
   double allreduce_sum(double x) const
   {
      check(sbgpu_allreduce_sum_f64_host(h_, &x, 1), "sbgpu_allreduce_sum_f64_host");
      return x;
   }
   int64_t allreduce_sum(int64_t x) const
   {
      check(sbgpu_allreduce_sum_i64_host(h_, &x, 1), "sbgpu_allreduce_sum_i64_host");
      return x;
   }
   sbgpu_comm_t *get() const { return h_; }

 private:
   sbgpu_comm_t *h_;
};

/* Sample::procSample's tail (src/alignments.cpp:1821-1829) over every isoform that is written.
 * With loci sharded over ranks, all-reduce `total_fpkm` between the two loops.                 */
inline double sum_fpkm(const std::vector<Isoform> &isoforms)
{
   double total = 0.0;
   for (const Isoform &t : isoforms)
      if (t.kept) total += t.FPKM;
   return total;
}
inline void finalize_tpm(std::vector<Isoform> &isoforms, double total_fpkm)
{
   for (Isoform &t : isoforms) {
      if (!t.kept) continue;
      t.TPM = 1e6 * t.FPKM / total_fpkm;
      t.TPM_s = std::to_string(t.TPM);
   }
}

/* The six sequence columns of the `-f` table under `-b genome.fa` (src/alignments.cpp:1622-1636) for every bin of
 * a quantified batch: ExonBin::bin_dnaseq + Kmer<string>::GCRatio / Entropy / HighGCStrech, one kernel launch.
 * `chrom_seq` holds the chromosome's bases as load_chrom_fasta keeps them (base 1 first).                      */
struct BinSequenceStats {
   std::vector<double> gc, entropy; /* per bin, batch order                                     */
   std::vector<uint8_t> flags;      /* bit 0..3 = stretch (20, 0.8), (20, 0.9), (40, 0.8), (40, 0.9) */
};
inline BinSequenceStats bin_sequence_stats(const Context &ctx, const LocusBatch &batch, const std::string &chrom_seq)
{
   const int64_t n_loci = (int64_t)batch.row_off.size() - 1, n_bins = batch.row_off.back();
   std::vector<int64_t> off(1, 0);
   std::vector<uint32_t> sl, sr;
   for (int64_t l = 0; l < n_loci; ++l) {
      const int64_t s0 = batch.seg_off[(size_t)l], nseg = batch.seg_off[(size_t)l + 1] - s0;
      for (int64_t b = batch.row_off[(size_t)l]; b < batch.row_off[(size_t)l + 1]; ++b) {
         for (int64_t s = 0; s < nseg; ++s)
            if ((batch.bin_key[(size_t)(b * batch.key_words + (s >> 5))] >> (s & 31)) & 1u) {
               sl.push_back(batch.seg_left[(size_t)(s0 + s)]);
               sr.push_back(batch.seg_right[(size_t)(s0 + s)]);
            }
         off.push_back((int64_t)sl.size());
      }
   }
   BinSequenceStats out;
   out.gc.resize((size_t)n_bins);
   out.entropy.resize((size_t)n_bins);
   out.flags.resize((size_t)n_bins);
   check(sbgpu_binseq_host(ctx.get(), (const uint8_t *)chrom_seq.data(), 1, (int64_t)chrom_seq.size(), n_bins, off.data(), sl.data(),
                           sr.data(), out.gc.data(), out.entropy.data(), out.flags.data()),
         "sbgpu_binseq_host");
   return out;
}

} // namespace sbgpu
#endif /* SBGPU_HOST_HPP_ */
