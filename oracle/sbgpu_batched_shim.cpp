// oracle/sbgpu_batched_shim.cpp -- TEST INFRASTRUCTURE ONLY (nothing under strawberry_amd/ refers to it).
//
// The BATCHED drop-in under the reference's own driver ("procSample must be restructured into collect ->
// one batched solve -> epilogue").  `make -C oracle ref` links oracle/_ref/strawberry_sbgpu_batched from the reference's
// UNMODIFIED objects -- Strawberry.cpp's main, BAM decode, clustering, assembly, LocusContext, the output code -- with
// three functions replaced: Sample::procSample (/root/reference/src/alignments.cpp:1736-1834) is weakened in a copy of
// alignments.o and EmSolver::init / EmSolver::run (src/estimate.cpp:366-488) in a copy of estimate.o (objcopy
// --weaken-symbol), and the definitions below take their place.
//
//   collect    the clusters are walked once with the reference's own classes (nextClusterRefDemand, finalizeCluster,
//              LocusContext's constructor: bins and weights are the reference's); LocusContext::estimate_abundances()
//              is called on every locus while the EmSolver below is in RECORD mode: init() appends the locus' (n, alpha)
//              to ONE sbgpu::EmBatch and returns false, which makes estimate_abundances return before it has touched
//              anything (src/estimate.cpp:305-314);
//   solve      ONE sbgpu_em_batch call for all loci of the sample (a chromosome at a time when -b asks for the genome's
//              sequence, which the reference loads per chromosome);
//   epilogue   REPLAY mode: estimate_abundances() again, locus by locus in cluster order -- init() / run() now hand the
//              device's status and theta of that locus to the reference's own epilogue (theta log, FPKM, Frac, the
//              kMinIsoformFrac filter, src/estimate.cpp:310-355) -- then what quantifyCluster does with a solved locus
//              (alignments.cpp:1524-1545), the TPM pass and print2gtf (:1821-1834).
//
// tests/test_reference_driver_gpu.py compares its files with the reference binary's, byte for byte;
// tools/dropin_timing.py times it beside strawberry_ref and the per-locus strawberry_sbgpu (profiles/r04_dropin.txt).
//
// The loop below restates the control flow of the function it replaces -- there is no other way to move the solve out
// of it; everything it calls is the reference's compiled code, declared by the reference's headers at build time.
#include "alignments.h" // the reference's: /root/reference/include/alignments.h:178-290 (Sample)
#include "estimate.hpp" // /root/reference/include/estimate.hpp:14-257 (LocusContext, EmSolver)

#include <chrono>
#include <climits>
#include <cstdlib>
#include <memory>

#include "sbgpu_host.hpp"

#ifdef SB_BATCHED_SOLVE_WITH_REFERENCE
// A second build of this file (oracle/_ref/strawberry_batched_refem) solves the collected batch with the REFERENCE's own
// EmSolver bodies (a renamed copy of them, as in em_dump_shim.cpp) instead of the device: it has no GPU in it, so the CPU
// suite can check that the restructured loop -- collect, solve, epilogue in cluster order -- reproduces the reference
// binary's files by itself, here, before the device version runs on the GPU box.
extern "C" bool sbref_em_init(EmSolver *, int, const std::vector<int> &, const std::vector<std::vector<double>> &);
extern "C" bool sbref_em_run(EmSolver *);
#endif

namespace {

#ifdef SB_BATCHED_SOLVE_WITH_REFERENCE
void solve_batch(sbgpu::EmBatch &b)
{
   const int64_t n = b.size();
   b.theta.assign((size_t)b.iso_off.back() + 1, 0.0);
   b.status.assign((size_t)n + 1, 0);
   b.iters.assign((size_t)n + 1, 0);
   for (int64_t l = 0; l < n; ++l) {
      const int64_t r0 = b.row_off[(size_t)l], r1 = b.row_off[(size_t)l + 1], j0 = b.iso_off[(size_t)l];
      const int niso = (int)(b.iso_off[(size_t)l + 1] - j0);
      std::vector<int> cnt(b.count.begin() + r0, b.count.begin() + r1);
      std::vector<std::vector<double>> alpha((size_t)(r1 - r0), std::vector<double>((size_t)niso));
      for (int64_t i = 0; i < r1 - r0; ++i)
         for (int j = 0; j < niso; ++j) alpha[(size_t)i][(size_t)j] = b.F[(size_t)(b.f_off[(size_t)l] + i * niso + j)];
      EmSolver em;
      const bool ok = sbref_em_init(&em, niso, cnt, alpha);
      const bool ran = ok && sbref_em_run(&em);
      b.status[(size_t)l] = !ok ? SBGPU_EM_INIT_EMPTY : !ran ? SBGPU_EM_DENOM_ZERO : SBGPU_EM_OK;
      for (int j = 0; j < niso; ++j) b.theta[(size_t)(j0 + j)] = em._theta[(size_t)j];
   }
}
#else
const sbgpu::Context &device_context()
{
   static const sbgpu::Context ctx(0); // throws (no CPU fallback) when there is no gfx950 device
   return ctx;
}
void solve_batch(sbgpu::EmBatch &b) { b.solve(device_context()); } // ONE sbgpu_em_batch call
#endif

// what travels between the two passes
struct Seam {
   bool record = true;
   sbgpu::EmBatch batch;
   int64_t cursor = 0;      // REPLAY: the locus whose init() comes next
   int64_t solved = -1;     // REPLAY: the locus run() belongs to
   std::vector<double> theta0_scratch;
} seam;

} // namespace

// ---------------------------------------------------------------------------------------------------- the seam
// estimate.hpp:241-243.  RECORD: collect and decline.  REPLAY: the reference's bool, theta_0 in _theta.
bool EmSolver::init(const int num_iso, const std::vector<int> &count, const std::vector<std::vector<double>> &model)
{
   if (seam.record) {
      seam.batch.add(num_iso, count, model);
      return false;
   }
   const int64_t l = seam.cursor++;
   seam.solved = l;
   double total = 0.0;
   for (int c : count) total += (double)c;
   _theta.assign((size_t)num_iso, total / num_iso); // theta_0 (estimate.cpp:374-375): what survives a zero denominator
   return seam.batch.init_ok(l);
}

// estimate.hpp:250.  false: a zero denominator (estimate.cpp:451-453), _theta untouched.
bool EmSolver::run()
{
   const int64_t l = seam.solved;
   if (!seam.batch.run_ok(l)) return false;
   const int64_t j0 = seam.batch.iso_off[(size_t)l], j1 = seam.batch.iso_off[(size_t)l + 1];
   _theta.assign(seam.batch.theta.begin() + j0, seam.batch.theta.begin() + j1);
   return true;
}

// ---------------------------------------------------------------------------------------------------- the driver loop
// replaces /root/reference/src/alignments.cpp:1736-1834
void Sample::procSample(FILE *pfile, FILE *plogfile, FILE *fragfile)
{
   _hit_factory->reset();
   std::vector<Isoform> isoforms;
   isoforms.reserve(1024);
   reset_refmRNAs();
   const RefSeqTable &ref_t = _hit_factory->_ref_table;
   if (fragfile != NULL) { // the -f table's header (alignments.cpp:1746-1752)
      std::vector<std::string> header = {"sample", "sample_frag_count", "gene_id", "gene_frag_count", "transcripts", "FPKMs",
                                         "conditional_probabilities", "class_probabilities", "path_symbol", "path_count",
                                         "path_gc_content", "path_hexmer_entropy", "gc_stretch_0.8_20", "gc_stretch_0.9_20",
                                         "gc_stretch_0.8_40", "gc_stretch_0.9_40"};
      pretty_print(fragfile, header, "\t");
   }
   struct Pending {
      std::shared_ptr<HitCluster> cluster;
      std::unique_ptr<LocusContext> est;
   };
   std::vector<Pending> pending;

   // SBGPU_DROPIN_TIMING=1: where the wall time of this function goes (tools/dropin_timing.py reads the line)
   using clk = std::chrono::steady_clock;
   auto secs = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double>(b - a).count(); };
   double t_solve = 0.0, t_epilogue = 0.0;
   int64_t n_solved = 0, n_calls = 0;
   const clk::time_point t_begin = clk::now();

   // solve what has been collected and run the reference's epilogue over it, in cluster order
   auto flush = [&]() {
      if (pending.empty()) return;
      const clk::time_point t0 = clk::now();
      solve_batch(seam.batch);
      const clk::time_point t1 = clk::now();
      t_solve += secs(t0, t1);
      n_solved += seam.batch.size();
      ++n_calls;
      seam.record = false;
      seam.cursor = 0;
      for (Pending &p : pending) {
         const bool success = p.est->estimate_abundances(); // the reference's epilogue over the device's theta
         if (success) {                                      // alignments.cpp:1524-1536
            const std::vector<Isoform> &iso = p.est->transcripts();
            isoforms.insert(isoforms.end(), iso.begin(), iso.end());
            std::cerr << ref_t.ref_real_name(p.cluster->ref_id()) << "\t" << p.cluster->left() << "\t" << p.cluster->right()
                      << " finishes abundances estimation" << std::endl;
            if (fragfile != NULL) printContext(*p.est, p.cluster, _fasta_getter, fragfile);
         }
      }
      pending.clear();
      seam.batch = sbgpu::EmBatch();
      seam.record = true;
      t_epilogue += secs(t1, clk::now());
   };

   int current_ref_id = INT_MAX;
   while (true) {
      std::shared_ptr<HitCluster> cluster(new HitCluster());
      if (-1 == nextClusterRefDemand(*cluster)) break;
      if (cluster->ref_id() == -1) continue;
      if (current_ref_id != cluster->ref_id()) {
         if (BIAS_CORRECTION) { // printContext reads the chromosome's sequence: finish the one before first
            flush();
            load_chrom_fasta(cluster->ref_id());
         }
         current_ref_id = cluster->ref_id();
      }
      finalizeCluster(cluster, true);
      Pending p;
      p.cluster = cluster;
      p.est.reset(new LocusContext(*this, plogfile, cluster, cluster->ref_mRNAs()));
      (void)p.est->estimate_abundances(); // RECORD: the locus' (n, alpha) joins the batch; nothing else happens
      pending.push_back(std::move(p));
   }
   flush();

   // alignments.cpp:1821-1834
   double total_fpkm = 0.0;
   for (const auto &iso : isoforms) total_fpkm += iso._FPKM;
   for (auto &iso : isoforms) {
      iso._TPM = 1e6 * iso._FPKM / total_fpkm;
      iso._TPM_s = std::to_string(iso._TPM);
   }
   for (const auto &iso : isoforms)
      iso._contig.print2gtf(pfile, _hit_factory->_ref_table, iso._FPKM_s, iso._frac_s, iso._TPM_s, iso._gene_str, iso._isoform_str,
                            iso._ref_gene_id, iso._ref_gene_name);
   const char *timing = std::getenv("SBGPU_DROPIN_TIMING");
   if (timing && timing[0] == '1') {
      const double total = secs(t_begin, clk::now());
      std::fprintf(stderr, "sbgpu_batched procSample: total %.3f s = collect (BAM pass 2, clustering, LocusContext: bins + weights) %.3f s + "
                           "solve %.3f s (%lld loci in %lld sbgpu_em_batch call%s) + epilogue and output %.3f s\n",
                   total, total - t_solve - t_epilogue, t_solve, (long long)n_solved, (long long)n_calls, n_calls == 1 ? "" : "s", t_epilogue);
   }
}
