// oracle/em_dump_shim.cpp -- TEST INFRASTRUCTURE ONLY (nothing under strawberry_amd/ refers to it).
//
// A tap on the seam of the reference program: `make -C oracle ref` links oracle/_ref/strawberry_dump from the
// reference's UNMODIFIED objects, with EmSolver::init / EmSolver::run (/root/reference/src/estimate.cpp:366-488, called at
// :305-308) weakened in one copy of estimate.o (the copy the program runs) and RENAMED to sbref_em_init / sbref_em_run in
// a second, all-weak copy (objcopy --redefine-sym; it contributes nothing but those two bodies).  The definitions below
// write the arguments of every init() -- the exact (n, alpha) a locus hands to the EM, as binary doubles -- and the
// solution of every run() to the file named by $SB_EM_DUMP, and forward to the reference's own bodies: the program's
// outputs are the reference binary's, byte for byte.  tools/make_e2e_golden.py uses it to capture the EM inputs of loci
// whose isoforms are ASSEMBLED contigs (BASELINE config 4; alignments.cpp:1658 assembleSample, :1091-1101
// reset_refmRNAs) into tests/golden/em_c4_assembled.npz.
//
// This file contains no reference code: it includes the reference's header and calls the reference's compiled functions.
#include "estimate.hpp" // the reference's: /root/reference/include/estimate.hpp:225-257

#include <cstdint>
#include <cstdio>
#include <cstdlib>

// the reference's own bodies under their objcopy names (member functions: `this` is the first argument)
extern "C" bool sbref_em_init(EmSolver *, int, const std::vector<int> &, const std::vector<std::vector<double>> &);
extern "C" bool sbref_em_run(EmSolver *);

namespace {
FILE *dump_file()
{
   static FILE *f = [] {
      const char *path = std::getenv("SB_EM_DUMP");
      return path ? std::fopen(path, "wb") : nullptr;
   }();
   return f;
}
void put_i32(FILE *f, int32_t v) { std::fwrite(&v, 4, 1, f); }
} // namespace

// record: 'I', niso, nrow, n[nrow] (int32), alpha[nrow][niso] (double), init's return value (int32)
bool EmSolver::init(const int num_iso, const std::vector<int> &count, const std::vector<std::vector<double>> &model)
{
   const bool ok = sbref_em_init(this, num_iso, count, model);
   if (FILE *f = dump_file()) {
      std::fputc('I', f);
      put_i32(f, num_iso);
      put_i32(f, (int32_t)count.size());
      for (int v : count) put_i32(f, v);
      for (const auto &row : model) std::fwrite(row.data(), 8, row.size(), f);
      put_i32(f, ok ? 1 : 0);
      std::fflush(f);
   }
   return ok;
}

// record: 'R', run's return value (int32), niso, theta[niso] (double)
bool EmSolver::run()
{
   const bool ok = sbref_em_run(this);
   if (FILE *f = dump_file()) {
      std::fputc('R', f);
      put_i32(f, ok ? 1 : 0);
      put_i32(f, (int32_t)_theta.size());
      std::fwrite(_theta.data(), 8, _theta.size(), f);
      std::fflush(f);
   }
   return ok;
}
