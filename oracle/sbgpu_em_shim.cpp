// oracle/sbgpu_em_shim.cpp -- TEST INFRASTRUCTURE ONLY (nothing under strawberry_amd/ refers to it).
//
// The drop-in, proven with the reference's own driver: `make -C oracle ref` links oracle/_ref/strawberry_sbgpu from
// the reference's UNMODIFIED objects -- Strawberry.cpp's main, BAM decode, clustering, LocusContext, the output code --
// with the two functions of the seam, EmSolver::init and EmSolver::run (/root/reference/src/estimate.cpp:366-488,
// called at :305-308), weakened in a copy of estimate.o (objcopy --weaken-symbol) and replaced by the definitions
// below: the reference's own class, declared by its own header at build time, with its two member functions
// forwarding to sbgpu::EmSolver (include/sbgpu_host.hpp) -- i.e. to the HIP kernels behind the C ABI.
// tests/test_reference_driver_gpu.py runs that binary on the toy BAMs and compares its output files with the
// reference binary's, byte for byte.
//
// This file contains no reference code: it includes the reference's header (as oracle/ref_shim.cpp does) and
// stores the solution in the private members the class already has.
#include "estimate.hpp" // the reference's: /root/reference/include/estimate.hpp:225-257

#include <chrono>
#include <cstdio>
#include <cstdlib>

#include "sbgpu_host.hpp"

namespace {
// SBGPU_DROPIN_TIMING=1: the wall time spent inside the per-locus device calls, reported when the program ends
// (tools/dropin_timing.py puts it beside the batched drop-in's one call)
struct SeamClock {
   double seconds = 0.0;
   long long calls = 0;
   ~SeamClock()
   {
      const char *t = std::getenv("SBGPU_DROPIN_TIMING");
      if (t && t[0] == '1')
         std::fprintf(stderr, "sbgpu per-locus seam: %lld EmSolver::init calls (one plan + upload + launch + synchronise each), %.3f s inside them\n", calls, seconds);
   }
} seam_clock;
const sbgpu::Context &device_context()
{
   static const sbgpu::Context ctx(0); // throws (no CPU fallback) when there is no gfx950 device
   return ctx;
}
} // namespace

// estimate.hpp:241-243.  The device solves the locus here; _theta holds theta_0 until run(), as in the reference.
bool EmSolver::init(const int num_iso, const std::vector<int> &count, const std::vector<std::vector<double>> &model)
{
   const auto t0 = std::chrono::steady_clock::now();
   sbgpu::EmSolver solver(device_context());
   const bool ok = solver.init(num_iso, count, model);
   seam_clock.seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
   ++seam_clock.calls;
   _theta = solver._theta;                       // theta_0 (estimate.cpp:374-375)
   const bool ran = ok && solver.run();
   _theta_after_zero = solver._theta;            // the solution (or theta_0 again after a zero denominator)
   _u.assign(1, ran ? 1 : 0);
   return ok;
}

// estimate.hpp:250.  false: a zero denominator (estimate.cpp:451-453), _theta untouched.
bool EmSolver::run()
{
   if (_u.empty() || _u[0] == 0) return false;
   _theta = _theta_after_zero;
   return true;
}
