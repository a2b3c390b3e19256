// examples/em_solver_kat.cpp -- sbgpu::EmSolver (include/sbgpu_host.hpp) used exactly like the
// reference's EmSolver at its call site (src/estimate.cpp:305-313), on the known-answer cases the
// survey captured from the reference.  Prints one line per case:
//   <name> init=<0|1> run=<0|1> theta...        (%.12g)
#include <cstdio>
#include <vector>

#include "sbgpu_host.hpp"

int main()
{
   struct Case {
      const char *name;
      int niso;
      std::vector<int> n;
      std::vector<std::vector<double>> F;
   };
   const std::vector<Case> cases = {
      {"toy", 2, {100, 50, 30}, {{.002, .001}, {.003, 0}, {0, .004}}},
      {"denom_zero", 2, {0, 5}, {{.1, 0}, {0, .1}}},
      {"all_dropped", 2, {3, 4}, {{1e-5, 1e-6}, {0, 1e-5}}},
      {"zero_col", 3, {10, 20}, {{.2, .1, 0}, {.05, .3, 0}}},
      {"row_dropped", 2, {7, 10, 20}, {{1e-6, 1e-6}, {.2, .1}, {.05, .3}}},
      {"single_iso", 1, {10, 20}, {{.2}, {.05}}},
      {"single_row", 3, {9}, {{.2, .1, .4}}},
   };
   try {
      sbgpu::Context ctx(0);
      for (const Case &c : cases) {
         bool success, ran = false;
         sbgpu::EmSolver em(ctx);
         success = em.init(c.niso, c.n, c.F);
         if (success) ran = em.run();
         std::printf("%s init=%d run=%d", c.name, (int)success, (int)ran);
         for (double t : em._theta) std::printf(" %.12g", t);
         std::printf("\n");
      }
   } catch (const std::exception &e) {
      std::fprintf(stderr, "error: %s\n", e.what());
      return 1;
   }
   return 0;
}
