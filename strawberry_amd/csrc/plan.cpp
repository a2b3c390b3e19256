// strawberry_amd/csrc/plan.cpp -- see plan.h
#include "plan.h"

#include <algorithm>
#include <map>
#include <tuple>

#include "../../include/sbgpu.h"

namespace sb {

static int pow2ceil(int64_t x)
{
   int p = 1;
   while (p < x) p <<= 1;
   return p;
}
static int ilog2i(int x)
{
   int l = 0;
   while ((1 << l) < x) ++l;
   return l;
}

int build_host_plan(int64_t n_loci, const int64_t *row_off, const int64_t *iso_off,
                    const int64_t *f_off, int n_cu, const PlanTuning &tune, HostPlan *out,
                    const char **err)
{
   *err = "";
   if (n_loci < 0 || (n_loci > 0 && (!row_off || !iso_off || !f_off))) {
      *err = "plan: null offsets";
      return SBGPU_EINVAL;
   }
   if (n_loci > INT32_MAX) {
      *err = "plan: more than 2^31-1 loci in one batch";
      return SBGPU_EINVAL;
   }
   HostPlan &p = *out;
   p = HostPlan();
   p.n_loci = n_loci;
   if (n_loci == 0) return SBGPU_OK;
   if (row_off[0] != 0 || iso_off[0] != 0 || f_off[0] != 0) {
      *err = "plan: offsets must start at 0";
      return SBGPU_EINVAL;
   }

   // pass 1: validate, decide the wave-kind tile height from the load
   int64_t wave_lanes_base = 0; // lanes the wave kind would occupy with the base tile (rh 2)
   for (int64_t l = 0; l < n_loci; ++l) {
      const int64_t nrow = row_off[l + 1] - row_off[l];
      const int64_t niso = iso_off[l + 1] - iso_off[l];
      if (nrow < 0 || niso < 1 || f_off[l + 1] - f_off[l] != nrow * niso || nrow > (1 << 28)) {
         *err = "plan: malformed locus (need niso >= 1, nrow >= 0, f_off step == nrow*niso)";
         return SBGPU_EINVAL;
      }
      if (niso > kMaxStreamIso) {
         *err = "plan: a locus has more than 512 isoforms";
         return SBGPU_ESHAPE;
      }
      p.algorithmic_bytes += nrow * niso * 8 + nrow * 4 + niso * 8 + 24;
      if (niso <= kMaxTileC) {
         int CPL, CL;
         layout_for(niso, &CPL, &CL);
         const int R = tile_rows(CPL, 2);
         const int64_t lanes = (int64_t)pow2ceil(std::max<int64_t>(1, (nrow + R - 1) / R)) * CL;
         if (lanes <= 64) wave_lanes_base += lanes;
      }
   }
   const int64_t simd_lanes = (int64_t)n_cu * 4 * 64;
   int wave_rmult = tune.wave_rmult;
   if (wave_rmult != 1 && wave_rmult != 2 && wave_rmult != 4) {
      // Base tile up to 8x the lanes the two waves per SIMD of its register budget hold (the
      // dispatcher back-fills as short waves retire: measured best on C2 and C3), double tile
      // (half the lanes per locus) beyond that.  The half tile (SBGPU_WAVE_RMULT=1)
      // doubles the lanes for a ~10 % shorter iteration: measured no better on C2, worse on C3.
      wave_rmult = (wave_lanes_base <= 16 * simd_lanes) ? 2 : 4;
   }

   std::map<std::tuple<int, int, int, int>, SizeClass> by_key;
   for (int64_t l = 0; l < n_loci; ++l) {
      const int64_t nrow = row_off[l + 1] - row_off[l];
      const int64_t niso = iso_off[l + 1] - iso_off[l];
      SizeClass k;
      k.kind = kStream;
      if (niso <= kMaxTileC) {
         layout_for(niso, &k.CPL, &k.CL);
         k.layout = layout_id(k.CPL, k.CL);
         // wave kind: smallest power-of-two group that holds the rows
         bool placed = false;
         {
            const int R = tile_rows(k.CPL, wave_rmult);
            const int64_t lanes = (int64_t)pow2ceil(std::max<int64_t>(1, (nrow + R - 1) / R)) * k.CL;
            if (lanes <= 64) {
               k.kind = (wave_rmult == 1) ? kWaveH : (wave_rmult == 2 ? kWave1 : kWave2);
               k.rmult = wave_rmult;
               k.R = R;
               k.G = (int)lanes;
               k.lbG = ilog2i(k.G);
               placed = true;
            }
         }
         // block kinds: the whole 256-lane workgroup is the group
         for (int tall = tune.light_block ? 0 : 1; tall < 2 && !placed; ++tall) {
            const int rm = tall ? kBlockTallRh : kBlockRh;
            if ((int64_t)(kBlockThreads / k.CL) * tile_rows(k.CPL, rm) >= nrow) {
               k.kind = tall ? kBlockTall : kBlock;
               k.rmult = rm;
               k.R = tile_rows(k.CPL, rm);
               k.G = kBlockThreads;
               k.lbG = 6;
               placed = true;
            }
         }
      }
      if (k.kind == kStream) {
         k.layout = k.CPL = k.CL = k.R = k.G = k.lbG = 0;
         k.rmult = 1;
         ++p.n_stream_loci;
      }
      SizeClass &sc = by_key[std::make_tuple(k.kind, k.layout, k.rmult, k.G)];
      if (sc.loci.empty()) {
         std::vector<int32_t> keep;
         sc = k;
      }
      sc.loci.push_back((int32_t)l);
      sc.work += (k.kind == kStream) ? nrow * niso : (int64_t)(k.G / std::max(1, k.CL)) * k.R * k.CPL * k.CL;
   }
   p.n_rows = row_off[n_loci];
   p.n_iso = iso_off[n_loci];
   p.n_elem = f_off[n_loci];

   // grids: one wave per batch of 64/G loci (one workgroup per locus in the block kind).
   // More waves than the chip holds is fine: the dispatcher starts the next one as soon as
   // a slot frees, and a wave that lives only as long as its own batch cannot chain several
   // 1000-iteration batches back to back.  Only beyond `max_waves` do the grids shrink and
   // waves pull further batches through the cursor.
   const int64_t max_waves = tune.max_waves > 0 ? tune.max_waves : (int64_t)1 << 20;
   int64_t waves_wanted = 0;
   for (auto &kv : by_key) {
      SizeClass &sc = kv.second;
      // heaviest loci first: they are the likeliest stragglers (LPT order)
      std::stable_sort(sc.loci.begin(), sc.loci.end(), [&](int32_t x, int32_t y) {
         const int64_t wx = (row_off[x + 1] - row_off[x]) * (iso_off[x + 1] - iso_off[x]);
         const int64_t wy = (row_off[y + 1] - row_off[y]) * (iso_off[y + 1] - iso_off[y]);
         return wx > wy;
      });
      const int64_t n = (int64_t)sc.loci.size();
      if (sc.kind == kWaveH || sc.kind == kWave1 || sc.kind == kWave2) {
         sc.block_threads = 64;
         sc.n_blocks = (int)((n + (64 / sc.G) - 1) / (64 / sc.G));
      } else if (sc.kind == kStream) {
         sc.block_threads = 1024;
         sc.n_blocks = (int)n;
      } else {
         sc.block_threads = sc.G;
         sc.n_blocks = (int)n;
      }
      waves_wanted += (int64_t)sc.n_blocks * (sc.block_threads / 64);
   }
   if (waves_wanted > max_waves) {
      const double f = (double)max_waves / (double)waves_wanted;
      for (auto &kv : by_key) {
         SizeClass &sc = kv.second;
         sc.n_blocks = std::max(1, (int)(sc.n_blocks * f + 0.5));
      }
   }
   for (auto &kv : by_key) p.classes.push_back(std::move(kv.second));
   std::stable_sort(p.classes.begin(), p.classes.end(), [](const SizeClass &x, const SizeClass &y) {
      if (x.kind != y.kind) return x.kind < y.kind;
      // Lowest block indices are dispatched first.  The makespan is set by the loci that run
      // all 1000 iterations, and an iteration costs more the more lanes (and column lanes) a
      // locus spans: those classes go first, the many short narrow ones fill in behind.
      const int cx = x.lbG + x.CPL + 4 * (x.layout / 8), cy = y.lbG + y.CPL + 4 * (y.layout / 8);
      if (cx != cy) return cx > cy;
      return x.work > y.work;
   });
   return SBGPU_OK;
}

} // namespace sb
