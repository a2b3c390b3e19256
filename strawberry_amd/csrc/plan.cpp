// strawberry_amd/csrc/plan.cpp -- see plan.h
#include "plan.h"

#include <algorithm>
#include <map>
#include <tuple>

#include "../../include/sbgpu.h"

namespace sb {

static int pow2ceil(int x)
{
   int p = 1;
   while (p < x) p <<= 1;
   return p;
}

int build_host_plan(int64_t n_loci, const int64_t *row_off, const int64_t *iso_off,
                    const int64_t *f_off, int n_cu, HostPlan *out, const char **err)
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
   std::map<std::tuple<int, int, int, int>, SizeClass> by_key;
   for (int64_t l = 0; l < n_loci; ++l) {
      const int64_t nrow = row_off[l + 1] - row_off[l];
      const int64_t niso = iso_off[l + 1] - iso_off[l];
      if (nrow < 0 || niso < 1 || f_off[l + 1] - f_off[l] != nrow * niso || nrow > INT32_MAX / 2) {
         *err = "plan: malformed locus (need niso >= 1, nrow >= 0, f_off step == nrow*niso)";
         return SBGPU_EINVAL;
      }
      if (niso > kMaxStreamIso) {
         *err = "plan: a locus has more than 512 isoforms";
         return SBGPU_ESHAPE;
      }
      p.algorithmic_bytes += nrow * niso * 8 + nrow * 4 + niso * 8 + 24;
      int kind = kStream, CPL = 0, CL = 0, R = 0, G = 0;
      if (niso <= kMaxTileC) {
         const int C = std::max(2, pow2ceil((int)niso));
         CPL = std::min(C, kMaxCPL);
         CL = C / CPL;
         R = kTileElems / CPL;
         // row lanes needed with R rows each; the group is CL x (row lanes)
         int64_t gr = std::max<int64_t>(1, (nrow + R - 1) / R);
         int64_t lanes = (int64_t)pow2ceil((int)std::min<int64_t>(gr, 1 << 20)) * CL;
         if (lanes <= 64) {
            kind = kTile;
            G = (int)lanes;
         } else {
            // one workgroup per locus: 256 or 512 lanes, R or 2R rows per row lane
            for (int mult = 1; mult <= 2 && kind != kTile; ++mult) {
               for (int gg = 256; gg <= 512 && kind != kTile; gg *= 2) {
                  if ((int64_t)(gg / CL) * R * mult >= nrow) {
                     kind = kTile;
                     G = gg;
                     R = R * mult;
                  }
               }
            }
         }
      }
      if (kind == kStream) {
         CPL = CL = R = G = 0;
         ++p.n_stream_loci;
      }
      SizeClass &sc = by_key[std::make_tuple(kind, CPL * 16 + CL, R, G)];
      sc.kind = kind;
      sc.CPL = CPL;
      sc.CL = CL;
      sc.R = R;
      sc.G = G;
      sc.loci.push_back((int32_t)l);
      sc.work += (kind == kTile) ? (int64_t)(G / CL) * R * CPL * CL : nrow * niso;
   }
   p.n_rows = row_off[n_loci];
   p.n_iso = iso_off[n_loci];
   p.n_elem = f_off[n_loci];

   // resident-wave budget: 4 SIMDs per CU, a few waves each
   const int64_t wave_budget = (int64_t)n_cu * 4 * 4;
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
      if (sc.kind == kTile) {
         if (sc.G <= 64) {
            sc.block_threads = 64;
            sc.n_blocks = (int)((n + (64 / sc.G) - 1) / (64 / sc.G));
            waves_wanted += sc.n_blocks;
         } else {
            sc.block_threads = sc.G;
            sc.n_blocks = (int)n;
            waves_wanted += (int64_t)sc.n_blocks * (sc.G / 64);
         }
      } else {
         sc.block_threads = 1024;
         sc.n_blocks = (int)n;
         waves_wanted += (int64_t)sc.n_blocks * 16;
      }
   }
   if (waves_wanted > wave_budget) {
      // more groups than the chip can hold: shrink every grid proportionally and
      // let the groups pull loci through the cursor until the list runs dry
      const double f = (double)wave_budget / (double)waves_wanted;
      for (auto &kv : by_key) {
         SizeClass &sc = kv.second;
         sc.n_blocks = std::max(1, (int)(sc.n_blocks * f + 0.5));
      }
   }
   for (auto &kv : by_key) p.classes.push_back(std::move(kv.second));
   std::sort(p.classes.begin(), p.classes.end(), [](const SizeClass &x, const SizeClass &y) {
      // workgroup-per-locus classes first (longest iterations), then by total work
      const int gx = x.kind == kStream ? 1 << 20 : x.G, gy = y.kind == kStream ? 1 << 20 : y.G;
      if ((gx > 64) != (gy > 64)) return gx > 64;
      return x.work > y.work;
   });
   return SBGPU_OK;
}

} // namespace sb
