// strawberry_amd/csrc/plan.cpp -- see plan.h
#include "plan.h"

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <thread>
#include <map>
#include <tuple>

#include "../../include/sbgpu.h"

namespace sb {

// How many iterations a locus is LIKELY to run, from its shape alone -- used only to order work (results never
// depend on it).  The EM converges slowly when a locus has about as many bins as isoforms (the likelihood is
// nearly flat along some direction of theta) and the slower the more isoforms it has; with many more bins than
// isoforms it converges in a few dozen steps.  Numbers: mean iteration counts of the reference's EmSolver on
// the C3 batch by nrow / niso, relative to the peak at nrow ~ niso, times (niso + 18) (tools/probe_iteration_shape.py:
// the loci that run all 1000 iterations are 97 % within the top 10 % of this score).
static int64_t predicted_iterations(int64_t nrow, int64_t niso)
{
   if (nrow <= 0 || niso <= 0) return 0;
   const double r = (double)nrow / (double)niso;
   const double w = r < 0.5 ? 0.10 : r < 0.75 ? 0.60 : r < 1.25 ? 1.00 : r < 1.5 ? 0.80 : r < 2 ? 0.55 : r < 3 ? 0.35 : r < 5 ? 0.27
                  : r < 10 ? 0.20 : 0.17;
   // What orders the classes is the LARGEST count among a class's loci, and loci of few isoforms with nrow ~ niso reach
   // the cap far more often than their mean suggests: hence the constant beside niso.  Measured on C3 (ms per step,
   // tools/probe_order.sh): + 0 1.40, + 6 1.38, + 12 1.36, + 18 1.32, + 24 1.32, + 30 1.34, + 45 1.42, + 150 1.62.
   const double it = 80.0 * w * (double)((niso < 24 ? niso : 24) + 18);
   return (int64_t)(it < 1000.0 ? it : 1000.0);
}


static int pow2ceil(int64_t x)
{
   int p = 1;
   while (p < x) p <<= 1;
   return p;
}
// host threads for the per-locus passes (SBGPU_HOST_THREADS, default min(16, hardware threads))
static unsigned plan_threads(int64_t n_items)
{
   unsigned nt = std::thread::hardware_concurrency();
   if (nt > 16) nt = 16;
   if (const char *e = std::getenv("SBGPU_HOST_THREADS")) nt = (unsigned)std::atoi(e);
   if (nt < 1) nt = 1;
   if (nt > 64) nt = 64;
   if ((int64_t)nt * 8192 > n_items) nt = (unsigned)std::max<int64_t>(1, n_items / 8192); // not worth a thread below 8192 loci
   return nt;
}
// f(begin, end, t) over [0, n) split into nt contiguous ranges
template <class F>
static void parallel_ranges(int64_t n, unsigned nt, F f)
{
   if (nt <= 1) {
      f((int64_t)0, n, 0u);
      return;
   }
   std::vector<std::thread> pool;
   for (unsigned t = 0; t < nt; ++t) pool.emplace_back([=]() { f(n * t / nt, n * (t + 1) / nt, t); });
   for (auto &th : pool) th.join();
}

static int ilog2i(int x)
{
   int l = 0;
   while ((1 << l) < x) ++l;
   return l;
}

// Modelled latency (cycles, one wave alone on its SIMD) of one EM iteration in a lane-rich layout, from the
// instruction costs tools/microbench.hip measures on MI355X (profiles/r02_microbench.txt): ~6 cycles per fp64 or DPP
// instruction, 62 per row for the division (v_rcp_f64 alone is 18) with its zero test, 18 per value and
// butterfly step, 19 for the matrix-pipe sum over lane bits 4-5, 32 for a lone permlane32 step.
static double lat_cycles(int cpl, int lb_cl, int r, int lb_gr)
{
   const double row_reduce = lb_gr == 0 ? 0.0 : lb_gr == 1 ? 32.0 : 18.0 * (lb_gr - 2) + 19.0;
   return 6.0 * cpl + 12.0 * r * cpl + 18.0 * r * lb_cl + 62.0 * r + cpl * row_reduce + 18.0 * cpl + 21.0 * lb_cl + 60.0;
}

bool lat_layout_for(int64_t nrow, int64_t niso, double lambda, LatLayout *out)
{
   bool found = false;
   double best = 0.0;
   for (int lb_cl = 0; lb_cl <= 4; ++lb_cl) {
      const int cl = 1 << lb_cl;
      const int64_t cpl = (niso + cl - 1) / cl;
      if (cpl > 4 || cpl < 1) continue;
      if (lb_cl > 0 && (niso + cl / 2 - 1) / (cl / 2) <= 1) continue; // half the column lanes already hold it with one column each
      for (int r : kLatRows) {
         if (r * cpl > kLatMaxTileElems) continue;
         // a locus gets the whole wave: 64 / cl row lanes (lanes are what the later phases have to spare, and a
         // compile-time lane count keeps the reductions free of branches)
         const int gr = 64 / cl;
         if ((std::max<int64_t>(nrow, 1) + r - 1) / r > gr) continue;
         const int lb_gr = ilog2i(gr);
         const double cyc = lat_cycles((int)cpl, lb_cl, r, lb_gr);
         const double score = cyc * (1.0 + lambda * (double)(gr * cl) / 64.0);
         if (!found || score < best) {
            found = true;
            best = score;
            out->cpl = (int)cpl, out->lb_cl = lb_cl, out->r = r, out->lbG = lb_gr + lb_cl, out->cycles = cyc;
         }
      }
   }
   return found;
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
   const unsigned nt = plan_threads(n_loci);
   int64_t wave_lanes_base = 0; // lanes the wave kind would occupy with the base tile (rh 2)
   {
      std::vector<int64_t> lanes_t(nt, 0), bytes_t(nt, 0);
      std::vector<int> bad_t(nt, 0);
      parallel_ranges(n_loci, nt, [&](int64_t b, int64_t e, unsigned t) {
         int64_t lanes_sum = 0, bytes = 0;
         int bad = 0;
         for (int64_t l = b; l < e; ++l) {
            const int64_t nrow = row_off[l + 1] - row_off[l];
            const int64_t niso = iso_off[l + 1] - iso_off[l];
            if (nrow < 0 || niso < 1 || f_off[l + 1] - f_off[l] != nrow * niso || nrow > (1 << 28)) {
               bad |= 1;
               continue;
            }
            if (niso > kMaxStreamIso) {
               bad |= 2;
               continue;
            }
            bytes += nrow * niso * 8 + nrow * 4 + niso * 8 + 24;
            if (niso <= kMaxTileC) {
               int CPL, CL;
               layout_for(niso, &CPL, &CL);
               const int R = tile_rows(CPL, 2);
               const int64_t lanes = (int64_t)pow2ceil(std::max<int64_t>(1, (nrow + R - 1) / R)) * CL;
               if (lanes <= 64) lanes_sum += lanes;
            }
         }
         lanes_t[t] = lanes_sum;
         bytes_t[t] = bytes;
         bad_t[t] = bad;
      });
      int bad = 0;
      for (unsigned t = 0; t < nt; ++t) {
         wave_lanes_base += lanes_t[t];
         p.algorithmic_bytes += bytes_t[t];
         bad |= bad_t[t];
      }
      if (bad & 1) {
         *err = "plan: malformed locus (need niso >= 1, nrow >= 0, f_off step == nrow*niso)";
         return SBGPU_EINVAL;
      }
      if (bad & 2) {
         *err = "plan: a locus has more than 512 isoforms";
         return SBGPU_ESHAPE;
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

   // Phases of the wave kind (plan.h: LatPhase).  Defaults, for batches that fill the chip: suspend at 32, 128 and
   // 512 iterations; the later phases weigh a layout's lanes 8, 2 and 0.25 times its latency.
   std::vector<int> limits = tune.phase_limits;
   std::vector<double> lambdas = tune.phase_lambda;
   if (limits.empty() && tune.phases_auto && n_loci >= 4096) {
      limits = {32, 128, 512};
      lambdas = {8.0, 2.0, 0.25};
   }
   {
      std::vector<int> ok;
      for (int v : limits)
         if (v > (ok.empty() ? 1 : ok.back()) && v < SBGPU_EM_MAX_ITER && ok.size() < 6) ok.push_back(v);
      limits = ok;
      while (lambdas.size() < limits.size()) lambdas.push_back(lambdas.empty() ? 1.0 : lambdas.back());
   }
   const bool phased = !limits.empty();

   // Classes are few (kinds x layouts x group sizes).  Each host thread classifies a contiguous range of
   // loci into its own table; the tables are merged in range order, so a class list stays in locus order.
   typedef std::tuple<int, int, int, int> ClassKey;
   struct Local {
      std::map<ClassKey, int> slot_of;
      std::vector<SizeClass> found;
      int64_t n_stream = 0;
   };
   std::vector<Local> local(nt);
   std::vector<int64_t> work_of((size_t)n_loci);
   parallel_ranges(n_loci, nt, [&](int64_t lb, int64_t le, unsigned t) {
      Local &L = local[t];
      int last_slot = -1;
      ClassKey last_key(-1, -1, -1, -1);
      for (int64_t l = lb; l < le; ++l) {
         const int64_t nrow = row_off[l + 1] - row_off[l];
         const int64_t niso = iso_off[l + 1] - iso_off[l];
         work_of[(size_t)l] = tune.order_by_work ? nrow * niso : predicted_iterations(nrow, niso) * ((int64_t)1 << 32) + nrow * niso;
         int kind = kStream, layout = 0, CPL = 0, CL = 0, rmult = 1, R = 0, G = 0, lbG = 0;
         if (niso <= kMaxTileC) {
            layout_for(niso, &CPL, &CL);
            layout = layout_id(CPL, CL);
            // wave kind: smallest power-of-two group that holds the rows
            bool placed = false;
            {
               const int Rw = tile_rows(CPL, wave_rmult);
               const int64_t lanes = (int64_t)pow2ceil(std::max<int64_t>(1, (nrow + Rw - 1) / Rw)) * CL;
               LatLayout ll;
               // a phased locus must fit a lane-rich layout too (the few that do not go to the workgroup kinds)
               if (lanes <= 64 && (!phased || lat_layout_for(nrow, niso, 0.0, &ll))) {
                  kind = (wave_rmult == 1) ? kWaveH : (wave_rmult == 2 ? kWave1 : kWave2);
                  rmult = wave_rmult;
                  R = Rw;
                  G = (int)lanes;
                  lbG = ilog2i(G);
                  placed = true;
               }
            }
            // block kinds: the whole 256-lane workgroup is the group
            for (int tall = tune.light_block ? 0 : 1; tall < 2 && !placed; ++tall) {
               const int rm = tall ? kBlockTallRh : kBlockRh;
               if ((int64_t)(kBlockThreads / CL) * tile_rows(CPL, rm) >= nrow) {
                  kind = tall ? kBlockTall : kBlock;
                  rmult = rm;
                  R = tile_rows(CPL, rm);
                  G = kBlockThreads;
                  lbG = 6;
                  placed = true;
               }
            }
         }
         if (kind == kStream) {
            layout = CPL = CL = R = G = lbG = 0;
            rmult = 1;
            ++L.n_stream;
         }
         const ClassKey key(kind, layout, rmult, G);
         int slot;
         if (key == last_key) {
            slot = last_slot; // neighbouring loci are often of one class
         } else {
            auto it = L.slot_of.find(key);
            if (it == L.slot_of.end()) {
               it = L.slot_of.emplace(key, (int)L.found.size()).first;
               SizeClass k;
               k.kind = kind, k.layout = layout, k.CPL = CPL, k.CL = CL, k.rmult = rmult, k.R = R, k.G = G, k.lbG = lbG;
               L.found.push_back(std::move(k));
            }
            slot = it->second;
            last_key = key;
            last_slot = slot;
         }
         SizeClass &sc = L.found[(size_t)slot];
         sc.loci.push_back((int32_t)l);
         sc.work += (kind == kStream) ? nrow * niso : (int64_t)(G / std::max(1, CL)) * R * CPL * CL;
      }
   });
   // merge in range order; the rest of the function walks the classes in key order
   std::map<ClassKey, SizeClass> merged;
   for (unsigned t = 0; t < nt; ++t) {
      p.n_stream_loci += local[t].n_stream;
      for (auto &kv : local[t].slot_of) {
         SizeClass &src = local[t].found[(size_t)kv.second];
         auto it = merged.find(kv.first);
         if (it == merged.end()) {
            merged.emplace(kv.first, std::move(src));
         } else {
            it->second.loci.insert(it->second.loci.end(), src.loci.begin(), src.loci.end());
            it->second.work += src.work;
         }
      }
   }
   std::vector<std::pair<ClassKey, SizeClass>> by_key;
   for (auto &kv : merged) by_key.emplace_back(kv.first, std::move(kv.second));
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
   {
      // likeliest stragglers first inside a class (predicted_iterations, then the heavier locus); ties keep
      // locus order.  Classes are independent: host threads take them one at a time.
      std::atomic<size_t> next(0);
      auto sorter = [&]() {
         for (;;) {
            const size_t ci = next.fetch_add(1);
            if (ci >= by_key.size()) break;
            std::vector<int32_t> &loci = by_key[ci].second.loci;
            std::sort(loci.begin(), loci.end(), [&](int32_t x, int32_t y) {
               const int64_t wx = work_of[(size_t)x], wy = work_of[(size_t)y];
               return wx != wy ? wx > wy : x < y;
            });
         }
      };
      if (nt <= 1) {
         sorter();
      } else {
         std::vector<std::thread> pool;
         for (unsigned t = 0; t < nt; ++t) pool.emplace_back(sorter);
         for (auto &th : pool) th.join();
      }
   }
   if (!tune.order_by_work)
      for (auto &kv : by_key)
         if (!kv.second.loci.empty()) kv.second.pred = work_of[(size_t)kv.second.loci[0]] >> 32;
   for (auto &kv : by_key) {
      SizeClass &sc = kv.second;
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
   // (a class's n_blocks stays its number of batches; when a kind wants more waves than `max_waves` the launch's
   // grid is cut instead and its workgroups stride over the batches)
   p.grid_scale = waves_wanted > max_waves ? (double)max_waves / (double)waves_wanted : 1.0;
   for (auto &kv : by_key) p.classes.push_back(std::move(kv.second));
   const bool by_pred = tune.classes_by_prediction;
   std::stable_sort(p.classes.begin(), p.classes.end(), [by_pred](const SizeClass &x, const SizeClass &y) {
      if (x.kind != y.kind) return x.kind < y.kind;
      // The makespan is set by the loci that run the most iterations: the classes likely to hold them (in steps
      // of 50 predicted iterations) start first -- C3: 1.67 -> 1.61 ms, with the stream priorities 1.49 ms.
      if (by_pred && x.pred / 50 != y.pred / 50) return x.pred > y.pred;
      // Lowest block indices are dispatched first.  The makespan is set by the loci that run
      // all 1000 iterations, and an iteration costs more the more lanes (and column lanes) a
      // locus spans: those classes go first, the many short narrow ones fill in behind.
      const int cx = x.lbG + x.CPL + 4 * (x.layout / 8), cy = y.lbG + y.CPL + 4 * (y.layout / 8);
      if (cx != cy) return cx > cy;
      return x.work > y.work;
   });

   // ---- later phases of the wave kind: every phased locus gets its class in each of them
   p.first_limit = SBGPU_EM_MAX_ITER;
   if (phased) {
      std::vector<int32_t> wave_loci;
      for (const SizeClass &sc : p.classes)
         if (sc.kind == kWaveH || sc.kind == kWave1 || sc.kind == kWave2) wave_loci.insert(wave_loci.end(), sc.loci.begin(), sc.loci.end());
      if (!wave_loci.empty()) {
         p.first_limit = limits[0];
         p.lat.resize(limits.size());
         for (size_t ph = 0; ph < limits.size(); ++ph) {
            LatPhase &lp = p.lat[ph];
            lp.it_limit = ph + 1 < limits.size() ? limits[ph + 1] : SBGPU_EM_MAX_ITER;
            lp.route.assign((size_t)n_loci, -1);
            const double lambda = lambdas[ph];
            if (lambda < 0.0) {
               // re-pack: the phase's classes are the wave kind's phase-0 classes, in their order
               lp.repack = true;
               for (const SizeClass &sc : p.classes) {
                  if (!(sc.kind == kWaveH || sc.kind == kWave1 || sc.kind == kWave2)) continue;
                  SizeClass c2 = sc;
                  c2.loci.clear();
                  for (int32_t l : sc.loci) lp.route[(size_t)l] = (int32_t)lp.classes.size();
                  lp.capacity.push_back((int32_t)sc.loci.size());
                  lp.max_blocks += sc.n_blocks;
                  lp.classes.push_back(std::move(c2));
               }
               continue;
            }
            // layouts in parallel, classes serially (they are few)
            std::vector<LatLayout> lay(wave_loci.size());
            parallel_ranges((int64_t)wave_loci.size(), nt, [&](int64_t b, int64_t e, unsigned) {
               for (int64_t k = b; k < e; ++k) {
                  const int32_t l = wave_loci[(size_t)k];
                  lat_layout_for(row_off[l + 1] - row_off[l], iso_off[l + 1] - iso_off[l], lambda, &lay[(size_t)k]);
               }
            });
            std::map<ClassKey, int> slot;
            std::vector<double> cyc;
            for (size_t k = 0; k < wave_loci.size(); ++k) {
               const LatLayout &ll = lay[k];
               const ClassKey key(ll.cpl, ll.lb_cl, ll.r, ll.lbG);
               auto it = slot.find(key);
               if (it == slot.end()) {
                  it = slot.emplace(key, (int)lp.classes.size()).first;
                  SizeClass sc;
                  sc.kind = kLat;
                  sc.CPL = ll.cpl, sc.CL = 1 << ll.lb_cl, sc.R = ll.r, sc.rmult = ll.r;
                  sc.layout = (ll.cpl - 1) | (ll.lb_cl << 2);
                  sc.lbG = ll.lbG, sc.G = 1 << ll.lbG;
                  sc.block_threads = 64;
                  lp.classes.push_back(sc);
                  lp.capacity.push_back(0);
                  cyc.push_back(ll.cycles);
               }
               lp.capacity[(size_t)it->second] += 1;
               lp.route[(size_t)wave_loci[k]] = it->second;
            }
            // slowest iterations first (lowest block indices are dispatched first); routes follow the permutation
            std::vector<int> order(lp.classes.size()), where(lp.classes.size());
            for (size_t i = 0; i < order.size(); ++i) order[i] = (int)i;
            std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return cyc[(size_t)x] > cyc[(size_t)y]; });
            std::vector<SizeClass> cls2;
            std::vector<int32_t> cap2;
            for (size_t i = 0; i < order.size(); ++i) {
               where[(size_t)order[i]] = (int)i;
               cls2.push_back(lp.classes[(size_t)order[i]]);
               cap2.push_back(lp.capacity[(size_t)order[i]]);
            }
            lp.classes.swap(cls2);
            lp.capacity.swap(cap2);
            for (int32_t l : wave_loci) lp.route[(size_t)l] = where[(size_t)lp.route[(size_t)l]];
            lp.max_blocks = 0;
            for (size_t i = 0; i < lp.classes.size(); ++i) {
               const int lpw = 64 >> lp.classes[i].lbG;
               lp.max_blocks += (lp.capacity[i] + lpw - 1) / lpw;
            }
         }
      }
   }
   return SBGPU_OK;
}

} // namespace sb
