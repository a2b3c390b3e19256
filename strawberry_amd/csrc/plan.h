// strawberry_amd/csrc/plan.h -- host-side size classing of a locus batch.
//
// The reference solves one locus at a time on a CPU thread
// (/root/reference/src/alignments.cpp:1782-1804).  On the GPU thousands of loci
// are in flight at once and their shapes are ragged (1..~2000 bins x 1..~200
// isoforms), so the batch is first sorted into size classes (register-tile
// layout x lanes per locus, see em_device.h); all classes of one kind run in ONE
// kernel launch whose workgroups look their class up in a descriptor table.
#pragma once

#include <cstdint>
#include <vector>

namespace sb {

// One kernel launch per kind and phase (a batch uses one wave kind, usually one block kind).
// `rh` = rows per row lane in units of half the base tile (kLayoutRHalf).
//  kWaveH/1/2   64/G groups per wave; rh = 1, 2, 4.  A plan uses ONE of them, chosen by load:
//               the half tile has the shortest iteration but needs twice the lanes per locus
//  kBlock       one 256-lane workgroup per locus, rh = 4 (optional, see PlanTuning)
//  kBlockTall   same with rh = 12: tiles up to 6x the base rows
//  kStream      anything larger: F re-read from L2 every iteration
enum ClassKind : int { kWaveH = 0, kWave1, kWave2, kBlock, kBlockTall, kStream, kNumKinds };
constexpr int kLat = kNumKinds; // the later phases' lane-rich wave layouts (their own tables; never a phase-0 kind)
constexpr int kBlockThreads = 256;
constexpr int kBlockRh = 2;
constexpr int kBlockTallRh = 12;

struct SizeClass {
   int kind = kWave1;
   int layout = 0;   // layout_id(CPL, CL)
   int CPL = 0, CL = 0; // columns per lane, column lanes (CPL*CL >= niso)
   int rmult = 2;    // rh: rows per row lane = rh * kLayoutRHalf[layout]
   int R = 0;
   int G = 0;        // lanes per locus = CL * row lanes
   int lbG = 0;      // log2(G) for the wave kind
   std::vector<int32_t> loci; // ordered by decreasing nrow*niso
   int n_blocks = 0;          // workgroups of this class inside its launch
   int block_threads = 0;
   int64_t work = 0;          // padded elements, for ordering
   int64_t pred = 0;          // largest predicted iteration count among its loci (plan.cpp), for ordering
};

// A later phase of the wave kind: the loci still running at the previous phase's iteration limit continue in
// lane-rich layouts (em_device.h: em_lat_kernel).  `route[l]` is the class of locus l in THIS phase (-1: the
// locus is not of the phased wave kind); a class's `loci` is empty on the host -- the lists are filled on the
// device -- and `capacity` says how many loci could reach it.
struct LatPhase {
   bool repack = false;              // true: the survivors are re-packed into their phase-0 layouts (the wave kind's own kernel)
   int it_limit = 0;                 // iterations (total) at which this phase suspends its loci; 1000 = the last phase
   std::vector<SizeClass> classes;
   std::vector<int32_t> capacity;    // per class
   std::vector<int32_t> route;       // [n_loci]
   int64_t max_blocks = 0;           // batches if every locus that can reach the phase does
};

struct HostPlan {
   int64_t n_loci = 0, n_rows = 0, n_iso = 0, n_elem = 0;
   int64_t algorithmic_bytes = 0;
   int64_t n_stream_loci = 0;
   std::vector<SizeClass> classes; // phase 0; grouped by kind; inside a kind heaviest first
   int first_limit = 1000;         // iteration limit of phase 0 for the wave kind (1000: no later phases)
   double grid_scale = 1.0;        // < 1: every launch's grid is this share of its batches (PlanTuning::max_waves)
   std::vector<LatPhase> lat;      // phases 1, 2, ...
};

// register-tile layouts: CPL exact columns per lane x CL column lanes (CPL*CL >= niso);
// layout id = (CPL - 1) + 8 * log2(CL)
constexpr int kLayoutRHalf[8] = {8, 4, 4, 4, 3, 3, 2, 2}; // half of the base rows per row lane, by CPL - 1
// rows per row lane for a tile of rh halves (same cap as em_device.h::tile_rows)
inline int tile_rows(int cpl, int rh)
{
   const int r = kLayoutRHalf[cpl - 1] * rh;
   return (r * cpl > 128) ? 128 / cpl : r;
}
inline int layout_id(int cpl, int cl)
{
   int lb = 0;
   while ((1 << lb) < cl) ++lb;
   return (cpl - 1) + 8 * lb;
}
// niso -> (CPL, CL): one column lane up to 8 isoforms, then 2 / 4 / 8 lanes of 5..8 columns
inline void layout_for(int64_t niso, int *cpl, int *cl)
{
   int c = 1;
   while (niso > 8 * c) c *= 2;
   *cl = c;
   *cpl = (int)((niso + c - 1) / c);
}
constexpr int kMaxTileC = 64;     // 8 column lanes x 8 columns; wider loci stream
constexpr int kMaxStreamIso = 512;

struct PlanTuning {
   int wave_rmult = 0;   // 0 = auto, 1 / 2 / 4 = force rh of the wave kind (half / base / double tile)
   bool light_block = true;  // use the base-tile block kind (<= 256 VGPRs, shares SIMDs with wave-form
                             // waves) for the loci it holds, the tall tile only for the rest
   int64_t max_waves = 0;  // grids shrink (waves pull several batches) only beyond this many waves; 0 = 2^20
   bool classes_by_prediction = true; // dispatch the classes with the largest predicted iteration counts first
                                      // (SBGPU_CLASS_ORDER=cost: by the cost of an iteration only; A/B measurements)
   bool order_by_work = false; // order a class by nrow * niso only (SBGPU_ORDER=work; A/B measurements) instead of
                               // by the iteration count predicted from the shape
   // Phases of the wave kind: iteration limits of all phases but the last (empty: one phase) and, per later phase,
   // the weight of a layout's lane count against its iteration latency (large early, when many loci are alive and
   // lanes are dear; ~0 in the last phase, when the chip is nearly empty and only the latency counts).
   bool phases_auto = false;         // use default limits for batches that fill the chip.  OFF: measured on C3 (profiles/
                                     // r02_phase_sweep_*.txt) every phase schedule is slower than one phase -- see DESIGN.md 3.1
   std::vector<int> phase_limits;
   std::vector<double> phase_lambda; // [phase - 1]; negative: the phase re-packs the survivors into their phase-0 layouts instead
};

// lane-rich layouts (em_device.h, em_lat_kernel): rows per lane on offer and the tile bound
constexpr int kLatRows[6] = {1, 2, 3, 4, 6, 8};
constexpr int kLatMaxTileElems = 32;
struct LatLayout {
   int cpl = 0, lb_cl = 0, r = 0, lbG = 0;
   double cycles = 0.0; // modelled latency of one iteration
};
// best lane-rich layout of a locus under the lane weight `lambda`; false when none holds it
bool lat_layout_for(int64_t nrow, int64_t niso, double lambda, LatLayout *out);

// Returns 0, or a negative SBGPU_E* code with `err` filled.
int build_host_plan(int64_t n_loci, const int64_t *row_off, const int64_t *iso_off,
                    const int64_t *f_off, int n_cu, const PlanTuning &tune, HostPlan *out,
                    const char **err);

} // namespace sb
