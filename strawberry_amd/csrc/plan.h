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
constexpr int kBlockThreads = 256;
constexpr int kBlockRh = 4;
constexpr int kBlockTallRh = 12;

struct SizeClass {
   int kind = kWave1;
   int layout = 0;   // index into kLayoutCPL/kLayoutCL
   int CPL = 0, CL = 0; // columns per lane, column lanes (CPL*CL >= niso)
   int rmult = 2;    // rh: rows per row lane = rh * kLayoutRHalf[layout]
   int R = 0;
   int G = 0;        // lanes per locus = CL * row lanes
   int lbG = 0;      // log2(G) for the wave kind
   std::vector<int32_t> loci; // ordered by decreasing nrow*niso
   int n_blocks = 0;          // workgroups of this class inside its launch
   int block_threads = 0;
   int64_t work = 0;          // padded elements, for ordering
};

struct HostPlan {
   int64_t n_loci = 0, n_rows = 0, n_iso = 0, n_elem = 0;
   int64_t algorithmic_bytes = 0;
   int64_t n_stream_loci = 0;
   std::vector<SizeClass> classes; // grouped by kind; inside a kind heaviest first
};

constexpr int kNumLayouts = 6;
constexpr int kLayoutCPL[kNumLayouts] = {2, 4, 8, 8, 8, 8};
constexpr int kLayoutCL[kNumLayouts] = {1, 1, 1, 2, 4, 8};
constexpr int kLayoutRHalf[kNumLayouts] = {4, 4, 2, 2, 2, 2}; // half of the base rows per row lane
constexpr int kMaxTileC = 64;     // 8 column lanes x 8 columns; wider loci stream
constexpr int kMaxStreamIso = 512;

struct PlanTuning {
   int wave_rmult = 0;   // 0 = auto, 1 / 2 / 4 = force rh of the wave kind (half / base / double tile)
   bool light_block = false; // also use the 2x-rows block kind (<= 256 VGPRs); off: it spills and
                             // fights the wave kind for the same SIMDs (measured slower on C3)
   int64_t max_waves = 0;  // grids shrink (waves pull several batches) only beyond this many waves; 0 = 2^20
};

// Returns 0, or a negative SBGPU_E* code with `err` filled.
int build_host_plan(int64_t n_loci, const int64_t *row_off, const int64_t *iso_off,
                    const int64_t *f_off, int n_cu, const PlanTuning &tune, HostPlan *out,
                    const char **err);

} // namespace sb
