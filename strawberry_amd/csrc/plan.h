// strawberry_amd/csrc/plan.h -- host-side size classing of a locus batch.
//
// The reference solves one locus at a time on a CPU thread
// (/root/reference/src/alignments.cpp:1782-1804).  On the GPU thousands of loci
// are in flight at once and their shapes are ragged (1..~2000 bins x 1..~200
// isoforms), so the batch is first sorted into size classes; each class is one
// kernel instantiation (see em_device.h).
#pragma once

#include <cstdint>
#include <vector>

namespace sb {

enum ClassKind : int { kTile = 0, kStream = 1 };

struct SizeClass {
   int kind;     // kTile / kStream
   int CPL, CL;  // tile: columns per lane, column lanes (CPL*CL >= niso)
   int R, G;     // tile: rows per row lane, lanes per locus = CL * row lanes (G > 64: one workgroup)
   std::vector<int32_t> loci; // ordered by decreasing nrow*niso
   int n_blocks = 0;          // launch grid
   int block_threads = 0;
   int64_t work = 0;          // sum of nrow*niso (padded) -- for ordering launches
};

struct HostPlan {
   int64_t n_loci = 0, n_rows = 0, n_iso = 0, n_elem = 0;
   int64_t algorithmic_bytes = 0;
   int64_t n_stream_loci = 0;
   std::vector<SizeClass> classes; // non-empty classes, heaviest first
};

constexpr int kTileElems = 32;    // R*CPL register tile per lane (x2 for the tall workgroup variants)
constexpr int kMaxCPL = 8;        // columns per lane
constexpr int kMaxTileC = 64;     // 8 column lanes x 8 columns; wider loci stream
constexpr int kMaxStreamIso = 512;

// Returns 0, or a negative SBGPU_E* code with `err` filled.
int build_host_plan(int64_t n_loci, const int64_t *row_off, const int64_t *iso_off,
                    const int64_t *f_off, int n_cu, HostPlan *out, const char **err);

} // namespace sb
