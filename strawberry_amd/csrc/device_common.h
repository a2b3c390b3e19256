// strawberry_amd/csrc/device_common.h -- shared by every kernel of libsbgpu.so
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sb {

// fp64 denormals flush to zero, like the reference build: it is compiled -Ofast
// (/root/reference/CMakeLists.txt:84), whose crtfastmath.o sets FTZ/DAZ.  That is
// observable: it decides WHEN a decaying theta_j becomes exactly 0 and a row denominator
// trips the `denom == 0` exit (src/estimate.cpp:451), and whether a far-tail insert-size
// density counts as 0.  MODE.FP_DENORM[7:6] = 0 (flush fp64/fp16 inputs and outputs).
__device__ __forceinline__ void set_fp64_flush_denormals()
{
   __builtin_amdgcn_s_setreg(1 | (6 << 6) | ((2 - 1) << 11), 0);
}

// fp32 variant (BASELINE config 5): flush fp32 denormals as well (MODE.FP_DENORM[5:4] = 0), like -Ofast would
__device__ __forceinline__ void set_all_flush_denormals()
{
   __builtin_amdgcn_s_setreg(1 | (4 << 6) | ((4 - 1) << 11), 0);
}
template <class T>
__device__ __forceinline__ void set_flush_denormals()
{
   if (sizeof(T) == 4) set_all_flush_denormals();
   else set_fp64_flush_denormals();
}

// XCD-aware tile order for kernels whose neighbouring tiles read neighbouring memory (a gather through a permutation that
// is local: sorted positions against arrival order inside a cluster).  The hardware deals workgroups round-robin over the 8
// XCDs, each with its own L2 (MI355X_MICROARCH.md, workgroup dispatch; observed, not promised -- this is for speed only): with
// tile = workgroup index, the lines two neighbouring tiles share are fetched from HBM by two L2s -- by all eight over a few
// tiles.  Here XCD x works through the x-th CONTIGUOUS eighth of the tiles, so a line is fetched by one L2.  The grid must
// be a multiple of 8 (xcd_grid); tiles beyond the last are the caller's bounds check.
__device__ __forceinline__ int64_t xcd_tile()
{
   const int64_t b = blockIdx.x, per = gridDim.x >> 3;
   return (b & 7) * per + (b >> 3);
}
inline unsigned xcd_grid(int64_t n_tiles) { return (unsigned)((n_tiles + 7) & ~(int64_t)7); }

// value of lane (lane ^ MASK), true xor for every MASK
template <int MASK>
__device__ __forceinline__ int xor_get_i(int x)
{
   if (MASK == 1) return __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, true);
   if (MASK == 2) return __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, true);
   if (MASK == 4) {
      // banks 0,2 (lanes 0-3, 8-11 of each row) read lane+4, banks 1,3 read lane-4
      int t = __builtin_amdgcn_update_dpp(0, x, 0x104 /*row_shl:4*/, 0xF, 0x5, false);
      return __builtin_amdgcn_update_dpp(t, x, 0x114 /*row_shr:4*/, 0xF, 0xA, false);
   }
   if (MASK == 8) return __builtin_amdgcn_update_dpp(0, x, 0x128 /*row_ror:8*/, 0xF, 0xF, true);
   if (MASK == 16) {
      // v_permlane16_swap: even 16-lane rows end up twice in a[0], odd rows twice in a[1]
      auto a = __builtin_amdgcn_permlane16_swap(x, x, false, false);
      return ((threadIdx.x & 16) ? a[0] : a[1]);
   }
   // MASK == 32
   auto a = __builtin_amdgcn_permlane32_swap(x, x, false, false);
   return ((threadIdx.x & 32) ? a[0] : a[1]);
}
// all-lanes min / max of a 32-bit value over the wave, as a uniform (SGPR) result
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v)
{
   v = min(v, (uint32_t)xor_get_i<1>((int)v));
   v = min(v, (uint32_t)xor_get_i<2>((int)v));
   v = min(v, (uint32_t)xor_get_i<4>((int)v));
   v = min(v, (uint32_t)xor_get_i<8>((int)v));
   v = min(v, (uint32_t)xor_get_i<16>((int)v));
   v = min(v, (uint32_t)xor_get_i<32>((int)v));
   return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v)
{
   v = max(v, (uint32_t)xor_get_i<1>((int)v));
   v = max(v, (uint32_t)xor_get_i<2>((int)v));
   v = max(v, (uint32_t)xor_get_i<4>((int)v));
   v = max(v, (uint32_t)xor_get_i<8>((int)v));
   v = max(v, (uint32_t)xor_get_i<16>((int)v));
   v = max(v, (uint32_t)xor_get_i<32>((int)v));
   return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}

// The range an element of a wave lies in -- "the last l with off[l] <= e", off ascending from off[0] <= e -- for the 64
// CONSECUTIVE elements e0 .. e0 + 63 of a wave (e = e0 + lane; elements from n_elems on ask as the last real one).  A lane
// that searches by itself makes log2(n_off) DEPENDENT loads (17 for 60 000 ranges); here the wave finds the ranges of
// its first and of its last element 64 ways at a time (three rounds of two loads for 60 000), and a lane then only
// searches between those two -- usually the same range or neighbours.  Every lane of the wave must call.
__device__ __forceinline__ int64_t wave_range_of(const int64_t *__restrict__ off, int64_t n_off, int64_t e0, int64_t n_elems)
{
   const int lane = (int)(threadIdx.x & 63u);
   const int64_t first = e0, last = e0 + 63 < n_elems ? e0 + 63 : n_elems - 1; // (uniform)
   int64_t lo_a = 0, hi_a = n_off, lo_b = 0, hi_b = n_off;                     // the answers lie in [lo, hi)
   while (hi_a - lo_a > 1 || hi_b - lo_b > 1) {
      const int64_t st_a = (hi_a - lo_a + 63) >> 6, st_b = (hi_b - lo_b + 63) >> 6;
      const int64_t ia = lo_a + lane * st_a, ib = lo_b + lane * st_b;
      const int64_t va = ia < hi_a ? off[ia] : INT64_MAX, vb = ib < hi_b ? off[ib] : INT64_MAX;
      const unsigned long long ma = __ballot(va <= first), mb = __ballot(vb <= last); // (prefixes of the lanes: off ascends)
      const int ka = ma ? 63 - __clzll((long long)ma) : 0, kb = mb ? 63 - __clzll((long long)mb) : 0;
      lo_a += ka * st_a, hi_a = lo_a + st_a < hi_a ? lo_a + st_a : hi_a;
      lo_b += kb * st_b, hi_b = lo_b + st_b < hi_b ? lo_b + st_b : hi_b;
   }
   const int64_t e = e0 + lane < n_elems ? e0 + lane : n_elems - 1;
   int64_t lo = lo_a, hi = lo_b + 1;
   while (hi - lo > 1) {
      const int64_t mid = (lo + hi) >> 1;
      if (off[mid] <= e) lo = mid;
      else hi = mid;
   }
   return lo;
}

} // namespace sb
