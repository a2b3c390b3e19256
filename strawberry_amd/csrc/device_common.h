// strawberry_amd/csrc/device_common.h -- shared by every kernel of libsbgpu.so
#pragma once

#include <hip/hip_runtime.h>

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

} // namespace sb
