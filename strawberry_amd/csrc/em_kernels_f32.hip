// strawberry_amd/csrc/em_kernels_f32.hip -- the fp32 instantiations of the fused EM kernel (BASELINE config 5:
// the fp32 side of the fp32-vs-fp64 tolerance sweep; F, theta and all arithmetic in fp32).  Not a parity target.
#include "em_device.h"
#include "plan.h"

namespace sb {
hipError_t launch_fused_f32(int kind, const FusedLaunchF32 &l, hipStream_t s)
{
   if (kind == kWave1)
      hipLaunchKernelGGL((em_fused_kernel<0, 2, float>), dim3(l.n_blocks), dim3(64), 0, s, l.a, l.ph);
   else if (kind == kWave2)
      hipLaunchKernelGGL((em_fused_kernel<0, 4, float>), dim3(l.n_blocks), dim3(64), 0, s, l.a, l.ph);
   else if (kind == kBlock)
      hipLaunchKernelGGL((em_fused_kernel<kBlockWaves, 2, float>), dim3(l.n_blocks), dim3(64 * kBlockWaves), 0, s, l.a, l.ph);
   else if (kind == kBlockTall)
      hipLaunchKernelGGL((em_fused_kernel<kBlockWaves, 12, float>), dim3(l.n_blocks), dim3(64 * kBlockWaves), 0, s, l.a, l.ph);
   else
      return hipErrorInvalidValue;
   return hipGetLastError();
}
} // namespace sb
