// strawberry_amd/csrc/em_kernels_block.hip -- one instantiation of the fused EM kernel per
// translation unit (they compile in parallel); see em_device.h.
#include "em_device.h"

namespace sb {
hipError_t launch_fused_block(const FusedLaunch &l, hipStream_t s)
{
   hipLaunchKernelGGL((em_fused_kernel<kBlockWaves, 2>), dim3(l.n_blocks), dim3(64 * kBlockWaves), 0, s, l.a, l.ph);
   return hipGetLastError();
}
#ifdef SB_STAMPS
SB_DEFINE_STAMP_READER(block)
#endif
} // namespace sb
