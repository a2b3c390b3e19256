// strawberry_amd/csrc/em_kernels_stream.hip -- one instantiation of the fused EM kernel per
// translation unit (they compile in parallel); see em_device.h.
#define SB_COMPILE_STREAM_KERNEL 1
#include "em_device.h"

namespace sb {
hipError_t launch_stream(const EmArgs &a, const ClassArgs &c, uint8_t *row_keep, int n_blocks, size_t lds_bytes,
                         hipStream_t s)
{
   hipLaunchKernelGGL(em_stream_kernel, dim3(n_blocks), dim3(kStreamThreads), lds_bytes, s, a, c, row_keep);
   return hipGetLastError();
}
} // namespace sb
