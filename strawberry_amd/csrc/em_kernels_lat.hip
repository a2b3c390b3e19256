// strawberry_amd/csrc/em_kernels_lat.hip -- the later phases' kernel (lane-rich wave layouts, em_device.h) and
// the one-workgroup kernel that turns a phase's survivor counts into the next launch's block table.
#define SB_COMPILE_LAT_KERNEL 1
#include "em_device.h"

namespace sb {
hipError_t launch_lat(const FusedLaunch &l, hipStream_t s)
{
   hipLaunchKernelGGL(em_lat_kernel, dim3(l.n_blocks), dim3(64), 0, s, l.a, l.ph);
   return hipGetLastError();
}
hipError_t launch_phase_prepare(ClassDesc *table, const int32_t *n_in, int n_classes, int32_t *total_blocks, hipStream_t s)
{
   hipLaunchKernelGGL(phase_prepare_kernel, dim3(1), dim3(256), 0, s, table, n_in, n_classes, total_blocks);
   return hipGetLastError();
}
} // namespace sb
