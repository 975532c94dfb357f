// vk_cells_streaming.hip: the cells kernel's instantiations for the streaming model (the kernels of the headline metric and of
// the BOSS configuration), a translation unit of its own because it is compiled with another machine scheduler -
// -mllvm -amdgpu-sched-strategy=iterative-ilp, see victor_amd/build.py, the Makefile and DESIGN.md section 5.
#define VK_KERNEL_TEMPLATES_ONLY
#include "vk_kernel_cells.h"

namespace vk {
#define VK_CELLS_STREAMING(NLR, NL) \
  template __global__ void vk_theory_cells_kernel<NLR, NL, 0, kModeStreaming, 0>(TheoryArgs); \
  template __global__ void vk_theory_cells_kernel<NLR, NL, 1, kModeStreaming, 0>(TheoryArgs);
VK_CELLS_STREAMING(1, 1) VK_CELLS_STREAMING(1, 2) VK_CELLS_STREAMING(1, 3)
VK_CELLS_STREAMING(2, 1) VK_CELLS_STREAMING(2, 2) VK_CELLS_STREAMING(2, 3)
VK_CELLS_STREAMING(3, 1) VK_CELLS_STREAMING(3, 2) VK_CELLS_STREAMING(3, 3)
#undef VK_CELLS_STREAMING
}  // namespace vk
