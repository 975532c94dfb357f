// vk_cells_streaming.hip: the cells kernel's instantiations for the streaming model (the kernels of the headline metric and of
// the BOSS configuration), a translation unit of its own because it is compiled with another machine scheduler -
// -mllvm -amdgpu-sched-strategy=iterative-ilp, see victor_amd/build.py, the Makefile and DESIGN.md section 5.
#define VK_KERNEL_TEMPLATES_ONLY
#include "vk_kernel_cells.h"
#include "vk_instances.h"

namespace vk {
VK_UNIT_CELLS_STREAMING(template)
}  // namespace vk
