// vk_cells_aniso.hip: the cells kernel's instantiations for the anisotropic real-space sum (NLR = 3, streaming model), a
// translation unit of its own because it is compiled with another machine scheduler (see Makefile / build.py).
#define VK_KERNEL_TEMPLATES_ONLY
#include "vk_kernel_cells.h"

namespace vk {
#define VK_CELLS_ANISO(NL, GRID) template __global__ void vk_theory_cells_kernel<3, NL, GRID, kModeStreaming, 0>(TheoryArgs);
VK_CELLS_ANISO(1, 0) VK_CELLS_ANISO(2, 0) VK_CELLS_ANISO(3, 0)
VK_CELLS_ANISO(1, 1) VK_CELLS_ANISO(2, 1) VK_CELLS_ANISO(3, 1)
#undef VK_CELLS_ANISO
}  // namespace vk
