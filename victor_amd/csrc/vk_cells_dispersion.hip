// vk_cells_dispersion.hip: the cells kernel for the dispersion model - explicit instantiations, a translation unit of its own so that the
// library's units compile side by side (vk_instances.h names what lives where; victor_amd/build.py, the Makefile).
#define VK_KERNEL_TEMPLATES_ONLY
#include "vk_kernel_cells.h"
#include "vk_instances.h"

namespace vk {
VK_UNIT_CELLS_DISPERSION(template)
}  // namespace vk
