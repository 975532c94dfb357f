// vk_generic.hip: the generic theory kernel and K1x - explicit instantiations, a translation unit of its own so that the
// library's units compile side by side (vk_instances.h names what lives where; victor_amd/build.py, the Makefile).
#define VK_KERNEL_TEMPLATES_ONLY
#include "vk_kernel_generic.h"
#include "vk_instances.h"

namespace vk {
VK_UNIT_GENERIC(template)
}  // namespace vk
