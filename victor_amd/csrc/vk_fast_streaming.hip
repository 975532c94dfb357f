// vk_fast_streaming.hip: the point-major kernel for the streaming model - explicit instantiations, a translation unit of its own so that the
// library's units compile side by side (vk_instances.h names what lives where; victor_amd/build.py, the Makefile).
#define VK_KERNEL_TEMPLATES_ONLY
#include "vk_kernel_fast.h"
#include "vk_instances.h"

namespace vk {
VK_UNIT_FAST_STREAMING(template)
}  // namespace vk
