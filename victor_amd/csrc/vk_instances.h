// vk_instances.h: which kernel instantiation is compiled in which translation unit.  The unit files (vk_cells_*.hip,
// vk_fast_*.hip, vk_generic.hip) expand a list with P = `template` (explicit instantiation: the code is generated there),
// victor_hip.hip expands ALL lists with P = `extern template` (the launch code only takes the kernels' addresses), so the two
// cannot drift apart: an instantiation the launch code reaches that is in no list is simply generated in victor_hip.hip.
// One hipcc per unit, side by side (victor_amd/build.py, the Makefile): the library's build time is that of its slowest unit.
#pragma once

#define VK_9(X, P, ...) X(P, 1, 1, __VA_ARGS__) X(P, 1, 2, __VA_ARGS__) X(P, 1, 3, __VA_ARGS__) \
                        X(P, 2, 1, __VA_ARGS__) X(P, 2, 2, __VA_ARGS__) X(P, 2, 3, __VA_ARGS__) \
                        X(P, 3, 1, __VA_ARGS__) X(P, 3, 2, __VA_ARGS__) X(P, 3, 3, __VA_ARGS__)
#define VK_3(X, P, ...) X(P, 1, __VA_ARGS__) X(P, 2, __VA_ARGS__) X(P, 3, __VA_ARGS__)

#define VK_CELLS_ONE(P, NLR, NL, GRID, MODE, SVA) P __global__ void vk_theory_cells_kernel<NLR, NL, GRID, MODE, SVA>(TheoryArgs);
#define VK_FAST_ONE(P, NLR, NL, GRID, MODE, SVA) P __global__ void vk_theory_fast_kernel<NLR, NL, GRID, MODE, SVA>(TheoryArgs);
#define VK_GENERIC_ONE(P, NLR, NL, RSD) P __global__ void vk_theory_kernel<RSD, NLR, NL>(TheoryArgs);
#define VK_XI_ONE(P, NLR, RSD) P __global__ void vk_xi_smu_kernel<RSD, NLR>(TheoryArgs);

// <NLR, NL, GRID (0 lattice / 1 union grid), MODE, SVA (anisotropic sigma_v(r, mu) patches in LDS)> for NLR, NL = 1..3

// vk_cells_streaming.hip (compiled under the iterative-ilp machine scheduler): the kernels of the headline metric and of BOSS
#define VK_UNIT_CELLS_STREAMING(P) \
  VK_9(VK_CELLS_ONE, P, 0, kModeStreaming, 0) VK_9(VK_CELLS_ONE, P, 1, kModeStreaming, 0)

// vk_cells_dispersion.hip
#define VK_UNIT_CELLS_DISPERSION(P) \
  VK_9(VK_CELLS_ONE, P, 0, kModeDispersion, 0) VK_9(VK_CELLS_ONE, P, 1, kModeDispersion, 0) \
  VK_9(VK_CELLS_ONE, P, 0, kModeDispersionFromData, 0) VK_9(VK_CELLS_ONE, P, 1, kModeDispersionFromData, 0) \
  VK_9(VK_CELLS_ONE, P, 0, kModeDispersion, 1)

// vk_cells_kaiser.hip: kaiser / euclid_special, streaming on a measured real-space ccf, streaming with sigma_v(r, mu)
#define VK_UNIT_CELLS_KAISER(P) \
  VK_9(VK_CELLS_ONE, P, 0, kModeKaiser, 0) VK_9(VK_CELLS_ONE, P, 1, kModeKaiser, 0) \
  VK_9(VK_CELLS_ONE, P, 0, kModeFromData, 0) VK_9(VK_CELLS_ONE, P, 1, kModeFromData, 0) \
  VK_9(VK_CELLS_ONE, P, 0, kModeStreaming, 1)

// vk_fast_streaming.hip: the point-major kernel (fewer than twenty points; one point per call)
#define VK_UNIT_FAST_STREAMING(P) \
  VK_9(VK_FAST_ONE, P, 0, kModeStreaming, 0) VK_9(VK_FAST_ONE, P, 1, kModeStreaming, 0) \
  VK_9(VK_FAST_ONE, P, 0, kModeFromData, 0) VK_9(VK_FAST_ONE, P, 1, kModeFromData, 0) \
  VK_9(VK_FAST_ONE, P, 0, kModeStreaming, 1)

// vk_fast_dispersion.hip
#define VK_UNIT_FAST_DISPERSION(P) \
  VK_9(VK_FAST_ONE, P, 0, kModeDispersion, 0) VK_9(VK_FAST_ONE, P, 1, kModeDispersion, 0) \
  VK_9(VK_FAST_ONE, P, 0, kModeDispersionFromData, 0) VK_9(VK_FAST_ONE, P, 1, kModeDispersionFromData, 0)

// vk_generic.hip: the generic theory kernel <RSD, NLR, NL> and K1x <RSD, NLR> for the four RSD models
#define VK_UNIT_GENERIC(P) \
  VK_9(VK_GENERIC_ONE, P, VK_RSD_STREAMING) VK_9(VK_GENERIC_ONE, P, VK_RSD_DISPERSION) \
  VK_9(VK_GENERIC_ONE, P, VK_RSD_KAISER) VK_9(VK_GENERIC_ONE, P, VK_RSD_EUCLID) \
  VK_3(VK_XI_ONE, P, VK_RSD_STREAMING) VK_3(VK_XI_ONE, P, VK_RSD_DISPERSION) VK_3(VK_XI_ONE, P, VK_RSD_KAISER) VK_3(VK_XI_ONE, P, VK_RSD_EUCLID)
