// vk_views.h: the plain structs that both the kernels (vk_common.h) and the host-compiled units (vk_host.h) see.  No HIP.
#pragma once

namespace vk {

struct PPView {           // a vk_pp living in global memory (device pointers)
  int n_int;
  int lead;
  double inv_h;
  const double* knots;
  const double* coef;
};

}  // namespace vk
