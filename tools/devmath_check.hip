// Standalone accuracy check of vk_devmath.h on the GPU:  hipcc --offload-arch=gfx950 -O3 -I victor_amd/csrc tools/devmath_check.hip -o /tmp/devmath_check
// Prints the maximum error in ulp of sqrt_rsqrt, recip and exp_nonpos, and the relative errors of the streaming integrand's cheaper forms
// (recip_nr, rsqrt_nr_x2, exp_gauss<0>, exp_gauss<1>), against the host's long-double values.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
#include "vk_devmath.h"

__global__ void run_nr(const double* x, double* out, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = vkm::rsqrt_nr(x[i]);
}

__global__ void run(const double* x, const double* a, double* g, double* ir, double* rc, double* ex, double* raw_rsq,
                    double* raw_rcp, int n) {
  __shared__ double tab[vkm::kExpNonposTab];
  for (int j = threadIdx.x; j < vkm::kExpNonposTab; j += blockDim.x) tab[j] = vkm::exp2_frac(j);
  __syncthreads();
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  vkm::sqrt_rsqrt(x[i], g[i], ir[i]);
  rc[i] = vkm::recip(x[i]);
  ex[i] = vkm::exp_nonpos(a[i], tab);
  raw_rsq[i] = __builtin_amdgcn_rsq(x[i]);
  raw_rcp[i] = __builtin_amdgcn_rcp(x[i]);
}

// vkm::exp_gauss<EXPT>: exp(-z^2/2) from (ynum, 1/SV) with y = kExpScale z; slots 0..3 of the inputs are special values
template <int EXPT>
__global__ void run_gauss(const double* yn, const double* isv, double* out, int n) {
  extern __shared__ double tab[];
  vkm::clamp_keeps_nan();
  for (int j = threadIdx.x; j < vkm::ExpCfg<EXPT>::kDoubles; j += blockDim.x) tab[j] = vkm::exp_table_slot<EXPT>(j);
  __syncthreads();
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = vkm::exp_gauss<EXPT>(yn[i], isv[i], tab, (unsigned)(threadIdx.x & 31) << 3);
}

// the one-step reciprocal and the doubled one-step 1/sqrt of the streaming integrand
__global__ void run_nr2(const double* x, double* rc, double* rs, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    rc[i] = vkm::recip_nr(x[i]);
    rs[i] = vkm::rsqrt_nr_x2(x[i]);
  }
}

// vkm::wave_sum (DPP reduction): every lane of a wavefront must receive the same total of the wave's 64 inputs
__global__ void run_wave_sum(const double* x, double* out, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = vkm::wave_sum(x[i]);
}

static double ulp_err(double got, long double want) {
  if (want == 0.0L) return got == 0.0 ? 0.0 : 1e300;
  int e;
  frexpl(want, &e);
  long double ulp = ldexpl(1.0L, e - 53);
  if (fabsl(want) < 2.3e-308L) ulp = 4.94e-324L;
  return (double)(fabsl((long double)got - want) / ulp);
}

int main() {
  const int n = 1 << 22;
  std::vector<double> x(n), a(n);
  unsigned long long s = 88172645463325252ULL;
  auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (double)(s >> 11) / 9007199254740992.0; };
  for (int i = 0; i < n; ++i) {
    x[i] = exp((rnd() * 2 - 1) * 20.0);                       // 2e-9 .. 5e8
    a[i] = (i & 7) == 0 ? -rnd() * 760.0 : -rnd() * 40.0;      // mostly the range the kernel sees, plus the underflow tail
  }
  a[0] = 0.0; a[1] = -745.2; a[2] = -1e-300; a[3] = -708.4;
  double *dx, *da, *dg, *dir, *drc, *dex, *d1, *d2;
  size_t nb = n * sizeof(double);
  hipMalloc(&dx, nb); hipMalloc(&da, nb); hipMalloc(&dg, nb); hipMalloc(&dir, nb); hipMalloc(&drc, nb); hipMalloc(&dex, nb);
  hipMalloc(&d1, nb); hipMalloc(&d2, nb);
  hipMemcpy(dx, x.data(), nb, hipMemcpyHostToDevice);
  hipMemcpy(da, a.data(), nb, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(run, dim3(n / 256), dim3(256), 0, 0, dx, da, dg, dir, drc, dex, d1, d2, n);
  std::vector<double> g(n), ir(n), rc(n), ex(n), r1(n), r2(n);
  hipMemcpy(g.data(), dg, nb, hipMemcpyDeviceToHost); hipMemcpy(ir.data(), dir, nb, hipMemcpyDeviceToHost);
  hipMemcpy(rc.data(), drc, nb, hipMemcpyDeviceToHost); hipMemcpy(ex.data(), dex, nb, hipMemcpyDeviceToHost);
  hipMemcpy(r1.data(), d1, nb, hipMemcpyDeviceToHost); hipMemcpy(r2.data(), d2, nb, hipMemcpyDeviceToHost);
  double m_g = 0, m_ir = 0, m_rc = 0, m_ex = 0, m_exn = 0, raw1 = 0, raw2 = 0;
  for (int i = 0; i < n; ++i) {
    long double xl = x[i];
    m_g = fmax(m_g, ulp_err(g[i], sqrtl(xl)));
    m_ir = fmax(m_ir, ulp_err(ir[i], 1.0L / sqrtl(xl)));
    m_rc = fmax(m_rc, ulp_err(rc[i], 1.0L / xl));
    long double el = expl((long double)a[i]);
    double ue = ulp_err(ex[i], el);
    if (a[i] > -700.0) m_exn = fmax(m_exn, ue); else m_ex = fmax(m_ex, ue);
    raw1 = fmax(raw1, fabs((double)((long double)r1[i] * sqrtl(xl) - 1.0L)));
    raw2 = fmax(raw2, fabs((double)((long double)r2[i] * xl - 1.0L)));
  }
  // rsqrt_nr (one Newton step): worst case and mean error in ulp
  std::vector<double> nr(n);
  hipLaunchKernelGGL(run_nr, dim3(n / 256), dim3(256), 0, 0, dx, dg, n);
  hipMemcpy(nr.data(), dg, nb, hipMemcpyDeviceToHost);
  double m_nr = 0, s_nr = 0;
  for (int i = 0; i < n; ++i) {
    const double u = ulp_err(nr[i], 1.0L / sqrtl((long double)x[i]));
    m_nr = fmax(m_nr, u);
    s_nr += u;
  }

  // exp_gauss: z = ynum * inv_sv / kExpScale in [-12, 12] mostly (the kernels see |z| <~ 8), tails out to the underflow;
  // the reference value is taken at the rounded product y' = fl(|ynum inv_sv|), which is an input of the range reduction
  std::vector<double> yn(n), isv(n), eg(n);
  for (int i = 0; i < n; ++i) {
    const double z = ((i & 15) == 0 ? 39.0 : 12.0) * (rnd() * 2 - 1);
    isv[i] = 0.5 + 1.5 * rnd();
    yn[i] = z * vkm::kExpScale / isv[i];
  }
  yn[0] = NAN; isv[0] = 1.0;                       // NaN must propagate (clamp_keeps_nan)
  yn[1] = 1e300; isv[1] = 1e5;                     // saturates at y' = 1 (|z| = 37.7): below the normal range, not NaN
  yn[2] = 0.0; isv[2] = 1.0;                       // exp(0) = 1
  yn[3] = -3000.0 * vkm::kExpScale; isv[3] = 1.0;  // beyond the clamp: likewise
  double *dyn, *disv, *deg;
  hipMalloc(&dyn, nb); hipMalloc(&disv, nb); hipMalloc(&deg, nb);
  hipMemcpy(dyn, yn.data(), nb, hipMemcpyHostToDevice);
  hipMemcpy(disv, isv.data(), nb, hipMemcpyHostToDevice);
  double g_rel[2] = {0, 0}, g_mean[2] = {0, 0};
  int gauss_special_ok = 1;
  for (int expt = 0; expt < 2; ++expt) {
    if (expt == 0) hipLaunchKernelGGL(run_gauss<0>, dim3(n / 256), dim3(256), vkm::ExpCfg<0>::kDoubles * 8, 0, dyn, disv, deg, n);
    else hipLaunchKernelGGL(run_gauss<1>, dim3(n / 256), dim3(256), vkm::ExpCfg<1>::kDoubles * 8, 0, dyn, disv, deg, n);
    hipMemcpy(eg.data(), deg, nb, hipMemcpyDeviceToHost);
    long double sum = 0;
    long cnt = 0;
    for (int i = 4; i < n; ++i) {
      const double yp = fabs(yn[i] * isv[i]);                         // one rounding, as the device's v_mul_f64
      const long double ys = (long double)yp * 512.0L;                 // = z sqrt(128/ln2)
      const long double el = expl(-ys * ys * (logl(2.0L) / 256.0L));
      if (el > 1e-290L) {                                              // |z| <= 36.5; beyond, only "tiny" is asserted (specials)
        const long double rel = ((long double)eg[i] - el) / el;
        g_rel[expt] = fmax(g_rel[expt], (double)fabsl(rel));
        sum += rel;
        ++cnt;
      } else if (!(eg[i] >= 0.0 && eg[i] < 1e-280)) {
        gauss_special_ok = 0;
      }
    }
    g_mean[expt] = (double)(sum / cnt);
    gauss_special_ok &= std::isnan(eg[0]) && eg[1] >= 0.0 && eg[1] < 1e-300 && eg[2] == 1.0 && eg[3] >= 0.0 && eg[3] < 1e-300;
  }
  // recip_nr / rsqrt_nr_x2: relative errors
  std::vector<double> rc2(n), rs2(n);
  hipLaunchKernelGGL(run_nr2, dim3(n / 256), dim3(256), 0, 0, dx, dg, dir, n);
  hipMemcpy(rc2.data(), dg, nb, hipMemcpyDeviceToHost);
  hipMemcpy(rs2.data(), dir, nb, hipMemcpyDeviceToHost);
  double m_rc2 = 0, m_rs2 = 0;
  for (int i = 0; i < n; ++i) {
    const long double xl = x[i];
    m_rc2 = fmax(m_rc2, (double)fabsl((long double)rc2[i] * xl - 1.0L));
    m_rs2 = fmax(m_rs2, (double)fabsl((long double)rs2[i] * sqrtl(xl) * 0.5L - 1.0L));
  }

  // wave_sum on the first 2^16 samples of x and of a (signed), 64 consecutive values per wavefront
  const int nw = 1 << 16;
  double *dws;
  hipMalloc(&dws, nw * sizeof(double));
  double ws_rel = 0;
  int ws_uniform = 1;
  for (int pass = 0; pass < 2; ++pass) {
    hipLaunchKernelGGL(run_wave_sum, dim3(nw / 256), dim3(256), 0, 0, pass ? da : dx, dws, nw);
    std::vector<double> w(nw);
    hipMemcpy(w.data(), dws, nw * sizeof(double), hipMemcpyDeviceToHost);
    const std::vector<double>& src = pass ? a : x;
    for (int g = 0; g < nw; g += 64) {
      long double want = 0, mag = 0;
      for (int l = 0; l < 64; ++l) { want += src[g + l]; mag += fabsl((long double)src[g + l]); }
      for (int l = 1; l < 64; ++l) ws_uniform &= (w[g + l] == w[g]);
      ws_rel = fmax(ws_rel, (double)(fabsl((long double)w[g] - want) / mag));
    }
  }
  printf("{\"rsqrt_nr_ulp_max\": %.3f, \"rsqrt_nr_ulp_mean\": %.3f, ", m_nr, s_nr / n);
  printf("\"gauss_rel_max\": [%.3e, %.3e], \"gauss_rel_mean\": [%.3e, %.3e], \"gauss_special_ok\": %d, \"recip_nr_rel\": %.3e, \"rsqrt_nr_x2_rel\": %.3e, ",
         g_rel[0], g_rel[1], g_mean[0], g_mean[1], gauss_special_ok, m_rc2, m_rs2);
  printf("\"wave_sum_rel\": %.3e, \"wave_sum_uniform\": %d, \"sqrt_ulp\": %.3f, \"rsqrt_ulp\": %.3f, \"recip_ulp\": %.3f, \"exp_ulp_normal\": %.3f, \"exp_ulp_denormal_tail\": %.3f, "
         "\"raw_v_rsq_f64_rel\": %.3e, \"raw_v_rcp_f64_rel\": %.3e}\n", ws_rel, ws_uniform, m_g, m_ir, m_rc, m_exn, m_ex, raw1, raw2);
  return 0;
}
