// Throughput of the FP64 / integer VALU opcodes the theory kernel uses, in cycles per wave-instruction per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/inst_rate.hip -o tools/inst_rate
// 4 waves per SIMD (1024-thread blocks, one per CU), 8 independent dependency chains per lane.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHAINS 8
#define ITERS 512

template <int OP>
__global__ __launch_bounds__(1024) void rate(double* out, unsigned long long* cyc, double seed) {
  double v[CHAINS];
  int iv[CHAINS];
  for (int c = 0; c < CHAINS; ++c) { v[c] = seed + 0.001 * (threadIdx.x + c); iv[c] = threadIdx.x + c; }
  const double k1 = seed * 0.999, k2 = seed * 1e-3;
  const int kinv = (int)(seed * 77) + threadIdx.x, kinv2 = kinv * 3;
  const unsigned long long smask = __builtin_amdgcn_ballot_w64((threadIdx.x & 1) != 0);
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) {
      if (OP == 0) v[c] = fma(v[c], k1, k2);
      if (OP == 1) v[c] = v[c] * k1;
      if (OP == 2) v[c] = v[c] + k2;
      if (OP == 3) asm volatile("v_min_f64 %0, %1, %2" : "=v"(v[c]) : "v"(v[c]), "v"(k1));
      if (OP == 4) asm volatile("v_rcp_f64 %0, %1" : "=v"(v[c]) : "v"(v[c]));
      if (OP == 5) asm volatile("v_rsq_f64 %0, %1" : "=v"(v[c]) : "v"(v[c]));
      if (OP == 6) asm volatile("v_fract_f64 %0, %1" : "=v"(v[c]) : "v"(v[c]));
      if (OP == 7) asm volatile("v_rndne_f64 %0, %1" : "=v"(v[c]) : "v"(v[c]));
      if (OP == 8) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(iv[c]) : "v"(v[c]));
      if (OP == 9) asm volatile("v_ldexp_f64 %0, %1, %2" : "=v"(v[c]) : "v"(v[c]), "v"(iv[c] & 1));
      if (OP == 10) asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(iv[c]) : "v"(iv[c]), "v"(iv[(c + 1) % CHAINS]));
      if (OP == 11) asm volatile("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(iv[c]) : "v"(iv[c]), "v"(3), "v"(iv[c]));
      if (OP == 12) asm volatile("v_add_u32 %0, %1, %2" : "=v"(iv[c]) : "v"(iv[c]), "v"(iv[(c + 1) % CHAINS]));
      if (OP == 13) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(iv[c]) : "v"(iv[c]), "v"(iv[(c + 1) % CHAINS]));
      if (OP == 14) asm volatile("v_mov_b64 %0, %1" : "=v"(v[c]) : "v"(v[(c + 1) % CHAINS]));
      if (OP == 15) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(v[c]) : "v"(iv[c]));
      if (OP == 16) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(iv[c]) : "v"(iv[c]), "v"(iv[c]), "v"(iv[c]));
      if (OP == 17) asm volatile("v_cmp_gt_f64 vcc, %0, %1" :: "v"(v[c]), "v"(k1) : "vcc");
      if (OP == 18) asm volatile("v_and_b32 %0, %1, %2" : "=v"(iv[c]) : "v"(iv[c]), "v"(iv[(c + 1) % CHAINS]));
      if (OP == 19) asm volatile("v_lshl_add_u32 %0, %1, 3, %2" : "=v"(iv[c]) : "v"(iv[c]), "v"(iv[(c + 1) % CHAINS]));
      if (OP == 20) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(iv[c]), "+v"(iv[(c + 1) % CHAINS]));
      if (OP == 21) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(iv[c]), "+v"(iv[(c + 1) % CHAINS]));
      if (OP == 22) asm volatile("s_nop 1\n v_mov_b32_dpp %0, %1 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(iv[c]) : "v"(iv[(c + 1) % CHAINS]));
      if (OP == 23) asm volatile("s_nop 1\n v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(iv[c]) : "v"(iv[(c + 1) % CHAINS]));
      if (OP == 24) asm volatile("s_nop 1\n v_mov_b32_dpp %0, %1 row_bcast:31 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(iv[c]) : "v"(iv[(c + 1) % CHAINS]));
      if (OP == 25) asm volatile("v_swap_b32 %0, %1" : "+v"(iv[c]), "+v"(iv[(c + 1) % CHAINS]));
      if (OP == 26) asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(iv[c]) : "v"(iv[c]), "v"(iv[(c + 1) % CHAINS]), "s"(smask));
      if (OP == 27) asm volatile("v_mov_b32_dpp %0, %1 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(iv[c]) : "v"(kinv));
      if (OP == 28) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(iv[c]) : "v"(kinv), "v"(kinv2));
      if (OP == 29) asm volatile("s_nop 1\n v_add_u32 %0, %1, %2" : "=v"(iv[c]) : "v"(iv[c]), "v"(iv[(c + 1) % CHAINS]));
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  double s = 0;
  for (int c = 0; c < CHAINS; ++c) s += v[c] + iv[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int OP>
double run(const char* name, double* d_out, unsigned long long* d_cyc, int blocks) {
  hipLaunchKernelGGL(rate<OP>, dim3(blocks), dim3(1024), 0, 0, d_out, d_cyc, 1.0001);
  hipDeviceSynchronize();
  std::vector<unsigned long long> c(blocks);
  hipMemcpy(c.data(), d_cyc, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  double avg = 0;
  for (auto x : c) avg += (double)x;
  avg /= blocks;
  // one block = 16 waves = 4 per SIMD; each wave issues ITERS*CHAINS instructions
  double per = avg / (4.0 * ITERS * CHAINS);
  printf("%-32s %7.2f cycles per wave-instruction per SIMD\n", name, per);
  return per;
}

int main() {
  const int blocks = 256;
  double* d_out; unsigned long long* d_cyc;
  hipMalloc(&d_out, blocks * 1024 * sizeof(double));
  hipMalloc(&d_cyc, blocks * sizeof(unsigned long long));
  for (int rep = 0; rep < 2; ++rep) {
    run<0>("v_fma_f64", d_out, d_cyc, blocks);
    run<1>("v_mul_f64", d_out, d_cyc, blocks);
    run<2>("v_add_f64", d_out, d_cyc, blocks);
    run<3>("v_min_f64", d_out, d_cyc, blocks);
    run<4>("v_rcp_f64", d_out, d_cyc, blocks);
    run<5>("v_rsq_f64", d_out, d_cyc, blocks);
    run<6>("v_fract_f64", d_out, d_cyc, blocks);
    run<7>("v_rndne_f64", d_out, d_cyc, blocks);
    run<8>("v_cvt_i32_f64", d_out, d_cyc, blocks);
    run<9>("v_ldexp_f64", d_out, d_cyc, blocks);
    run<10>("v_mul_lo_u32", d_out, d_cyc, blocks);
    run<11>("v_mad_u32_u24", d_out, d_cyc, blocks);
    run<12>("v_add_u32", d_out, d_cyc, blocks);
    run<13>("v_cndmask_b32", d_out, d_cyc, blocks);
    run<14>("v_mov_b64", d_out, d_cyc, blocks);
    run<15>("v_cvt_f64_i32", d_out, d_cyc, blocks);
    run<16>("v_fma_f32", d_out, d_cyc, blocks);
    run<17>("v_cmp_gt_f64", d_out, d_cyc, blocks);
    run<18>("v_and_b32", d_out, d_cyc, blocks);
    run<19>("v_lshl_add_u32", d_out, d_cyc, blocks);
    run<20>("v_permlane32_swap", d_out, d_cyc, blocks);
    run<21>("v_permlane16_swap", d_out, d_cyc, blocks);
    run<22>("dpp row_ror:8 (+s_nop 1)", d_out, d_cyc, blocks);
    run<23>("dpp quad_perm (+s_nop 1)", d_out, d_cyc, blocks);
    run<24>("dpp row_bcast:31 (+s_nop 1)", d_out, d_cyc, blocks);
    run<25>("v_swap_b32", d_out, d_cyc, blocks);
    run<26>("v_cndmask_b32_e64 (sgpr mask)", d_out, d_cyc, blocks);
    run<27>("dpp mov, no hazard", d_out, d_cyc, blocks);
    run<28>("v_cndmask_b32 vcc, no dep", d_out, d_cyc, blocks);
    run<29>("s_nop 1 + v_add_u32", d_out, d_cyc, blocks);
  }
  return 0;
}
