// micro-benchmark: issue cost of v_rcp_f64 / v_rcp_f32-seeded reciprocal / v_fma_f64 / exp on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int MODE>
__global__ void k(double* out, int iters, double seed) {
  double a[8];
  for (int i = 0; i < 8; ++i) a[i] = seed + threadIdx.x * 1e-3 + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (MODE == 0) a[i] = __builtin_fma(a[i], 1.0000001, 1e-9);
      if (MODE == 1) a[i] = __builtin_amdgcn_rcp(a[i]) + 1.5;
      if (MODE == 2) { float r = __builtin_amdgcn_rcpf((float)a[i]); a[i] = (double)r + 1.5; }
      if (MODE == 3) a[i] = __builtin_amdgcn_sqrt(a[i]) + 1.5;
      if (MODE == 4) a[i] = __builtin_amdgcn_rsq(a[i]) + 1.5;
      if (MODE == 5) a[i] = a[i] * 1.0000001 + 1e-9;   // same as 0 (contract)
      if (MODE == 6) { a[i] = __builtin_fma(a[i], 1.0000001, 1e-9); a[i] = __builtin_fma(a[i], 0.9999999, 1e-9); }
    }
  }
  double s = 0; for (int i = 0; i < 8; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE> void run(const char* name, int waves_per_simd) {
  int blocks = 256 * waves_per_simd;   // 256-thread blocks: 4 waves, one per SIMD
  double* d; hipMalloc(&d, sizeof(double) * blocks * 256);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  int iters = 20000;
  k<MODE><<<blocks, 256>>>(d, 100, 2.0); hipDeviceSynchronize();
  hipEventRecord(e0); k<MODE><<<blocks, 256>>>(d, iters, 2.0); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double instr_per_simd = (double)iters * 8 * waves_per_simd * (MODE == 6 ? 2 : 1);
  printf("%-22s waves/SIMD=%d  %.3f ms  -> %.2f ns per wave-instr per SIMD (%.1f cycles @2.4GHz)\n", name, waves_per_simd, ms,
         ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
  hipFree(d);
}
int main() {
  for (int w : {1, 2, 4}) {
    run<0>("v_fma_f64", w); run<6>("2x v_fma_f64", w); run<1>("v_rcp_f64 + add", w); run<2>("cvt+v_rcp_f32+cvt+add", w);
    run<3>("v_sqrt_f64 + add", w); run<4>("v_rsq_f64 + add", w);
  }
  return 0;
}
