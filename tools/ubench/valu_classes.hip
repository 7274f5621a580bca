// micro-benchmark: issue cost (shader cycles per wave-instruction on one SIMD) of the VALU instruction
// classes the LBL kernels are made of, measured in-kernel with s_memtime at 1 / 3 waves per SIMD, plus
// the clock the chip holds while doing it (s_memtime ticks per s_memrealtime 100-MHz tick).
//   hipcc --offload-arch=gfx950 -O3 -o valu_classes tools/ubench/valu_classes.hip && ./valu_classes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

#define KERNEL(NAME, BODY)                                                                          \
  __global__ void NAME(unsigned long long* out, int iters, double seed) {                           \
    double a0 = seed + threadIdx.x * 1e-3, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5,     \
           a6 = a0 + 6, a7 = a0 + 7;                                                                \
    double b = 1.0000001, c = 1e-9;                                                                 \
    double e0 = a0 * 0.5, e1 = a1 * 0.5, e2 = a2 * 0.5, e3 = a3 * 0.5, e4 = a4 * 0.5, e5 = a5 * 0.5, e6 = a6 * 0.5, e7 = a7 * 0.5;                                                                 \
    float f0 = (float)a0, f1 = (float)a1, f2 = (float)a2, f3 = (float)a3, f4 = (float)a4, f5 = (float)a5,       \
          f6 = (float)a6, f7 = (float)a7;                                                           \
    int q0 = threadIdx.x, q1 = q0 + 1, q2 = q0 + 2, q3 = q0 + 3, q4 = q0 + 4, q5 = q0 + 5, q6 = q0 + 6, q7 = q0 + 7; \
    const double sc = seed * 0.5;  (void)sc;                                                        \
    const int qsrc = threadIdx.x * 3; (void)qsrc;                                                   \
    const unsigned long long smask = 0x5a5a5a5a5a5a5a5aull * (unsigned long long)(unsigned)iters; (void)smask;                                                        \
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();    \
    for (int it = 0; it < iters; ++it) { BODY BODY }                                                \
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();    \
    double s = e0 + e1 + e2 + e3 + e4 + e5 + e6 + e7 + a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7 + q0 + q1 + q2 + q3 + q4 + q5 + q6 + q7; \
    const int g = blockIdx.x * blockDim.x + threadIdx.x;                                            \
    if ((threadIdx.x & 63) == 0) { out[2 * (g / 64)] = t1 - t0; out[2 * (g / 64) + 1] = r1 - r0; }  \
    if (s == 123.456) out[0] = 0;                                                                   \
  }

#define A_FMA(n) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a##n) : "v"(b), "v"(c));
#define A_FMAS(n) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a##n) : "v"(b), "s"(sc));
#define A_MUL(n) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a##n) : "v"(b));
#define A_ADD(n) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a##n) : "v"(c));
#define A_MAX(n) asm volatile("v_max_f64 %0, %0, %1" : "+v"(a##n) : "v"(c));
#define A_RCP(n) asm volatile("v_rcp_f64 %0, %0" : "+v"(a##n));
#define A_RSQ(n) asm volatile("v_rsq_f64 %0, %0" : "+v"(a##n));
#define A_SQRT(n) asm volatile("v_sqrt_f64 %0, %0" : "+v"(a##n));
#define A_RNDNE(n) asm volatile("v_rndne_f64 %0, %0" : "+v"(a##n));
#define A_CVTI(n) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(q##n) : "v"(a##n));
#define A_CVTD(n) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(a##n) : "v"(q##n));
#define A_LDEXP(n) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(a##n) : "v"(q##n));
#define A_FRMANT(n) asm volatile("v_frexp_mant_f64 %0, %0" : "+v"(a##n));
#define A_FREXP(n) asm volatile("v_frexp_exp_i32_f64 %0, %1" : "=v"(q##n) : "v"(a##n));
#define A_CMP(n) asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(a##n), "v"(b) : "vcc");
#define A_CND(n) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(q##n) : "v"(qsrc), "s"(smask));
#define A_MOV(n) asm volatile("v_mov_b32 %0, %1" : "=v"(q##n) : "v"(q0));
#define A_MOV64(n) asm volatile("v_mov_b64 %0, %1" : "=v"(a##n) : "v"(b));
#define A_ADDU(n) asm volatile("v_add_u32 %0, %0, %1" : "+v"(q##n) : "v"(q0));
#define A_FMA32(n) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f##n) : "v"(f0));
#define A_PKFMA32(n) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(a##n) : "v"(b));
#define A_RCP32(n) asm volatile("v_rcp_f32 %0, %0" : "+v"(f##n));
#define A_EXP32(n) asm volatile("v_exp_f32 %0, %0" : "+v"(f##n));
#define A_CVT32(n) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f##n) : "v"(a##n));
#define A_CVT64(n) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a##n) : "v"(f##n));
#define A_RDLANE(n) asm volatile("v_readlane_b32 s20, %0, 3\n v_writelane_b32 %0, s20, 5" : "+v"(q##n) : : "s20");
#define A_FRACT(n) asm volatile("v_fract_f64 %0, %0" : "+v"(a##n));
#define A_FLOOR(n) asm volatile("v_floor_f64 %0, %0" : "+v"(a##n));
#define A_DIVFIX(n) asm volatile("v_div_fixup_f64 %0, %0, %1, %1" : "+v"(a##n) : "v"(b));

// the far-line body's mix: one reciprocal + six FMA-class instructions
#define A_MIX(n) asm volatile("v_rcp_f64 %0, %0\n v_fma_f64 %1, %1, %2, %3\n v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3\n" \
                              "v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3\n v_fma_f64 %0, %0, %2, %3" \
                              : "+v"(a##n), "+v"(e##n) : "v"(b), "v"(c));
#define A_DEP(n) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a0) : "v"(b), "v"(c));
KERNEL(k_fma, REP8(A_FMA))
KERNEL(k_mix, REP8(A_MIX))
KERNEL(k_dep, REP8(A_DEP))
KERNEL(k_fmas, REP8(A_FMAS))
KERNEL(k_mul, REP8(A_MUL))
KERNEL(k_add, REP8(A_ADD))
KERNEL(k_max, REP8(A_MAX))
KERNEL(k_rcp, REP8(A_RCP))
KERNEL(k_rsq, REP8(A_RSQ))
KERNEL(k_sqrt, REP8(A_SQRT))
KERNEL(k_rndne, REP8(A_RNDNE))
KERNEL(k_cvti, REP8(A_CVTI))
KERNEL(k_cvtd, REP8(A_CVTD))
KERNEL(k_ldexp, REP8(A_LDEXP))
KERNEL(k_frmant, REP8(A_FRMANT))
KERNEL(k_frexp, REP8(A_FREXP))
KERNEL(k_cmp, REP8(A_CMP))
KERNEL(k_cnd, REP8(A_CND))
KERNEL(k_mov, REP8(A_MOV))
KERNEL(k_mov64, REP8(A_MOV64))
KERNEL(k_addu, REP8(A_ADDU))
KERNEL(k_fma32, REP8(A_FMA32))
KERNEL(k_pkfma32, REP8(A_PKFMA32))
KERNEL(k_rcp32, REP8(A_RCP32))
KERNEL(k_exp32, REP8(A_EXP32))
KERNEL(k_cvt32, REP8(A_CVT32))
KERNEL(k_cvt64, REP8(A_CVT64))
KERNEL(k_rdlane, REP8(A_RDLANE))
KERNEL(k_fract, REP8(A_FRACT))
KERNEL(k_floor, REP8(A_FLOOR))
KERNEL(k_divfix, REP8(A_DIVFIX))

typedef void (*kern_t)(unsigned long long*, int, double);

static void run(const char* name, kern_t k, int waves_per_simd, int per_iter) {
  const int blocks = 256 * waves_per_simd;   // 256-thread blocks: 4 waves, one per SIMD, 256 CUs
  const int nw = blocks * 4;
  unsigned long long* d;
  hipMalloc(&d, sizeof(unsigned long long) * 2 * nw);
  const int iters = 4000;
  hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, 100, 2.0);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, iters, 2.0);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(2 * nw);
  hipMemcpy(h.data(), d, sizeof(unsigned long long) * 2 * nw, hipMemcpyDeviceToHost);
  std::vector<double> cyc(nw), clk(nw);
  for (int i = 0; i < nw; ++i) { cyc[i] = (double)h[2 * i]; clk[i] = h[2 * i + 1] ? (double)h[2 * i] / (double)h[2 * i + 1] * 100.0 : 0; }
  std::nth_element(cyc.begin(), cyc.begin() + nw / 2, cyc.end());
  std::nth_element(clk.begin(), clk.begin() + nw / 2, clk.end());
  const double n = (double)iters * 16 * per_iter;       // BODY BODY = 16 instances per iteration
  printf("%-28s waves/SIMD=%d  %7.3f ms  %6.2f cycles/instr/SIMD (in-kernel)  clock %.0f MHz\n", name, waves_per_simd, ms,
         cyc[nw / 2] / n / waves_per_simd, clk[nw / 2]);
  hipFree(d);
}

int main() {
  for (int w : {1, 3, 4}) {
    run("v_fma_f64", k_fma, w, 1); run("v_fma_f64 dependent chain", k_dep, w, 1);
    run("rcp + 6 fma (per 7 instr)", k_mix, w, 1); run("v_fma_f64 (SGPR src)", k_fmas, w, 1); run("v_mul_f64", k_mul, w, 1);
    run("v_add_f64", k_add, w, 1); run("v_max_f64", k_max, w, 1);
    run("v_rcp_f64", k_rcp, w, 1); run("v_rsq_f64", k_rsq, w, 1); run("v_sqrt_f64", k_sqrt, w, 1);
    run("v_rndne_f64", k_rndne, w, 1); run("v_floor_f64", k_floor, w, 1); run("v_fract_f64", k_fract, w, 1);
    run("v_cvt_i32_f64", k_cvti, w, 1); run("v_cvt_f64_i32", k_cvtd, w, 1);
    run("v_ldexp_f64", k_ldexp, w, 1); run("v_frexp_mant_f64", k_frmant, w, 1); run("v_frexp_exp_i32_f64", k_frexp, w, 1);
    run("v_div_fixup_f64", k_divfix, w, 1);
    run("v_cmp_lt_f64", k_cmp, w, 1); run("v_cndmask_b32", k_cnd, w, 1); run("v_mov_b32", k_mov, w, 1);
    run("v_mov_b64", k_mov64, w, 1); run("v_add_u32", k_addu, w, 1);
    run("v_readlane+v_writelane", k_rdlane, w, 2);
    run("v_fma_f32", k_fma32, w, 1); run("v_pk_fma_f32", k_pkfma32, w, 1); run("v_rcp_f32", k_rcp32, w, 1);
    run("v_exp_f32", k_exp32, w, 1); run("v_cvt_f32_f64", k_cvt32, w, 1); run("v_cvt_f64_f32", k_cvt64, w, 1);
  }
  return 0;
}
