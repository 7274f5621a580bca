// mwrt_kernels.hip.h -- hand-written gfx950 (CDNA4) kernels of the LBL forward operator.
//
// Mapping: one workgroup = one (profile, frequency-chunk); one LANE = one LEVEL
// of that profile.  Line tables are wave-uniform and travel through the scalar unit (s_load),
// the chunk's frequencies are broadcast-read from LDS, and every per-(level,line) transcendental
// is evaluated once per lane and reused for the NFC frequencies of the chunk.  Phase K1 leaves zenith layer optical depths and
// Planck functions in LDS; phase K2 integrates the slant-path RTE out of LDS for all
// (frequency, angle) pairs; nothing but the 4 profile fields and the TBs touches HBM.
//
// Arithmetic restates pyrtlib [EXT] (not in /root/reference): RTEquation.vapor,
// clearsky_absorption -> H2OAbsModel.h2o_absorption / O2AbsModel.o2_absorption /
// N2AbsModel.n2_absorption, exponential_integration, planck, bright -- the routines that
// TbCloudRTE.execute() runs when called from reference python_src/proc/PyRTlib_processing.py:126.
// fp64 throughout ("dtype": "f64"); no MFMA: this is elementwise + scan work.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/mwrt.h"

namespace mwrt {

constexpr int WAVE = 64;
constexpr double TAUMAX = 125.0;
constexpr double TRANS_MIN = 5.1664206328378610e-55;   // exp(-TAUMAX)

// Device image of the tables: the ABI record plus host-precomputed reciprocals.  It is read
// through a CONSTANT-address-space pointer: the tables never change while a kernel runs, and
// that is what lets the compiler fetch them with s_load (scalar cache, SGPR operands) instead of
// 64 identical vector loads per wave.
// Per-line records (array of structs): what the fused kernel's line loops read for line k sits in
// one contiguous 128-byte record, so the compiler fetches it with one or two wide s_load and a
// single wait per iteration instead of a dozen dwordx2 loads in three dependent groups.
struct O2Rec { double f, s300rf2, be, w300, y0, y1, g0, g1, dnu0, dnu1, pad[6]; };      // 128 B
struct H2ORec { double fl, s1, b2, w0, x, w0s, xs, sh, xh, shs, xhs, aair, aself, w2, pad[2]; };   // 128 B
struct ModelFlat : mwrt_model_desc {
  double o2_rf2[MWRT_MAX_O2_LINES];      // 1 / F_k^2
  O2Rec o2r[MWRT_MAX_O2_LINES];
  H2ORec h2or[MWRT_MAX_H2O_LINES];
};
typedef const __attribute__((address_space(4))) ModelFlat* cmodel;
typedef const __attribute__((address_space(4))) double* cdoubles;

struct LaunchGeom {          // host-computed K2 work split (see plan_k2 in mwrt.hip), per K2 pass
  int nseg[2];               // level segments per (freq, angle) pair
  int seglen[2];             // layers per segment
  int npart;                 // doubles of segment partials (B, T) the largest pass needs
  int ldrow;                 // padded LDS row length (doubles) of tau/boft: conflict-free for b64
  unsigned magic_nseg[2];    // ceil(2^32 / nseg), ceil(2^32 / nang): n / d = umulhi(n, magic) for n < 65536, 1 < d <= 1024
  unsigned magic_nang;       // (the work-item index splits cost two integer divisions per item otherwise)
};

// n / d for the small operands of the K2 work split (see LaunchGeom): one v_mul_hi_u32
__device__ __forceinline__ int div_small(int n, int d, unsigned magic) {
  return d == 1 ? n : (int)__umulhi((unsigned)n, magic);
}

// ---------------------------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------------------------
#ifndef MWRT_EXACT_DIV
#define MWRT_EXACT_DIV 0
#endif
// timing-only ablation builds (tools/ablate.sh): bit 1 skips the O2 line loop, 2 the H2O Lorentz
// loop, 4 the speed-dependent loop, 8 the K2 integration, 16 the layer step of the TAU absorption kernels, 32 their
// stores, 64 the scalar loads of their interpolation weights.  Always 0 in the shipped library.
#ifndef MWRT_ABLATE
#define MWRT_ABLATE 0
#endif
// diagnostic build (tools/phase_timeline.sh): lane 0 of every wave of k_tb_fused stamps the 100-MHz wall clock at its phase
// boundaries into FusedArgs::phase [workgroup][wave][10] (slots 8, 9: HW_ID and XCC_ID of the wave).  Always 0 in the shipped library.
#ifndef MWRT_PHASE_CLOCK
#define MWRT_PHASE_CLOCK 0
#endif
// Issue priority by phase.  A SIMD issues from its OLDEST ready wave first, so of the three or four workgroups a CU holds the
// first one dispatched runs almost as if alone and the last one gets the gaps: in the single resident round of the headline
// shape (1000 workgroups, four per CU, all started within 0.3 us) the phase stamps of a -DMWRT_PHASE_CLOCK=1 build show the
// four workgroups of every CU leaving at 70 / 81 / 108 / 111 us -- the last ones run their final 30 us with one or two waves
// per SIMD, latency-bound.  s_setprio beats age: a wave in an EARLIER phase gets the higher priority (3 water lines,
// 2 oxygen lines, 1 layer step + first RTE pass, 0 second RTE pass), so the laggards of a SIMD catch up at every phase
// change and all waves finish together (88 ... 101 us): 116 -> 105 us.  -DMWRT_NO_SETPRIO=1 builds without it (A/B timing).
#ifndef MWRT_NO_SETPRIO
#define MWRT_NO_SETPRIO 0
#endif
#if MWRT_NO_SETPRIO
#define MWRT_SETPRIO(n) do { } while (0)
#else
#define MWRT_SETPRIO(n) __builtin_amdgcn_s_setprio(n)
#endif
#if MWRT_PHASE_CLOCK
#define MWRT_STAMP(k) do { if (A.phase && lane == 0) A.phase[((int64_t)blockIdx.x * 4 + wave) * 10 + (k)] = (long long)wall_clock64(); } while (0)
#else
#define MWRT_STAMP(k) do { } while (0)
#endif

// x / d with v_rcp_f64 + two Newton steps (~1.5 ulp; parity bar is 1e-6 K, budget 0.01 K).
__device__ __forceinline__ double fdiv(double x, double d) {
#if MWRT_EXACT_DIV
  return x / d;
#else
  double r = __builtin_amdgcn_rcp(d);
  double e = __builtin_fma(-d, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-d, r, 1.0);
  r = __builtin_fma(r, e, r);
  return x * r;
#endif
}

// exp(x) for the kernels' bounded arguments: Cody-Waite reduction + a degree-11 polynomial on |r| <= ln2/2
// (1 + r + r^2 g(r), g fitted at the Chebyshev nodes of the interval: 1.7e-17 relative; the degree-13 Taylor
// series it replaces had 4e-18 and two more steps), scaled by v_ldexp_f64 (which also gives the right 0 / inf at
// the range ends).
// ocml's exp spends two VALU instructions per Horner step (v_mov of the 64-bit constant + v_fmac);
// here each constant rides in an SGPR pair (materialised by s_mov, off the VALU port), so a step
// is ONE v_fma_f64.  ~19 VALU instead of ~35; max relative error measured < 4e-16.
#define MWRT_FMA_SC(p, r, c) asm("v_fma_f64 %0, %1, %2, %3" : "=v"(p) : "v"(p), "v"(r), "s"(c))
__device__ __forceinline__ double fexp(double x) {
#if MWRT_EXACT_DIV
  return exp(x);
#else
  const double k = __builtin_rint(x * 1.4426950408889634074);
  double r = __builtin_fma(k, -6.93147180369123816490e-01, x);
  r = __builtin_fma(k, -1.90821492927058770002e-10, r);
  double p = 2.5100569275813683e-08;
  MWRT_FMA_SC(p, r, 2.762032742826824e-07);
  MWRT_FMA_SC(p, r, 2.75572680728901e-06);
  MWRT_FMA_SC(p, r, 2.4801520792572694e-05);
  MWRT_FMA_SC(p, r, 0.00019841269863303223);
  MWRT_FMA_SC(p, r, 0.0013888888917538296);
  MWRT_FMA_SC(p, r, 0.008333333333330011);
  MWRT_FMA_SC(p, r, 0.04166666666662348);
  MWRT_FMA_SC(p, r, 0.16666666666666669);
  MWRT_FMA_SC(p, r, 0.5000000000000001);
  p = __builtin_fma(p, r, 1.0);
  p = __builtin_fma(p, r, 1.0);
  return __builtin_amdgcn_ldexp(p, (int)k);
#endif
}

// (2 atanh(s)/s - 2)/z = 2/3 + 2z/5 + ..., z = s^2 <= 0.1716^2, as a degree-6 polynomial fitted at the Chebyshev nodes of
// the interval: 2 atanh(s)/s to 4.6e-18 relative in 7 steps (the Taylor series needs 10 for 5e-17)
__device__ __forceinline__ double two_atanh_tail(double z) {
  double p = 0.14616878919029822;
  MWRT_FMA_SC(p, z, 0.15331686868638428);
  MWRT_FMA_SC(p, z, 0.1818289017031397);
  MWRT_FMA_SC(p, z, 0.22222211120449298);
  MWRT_FMA_SC(p, z, 0.2857142862606338);
  MWRT_FMA_SC(p, z, 0.3999999999989931);
  MWRT_FMA_SC(p, z, 0.666666666666667);
  return p;
}

// log(x), x > 0 finite and normal (layer ratios of positive absorption coefficients): frexp to
// m in [sqrt(1/2), sqrt(2)), s = (m-1)/(m+1), log m = 2 s (1 + z/3 + z^2/5 + ...) = s (2 + z two_atanh_tail(z)), z = s^2
// (|s| <= 0.1716).  Keeps full RELATIVE accuracy as x -> 1, which is what the
// log-mean of two nearly equal levels needs.  ~33 VALU against ~50 for ocml's log.
__device__ __forceinline__ double flog(double x) {
#if MWRT_EXACT_DIV
  return log(x);
#else
  int e = __builtin_amdgcn_frexp_exp(x);
  double m = __builtin_amdgcn_frexp_mant(x);            // [0.5, 1)
  const bool lo = m < 0.70710678118654752440;
  m = lo ? m + m : m;
  e = lo ? e - 1 : e;
  const double num = m - 1.0, den = m + 1.0;
  double r = __builtin_amdgcn_rcp(den);
  r = __builtin_fma(r, __builtin_fma(-den, r, 1.0), r);
  r = __builtin_fma(r, __builtin_fma(-den, r, 1.0), r);
  const double s = num * r;
  const double z = s * s;
  const double p = two_atanh_tail(z);                   // (2 atanh(s)/s - 2) / z
  const double ed = (double)e;
  const double lm = __builtin_fma(s * z, p, s + s);     // log(m)
  return __builtin_fma(ed, 6.93147180369123816490e-01, __builtin_fma(ed, 1.90821492927058770002e-10, lm));
#endif
}

// exp(x) for |x| <= 1/8 with no range reduction: 1 + x + x^2 g(x), g of degree 7 fitted at the Chebyshev nodes
// (2.2e-18 relative).  9 VALU.
// Thin layers (tau * airmass <= 1/8) are the rule for the K-band channels at every level and angle.
constexpr double EXP_SMALL_X = 0.125;
__device__ __forceinline__ double fexp_small(double x) {
  double p = 2.756514908613403e-06;
  MWRT_FMA_SC(p, x, 2.4810200365624755e-05);
  MWRT_FMA_SC(p, x, 0.00019841269076602318);
  MWRT_FMA_SC(p, x, 0.001388888804772704);
  MWRT_FMA_SC(p, x, 0.008333333333357229);
  MWRT_FMA_SC(p, x, 0.04166666666692954);
  MWRT_FMA_SC(p, x, 0.16666666666666666);
  MWRT_FMA_SC(p, x, 0.4999999999999999);
  p = __builtin_fma(p, x, 1.0);
  return __builtin_fma(p, x, 1.0);
}

// tanh(x/2) = (1 - e^-x) / (1 + e^-x) for 0 <= x <= 1/8: x (1/2 + u g(u)), u = x^2, g of degree 3 fitted at the Chebyshev
// nodes of [0, 1/64] (6e-17 relative).  6 VALU; in the thin-layer RTE step it replaces 1 - E, 1 + E and their quotient (9 issue slots), and has none
// of the cancellation of 1 - E.
__device__ __forceinline__ double ftanh_half_small(double x) {
  const double u = x * x;
  double p = 4.257889640378761e-05;
  MWRT_FMA_SC(p, u, -0.0004216256671577941);
  MWRT_FMA_SC(p, u, 0.004166666662552237);
  MWRT_FMA_SC(p, u, -0.04166666666666466);
  p = __builtin_fma(p, u, 0.5);
  return p * x;
}

// ... and for |x| <= 1/64: exp to degree 6 (truncation 9e-18), tanh(x/2) through x^7 (next term 3e-21 relative). 7 + 5 VALU.
constexpr double EXP_TINY_X = 0.015625;
__device__ __forceinline__ double fexp_tiny(double x) {
  double p = 1.3888888888888889e-03;                 // 1/6!
  MWRT_FMA_SC(p, x, 8.3333333333333332e-03);         // 1/5!
  MWRT_FMA_SC(p, x, 4.1666666666666664e-02);         // 1/4!
  MWRT_FMA_SC(p, x, 1.6666666666666666e-01);         // 1/3!
  p = __builtin_fma(p, x, 0.5);
  p = __builtin_fma(p, x, 1.0);
  return __builtin_fma(p, x, 1.0);
}
__device__ __forceinline__ double ftanh_half_tiny(double x) {
  const double u = x * x;
  double p = -4.2162698412698413e-04;                // -17/40320
  MWRT_FMA_SC(p, u, 4.1666666666666666e-03);         // 1/240
  MWRT_FMA_SC(p, u, -4.1666666666666664e-02);        // -1/24
  p = __builtin_fma(p, u, 0.5);
  return p * x;
}

// Wave votes straight from the comparison mask: HIP's __all / __any take an int, and the compiler materialises it
// (v_cndmask 0/1, v_cmp_ne) before comparing with exec -- two VALU instructions and a VALU -> SALU hazard per vote.
typedef unsigned long long wmask;
__device__ __forceinline__ bool wave_all(bool p) { return __builtin_amdgcn_ballot_w64(!p) == 0ull; }
__device__ __forceinline__ bool wave_any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }
// ... and a conjunction of comparisons as the AND of their masks on the scalar unit (pass each comparison separately)
__device__ __forceinline__ wmask wballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
template <class... B>
__device__ __forceinline__ bool wave_all_of(B... b) { return (wballot(b) & ...) == wballot(true); }

// max over the 16 lanes of a DPP row (lanes 16k .. 16k+15), delivered to all of them: row_ror 8, 4, 2, 1.
// Four VALU instructions, no LDS crossbar.
__device__ __forceinline__ float row16_max(float m) {
  m = fmaxf(m, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(m), 0x128, 0xf, 0xf, false)));
  m = fmaxf(m, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(m), 0x124, 0xf, 0xf, false)));
  m = fmaxf(m, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(m), 0x122, 0xf, 0xf, false)));
  m = fmaxf(m, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(m), 0x121, 0xf, 0xf, false)));
  return m;
}
constexpr int K2_SORT_MIN_SEGLEN = 16;     // shorter segments: dealing the items out costs more than the thin step saves

// Planck function in pyrtlib's units, B = 1 / (exp(x) - 1), x = h f / (k T).  In the microwave x is a few
// 1e-3: when the whole wave has x <= 1/32, B = (1/x) * x/(e^x - 1) with the Bernoulli series
// x/(e^x - 1) = 1 - x/2 + x^2/12 - x^4/720 + x^6/30240 (next term 8e-19) and 1/x = (k T / h) * (1/f) from
// per-level and per-frequency factors the caller holds: 8 VALU, no exp, no reciprocal, and none of the
// cancellation of exp(x) - 1 (which costs pyrtlib itself ~3e-14 relative; far below the parity bar).
constexpr double PLANCK_SMALL_X = 0.03125;
__device__ __forceinline__ double planck_b(double x, double inv_x) {
  if (wave_all_of(x <= PLANCK_SMALL_X, x > 0.0)) {
    const double u = x * x;
    double g = 3.3068783068783071e-05;                 // 1/30240
    MWRT_FMA_SC(g, u, -1.3888888888888889e-03);        // -1/720
    MWRT_FMA_SC(g, u, 8.3333333333333329e-02);         // 1/12
    g = __builtin_fma(g, u, 1.0);
    g = __builtin_fma(x, -0.5, g);
    return g * inv_x;
  }
  return fdiv(1.0, fexp(x) - 1.0);
}

// inner-loop variant: one Newton step (v_rcp_f64 is good to ~2^-23, so ~2^-46 ~ 1.4e-14 relative)
__device__ __forceinline__ double fdiv1(double x, double d) {
#if MWRT_EXACT_DIV
  return x / d;
#else
  double r = __builtin_amdgcn_rcp(d);
  r = __builtin_fma(r, __builtin_fma(-d, r, 1.0), r);
  return x * r;
#endif
}

// Frequencies are wave-uniform.  Held in SGPRs, {f, f^2} x 14 is 56 scalar registers and the
// allocator spills them into VGPR lanes (v_readlane per use).  They live in LDS instead and are
// re-read by broadcast each line iteration; the fence stops the compiler hoisting the reads
// back into (vector) registers across the line loop.
#define LDS_RELOAD_FENCE() asm volatile("" ::: "memory")
// A wave-uniform if / else whose sides are both free of side effects gets flattened by the optimiser into "evaluate
// both, select" -- the opposite of what a wave vote is for.  An empty volatile asm cannot be speculated: placed at the
// top of each side it keeps the branch a branch.
#define KEEP_BRANCH() asm volatile("")

// "Far" lines: every frequency of the chunk is at least FAR_MIN_GHZ (+ the shift allowance) away
// from the line centre, so D1*D2 may be formed as a polynomial in f^2 without harmful cancellation.
constexpr double FAR_MIN_GHZ = 0.2;
constexpr double FAR_SHIFT_GHZ = 0.05;     // O2: |dnu| allowance, checked per line by wave vote
constexpr double FAR_H2O_GHZ = 5.0;        // H2O: covers any pressure shift (< 1 GHz) with margin

// Wave-uniform bit sets over line indices (line k of the table against the chunk's frequencies).  They steer the line
// loops: each loop walks ONE set with ONE loop body, so the NFC accumulators never cross a
// control-flow join between differently allocated variants (the v_mov copies that cost).
struct LineMasks {
  unsigned long long o2_far;   // every chunk frequency >= FAR_MIN_GHZ + FAR_SHIFT_GHZ from the line centre
  unsigned h2o_far;            // ... >= FAR_H2O_GHZ from the line centre
  unsigned h2o_none;           // both Lorentz terms beyond the 750-GHz cutoff for every frequency (FAR_H2O_GHZ margin)
  unsigned h2o_res;            // negative-frequency term beyond the cutoff for every frequency (e.g. 752 GHz from 22 GHz)
  unsigned h2o_sd;             // speed-dependent lines (W2 > 0)
  unsigned h2o_sdfar;          // ... of those, the ones whose special shape (inside 10 half-widths) cannot reach any frequency of
                               // the chunk by the host's bound: treated as plain lines, re-checked per level (wave vote)
  unsigned h2o_sdint;          // speed-dependent lines far enough from the chunk (>= 3 GHz and 5 spans) for the half-sampled shape
  unsigned h2o_vfar;           // "very far" lines: summed as ONE Taylor polynomial in f^2 about the chunk's middle (vfar_add)
  unsigned long long o2_vfar;
  double vf_u0, vf_h, vf_invh; // middle and half range of the chunk's f^2 values [GHz^2] (vf_h >= 1), 1 / vf_h
};

// The sets depend on the chunk's frequencies and the table only: the host computes them once per (model, frequency
// list, chunk width) -- csrc/mwrt.hip chunk_masks() -- and the kernels fetch their chunk's record through the scalar
// cache (round 2 had every workgroup derive them: ~180 VALU per lane and chunk).
typedef const __attribute__((address_space(4))) LineMasks* cmasks;
__device__ __forceinline__ LineMasks load_masks(const LineMasks* table, int chunk) {
  const cmasks q = (cmasks)table + chunk;
  LineMasks lm;
  lm.o2_far = q->o2_far; lm.h2o_far = q->h2o_far; lm.h2o_none = q->h2o_none; lm.h2o_res = q->h2o_res; lm.h2o_sd = q->h2o_sd;
  lm.h2o_sdfar = q->h2o_sdfar; lm.h2o_sdint = q->h2o_sdint;
  lm.h2o_vfar = q->h2o_vfar; lm.o2_vfar = q->o2_vfar; lm.vf_u0 = q->vf_u0; lm.vf_h = q->vf_h; lm.vf_invh = q->vf_invh;
  return lm;
}

struct cplx { double re, im; };
__device__ __forceinline__ cplx cmul(cplx a, cplx b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
__device__ __forceinline__ cplx cadd(cplx a, double r) { return {a.re + r, a.im}; }
// 1 / b for complex b (one real reciprocal): the speed-dependent shape divides twice by the same
// per-(level, line) quantity, so the loop body multiplies by this instead
__device__ __forceinline__ cplx crecip(cplx b) {
  const double d = __builtin_fma(b.re, b.re, b.im * b.im);
#if MWRT_EXACT_DIV
  const double r = 1.0 / d;
#else
  double r = __builtin_amdgcn_rcp(d);
  r = __builtin_fma(r, __builtin_fma(-d, r, 1.0), r);
  r = __builtin_fma(r, __builtin_fma(-d, r, 1.0), r);
#endif
  return {b.re * r, -b.im * r};
}
__device__ __forceinline__ cplx cdiv(cplx a, cplx b) { return cmul(a, crecip(b)); }
// sqrt(x), x > 0 finite and far from the denormal range: v_rsq_f64 seed (~2^-23) + one coupled
// Newton step on (g ~ sqrt x, h ~ 1/(2 sqrt x)) -> ~2^-45 relative
__device__ __forceinline__ double fsqrt(double x) {
#if MWRT_EXACT_DIV
  return sqrt(x);
#else
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y;
  const double h = 0.5 * y;
  return __builtin_fma(__builtin_fma(-g, g, x), h, g);      // x == 0 gives NaN: callers never use that lane
#endif
}
__device__ __forceinline__ cplx csqrt_principal(cplx z) {
  const double r = fsqrt(__builtin_fma(z.re, z.re, z.im * z.im));
  // one square root and one division, selected by the sign of Re z:
  //   Re z >= 0: a = sqrt((r + Re z)/2), result (a, Im z / 2a);  Re z < 0: b = sqrt((r - Re z)/2), result (|Im z| / 2b, +-b)
  const bool pos = z.re >= 0.0;
  const double a = fsqrt(0.5 * (r + fabs(z.re)));
  const double q = fdiv1(pos ? z.im : fabs(z.im), a + a);    // z == 0 is outside the SD shape's domain (Re Xc > 0)
  return pos ? cplx{a, q} : cplx{q, copysign(a, z.im)};
}

// Rosenkranz DCERROR [EXT]: Hui, Armstrong & Wray (1978) rational approximation of the complex
// error function, upper half plane (y >= 0 always holds here: y = Re(principal sqrt)).
__device__ __forceinline__ cplx dcerror_upper(double x, double y) {
  const double a0 = 122.607931777104326, a1 = 214.382388694706425, a2 = 181.928533092181549,
               a3 = 93.155580458138441, a4 = 30.180142196210589, a5 = 5.912626209773153,
               a6 = 0.564189583562615;
  const double b0 = 122.607931773875350, b1 = 352.730625110963558, b2 = 457.334478783897737,
               b3 = 348.703917719495792, b4 = 170.354001821091472, b5 = 53.992906912940207,
               b6 = 10.479857114260399;
  const cplx zh = {fabs(y), -x};
  // Both polynomials have REAL coefficients: at a complex point they cost two real FMAs per coefficient (instead of
  // the four of a complex Horner step) through the quadratic z^2 = r z - s, r = 2 Re z, s = |z|^2:
  //   b_n = a_n,  b_{n-1} = a_{n-1} + r b_n,  b_k = a_k + r b_{k+1} - s b_{k+2},  p(z) = a_0 + z b_1 - s b_2
  // (agrees with complex Horner to < 5e-15 relative over |z| <= 100; the rational itself is Hui's, ~1e-6).
  const double r = zh.re + zh.re;
  const double ms = -__builtin_fma(zh.re, zh.re, zh.im * zh.im);
  auto step = [&](double c, double b1, double b2) -> double { return __builtin_fma(r, b1, __builtin_fma(ms, b2, c)); };
  double n2 = a6, n1 = __builtin_fma(r, a6, a5), nt;
  nt = step(a4, n1, n2); n2 = n1; n1 = nt;
  nt = step(a3, n1, n2); n2 = n1; n1 = nt;
  nt = step(a2, n1, n2); n2 = n1; n1 = nt;
  nt = step(a1, n1, n2); n2 = n1; n1 = nt;
  const cplx as = {__builtin_fma(zh.re, n1, __builtin_fma(ms, n2, a0)), zh.im * n1};
  double d2 = 1.0, d1 = r + b6, dt;
  dt = __builtin_fma(r, d1, ms + b5); d2 = d1; d1 = dt;
  dt = step(b4, d1, d2); d2 = d1; d1 = dt;
  dt = step(b3, d1, d2); d2 = d1; d1 = dt;
  dt = step(b2, d1, d2); d2 = d1; d1 = dt;
  dt = step(b1, d1, d2); d2 = d1; d1 = dt;
  const cplx bs = {__builtin_fma(zh.re, d1, __builtin_fma(ms, d2, b0)), zh.im * d1};
  return cdiv(as, bs);
}

// deterministic workgroup sum (fixed order: lanes by butterfly, then waves in index order)
__device__ __forceinline__ double block_sum(double v, double* scratch /*[nwaves]*/, int tid, int nthreads) {
#pragma unroll
  for (int o = WAVE / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
  __syncthreads();
  if ((tid & (WAVE - 1)) == 0) scratch[tid / WAVE] = v;
  __syncthreads();
  double s = 0.0;
  for (int w = 0; w < nthreads / WAVE; ++w) s += scratch[w];
  return s;
}

// ---------------------------------------------------------------------------------------------
// per-level state shared by the H2O / O2 / N2 evaluations (RTEquation.vapor +
// clearsky_absorption preamble [EXT], incl. pyrtlib's kPa round trip)
// ---------------------------------------------------------------------------------------------
struct LevelState {
  double t;       // K    (300/(300/tk))
  double p;       // hPa  ((pdrykpa+ekpa)*10)
  double rho;     // g m-3
  double pdry;    // hPa  (pdrykpa*10)
};

__device__ __forceinline__ double goff_gratch_e(double tk, double rh) {
  const double LN10 = 2.302585092994045684;
  const double INV_LN10 = 0.434294481903251828;
  double y = 373.16 / tk;
  double es = -7.90298 * (y - 1.0) + 5.02808 * (flog(y) * INV_LN10)
            - 1.3816e-07 * (fexp(LN10 * (11.344 * (1.0 - (1.0 / y)))) - 1.0)
            + 0.0081328 * (fexp(LN10 * (-3.49149 * (y - 1.0))) - 1.0) + 3.0057148979490314 /*log10(1013.246)*/;
  return rh * fexp(LN10 * es);
}

__device__ __forceinline__ LevelState level_state(double p_hpa, double tk, double e) {
  const double rvap = (0.01 * 8.314510) / 18.01528;
  double v = 300.0 / tk;
  double ekpa = e / 10.0;
  double pdrykpa = p_hpa / 10.0 - ekpa;
  LevelState s;
  s.t = 300.0 / v;
  s.p = (pdrykpa + ekpa) * 10.0;
  s.rho = ekpa * 10.0 / (rvap * s.t);
  s.pdry = pdrykpa * 10.0;
  return s;
}

// ---------------------------------------------------------------------------------------------
// K1a: H2O lines + continuum for NFC uniform frequencies (H2OAbsModel.h2o_absorption [EXT])
//
// Per line the two Lorentz terms share ONE reciprocal:
//   s*[m1*(w/(D1) - base) + m2*(w/(D2) - base)] = (s w) (m1 D2 + m2 D1)/(D1 D2) - (m1+m2)(s base)
// with m = 1.0/0.0 for the 750-GHz cutoff.  The speed-dependent 22/183-GHz resonant term is
// evaluated in a second, short loop over the SD lines only (keeps the hot loop branch-free).
// ---------------------------------------------------------------------------------------------
struct H2OLine {          // per-(level, line) quantities, frequency independent
  double c1;              // line centre + pressure shift
  double w0, wsq;         // half width, squared
  double sw;              // S/fl^2 * w0
  double sbase;           // S/fl^2 * base
  double s;               // S/fl^2
  double base;
};

__device__ __forceinline__ H2OLine h2o_line(cmodel M, int k, double pda, double pvap, double ti, double tiln,
                                            double ti2, bool shifted) {
  H2OLine q;
  const auto& R = M->h2or[k];
  const double fl = R.fl;
  q.w0 = R.w0 * pda * fexp(R.x * tiln) + R.w0s * pvap * fexp(R.xs * tiln);
  double shift = 0.0;
  if (shifted) {
    // exponents / ln-T coefficients that are zero in the table cost nothing (uniform branches)
    const double xh = R.xh, xhs = R.xhs, aa = R.aair, as = R.aself;
    double sf = R.sh * pda, ss = R.shs * pvap;
    if (aa != 0.0) sf *= (1.0 - aa * tiln);
    if (as != 0.0) ss *= (1.0 - as * tiln);
    if (xh != 0.0) sf *= fexp(xh * tiln);
    if (xhs != 0.0) ss *= fexp(xhs * tiln);
    shift = sf + ss;
  }
  q.wsq = q.w0 * q.w0;
  q.s = R.s1 * ti2 * fexp(R.b2 * (1.0 - ti));                 // R.s1 = S1 / fl^2: the f^2 is applied at the end
  q.base = fdiv1(q.w0, 562500.0 + q.wsq);
  q.c1 = fl + shift;
  q.sw = q.s * q.w0;
  q.sbase = q.s * q.base;
  return q;
}

// Far-line bodies shared by the H2O and O2 loops: the two Lorentz terms of a line whose centre is
// far from every frequency of the chunk collapse to one rational function of f^2,
//   (f^2 P + Q) / (f^4 + A2 f^2 + Bc),   A2 = 2 (w^2 - c^2),  Bc = (c^2 + w^2)^2
// (cancellation in the denominator <= f^2 / (4 FAR_MIN^2) ulp ~ 2e-12).
struct FarLine { double P, Q, A2, Bc; };

// FOUR far lines per frequency through ONE reciprocal:
//   n0/d0 + n1/d1 + n2/d2 + n3/d3 = ((n0 d1 + n1 d0) d2 d3 + (n2 d3 + n3 d2) d0 d1) / (d0 d1 d2 d3).
// v_rcp_f64 costs about three FMA issue slots and delivers 2^-23, so a reciprocal + Newton step is 5 of
// the 9 slots a line-frequency term costs on its own; shared by four lines the term costs 6.6.
// 24 FMA-class instructions + 1 rcp per frequency (products stay < 1e48 for centres <= 1 THz).
// A wave issues in order, and a SIMD holds only three of these waves: the reciprocal and the Newton step that end a
// frequency are a serial tail, and the compiler (scheduling for register pressure) runs one frequency after the other.
// Here two frequencies go side by side and the loop is software-pipelined by hand: the reciprocals of one pair are issued,
// then the 52 independent instructions of the NEXT pair's numerators and denominators, then the first pair's Newton step and
// accumulation; the f^2 values are read from LDS two pairs ahead.  Scheduling barriers keep the compiler from undoing it.
#define MWRT_STAGE() __builtin_amdgcn_sched_barrier(0)
template <int NFC>
__device__ __forceinline__ void far_quad_accumulate(const double* sfq, const FarLine& a, const FarLine& b,
                                                    const FarLine& c, const FarLine& d, double (&sum)[NFC]) {
  static_assert(NFC % 2 == 0, "frequencies are taken in pairs");
  auto front = [&](const double (&f2s)[2], double (&den)[2], double (&num)[2]) {
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const double f2 = f2s[g];
      const double d0 = __builtin_fma(f2, f2 + a.A2, a.Bc);
      const double d1 = __builtin_fma(f2, f2 + b.A2, b.Bc);
      const double d2 = __builtin_fma(f2, f2 + c.A2, c.Bc);
      const double d3 = __builtin_fma(f2, f2 + d.A2, d.Bc);
      const double n0 = __builtin_fma(f2, a.P, a.Q);
      const double n1 = __builtin_fma(f2, b.P, b.Q);
      const double n2 = __builtin_fma(f2, c.P, c.Q);
      const double n3 = __builtin_fma(f2, d.P, d.Q);
      const double p01 = d0 * d1, p23 = d2 * d3;
      const double m01 = __builtin_fma(n0, d1, n1 * d0);
      const double m23 = __builtin_fma(n2, d3, n3 * d2);
      den[g] = p01 * p23;
      num[g] = __builtin_fma(m01, p23, m23 * p01);
    }
  };
  double den[2], num[2], r[2];
  double f2c[2] = {sfq[1], sfq[3]};
  double f2n[2] = {sfq[(NFC > 2) ? 5 : 1], sfq[(NFC > 2) ? 7 : 3]};
  front(f2c, den, num);
  r[0] = __builtin_amdgcn_rcp(den[0]); r[1] = __builtin_amdgcn_rcp(den[1]);
#pragma unroll
  for (int j = 0; j < NFC; j += 2) {
    double den_n[2] = {1.0, 1.0}, num_n[2] = {0.0, 0.0}, f2p[2] = {0.0, 0.0};
    if (j + 4 < NFC) { f2p[0] = sfq[2 * (j + 4) + 1]; f2p[1] = sfq[2 * (j + 5) + 1]; }      // in flight for two trips
    MWRT_STAGE();
    if (j + 2 < NFC) front(f2n, den_n, num_n);
    MWRT_STAGE();
    const double e0 = __builtin_fma(-den[0], r[0], 1.0), e1 = __builtin_fma(-den[1], r[1], 1.0);
    r[0] = __builtin_fma(r[0], e0, r[0]); r[1] = __builtin_fma(r[1], e1, r[1]);
    sum[j] = __builtin_fma(num[0], r[0], sum[j]); sum[j + 1] = __builtin_fma(num[1], r[1], sum[j + 1]);
    if (j + 2 < NFC) {
      den[0] = den_n[0]; den[1] = den_n[1]; num[0] = num_n[0]; num[1] = num_n[1];
      r[0] = __builtin_amdgcn_rcp(den[0]); r[1] = __builtin_amdgcn_rcp(den[1]);
      f2n[0] = f2p[0]; f2n[1] = f2p[1];
    }
  }
  MWRT_STAGE();
}

// VERY far lines: a line whose poles in u = f^2 (u ~ c^2 -+ 2 i c w) lie at >= 1/VF_RATIO_MAX half ranges from the middle u0
// of the chunk's f^2 values -- the submillimetre lines seen from a 22-58 GHz chunk -- is analytic across the chunk with
// room to spare: its term (P u + Q)/(u^2 + A2 u + Bc) is expanded in x = (u - u0)/h, |x| <= 1,
//   d(u) = d0 + d1 h x + h^2 x^2,   c_0 = n(u0)/d0,  c_1 = (P h - d1 h c_0)/d0,  c_j = -(d1 h c_{j-1} + h^2 c_{j-2})/d0,
// and ALL such lines of a species share one polynomial: ~40 instructions per line instead of 7 per line and frequency, plus
// one Horner evaluation per frequency.  Truncation after x^7: sum_{j>=8} (j+1) r^j <= 4e-14 of the line's own term at
// r = VF_RATIO_MAX (the poles' distance ratio), and such a line is a few percent of the absorption at most; the host picks
// the lines (chunk_masks) with the shift / width allowances.
constexpr int VF_TERMS = 8;
constexpr double VF_RATIO_MAX = 0.016;
constexpr int VF_MIN_FREQS = 7;              // ... and chunks with fewer frequencies than this are served directly
constexpr int VF_MIN_LINES = 4;              // fewer lines than this do not pay for the Horner pass (8 per frequency)
__device__ __forceinline__ void vfar_add(const FarLine& fl, double u0, double h, double (&acc)[VF_TERMS]) {
  const double d0 = __builtin_fma(u0, u0 + fl.A2, fl.Bc);
  const double d1h = __builtin_fma(fl.A2, h, (2.0 * u0) * h);
  double rd = __builtin_amdgcn_rcp(d0);
  rd = __builtin_fma(rd, __builtin_fma(-d0, rd, 1.0), rd);
  const double a = -d1h * rd, b = (-(h * h)) * rd;
  double cm2 = __builtin_fma(fl.P, u0, fl.Q) * rd;
  double cm1 = __builtin_fma(a, cm2, (fl.P * h) * rd);
  acc[0] += cm2;
  acc[1] += cm1;
#pragma unroll
  for (int j = 2; j < VF_TERMS; ++j) {
    const double cj = __builtin_fma(a, cm1, b * cm2);
    acc[j] += cj;
    cm2 = cm1; cm1 = cj;
  }
}
template <int NFC>
__device__ __forceinline__ void vfar_eval(const double* sfq, double invh, double mu /* = -u0 / h */, const double (&acc)[VF_TERMS], double (&sum)[NFC]) {
  // the Horner chains of several frequencies side by side (each is VF_TERMS - 1 dependent FMAs)
  constexpr int G = (NFC % 7 == 0) ? 7 : ((NFC % 4 == 0) ? 4 : 2);
  static_assert(NFC % G == 0, "group width");
#pragma unroll
  for (int j0 = 0; j0 < NFC; j0 += G) {
    double x[G], p[G];
#pragma unroll
    for (int g = 0; g < G; ++g) { x[g] = __builtin_fma(sfq[2 * (j0 + g) + 1], invh, mu); p[g] = acc[VF_TERMS - 1]; }
#pragma unroll
    for (int k = VF_TERMS - 2; k >= 0; --k) {
#pragma unroll
      for (int g = 0; g < G; ++g) p[g] = __builtin_fma(p[g], x[g], acc[k]);
      MWRT_STAGE();
    }
#pragma unroll
    for (int g = 0; g < G; ++g) sum[j0 + g] += p[g];
  }
}

// ... and TWO far lines through one reciprocal (what a quad loop leaves over, when it leaves two or three)
template <int NFC>
__device__ __forceinline__ void far_pair_accumulate(const double* sfq, const FarLine& a, const FarLine& b, double (&sum)[NFC]) {
#pragma unroll
  for (int j = 0; j < NFC; ++j) {
    const double f2 = sfq[2 * j + 1];
    const double d0 = __builtin_fma(f2, f2 + a.A2, a.Bc);
    const double d1 = __builtin_fma(f2, f2 + b.A2, b.Bc);
    const double n0 = __builtin_fma(f2, a.P, a.Q);
    const double n1 = __builtin_fma(f2, b.P, b.Q);
    const double den = d0 * d1;
    const double num = __builtin_fma(n0, d1, n1 * d0);
    double r = __builtin_amdgcn_rcp(den);
    r = __builtin_fma(r, __builtin_fma(-den, r, 1.0), r);
    sum[j] = __builtin_fma(num, r, sum[j]);
  }
}

// lowest `n` set bits of `m` (n < 4): the lines a quad loop leaves to the general loop
__device__ __forceinline__ unsigned long long lowest_bits(unsigned long long m, int n) {
  unsigned long long out = 0;
  for (int i = 0; i < n; ++i) { const unsigned long long b = m & (0ull - m); out |= b; m ^= b; }
  return out;
}

// Half-sampled speed-dependent shape (16-frequency chunks of a fine grid, chunk >= 3 GHz and 5 spans from the line centre).
// The SD resonant shape costs ~92 VALU per (level, frequency); its DIFFERENCE from the Lorentzian it replaces is small
// (<= 2 % of it) and smooth across a chunk, so it is evaluated at 9 of the 16 frequencies (slots 0, 2, ..., 14 and 15) and
// interpolated to the other 7 with host-computed Lagrange weights: error <= 4e-12 of the line's Lorentzian
// (probe on the oracle's formulas), against the 1e-10 the windowed path works to.  Where a lane is inside 10
// half-widths the odd slots then get Lorentzian + interpolated difference; outside, the plain Lorentzian as always.
constexpr int SD_NODES = 9, SD_TARGETS = 7;
__host__ __device__ constexpr int sd_node_slot(int n) { return n < 8 ? 2 * n : 15; }

// Window mode (k_absorb_win, fine spectral grids).  The lines far from a whole WINDOW of chunks are summed at a few
// Chebyshev nodes of the window (NODES = true: raw line sums out, no continuum, no SD lines; a line whose per-lane
// vote fails is reported in *failed and left out) and interpolated to each chunk's frequencies; the chunk call then
// starts from those sums (init_sum, init_bsum) and skips the lines in `excl`.  Outside window mode: excl = 0, null.
template <int NFC, bool NODES = false>
__device__ __forceinline__ void h2o_absorb(cmodel M, const LevelState& L, const double* sfq /*LDS: {f, f^2} per slot*/,
                                           const LineMasks& lm, double (&awet)[NFC], unsigned excl = 0u,
                                           const double* init_sum = nullptr, double init_bsum = 0.0,
                                           unsigned* failed = nullptr, double* bsum_out = nullptr,
                                           cdoubles sdw = nullptr /* [SD_TARGETS][SD_NODES] weights of this chunk, or null */) {
  const double t = L.t;
  const double pvap = fdiv(L.rho * t, M->h2o_pvap_div);
  const double pda = L.p - pvap;
  const double den = M->h2o_den_coef * L.rho;
  const double lnc = flog(fdiv(M->h2o_reftcon, t));
  const double con0 = (M->h2o_cf * pda * fexp(M->h2o_xcf * lnc) + M->h2o_cs * pvap * fexp(M->h2o_xcs * lnc)) * pvap;
  const double ti = fdiv(M->h2o_reftline, t);
  const double tiln = flog(ti);
  const double ti2 = fexp(2.5 * tiln);
  const bool shifted = M->h2o_shift_mode != 0;
  double sum[NFC];
#pragma unroll
  for (int j = 0; j < NFC; ++j) sum[j] = init_sum ? init_sum[j] : 0.0;

  const int nl = M->n_h2o;
  const unsigned all = ((nl >= 32) ? 0xffffffffu : ((1u << nl) - 1u)) & ~excl;
  // The 750-GHz cutoff of each Lorentz term depends on the lane only through the (tiny) pressure
  // shift.  Three loops, each with ONE body:
  //   A  far lines (table centre >= FAR_H2O_GHZ from every frequency), both terms inside the cutoff for
  //      every lane (wave vote at the chunk's extreme frequencies): the rational form, no masks.
  //      A line that fails the vote is handed to loop B.
  //   B  everything else that is not speed dependent: resonant-only / near-centre / masked forms.
  //   C  speed-dependent lines (22 / 183 GHz in R20SD+): Lorentz pair + the SD resonant shape.
  // Lines whose two terms are beyond the cutoff for every frequency (e.g. 916 GHz from 22 GHz) are skipped.
  const double fmin = sfq[2 * NFC], fmax = sfq[2 * NFC + 1];
  double bsum = init_bsum;                                    // sum of (count * s * base), frequency independent
  const unsigned sd_eff = lm.h2o_sd & ~lm.h2o_sdfar;          // lines that go to the speed-dependent loop straight away
  unsigned sd_extra = 0u;                                     // ... and the "out of reach" ones a level of this wave takes back
  unsigned deferred = (~lm.h2o_far | lm.h2o_res) & ~sd_eff & ~lm.h2o_none & all;
  const unsigned setA = (MWRT_ABLATE & 2) ? 0u : (lm.h2o_far & ~lm.h2o_res & ~sd_eff & ~lm.h2o_none & all);
  // ... of which the very far ones go through one Taylor polynomial (vfar_add) and the rest
  // FOUR at a time through far_quad_accumulate; the count mod 4 left over joins loop B
  const unsigned setV = NODES ? 0u : (setA & lm.h2o_vfar);
  const unsigned setQ = setA & ~setV;
  const unsigned leftA = (unsigned)lowest_bits(setQ, __builtin_popcount(setQ) & 3);
  const unsigned leftP = (__builtin_popcount(leftA) >= 2) ? (unsigned)lowest_bits(leftA, 2) : 0u;      // ... two of them as a pair
  deferred |= leftA & ~leftP;
  auto far_setup = [&](int k, FarLine& fl) {
    const H2OLine q = h2o_line(M, k, pda, pvap, ti, tiln, ti2, shifted);
    // both terms inside the cutoff for every lane?
    const bool plain = wave_all_of(q.c1 - fmin < 750.0, fmax - q.c1 < 750.0, q.c1 - fmin > -750.0, fmax + q.c1 < 750.0, fmin + q.c1 > -750.0);
    // both terms in:  s w (D1 + D2)/(D1 D2) - 2 s base,  D1 + D2 = 2 f^2 + 2 (c^2 + w^2)
    const double cc = __builtin_fma(q.c1, q.c1, q.wsq);
    fl.A2 = 2.0 * __builtin_fma(-q.c1, q.c1, q.wsq);
    fl.Bc = cc * cc;
    double P = 2.0 * q.sw, bs = 2.0 * q.sbase;
    // a speed-dependent line is a plain line only where its special shape (inside 10 half-widths, ABH2O_SD) is out of
    // reach of every frequency in [fmin, fmax] at this level
    const bool sdline = M->h2o_w2[k] > 0.0;
    const bool reach = sdline && !wave_all(10.0 * q.w0 < ::fmin(fabs(q.c1 - fmin), fabs(q.c1 - fmax)));
    if (reach || !plain) {                                                // SD within reach / cutoff not uniform
      if constexpr (NODES) *failed |= 1u << k;
      else if (reach) sd_extra |= 1u << k;                                       // ... the speed-dependent loop's job
      else deferred |= 1u << k;                                                  // ... loop B's job
      P = 0.0; bs = 0.0;
    }
    fl.P = P;
    fl.Q = P * cc;
    bsum += bs;
  };
  if (setV) {
    double acc[VF_TERMS];
#pragma unroll
    for (int j = 0; j < VF_TERMS; ++j) acc[j] = 0.0;
    for (unsigned m = setV; m; m &= m - 1u) {
      FarLine q;
      far_setup(__builtin_ctz(m), q);
      vfar_add(q, lm.vf_u0, lm.vf_h, acc);
    }
    LDS_RELOAD_FENCE();
    vfar_eval<NFC>(sfq, lm.vf_invh, -lm.vf_u0 * lm.vf_invh, acc, sum);
  }
  for (unsigned m = setQ & ~leftA; m;) {
    FarLine q0, q1, q2, q3;
    far_setup(__builtin_ctz(m), q0); m &= m - 1u;
    far_setup(__builtin_ctz(m), q1); m &= m - 1u;
    far_setup(__builtin_ctz(m), q2); m &= m - 1u;
    far_setup(__builtin_ctz(m), q3); m &= m - 1u;
    LDS_RELOAD_FENCE();
    far_quad_accumulate<NFC>(sfq, q0, q1, q2, q3, sum);
  }
  if (leftP) {
    FarLine q0, q1;
    unsigned m = leftP;
    far_setup(__builtin_ctz(m), q0); m &= m - 1u;
    far_setup(__builtin_ctz(m), q1);
    LDS_RELOAD_FENCE();
    far_pair_accumulate<NFC>(sfq, q0, q1, sum);
  }
  if (MWRT_ABLATE & 2) deferred = 0u;
  for (unsigned m = deferred; m; m &= m - 1u) {
    const int k = __builtin_ctz(m);
    const H2OLine q = h2o_line(M, k, pda, pvap, ti, tiln, ti2, shifted);
    const bool d1_in = (q.c1 - fmin < 750.0) && (fmax - q.c1 < 750.0) && (q.c1 - fmin > -750.0);
    const bool d1_out = (q.c1 - fmax >= 750.0) || (fmin - q.c1 >= 750.0);
    const bool d2_in = fmax + q.c1 < 750.0 && fmin + q.c1 > -750.0;
    const bool d2_out = fmin + q.c1 >= 750.0;
    if (wave_all(d1_out && d2_out)) continue;
    if (M->h2o_w2[k] > 0.0 && !wave_all(10.0 * q.w0 < ::fmin(fabs(q.c1 - fmin), fabs(q.c1 - fmax)))) {
      // a speed-dependent line within reach of its special shape: not a plain line at this level
      if constexpr (NODES) *failed |= 1u << k; else sd_extra |= 1u << k;
      continue;
    }
    LDS_RELOAD_FENCE();
    if (wave_all(d1_in && d2_in)) {                               // next to a line centre: detunings formed directly
      bsum = __builtin_fma(2.0, q.sbase, bsum);
#pragma unroll
      for (int j = 0; j < NFC; ++j) {
        const double f = sfq[2 * j];
        const double d1 = f - q.c1;
        const double d2 = f + q.c1;
        const double D1 = __builtin_fma(d1, d1, q.wsq);
        const double D2 = __builtin_fma(d2, d2, q.wsq);
        const double den12 = D1 * D2;
        double r = __builtin_amdgcn_rcp(den12);
        r = __builtin_fma(r, __builtin_fma(-den12, r, 1.0), r);
        sum[j] = __builtin_fma((D1 + D2) * r, q.sw, sum[j]);
      }
    } else if (wave_all(d1_in && d2_out)) {                       // resonant term only (e.g. 752 GHz seen from 22 GHz)
      bsum += q.sbase;
#pragma unroll
      for (int j = 0; j < NFC; ++j) {
        const double d1 = sfq[2 * j] - q.c1;
        const double D1 = __builtin_fma(d1, d1, q.wsq);
        double r = __builtin_amdgcn_rcp(D1);
        r = __builtin_fma(r, __builtin_fma(-D1, r, 1.0), r);
        sum[j] = __builtin_fma(r, q.sw, sum[j]);
      }
    } else if constexpr (NODES) {                               // not a smooth function of f across the window
      *failed |= 1u << k;
    } else {                                                    // cutoff differs between lanes / frequencies: masks
#pragma unroll
      for (int j = 0; j < NFC; ++j) {
        const double f = sfq[2 * j];
        const double d1 = f - q.c1;
        const double d2 = f + q.c1;
        const double D1 = __builtin_fma(d1, d1, q.wsq);
        const double D2 = __builtin_fma(d2, d2, q.wsq);
        const double m1 = (fabs(d1) < 750.0) ? 1.0 : 0.0;
        const double m2 = (fabs(d2) < 750.0) ? 1.0 : 0.0;
        const double num = __builtin_fma(m2, D1, m1 * D2);
        const double r = fdiv1(num, D1 * D2);
        sum[j] = __builtin_fma(r, q.sw, sum[j]);
        sum[j] = __builtin_fma(-(m1 + m2), q.sbase, sum[j]);
      }
    }
  }
  if constexpr (NODES) {                                        // raw line sums at the nodes; bsum travels separately
#pragma unroll
    for (int j = 0; j < NFC; ++j) awet[j] = sum[j];
    *bsum_out = bsum;
    return;
  }
#pragma unroll
  for (int j = 0; j < NFC; ++j) sum[j] -= bsum;
  // speed-dependent lines (ABH2O_SD): the resonant term inside |d1| < 10 w0 is the quadratic-speed-dependent
  // shape      Xc = (w0 - 1.5 w2 + i (d1 + 1.5 delta2)) / (w2 - i delta2),
  //            SD = 2 (1 - sqrt(pi) Xrt w(i Xrt)) / (w2 - i delta2),   Xrt = sqrt(Xc)
  // instead of the Lorentzian; outside it, and for the second term, the plain cutoff Lorentzians.
  const unsigned setC = (MWRT_ABLATE & 4) ? 0u : ((sd_eff | sd_extra) & all);
  for (unsigned m = setC; m; m &= m - 1u) {
    const int k = __builtin_ctz(m);
    const H2OLine q = h2o_line(M, k, pda, pvap, ti, tiln, ti2, shifted);
    const double w2 = M->h2o_w2[k] * pda * fexp(M->h2o_xw2[k] * tiln) + M->h2o_w2s[k] * pvap * fexp(M->h2o_xw2s[k] * tiln);
    const double delta2 = M->h2o_d2[k] * pda + M->h2o_d2s[k] * pvap;
    const cplx iden2 = crecip(cplx{w2, -delta2});              // 1 / (w2 - i delta2), once per (level, line)
    const double sdlim = (w2 > 0.0) ? 10.0 * q.w0 : -1.0;     // width2 == 0 at this level: plain Lorentz
    const double xre = q.w0 - 1.5 * w2, xim0 = 1.5 * delta2;
    LDS_RELOAD_FENCE();
    // pass 1: the cutoff Lorentzians, the resonant one masked out where the SD shape takes over
    const bool d1_in = (q.c1 - fmin < 750.0) && (fmax - q.c1 < 750.0) && (q.c1 - fmin > -750.0);
    const bool d2_in = fmax + q.c1 < 750.0 && fmin + q.c1 > -750.0;
    // half-sampled shape for this line and chunk?  (needs both Lorentz terms inside the cutoff: the common case)
    bool half = false;
    if constexpr (NFC == 16 && !NODES) half = sdw != nullptr && ((lm.h2o_sdint >> k) & 1u) && wave_all(d1_in && d2_in);
    if (wave_all(d1_in && d2_in)) {                                // both inside the cutoff everywhere (22 / 183 GHz lines)
      const double sbase2 = q.sbase + q.sbase;
#pragma unroll
      for (int j = 0; j < NFC; ++j) {
        const double f = sfq[2 * j];
        const double d1 = f - q.c1;
        const double d2 = f + q.c1;
        const double D1 = __builtin_fma(d1, d1, q.wsq);
        const double D2 = __builtin_fma(d2, d2, q.wsq);
        // (half-sampled: the odd slots keep their Lorentzian and get the interpolated difference in pass 2)
        const bool inner = fabs(d1) < sdlim && !(half && (j & 1) && j != 15);
        const double den12 = D1 * D2;
        double r = __builtin_amdgcn_rcp(den12);
        r = __builtin_fma(r, __builtin_fma(-den12, r, 1.0), r);
        const double num = inner ? D1 : D1 + D2;
        sum[j] = __builtin_fma(num * r, q.sw, sum[j]);
        sum[j] -= inner ? q.sbase : sbase2;
      }
    } else {
#pragma unroll
      for (int j = 0; j < NFC; ++j) {
        const double f = sfq[2 * j];
        const double d1 = f - q.c1;
        const double d2 = f + q.c1;
        const double D1 = __builtin_fma(d1, d1, q.wsq);
        const double D2 = __builtin_fma(d2, d2, q.wsq);
        const double a1 = fabs(d1);
        const double m1 = (a1 < 750.0 && !(a1 < sdlim)) ? 1.0 : 0.0;
        const double m2 = (fabs(d2) < 750.0) ? 1.0 : 0.0;
        const double num = __builtin_fma(m2, D1, m1 * D2);
        const double r = fdiv1(num, D1 * D2);
        sum[j] = __builtin_fma(r, q.sw, sum[j]);
        sum[j] = __builtin_fma(-(m1 + m2), q.sbase, sum[j]);
      }
    }
    // pass 2: the SD resonant shape, frequency by frequency, only where some lane of the wave is inside
    // 10 half-widths (the branch is wave-uniform, so nothing of one frequency interleaves with the next)
    LDS_RELOAD_FENCE();
    auto sd_shape = [&](double d1) -> double {                  // Re SD at detuning d1
      const cplx xc = cmul(cplx{xre, d1 + xim0}, iden2);
      const cplx xrt = csqrt_principal(xc);
      const cplx w = dcerror_upper(-xrt.im, xrt.re);
      const cplx pxw = cmul(cplx{1.77245385090551603 * xrt.re, 1.77245385090551603 * xrt.im}, w);
      return __builtin_fma(2.0 * (1.0 - pxw.re), iden2.re, 2.0 * pxw.im * iden2.im);      // Re((2 - 2 pxw) iden2)
    };
    bool done = false;
    if constexpr (NFC == 16 && !NODES) {
      if (half) {
        done = true;
        // the chunk lies on one side of the line: the closest frequency is one of its ends
        const double dmin = ::fmin(fabs(sfq[0] - q.c1), fabs(sfq[2 * 15] - q.c1));
        if (wave_any(dmin < sdlim)) {
          double dn[SD_NODES];
#pragma unroll
          for (int n = 0; n < SD_NODES; ++n) {
            const int j = sd_node_slot(n);
            const double d1 = sfq[2 * j] - q.c1;
            const double sdre = sd_shape(d1);
            const double lres = fdiv1(q.w0, __builtin_fma(d1, d1, q.wsq));     // the Lorentzian the shape replaces
            dn[n] = sdre - lres;
            const double r1 = (fabs(d1) < sdlim) ? sdre - q.base : 0.0;
            sum[j] = __builtin_fma(q.s, r1, sum[j]);
          }
#pragma unroll
          for (int i = 0; i < SD_TARGETS; ++i) {
            const int j = 2 * i + 1;
            double dl = 0.0;
#pragma unroll
            for (int n = 0; n < SD_NODES; ++n) dl = __builtin_fma(sdw[i * SD_NODES + n], dn[n], dl);
            const bool inner = fabs(sfq[2 * j] - q.c1) < sdlim;
            sum[j] = __builtin_fma(q.s, inner ? dl : 0.0, sum[j]);
          }
        }
      }
    }
    if (!done) {
#pragma unroll
      for (int j = 0; j < NFC; ++j) {
        const double d1 = sfq[2 * j] - q.c1;
        const bool inner = fabs(d1) < sdlim;
        if (wave_any(inner)) {
          const double r1 = inner ? sd_shape(d1) - q.base : 0.0;
          sum[j] = __builtin_fma(q.s, r1, sum[j]);
        }
      }
    }
  }
  const bool dry = !(L.rho > 0.0);
#pragma unroll
  for (int j = 0; j < NFC; ++j) {
    const double f2 = sfq[2 * j + 1];
    awet[j] = dry ? 0.0 : (3.183e-05 * den * sum[j] + con0) * f2;
  }
}

// ---------------------------------------------------------------------------------------------
// K1b: O2 lines + non-resonant + N2 continuum (O2AbsModel.o2_absorption / N2AbsModel [EXT])
//   S (f/F)^2 [ (w g + d1 Y)/D1 + (w g - d2 Y)/D2 ]  with one reciprocal per line and frequency
// ---------------------------------------------------------------------------------------------
struct O2Line {            // per-(level, line) quantities, frequency independent (all carry HALF the line's weight:
  double c1, df2;          //  the common factor 2 of P and Q is applied once, in the final scale)
  double P, Q;             //  n1/D1 + n2/D2 = 2 (f^2 P + Q) / (D1 D2),  P = a + c b,  Q = (c^2 + w^2)(a - c b)
  double cc;               //  c^2 + w^2
  double dnu;
};

template <int NFC, bool NODES = false>
__device__ __forceinline__ void dry_absorb(cmodel M, const LevelState& L, const double* sfq /*LDS: {f, f^2} per slot*/,
                                           const LineMasks& lm, double (&adry)[NFC], unsigned long long excl = 0ull,
                                           const double* init_sum = nullptr, unsigned long long* failed = nullptr) {
  const double temp = L.t;
  const double pres = L.p;
  const double th = fdiv(300.0, temp);
  const double th1 = th - 1.0;
  const double lnth = flog(th);
  const double b = fexp(M->o2_x * lnth);
  const double preswv = fdiv(L.rho * temp, M->o2_pvap_div);
  const double presda = pres - preswv;
  const double den = 0.001 * (presda * b + M->o2_wv_factor * preswv * th);
  const double dens = 0.001 * (presda + M->o2_wv_factor * preswv) * th;
  const double dfnr = M->o2_wb300 * den;
  const double pe2 = den * den;
  const bool second = M->o2_mix_mode != 0;
  const double ymul = second ? den : 0.001 * pres * b;
  const bool line1_dens = !second && M->o2_line1_dens;

  double sum[NFC];
#pragma unroll
  for (int j = 0; j < NFC; ++j) sum[j] = init_sum ? init_sum[j] : 0.0;

  // With d1 = f - c, d2 = f + c, D = d^2 + w^2, n1 = a + d1 b, n2 = a - d2 b the two terms of a line
  // share one reciprocal and the numerator collapses to a polynomial in f^2:
  //   n1/D1 + n2/D2 = (f^2 P + Q) / (D1 D2),  P = 2 (a + c b),  Q = 2 (c^2 + w^2)(a - c b)
  long long be_prev = -1;
  double ebe = 1.0;
  auto line_setup = [&](int k) -> O2Line {
    const auto& R = M->o2r[k];
    const double y = ymul * __builtin_fma(R.y1, th1, R.y0);
    double dnu = 0.0, gfac = 1.0;
    if (second) {
      dnu = pe2 * __builtin_fma(R.dnu1, th1, R.dnu0);
      gfac = __builtin_fma(pe2, __builtin_fma(R.g1, th1, R.g0), 1.0);
    }
    const double df = R.w300 * ((k == 0 && line1_dens) ? dens : den);
    // N- / N+ partners share BE: the exponential is redone only when the table value changes
    // (compared as bit patterns so the test stays on the scalar unit)
    const long long be_bits = __builtin_bit_cast(long long, R.be);
    if (be_bits != be_prev) { ebe = fexp(-R.be * th1); be_prev = be_bits; }
    const double str = R.s300rf2 * ebe;                               // S300 / F^2 (the f^2 is applied at the end)
    O2Line q;
    q.dnu = dnu;
    q.c1 = R.f + dnu;
    q.df2 = df * df;
    const double a = (str * df) * gfac;
    const double cb = q.c1 * (str * y);
    q.cc = __builtin_fma(q.c1, q.c1, q.df2);
    q.P = a + cb;
    q.Q = q.cc * (a - cb);
    return q;
  };

  const int nl = M->n_o2;
  const unsigned long long all = ((nl >= 64) ? ~0ull : ((1ull << nl) - 1ull)) & ~excl;
  // loop A: far lines -- polynomial denominator; a line whose shift |dnu| exceeds the allowance at any
  // level of this wave is handed to loop B
  unsigned long long near = ~lm.o2_far & all;
  const unsigned long long setA = (MWRT_ABLATE & 1) ? 0ull : (lm.o2_far & all);
  // ... four lines at a time (far_quad_accumulate); the count mod 4 left over joins loop B
  const unsigned long long setV = NODES ? 0ull : (setA & lm.o2_vfar);      // very far lines: one Taylor polynomial (vfar_add)
  const unsigned long long setQ = setA & ~setV;
  const unsigned long long leftA = lowest_bits(setQ, __builtin_popcountll(setQ) & 3);
  const unsigned long long leftP = (__builtin_popcountll(leftA) >= 2) ? lowest_bits(leftA, 2) : 0ull;   // ... two of them as a pair
  near |= leftA & ~leftP;
  auto far_setup = [&](int k, FarLine& fl) {
    const O2Line q = line_setup(k);
    double P = q.P, Q = q.Q;
    if (!NODES && second && !wave_all(fabs(q.dnu) < FAR_SHIFT_GHZ)) { near |= 1ull << k; P = 0.0; Q = 0.0; }
    fl.P = P; fl.Q = Q;
    fl.A2 = 2.0 * __builtin_fma(-q.c1, q.c1, q.df2);
    fl.Bc = q.cc * q.cc;
  };
  if (setV) {
    double acc[VF_TERMS];
#pragma unroll
    for (int j = 0; j < VF_TERMS; ++j) acc[j] = 0.0;
    for (unsigned long long m = setV; m; m &= m - 1ull) {
      FarLine q;
      far_setup(__builtin_ctzll(m), q);
      vfar_add(q, lm.vf_u0, lm.vf_h, acc);
    }
    LDS_RELOAD_FENCE();
    vfar_eval<NFC>(sfq, lm.vf_invh, -lm.vf_u0 * lm.vf_invh, acc, sum);
  }
  for (unsigned long long m = setQ & ~leftA; m;) {
    FarLine q0, q1, q2, q3;
    far_setup(__builtin_ctzll(m), q0); m &= m - 1ull;
    far_setup(__builtin_ctzll(m), q1); m &= m - 1ull;
    far_setup(__builtin_ctzll(m), q2); m &= m - 1ull;
    far_setup(__builtin_ctzll(m), q3); m &= m - 1ull;
    LDS_RELOAD_FENCE();
    far_quad_accumulate<NFC>(sfq, q0, q1, q2, q3, sum);
  }
  if (leftP) {
    FarLine q0, q1;
    unsigned long long m = leftP;
    far_setup(__builtin_ctzll(m), q0); m &= m - 1ull;
    far_setup(__builtin_ctzll(m), q1);
    LDS_RELOAD_FENCE();
    far_pair_accumulate<NFC>(sfq, q0, q1, sum);
  }
  // loop B: lines next to a chunk frequency -- D1, D2 formed from the detunings directly (no cancellation)
  if (MWRT_ABLATE & 1) near = 0ull;
  be_prev = -1;
  for (unsigned long long m = near; m; m &= m - 1ull) {
    const int k = __builtin_ctzll(m);
    const O2Line q = line_setup(k);
    LDS_RELOAD_FENCE();
#pragma unroll
    for (int j = 0; j < NFC; ++j) {
      const double f = sfq[2 * j], f2 = sfq[2 * j + 1];
      const double d1 = f - q.c1;
      const double d2 = f + q.c1;
      const double D1 = __builtin_fma(d1, d1, q.df2);
      const double D2 = __builtin_fma(d2, d2, q.df2);
      const double den12 = D1 * D2;
      double r = __builtin_amdgcn_rcp(den12);
      r = __builtin_fma(r, __builtin_fma(-den12, r, 1.0), r);
      sum[j] = __builtin_fma(__builtin_fma(f2, q.P, q.Q), r, sum[j]);
    }
  }
  if constexpr (NODES) {                                        // raw half-weight line sums at the nodes
#pragma unroll
    for (int j = 0; j < NFC; ++j) adry[j] = sum[j];
    (void)failed;
    return;
  }
  const double scale2 = 2.0 * M->o2_coef * presda * th * th * th;     // the 2 of P and Q
  // N2 collision-induced continuum (ABSN2): p^2 f^2 th^m
  const double pn2 = M->n2_ptot ? pres : L.pdry;
  const double n2c = M->n2_n * M->n2_l * pn2 * pn2 * fexp(M->n2_m * lnth);
  const double nr0 = 0.5 * M->o2_nonres * dfnr;
  const double dfnr2 = dfnr * dfnr;
#pragma unroll
  for (int j = 0; j < NFC; ++j) {
    const double f2 = sfq[2 * j + 1];
    const double hnonres = fdiv(nr0 * f2, th * (f2 + dfnr2));          // half the non-resonant term
    double o2 = scale2 * __builtin_fma(sum[j], f2, hnonres);
    o2 = fmax(o2, 0.0);
    adry[j] = o2 + n2c * sfq[2 * NFC + 2 + j] * f2;          // N2 frequency-dependence factor, per slot
  }
}

// RTEquation.exponential_integration [EXT]: log-mean ("exponential decay") layer value.
// Branch order is the contract.  Returns NaN-flag through `neg`.
//
// The log-mean itself: with s = (x1 - x0)/(x1 + x0),  ln(x1/x0) = 2 atanh(s), so
//   (x1 - x0)/ln(x1/x0) = (x1 + x0)/2 * s/atanh(s),   s/atanh(s) = 1 - s^2/3 - 4 s^4/45 - ...
// Adjacent levels of a sounding differ by a few percent, so |s| <= LOGMEAN_SMALL_S for a whole wave is
// the usual case (wave vote): one division and a 10-term series instead of a division, a full log
// (frexp, second division, series) and a third division.  It is also better conditioned than the
// quotient form, which loses up to 1e-7 relative when x1 - x0 is just above the 1e-9 switch.
constexpr double LOGMEAN_SMALL_S = 0.1715;      // |s| <= this: inside the interval s_over_atanh was fitted on

// s / atanh(s) = 1 - z/3 - 4 z^2/45 - 44 z^3/945 - ... (z = s^2) = 1 + z g(z), g of degree 6 fitted at the Chebyshev nodes
// of [0, 0.1716^2] (1.1e-18 relative; the Taylor series needs ten terms): the log-mean is (x1 + x0)/2 times this
__device__ __forceinline__ double s_over_atanh(double z) {
  double q = -0.014721548786632996;
  MWRT_FMA_SC(q, z, -0.01673734010479199);
  MWRT_FMA_SC(q, z, -0.02179782053718046);
  MWRT_FMA_SC(q, z, -0.03019399300258251);
  MWRT_FMA_SC(q, z, -0.04656084661263456);
  MWRT_FMA_SC(q, z, -0.08888888888879345);
  MWRT_FMA_SC(q, z, -0.33333333333333337);
  return __builtin_fma(q, z, 1.0);
}

// The same quotient on the next band, |s| <= LOGMEAN_MID_S (adjacent absorptions up to 4 : 1, the coarse top of a
// sounding): a (5,5) rational fit in z = s^2 on Chebyshev nodes of [0, 0.36], 7.7e-17 relative in exact arithmetic.
// Ten FMAs and one division, about half of log_mean_any.
constexpr double LOGMEAN_MID_S = 0.6;
__device__ __forceinline__ double s_over_atanh_mid(double z) {
  double pn = -1.24405457430790653678e-02, qd = -1.87173749238921052448e-03;
  MWRT_FMA_SC(pn, z, 2.32467124128396325363e-01);
  MWRT_FMA_SC(qd, z, 9.09554662683649162425e-02);
  MWRT_FMA_SC(pn, z, -1.25179985749945586930e+00);
  MWRT_FMA_SC(qd, z, -7.29233489586724676923e-01);
  MWRT_FMA_SC(pn, z, 2.79764691392900100515e+00);
  MWRT_FMA_SC(qd, z, 2.07624733687436672750e+00);
  MWRT_FMA_SC(pn, z, -2.76419873116542634427e+00);
  MWRT_FMA_SC(qd, z, -2.43086539783209769889e+00);
  pn = __builtin_fma(pn, z, 1.0);
  qd = __builtin_fma(qd, z, 1.0);
  return fdiv1(pn, qd);
}

// The log-mean for ANY ratio of two positive values with one division for the logarithm and one for the quotient:
//   x1/x0 = 2^e m,  m in [1/sqrt 2, sqrt 2]  (e from the exponent fields, x0' = x0 2^e),
//   s' = (x1 - x0')/(x1 + x0'),  ln(x1/x0) = e ln 2 + 2 s' (atanh(s')/s'),  result = (x1 - x0) / ln(x1/x0).
// ln keeps full RELATIVE accuracy as x1 -> x0 (e = 0, ln = 2 s' (1 + z/3 + ...)), which the quotient of a generic
// log cannot.  ~45 VALU, no branch: what a wave runs when some lane's levels are far apart (the coarse top of a
// sounding shares its wave with finely spaced levels).
__device__ __forceinline__ double log_mean_any(double x1, double x0, double d) {
  int e = __builtin_amdgcn_frexp_exp(x1) - __builtin_amdgcn_frexp_exp(x0);
  double x0s = __builtin_amdgcn_ldexp(x0, e);                      // x1 / x0s in (1/2, 2)
  const bool hi = x1 > 1.41421356237309504880 * x0s;
  const bool lo = x1 * 1.41421356237309504880 < x0s;
  x0s = hi ? x0s + x0s : (lo ? 0.5 * x0s : x0s);
  e = hi ? e + 1 : (lo ? e - 1 : e);
  const double sp = fdiv1(x1 - x0s, x1 + x0s);                     // |s'| <= 0.1716
  const double z = sp * sp;
  const double p = __builtin_fma(two_atanh_tail(z), z, 2.0);      // 2 atanh(s)/s = 2 + 2z/3 + 2z^2/5 + ...
  const double ed = (double)e;
  const double ln = __builtin_fma(ed, 6.93147180369123816490e-01, __builtin_fma(ed, 1.90821492927058770002e-10, sp * p));
  return fdiv1(d, ln);
}

template <bool ZEROFLG = true>
__device__ __forceinline__ double layer_value(double x1, double x0, bool& neg, bool live = true) {
  // live = this lane holds a layer (its result is used): the wave votes ignore the others
  const double d = x1 - x0;
  const double sm = x1 + x0;
  const bool same = fabs(d) < 1e-09;
  // x0 < 0 | x1 < 0 | x0 == 0 | x1 == 0 in one comparison (NaN inputs never reach this point)
  const bool nonpos = !(fmin(x1, x0) > 0.0);
  double r;
  const double s = fdiv1(d, sm);
  // the votes as algebra on comparison masks (each ballot is its v_cmp; combining bools first costs two VALU per vote)
  const wmask m_live = __builtin_amdgcn_ballot_w64(live);
  const wmask m_special = (__builtin_amdgcn_ballot_w64(nonpos) | __builtin_amdgcn_ballot_w64(same)) & m_live;
  const wmask m_plain = m_live & ~m_special;                     // lanes whose log-mean is the generic one
  if ((m_plain & ~__builtin_amdgcn_ballot_w64(fabs(s) <= LOGMEAN_SMALL_S)) == 0ull) {
    KEEP_BRANCH();
    r = (0.5 * sm) * s_over_atanh(s * s);
  } else if ((m_plain & ~__builtin_amdgcn_ballot_w64(fabs(s) <= LOGMEAN_MID_S)) == 0ull) {
    KEEP_BRANCH();
    r = (0.5 * sm) * s_over_atanh_mid(s * s);
  } else {
    KEEP_BRANCH();
    r = log_mean_any(x1, x0, d);
  }
  if (m_special != 0ull) {                                     // rare below the stratosphere: wave-uniform skip (and 12 VGPRs fewer live)
    const bool negative = (x0 < 0.0) | (x1 < 0.0);
    const bool zero = x0 == 0.0 || x1 == 0.0;
    if (negative && live) neg = true;
    r = zero ? (ZEROFLG ? sm * 0.5 : 0.0) : r;                 // zeroflg = True for wet & dry, False for liquid & ice
    r = same ? x1 : r;
    r = negative ? 0.0 : r;
  }
  return r;
}

// Four layer values behind ONE pair of wave votes (the TAU absorption kernels make 32 per lane and chunk).
// x1[k] in, layer value out (in place); x0[k] = the level below.
template <bool ZEROFLG = true>
__device__ __forceinline__ void layer_value4(double (&x1)[4], const double (&x0)[4], bool& neg, bool live) {
  double s[4];
  // votes as algebra on comparison masks (see layer_value)
  const wmask m_live = wballot(live);
  wmask m_special = 0ull, m_notsmall = 0ull, m_notmid = 0ull;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const double d = x1[k] - x0[k];
    s[k] = fdiv1(d, x1[k] + x0[k]);
    const wmask sp = wballot(!(fmin(x1[k], x0[k]) > 0.0)) | wballot(fabs(d) < 1e-09);
    m_special |= sp;
    m_notsmall |= ~(sp | wballot(fabs(s[k]) <= LOGMEAN_SMALL_S));
    m_notmid |= ~(sp | wballot(fabs(s[k]) <= LOGMEAN_MID_S));
  }
  m_special &= m_live;
  double r[4];
  if ((m_live & m_notsmall) == 0ull) {
    KEEP_BRANCH();
#pragma unroll
    for (int k = 0; k < 4; ++k) r[k] = (0.5 * (x1[k] + x0[k])) * s_over_atanh(s[k] * s[k]);
  } else if ((m_live & m_notmid) == 0ull) {
    KEEP_BRANCH();
#pragma unroll
    for (int k = 0; k < 4; ++k) r[k] = (0.5 * (x1[k] + x0[k])) * s_over_atanh_mid(s[k] * s[k]);
  } else {
    KEEP_BRANCH();
#pragma unroll
    for (int k = 0; k < 4; ++k) r[k] = log_mean_any(x1[k], x0[k], x1[k] - x0[k]);
  }
  if (m_special != 0ull) {
    KEEP_BRANCH();
#pragma unroll 1
    for (int k = 0; k < 4; ++k) {
      const double d = x1[k] - x0[k];
      const bool negative = (x0[k] < 0.0) | (x1[k] < 0.0);
      const bool zero = x0[k] == 0.0 || x1[k] == 0.0;
      if (negative && live) neg = true;
      double q = zero ? (ZEROFLG ? (x1[k] + x0[k]) * 0.5 : 0.0) : r[k];
      q = (fabs(d) < 1e-09) ? x1[k] : q;
      r[k] = negative ? 0.0 : q;
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) x1[k] = r[k];
}

// ---------------------------------------------------------------------------------------------
// Opt-in physics the reference leaves at pyrtlib's defaults: cloud liquid / ice
// absorption (cloudy=True + init_cloudy) and spherical refracted ray tracing (ray_tracing=True).
// Only the OPT instantiations of the fused kernel contain this code; the clear-sky kernels are untouched.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ cplx clog_(cplx w) {       // principal complex logarithm
  return {0.5 * flog(__builtin_fma(w.re, w.re, w.im * w.im)), atan2(w.im, w.re)};
}

// RTEquation.cloudy_absorption + LiqAbsModel.liquid_water_absorption [EXT]: Np/km at one level.
// The frequency-independent part of the liquid model is built once per level (CloudLevel), the per-frequency
// part is a handful of complex operations; ice is (8.18645 / wavelength[cm]) * deni * 0.000959553 dB/km.
struct CloudLevel {
  int mode;                      // liq_mode of the model
  double eps0;                   // static dielectric constant
  // mode 0 (Liebe, Hufford & Manabe 1991 / MPM93 double Debye):  a = eps1, b = 1/fp, c = 1/fs
  // mode 1 (Rosenkranz 2015: Patek 2009 static constant, Ellison 2007 Debye term, B-band term):
  //         a = delta, b = sd, c = delta_B, z1 and 1/cnorm complex
  double a, b, c;
  cplx z1, icnorm;
};
constexpr double CLOUD_KICE = 8.18645 * 0.000959553 * (0.1 * 2.302585092994045684) / 29.9792458;

__device__ __forceinline__ CloudLevel cloud_level(cmodel M, double tk) {
  CloudLevel c{};
  c.mode = M->liq_mode;
  if (c.mode == 0) {
    const double theta1 = 1.0 - fdiv(300.0, tk);
    c.eps0 = 77.66 - 103.3 * theta1;
    c.a = 0.0671 * c.eps0;
    const double fp = (316.0 * theta1 + 146.4) * theta1 + 20.2;
    c.b = fdiv(1.0, fp);
    c.c = fdiv(1.0, 39.8 * fp);
  } else {
    const double tc = tk - 273.15;
    const double lth = flog(fdiv(300.0, tk));
    c.eps0 = -43.7527 * fexp(0.05 * lth) + 299.504 * fexp(1.47 * lth) - 399.364 * fexp(2.11 * lth) + 221.327 * fexp(2.31 * lth);
    c.a = 80.69715 * fexp(-tc * (1.0 / 226.45));
    c.b = 1164.023 * fexp(fdiv(-651.4728, tc + 133.07));
    c.c = 4.008724 * fexp(-tc * (1.0 / 103.05));
    const double f1 = 10.46012 + tc * (0.1454962 + tc * (0.063267156 + tc * 0.00093786645));
    c.z1 = cplx{-0.75 * f1, f1};
    c.icnorm = crecip(clog_(cdiv(cplx{-4500.0, 2000.0}, c.z1)));      // 1/cnorm; 1/conj(cnorm) is its conjugate
  }
  return c;
}

// liquid absorption for water content `denl` [g m-3] at frequency f [GHz]
__device__ __forceinline__ double liquid_abs(const CloudLevel& c, double f, double denl) {
  cplx eps;
  if (c.mode == 0) {
    const double eps2 = 3.52;
    const cplx t1 = cdiv(cplx{c.eps0 - c.a, 0.0}, cplx{1.0, f * c.b});
    const cplx t2 = cdiv(cplx{c.a - eps2, 0.0}, cplx{1.0, f * c.c});
    eps = cplx{t1.re + t2.re + eps2, t1.im + t2.im};
  } else {
    const cplx z2 = {-4500.0, 2000.0};
    const double hdelta = 0.5 * c.c;
    const cplx kap0 = cdiv(cplx{0.0, -c.a * f}, cplx{c.b, f});                   // -delta z / (sd + z), z = i f
    const cplx lp = clog_(cdiv(cplx{-z2.re, f - z2.im}, cplx{-c.z1.re, f - c.z1.im}));
    const cplx lj = clog_(cdiv(cplx{-z2.re, f + z2.im}, cplx{-c.z1.re, f + c.z1.im}));
    const cplx chip = cmul(cplx{hdelta * lp.re, hdelta * lp.im}, c.icnorm);
    const cplx chij = cmul(cplx{hdelta * lj.re, hdelta * lj.im}, cplx{c.icnorm.re, -c.icnorm.im});
    eps = cplx{c.eps0 + (kap0.re + (chip.re + chij.re - c.c)), kap0.im + (chip.im + chij.im)};
  }
  const cplx re = cdiv(cplx{eps.re - 1.0, eps.im}, cplx{eps.re + 2.0, eps.im});
  return -0.06286 * re.im * f * denl;
}

// RTEquation.refractivity [EXT] (Thayer 1974): refractive index at one level
__device__ __forceinline__ double thayer_refindex(double p, double tk, double e) {
  const double pa = p - e, tc = tk - 273.16, tk2 = tk * tk, tc2 = tc * tc;
  const double rza = 1.0 + pa * (5.79e-07 * (1.0 + 0.52 / tk) - (0.00094611 * tc) / tk2);
  const double rzw = 1.0 + 1650.0 * (e / (tk * tk2)) * (1.0 - 0.01317 * tc + 0.000175 * tc2 + 1.44e-06 * (tc2 * tc));
  const double wetn = (64.79 * (e / tk) + 377600.0 * (e / tk2)) * rzw;
  const double dryn = 77.6036 * (pa / tk) * rza;
  return 1.0 + (dryn + wetn) * 1e-06;
}

// O3AbsModel.o3_absorption [EXT, recalled from Rosenkranz's o3abs -- unverified; the line list is data, include/mwrt.h
// mwrt_model_desc.n_x]: the extra trace species joins the DRY absorption of this level for the chunk's frequencies
// (RTEquation.clearsky_absorption(..., o3n) [EXT]).  Generic Van Vleck-Weisskopf lines with a Voigt half width; one
// reciprocal per line and frequency.  Opt-in path, not tuned.
template <int NFC>
__device__ __forceinline__ void x_absorb(cmodel M, double tk, double p, double numden, const double* sfq, double (&adry)[NFC]) {
  const double ti = fdiv(M->x_reft, tk);
  const double tiln = flog(ti);
  const double qvinv = (M->x_qvib_t > 0.0) ? 1.0 - fexp(-fdiv(M->x_qvib_t, tk)) : 1.0;
  const double sq = 4.3e-07 * fsqrt(fdiv(tk, M->x_mass));
  double sum[NFC];
#pragma unroll
  for (int j = 0; j < NFC; ++j) sum[j] = 0.0;
  const int nx = M->n_x;
  for (int k = 0; k < nx; ++k) {
    const double fl = M->x_fl[k];
    const double wc = M->x_w[k] * p * fexp(M->x_x[k] * tiln);
    const double bd = sq * fl;
    const double w = 0.5346 * wc + fsqrt(__builtin_fma(0.2166 * wc, wc, 0.6931 * (bd * bd)));
    const double wsq = w * w;
    const double sw = (M->x_s1[k] * fexp(M->x_b[k] * (1.0 - ti))) * fdiv(w, fl * fl);     // the f^2 of (f/FL)^2 is applied at the end
    LDS_RELOAD_FENCE();
#pragma unroll
    for (int j = 0; j < NFC; ++j) {
      const double f = sfq[2 * j];
      const double d1 = f - fl, d2 = f + fl;
      const double D1 = __builtin_fma(d1, d1, wsq), D2 = __builtin_fma(d2, d2, wsq);
      const double den12 = D1 * D2;
      double r = __builtin_amdgcn_rcp(den12);
      r = __builtin_fma(r, __builtin_fma(-den12, r, 1.0), r);
      sum[j] = __builtin_fma((D1 + D2) * r, sw, sum[j]);
    }
  }
  const double pref = ((M->x_coef * numden) * qvinv) * fexp(2.5 * tiln);
#pragma unroll
  for (int j = 0; j < NFC; ++j) adry[j] = __builtin_fma(pref * sfq[2 * j + 1], sum[j], adry[j]);
}

// RTEquation.ray_tracing [EXT] (TBMODEL RAYTRAC: Dutton, Thayer & Westwater after Bean & Dutton fig. 3.20).
// Stores the PATH FACTOR ds_i / dz_i per layer, amf [nprof][nang][nlev] (entry 0 = 0): the slant-path
// integration multiplies the zenith layer optical depth by it, exactly where the plane-parallel path
// multiplies by 1/sin(elev).
//
// Workgroup = profile, LANE = LEVEL, loop over angles.  The reference walks the levels serially, carrying
// (phi, tau, r, tan theta) of the level below; but theta_i depends only on level i and the ground, and the
// carried sums enter ds only through differences,
//     phi_i - phi_{i-1} = (dtheta_i - dtheta_{i-1}) + dtau_i,      |tau_i - tau_{i-1}| = |dtau_i|,
// so every layer needs just its lower neighbour (one __shfl_up, wave seams through LDS): no scan, and the
// differences are formed directly instead of from two running sums (slightly better conditioned than the
// reference's own order; agreement ~1e-12 relative in ds).  A trapped ray (ducting: argth <= 0 at any level)
// gives NaN factors for that angle and duct[profile] = 1.  libm calls (asin, tan, ...): ~3 % of the opt-in
// path's arithmetic, not tuned further.
#ifdef MWRT_HOST_TU      // non-template kernels live in ONE translation unit (csrc/mwrt.hip)
constexpr double EARTH_RADIUS_KM = 6370.949;
__global__ void __launch_bounds__(1024)
k_ray_paths(const double* __restrict__ z, const double* __restrict__ p, const double* __restrict__ t,
            const double* __restrict__ rh, int nlev, const double* __restrict__ elev_deg, int nang,
            double* __restrict__ amf, uint8_t* __restrict__ duct) {
  __shared__ double seam[16][3];            // last lane of each wave: refractive index, tan(theta), dtheta
  const int64_t prof = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & (WAVE - 1), wave = tid / WAVE;
  const bool active = tid < nlev;
  const int64_t off = prof * nlev + (active ? tid : 0);
  const double qnan = __builtin_nan("");
  const double z0 = z[prof * nlev];
  const double zi = z[off] - z0;
  const double pi = p[off], ti = t[off], rhi = rh[off];
  const bool bad_prof = __syncthreads_or(active && (isnan(zi) || isnan(pi) || isnan(ti) || isnan(rhi)));
  const double ni = thayer_refindex(pi, ti, goff_gratch_e(ti, rhi));
  // neighbour level i-1 (level-only quantities)
  if (lane == WAVE - 1) seam[wave][0] = ni;
  __syncthreads();
  double nprev = __shfl_up(ni, 1, WAVE);
  double zprev = __shfl_up(zi, 1, WAVE);
  if (lane == 0 && wave > 0) { nprev = seam[wave - 1][0]; zprev = z[off - 1] - z0; }
  const double n0 = thayer_refindex(p[prof * nlev], t[prof * nlev], goff_gratch_e(t[prof * nlev], rh[prof * nlev]));
  const double rs = EARTH_RADIUS_KM + 0.0 + z0;
  const double r = EARTH_RADIUS_KM + zi + z0;
  const double rl = EARTH_RADIUS_KM + zprev + z0;
  const double dz = zi - zprev;
  double refbar;
  if (ni == nprev || ni == 1.0 || nprev == 1.0) refbar = (ni + nprev) * 0.5;
  else refbar = 1.0 + (nprev - ni) / (log((nprev - 1.0) / (ni - 1.0)));
  for (int a = 0; a < nang; ++a) {
    double* out = amf + (prof * nang + a) * nlev;
    const double angle = elev_deg[a];
    if (bad_prof || isnan(angle)) {            // NaN inputs are the main kernel's business
      if (active) out[tid] = tid == 0 ? 0.0 : qnan;
      continue;
    }
    if ((angle >= 89.0 && angle <= 91.0) || (angle >= -91.0 && angle <= -89.0)) {
      if (active) out[tid] = (tid > 0 && dz != 0.0) ? 1.0 : 0.0;
      continue;
    }
    const double theta0 = angle * (M_PI / 180.0);
    const double costh0 = cos(theta0), sina = sin(theta0 * 0.5);
    const double a0 = 2.0 * (sina * sina);
    // my level: theta_i, dtheta_i (level 0 is the ground: theta0, 0)
    const double argdth = zi / rs - ((n0 - ni) * costh0 / ni);
    const double argth = 0.5 * (a0 + argdth) / r;
    const bool trapped_here = active && tid > 0 && !(argth > 0.0);
    double theta = theta0, dtheta = 0.0;
    if (tid > 0 && argth > 0.0) {
      const double sint = sqrt(r * argth);
      theta = 2.0 * asin(sint);
      if ((theta - 2.0 * theta0) <= 0.0) {
        const double dendth = 2.0 * (sint + sina) * cos((theta + theta0) * 0.25);
        const double sind4 = (0.5 * argdth - zi * argth) / dendth;
        dtheta = 4.0 * asin(sind4);
        theta = theta0 + dtheta;
      } else {
        dtheta = theta - theta0;
      }
    }
    const double tanth = tan(theta);
    __syncthreads();                             // previous angle's seam reads are done
    if (lane == WAVE - 1) { seam[wave][1] = tanth; seam[wave][2] = dtheta; }
    const bool trapped = __syncthreads_or(trapped_here);
    double tanthl = __shfl_up(tanth, 1, WAVE);
    double dthl = __shfl_up(dtheta, 1, WAVE);
    if (lane == 0 && wave > 0) { tanthl = seam[wave - 1][1]; dthl = seam[wave - 1][2]; }
    double f = 0.0;
    if (tid > 0) {
      const double cthbar = ((1.0 / tanth) + (1.0 / tanthl)) * 0.5;
      const double dtau = cthbar * (nprev - ni) / refbar;
      const double dphi = (dtheta - dthl) + dtau;
      const double sh = sin(dphi * 0.5);
      double dsi = sqrt(dz * dz + 4.0 * r * rl * (sh * sh));
      if (dtau != 0.0) {
        const double dtaua = fabs(dtau);
        dsi = dsi * (dtaua / (2.0 * sin(dtaua * 0.5)));
      }
      f = (dz != 0.0) ? dsi / dz : 0.0;
    }
    if (active) out[tid] = trapped ? (tid == 0 ? 0.0 : qnan) : f;     // pyrtlib gives up on the whole ray
    if (trapped && tid == 0) duct[prof] = 1;
  }
}

#endif  // MWRT_HOST_TU

// ---------------------------------------------------------------------------------------------
// fused kernel: profile in -> TB out
// ---------------------------------------------------------------------------------------------
constexpr int MAX_MULTI = 8;   // absorption models evaluated by one launch (the wrapper runs four)

struct FusedArgs {
  // blockIdx.x enumerates (model, profile): outputs are [nmodels][nprof]..., inputs [nprof]...
  const ModelFlat* Ms[MAX_MULTI];
  int64_t nprof_in;        // profiles per model
  const double* z; const double* p; const double* t; const double* rh;   // [nprof][nlev]
  const double* frq;       // [nf] device
  const double* airmass;   // [nang] device: 1/sin(elev)
  double* tb;              // [nprof][nang][nf]
  uint8_t* valid;          // [nprof]
  double* tbatm; double* tmr; double* tauwet; double* taudry;   // optional [nprof][nang][nf]
  double* taulay;          // optional [nprof][nf][nlev]
  int nlev, nf, nang;
  int write_valid;         // 1: this launch has one workgroup per profile and sets valid = 1 itself
  LaunchGeom g;
  // by-products and opt-in physics (read by the OPT / EXTRAS instantiations only; null / 0 otherwise)
  const double* denliq; const double* denice;   // [nprof][nlev] g m-3, either may be null
  const double* amf;       // [nprof][nang][nlev] ray-traced path factor ds/dz, or null (plane-parallel)
  const uint8_t* duct;     // [nprof] 1: a ray of this profile was trapped (valid = 3)
  double* tauliq; double* tauice;               // optional [nprof][nang][nf]
  // ALPHA instantiation (RTE from materialised absorption): awet, adry [nprof][nf][nlev] as k_absorb writes them
  const double* awet_in; const double* adry_in;
  const LineMasks* masks[MAX_MULTI];   // per model: LineMasks of every frequency chunk (host-computed)
  const double* o3n;       // OPT: ozone number density [nprof][nlev] molecules m-3, or null
#if MWRT_PHASE_CLOCK
  long long* phase;        // diagnostic build only: [nprof][4][10] wall-clock stamps + HW_ID, XCC_ID
#endif
};

// NaN / negative-absorption exit: every output of this (profile, chunk) becomes NaN
__device__ __forceinline__ void blank_outputs(const FusedArgs& A, int64_t prof, int jbase, int nfc, int tid, int nthreads) {
  const double qnan = __builtin_nan("");
  const int nang = A.nang, nlev = A.nlev;
  for (int it = tid; it < nfc * nang; it += nthreads) {
    const int j = it / nang, a = it % nang;
    const int64_t o = (prof * nang + a) * A.nf + jbase + j;
    A.tb[o] = qnan;
    if (A.tbatm) A.tbatm[o] = qnan;
    if (A.tmr) A.tmr[o] = qnan;
    if (A.tauwet) A.tauwet[o] = qnan;
    if (A.taudry) A.taudry[o] = qnan;
    if (A.tauliq) A.tauliq[o] = qnan;
    if (A.tauice) A.tauice[o] = qnan;
  }
  if (A.taulay) for (int it = tid; it < nfc * nlev; it += nthreads)
    A.taulay[(prof * A.nf + jbase + it / nlev) * nlev + it % nlev] = qnan;
}

// NFC = frequencies per workgroup (accumulators in registers during K1);
// NFK = frequencies per K2 pass (rows of tau / B kept in LDS at a time): NFC = NPASS * NFK.
// Keeping only NFK rows resident holds the workgroup under 40 KB of LDS, so FOUR 192-thread
// workgroups (12 waves = 3 per SIMD) fit a CU and a 1000-profile batch is one resident round.
// (the TB-only variants are pinned to 3 waves per SIMD -- the clear-sky one sits at 161 of 168 VGPRs on its own and
// twelve more cost a third of the throughput; the cloud / ray-tracing one lands one register above the step and
// spills two; the RTE-from-absorption variant is pinned to the 4 waves its 256-thread launch relies on)
#ifndef MWRT_MIN_WAVES
#define MWRT_MIN_WAVES 1
#endif
// ALPHA = the K2 half alone: absorption coefficients are READ from HBM (what k_absorb wrote) instead of
// evaluated -- the two-kernel K1 -> alpha -> K2 form of the fine-grid configuration, and the entry for callers
// who bring their own absorption.
template <int NFC, int NFK, int MAXT, bool OPT = false, bool EXTRAS = false, bool ALPHA = false>
__global__ void __launch_bounds__(MAXT, (MAXT <= 256 ? (EXTRAS ? MWRT_MIN_WAVES : (ALPHA ? 4 : 3)) : 1))
k_tb_fused(const FusedArgs A) {
  constexpr int NPASS = (NFC + NFK - 1) / NFK;             // the last pass may hold fewer rows (14 = 8 + 6)
  static_assert(NPASS <= 2, "LaunchGeom carries the split of two passes");
  extern __shared__ __attribute__((aligned(16))) double lds[];     // 16-B base: wide ds_read stays aligned (guide G17)
  const int tid = threadIdx.x;
  const int lane = tid & (WAVE - 1);
  const int wave = tid / WAVE;
  const int nthreads = blockDim.x;
  const int nwaves = nthreads / WAVE;
  const int64_t prof = blockIdx.x;                       // output row: model * nprof_in + profile
  const int mi = (int)(prof / A.nprof_in);
  const int64_t pin = prof - mi * A.nprof_in;            // input profile
  const int jbase = blockIdx.y * NFC;
  const int nfc = min(NFC, A.nf - jbase);
  const int nlev = A.nlev, nang = A.nang, ld = A.g.ldrow;
  const cmodel M = (cmodel)A.Ms[mi];
  const cdoubles cfrq = (cdoubles)A.frq;
  const cdoubles cam = (cdoubles)A.airmass;

  double* tau = lds;                                     // [NFK][ld] zenith layer optical depth (wet+dry)
  double* bof = tau + (size_t)NFK * ld;                  // [NFK][ld] Planck function B(T_i, f_j)
  double* part = bof + (size_t)NFK * ld;                 // [pairs*nseg][2] segment partials (B, T)
  double* scratch = part + (size_t)A.g.npart;            // [16] block_sum scratch
  double* edge = scratch + 16;                           // [nwaves][2*NFC] last lane of each wave
  constexpr int GRP = 16;                                // levels per group of the layer-tau maxima
  const int ngrp = nthreads / GRP;
  float* gmax = (float*)(edge + (size_t)nwaves * 2 * NFC);   // [NFK][ngrp] largest zenith layer tau of 16 levels of a row
  int* wcnt = (int*)(gmax + (size_t)NFK * ngrp);          // [nwaves] thin work items per wave
  int* perm = wcnt + nwaves;                              // [nthreads] work items, thin ones first
  __shared__ int s_flag;
  __shared__ double sfq[5 * NFC + 2];                     // {f, f^2} per slot, {fmin, fmax}, N2 fdep, 1/f, cosmic-background Planck term per slot

  MWRT_STAMP(0);
#if MWRT_PHASE_CLOCK
  if (A.phase && lane == 0) {
    A.phase[((int64_t)blockIdx.x * 4 + wave) * 10 + 8] = (long long)__builtin_amdgcn_s_getreg((31 << 11) | 4);     // HW_REG_HW_ID
    A.phase[((int64_t)blockIdx.x * 4 + wave) * 10 + 9] = (long long)__builtin_amdgcn_s_getreg((31 << 11) | 20);    // HW_REG_XCC_ID
  }
#endif
  // uniform frequency chunk; slots beyond nfc reuse the last valid one (results discarded)
  if (tid == 0) s_flag = 0;
  if (tid < NFC) {
    const double f = cfrq[jbase + min(tid, nfc - 1)];
    sfq[2 * tid] = f; sfq[2 * tid + 1] = f * f;
    double fdep = 1.0;
    if (M->n2_fdep) { const double q = f * (1.0 / 450.0); fdep = 0.5 + fdiv(0.5, 1.0 + q * q); }
    sfq[2 * NFC + 2 + tid] = fdep;
    sfq[3 * NFC + 2 + tid] = fdiv(1.0, f);
    // B(T_cosmic, f): one value per frequency, not per (frequency, angle) pair
    sfq[4 * NFC + 2 + tid] = fdiv(1.0, fexp(fdiv(f * (1e9 * M->planck_h / M->boltzmann_k), M->t_cosmic)) - 1.0);
  }
  if (tid == WAVE - 1) {
    double lo = cfrq[jbase], hi = lo;
    for (int j = 1; j < nfc; ++j) { const double f = cfrq[jbase + j]; lo = fmin(lo, f); hi = fmax(hi, f); }
    sfq[2 * NFC] = lo; sfq[2 * NFC + 1] = hi;
  }
  __syncthreads();

  const bool active = tid < nlev;
  // a wave beyond the top level (the RTE-from-absorption launch adds one for the K2 work items) holds no level:
  // it skips the per-level phases wave-uniformly and only meets the barriers
  const bool wave_live = wave * WAVE < nlev;
  const int64_t off = pin * nlev + (active ? tid : 0);
  const double zi = A.z[off], ti = A.t[off];
  const double pi = ALPHA ? 0.0 : A.p[off], rhi = ALPHA ? 0.0 : A.rh[off];
  if (active && (isnan(zi) || isnan(pi) || isnan(ti) || isnan(rhi))) atomicOr(&s_flag, 1);
  double awet[NFC], adry[NFC];
  if constexpr (ALPHA) {
    bool bad = false;
    if (wave_live) {
#pragma unroll
      for (int j = 0; j < NFC; ++j) {
        const int64_t o = (pin * A.nf + jbase + min(j, nfc - 1)) * nlev + (active ? tid : 0);     // 512-B rows per wave
        awet[j] = A.awet_in[o];
        adry[j] = A.adry_in[o];
        bad = bad || isnan(awet[j]) || isnan(adry[j]);
      }
    } else {
#pragma unroll
      for (int j = 0; j < NFC; ++j) { awet[j] = 0.0; adry[j] = 0.0; }
    }
    if (active && bad) atomicOr(&s_flag, 1);
  }
  double denl = 0.0, deni = 0.0, o3n = 0.0;
  if constexpr (OPT) {
    if (A.denliq) denl = A.denliq[off];
    if (A.denice) deni = A.denice[off];
    if (A.o3n) o3n = A.o3n[off];
    if (active && (isnan(denl) || isnan(deni) || isnan(o3n))) atomicOr(&s_flag, 1);
  }
  __syncthreads();
  if (s_flag) {                               // check_for_nans: outputs stay NaN, valid = 0
    blank_outputs(A, prof, jbase, nfc, tid, nthreads);
    if (tid == 0) A.valid[prof] = 0;
    return;
  }

  // ---- phase K1: absorption at my level for the NFC frequencies ----
  if constexpr (!ALPHA) {
    const double e = goff_gratch_e(ti, rhi);
    const LevelState L = level_state(pi, ti, e);
    const LineMasks lm = load_masks(A.masks[mi], blockIdx.y);
    MWRT_STAMP(1);
    MWRT_SETPRIO(3);
    h2o_absorb<NFC>(M, L, sfq, lm, awet);
    MWRT_STAMP(2);
    MWRT_SETPRIO(2);
    dry_absorb<NFC>(M, L, sfq, lm, adry);
    MWRT_SETPRIO(1);
    MWRT_STAMP(3);
    if constexpr (OPT) {
      if (A.o3n) x_absorb<NFC>(M, ti, pi, o3n, sfq, adry);       // clearsky_absorption(..., o3n): ozone joins the dry term
    }
  }
  // neighbour level i-1: lane-1 through the crossbar, wave seams through a 2*NFC-double edge row
  if (lane == WAVE - 1) {
#pragma unroll
    for (int j = 0; j < NFC; ++j) { edge[wave * 2 * NFC + j] = awet[j]; edge[wave * 2 * NFC + NFC + j] = adry[j]; }
  }
  __syncthreads();
  // Wet and dry rows are kept apart only where the opacity columns are wanted (EXTRAS); otherwise td[] holds
  // the layer total (wet + dry, then + ice + liquid) and tw[] is never materialised: 28 registers fewer.
  double tw[EXTRAS ? NFC : 1], td[NFC];
  bool neg = false;
  {
    const double z0 = A.z[pin * nlev];        // execute() works in height above the antenna
    const double dz = (active && tid > 0) ? ((zi - z0) - (A.z[off - 1] - z0)) : 0.0;
    const bool seam = (lane == 0) && (wave > 0);
    const bool has_prev = active && tid > 0;
    if (!wave_live) {
#pragma unroll
      for (int j = 0; j < NFC; ++j) { td[j] = 0.0; if constexpr (EXTRAS) tw[j] = 0.0; }
    } else
#pragma unroll
    for (int j = 0; j < NFC; ++j) {
#pragma clang fp contract(off)               // wet * dz + dry * dz rounds the same way in every instantiation
      double pw = __shfl_up(awet[j], 1, WAVE);
      double pd = __shfl_up(adry[j], 1, WAVE);
      if (seam) { pw = edge[(wave - 1) * 2 * NFC + j]; pd = edge[(wave - 1) * 2 * NFC + NFC + j]; }
      const double lw = layer_value(awet[j], pw, neg, has_prev), ld_ = layer_value(adry[j], pd, neg, has_prev);
      const double twj = has_prev ? lw * dz : 0.0;
      const double tdj = has_prev ? ld_ * dz : 0.0;
      if constexpr (EXTRAS) { tw[j] = twj; td[j] = tdj; }
      else td[j] = twj + tdj;
    }
  }
  // cloud liquid / ice (opt-in): same layer rule with zeroflg = False; tau = ((wet + dry) + ice) + liquid
  // (kept as separate arrays only where the opacity columns are wanted; otherwise folded into the dry row)
  constexpr bool CLOUD_ROWS = OPT && EXTRAS;
  double tl[CLOUD_ROWS ? NFC : 1], tci[CLOUD_ROWS ? NFC : 1];
  if constexpr (OPT) {
#pragma unroll
    for (int j = 0; j < (CLOUD_ROWS ? NFC : 1); ++j) { tl[j] = 0.0; tci[j] = 0.0; }
    // Skipped when the profile holds no cloud at all (workgroup vote).  The liquid absorption of every level goes
    // through the (still unused) tau / Planck rows of LDS, [frequency][level], so each lane reads its own and its
    // lower neighbour's value: no crossbar, no wave seams, and the per-level model state dies before the layer loop.
    const bool cloud_here = active && (denl > 0.0 || deni > 0.0);
    if ((A.denliq || A.denice) && __syncthreads_or(cloud_here)) {
      auto row = [&](int j) -> double* { return (j < NFK ? tau + (size_t)j * ld : bof + (size_t)(j - NFK) * ld); };
      {
        const CloudLevel cl = cloud_level(M, ti);
        if (active) {
#pragma unroll
          for (int j = 0; j < NFC; ++j) row(j)[tid] = (denl > 0.0) ? liquid_abs(cl, sfq[2 * j], denl) : 0.0;
        }
      }
      __syncthreads();
      const bool has_below = active && tid > 0;
      const double deni_prev = (has_below && A.denice) ? A.denice[off - 1] : 0.0;
      const double z0 = A.z[pin * nlev];
      const double dz = has_below ? ((zi - z0) - (A.z[off - 1] - z0)) : 0.0;
#pragma unroll
      for (int j = 0; j < NFC; ++j) {
        const double f = sfq[2 * j];
        const double al = active ? row(j)[tid] : 0.0;
        const double pl = has_below ? row(j)[tid - 1] : 0.0;
        const double ai = (deni > 0.0) ? CLOUD_KICE * f * deni : 0.0;
        const double pc = (deni_prev > 0.0) ? CLOUD_KICE * f * deni_prev : 0.0;
        const double ll = layer_value<false>(al, pl, neg, has_below), li = layer_value<false>(ai, pc, neg, has_below);
        const double tlj = has_below ? ll * dz : 0.0;
        const double tij = has_below ? li * dz : 0.0;
        if constexpr (CLOUD_ROWS) { tl[j] = tlj; tci[j] = tij; }
        else td[j] = (td[j] + tij) + tlj;
      }
      __syncthreads();                          // the rows are about to be refilled with tau / B
    }
  }
  MWRT_STAMP(4);
  if (neg) atomicOr(&s_flag, 2);
  __syncthreads();
  if (s_flag) {                               // pyrtlib raises ValueError here: flag 2, NaN out
    blank_outputs(A, prof, jbase, nfc, tid, nthreads);
    if (tid == 0) A.valid[prof] = 2;
    return;
  }
  const double hk = 1e9 * M->planck_h / M->boltzmann_k;
  const double inv_hk = 1e-9 * M->boltzmann_k / M->planck_h;
  // The TB-only instantiation (EXTRAS = false) carries none of the by-product code: the compiler would
  // otherwise evaluate tbatm / tmr speculatively and keep the opacity sums' registers alive.
  bool want_tau = false;
  if constexpr (EXTRAS) {
    if (A.taulay && active) {
#pragma unroll
      for (int j = 0; j < NFC; ++j)
        if (j < nfc) {
          double tz = tw[j] + td[j];
          if constexpr (CLOUD_ROWS) tz = (tz + tci[j]) + tl[j];
          A.taulay[(prof * A.nf + jbase + j) * nlev + tid] = tz;
        }
    }
    want_tau = (A.tauwet != nullptr) || (A.taudry != nullptr) || (A.tauliq != nullptr) || (A.tauice != nullptr);
  }

  // ---- phase K2: slant-path RTE (RTEquation.planck, from_sat = False [EXT]), NFK rows at a time ----
#pragma unroll
  for (int h = 0; h < NPASS; ++h) {
    const int nfk = min(NFK, nfc - h * NFK);            // frequencies live in this pass (uniform)
    if (nfk <= 0) break;
    if (h > 0) MWRT_SETPRIO(0);
    if (h > 0) __syncthreads();                          // previous pass has finished reading LDS
    const int nseg = A.g.nseg[h], seglen = A.g.seglen[h];
    const int npairs = nfk * nang;
    const int items = (MWRT_ABLATE & 8) ? 0 : npairs * nseg;
    // wave-uniform: this pass deals its work items to the lanes thin ones first (see below)
    const bool sorted = !(OPT && A.amf) && items <= nthreads && seglen >= K2_SORT_MIN_SEGLEN;
    if (wave_live) {
      const double hkt = fdiv(hk, ti);                  // h / (k T) per GHz
      const double kth = ti * inv_hk;                   // its inverse, without a second division per lane
#pragma unroll
      for (int jj = 0; jj < NFK; ++jj) {
        const int j = h * NFK + jj;
        if (j < NFC) {
          double tz = td[j];
          if constexpr (EXTRAS) tz = tw[j] + td[j];
          if constexpr (CLOUD_ROWS) tz = (tz + tci[j]) + tl[j];
          const double bz = planck_b(sfq[2 * j] * hkt, sfq[3 * NFC + 2 + j] * kth);
          if (active) { tau[jj * ld + tid] = tz; bof[jj * ld + tid] = bz; }
          // largest layer value of each group of 16 levels (rounded up): decides once per work item whether
          // every layer of its segment is thin at its airmass
          if (sorted) {
            const float m = row16_max(active ? (float)tz * 1.0000005f : 0.0f);
            if ((lane & (GRP - 1)) == 0) gmax[jj * ngrp + tid / GRP] = m;
          }
        }
      }
    }
    // optional zenith opacity sums (tauwet / taudry columns); deterministic order
    double swet[EXTRAS ? NFK : 1], sdry[EXTRAS ? NFK : 1], sliq[EXTRAS ? NFK : 1], sice[EXTRAS ? NFK : 1];
    const bool rays = OPT && A.amf != nullptr;
    if constexpr (EXTRAS) {
      if (want_tau && !rays) {
#pragma unroll
        for (int jj = 0; jj < NFK; ++jj) {
          constexpr int JL = NFC - 1;
          const int j = min(h * NFK + jj, JL);               // rows past the chunk repeat the last one (never read)
          swet[jj] = block_sum(tw[j], scratch, tid, nthreads);
          sdry[jj] = block_sum(td[j], scratch, tid, nthreads);
          if constexpr (OPT) {
            sliq[jj] = block_sum(tl[j], scratch, tid, nthreads);
            sice[jj] = block_sum(tci[j], scratch, tid, nthreads);
          } else {
            sliq[jj] = 0.0; sice[jj] = 0.0;
          }
        }
      }
    }
    if constexpr (EXTRAS && OPT) {
      if (want_tau && rays) {
        // ray-traced paths: the opacity columns are sums of layer value x path factor, one species at a
        // time through the tau rows (the DataFrame path of a single execute(); not a throughput path)
        const int npairs_r = nfk * nang;
        for (int sp = 0; sp < 4; ++sp) {
          double* outp = sp == 0 ? A.tauwet : sp == 1 ? A.taudry : sp == 2 ? A.tauliq : A.tauice;
          __syncthreads();
          if (active) {
#pragma unroll
            for (int jj = 0; jj < NFK; ++jj) {
              const int j = h * NFK + jj;
              if (j < NFC) tau[jj * ld + tid] = sp == 0 ? tw[j] : sp == 1 ? td[j] : sp == 2 ? tl[j] : tci[j];
            }
          }
          __syncthreads();
          if (outp) for (int pr = tid; pr < npairs_r; pr += nthreads) {
            const int jj = pr / nang, a = pr - jj * nang;
            const double* fr = A.amf + (pin * nang + a) * nlev;
            double acc = 0.0;
            for (int i = 1; i < nlev; ++i) acc += tau[jj * ld + i] * fr[i];
            outp[(prof * nang + a) * A.nf + jbase + h * NFK + jj] = acc;
          }
        }
        __syncthreads();
        if (active) {
#pragma unroll
          for (int jj = 0; jj < NFK; ++jj) {
            const int j = h * NFK + jj;
            if (j < NFC) tau[jj * ld + tid] = ((tw[j] + td[j]) + tci[j]) + tl[j];
          }
        }
      }
    }
    __syncthreads();

    // Work item = (pair, level segment).  A layer is thin when tau * airmass <= 1/8: an item whose whole
    // segment is thin takes the division-free step.  The choice is per WAVE, so the items are dealt to the lanes
    // thin ones first (ballot ranks + one pass through LDS): at most one wave mixes both kinds and runs the
    // general step for all its lanes.  The maxima that decide it are per 16 levels of a row (gmax).
    int nthin_all = 0;
    if (sorted) {
      const bool mine = tid < items;
      const int it = mine ? tid : 0;
      const int pr = div_small(it, nseg, A.g.magic_nseg[h]), seg = it - pr * nseg;
      const int jj = div_small(pr, nang, A.g.magic_nang), a = pr - jj * nang;
      const int lo = 1 + seg * seglen, hi = min(lo + seglen, nlev);
      float gm = 0.0f;
      for (int g = lo / GRP; g <= (hi - 1) / GRP; ++g) gm = fmaxf(gm, gmax[jj * ngrp + g]);
      // a NaN airmass (its rows come out NaN either way) counts as thin
      const bool thin = mine && !((double)gm * cam[a] > EXP_SMALL_X);
      const unsigned long long bal = __ballot(thin);
      const int rank = __popcll(bal & ((1ull << lane) - 1ull));
      if (lane == 0) wcnt[wave] = __popcll(bal);
      __syncthreads();
      int before = 0, nthin = 0;
      for (int w = 0; w < nwaves; ++w) { const int c = wcnt[w]; nthin += c; if (w < wave) before += c; }
      if (mine) perm[thin ? before + rank : nthin + tid - (before + rank)] = (a << 11) | (jj << 7) | seg;
      nthin_all = nthin;
      __syncthreads();
    }
    for (int it0 = tid; it0 < items; it0 += nthreads) {
      int seg, jj, a;
      if (sorted) {
        const int key = perm[it0];
        seg = key & 127; jj = (key >> 7) & 15; a = key >> 11;
      } else {
        const int pr = div_small(it0, nseg, A.g.magic_nseg[h]);
        seg = it0 - pr * nseg; jj = div_small(pr, nang, A.g.magic_nang); a = pr - jj * nang;
      }
      const int it = (jj * nang + a) * nseg + seg;
      const double am = cam[a];
      const int lo = 1 + seg * seglen;
      const int hi = min(lo + seglen, nlev);
      const double* tj = tau + jj * ld;
      const double* bj = bof + jj * ld;
      const double* fr = nullptr;
      if constexpr (OPT) fr = A.amf ? A.amf + (pin * nang + a) * nlev : nullptr;
      double T = 1.0, B = 0.0;
      double bprev = (lo < nlev) ? bj[lo - 1] : 0.0;
      if (!sorted) {
        // short segments, several rounds, or ray-traced path factors (they vary with the level): thin or not is
        // voted per step
        for (int i = lo; i < hi; ++i) {
          const double tl = tj[i] * ((OPT && fr) ? fr[i] : am);
          const double E = wave_all(fabs(tl) <= EXP_SMALL_X) ? fexp_small(-tl) : fexp(-tl);
          const double bi = bj[i];
          const double lay = fdiv1(__builtin_fma(bi, E, bprev), 1.0 + E);
          B = __builtin_fma(lay * T, 1.0 - E, B);
          T *= E;
          bprev = bi;
        }
      } else if (wave_all(it0 < nthin_all)) {
        // boflay (1 - E) = (B_{i-1} + B_i E) (1 - E)/(1 + E) = (B_{i-1} + B_i E) tanh(tau/2)
        auto step = [&](double tz, double bi) {
          const double tl = tz * am;
          const double E = fexp_small(-tl);
          const double th = ftanh_half_small(tl);
          B = __builtin_fma(__builtin_fma(bi, E, bprev) * T, th, B);
          T *= E;
          bprev = bi;
        };
        // two layers per trip, the next trip's rows already in flight (index hi <= nlev is inside the padded row)
        int i = lo;
        double t0 = (i < hi) ? tj[i] : 0.0, b0 = (i < hi) ? bj[i] : 0.0;
        for (; i + 1 < hi; i += 2) {
          const double t1 = tj[i + 1], b1 = bj[i + 1];
          const double t2 = tj[i + 2], b2 = bj[i + 2];
          step(t0, b0); step(t1, b1);
          t0 = t2; b0 = b2;
        }
        if (i < hi) step(t0, b0);
      } else {
        auto step = [&](double tz, double bi) {
          const double tl = tz * am;
          const double E = fexp(-tl);
          const double lay = fdiv1(__builtin_fma(bi, E, bprev), 1.0 + E);
          B = __builtin_fma(lay * T, 1.0 - E, B);
          T *= E;
          bprev = bi;
        };
        int i = lo;
        double t0 = (i < hi) ? tj[i] : 0.0, b0 = (i < hi) ? bj[i] : 0.0;
        for (; i + 1 < hi; i += 2) {
          const double t1 = tj[i + 1], b1 = bj[i + 1];
          const double t2 = tj[i + 2], b2 = bj[i + 2];
          step(t0, b0); step(t1, b1);
          t0 = t2; b0 = b2;
        }
        if (i < hi) step(t0, b0);
      }
      part[2 * it + 0] = B; part[2 * it + 1] = T;
    }
    MWRT_STAMP(5 + h);
    __syncthreads();
    for (int pr = tid; pr < npairs; pr += nthreads) {
      const int jj = div_small(pr, nang, A.g.magic_nang);
      const int a = pr - jj * nang;
      const int j = h * NFK + jj;
      double B = 0.0, T = 1.0;
      for (int sg = 0; sg < nseg; ++sg) {
        const double* q = part + 2 * (pr * nseg + sg);
        B = __builtin_fma(T, q[0], B);
        T *= q[1];
      }
      const double hvk = cfrq[jbase + j] * hk;
      double boftotl, boftmr;
      // T is exp(-tauprof) of the whole path; pyrtlib's "tauprof < 125" cut is T > exp(-125) (beyond it the
      // cosmic term is 1e-54 of B either way)
      if (T > TRANS_MIN) {
        const double ex = T;
        const double bbg = sfq[4 * NFC + 2 + j];
        boftotl = __builtin_fma(bbg, ex, B);
        boftmr = EXTRAS ? fdiv(B, 1.0 - ex) : 0.0;
      } else {
        boftotl = B; boftmr = B;
      }
      const int64_t o = (prof * nang + a) * A.nf + jbase + j;
      A.tb[o] = fdiv(hvk, flog(1.0 + fdiv(1.0, boftotl)));
      if constexpr (EXTRAS) {
        if (A.tbatm) A.tbatm[o] = fdiv(hvk, flog(1.0 + fdiv(1.0, B)));
        if (A.tmr) A.tmr[o] = fdiv(hvk, flog(1.0 + fdiv(1.0, boftmr)));
        if (want_tau && !rays) {
          const double am = cam[a];
          double sw = 0.0, sd = 0.0, sl = 0.0, si = 0.0;
#pragma unroll
          for (int q2 = 0; q2 < NFK; ++q2) if (q2 == jj) { sw = swet[q2]; sd = sdry[q2]; sl = sliq[q2]; si = sice[q2]; }
          if (A.tauwet) A.tauwet[o] = sw * am;
          if (A.taudry) A.taudry[o] = sd * am;
          if (A.tauliq) A.tauliq[o] = sl * am;
          if (A.tauice) A.tauice[o] = si * am;
        }
      }
    }
  }
  if constexpr (OPT) {
    // a trapped ray (ducting) leaves its angle NaN and marks the profile 3
    if (A.duct && A.duct[pin]) { if (tid == 0) A.valid[prof] = 3; return; }
  }
  if (A.write_valid && tid == 0) A.valid[prof] = 1;
  MWRT_STAMP(7);
}

// ---------------------------------------------------------------------------------------------
// K1 alone: awet / adry [nprof][nf][nlev] (RTEquation.clearsky_absorption [EXT]) -- or, TAU = true, K1 + the layer
// step: zenith layer optical depth tau [nprof][nlev][fpitch] (exponential_integration(zeroflg = True) on wet and
// dry, summed), 8 B per (profile, level, frequency) instead of 16, frequency fastest: what k_rte_tau reads with
// lane = frequency.
//
// TAU-mode level mapping.  The layer step needs level i-1 next to level i.  Lanes are levels, so the neighbour is
// one lane down (a DPP shift, no LDS) -- except across wave seams.  Instead of passing seam values through LDS
// behind a workgroup barrier, each wave REPEATS the last level of the wave below in its lane 0:
//     level(wave, lane) = 63 * wave + lane,    lane 0 of waves >= 1 is a duplicate that stores nothing.
// 3 waves cover 190 levels (the reference's 180), and the waves of a workgroup never wait for each other inside
// the frequency loop.
// ---------------------------------------------------------------------------------------------
constexpr int TAU_NFC = 16;                   // tau rows are written in 16-frequency (128-byte) pieces
__host__ __device__ constexpr int tau_threads(int nlev) { return ((nlev - 1 + (WAVE - 2)) / (WAVE - 1)) * WAVE; }

struct TauOut {
  const double* z;         // [nprof][nlev] km (layer thickness)
  double* tau;             // [nprof][nlev][fpitch]; row 0 (the ground level) is 0
  uint8_t* valid;          // [nprof], preset to 1 by the host; lowered to 0 (NaN input) / raised to 2 (negative absorption)
  int fpitch;              // doubles between consecutive levels: a multiple of 16, >= 16 * ceil(nf / 16)
};

// value of lane - 1 (lane 0 keeps its own): GFX9 wave_shr:1, two v_mov_b32_dpp, no LDS crossbar
__device__ __forceinline__ double lane_below(double v) {
  const long long b = __builtin_bit_cast(long long, v);
  const int lo = (int)b, hi = (int)(b >> 32);
  const int plo = __builtin_amdgcn_update_dpp(lo, lo, 0x138, 0xf, 0xf, false);
  const int phi = __builtin_amdgcn_update_dpp(hi, hi, 0x138, 0xf, 0xf, false);
  return __builtin_bit_cast(double, ((long long)phi << 32) | (unsigned)plo);
}

// One chunk of layer optical depths: lane = level holds tz[16] = its row's 16 frequencies = one 128-byte line, written
// in eight 16-byte pieces.  (Transposing 4 x 4 blocks of pieces across each quad of lanes first, so that a store
// instruction has every quad write 64 contiguous bytes, was measured: same-box A/B 3.93 vs 3.82 ms -- the 128 DPP
// moves cost more than the fuller memory requests save; so were nontemporal stores: no difference; a fully coalesced
// (level-fastest, i.e. wrong) layout as a timing experiment: 3.60 / 3.72 vs 3.74 / 3.75 ms -- the pattern is not the cost.)
//   lev0 = level of lane 0 of this wave; a lane's row is stored when `row_ok(level, lane)` (duplicate / padding rows are not).
template <class RowOk>
__device__ __forceinline__ void store_tau_chunk(const double (&tz)[TAU_NFC], double* tau_prof /* + jbase */, int64_t fpitch,
                                                int lev0, int lane, RowOk row_ok) {
  if (row_ok(lev0 + lane, lane)) {
    double2* row = (double2*)(tau_prof + (int64_t)(lev0 + lane) * fpitch);
#pragma unroll
    for (int j = 0; j < TAU_NFC; j += 2) row[j / 2] = double2{tz[j], tz[j + 1]};
  }
}

// the chunk's frequency table {f, f^2} x NFC, {fmin, fmax}, N2 factor x NFC in a WAVE-PRIVATE piece of LDS: filled
// and read by the same wave, so no workgroup barrier separates consecutive chunks
template <int NFC, class ModelPtr>
__device__ __forceinline__ void fill_chunk_table(double* sfq, ModelPtr M, cdoubles cfrq, int jbase, int nfc, int lane) {
  __builtin_amdgcn_wave_barrier();
  asm volatile("" ::: "memory");                      // the previous chunk's reads stay above the refill
  if (lane < NFC) {
    const double f = cfrq[jbase + min(lane, nfc - 1)];
    sfq[2 * lane] = f; sfq[2 * lane + 1] = f * f;
    double fdep = 1.0;
    if (M->n2_fdep) { const double q = f * (1.0 / 450.0); fdep = 0.5 + fdiv(0.5, 1.0 + q * q); }
    sfq[2 * NFC + 2 + lane] = fdep;
  }
  if (lane == WAVE - 1) {
    double lo = cfrq[jbase], hi = lo;
    for (int j = 1; j < nfc; ++j) { const double f = cfrq[jbase + j]; lo = fmin(lo, f); hi = fmax(hi, f); }
    sfq[2 * NFC] = lo; sfq[2 * NFC + 1] = hi;
  }
  __builtin_amdgcn_wave_barrier();
  asm volatile("" ::: "memory");
}

// NaN rows for chunks [c0, c0 + nch) of one profile (NaN input, negative absorption): k_rte_tau turns them into NaN TBs
__device__ __forceinline__ void blank_tau(const TauOut& T, int64_t prof, int nlev, int c0, int nch, int tid, int nthreads) {
  const double qnan = __builtin_nan("");
  const int w = nch * TAU_NFC;
  for (int it = tid; it < nlev * w; it += nthreads) {
    const int l = it / w, k = it - l * w;
    T.tau[(prof * nlev + l) * (int64_t)T.fpitch + c0 * TAU_NFC + k] = qnan;
  }
}

struct AbsorbArgs {
  const ModelFlat* M;
  const double* p; const double* t; const double* rh;
  const double* frq;
  double* awet; double* adry;
  int nlev, nf;
  TauOut T;                // TAU instantiations only
  const LineMasks* masks;  // [nchunks]
};

template <int NFC, int MAXT, bool TAU = false>
__global__ void __launch_bounds__(MAXT)
k_absorb(const AbsorbArgs A) {
  static_assert(!TAU || NFC == TAU_NFC, "tau rows are written in 16-frequency pieces");
  const int tid = threadIdx.x;
  const int lane = tid & (WAVE - 1), wave = tid / WAVE;
  const int64_t prof = blockIdx.x;
  const int jbase = blockIdx.y * NFC;
  const int nfc = min(NFC, A.nf - jbase);
  const cmodel M = (cmodel)A.M;
  const cdoubles cfrq = (cdoubles)A.frq;
  __shared__ double sfq_w[MAXT / WAVE][3 * NFC + 2];
  double* sfq = sfq_w[wave];
  fill_chunk_table<NFC>(sfq, M, cfrq, jbase, nfc, lane);
  const int lev = TAU ? wave * (WAVE - 1) + lane : tid;
  const bool active = lev < A.nlev;
  const int64_t off = prof * A.nlev + (active ? lev : 0);
  const double pi = A.p[off], ti = A.t[off], rhi = A.rh[off];
  double zi = 0.0;
  if constexpr (TAU) {
    zi = A.T.z[off];
    if (__syncthreads_or(active && (isnan(zi) || isnan(pi) || isnan(ti) || isnan(rhi)))) {   // check_for_nans
      blank_tau(A.T, prof, A.nlev, blockIdx.y, 1, tid, blockDim.x);
      if (tid == 0) A.T.valid[prof] = 0;
      return;
    }
  }
  double awet[NFC], adry[NFC];
  const double e = goff_gratch_e(ti, rhi);
  const LevelState L = level_state(pi, ti, e);
  const LineMasks lm = load_masks(A.masks, blockIdx.y);
  h2o_absorb<NFC>(M, L, sfq, lm, awet);
  dry_absorb<NFC>(M, L, sfq, lm, adry);
  if constexpr (!TAU) {
    if (active) {
#pragma unroll
      for (int j = 0; j < NFC; ++j) {
        if (j < nfc) {
          const int64_t o = (prof * A.nf + jbase + j) * A.nlev + tid;
          A.awet[o] = awet[j];
          A.adry[o] = adry[j];
        }
      }
    }
  } else {
    const bool has_prev = active && lev > 0 && lane > 0;
    const double z0 = A.T.z[prof * A.nlev];
    const double dz = has_prev ? ((zi - z0) - (A.T.z[off - 1] - z0)) : 0.0;
    bool neg = false;
    double tz[NFC];
#pragma unroll
    for (int j = 0; j < NFC; j += 4) {
#pragma clang fp contract(off)               // wet * dz + dry * dz rounds as in the fused kernel
      double w4[4] = {awet[j], awet[j + 1], awet[j + 2], awet[j + 3]};
      double d4[4] = {adry[j], adry[j + 1], adry[j + 2], adry[j + 3]};
      const double wb[4] = {lane_below(w4[0]), lane_below(w4[1]), lane_below(w4[2]), lane_below(w4[3])};
      const double db[4] = {lane_below(d4[0]), lane_below(d4[1]), lane_below(d4[2]), lane_below(d4[3])};
      layer_value4(w4, wb, neg, has_prev);
      layer_value4(d4, db, neg, has_prev);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const double twj = has_prev ? w4[k] * dz : 0.0;
        const double tdj = has_prev ? d4[k] * dz : 0.0;
        tz[j + k] = twj + tdj;
      }
    }
    {
      const int nlev = A.nlev;
      store_tau_chunk(tz, A.T.tau + prof * nlev * (int64_t)A.T.fpitch + jbase, A.T.fpitch, wave * (WAVE - 1), lane,
                      [&](int l, int ln) { return l < nlev && (ln > 0 || wave == 0); });
    }
    if (__syncthreads_or(neg)) {              // pyrtlib raises ValueError here: flag 2, NaN out
      blank_tau(A.T, prof, A.nlev, blockIdx.y, 1, tid, blockDim.x);
      if (tid == 0) A.T.valid[prof] = 2;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// K1 on fine spectral grids: windowed evaluation (k_absorb_win)
//
// On a grid of hundreds of frequencies most of a chunk's 65 lines are far from the whole NEIGHBOURHOOD of the
// chunk, and their sum is an analytic function of f there (poles at c_k +- i w_k, at least WIN_MARGIN_GHZ beyond
// the window's ends).  A workgroup therefore owns a WINDOW of WIN_CHUNKS consecutive 16-frequency chunks of one
// profile: it sums the window-far lines once at WIN_NODES Chebyshev nodes of the window (same line bodies as
// everywhere else), and for each chunk interpolates those sums to the chunk's frequencies with a precomputed
// Lagrange matrix (wave-uniform, from the host: depends on the frequencies only) and adds the remaining lines
// -- near the window, speed dependent, or failing a per-level vote -- directly.  The per-(level, line) setup of
// the far lines is paid once per window instead of once per chunk.
// Interpolation error: <= 1e-10 of the line sum for spans <= 6 GHz (16 nodes, 4 GHz margin; tools/window_probe.py
// reproduces the bound on the oracle), i.e. invisible against the 1e-6 K parity bar -- and tested against it.
//
// TAU = true: the chunk ends with the layer step (see k_absorb) and writes the zenith layer optical depth,
// [level][frequency], 8 B per point; the fine-grid TB path is this kernel followed by k_rte_tau.
// ---------------------------------------------------------------------------------------------
constexpr int WIN_NODES = 16;          // O2: lines from WIN_MARGIN_GHZ beyond the window
constexpr int WIN_NODES_H = 8;         // H2O: lines from WIN_H2O_MARGIN_GHZ beyond it -- so smooth across the window that 8
                                       // nodes do (convergence ~ 25^-n); a third less LDS = a fourth workgroup per CU
constexpr int WIN_CHUNKS = 8;          // chunks of a base window
constexpr int WIN_CHUNKS_MAX = 16;     // ... of a merged one (two neighbours with no line near either: one node phase for both)
constexpr int WIN_NFC = 16;

struct WinDesc {                       // one per window, built by the host (csrc/mwrt.hip: build_windows)
  double fnode[WIN_NODES];             // Chebyshev nodes of [f_lo, f_hi], GHz
  double fnode_h[WIN_NODES_H];
  double flo, fhi;                     // the window itself
  unsigned long long o2_far;           // O2 lines >= WIN_MARGIN_GHZ beyond the window
  unsigned h2o_far_both, h2o_far_res;  // H2O lines >= WIN_H2O_MARGIN_GHZ beyond the window with a cutoff state uniform across it
  int first_chunk, nchunks;            // chunks [first_chunk, first_chunk + nchunks) of the frequency list
  int pad0, pad1;
};

struct AbsorbWinArgs {
  const ModelFlat* M;
  const double* p; const double* t; const double* rh;
  const double* frq;
  const WinDesc* win;                  // [nwin]
  const double* lagrange;              // [nwin][WIN_CHUNKS_MAX][WIN_NODES][WIN_NFC]: weight of node m for target j of chunk c
  const double* lagrange_h;            // [nwin][WIN_CHUNKS_MAX][WIN_NODES_H][WIN_NFC]
  const LineMasks* masks;              // [nchunks of the list]: line_masks() of every chunk, precomputed (it depends on the
                                       // frequencies and the table only)
  const double* lag_sd;                // [nchunks][SD_TARGETS][SD_NODES]: the half-sampled SD shape's weights per chunk
  double* awet; double* adry;
  int nlev, nf;
  TauOut T;                            // TAU instantiations only
};

template <int MAXT, bool TAU = false>
__global__ void __launch_bounds__(MAXT, (MAXT <= 256 ? 3 : 1))
k_absorb_win(const AbsorbWinArgs A) {
  constexpr int NFC = WIN_NFC, NN = WIN_NODES, NH = WIN_NODES_H;
  static_assert(NN == NFC, "the node set is evaluated through the NFC-wide line bodies");
  static_assert(NFC == TAU_NFC, "tau rows are written in 16-frequency pieces");
  const int tid = threadIdx.x, lane = tid & (WAVE - 1), wave = tid / WAVE;
  const int64_t prof = blockIdx.x;
  const cmodel M = (cmodel)A.M;
  const cdoubles cfrq = (cdoubles)A.frq;
  typedef const __attribute__((address_space(4))) WinDesc* cwin;
  const cwin D = (cwin)(A.win + blockIdx.y);
  const cdoubles Lw = (cdoubles)(A.lagrange + (size_t)blockIdx.y * WIN_CHUNKS_MAX * NFC * NN);
  const cdoubles Lwh = (cdoubles)(A.lagrange_h + (size_t)blockIdx.y * WIN_CHUNKS_MAX * NFC * NH);
  __shared__ double sfn[3 * NFC + 2];                        // the nodes, laid out like a chunk
  __shared__ double sfn_h[3 * NH + 2];                       // the H2O nodes
  __shared__ double sfq_w[MAXT / WAVE][3 * NFC + 2];         // the current chunk, one copy per wave (fill_chunk_table)
  double* sfq = sfq_w[wave];
  if (tid < NN) {
    const double f = D->fnode[tid];
    sfn[2 * tid] = f; sfn[2 * tid + 1] = f * f; sfn[2 * NFC + 2 + tid] = 1.0;
  }
  if (tid >= NN && tid < NN + NH) {                          // (a one-wave workgroup has lanes 16 .. 23 too)
    const double f = D->fnode_h[tid - NN];
    sfn_h[2 * (tid - NN)] = f; sfn_h[2 * (tid - NN) + 1] = f * f; sfn_h[2 * NH + 2 + (tid - NN)] = 1.0;
  }
  if (tid == WAVE - 1) {                                     // cutoff / range votes at the nodes speak for the whole window
    sfn[2 * NFC] = D->flo; sfn[2 * NFC + 1] = D->fhi;
    sfn_h[2 * NH] = D->flo; sfn_h[2 * NH + 1] = D->fhi;
  }
  const int lev = TAU ? wave * (WAVE - 1) + lane : tid;
  const bool active = lev < A.nlev;
  const int64_t off = prof * A.nlev + (active ? lev : 0);
  const double pi = A.p[off], ti = A.t[off], rhi = A.rh[off];
  const int nch = D->nchunks;
  double zi = 0.0;
  bool bad = false;
  if constexpr (TAU) { zi = A.T.z[off]; bad = active && (isnan(zi) || isnan(pi) || isnan(ti) || isnan(rhi)); }
  if (__syncthreads_or(bad)) {                               // (also publishes sfn) check_for_nans: NaN out, valid = 0
    blank_tau(A.T, prof, A.nlev, D->first_chunk, nch, tid, blockDim.x);
    if (tid == 0) A.T.valid[prof] = 0;
    return;
  }
  const double e = goff_gratch_e(ti, rhi);
  const LevelState L = level_state(pi, ti, e);

  // ---- window-far lines at the nodes ----
  // The node sums live in LDS, [node][thread] (each lane reads back only its own column: conflict-free), not in
  // registers: 32 doubles per lane would cost the kernel two waves of occupancy.
  extern __shared__ __attribute__((aligned(16))) double wlds[];
  const int nthreads = blockDim.x;
  double* Sh_l = wlds;                                        // [NH][nthreads]
  double* So_l = wlds + (size_t)NH * nthreads;                // [NN][nthreads]
  const unsigned wf_both = D->h2o_far_both, wf_res = D->h2o_far_res;
  const unsigned long long wf_o2 = D->o2_far;
  double bsum_far = 0.0;
  unsigned failed_h = 0u;
  unsigned long long failed_o = 0ull;
  {
    LineMasks ln;
    ln.o2_far = wf_o2; ln.h2o_far = wf_both | wf_res; ln.h2o_res = wf_res; ln.h2o_none = 0u; ln.h2o_sd = 0u; ln.h2o_sdfar = 0u; ln.h2o_sdint = 0u;
    {
      double Sh[NH];
      h2o_absorb<NH, true>(M, L, sfn_h, ln, Sh, ~(wf_both | wf_res), nullptr, 0.0, &failed_h, &bsum_far);
#pragma unroll
      for (int m = 0; m < NH; ++m) Sh_l[m * nthreads + tid] = Sh[m];
    }
    double S[NN];
    dry_absorb<NFC, true>(M, L, sfn, ln, S, ~wf_o2, nullptr, &failed_o);
#pragma unroll
    for (int m = 0; m < NN; ++m) So_l[m * nthreads + tid] = S[m];
  }
  const unsigned excl_h = (wf_both | wf_res) & ~failed_h;     // a line that failed its vote at some level of this wave
  const unsigned long long excl_o = wf_o2 & ~failed_o;        // was left out of the node sums: evaluated directly
  // node sums -> a chunk's frequencies: out[j] = sum_m Lt[m][j] S[m]; the matrix is wave-uniform (scalar loads),
  // stored node-major so one node's 16 weights are one contiguous load
  auto interpolate = [&](const double* S_l, cdoubles Lt, int nn, double (&out)[NFC]) {
#pragma unroll
    for (int j = 0; j < NFC; ++j) out[j] = 0.0;
#pragma unroll 1
    for (int m = 0; m < nn; ++m) {                            // one node per trip: 16 scalar weights live at a time
      const double sm = S_l[m * nthreads + tid];
#pragma unroll
      for (int j = 0; j < NFC; ++j) out[j] = __builtin_fma((MWRT_ABLATE & 64) ? 0.0625 + 0.001 * j : Lt[m * NFC + j], sm, out[j]);
    }
  };

  // layer thickness below this level (TAU)
  bool has_prev = false, neg = false;
  double dz = 0.0;
  if constexpr (TAU) {
    has_prev = active && lev > 0 && lane > 0;
    const double z0 = A.T.z[prof * A.nlev];
    dz = has_prev ? ((zi - z0) - (A.T.z[off - 1] - z0)) : 0.0;
  }

  // ---- the window's chunks (no workgroup barrier inside: the waves drift apart and fill each other's stalls) ----
  for (int c = 0; c < nch; ++c) {
    const int jbase = (D->first_chunk + c) * NFC;
    const int nfc = min(NFC, A.nf - jbase);
    fill_chunk_table<NFC>(sfq, M, cfrq, jbase, nfc, lane);
    const cdoubles Lt = Lw + (size_t)c * NN * NFC;            // [node][target] of this chunk
    const cdoubles Lth = Lwh + (size_t)c * NH * NFC;
    const LineMasks lm = load_masks(A.masks, D->first_chunk + c);
    // The level state is the same for every chunk, and the compiler would hoist every per-(level, line) quantity
    // of the direct lines out of the chunk loop (hundreds of registers).  Laundering it keeps them inside.
    LevelState Lc = L;
    asm volatile("" : "+v"(Lc.t), "+v"(Lc.p), "+v"(Lc.rho), "+v"(Lc.pdry));
    double init[NFC], awet[NFC], adry[NFC];
    if constexpr (!TAU) {
      interpolate(Sh_l, Lth, NH, init);
      h2o_absorb<NFC>(M, Lc, sfq, lm, awet, excl_h, init, bsum_far, nullptr, nullptr,
                      (cdoubles)(A.lag_sd + (size_t)(D->first_chunk + c) * SD_TARGETS * SD_NODES));
      if (active) {
#pragma unroll
        for (int j = 0; j < NFC; ++j)
          if (j < nfc) A.awet[(prof * A.nf + jbase + j) * A.nlev + tid] = awet[j];
      }
      interpolate(So_l, Lt, NN, init);
      dry_absorb<NFC>(M, Lc, sfq, lm, adry, excl_o, init);
      if (active) {
#pragma unroll
        for (int j = 0; j < NFC; ++j)
          if (j < nfc) A.adry[(prof * A.nf + jbase + j) * A.nlev + tid] = adry[j];
      }
    } else {
      // One species is evaluated, run through the layer rule and PARKED (16 doubles) while the other is evaluated.
      // Wet first measured better than dry first on the real translation unit: 40 spilled VGPRs / 156 B of scratch per
      // lane against 52 / 212, and 3.70-3.77 against 3.83-3.93 ms on one box.
      auto layer_rows = [&](double (&x)[NFC]) {                // absorption -> layer optical depth of the layer below, in place
#pragma unroll
        for (int j = 0; j < NFC; j += 4) {
#pragma clang fp contract(off)
          double v4[4] = {x[j], x[j + 1], x[j + 2], x[j + 3]};
          const double vb[4] = {lane_below(v4[0]), lane_below(v4[1]), lane_below(v4[2]), lane_below(v4[3])};
          if (!(MWRT_ABLATE & 16)) layer_value4(v4, vb, neg, has_prev);
#pragma unroll
          for (int k = 0; k < 4; ++k) x[j + k] = has_prev ? v4[k] * dz : 0.0;
        }
      };
      interpolate(Sh_l, Lth, NH, init);
      h2o_absorb<NFC>(M, Lc, sfq, lm, awet, excl_h, init, bsum_far, nullptr, nullptr,
                      (cdoubles)(A.lag_sd + (size_t)(D->first_chunk + c) * SD_TARGETS * SD_NODES));
      layer_rows(awet);
      interpolate(So_l, Lt, NN, init);
      dry_absorb<NFC>(M, Lc, sfq, lm, adry, excl_o, init);
      layer_rows(adry);
#pragma unroll
      for (int j = 0; j < NFC; ++j) {
#pragma clang fp contract(off)               // wet * dz + dry * dz rounds as in the fused kernel
        adry[j] = awet[j] + adry[j];
      }
      if (!((MWRT_ABLATE & 32) && adry[0] != -1.0)) {
        const int nlev = A.nlev;
        store_tau_chunk(adry, A.T.tau + prof * nlev * (int64_t)A.T.fpitch + jbase, A.T.fpitch, wave * (WAVE - 1), lane,
                        [&](int l, int ln) { return l < nlev && (ln > 0 || wave == 0); });
      }
    }
  }
  if constexpr (TAU) {
    if (__syncthreads_or(neg)) {                // pyrtlib raises ValueError here: flag 2, NaN out
      blank_tau(A.T, prof, A.nlev, D->first_chunk, nch, tid, nthreads);
      if (tid == 0) A.T.valid[prof] = 2;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// K2 on fine spectral grids: downwelling Planck-space RTE (RTEquation.planck, from_sat = False, + bright [EXT]) from
// zenith layer optical depths in HBM, tau [nprof][nlev][fpitch] as the TAU absorption kernels write them.
//
// LANE = FREQUENCY.  A wave owns 64 consecutive frequencies of one profile and walks the levels serially: each step
// is one coalesced 512-byte row read (issued PF levels ahead), the Planck function of the level once per frequency,
// and the NA slant-path recursions in registers (B_a, T_a per elevation).  No LDS traffic in the loop (two
// broadcast reads of the level's h/kT), no barriers, no work split to recombine; thin or general layer step is voted
// per (level, elevation) by the wave -- neighbouring frequencies have neighbouring optical depths.
// Algorithmic traffic: 8 B per (profile, level, frequency) in, 8 B per TB out.
// ---------------------------------------------------------------------------------------------
struct RteTauArgs {
  const ModelFlat* M;
  const double* tau;       // [nprof][nlev][fpitch]
  const double* t;         // [nprof][nlev] K
  const double* frq;       // [nf] GHz
  const double* airmass;   // [nang]; elevations a0 .. a0 + NA - 1 are this launch's
  double* tb;              // [nprof][nang][nf]
  const uint8_t* valid;    // [nprof] as the absorption kernel left it: != 1 -> NaN rows
  int nlev, nf, nang, fpitch, a0;
};

constexpr int RTE_THREADS = 256;
constexpr int RTE_PF = 8;        // levels in flight per lane

template <int NA>
__global__ void __launch_bounds__(RTE_THREADS)
k_rte_tau(const RteTauArgs A) {
  extern __shared__ __attribute__((aligned(16))) double hkl[];    // {h/(k T_i) per GHz, its inverse} per level
  const int tid = threadIdx.x;
  const int64_t prof = blockIdx.x;
  const int nlev = A.nlev, nf = A.nf;
  const cmodel M = (cmodel)A.M;
  const cdoubles cam = (cdoubles)A.airmass;
  const double hk = 1e9 * M->planck_h / M->boltzmann_k;
  const double inv_hk = 1e-9 * M->boltzmann_k / M->planck_h;
  for (int l = tid; l < nlev; l += RTE_THREADS) {
    const double ti = A.t[prof * nlev + l];
    hkl[2 * l] = fdiv(hk, ti);
    hkl[2 * l + 1] = ti * inv_hk;
  }
  __syncthreads();
  const int f0 = blockIdx.y * RTE_THREADS + (tid & ~(WAVE - 1));
  if (f0 >= nf) return;                                           // a wave past the last frequency
  const int fi = blockIdx.y * RTE_THREADS + tid;
  const bool live = fi < nf;
  const int fc = live ? fi : nf - 1;                              // idle lanes shadow the last frequency (votes stay clean)
  const double qnan = __builtin_nan("");
  if (A.valid[prof] != 1) {                                       // NaN input / negative absorption: every TB of the profile is NaN
    if (live) {
#pragma unroll
      for (int a = 0; a < NA; ++a) A.tb[(prof * A.nang + A.a0 + a) * nf + fi] = qnan;
    }
    return;
  }
  const double f = A.frq[fc];
  const double rf = fdiv(1.0, f);
  double am[NA], B[NA], T[NA];
  double am_max = 0.0;                                            // NaN air masses (their rows come out NaN either way) aside
#pragma unroll
  for (int a = 0; a < NA; ++a) { am[a] = cam[A.a0 + a]; B[a] = 0.0; T[a] = 1.0; am_max = fmax(am_max, fabs(am[a])); }
  const double* col = A.tau + prof * nlev * (int64_t)A.fpitch + fc;
  const int64_t pitch = A.fpitch;
  double bprev = planck_b(f * hkl[0], rf * hkl[1]);
  double cur[RTE_PF];
#pragma unroll
  for (int k = 0; k < RTE_PF; ++k) cur[k] = col[(int64_t)min(1 + k, nlev - 1) * pitch];
  for (int i0 = 1; i0 < nlev; i0 += RTE_PF) {
    double nxt[RTE_PF];
#pragma unroll
    for (int k = 0; k < RTE_PF; ++k) nxt[k] = col[(int64_t)min(i0 + RTE_PF + k, nlev - 1) * pitch];
#pragma unroll
    for (int k = 0; k < RTE_PF; ++k) {
      const int i = i0 + k;
      if (i < nlev) {
        const double tz = cur[k];
        const double bi = planck_b(f * hkl[2 * i], rf * hkl[2 * i + 1]);
        // boflay (1 - E) = (B_{i-1} + B_i E) (1 - E)/(1 + E) = (B_{i-1} + B_i E) tanh(tau/2): the thin-layer step
        auto thin_step = [&](int a, double tl) {
          const double E = fexp_small(-tl);
          const double th = ftanh_half_small(tl);
          B[a] = __builtin_fma(__builtin_fma(bi, E, bprev) * T[a], th, B[a]);
          T[a] *= E;
        };
        if (wave_all(!(fabs(tz) * am_max > EXP_TINY_X))) {           // tiny at the longest path: the short series for every elevation
          KEEP_BRANCH();
#pragma unroll
          for (int a = 0; a < NA; ++a) {
            const double tl = tz * am[a];
            const double E = fexp_tiny(-tl);
            B[a] = __builtin_fma(__builtin_fma(bi, E, bprev) * T[a], ftanh_half_tiny(tl), B[a]);
            T[a] *= E;
          }
        } else if (wave_all(!(fabs(tz) * am_max > EXP_SMALL_X))) {   // thin at the longest path: thin at all of them, one vote
          KEEP_BRANCH();
#pragma unroll
          for (int a = 0; a < NA; ++a) thin_step(a, tz * am[a]);
        } else {
#pragma unroll
          for (int a = 0; a < NA; ++a) {
            const double tl = tz * am[a];
            if (wave_all(!(fabs(tl) > EXP_SMALL_X))) {
              thin_step(a, tl);
            } else {
              const double E = fexp(-tl);
              const double lay = fdiv1(__builtin_fma(bi, E, bprev), 1.0 + E);
              B[a] = __builtin_fma(lay * T[a], 1.0 - E, B[a]);
              T[a] *= E;
            }
          }
        }
        bprev = bi;
      }
    }
#pragma unroll
    for (int k = 0; k < RTE_PF; ++k) cur[k] = nxt[k];
  }
  if (!live) return;
  const double hvk = f * hk;
  const double bbg = fdiv(1.0, fexp(fdiv(hvk, M->t_cosmic)) - 1.0);   // B(T_cosmic, f)
#pragma unroll
  for (int a = 0; a < NA; ++a) {
    // T is exp(-tauprof) of the whole path; pyrtlib's "tauprof < 125" cut is T > exp(-125)
    const double boftotl = (T[a] > TRANS_MIN) ? __builtin_fma(bbg, T[a], B[a]) : B[a];
    A.tb[(prof * A.nang + A.a0 + a) * nf + fi] = fdiv(hvk, flog(1.0 + fdiv(1.0, boftotl)));
  }
}

#ifdef MWRT_HOST_TU
// ---------------------------------------------------------------------------------------------
// K-matrix: dTB/dT, dTB/de, dTB/d(layer thickness) per level from ONE pass -- the adjoint of the layer rule and of
// the Planck-space recursion (RTEquation.exponential_integration / planck / bright [EXT]) applied to absorption
// derivatives.  The reference parses exactly this block out of RTTOV-gb's K run (RTTOV_gb_processing.py:286-300,
// :418-432); round 2 produced it with 3 nlev + 1 forward runs.
//
// Inputs: absorption at the profile's levels and at four locally perturbed states (T +- dT at fixed e, e (1 +- re) at
// fixed T), each [nprof][nf][nlev] as k_absorb writes them -- absorption is a LOCAL function of (p, T, e), so its
// partial derivatives are central differences of five evaluations per level, whatever the number of levels.
// Everything downstream is differentiated analytically:
//   tau_l = (LM(aw_l, aw_{l-1}) + LM(ad_l, ad_{l-1})) dz_l am,
//   B_tot = sum_l c_l T_{l-1} + B_cosmic T_n,   c_l = (b_{l-1} + b_l E_l)(1 - E_l)/(1 + E_l),  E_l = exp(-tau_l),
//   dB_tot/dtau_l = T_{l-1} dc_l/dtau_l - (B_tot - sum_{m<=l} c_m T_{m-1})     (everything above layer l is dimmed),
//   dTB/dB_tot = hvk / (ln^2(1 + 1/B) B (B + 1)),   db/dT = b (b + 1) hvk / T^2.
// One thread per (profile, frequency, elevation): two serial walks over the levels (the first for B_tot).  An analysis
// product, not a throughput path.
// ---------------------------------------------------------------------------------------------
struct JacArgs {
  const ModelFlat* M;
  const double* z; const double* t;          // [nprof][nlev]
  const double* a[5][2];                     // {base, T+, T-, e+, e-} x {wet, dry}: [nprof][nf][nlev]
  const double* de;                          // [nprof][nlev] the absolute vapour-pressure step used for e+-  (hPa)
  double dT;
  const double* frq; const double* airmass;
  double* tb;                                // [nprof][nang][nf]
  double* dtb_dt; double* dtb_de; double* dtb_ddz;   // [nprof][nang][nf][nlev]
  uint8_t* valid;                            // [nprof] preset to 1; 0 NaN input, 2 negative absorption
  int64_t nprof; int nlev, nf, nang;
};

// log-mean layer value and its two partial derivatives, branch for branch as layer_value<true>
__device__ __forceinline__ double layer_value_grad(double x1, double x0, double& d1, double& d0, bool& neg) {
  if (x0 < 0.0 || x1 < 0.0) { neg = true; d1 = d0 = 0.0; return 0.0; }
  const double d = x1 - x0;
  if (fabs(d) < 1e-09) { d1 = 1.0; d0 = 0.0; return x1; }
  if (x0 == 0.0 || x1 == 0.0) { d1 = d0 = 0.5; return 0.5 * (x1 + x0); }
  const double sm = x1 + x0, s = d / sm;
  if (fabs(s) <= LOGMEAN_SMALL_S) {
    // L = sm/2 q(z), q = s/atanh(s), z = s^2:  dL/dx1 = q/2 + 2 s q'(z) x0/sm,  dL/dx0 = q/2 - 2 s q'(z) x1/sm
    const double z = s * s;
    const double c[10] = {-3.3333333333333333333e-01, -8.8888888888888888889e-02, -4.6560846560846560847e-02,
                          -3.0194003527336860670e-02, -2.1796804019026241248e-02, -1.6787551856334925118e-02,
                          -1.3502765051265933100e-02, -1.1203745637718733130e-02, -9.5160731945278989134e-03,
                          -8.2312065673505011548e-03};
    double q = 0.0, qp = 0.0;
    for (int k = 9; k >= 0; --k) { qp = qp * z + (k + 1) * c[k]; q = (q + c[k]) * z; }
    q += 1.0;
    d1 = 0.5 * q + 2.0 * s * qp * x0 / sm;
    d0 = 0.5 * q - 2.0 * s * qp * x1 / sm;
    return 0.5 * sm * q;
  }
  const double ln = log(x1 / x0), L = d / ln;
  d1 = (1.0 - L / x1) / ln;
  d0 = (L / x0 - 1.0) / ln;
  return L;
}

__global__ void __launch_bounds__(64)
k_tb_jacobian(const JacArgs A) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int nlev = A.nlev, nf = A.nf, nang = A.nang;
  if (gid >= A.nprof * nf * nang) return;
  const int a = (int)(gid % nang);
  const int j = (int)((gid / nang) % nf);
  const int64_t prof = gid / ((int64_t)nang * nf);
  const cmodel M = (cmodel)A.M;
  const double qnan = __builtin_nan("");
  const int64_t orow = ((prof * nang + a) * nf + j);
  double* o_t = A.dtb_dt + orow * nlev; double* o_e = A.dtb_de + orow * nlev; double* o_z = A.dtb_ddz + orow * nlev;
  const int64_t arow = (prof * nf + j) * nlev;
  const double* z = A.z + prof * nlev; const double* t = A.t + prof * nlev;
  const double am = A.airmass[a];
  const double f = A.frq[j];
  const double hvk = f * (1e9 * M->planck_h / M->boltzmann_k);
  bool bad = false;
  for (int l = 0; l < nlev; ++l)
    bad = bad || isnan(z[l]) || isnan(t[l]) || isnan(A.a[0][0][arow + l]) || isnan(A.a[0][1][arow + l]);
  auto blank = [&](uint8_t flag) {
    A.tb[orow] = qnan;
    for (int l = 0; l < nlev; ++l) { o_t[l] = qnan; o_e[l] = qnan; o_z[l] = qnan; }
    if (flag != 1) A.valid[prof] = flag;
  };
  if (bad) { blank(0); return; }
  const double* aw = A.a[0][0] + arow; const double* ad = A.a[0][1] + arow;
  // ---- walk 1: B_tot ----
  bool neg = false;
  double Btot = 0.0, T = 1.0;
  {
    double bprev = 1.0 / (exp(hvk / t[0]) - 1.0);
    for (int l = 1; l < nlev; ++l) {
      double g1, g0;
      const double dz = (z[l] - z[0]) - (z[l - 1] - z[0]);
      const double tau = (layer_value_grad(aw[l], aw[l - 1], g1, g0, neg) * dz + layer_value_grad(ad[l], ad[l - 1], g1, g0, neg) * dz) * am;
      const double E = exp(-tau);
      const double bl = 1.0 / (exp(hvk / t[l]) - 1.0);
      Btot += (bprev + bl * E) / (1.0 + E) * T * (1.0 - E);
      T *= E;
      bprev = bl;
    }
  }
  if (neg) { blank(2); return; }
  const double bbg = 1.0 / (exp(hvk / M->t_cosmic) - 1.0);
  const bool with_bg = T > TRANS_MIN;
  if (with_bg) Btot += bbg * T;
  const double Lg = log(1.0 + 1.0 / Btot);
  A.tb[orow] = hvk / Lg;
  const double dTB_dB = hvk / (Lg * Lg * Btot * (Btot + 1.0));
  if (isnan(am)) { blank(1); return; }                          // a NaN elevation: its rows are NaN, the profile stays valid
  // ---- walk 2: derivatives ----
  const double inv2dT = 0.5 / A.dT;
  auto dA = [&](int species, int lvl, bool wrt_e) {             // d(absorption)/dT or /de at a level, central difference
    const double hi = A.a[wrt_e ? 3 : 1][species][arow + lvl], lo = A.a[wrt_e ? 4 : 2][species][arow + lvl];
    return wrt_e ? (hi - lo) / (2.0 * A.de[prof * nlev + lvl]) : (hi - lo) * inv2dT;
  };
  double S = 0.0;                                               // sum_{m <= l} c_m T_{m-1}
  double Tm = 1.0;
  double b0 = 1.0 / (exp(hvk / t[0]) - 1.0);
  double acc_t = 0.0, acc_e = 0.0;                              // contributions to level l-1 collected so far
  o_z[0] = 0.0;
  for (int l = 1; l < nlev; ++l) {
    double w1, w0, d1, d0;
    const double dz = (z[l] - z[0]) - (z[l - 1] - z[0]);
    const double Lw = layer_value_grad(aw[l], aw[l - 1], w1, w0, neg), Ld = layer_value_grad(ad[l], ad[l - 1], d1, d0, neg);
    const double tau = (Lw * dz + Ld * dz) * am;
    const double E = exp(-tau);
    const double b1 = 1.0 / (exp(hvk / t[l]) - 1.0);
    const double opE = 1.0 + E, omE = 1.0 - E;
    const double c = (b0 + b1 * E) / opE * omE;
    S += c * Tm;
    const double dc_dtau = E * (2.0 * b0 + 2.0 * b1 * E - b1 + b1 * E * E) / (opE * opE);
    // everything above layer l (later layers and the cosmic term) is dimmed by E_l
    const double g = dTB_dB * (Tm * dc_dtau - (Btot - S));                       // dTB/dtau_l
    const double gk = g * am * dz;
    // level l-1 (lower end of the layer) and level l (upper end)
    acc_t += gk * (w0 * dA(0, l - 1, false) + d0 * dA(1, l - 1, false)) + dTB_dB * Tm * (omE / opE) * (b0 * (b0 + 1.0) * hvk / (t[l - 1] * t[l - 1]));
    acc_e += gk * (w0 * dA(0, l - 1, true) + d0 * dA(1, l - 1, true));
    o_t[l - 1] = acc_t; o_e[l - 1] = acc_e;
    acc_t = gk * (w1 * dA(0, l, false) + d1 * dA(1, l, false)) + dTB_dB * Tm * (E * omE / opE) * (b1 * (b1 + 1.0) * hvk / (t[l] * t[l]));
    acc_e = gk * (w1 * dA(0, l, true) + d1 * dA(1, l, true));
    o_z[l] = (dz != 0.0) ? g * tau / dz : 0.0;                                  // dTB / d(thickness of layer l) [K/km]
    Tm *= E;
    b0 = b1;
  }
  o_t[nlev - 1] = acc_t; o_e[nlev - 1] = acc_e;
}

// diagnostic: the local exp / log / division helpers on caller-supplied arguments (mwrt_selftest_math)
__global__ void k_selftest_math(const double* x, const double* y, double* out_exp, double* out_log, double* out_div,
                                double* out_div1, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  out_exp[i] = fexp(x[i]);
  out_log[i] = flog(y[i]);
  out_div[i] = fdiv(x[i], y[i]);
  out_div1[i] = fdiv1(x[i], y[i]);
}

#endif  // MWRT_HOST_TU

}  // namespace mwrt
