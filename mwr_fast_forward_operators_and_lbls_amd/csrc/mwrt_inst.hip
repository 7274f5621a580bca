// mwrt_inst.hip -- one translation unit per frequency-chunk width (compile with -DMWRT_INST_NFC=8|14|16):
// the fused-kernel and absorption-kernel instantiations of that width and their launchers.
#include "mwrt_inst.hip.h"

#ifndef MWRT_INST_NFC
#error "compile with -DMWRT_INST_NFC=8, 14 or 16"
#endif

namespace mwrt {

namespace {

constexpr int NFC = MWRT_INST_NFC;
constexpr int NFK = 8;

template <int MAXT, bool OPT, bool EXTRAS, bool ALPHA>
hipError_t launch_one(const FusedArgs& a, dim3 grid, dim3 block, size_t lds, hipStream_t st) {
  auto k = k_tb_fused<NFC, NFK, MAXT, OPT, EXTRAS, ALPHA>;
  hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k, grid, block, lds, st, a);
  return hipGetLastError();
}

// workgroup size class: 256 threads up to 256 levels; 512 threads get 256 VGPRs per lane (no scratch);
// only > 512 levels fall to the 1024-thread instantiation, whose 128-VGPR cap spills
// (profiles/r02_tall_profiles.txt)
template <bool OPT, bool EXTRAS, bool ALPHA = false>
hipError_t launch_by_size(const FusedArgs& a, dim3 grid, dim3 block, size_t lds, hipStream_t st) {
  if (block.x <= 256) return launch_one<256, OPT, EXTRAS, ALPHA>(a, grid, block, lds, st);
  if (block.x <= 512) return launch_one<512, OPT, EXTRAS, ALPHA>(a, grid, block, lds, st);
  return launch_one<1024, OPT, EXTRAS, ALPHA>(a, grid, block, lds, st);
}

}  // namespace

#define MWRT_CAT2(a, b) a##b
#define MWRT_CAT(a, b) MWRT_CAT2(a, b)

hipError_t MWRT_CAT(launch_fused_nfc, MWRT_INST_NFC)(const FusedArgs& a, dim3 grid, dim3 block, size_t lds, hipStream_t st,
                                                      int variant) {
  switch (variant) {
    case FUSED_TB_ONLY: return launch_by_size<false, false>(a, grid, block, lds, st);
    case FUSED_OPT: return launch_by_size<true, false>(a, grid, block, lds, st);
    case FUSED_FROM_ALPHA: return launch_by_size<false, false, true>(a, grid, block, lds, st);
    default: return launch_by_size<true, true>(a, grid, block, lds, st);
  }
}

hipError_t MWRT_CAT(launch_absorb_nfc, MWRT_INST_NFC)(const AbsorbArgs& a, dim3 grid, dim3 block, hipStream_t st) {
  if (block.x <= 256) hipLaunchKernelGGL((k_absorb<NFC, 256>), grid, block, 0, st, a);
  else if (block.x <= 512) hipLaunchKernelGGL((k_absorb<NFC, 512>), grid, block, 0, st, a);
  else hipLaunchKernelGGL((k_absorb<NFC, 1024>), grid, block, 0, st, a);
  return hipGetLastError();
}

#if MWRT_INST_NFC == 16
namespace {
template <int MAXT, bool TAU>
hipError_t launch_win_one(const AbsorbWinArgs& a, dim3 grid, dim3 block, size_t lds, hipStream_t st) {
  auto k = k_absorb_win<MAXT, TAU>;
  hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k, grid, block, lds, st, a);
  return hipGetLastError();
}
template <int NA>
hipError_t launch_rte_one(const RteTauArgs& a, dim3 grid, size_t lds, hipStream_t st) {
  hipLaunchKernelGGL((k_rte_tau<NA>), grid, dim3(RTE_THREADS), lds, st, a);
  return hipGetLastError();
}
}  // namespace

// bytes of LDS a windowed absorption workgroup of `threads` lanes needs (dynamic node sums + static tables)
size_t absorb_win_lds_bytes(int threads) {
  const int maxt = threads <= 256 ? 256 : 512;
  return sizeof(double) * ((size_t)(WIN_NODES + WIN_NODES_H) * threads + (3 * WIN_NFC + 2) * (1 + maxt / WAVE) +
                           (3 * WIN_NODES_H + 2)) + 64;
}

hipError_t launch_absorb_win(const AbsorbWinArgs& a, dim3 grid, dim3 block, hipStream_t st, bool tau) {
  // node sums in LDS: 16 (O2) + 8 (H2O) doubles per thread -- 36 KB at 192 threads: four workgroups per CU
  const size_t lds = sizeof(double) * (WIN_NODES + WIN_NODES_H) * block.x;
  if (block.x <= 256)
    return tau ? launch_win_one<256, true>(a, grid, block, lds, st) : launch_win_one<256, false>(a, grid, block, lds, st);
  return tau ? launch_win_one<512, true>(a, grid, block, lds, st) : launch_win_one<512, false>(a, grid, block, lds, st);
}

// K1 + layer step, every line at every frequency: zenith layer optical depth [nprof][nlev][fpitch]
hipError_t launch_absorb_tau(const AbsorbArgs& a, dim3 grid, dim3 block, hipStream_t st) {
  if (block.x <= 256) hipLaunchKernelGGL((k_absorb<16, 256, true>), grid, block, 0, st, a);
  else if (block.x <= 512) hipLaunchKernelGGL((k_absorb<16, 512, true>), grid, block, 0, st, a);
  else hipLaunchKernelGGL((k_absorb<16, 1024, true>), grid, block, 0, st, a);
  return hipGetLastError();
}

// RTE from layer optical depths; `na` elevations (a.a0 .. a.a0 + na - 1) per launch, na in 1..8 or 10
hipError_t launch_rte_tau(const RteTauArgs& a, dim3 grid, size_t lds, hipStream_t st, int na) {
  switch (na) {
    case 1: return launch_rte_one<1>(a, grid, lds, st);
    case 2: return launch_rte_one<2>(a, grid, lds, st);
    case 3: return launch_rte_one<3>(a, grid, lds, st);
    case 4: return launch_rte_one<4>(a, grid, lds, st);
    case 5: return launch_rte_one<5>(a, grid, lds, st);
    case 6: return launch_rte_one<6>(a, grid, lds, st);
    case 7: return launch_rte_one<7>(a, grid, lds, st);
    case 8: return launch_rte_one<8>(a, grid, lds, st);
    case 10: return launch_rte_one<10>(a, grid, lds, st);
    default: return hipErrorInvalidValue;
  }
}
#endif

}  // namespace mwrt
