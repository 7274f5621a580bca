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
hipError_t launch_absorb_win(const AbsorbWinArgs& a, dim3 grid, dim3 block, hipStream_t st) {
  // node sums in LDS: 2 x WIN_NODES doubles per thread (48 KB at 192 threads: three workgroups per CU)
  const size_t lds = sizeof(double) * 2 * WIN_NODES * block.x;
  if (block.x <= 256) {
    hipError_t e = hipFuncSetAttribute((const void*)k_absorb_win<256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_absorb_win<256>), grid, block, lds, st, a);
  } else {
    hipError_t e = hipFuncSetAttribute((const void*)k_absorb_win<512>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_absorb_win<512>), grid, block, lds, st, a);
  }
  return hipGetLastError();
}
#endif

}  // namespace mwrt
