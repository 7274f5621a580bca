// mwrt_inst.hip.h -- launch entry points of the per-NFC translation units (csrc/mwrt_inst.hip).
//
// The fused kernel exists in 3 frequency-chunk widths x 3 workgroup sizes x 3 feature sets; compiled in one
// translation unit that is ~2 minutes of hipcc.  Each chunk width is its own translation unit
// (-DMWRT_INST_NFC=8|14|16), built in parallel by build.py and linked into libmwrt.so.
#pragma once
#include "mwrt_kernels.hip.h"

namespace mwrt {

// feature set of a fused-kernel instantiation
enum FusedVariant {
  FUSED_TB_ONLY = 0,   // clear sky, plane-parallel, TB only: the throughput path (bench, the wrapper's batched call)
  FUSED_OPT = 1,       // + cloud liquid / ice and ray-traced paths (mwrt_tb_options), TB only
  FUSED_FULL = 2,      // + the other DataFrame columns and layer optical depths (mwrt_tb_extras)
  FUSED_FROM_ALPHA = 3 // layer integration + RTE from absorption coefficients already in HBM (no K1), TB only
};

#define MWRT_DECLARE_INST(N)                                                                                         \
  hipError_t launch_fused_nfc##N(const FusedArgs& a, dim3 grid, dim3 block, size_t lds, hipStream_t st, int variant); \
  hipError_t launch_absorb_nfc##N(const AbsorbArgs& a, dim3 grid, dim3 block, hipStream_t st);
MWRT_DECLARE_INST(8)
MWRT_DECLARE_INST(14)
MWRT_DECLARE_INST(16)
#undef MWRT_DECLARE_INST
// the windowed fine-grid absorption kernel exists for the 16-wide chunks only (csrc/mwrt_inst.hip, NFC = 16 unit)
hipError_t launch_absorb_win(const AbsorbWinArgs& a, dim3 grid, dim3 block, hipStream_t st, bool tau);
size_t absorb_win_lds_bytes(int threads);
// ... and so do the layer-optical-depth form of the every-line absorption kernel and the RTE kernel that reads it
hipError_t launch_absorb_tau(const AbsorbArgs& a, dim3 grid, dim3 block, hipStream_t st);
hipError_t launch_rte_tau(const RteTauArgs& a, dim3 grid, size_t lds, hipStream_t st, int na);

}  // namespace mwrt
