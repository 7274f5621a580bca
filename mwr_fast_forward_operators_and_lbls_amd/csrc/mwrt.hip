// mwrt.hip -- C-ABI host side of libmwrt.so (declarations + reference citations: include/mwrt.h).
//
// HIP only: there is no CPU fallback in this library.  Without a GPU mwrt_device_count()
// returns 0 and mwrt_create() fails with MWRT_ERR_NO_DEVICE.
#define MWRT_HOST_TU 1      // this translation unit owns the non-template kernels
#include "mwrt_inst.hip.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <new>
#include <atomic>

using namespace mwrt;

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) { g_err = msg; return code; }

#define HIP_TRY(expr)                                                                        \
  do {                                                                                       \
    hipError_t e_ = (expr);                                                                  \
    if (e_ != hipSuccess)                                                                    \
      return fail(e_ == hipErrorOutOfMemory ? MWRT_ERR_OUT_OF_MEMORY : MWRT_ERR_HIP,         \
                  std::string(#expr) + ": " + hipGetErrorString(e_));                        \
  } while (0)

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  hipError_t reserve(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    if (p) { hipError_t e = hipFree(p); p = nullptr; cap = 0; if (e != hipSuccess) return e; }
    size_t want = bytes + bytes / 8 + 256;
    hipError_t e = hipMalloc(&p, want);
    if (e == hipSuccess) cap = want;
    return e;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
  template <class T> T* as() { return static_cast<T*>(p); }
};

// Content-keyed, immutable device copies of small host arrays.
struct ParamCache {
  struct Entry { std::vector<double> host; double* dev = nullptr; unsigned long stamp = 0; };
  std::vector<Entry> entries;
  unsigned long clock = 0;
  static constexpr size_t MAX_ENTRIES = 32;
  // device pointer holding exactly src[0..n); uploads through `copy_stream` (never a capturing stream)
  hipError_t get(const double* src, int n, hipStream_t copy_stream, const double** out) {
    for (Entry& e : entries)
      if ((int)e.host.size() == n && std::memcmp(e.host.data(), src, sizeof(double) * (size_t)n) == 0) {
        e.stamp = ++clock; *out = e.dev; return hipSuccess;
      }
    if (entries.size() >= MAX_ENTRIES) {
      // evict the least recently used copy -- only after everything queued on the device has drained
      hipError_t e = hipDeviceSynchronize();
      if (e != hipSuccess) return e;
      size_t lru = 0;
      for (size_t i = 1; i < entries.size(); ++i) if (entries[i].stamp < entries[lru].stamp) lru = i;
      (void)hipFree(entries[lru].dev);
      entries.erase(entries.begin() + (long)lru);
    }
    Entry ne;
    ne.host.assign(src, src + n);
    hipError_t e = hipMalloc((void**)&ne.dev, sizeof(double) * (size_t)(n > 0 ? n : 1));
    if (e != hipSuccess) return e;
    // a fresh buffer nobody reads yet: copy on the context's own stream and wait for it, so the
    // bytes are in HBM before any stream (the caller's included) can launch a reader
    e = hipMemcpyAsync(ne.dev, ne.host.data(), sizeof(double) * (size_t)n, hipMemcpyHostToDevice, copy_stream);
    if (e == hipSuccess) e = hipStreamSynchronize(copy_stream);
    if (e != hipSuccess) { (void)hipFree(ne.dev); return e; }
    ne.stamp = ++clock;
    entries.push_back(std::move(ne));
    *out = entries.back().dev;
    return hipSuccess;
  }
  void release() { for (Entry& e : entries) (void)hipFree(e.dev); entries.clear(); }
};

}  // namespace

struct mwrt_context {
  int device = 0;
  hipStream_t stream = nullptr;
  int lds_max = 65536;
  int num_cus = 256;            // compute units of the device (MI355X: 256)
  // small per-call parameter arrays (frq, airmass): content-keyed device copies.  A copy is never
  // overwritten or freed while the context lives (bar LRU eviction behind a device-wide drain), so
  // launches still queued on ANY stream and captured hipGraphs keep reading valid memory.
  ParamCache frq_cache, am_cache, elev_cache;
  // fine-grid absorption: window descriptors + Lagrange matrices per (model, frequency list), immutable like ParamCache
  int absorption_mode = 0;      // 0 auto, 1 direct, 2 windowed
  int chunk_width = 0;          // 0 auto, 8 / 14 / 16: frequencies per workgroup of the fused TB kernel (mwrt_set_chunk_width)
  struct WinEntry { uint64_t model_id; std::vector<double> frq; char* d_blob; size_t off_lag, off_lagh, off_lagsd; int nwin; };
  std::vector<WinEntry> win_cache;
  // line classification of every frequency chunk (LineMasks), per (model, frequency list, chunk width): depends on the
  // frequencies and the table only, so the host computes it once instead of every workgroup voting on it
  struct MaskEntry { uint64_t model_id; int nfc; std::vector<double> frq; LineMasks* d_masks; };
  std::vector<MaskEntry> mask_cache;
  // ray-tracing workspace: path factors [nprof][nang][nlev] and the per-profile ducting flag
  DevBuf d_amf, d_duct;
  // fine-grid two-kernel path: materialised absorption of one profile batch (awet | adry)
  DevBuf d_alpha;
  size_t alpha_batch_bytes = (size_t)4 << 30;   // 4 GiB of a 288-GB card: configs[4]'s per-GPU share is one batch
  // the two workspaces above are shared by consecutive calls: a call that uses one on a different stream than
  // the previous user first waits (on the device) for that user's last kernel
  hipEvent_t ws_event = nullptr;
  hipStream_t ws_stream = nullptr;
  bool ws_used = false;
  // staging for the host-buffer entry points
  DevBuf d_in, d_out, d_valid, d_ex;
  // timing: a ring of hipEvent pairs recorded around every kernel launch, on the launch stream
  bool timing = false;
  std::vector<hipEvent_t> ev0, ev1;
  long ev_count = 0;
};
constexpr int TIMING_RING = 512;

struct mwrt_model {
  ModelFlat* d_desc = nullptr;
  ModelFlat h_desc;
  uint64_t id = 0;              // process-unique: caches keyed by it survive a destroy / create that reuses the address
};

namespace {

// K2 work split: items = pairs x nseg over `threads` lanes; cost ~ rounds x seglen (+ combine)
// segments per (frequency, angle) pair for one K2 pass of `npairs` pairs: fill the workgroup in one round
int plan_k2_pass(int nlev, int npairs, int threads) {
  const int layers = nlev - 1;
  int best = 1; long best_cost = -1;
  for (int ns = 1; ns <= 64 && ns <= (layers > 0 ? layers : 1); ++ns) {
    const int sl = (layers + ns - 1) / ns;
    const long rounds = ((long)npairs * ns + threads - 1) / threads;
    const long cost = rounds * (sl * 8L + 4) + ns;      // 8 ~ relative cost of a layer step vs a combine step
    if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = ns; }
  }
  return best;
}

// rows kept in LDS per K2 pass; a 14-wide chunk runs as passes of 8 and 6 rows, each with its own split
int nfk_of(int /*nfc*/) { return 8; }

unsigned magic_of(int d) { return d <= 1 ? 0u : (unsigned)((((uint64_t)1 << 32) + (uint64_t)d - 1) / (uint64_t)d); }
void set_magics(LaunchGeom* g, int nang) {
  g->magic_nseg[0] = magic_of(g->nseg[0]);
  g->magic_nseg[1] = magic_of(g->nseg[1]);
  g->magic_nang = magic_of(nang);
}

void set_pass(LaunchGeom* g, int h, int nlev, int ns) {
  g->nseg[h] = ns < 1 ? 1 : ns;
  g->seglen[h] = (nlev - 1 + g->nseg[h] - 1) / g->nseg[h];
  if (g->seglen[h] < 1) g->seglen[h] = 1;
}

LaunchGeom plan_k2(int nlev, int nfc, int nf, int nang, int threads) {
  LaunchGeom g;
  const int nfk = nfk_of(nfc);
  const int rows0 = std::min(nfk, std::min(nfc, nf));
  const int rows1 = std::max(0, std::min(nfc, nf) - nfk);
  set_pass(&g, 0, nlev, plan_k2_pass(nlev, rows0 * nang, threads));
  set_pass(&g, 1, nlev, rows1 > 0 ? plan_k2_pass(nlev, rows1 * nang, threads) : 1);
  g.npart = 2 * nang * std::max(rows0 * g.nseg[0], rows1 * g.nseg[1]);
  set_magics(&g, nang);
  // row stride in doubles: odd multiple of 2 dwords keeps ds_read_b64 rows on distinct banks
  int ld = nlev + 1;
  if ((ld & 1) == 0) ld += 1;
  g.ldrow = ld;
  return g;
}

size_t fused_lds_bytes(int nfc, const LaunchGeom& g, int /*nang*/, int threads) {
  const int nfk = nfk_of(nfc);
  // float gmax[nfk][threads/16], int wcnt[nwaves], int perm[threads]
  const size_t sort_doubles = ((size_t)nfk * (threads / 16) + (threads / WAVE) + threads + 1) / 2;
  return sizeof(double) * ((size_t)2 * nfk * g.ldrow + (size_t)g.npart + 16 +
                           (size_t)(threads / WAVE) * 2 * nfc + sort_doubles);
}

// K2 split for a chunk width, shrunk until the workgroup's LDS fits; false if it cannot
bool plan_fused(const mwrt_context* c, int nfc, int nlev, int nf, int nang, LaunchGeom* g, size_t* lds, int threads = 0);

int upload_small(mwrt_context* c, ParamCache& cache, const double* src, int n, const double** dev) {
  HIP_TRY(cache.get(src, n, c->stream, dev));
  return MWRT_OK;
}

// Workspace hand-over between streams (ray-path factors, materialised absorption): stream-ordered, no host wait.
int workspace_acquire(mwrt_context* c, hipStream_t st) {
  if (!c->ws_event) HIP_TRY(hipEventCreateWithFlags(&c->ws_event, hipEventDisableTiming));
  if (c->ws_used && c->ws_stream != st) HIP_TRY(hipStreamWaitEvent(st, c->ws_event, 0));
  return MWRT_OK;
}
int workspace_release(mwrt_context* c, hipStream_t st) {
  HIP_TRY(hipEventRecord(c->ws_event, st));
  c->ws_stream = st; c->ws_used = true;
  return MWRT_OK;
}

// `stream` argument of the *_device entry points: NULL = the context's own (non-blocking) stream,
// MWRT_STREAM_LEGACY = the caller's legacy default stream (hipStream_t 0), else the handle itself
hipStream_t resolve_stream(const mwrt_context* c, void* stream) {
  if (!stream) return c->stream;
  if (stream == MWRT_STREAM_LEGACY) return (hipStream_t) nullptr;
  return (hipStream_t)stream;
}

int check_common(const mwrt_context* c, const mwrt_model* m, int64_t nprof, int32_t nlev, int32_t nf) {
  if (!c || !m) return fail(MWRT_ERR_INVALID_ARGUMENT, "null context or model");
  if (nprof < 0 || nf < 1) return fail(MWRT_ERR_INVALID_ARGUMENT, "nprof < 0 or nf < 1");
  if (nlev < 2) return fail(MWRT_ERR_INVALID_ARGUMENT, "nlev < 2");
  if (nlev > MWRT_MAX_LEVELS) return fail(MWRT_ERR_UNSUPPORTED, "nlev > MWRT_MAX_LEVELS (one lane per level)");
  if (nprof > 2147483647LL) return fail(MWRT_ERR_UNSUPPORTED, "nprof exceeds grid limit");
  return MWRT_OK;
}

bool any_nan(const double* x, int n) {
  for (int i = 0; i < n; ++i) if (std::isnan(x[i])) return true;
  return false;
}

void timing_begin(mwrt_context* c, hipStream_t st) {
  if (c->timing) (void)hipEventRecord(c->ev0[c->ev_count % TIMING_RING], st);
}
void timing_end(mwrt_context* c, hipStream_t st) {
  if (c->timing) { (void)hipEventRecord(c->ev1[c->ev_count % TIMING_RING], st); c->ev_count++; }
}

int launch_fused(mwrt_context* c, int nfc, FusedArgs a, int64_t nprof, hipStream_t st, int variant) {
  int threads = ((a.nlev + WAVE - 1) / WAVE) * WAVE;
  // The RTE-from-absorption kernel is light on registers (4 waves per SIMD fit): a fourth wave that holds no
  // level still takes its share of the (frequency, angle, segment) items, 90 serial layer steps instead of 120
  if (variant == FUSED_FROM_ALPHA && threads < 256 && a.nang > 1) threads = 256;
  const int nchunks = (a.nf + nfc - 1) / nfc;
  size_t lds = 0;
  if (!plan_fused(c, nfc, a.nlev, a.nf, a.nang, &a.g, &lds, threads))
    return fail(MWRT_ERR_UNSUPPORTED, "LDS budget exceeded (nlev x nang too large)");
  dim3 grid((unsigned)nprof /* = nmodels x profiles */, (unsigned)nchunks), block(threads);
  // valid[] = 1 is written by the kernel itself when one workgroup owns the profile; with several
  // frequency chunks per profile the flags are preset here and the kernel only lowers/raises them
  a.write_valid = nchunks == 1;
  if (!a.write_valid) HIP_TRY(hipMemsetAsync(a.valid, 1, (size_t)nprof, st));
#if MWRT_PHASE_CLOCK
  // diagnostic build: stamps of the LAST launch go to $MWRT_PHASE_DUMP as raw int64 [nprof][4][10] (100-MHz wall clock; slots 8, 9 = HW_ID, XCC_ID)
  static long long* d_phase = nullptr; static size_t phase_cap = 0;
  const char* dump = std::getenv("MWRT_PHASE_DUMP");
  const size_t phase_n = (size_t)nprof * 4 * 10;
  a.phase = nullptr;
  if (dump && nchunks == 1) {
    if (phase_cap < phase_n) { if (d_phase) (void)hipFree(d_phase); HIP_TRY(hipMalloc((void**)&d_phase, phase_n * 8)); phase_cap = phase_n; }
    HIP_TRY(hipMemsetAsync(d_phase, 0, phase_n * 8, st));
    a.phase = d_phase;
  }
#endif
  timing_begin(c, st);
  hipError_t e;
  switch (nfc) {                                    // one translation unit per chunk width (csrc/mwrt_inst.hip)
    case 8: e = launch_fused_nfc8(a, grid, block, lds, st, variant); break;
    case 14: e = launch_fused_nfc14(a, grid, block, lds, st, variant); break;
    default: e = launch_fused_nfc16(a, grid, block, lds, st, variant); break;
  }
  timing_end(c, st);
  HIP_TRY(e);
#if MWRT_PHASE_CLOCK
  if (a.phase) {
    std::vector<long long> h(phase_n);
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipMemcpy(h.data(), d_phase, phase_n * 8, hipMemcpyDeviceToHost));
    if (FILE* f = std::fopen(dump, "wb")) { std::fwrite(h.data(), 8, phase_n, f); std::fclose(f); }
  }
#endif
  return MWRT_OK;
}

int launch_absorb(mwrt_context* c, int nfc, const AbsorbArgs& a, int64_t nprof, hipStream_t st) {
  const int threads = ((a.nlev + WAVE - 1) / WAVE) * WAVE;
  const int nchunks = (a.nf + nfc - 1) / nfc;
  dim3 grid((unsigned)nprof, (unsigned)nchunks), block(threads);
  timing_begin(c, st);
  hipError_t e;
  switch (nfc) {
    case 8: e = launch_absorb_nfc8(a, grid, block, st); break;
    case 14: e = launch_absorb_nfc14(a, grid, block, st); break;
    default: e = launch_absorb_nfc16(a, grid, block, st); break;
  }
  timing_end(c, st);
  HIP_TRY(e);
  return MWRT_OK;
}

bool plan_fused(const mwrt_context* c, int nfc, int nlev, int nf, int nang, LaunchGeom* g, size_t* lds, int threads) {
  if (threads <= 0) threads = ((nlev + WAVE - 1) / WAVE) * WAVE;
  *g = plan_k2(nlev, nfc, nf, nang, threads);
  *lds = fused_lds_bytes(nfc, *g, nang, threads);
  while (*lds > (size_t)c->lds_max && (g->nseg[0] > 1 || g->nseg[1] > 1)) {     // shrink the partials if LDS is short
    const int nfk = nfk_of(nfc);
    const int rows0 = std::min(nfk, std::min(nfc, nf)), rows1 = std::max(0, std::min(nfc, nf) - nfk);
    set_pass(g, 0, nlev, (g->nseg[0] + 1) / 2);
    set_pass(g, 1, nlev, (g->nseg[1] + 1) / 2);
    g->npart = 2 * nang * std::max(rows0 * g->nseg[0], rows1 * g->nseg[1]);
    set_magics(g, nang);
    *lds = fused_lds_bytes(nfc, *g, nang, threads);
  }
  return *lds <= (size_t)c->lds_max;
}

// frequency-chunk width: 14 HATPRO channels fit one chunk exactly; other counts use 16 / 8
int pick_nfc(int nf) {
  if (nf % 14 == 0 || nf <= 14) return (nf <= 8) ? 8 : 14;
  return 16;
}

// ... unless the caller fixed the width (mwrt_set_chunk_width: 8 splits a 14-channel profile over two workgroups, each with
// the full per-(level, line) set-up but half the line-frequency work -- one profile 58 instead of 75 us, 256 profiles 61
// instead of 76, 512 profiles 77 instead of 83 at seven elevations; not the default because results would then depend, in the
// 13th digit, on how a caller batches its profiles) or the profile is so tall that the wide chunk's LDS rows do not fit: then 8
int pick_nfc_fused(const mwrt_context* c, int nlev, int nf, int nang) {
  int nfc = c->chunk_width ? c->chunk_width : pick_nfc(nf);
  LaunchGeom g; size_t lds;
  if (!plan_fused(c, nfc, nlev, nf, nang, &g, &lds)) nfc = 8;
  return nfc;
}

// ---- fine-grid absorption: windows of WIN_CHUNKS chunks, Chebyshev nodes, Lagrange matrices ----
constexpr double WIN_MARGIN_GHZ = 4.0;       // an O2 line is window-far when its centre is this far beyond the window (16 nodes)
constexpr double WIN_H2O_MARGIN_GHZ = 30.0;  // an H2O line: this far (8 nodes; the H2O table is sparse, few lines come closer)
constexpr double WIN_MAX_SPAN_GHZ = 6.0;     // widest window the 16-node interpolation is used on
constexpr double WIN_CUTOFF_GUARD_GHZ = 5.0; // the H2O 750-GHz cutoff must be this clearly in or out (pressure shifts < 1 GHz)

bool windows_eligible(const double* frq, int nf) {
  if (nf < WIN_CHUNKS * WIN_NFC) return false;
  for (int j = 1; j < nf; ++j) if (!(frq[j] > frq[j - 1])) return false;
  const int per = WIN_CHUNKS * WIN_NFC;
  for (int b = 0; b < nf; b += per) {
    const int e = std::min(nf, b + per) - 1;
    if (e == b) return false;                              // a one-frequency window has no span to put nodes on
    if (frq[e] - frq[b] > WIN_MAX_SPAN_GHZ) return false;
  }
  return true;
}

// upper bound of a speed-dependent H2O line's half width anywhere in an atmosphere (dry air <= 1100 hPa, vapour
// <= 150 hPa, T >= 148 K): the host may put such a line in a window's far set only where 10 half-widths cannot
// reach the window; the kernel re-checks per level and takes the line back if they can
double sd_halfwidth_bound(const mwrt_model_desc& t, int k) {
  return t.h2o_w0[k] * 1100.0 * std::pow(2.0, std::max(t.h2o_x[k], 0.0)) +
         t.h2o_w0s[k] * 150.0 * std::pow(2.0, std::max(t.h2o_xs[k], 0.0));
}

// line_masks of every chunk of `nfc` frequencies (what the kernels' line loops are steered by; mwrt_kernels.hip.h LineMasks)
void chunk_masks(const mwrt_model_desc& t, const double* frq, int nf, int nfc, std::vector<LineMasks>* out) {
  const int nchunks = (nf + nfc - 1) / nfc;
  out->assign(nchunks, LineMasks{});
  for (int ch = 0; ch < nchunks; ++ch) {
    const int j0 = ch * nfc, j1 = std::min(nf, j0 + nfc);
    LineMasks& lm = (*out)[ch];
    // very far lines (vfar_add): poles of the line's term in u = f^2, u ~ c^2 -+ 2 i c w, at >= 1/VF_RATIO_MAX half ranges
    // from the middle of the chunk's f^2 values -- with 2 GHz of allowance for pressure shifts and 10 GHz for the half width
    double ulo = 1e300, uhi = 0.0;
    for (int j = j0; j < j1; ++j) { ulo = std::min(ulo, frq[j] * frq[j]); uhi = std::max(uhi, frq[j] * frq[j]); }
    lm.vf_u0 = 0.5 * (ulo + uhi);
    lm.vf_h = std::max(0.5 * (uhi - ulo), 1.0);
    lm.vf_invh = 1.0 / lm.vf_h;
    auto very_far = [&](double c) {
      const double cl = std::max(c - 2.0, 0.0), ch = c + 2.0;
      const double plo = cl * cl - 100.0, phi = ch * ch;             // real part of the poles lies in [plo, phi]
      const double dist = (lm.vf_u0 < plo) ? plo - lm.vf_u0 : ((lm.vf_u0 > phi) ? lm.vf_u0 - phi : 0.0);
      return lm.vf_h <= VF_RATIO_MAX * dist;
    };
    // (a line costs ~40 instructions in the polynomial against 7 per frequency directly: not worth it under 7 frequencies)
    static const bool no_vfar_env = std::getenv("MWRT_NO_VFAR") != nullptr;               // diagnostic: time the direct sums
    const bool no_vfar = no_vfar_env || (j1 - j0) < VF_MIN_FREQS;
    for (int k = 0; k < t.n_o2; ++k) {
      double dmin = 1e300;
      for (int j = j0; j < j1; ++j) dmin = std::min(dmin, std::fabs(frq[j] - t.o2_f[k]));
      if (dmin >= FAR_MIN_GHZ + FAR_SHIFT_GHZ) lm.o2_far |= 1ull << k;
      if (!no_vfar && very_far(t.o2_f[k])) lm.o2_vfar |= 1ull << k;
    }
    if (__builtin_popcountll(lm.o2_vfar) < VF_MIN_LINES) lm.o2_vfar = 0;
    for (int k = 0; k < t.n_h2o; ++k) {
      double dmin = 1e300, smin = 1e300;
      for (int j = j0; j < j1; ++j) { dmin = std::min(dmin, std::fabs(frq[j] - t.h2o_fl[k])); smin = std::min(smin, std::fabs(frq[j] + t.h2o_fl[k])); }
      if (dmin >= FAR_H2O_GHZ) lm.h2o_far |= 1u << k;
      if (!no_vfar && very_far(t.h2o_fl[k])) lm.h2o_vfar |= 1u << k;
      if (dmin >= 750.0 + FAR_H2O_GHZ && smin >= 750.0 + FAR_H2O_GHZ) lm.h2o_none |= 1u << k;
      if (smin >= 750.0 + FAR_H2O_GHZ) lm.h2o_res |= 1u << k;
      if (t.h2o_w2[k] > 0.0) {
        lm.h2o_sd |= 1u << k;
        if (10.0 * sd_halfwidth_bound(t, k) < dmin - 1.0) lm.h2o_sdfar |= 1u << k;       // its special shape cannot reach the chunk
        // half-sampled shape: a full 16-frequency chunk of increasing frequencies, >= 3 GHz and 5 spans from the centre
        bool inc = nfc == 16 && j1 - j0 == 16;
        for (int j = j0 + 1; inc && j < j1; ++j) inc = frq[j] > frq[j - 1];
        static const bool no_half = std::getenv("MWRT_NO_SD_HALF") != nullptr;            // diagnostic: time the full sampling
        if (inc && !no_half && dmin >= 3.0 && dmin >= 5.0 * (frq[j1 - 1] - frq[j0])) lm.h2o_sdint |= 1u << k;
      }
    }
    if (__builtin_popcount(lm.h2o_vfar) < VF_MIN_LINES) lm.h2o_vfar = 0;
  }
}

// device copy of chunk_masks(...), immutable and cached like the window descriptors
int get_masks(mwrt_context* c, const mwrt_model* m, const double* frq, int nf, int nfc, const LineMasks** out) {
  for (size_t i = 0; i < c->mask_cache.size(); ++i) {
    auto& e = c->mask_cache[i];
    if (e.model_id == m->id && e.nfc == nfc && (int)e.frq.size() == nf && std::memcmp(e.frq.data(), frq, sizeof(double) * nf) == 0) {
      *out = e.d_masks;
      // least recently USED goes first: a hit moves to the back, so a warm-up call keeps what it touched (a later miss of the
      // same sequence must not evict it -- the eviction drains the device, which a capturing stream refuses)
      std::rotate(c->mask_cache.begin() + (long)i, c->mask_cache.begin() + (long)i + 1, c->mask_cache.end());
      return MWRT_OK;
    }
  }
  if (c->mask_cache.size() >= 64) {                     // bounded: drop the least recently used entry behind a device-wide drain
    HIP_TRY(hipDeviceSynchronize());
    (void)hipFree(c->mask_cache.front().d_masks);
    c->mask_cache.erase(c->mask_cache.begin());
  }
  std::vector<LineMasks> host;
  chunk_masks(m->h_desc, frq, nf, nfc, &host);
  mwrt_context::MaskEntry e{m->id, nfc, std::vector<double>(frq, frq + nf), nullptr};
  HIP_TRY(hipMalloc((void**)&e.d_masks, sizeof(LineMasks) * host.size()));
  hipError_t err = hipMemcpyAsync(e.d_masks, host.data(), sizeof(LineMasks) * host.size(), hipMemcpyHostToDevice, c->stream);
  if (err == hipSuccess) err = hipStreamSynchronize(c->stream);
  if (err != hipSuccess) { (void)hipFree(e.d_masks); HIP_TRY(err); }
  c->mask_cache.push_back(std::move(e));
  *out = c->mask_cache.back().d_masks;
  return MWRT_OK;
}

// Chebyshev nodes of [flo, fhi] and the barycentric Lagrange weights of every target frequency of the window,
// stored [chunk][node][target]; targets past the last frequency repeat it (their results are discarded)
template <int NNODES>
void window_nodes(const double* frq, int b, int e, int nchunks, double* fnode, double* blk_base) {
  const int per = nchunks * WIN_NFC;
  const double flo = frq[b], fhi = frq[e];
  long double x[NNODES], bw[NNODES];
  for (int m = 0; m < NNODES; ++m)
    fnode[m] = (double)(0.5L * (flo + fhi) + 0.5L * (fhi - flo) * cosl(M_PIl * (2 * m + 1) / (2.0L * NNODES)));
  for (int m = 0; m < NNODES; ++m) x[m] = fnode[m];                // weights for the nodes as the kernel sees them
  for (int m = 0; m < NNODES; ++m) {
    long double prod = 1.0L;
    for (int k = 0; k < NNODES; ++k) if (k != m) prod *= (x[m] - x[k]);
    bw[m] = 1.0L / prod;
  }
  for (int r = 0; r < per; ++r) {
    const long double f = frq[std::min(b + r, e)];
    const int cidx = r / WIN_NFC, j = r % WIN_NFC;
    double* blk = blk_base + (size_t)cidx * NNODES * WIN_NFC;
    int hit = -1;
    for (int m = 0; m < NNODES; ++m) if (f == x[m]) hit = m;
    if (hit >= 0) { blk[hit * WIN_NFC + j] = 1.0; continue; }
    long double q[NNODES], sum = 0.0L;
    for (int m = 0; m < NNODES; ++m) { q[m] = bw[m] / (f - x[m]); sum += q[m]; }
    for (int m = 0; m < NNODES; ++m) blk[m * WIN_NFC + j] = (double)(q[m] / sum);
  }
}

struct WindowSet {
  std::vector<WinDesc> wins;
  std::vector<double> lag, lag_h;       // [nwin][WIN_CHUNKS_MAX][nodes][WIN_NFC]
  std::vector<double> lag_sd;           // [nchunks][SD_TARGETS][SD_NODES]: odd slots of a chunk from slots 0, 2, ..., 14, 15
};

// far-line sets of a window [flo, fhi]: an O2 line is far beyond max(4 GHz, 1.6 half-spans), an H2O line beyond
// max(30 GHz, 11.8 half-spans) -- the distance-to-half-span ratios the 16- and 8-node interpolations were sized for
void window_far_sets(const mwrt_model_desc& t, double flo, double fhi, WinDesc* d) {
  const double half = 0.5 * (fhi - flo);
  const double mo = std::max(WIN_MARGIN_GHZ, 1.6 * half), mh = std::max(WIN_H2O_MARGIN_GHZ, 11.8 * half);
  d->o2_far = 0; d->h2o_far_both = 0; d->h2o_far_res = 0;
  for (int k = 0; k < t.n_o2; ++k) {
    const double c = t.o2_f[k];
    if (c < flo - mo || c > fhi + mo) d->o2_far |= 1ull << k;
  }
  for (int k = 0; k < t.n_h2o; ++k) {
    const double c = t.h2o_fl[k];
    if (!(c < flo - mh || c > fhi + mh)) continue;
    // a speed-dependent line stays direct wherever its special shape (inside 10 half-widths) could reach the window
    if (t.h2o_w2[k] > 0.0 && !(10.0 * sd_halfwidth_bound(t, k) < std::min(std::fabs(c - flo), std::fabs(c - fhi)) - 1.0)) continue;
    const double g = WIN_CUTOFF_GUARD_GHZ;
    const bool d1_in = std::fabs(flo - c) < 750.0 - g && std::fabs(fhi - c) < 750.0 - g;
    const bool d2_in = fhi + c < 750.0 - g;
    const bool d2_out = flo + c >= 750.0 + g;
    if (d1_in && d2_in) d->h2o_far_both |= 1u << k;
    else if (d1_in && d2_out) d->h2o_far_res |= 1u << k;
    // anything else (a cutoff crossing the window, or both terms out) is left to the per-chunk loops
  }
}

void build_windows(const mwrt_model_desc& t, const double* frq, int nf, WindowSet* ws) {
  const int per = WIN_CHUNKS * WIN_NFC;
  const int nchunks = (nf + WIN_NFC - 1) / WIN_NFC;
  const int nbase = (nf + per - 1) / per;
  // base windows of WIN_CHUNKS chunks; two neighbours are MERGED (one node phase for both) when the merged window
  // keeps every O2 line far and loses no H2O line from the far set: the out-of-band stretches of a spectrum
  struct Span { int c0, nch; };
  std::vector<Span> spans;
  auto bounds = [&](const Span& sp, int* b, int* e) { *b = sp.c0 * WIN_NFC; *e = std::min(nf, (sp.c0 + sp.nch) * WIN_NFC) - 1; };
  for (int w = 0; w < nbase; ++w) spans.push_back({w * WIN_CHUNKS, std::min(WIN_CHUNKS, nchunks - w * WIN_CHUNKS)});
  const bool merge = std::getenv("MWRT_WIN_NOMERGE") == nullptr;                       // diagnostic: time the unmerged windows
  for (size_t i = 0; i + 1 < spans.size();) {
    const Span m{spans[i].c0, spans[i].nch + spans[i + 1].nch};
    bool ok = merge && spans[i].nch == WIN_CHUNKS && m.nch <= WIN_CHUNKS_MAX;
    if (ok) {
      int b, e; bounds(m, &b, &e);
      WinDesc dm{}, d0{}, d1{};
      window_far_sets(t, frq[b], frq[e], &dm);
      int b0, e0, b1, e1; bounds(spans[i], &b0, &e0); bounds(spans[i + 1], &b1, &e1);
      window_far_sets(t, frq[b0], frq[e0], &d0);
      window_far_sets(t, frq[b1], frq[e1], &d1);
      const unsigned long long all_o2 = t.n_o2 >= 64 ? ~0ull : ((1ull << t.n_o2) - 1ull);
      ok = dm.o2_far == all_o2 &&                                                     // no O2 line anywhere near
           (dm.h2o_far_both | dm.h2o_far_res) == ((d0.h2o_far_both | d0.h2o_far_res) & (d1.h2o_far_both | d1.h2o_far_res));
    }
    if (ok) { spans[i] = m; spans.erase(spans.begin() + (long)i + 1); ++i; }           // (a merged window is not merged again)
    else ++i;
  }
  // Workgroups are dispatched in grid order (profiles fastest, then windows): the EXPENSIVE windows go first, so that the
  // last, partly filled round of the launch is made of cheap ones (the oxygen band sits at the end of a 20-60 GHz grid).
  // Cost per frequency, roughly, in instructions: a floor, 12 per oxygen line evaluated directly, the speed-dependent shape
  // where some level can be inside its 10 half-widths.
  if (std::getenv("MWRT_WIN_GRID_ORDER") == nullptr) {                                   // (diagnostic: keep the grid order)
    auto cost = [&](const Span& sp) {
      int b, e; bounds(sp, &b, &e);
      WinDesc d{};
      window_far_sets(t, frq[b], frq[e], &d);
      const unsigned long long all_o2 = t.n_o2 >= 64 ? ~0ull : ((1ull << t.n_o2) - 1ull);
      double per_f = 150.0 + 12.0 * __builtin_popcountll(~d.o2_far & all_o2);
      for (int k = 0; k < t.n_h2o; ++k) {
        if (!(t.h2o_w2[k] > 0.0)) continue;
        const double c = t.h2o_fl[k];
        const double dist = (c < frq[b]) ? frq[b] - c : ((c > frq[e]) ? c - frq[e] : 0.0);
        per_f += 50.0 * std::max(0.0, 1.0 - dist / (10.0 * sd_halfwidth_bound(t, k)));
      }
      return per_f * (e - b + 1);
    };
    std::stable_sort(spans.begin(), spans.end(), [&](const Span& x, const Span& y) { return cost(x) > cost(y); });
  }
  // Lagrange weights of the half-sampled speed-dependent shape, per chunk (zero for a partial last chunk: never used)
  ws->lag_sd.assign((size_t)nchunks * SD_TARGETS * SD_NODES, 0.0);
  for (int ch = 0; ch < nchunks; ++ch) {
    if ((ch + 1) * WIN_NFC > nf) continue;
    const double* f = frq + (size_t)ch * WIN_NFC;
    for (int i = 0; i < SD_TARGETS; ++i) {
      const long double x = f[2 * i + 1];
      for (int n = 0; n < SD_NODES; ++n) {
        long double w = 1.0L;
        for (int q = 0; q < SD_NODES; ++q)
          if (q != n) w *= (x - (long double)f[sd_node_slot(q)]) / ((long double)f[sd_node_slot(n)] - (long double)f[sd_node_slot(q)]);
        ws->lag_sd[((size_t)ch * SD_TARGETS + i) * SD_NODES + n] = (double)w;
      }
    }
  }
  const int nwin = (int)spans.size();
  const int perm = WIN_CHUNKS_MAX * WIN_NFC;
  ws->wins.assign(nwin, WinDesc{});
  ws->lag.assign((size_t)nwin * perm * WIN_NODES, 0.0);
  ws->lag_h.assign((size_t)nwin * perm * WIN_NODES_H, 0.0);
  for (int w = 0; w < nwin; ++w) {
    WinDesc& d = ws->wins[w];
    int b, e; bounds(spans[w], &b, &e);
    d.flo = frq[b]; d.fhi = frq[e];
    d.first_chunk = spans[w].c0;
    d.nchunks = spans[w].nch;
    window_nodes<WIN_NODES>(frq, b, e, d.nchunks, d.fnode, ws->lag.data() + (size_t)w * perm * WIN_NODES);
    window_nodes<WIN_NODES_H>(frq, b, e, d.nchunks, d.fnode_h, ws->lag_h.data() + (size_t)w * perm * WIN_NODES_H);
    window_far_sets(t, d.flo, d.fhi, &d);
  }
}

struct WinPtrs { const WinDesc* win; const double* lag; const double* lag_h; const double* lag_sd; const LineMasks* masks; int nwin; };   // masks: get_masks(.., WIN_NFC)

int get_windows(mwrt_context* c, const mwrt_model* m, const double* frq, int nf, WinPtrs* out) {
  for (size_t i = 0; i < c->win_cache.size(); ++i) {
    auto& e = c->win_cache[i];
    if (e.model_id == m->id && (int)e.frq.size() == nf && std::memcmp(e.frq.data(), frq, sizeof(double) * nf) == 0) {
      *out = WinPtrs{(const WinDesc*)e.d_blob, (const double*)(e.d_blob + e.off_lag), (const double*)(e.d_blob + e.off_lagh),
                     (const double*)(e.d_blob + e.off_lagsd), nullptr, e.nwin};
      std::rotate(c->win_cache.begin() + (long)i, c->win_cache.begin() + (long)i + 1, c->win_cache.end());   // least recently used first
      return get_masks(c, m, frq, nf, WIN_NFC, &out->masks);
    }
  }
  if (c->win_cache.size() >= 16) {                      // bounded: drop the least recently used entry behind a device-wide drain
    HIP_TRY(hipDeviceSynchronize());
    (void)hipFree(c->win_cache.front().d_blob);
    c->win_cache.erase(c->win_cache.begin());
  }
  WindowSet ws;
  build_windows(m->h_desc, frq, nf, &ws);
  auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
  mwrt_context::WinEntry e{m->id, std::vector<double>(frq, frq + nf), nullptr, 0, 0, 0, (int)ws.wins.size()};
  e.off_lag = up(sizeof(WinDesc) * ws.wins.size());
  e.off_lagh = e.off_lag + up(sizeof(double) * ws.lag.size());
  e.off_lagsd = e.off_lagh + up(sizeof(double) * ws.lag_h.size());
  const size_t total = e.off_lagsd + up(sizeof(double) * ws.lag_sd.size());
  std::vector<char> host(total, 0);
  std::memcpy(host.data(), ws.wins.data(), sizeof(WinDesc) * ws.wins.size());
  std::memcpy(host.data() + e.off_lag, ws.lag.data(), sizeof(double) * ws.lag.size());
  std::memcpy(host.data() + e.off_lagh, ws.lag_h.data(), sizeof(double) * ws.lag_h.size());
  std::memcpy(host.data() + e.off_lagsd, ws.lag_sd.data(), sizeof(double) * ws.lag_sd.size());
  HIP_TRY(hipMalloc((void**)&e.d_blob, total));
  hipError_t err = hipMemcpyAsync(e.d_blob, host.data(), total, hipMemcpyHostToDevice, c->stream);
  if (err == hipSuccess) err = hipStreamSynchronize(c->stream);
  if (err != hipSuccess) { (void)hipFree(e.d_blob); HIP_TRY(err); }
  c->win_cache.push_back(std::move(e));
  const auto& k = c->win_cache.back();
  *out = WinPtrs{(const WinDesc*)k.d_blob, (const double*)(k.d_blob + k.off_lag), (const double*)(k.d_blob + k.off_lagh),
                 (const double*)(k.d_blob + k.off_lagsd), nullptr, k.nwin};
  return get_masks(c, m, frq, nf, WIN_NFC, &out->masks);
}

// plane-parallel air mass 1 / sin(elev) per elevation (NaN elevation -> NaN air mass: that angle's rows come out NaN)
int airmass_of(const double* elev, int nang, std::vector<double>* am) {
  am->resize(nang);
  for (int a = 0; a < nang; ++a) {
    // The wrapper tests ang = [elevation_k] per k (PyRTlib_processing.py:106, :117) and skips only that
    // k: a NaN elevation blanks its own [:, k, :] rows and nothing else.  Its air mass is NaN, which
    // the slant-path integration carries into every output of that angle; valid[] is about the
    // profile's own data and stays 1.
    if (std::isnan(elev[a])) { (*am)[a] = std::nan(""); continue; }
    // a path at or below the horizon has no plane-parallel air mass
    if (!(elev[a] > 0.0 && elev[a] < 180.0))
      return fail(MWRT_ERR_INVALID_ARGUMENT, "elevation angles must lie in (0, 180) degrees");
    (*am)[a] = 1.0 / std::sin(elev[a] * M_PI / 180.0);
  }
  return MWRT_OK;
}

// ---- fine-grid two-kernel form: K1 + layer step -> zenith layer optical depth in HBM -> RTE ----
int tau_pitch_of(int nf) { return ((nf + TAU_NFC - 1) / TAU_NFC) * TAU_NFC; }

// can the windowed absorption kernel serve this call (frequency list, level count, LDS)?
bool windowed_ok(const mwrt_context* c, const double* frq, int nf, int threads) {
  return windows_eligible(frq, nf) && threads <= 512 && absorb_win_lds_bytes(threads) <= (size_t)c->lds_max;
}

// K1 (+ layer step): d_tau [nprof][nlev][fpitch], d_valid [nprof].  Windowed kernel when the list qualifies
// (and the mode allows), else every line at every frequency.
int layer_tau_launch(mwrt_context* c, const mwrt_model* m, int64_t nprof, int nlev, const double* d_z, const double* d_p,
                     const double* d_t, const double* d_rh, int nf, const double* frq, const double* dev_frq,
                     double* d_tau, int fpitch, uint8_t* d_valid, hipStream_t st) {
  const int threads = tau_threads(nlev);
  if (threads > 1024) return fail(MWRT_ERR_UNSUPPORTED, "layer optical depths: nlev > 1009");
  const bool eligible = windowed_ok(c, frq, nf, threads);
  if (c->absorption_mode == 2 && !eligible)
    return fail(MWRT_ERR_UNSUPPORTED, "windowed absorption needs >= 128 strictly increasing frequencies in windows <= 6 GHz wide, "
                                      "<= 505 levels");
  HIP_TRY(hipMemsetAsync(d_valid, 1, (size_t)nprof, st));
  TauOut T{d_z, d_tau, d_valid, fpitch};
  if (eligible && c->absorption_mode != 1) {
    WinPtrs wp{};
    int rc = get_windows(c, m, frq, nf, &wp); if (rc) return rc;
    AbsorbWinArgs w{};
    w.M = m->d_desc; w.p = d_p; w.t = d_t; w.rh = d_rh; w.frq = dev_frq;
    w.win = wp.win; w.lagrange = wp.lag; w.lagrange_h = wp.lag_h; w.masks = wp.masks; w.lag_sd = wp.lag_sd;
    w.nlev = nlev; w.nf = nf; w.T = T;
    timing_begin(c, st);
    hipError_t e = launch_absorb_win(w, dim3((unsigned)nprof, (unsigned)wp.nwin), dim3(threads), st, true);
    timing_end(c, st);
    HIP_TRY(e);
    return MWRT_OK;
  }
  AbsorbArgs a{};
  a.M = m->d_desc; a.p = d_p; a.t = d_t; a.rh = d_rh; a.frq = dev_frq; a.nlev = nlev; a.nf = nf; a.T = T;
  { int rc = get_masks(c, m, frq, nf, TAU_NFC, &a.masks); if (rc) return rc; }
  timing_begin(c, st);
  hipError_t e = launch_absorb_tau(a, dim3((unsigned)nprof, (unsigned)((nf + TAU_NFC - 1) / TAU_NFC)), dim3(threads), st);
  timing_end(c, st);
  HIP_TRY(e);
  return MWRT_OK;
}

// K2: TBs from layer optical depths; the elevations go through in groups of <= 8 (or 10) per launch
int rte_tau_launch(mwrt_context* c, const mwrt_model* m, int64_t nprof, int nlev, const double* d_tau, int fpitch,
                   const double* d_t, int nf, const double* dev_frq, int nang, const double* dev_am, double* d_tb,
                   const uint8_t* d_valid, hipStream_t st) {
  RteTauArgs r{};
  r.M = m->d_desc; r.tau = d_tau; r.t = d_t; r.frq = dev_frq; r.airmass = dev_am; r.tb = d_tb; r.valid = d_valid;
  r.nlev = nlev; r.nf = nf; r.nang = nang; r.fpitch = fpitch;
  const dim3 grid((unsigned)nprof, (unsigned)((nf + RTE_THREADS - 1) / RTE_THREADS));
  const size_t lds = sizeof(double) * 2 * (size_t)nlev;
  for (int a0 = 0; a0 < nang;) {
    const int rem = nang - a0;
    const int na = rem <= 8 ? rem : (rem == 10 ? 10 : (rem == 9 ? 5 : 8));
    r.a0 = a0;
    timing_begin(c, st);
    hipError_t e = launch_rte_tau(r, grid, lds, st, na);
    timing_end(c, st);
    HIP_TRY(e);
    a0 += na;
  }
  return MWRT_OK;
}

}  // namespace

extern "C" {

int mwrt_version(void) { return MWRT_VERSION; }

size_t mwrt_model_desc_size(void) { return sizeof(mwrt_model_desc); }

int mwrt_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
  return n;
}

const char* mwrt_last_error(void) { return g_err.c_str(); }

int mwrt_create(int device_id, mwrt_context** out) {
  if (!out) return fail(MWRT_ERR_INVALID_ARGUMENT, "out is null");
  *out = nullptr;
  const int n = mwrt_device_count();
  if (n <= 0) return fail(MWRT_ERR_NO_DEVICE, "no HIP device: this library has no CPU path");
  if (device_id < 0 || device_id >= n) return fail(MWRT_ERR_INVALID_ARGUMENT, "device_id out of range");
  HIP_TRY(hipSetDevice(device_id));
  mwrt_context* c = new (std::nothrow) mwrt_context();
  if (!c) return fail(MWRT_ERR_OUT_OF_MEMORY, "host allocation failed");
  c->device = device_id;
  hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
  if (e != hipSuccess) { delete c; return fail(MWRT_ERR_HIP, hipGetErrorString(e)); }
  int lds = 0;
  if (hipDeviceGetAttribute(&lds, hipDeviceAttributeMaxSharedMemoryPerBlock, device_id) == hipSuccess && lds > 0)
    c->lds_max = lds;
  int cus = 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device_id) == hipSuccess && cus > 0) c->num_cus = cus;
  if (const char* mb = std::getenv("MWRT_ALPHA_BATCH_MB")) {       // diagnostic: size of a fine-grid profile batch
    const long v = std::atol(mb);
    if (v > 0) c->alpha_batch_bytes = (size_t)v << 20;
  }
  *out = c;
  return MWRT_OK;
}

int mwrt_destroy(mwrt_context* c) {
  if (!c) return MWRT_OK;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  for (auto& e : c->win_cache) (void)hipFree(e.d_blob);
  c->win_cache.clear();
  for (auto& e : c->mask_cache) (void)hipFree(e.d_masks);
  c->mask_cache.clear();
  c->frq_cache.release(); c->am_cache.release(); c->elev_cache.release(); c->d_amf.release(); c->d_duct.release(); c->d_alpha.release();
  c->d_in.release(); c->d_out.release();
  c->d_valid.release(); c->d_ex.release();
  for (hipEvent_t e : c->ev0) (void)hipEventDestroy(e);
  for (hipEvent_t e : c->ev1) (void)hipEventDestroy(e);
  if (c->ws_event) (void)hipEventDestroy(c->ws_event);
  (void)hipStreamDestroy(c->stream);
  delete c;
  return MWRT_OK;
}

int mwrt_model_create(mwrt_context* c, const mwrt_model_desc* desc, mwrt_model** out) {
  if (!c || !desc || !out) return fail(MWRT_ERR_INVALID_ARGUMENT, "null argument");
  *out = nullptr;
  if (desc->n_h2o < 0 || desc->n_h2o > MWRT_MAX_H2O_LINES || desc->n_o2 < 0 || desc->n_o2 > MWRT_MAX_O2_LINES)
    return fail(MWRT_ERR_INVALID_ARGUMENT, "line counts out of range");
  if (desc->n_x < 0 || desc->n_x > MWRT_MAX_X_LINES) return fail(MWRT_ERR_INVALID_ARGUMENT, "extra-species line count out of range");
  HIP_TRY(hipSetDevice(c->device));
  mwrt_model* m = new (std::nothrow) mwrt_model();
  if (!m) return fail(MWRT_ERR_OUT_OF_MEMORY, "host allocation failed");
  static std::atomic<uint64_t> next_id{1};
  m->id = next_id.fetch_add(1);
  static_cast<mwrt_model_desc&>(m->h_desc) = *desc;
  std::memset(m->h_desc.o2r, 0, sizeof(m->h_desc.o2r));
  std::memset(m->h_desc.h2or, 0, sizeof(m->h_desc.h2or));
  for (int k = 0; k < MWRT_MAX_O2_LINES; ++k) {
    const double rf2 = (k < desc->n_o2 && desc->o2_f[k] != 0.0) ? 1.0 / (desc->o2_f[k] * desc->o2_f[k]) : 0.0;
    m->h_desc.o2_rf2[k] = rf2;
    O2Rec& r = m->h_desc.o2r[k];
    r.f = desc->o2_f[k]; r.s300rf2 = desc->o2_s300[k] * rf2; r.be = desc->o2_be[k]; r.w300 = desc->o2_w300[k];
    r.y0 = desc->o2_y0[k]; r.y1 = desc->o2_y1[k]; r.g0 = desc->o2_g0[k]; r.g1 = desc->o2_g1[k];
    r.dnu0 = desc->o2_dnu0[k]; r.dnu1 = desc->o2_dnu1[k];
  }
  for (int k = 0; k < MWRT_MAX_H2O_LINES; ++k) {
    H2ORec& r = m->h_desc.h2or[k];
    const double fl = desc->h2o_fl[k];
    r.fl = fl; r.s1 = (k < desc->n_h2o && fl != 0.0) ? desc->h2o_s1[k] / (fl * fl) : 0.0; r.b2 = desc->h2o_b2[k];
    r.w0 = desc->h2o_w0[k]; r.x = desc->h2o_x[k]; r.w0s = desc->h2o_w0s[k]; r.xs = desc->h2o_xs[k];
    r.sh = desc->h2o_sh[k]; r.xh = desc->h2o_xh[k]; r.shs = desc->h2o_shs[k]; r.xhs = desc->h2o_xhs[k];
    r.aair = desc->h2o_aair[k]; r.aself = desc->h2o_aself[k]; r.w2 = desc->h2o_w2[k];
  }
  hipError_t e = hipMalloc((void**)&m->d_desc, sizeof(ModelFlat));
  if (e == hipSuccess) e = hipMemcpy(m->d_desc, &m->h_desc, sizeof(ModelFlat), hipMemcpyHostToDevice);
  if (e != hipSuccess) { if (m->d_desc) (void)hipFree(m->d_desc); delete m; return fail(MWRT_ERR_HIP, hipGetErrorString(e)); }
  *out = m;
  return MWRT_OK;
}

int mwrt_model_destroy(mwrt_context* c, mwrt_model* m) {
  if (!m) return MWRT_OK;
  if (c) {
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();                       // launches on any stream may still read the tables / windows
    for (size_t i = c->win_cache.size(); i-- > 0;)      // window descriptors are keyed by the model: drop this one's
      if (c->win_cache[i].model_id == m->id) {
        (void)hipFree(c->win_cache[i].d_blob);
        c->win_cache.erase(c->win_cache.begin() + (long)i);
      }
    for (size_t i = c->mask_cache.size(); i-- > 0;)
      if (c->mask_cache[i].model_id == m->id) {
        (void)hipFree(c->mask_cache[i].d_masks);
        c->mask_cache.erase(c->mask_cache.begin() + (long)i);
      }
  }
  if (m->d_desc) (void)hipFree(m->d_desc);
  delete m;
  return MWRT_OK;
}

// core of every TB entry point: nmodels absorption models x nprof profiles in ONE launch
static int tb_launch(mwrt_context* c, int nmodels, const mwrt_model* const* ms, int64_t nprof, int32_t nlev,
                     const double* d_z, const double* d_p, const double* d_t, const double* d_rh,
                     int32_t nf, const double* frq, int32_t nang, const double* elev,
                     double* d_tb, uint8_t* d_valid, const mwrt_tb_extras* ex, void* stream,
                     const mwrt_tb_options* opt = nullptr, const double* d_awet = nullptr, const double* d_adry = nullptr) {
  if (nmodels < 1 || nmodels > MAX_MULTI || !ms) return fail(MWRT_ERR_INVALID_ARGUMENT, "nmodels must be 1..8");
  const bool cloudy = opt && (opt->denliq || opt->denice);
  const bool rays = opt && opt->ray_tracing != 0;
  const bool ozone = opt && opt->o3n;
  const bool use_opt = cloudy || rays || ozone;
  if (ozone)
    for (int i = 0; i < nmodels; ++i)
      if (ms[i] && ms[i]->h_desc.n_x <= 0)
        return fail(MWRT_ERR_UNSUPPORTED, "o3n given but the model carries no extra-species line table (mwrt_model_desc.n_x = 0)");
  for (int i = 0; i < nmodels; ++i) {
    int rc = check_common(c, ms[i], nprof, nlev, nf);
    if (rc) return rc;
  }
  if (nang < 1 || nang > MWRT_MAX_ANGLES) return fail(MWRT_ERR_INVALID_ARGUMENT, "nang out of range");
  const bool from_alpha = d_awet != nullptr;
  if (!d_z || !d_t || !frq || !elev || !d_tb || !d_valid || (!from_alpha && (!d_p || !d_rh)) || (from_alpha && !d_adry))
    return fail(MWRT_ERR_INVALID_ARGUMENT, "null buffer");
  const int64_t rows = nprof * nmodels;
  if (rows > 2147483647LL) return fail(MWRT_ERR_UNSUPPORTED, "nmodels x nprof exceeds grid limit");
  HIP_TRY(hipSetDevice(c->device));
  hipStream_t st = resolve_stream(c, stream);
  if (nprof == 0) return MWRT_OK;
  const size_t nout = (size_t)rows * nang * nf;
  bool all_elev_nan = true;
  for (int a = 0; a < nang; ++a) all_elev_nan = all_elev_nan && std::isnan(elev[a]);
  if (any_nan(frq, nf) || all_elev_nan) {
    // check_for_nans covers frqs and ang too (PyRTlib_processing.py:77-78): a NaN frequency (the
    // wrapper's frqs array is shared by every call) leaves everything NaN, valid = 0
    HIP_TRY(hipMemsetAsync(d_tb, 0xFF, nout * sizeof(double), st));
    HIP_TRY(hipMemsetAsync(d_valid, 0, (size_t)rows, st));
    if (ex) {
      if (ex->tbatm) HIP_TRY(hipMemsetAsync(ex->tbatm, 0xFF, nout * sizeof(double), st));
      if (ex->tmr) HIP_TRY(hipMemsetAsync(ex->tmr, 0xFF, nout * sizeof(double), st));
      if (ex->tauwet) HIP_TRY(hipMemsetAsync(ex->tauwet, 0xFF, nout * sizeof(double), st));
      if (ex->taudry) HIP_TRY(hipMemsetAsync(ex->taudry, 0xFF, nout * sizeof(double), st));
      if (ex->taulay) HIP_TRY(hipMemsetAsync(ex->taulay, 0xFF, (size_t)rows * nf * nlev * sizeof(double), st));
      if (ex->tauliq) HIP_TRY(hipMemsetAsync(ex->tauliq, 0xFF, nout * sizeof(double), st));
      if (ex->tauice) HIP_TRY(hipMemsetAsync(ex->tauice, 0xFF, nout * sizeof(double), st));
    }
    return MWRT_OK;
  }
  std::vector<double> am;
  int rc = airmass_of(elev, nang, &am); if (rc) return rc;
  const double *dev_frq = nullptr, *dev_am = nullptr;
  rc = upload_small(c, c->frq_cache, frq, nf, &dev_frq); if (rc) return rc;
  rc = upload_small(c, c->am_cache, am.data(), nang, &dev_am); if (rc) return rc;

  FusedArgs a{};
  for (int i = 0; i < nmodels; ++i) a.Ms[i] = ms[i]->d_desc;
  a.nprof_in = nprof;
  a.z = d_z; a.p = d_p; a.t = d_t; a.rh = d_rh;
  a.frq = dev_frq; a.airmass = dev_am;
  a.tb = d_tb; a.valid = d_valid;
  if (ex) { a.tbatm = ex->tbatm; a.tmr = ex->tmr; a.tauwet = ex->tauwet; a.taudry = ex->taudry; a.taulay = ex->taulay; }
  a.nlev = nlev; a.nf = nf; a.nang = nang;
  if (ex) { a.tauliq = ex->tauliq; a.tauice = ex->tauice; }
  // (clear sky: the kernel writes the cloud columns itself -- 0 x air mass for good rows, NaN for blanked profiles
  // and NaN elevations, like every other column)
  if (cloudy) { a.denliq = opt->denliq; a.denice = opt->denice; }
  if (ozone) a.o3n = opt->o3n;
  if (rays) {
    // RTEquation.refractivity + ray_tracing [EXT] as a pre-kernel on the same stream: path factor ds/dz per
    // (profile, angle, layer) into the context's workspace (grown only, never shrunk)
    const double* dev_elev = nullptr;
    rc = upload_small(c, c->elev_cache, elev, nang, &dev_elev); if (rc) return rc;
    const size_t need = (size_t)nprof * nang * nlev * sizeof(double);
    if (need > c->d_amf.cap || (size_t)nprof > c->d_duct.cap) {
      HIP_TRY(hipDeviceSynchronize());                // queued launches may still read the old workspace
      HIP_TRY(c->d_amf.reserve(need));
      HIP_TRY(c->d_duct.reserve((size_t)nprof));
    }
    rc = workspace_acquire(c, st); if (rc) return rc;
    HIP_TRY(hipMemsetAsync(c->d_duct.p, 0, (size_t)nprof, st));
    const int rthreads = ((nlev + WAVE - 1) / WAVE) * WAVE;
    hipLaunchKernelGGL(k_ray_paths, dim3((unsigned)nprof), dim3(rthreads), 0, st, d_z, d_p, d_t, d_rh, (int)nlev, dev_elev,
                       (int)nang, c->d_amf.as<double>(), c->d_duct.as<uint8_t>());
    HIP_TRY(hipGetLastError());
    a.amf = c->d_amf.as<double>();
    a.duct = c->d_duct.as<uint8_t>();
  }
  const bool extras = ex && (ex->tbatm || ex->tmr || ex->tauwet || ex->taudry || ex->taulay || ex->tauliq || ex->tauice);
  int variant = extras ? FUSED_FULL : (use_opt ? FUSED_OPT : FUSED_TB_ONLY);
  if (from_alpha) {
    if (extras || use_opt || nmodels != 1)
      return fail(MWRT_ERR_UNSUPPORTED, "RTE from absorption: one model, no extras, no options");
    a.awet_in = d_awet; a.adry_in = d_adry;
    variant = FUSED_FROM_ALPHA;
  }
  // Fine spectral grids (BASELINE configs[4]): K1 -> tau -> K2.  The windowed absorption kernel (k_absorb_win) is
  // ~1.8x the fused kernel's K1 on such grids and ends each chunk with the layer step, so what crosses HBM is the
  // zenith layer optical depth: 8 B per (profile, level, frequency), written once, read once by k_rte_tau
  // (lane = frequency).  Profile batches of <= alpha_batch_bytes, both kernels on the caller's stream.
  if (variant == FUSED_TB_ONLY && nmodels == 1 && c->absorption_mode != 1 && windowed_ok(c, frq, nf, tau_threads(nlev))) {
    const int fpitch = tau_pitch_of(nf);
    const size_t per_prof = (size_t)nlev * fpitch * sizeof(double);
    const int64_t batch = std::max<int64_t>(1, std::min<int64_t>(nprof, (int64_t)(c->alpha_batch_bytes / per_prof)));
    bool have_ws = true;
    if ((size_t)batch * per_prof > c->d_alpha.cap) {
      HIP_TRY(hipDeviceSynchronize());                // queued launches may still read the old workspace
      const hipError_t e = c->d_alpha.reserve((size_t)batch * per_prof);
      if (e == hipErrorOutOfMemory) { (void)hipGetLastError(); have_ws = false; }   // no room for tau: the fused kernel
      else HIP_TRY(e);                                                              // needs no workspace at all
    }
    if (have_ws) {
      rc = workspace_acquire(c, st); if (rc) return rc;
      for (int64_t b0 = 0; b0 < nprof; b0 += batch) {
        const int64_t nb = std::min(batch, nprof - b0);
        double* d_tau = c->d_alpha.as<double>();
        rc = layer_tau_launch(c, ms[0], nb, nlev, d_z + b0 * nlev, d_p + b0 * nlev, d_t + b0 * nlev, d_rh + b0 * nlev, nf, frq,
                              dev_frq, d_tau, fpitch, d_valid + b0, st);
        if (rc) return rc;
        rc = rte_tau_launch(c, ms[0], nb, nlev, d_tau, fpitch, d_t + b0 * nlev, nf, dev_frq, nang, dev_am,
                            d_tb + (size_t)b0 * nang * nf, d_valid + b0, st);
        if (rc) return rc;
      }
      return workspace_release(c, st);
    }
  }
  const int nfc_main = pick_nfc_fused(c, nlev, nf, nang);
  if (variant != FUSED_FROM_ALPHA)
    for (int i = 0; i < nmodels; ++i) { rc = get_masks(c, ms[i], frq, nf, nfc_main, &a.masks[i]); if (rc) return rc; }
  rc = launch_fused(c, nfc_main, a, rows, st, variant);
  if (rc == MWRT_OK && rays) rc = workspace_release(c, st);
  return rc;
}

int mwrt_tb_batch_device(mwrt_context* c, const mwrt_model* m, int64_t nprof, int32_t nlev,
                         const double* d_z, const double* d_p, const double* d_t, const double* d_rh,
                         int32_t nf, const double* frq, int32_t nang, const double* elev,
                         double* d_tb, uint8_t* d_valid, const mwrt_tb_extras* ex, void* stream) {
  if (!c || !m) return fail(MWRT_ERR_INVALID_ARGUMENT, "null context or model");
  return tb_launch(c, 1, &m, nprof, nlev, d_z, d_p, d_t, d_rh, nf, frq, nang, elev, d_tb, d_valid, ex, stream);
}

int mwrt_tb_batch_opt_device(mwrt_context* c, const mwrt_model* m, int64_t nprof, int32_t nlev,
                             const double* d_z, const double* d_p, const double* d_t, const double* d_rh,
                             int32_t nf, const double* frq, int32_t nang, const double* elev,
                             double* d_tb, uint8_t* d_valid, const mwrt_tb_extras* ex, const mwrt_tb_options* opt,
                             void* stream) {
  if (!c || !m) return fail(MWRT_ERR_INVALID_ARGUMENT, "null context or model");
  return tb_launch(c, 1, &m, nprof, nlev, d_z, d_p, d_t, d_rh, nf, frq, nang, elev, d_tb, d_valid, ex, stream, opt);
}

int mwrt_tb_from_absorption_device(mwrt_context* c, const mwrt_model* m, int64_t nprof, int32_t nlev,
                                   const double* d_z, const double* d_t, int32_t nf, const double* frq, int32_t nang,
                                   const double* elev, const double* d_awet, const double* d_adry,
                                   double* d_tb, uint8_t* d_valid, void* stream) {
  if (!c || !m) return fail(MWRT_ERR_INVALID_ARGUMENT, "null context or model");
  if (!d_awet || !d_adry) return fail(MWRT_ERR_INVALID_ARGUMENT, "null buffer");
  return tb_launch(c, 1, &m, nprof, nlev, d_z, nullptr, d_t, nullptr, nf, frq, nang, elev, d_tb, d_valid, nullptr, stream,
                   nullptr, d_awet, d_adry);
}

int mwrt_tb_batch_multi_device(mwrt_context* c, int32_t nmodels, const mwrt_model* const* models, int64_t nprof,
                               int32_t nlev, const double* d_z, const double* d_p, const double* d_t,
                               const double* d_rh, int32_t nf, const double* frq, int32_t nang, const double* elev,
                               double* d_tb, uint8_t* d_valid, void* stream) {
  if (!c) return fail(MWRT_ERR_INVALID_ARGUMENT, "null context");
  return tb_launch(c, nmodels, models, nprof, nlev, d_z, d_p, d_t, d_rh, nf, frq, nang, elev, d_tb, d_valid, nullptr,
                   stream);
}

int mwrt_tb_batch_multi(mwrt_context* c, int32_t nmodels, const mwrt_model* const* models, int64_t nprof, int32_t nlev,
                        const double* z, const double* p, const double* t, const double* rh,
                        int32_t nf, const double* frq, int32_t nang, const double* elev,
                        double* tb, uint8_t* valid) {
  if (!c || !models || nmodels < 1 || nmodels > MAX_MULTI) return fail(MWRT_ERR_INVALID_ARGUMENT, "bad context / models");
  if (!z || !p || !t || !rh || !frq || !elev || !tb || !valid) return fail(MWRT_ERR_INVALID_ARGUMENT, "null buffer");
  if (nprof < 0 || nlev < 2 || nf < 1 || nang < 1) return fail(MWRT_ERR_INVALID_ARGUMENT, "bad sizes");
  if (nprof == 0) return MWRT_OK;
  HIP_TRY(hipSetDevice(c->device));
  hipStream_t st = c->stream;
  const size_t nin = (size_t)nprof * nlev, rows = (size_t)nprof * nmodels, nout = rows * nang * nf;
  HIP_TRY(c->d_in.reserve(4 * nin * sizeof(double)));
  HIP_TRY(c->d_out.reserve(nout * sizeof(double)));
  HIP_TRY(c->d_valid.reserve(rows));
  double* din = c->d_in.as<double>();
  const double* src[4] = {z, p, t, rh};
  for (int k = 0; k < 4; ++k)                        // the profiles cross PCIe once for all models
    HIP_TRY(hipMemcpyAsync(din + k * nin, src[k], nin * sizeof(double), hipMemcpyHostToDevice, st));
  int rc = tb_launch(c, nmodels, models, nprof, nlev, din, din + nin, din + 2 * nin, din + 3 * nin, nf, frq, nang, elev,
                     c->d_out.as<double>(), c->d_valid.as<uint8_t>(), nullptr, st);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(tb, c->d_out.p, nout * sizeof(double), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(valid, c->d_valid.p, rows, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  const double qnan = std::nan("");
  for (size_t i = 0; i < rows; ++i)
    if (valid[i] == 2) for (size_t o = 0; o < (size_t)nang * nf; ++o) tb[i * nang * nf + o] = qnan;
  return MWRT_OK;
}

int mwrt_tb_batch(mwrt_context* c, const mwrt_model* m, int64_t nprof, int32_t nlev,
                  const double* z, const double* p, const double* t, const double* rh,
                  int32_t nf, const double* frq, int32_t nang, const double* elev,
                  double* tb, uint8_t* valid, const mwrt_tb_extras* ex) {
  return mwrt_tb_batch_opt(c, m, nprof, nlev, z, p, t, rh, nf, frq, nang, elev, tb, valid, ex, nullptr);
}

int mwrt_tb_batch_opt(mwrt_context* c, const mwrt_model* m, int64_t nprof, int32_t nlev,
                      const double* z, const double* p, const double* t, const double* rh,
                      int32_t nf, const double* frq, int32_t nang, const double* elev,
                      double* tb, uint8_t* valid, const mwrt_tb_extras* ex, const mwrt_tb_options* opt) {
  int rc = check_common(c, m, nprof, nlev, nf);
  if (rc) return rc;
  if (nang < 1 || nang > MWRT_MAX_ANGLES) return fail(MWRT_ERR_INVALID_ARGUMENT, "nang out of range");
  if (!z || !p || !t || !rh || !frq || !elev || !tb || !valid) return fail(MWRT_ERR_INVALID_ARGUMENT, "null buffer");
  if (nprof == 0) return MWRT_OK;
  HIP_TRY(hipSetDevice(c->device));
  hipStream_t st = c->stream;
  const size_t nin = (size_t)nprof * nlev, nout = (size_t)nprof * nang * nf;
  const size_t nlay = (size_t)nprof * nf * nlev;
  const bool has_liq = opt && opt->denliq, has_ice = opt && opt->denice, has_o3 = opt && opt->o3n;
  HIP_TRY(c->d_in.reserve((4 + (has_liq ? 1 : 0) + (has_ice ? 1 : 0) + (has_o3 ? 1 : 0)) * nin * sizeof(double)));
  HIP_TRY(c->d_out.reserve(nout * sizeof(double)));
  HIP_TRY(c->d_valid.reserve((size_t)nprof));
  double* din = c->d_in.as<double>();
  const double* src[4] = {z, p, t, rh};
  for (int k = 0; k < 4; ++k)
    HIP_TRY(hipMemcpyAsync(din + k * nin, src[k], nin * sizeof(double), hipMemcpyHostToDevice, st));
  mwrt_tb_options dopt{};
  if (opt) {
    double* q = din + 4 * nin;
    if (has_liq) { HIP_TRY(hipMemcpyAsync(q, opt->denliq, nin * sizeof(double), hipMemcpyHostToDevice, st)); dopt.denliq = q; q += nin; }
    if (has_ice) { HIP_TRY(hipMemcpyAsync(q, opt->denice, nin * sizeof(double), hipMemcpyHostToDevice, st)); dopt.denice = q; q += nin; }
    if (has_o3) { HIP_TRY(hipMemcpyAsync(q, opt->o3n, nin * sizeof(double), hipMemcpyHostToDevice, st)); dopt.o3n = q; }
    dopt.ray_tracing = opt->ray_tracing;
  }
  constexpr int NEX = 7;                               // tbatm tmr tauwet taudry taulay tauliq tauice
  mwrt_tb_extras dex{};
  double* host_ex[NEX] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  double** slots[NEX] = {&dex.tbatm, &dex.tmr, &dex.tauwet, &dex.taudry, &dex.taulay, &dex.tauliq, &dex.tauice};
  auto exlen = [&](int k) { return k == 4 ? nlay : nout; };
  if (ex) {
    host_ex[0] = ex->tbatm; host_ex[1] = ex->tmr; host_ex[2] = ex->tauwet; host_ex[3] = ex->taudry; host_ex[4] = ex->taulay;
    host_ex[5] = ex->tauliq; host_ex[6] = ex->tauice;
    size_t need = 0;
    for (int k = 0; k < NEX; ++k) if (host_ex[k]) need += exlen(k);
    HIP_TRY(c->d_ex.reserve(need * sizeof(double) + 8));
    double* q = c->d_ex.as<double>();
    for (int k = 0; k < NEX; ++k) if (host_ex[k]) { *slots[k] = q; q += exlen(k); }
  }
  rc = mwrt_tb_batch_opt_device(c, m, nprof, nlev, din, din + nin, din + 2 * nin, din + 3 * nin, nf, frq, nang, elev,
                                c->d_out.as<double>(), c->d_valid.as<uint8_t>(), ex ? &dex : nullptr, opt ? &dopt : nullptr,
                                st);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(tb, c->d_out.p, nout * sizeof(double), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(valid, c->d_valid.p, (size_t)nprof, hipMemcpyDeviceToHost, st));
  if (ex) {
    for (int k = 0; k < NEX; ++k)
      if (host_ex[k]) HIP_TRY(hipMemcpyAsync(host_ex[k], *slots[k], exlen(k) * sizeof(double), hipMemcpyDeviceToHost, st));
  }
  HIP_TRY(hipStreamSynchronize(st));
  // a profile flagged 2 (negative absorption: pyrtlib raises for the whole execute()) is blanked
  // as a whole, whichever frequency chunk met it
  const double qnan = std::nan("");
  for (int64_t i = 0; i < nprof; ++i) {
    if (valid[i] != 2) continue;
    for (size_t o = 0; o < (size_t)nang * nf; ++o) tb[(size_t)i * nang * nf + o] = qnan;
    for (int k = 0; k < NEX; ++k) {
      if (!host_ex[k]) continue;
      const size_t per = k == 4 ? (size_t)nf * nlev : (size_t)nang * nf;
      for (size_t o = 0; o < per; ++o) host_ex[k][(size_t)i * per + o] = qnan;
    }
  }
  return MWRT_OK;
}

int mwrt_absorption_batch_device(mwrt_context* c, const mwrt_model* m, int64_t nprof, int32_t nlev,
                                 const double* d_p, const double* d_t, const double* d_rh,
                                 int32_t nf, const double* frq, double* d_awet, double* d_adry, void* stream) {
  int rc = check_common(c, m, nprof, nlev, nf);
  if (rc) return rc;
  if (!d_p || !d_t || !d_rh || !frq || !d_awet || !d_adry) return fail(MWRT_ERR_INVALID_ARGUMENT, "null buffer");
  if (any_nan(frq, nf)) return fail(MWRT_ERR_INVALID_ARGUMENT, "NaN frequency");
  HIP_TRY(hipSetDevice(c->device));
  hipStream_t st = resolve_stream(c, stream);
  if (nprof == 0) return MWRT_OK;
  const double* dev_frq = nullptr;
  rc = upload_small(c, c->frq_cache, frq, nf, &dev_frq); if (rc) return rc;
  const int wthreads = ((nlev + WAVE - 1) / WAVE) * WAVE;
  const bool eligible = windowed_ok(c, frq, nf, wthreads);            // node sums in LDS: 2 x 16 doubles per thread
  if (c->absorption_mode == 2 && !eligible)
    return fail(MWRT_ERR_UNSUPPORTED, "windowed absorption needs >= 128 strictly increasing frequencies in windows <= 6 GHz wide");
  if (eligible && c->absorption_mode != 1) {
    WinPtrs wp{};
    rc = get_windows(c, m, frq, nf, &wp); if (rc) return rc;
    AbsorbWinArgs w{};
    w.M = m->d_desc; w.p = d_p; w.t = d_t; w.rh = d_rh; w.frq = dev_frq;
    w.win = wp.win; w.lagrange = wp.lag; w.lagrange_h = wp.lag_h; w.masks = wp.masks; w.lag_sd = wp.lag_sd;
    w.awet = d_awet; w.adry = d_adry; w.nlev = nlev; w.nf = nf;
    timing_begin(c, st);
    hipError_t e = launch_absorb_win(w, dim3((unsigned)nprof, (unsigned)wp.nwin), dim3(wthreads), st, false);
    timing_end(c, st);
    HIP_TRY(e);
    return MWRT_OK;
  }
  AbsorbArgs a{};
  a.M = m->d_desc; a.p = d_p; a.t = d_t; a.rh = d_rh; a.frq = dev_frq;
  a.awet = d_awet; a.adry = d_adry; a.nlev = nlev; a.nf = nf;
  rc = get_masks(c, m, frq, nf, pick_nfc(nf), &a.masks); if (rc) return rc;
  return launch_absorb(c, pick_nfc(nf), a, nprof, st);
}

int mwrt_absorption_batch(mwrt_context* c, const mwrt_model* m, int64_t nprof, int32_t nlev,
                          const double* p, const double* t, const double* rh,
                          int32_t nf, const double* frq, double* awet, double* adry) {
  int rc = check_common(c, m, nprof, nlev, nf);
  if (rc) return rc;
  if (!p || !t || !rh || !frq || !awet || !adry) return fail(MWRT_ERR_INVALID_ARGUMENT, "null buffer");
  if (nprof == 0) return MWRT_OK;
  HIP_TRY(hipSetDevice(c->device));
  hipStream_t st = c->stream;
  const size_t nin = (size_t)nprof * nlev, nout = (size_t)nprof * nf * nlev;
  HIP_TRY(c->d_in.reserve(3 * nin * sizeof(double)));
  HIP_TRY(c->d_out.reserve(2 * nout * sizeof(double)));
  double* din = c->d_in.as<double>();
  const double* src[3] = {p, t, rh};
  for (int k = 0; k < 3; ++k)
    HIP_TRY(hipMemcpyAsync(din + k * nin, src[k], nin * sizeof(double), hipMemcpyHostToDevice, st));
  double* dout = c->d_out.as<double>();
  rc = mwrt_absorption_batch_device(c, m, nprof, nlev, din, din + nin, din + 2 * nin, nf, frq, dout, dout + nout, st);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(awet, dout, nout * sizeof(double), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(adry, dout + nout, nout * sizeof(double), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  return MWRT_OK;
}

int mwrt_layer_tau_pitch(int32_t nf) { return nf < 1 ? 0 : tau_pitch_of(nf); }

int mwrt_layer_tau_batch_device(mwrt_context* c, const mwrt_model* m, int64_t nprof, int32_t nlev,
                                const double* d_z, const double* d_p, const double* d_t, const double* d_rh,
                                int32_t nf, const double* frq, double* d_tau, int32_t tau_pitch, uint8_t* d_valid,
                                void* stream) {
  int rc = check_common(c, m, nprof, nlev, nf);
  if (rc) return rc;
  if (!d_z || !d_p || !d_t || !d_rh || !frq || !d_tau || !d_valid) return fail(MWRT_ERR_INVALID_ARGUMENT, "null buffer");
  if (tau_pitch < tau_pitch_of(nf) || tau_pitch % TAU_NFC != 0)
    return fail(MWRT_ERR_INVALID_ARGUMENT, "tau_pitch must be a multiple of 16 and >= mwrt_layer_tau_pitch(nf)");
  if (any_nan(frq, nf)) return fail(MWRT_ERR_INVALID_ARGUMENT, "NaN frequency");
  HIP_TRY(hipSetDevice(c->device));
  hipStream_t st = resolve_stream(c, stream);
  if (nprof == 0) return MWRT_OK;
  const double* dev_frq = nullptr;
  rc = upload_small(c, c->frq_cache, frq, nf, &dev_frq); if (rc) return rc;
  return layer_tau_launch(c, m, nprof, nlev, d_z, d_p, d_t, d_rh, nf, frq, dev_frq, d_tau, tau_pitch, d_valid, st);
}

int mwrt_tb_from_layer_tau_device(mwrt_context* c, const mwrt_model* m, int64_t nprof, int32_t nlev,
                                  const double* d_tau, int32_t tau_pitch, const double* d_t,
                                  int32_t nf, const double* frq, int32_t nang, const double* elev,
                                  const uint8_t* d_valid, double* d_tb, void* stream) {
  int rc = check_common(c, m, nprof, nlev, nf);
  if (rc) return rc;
  if (nang < 1 || nang > MWRT_MAX_ANGLES) return fail(MWRT_ERR_INVALID_ARGUMENT, "nang out of range");
  if (!d_tau || !d_t || !frq || !elev || !d_valid || !d_tb) return fail(MWRT_ERR_INVALID_ARGUMENT, "null buffer");
  if (tau_pitch < nf) return fail(MWRT_ERR_INVALID_ARGUMENT, "tau_pitch < nf");
  if (any_nan(frq, nf)) return fail(MWRT_ERR_INVALID_ARGUMENT, "NaN frequency");
  HIP_TRY(hipSetDevice(c->device));
  hipStream_t st = resolve_stream(c, stream);
  if (nprof == 0) return MWRT_OK;
  std::vector<double> am;
  rc = airmass_of(elev, nang, &am); if (rc) return rc;
  const double *dev_frq = nullptr, *dev_am = nullptr;
  rc = upload_small(c, c->frq_cache, frq, nf, &dev_frq); if (rc) return rc;
  rc = upload_small(c, c->am_cache, am.data(), nang, &dev_am); if (rc) return rc;
  return rte_tau_launch(c, m, nprof, nlev, d_tau, tau_pitch, d_t, nf, dev_frq, nang, dev_am, d_tb, d_valid, st);
}

// K-matrix (dTB/dT, dTB/de, dTB/d thickness per level) -- see k_tb_jacobian.  HOST buffers, synchronous.
int mwrt_tb_jacobian_batch(mwrt_context* c, const mwrt_model* m, int64_t nprof, int32_t nlev,
                           const double* z, const double* p, const double* t, const double* rh,
                           int32_t nf, const double* frq, int32_t nang, const double* elev,
                           double* tb, double* dtb_dt, double* dtb_de, double* dtb_ddz, uint8_t* valid) {
  int rc = check_common(c, m, nprof, nlev, nf);
  if (rc) return rc;
  if (nang < 1 || nang > MWRT_MAX_ANGLES) return fail(MWRT_ERR_INVALID_ARGUMENT, "nang out of range");
  if (!z || !p || !t || !rh || !frq || !elev || !tb || !dtb_dt || !dtb_de || !dtb_ddz || !valid)
    return fail(MWRT_ERR_INVALID_ARGUMENT, "null buffer");
  if (any_nan(frq, nf)) return fail(MWRT_ERR_INVALID_ARGUMENT, "NaN frequency");
  if (nprof == 0) return MWRT_OK;
  HIP_TRY(hipSetDevice(c->device));
  hipStream_t st = c->stream;
  std::vector<double> am;
  rc = airmass_of(elev, nang, &am); if (rc) return rc;
  const double *dev_frq = nullptr, *dev_am = nullptr;
  rc = upload_small(c, c->frq_cache, frq, nf, &dev_frq); if (rc) return rc;
  rc = upload_small(c, c->am_cache, am.data(), nang, &dev_am); if (rc) return rc;
  constexpr double DT = 0.01, REL_E = 1e-4, MIN_DE = 1e-7;       // local steps of the absorption derivatives
  // Goff-Gratch over water, as RTEquation.vapor [EXT] (host copy: only used to keep e fixed while T moves)
  auto es_of = [](double tk) {
    const double y = 373.16 / tk;
    const double es = -7.90298 * (y - 1.0) + 5.02808 * std::log10(y) - 1.3816e-07 * (std::pow(10.0, 11.344 * (1.0 - (1.0 / y))) - 1.0) +
                      0.0081328 * (std::pow(10.0, -3.49149 * (y - 1.0)) - 1.0) + std::log10(1013.246);
    return std::pow(10.0, es);
  };
  // profile batches: the five absorption sets + three Jacobian arrays of one batch stay under ~1 GiB
  const size_t per_prof = sizeof(double) * ((size_t)10 * nf * nlev + (size_t)3 * nang * nf * nlev + (size_t)nang * nf + 12 * (size_t)nlev) + 1;
  const int64_t batch = std::max<int64_t>(1, std::min<int64_t>(nprof, (int64_t)(((size_t)1 << 30) / per_prof)));
  DevBuf ws;                                                      // own workspace: freed on return
  HIP_TRY(ws.reserve((size_t)batch * per_prof + 4096));
  for (int64_t b0 = 0; b0 < nprof; b0 += batch) {
    const int64_t nb = std::min(batch, nprof - b0);
    const size_t nin = (size_t)nb * nlev, nabs = (size_t)nb * nf * nlev, njac = (size_t)nb * nang * nf * nlev, ntb = (size_t)nb * nang * nf;
    // host side: the 12 level arrays  z p t rh | t+ rh(T+) | t- rh(T-) | rh(e+) rh(e-) | de | (pad)
    std::vector<double> h(12 * nin);
    double *hz = h.data(), *hp = hz + nin, *ht = hp + nin, *hrh = ht + nin, *htp = hrh + nin, *hrp = htp + nin, *htm = hrp + nin,
           *hrm = htm + nin, *hep = hrm + nin, *hem = hep + nin, *hde = hem + nin;
    for (size_t i = 0; i < nin; ++i) {
      const size_t g = (size_t)b0 * nlev + i;
      hz[i] = z[g]; hp[i] = p[g]; ht[i] = t[g]; hrh[i] = rh[g];
      const double es = es_of(t[g]), e = rh[g] * es;
      htp[i] = t[g] + DT; hrp[i] = e / es_of(t[g] + DT);
      htm[i] = t[g] - DT; hrm[i] = e / es_of(t[g] - DT);
      const double de = std::max(std::fabs(e) * REL_E, MIN_DE);
      const double lo = std::max(e - de, 0.0);                     // a dry level: the step stays on the non-negative side
      hep[i] = (lo + 2.0 * de) / es; hem[i] = lo / es; hde[i] = de;
    }
    double* base = ws.as<double>();
    double* d_lev = base;                                          // 12 x nin
    double* d_abs = d_lev + 12 * nin;                              // 10 x nabs
    double* d_jac = d_abs + 10 * nabs;                             // 3 x njac
    double* d_tb = d_jac + 3 * njac;
    uint8_t* d_valid = (uint8_t*)(d_tb + ntb);
    HIP_TRY(hipMemcpyAsync(d_lev, h.data(), sizeof(double) * 12 * nin, hipMemcpyHostToDevice, st));
    const double *dz_ = d_lev, *dp_ = d_lev + nin, *dt_ = d_lev + 2 * nin, *drh_ = d_lev + 3 * nin;
    const double* T_of[5] = {dt_, d_lev + 4 * nin, d_lev + 6 * nin, dt_, dt_};
    const double* RH_of[5] = {drh_, d_lev + 5 * nin, d_lev + 7 * nin, d_lev + 8 * nin, d_lev + 9 * nin};
    JacArgs a{};
    for (int v = 0; v < 5; ++v) {
      double* aw = d_abs + (size_t)(2 * v) * nabs; double* ad = aw + nabs;
      rc = mwrt_absorption_batch_device(c, m, nb, nlev, dp_, T_of[v], RH_of[v], nf, frq, aw, ad, st);
      if (rc) return rc;
      a.a[v][0] = aw; a.a[v][1] = ad;
    }
    a.M = m->d_desc; a.z = dz_; a.t = dt_; a.de = d_lev + 10 * nin; a.dT = DT; a.frq = dev_frq; a.airmass = dev_am;
    a.tb = d_tb; a.dtb_dt = d_jac; a.dtb_de = d_jac + njac; a.dtb_ddz = d_jac + 2 * njac; a.valid = d_valid;
    a.nprof = nb; a.nlev = nlev; a.nf = nf; a.nang = nang;
    HIP_TRY(hipMemsetAsync(d_valid, 1, (size_t)nb, st));
    const int64_t nthreads = nb * nf * nang;
    timing_begin(c, st);
    hipLaunchKernelGGL(k_tb_jacobian, dim3((unsigned)((nthreads + 63) / 64)), dim3(64), 0, st, a);
    timing_end(c, st);
    HIP_TRY(hipGetLastError());
    const size_t o = (size_t)b0 * nang * nf;
    HIP_TRY(hipMemcpyAsync(tb + o, d_tb, sizeof(double) * ntb, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(dtb_dt + o * nlev, a.dtb_dt, sizeof(double) * njac, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(dtb_de + o * nlev, a.dtb_de, sizeof(double) * njac, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(dtb_ddz + o * nlev, a.dtb_ddz, sizeof(double) * njac, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(valid + b0, d_valid, (size_t)nb, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
  }
  ws.release();
  // a profile flagged 0 / 2 is blanked as a whole, whichever thread met it
  const double qnan = std::nan("");
  for (int64_t i = 0; i < nprof; ++i) {
    if (valid[i] == 1) continue;
    const size_t o = (size_t)i * nang * nf;
    for (size_t k = 0; k < (size_t)nang * nf; ++k) tb[o + k] = qnan;
    for (size_t k = 0; k < (size_t)nang * nf * nlev; ++k) { dtb_dt[o * nlev + k] = qnan; dtb_de[o * nlev + k] = qnan; dtb_ddz[o * nlev + k] = qnan; }
  }
  return MWRT_OK;
}

int mwrt_set_chunk_width(mwrt_context* c, int width) {
  if (!c) return fail(MWRT_ERR_INVALID_ARGUMENT, "null context");
  if (width != 0 && width != 8 && width != 14 && width != 16) return fail(MWRT_ERR_INVALID_ARGUMENT, "chunk width: 0 (automatic), 8, 14 or 16");
  c->chunk_width = width;
  return MWRT_OK;
}

int mwrt_set_absorption_mode(mwrt_context* c, int mode) {
  if (!c || mode < 0 || mode > 2) return fail(MWRT_ERR_INVALID_ARGUMENT, "mode must be 0, 1 or 2");
  c->absorption_mode = mode;
  return MWRT_OK;
}

int mwrt_selftest_math(mwrt_context* c, int32_t n, const double* x, const double* y_pos,
                       double* exp_x, double* log_y, double* x_div_y, double* x_div1_y) {
  if (!c || n < 0 || !x || !y_pos || !exp_x || !log_y || !x_div_y || !x_div1_y)
    return fail(MWRT_ERR_INVALID_ARGUMENT, "null argument");
  if (n == 0) return MWRT_OK;
  HIP_TRY(hipSetDevice(c->device));
  const size_t b = sizeof(double) * (size_t)n;
  HIP_TRY(c->d_in.reserve(2 * b));
  HIP_TRY(c->d_out.reserve(4 * b));
  double* din = c->d_in.as<double>();
  double* dout = c->d_out.as<double>();
  HIP_TRY(hipMemcpyAsync(din, x, b, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemcpyAsync(din + n, y_pos, b, hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(k_selftest_math, dim3((n + 255) / 256), dim3(256), 0, c->stream, din, din + n, dout, dout + n,
                     dout + 2 * (size_t)n, dout + 3 * (size_t)n, n);
  HIP_TRY(hipGetLastError());
  double* dst[4] = {exp_x, log_y, x_div_y, x_div1_y};
  for (int k = 0; k < 4; ++k) HIP_TRY(hipMemcpyAsync(dst[k], dout + (size_t)k * n, b, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return MWRT_OK;
}

int mwrt_synchronize(mwrt_context* c, void* stream) {
  if (!c) return fail(MWRT_ERR_INVALID_ARGUMENT, "null context");
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipStreamSynchronize(resolve_stream(c, stream)));
  return MWRT_OK;
}

int mwrt_set_timing(mwrt_context* c, int enabled) {
  if (!c) return fail(MWRT_ERR_INVALID_ARGUMENT, "null context");
  HIP_TRY(hipSetDevice(c->device));
  if (enabled && c->ev0.empty()) {
    c->ev0.resize(TIMING_RING); c->ev1.resize(TIMING_RING);
    for (int i = 0; i < TIMING_RING; ++i) { HIP_TRY(hipEventCreate(&c->ev0[i])); HIP_TRY(hipEventCreate(&c->ev1[i])); }
  }
  c->timing = enabled != 0;
  c->ev_count = 0;
  return MWRT_OK;
}

int mwrt_timing_collect(mwrt_context* c, double* total_ms, int32_t* launches) {
  if (!c || !total_ms || !launches) return fail(MWRT_ERR_INVALID_ARGUMENT, "null argument");
  HIP_TRY(hipSetDevice(c->device));
  const long n = c->ev_count < TIMING_RING ? c->ev_count : TIMING_RING;
  double sum = 0.0;
  for (long i = 0; i < n; ++i) {
    HIP_TRY(hipEventSynchronize(c->ev1[i]));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, c->ev0[i], c->ev1[i]));
    sum += ms;
  }
  *total_ms = sum;
  *launches = (int32_t)n;
  c->ev_count = 0;
  return MWRT_OK;
}

int mwrt_last_kernel_ms(mwrt_context* c, double* ms_out) {
  if (!c || !ms_out) return fail(MWRT_ERR_INVALID_ARGUMENT, "null argument");
  if (c->ev_count < 1) return fail(MWRT_ERR_INVALID_ARGUMENT, "no timed launch pending");
  const long i = (c->ev_count - 1) % TIMING_RING;
  HIP_TRY(hipEventSynchronize(c->ev1[i]));
  float ms = 0.f;
  HIP_TRY(hipEventElapsedTime(&ms, c->ev0[i], c->ev1[i]));
  *ms_out = ms;
  return MWRT_OK;
}

}  // extern "C"
