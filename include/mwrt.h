/*
 * mwrt.h -- C ABI of the MI355X-native line-by-line microwave forward operator.
 *
 * Drop-in boundary for ONE path of apschera2023uzk/MWR_fast_forward_operators_and_LBLs:
 * the pyrtlib LBL call made by python_src/proc/PyRTlib_processing.py:123-127
 *
 *     rte = TbCloudRTE(z[::-1], p[::-1], t[::-1], rh[::-1], frqs, ang)   (:123)
 *     rte.init_absmdl(mdl)                                               (:124)
 *     rte.satellite = False                                              (:125)
 *     df = rte.execute(); tbs[i,:,k,j] = df["tbtotal"].values            (:126-127)
 *
 * The reference has no FFI layer of its own: these entry points are
 * what a ctypes stub placed behind that Python call surface binds.
 * Plain pointers and sizes only; the library never throws across the boundary; every
 * function returns an mwrt_status (0 = ok, <0 = error, text via mwrt_last_error()).
 *
 * Array conventions (all float64, C-contiguous, caller-owned):
 *   profiles   [nprof][nlev], level 0 = ground, level nlev-1 = top -- what TbCloudRTE sees
 *              after the wrapper's [::-1] (:123); z in km, p in hPa, T in K, rh as fraction
 *              (:109-114).
 *   frq_ghz    [nf]    (:87-88)
 *   elev_deg   [nang]  ELEVATION angles in (0, 180), 90 = zenith (:106, :37); the path is
 *              plane-parallel (air mass 1/sin elev), other values are MWRT_ERR_INVALID_ARGUMENT
 *   tb_out     [nprof][nang][nf]  == pyrtlib's DataFrame row order (angle-major) per profile
 *   valid_out  [nprof] 1 = ok; 0 = NaN in the inputs of that profile (the wrapper's
 *              check_for_nans, :71-79, :117-119: outputs stay NaN); 2 = negative absorption
 *              met in the layer integration (pyrtlib raises ValueError there); 3 = a ray of this
 *              profile was trapped (ducting) while ray tracing (that angle's outputs are NaN).
 *
 * NaN rules (check_for_nans is evaluated per (time, Crop, elevation), :101-119):
 *   NaN in a profile's z/p/T/rh   -> that profile's outputs NaN, valid = 0;
 *   NaN in elev_deg[k]            -> only the [:, k, :] rows are NaN (the reference skips that k
 *                                    alone, :106, :117); the other angles are computed and valid
 *                                    stays 1 -- valid describes the profile's own data;
 *   NaN in frq_ghz (or every elevation NaN) -> every output NaN, valid = 0 (frqs is shared by all
 *                                    calls of the wrapper, :87-88).
 *
 * Reproducibility: results are deterministic for a given call.  The kernels process the
 * frequencies in chunks (14 or 16 per workgroup) and choose per chunk between algebraically
 * equal forms of a line's denominator (polynomial in f^2 away from line centres, direct
 * detunings next to them), so the TB of one frequency may differ by <= 1e-8 K depending on
 * which other frequencies share its call.  Likewise the layer integration picks, per wave of
 * (frequency, elevation) pairs, between two algebraically equal forms of a layer's emission
 * (thin layers: series, no division), so a TB may differ by <= 1e-10 K depending on which
 * other elevations share its call.  Against the 1e-6 K parity bar both are invisible.
 *
 * Streams: the *_device entry points are asynchronous on `stream`:
 *   NULL               the context's own stream (hipStreamNonBlocking: NOT ordered with the
 *                      legacy default stream -- synchronise with mwrt_synchronize(ctx, NULL));
 *   MWRT_STREAM_LEGACY the caller's legacy default stream (hipStream_t 0, what
 *                      torch.cuda.current_stream().cuda_stream reads as 0);
 *   anything else      that hipStream_t.
 * Work is ordered on that stream like any kernel launch; consumers on other streams need an
 * event.  frq_ghz / elev_deg are host arrays: the first call with new values makes an immutable
 * device copy (a short host-side wait on the context's stream); later calls with the same values
 * neither allocate nor synchronise, which is what makes the call hipGraph-capturable after one
 * warm-up call.
 */
#ifndef MWRT_H
#define MWRT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MWRT_VERSION 301          /* 0.3.1: + mwrt_set_chunk_width; 0.3.0: layer-optical-depth two-kernel form (mwrt_layer_tau_*, mwrt_tb_from_layer_tau_device) */
#define MWRT_MAX_H2O_LINES 32
#define MWRT_MAX_O2_LINES 64
#define MWRT_MAX_X_LINES 64       /* lines of the extra trace species (ozone) */
#define MWRT_MAX_LEVELS 1024      /* one lane per level, one workgroup per profile */
#define MWRT_MAX_ANGLES 64
#define MWRT_STREAM_LEGACY ((void*)(intptr_t)-1)   /* `stream` value meaning hipStream_t 0 */

typedef enum {
  MWRT_OK = 0,
  MWRT_ERR_INVALID_ARGUMENT = -1,
  MWRT_ERR_NO_DEVICE = -2,
  MWRT_ERR_HIP = -3,
  MWRT_ERR_OUT_OF_MEMORY = -4,
  MWRT_ERR_UNSUPPORTED = -5
} mwrt_status;

/* Replaces pyrtlib's process-global model state set by init_absmdl(str) (:124): an explicit,
 * immutable table record.  Field-for-field image of spectroscopy.ModelTables (Python host). */
typedef struct mwrt_model_desc {
  int32_t n_h2o, n_o2;
  int32_t h2o_shift_mode;   /* 0 none (R98); 2 air+self shift with ln-T coefficients (R17+) */
  int32_t o2_mix_mode;      /* 0 first order on total pressure (R98/R17); 1 second order on den (R19+) */
  int32_t o2_line1_dens;    /* R98: 118.75-GHz width uses DENS (first-order mixing mode only) */
  int32_t n2_fdep;          /* absn2 frequency-dependence factor on/off */
  int32_t n2_ptot;          /* 1: N2 at total pressure (pre-2019, folded into the O2 routine) */
  int32_t liq_mode;         /* cloud liquid (opt-in): 0 Liebe 1991 / MPM93 double Debye; 1 Rosenkranz 2015 */
  double h2o_reftcon, h2o_reftline, h2o_cf, h2o_xcf, h2o_cs, h2o_xcs, h2o_pvap_div, h2o_den_coef;
  double o2_x, o2_wb300, o2_pvap_div, o2_wv_factor, o2_nonres, o2_coef;
  double n2_l, n2_m, n2_n;
  double t_cosmic, planck_h, boltzmann_k;
  double h2o_fl[MWRT_MAX_H2O_LINES], h2o_s1[MWRT_MAX_H2O_LINES], h2o_b2[MWRT_MAX_H2O_LINES];
  double h2o_w0[MWRT_MAX_H2O_LINES], h2o_x[MWRT_MAX_H2O_LINES];
  double h2o_w0s[MWRT_MAX_H2O_LINES], h2o_xs[MWRT_MAX_H2O_LINES];
  double h2o_sh[MWRT_MAX_H2O_LINES], h2o_xh[MWRT_MAX_H2O_LINES];
  double h2o_shs[MWRT_MAX_H2O_LINES], h2o_xhs[MWRT_MAX_H2O_LINES];
  double h2o_aair[MWRT_MAX_H2O_LINES], h2o_aself[MWRT_MAX_H2O_LINES];
  double h2o_w2[MWRT_MAX_H2O_LINES], h2o_xw2[MWRT_MAX_H2O_LINES];
  double h2o_w2s[MWRT_MAX_H2O_LINES], h2o_xw2s[MWRT_MAX_H2O_LINES];
  double h2o_d2[MWRT_MAX_H2O_LINES], h2o_d2s[MWRT_MAX_H2O_LINES];
  double o2_f[MWRT_MAX_O2_LINES], o2_s300[MWRT_MAX_O2_LINES], o2_be[MWRT_MAX_O2_LINES];
  double o2_w300[MWRT_MAX_O2_LINES], o2_y0[MWRT_MAX_O2_LINES], o2_y1[MWRT_MAX_O2_LINES];
  double o2_g0[MWRT_MAX_O2_LINES], o2_g1[MWRT_MAX_O2_LINES];
  double o2_dnu0[MWRT_MAX_O2_LINES], o2_dnu1[MWRT_MAX_O2_LINES];
  /* Extra trace species (ozone): pyrtlib's TbCloudRTE(..., o3n=...) adds O3AbsModel.o3_absorption to the dry
   * absorption [EXT; Rosenkranz o3abs].  The reference builds an O3 profile for the sibling model
   * (python_src/proc/ARMS_gb_processing.py:94-99) and leaves o3n at None on the LBL path, so this is opt-in
   * (mwrt_tb_options.o3n) and DATA-FREE here: the line list could not be restated offline; n_x = 0 means "no table"
   * and a call that passes o3n is refused.  tools/export_pyrtlib_tables.py dumps pyrtlib's list into these fields.
   *   alpha_x [Np/km] = x_coef * n [molecules m-3] * qvinv * ti^2.5 * sum_k S1_k exp(B_k (1 - ti)) (f/FL_k)^2
   *                     * [ w_k / ((f - FL_k)^2 + w_k^2) + w_k / ((f + FL_k)^2 + w_k^2) ],
   *   ti = x_reft / T,  qvinv = 1 - exp(-x_qvib_t / T)  (1 if x_qvib_t <= 0),
   *   w_k = 0.5346 wc + sqrt(0.2166 wc^2 + 0.6931 bd^2)   (Voigt half width, Olivero & Longbothum 1977),
   *   wc = W_k p ti^X_k  (p total, hPa),  bd = 4.3e-7 sqrt(T / x_mass) FL_k  (Doppler 1/e half width). */
  int32_t n_x;
  int32_t x_reserved;
  double x_reft, x_qvib_t, x_mass, x_coef;
  double x_fl[MWRT_MAX_X_LINES], x_s1[MWRT_MAX_X_LINES], x_b[MWRT_MAX_X_LINES];
  double x_w[MWRT_MAX_X_LINES], x_x[MWRT_MAX_X_LINES];
} mwrt_model_desc;

/* Optional by-products of execute() (the other DataFrame columns pyrtlib returns; the
 * reference reads only "tbtotal", :127).  Any pointer may be NULL. */
typedef struct mwrt_tb_extras {
  double* tbatm;    /* [nprof][nang][nf] */
  double* tmr;      /* [nprof][nang][nf] */
  double* tauwet;   /* [nprof][nang][nf] slant-path opacity, Np */
  double* taudry;   /* [nprof][nang][nf] */
  double* taulay;   /* [nprof][nf][nlev] ZENITH layer optical depth (wet+dry+ice+liquid), entry 0 = 0 */
  double* tauliq;   /* [nprof][nang][nf] cloud liquid opacity (0 unless mwrt_tb_options.denliq is given) */
  double* tauice;   /* [nprof][nang][nf] cloud ice opacity */
} mwrt_tb_extras;

/* Physics pyrtlib offers and the reference leaves at its defaults (TbCloudRTE(..., ray_tracing=False,
 * cloudy=False); the author prints rte.cloudy at old_processing.py:558-563).  STRICTLY OPT-IN: a NULL
 * options pointer, or all-zero options, is the reference's clear-sky plane-parallel path bit for bit.
 *   denliq / denice  cloud liquid / ice density profiles [nprof][nlev] in g m-3 (what init_cloudy takes);
 *                    the upstream producer stores kg/kg: python_src/preproc/derive_cloud_water.py:68-142,
 *                    preprocessing4all.py:811-812, :1199-1200 ("Level_Liquid", "Level_Ice").
 *                    RTEquation.cloudy_absorption + exponential_integration(zeroflg = False).
 *   o3n              ozone number density profiles (see mwrt_model_desc.n_x): opt-in, and refused without a line table.
 *   ray_tracing      != 0: spherical refracted slant paths (RTEquation.refractivity, Thayer 1974, and
 *                    RTEquation.ray_tracing, TBMODEL RAYTRAC) instead of dz / sin(elev) -- matters for the
 *                    4.2 ... 8.4 degree elevations of PyRTlib_processing.py:37.
 * On the *_device entry point denliq / denice are DEVICE pointers. */
typedef struct mwrt_tb_options {
  const double* denliq;
  const double* denice;
  int32_t ray_tracing;
  int32_t reserved0;
  const double* o3n;       /* ozone number density [nprof][nlev], molecules m-3 (pyrtlib's o3n), or NULL; needs a model
                              with n_x > 0 (else MWRT_ERR_UNSUPPORTED); added to the dry absorption of every level */
} mwrt_tb_options;

typedef struct mwrt_context mwrt_context;   /* one per (host thread, GPU): device, stream, workspace */
typedef struct mwrt_model mwrt_model;       /* device-resident copy of an mwrt_model_desc */

int mwrt_version(void);
/* sizeof(mwrt_model_desc) as compiled into the library (binding self-check). */
size_t mwrt_model_desc_size(void);
/* Number of usable HIP devices; 0 when there is no GPU / driver (never an error). */
int mwrt_device_count(void);
/* Thread-local text of the last failure on this thread ("" if none). */
const char* mwrt_last_error(void);

int mwrt_create(int device_id, mwrt_context** out);
int mwrt_destroy(mwrt_context* ctx);
int mwrt_model_create(mwrt_context* ctx, const mwrt_model_desc* desc, mwrt_model** out);
int mwrt_model_destroy(mwrt_context* ctx, mwrt_model* model);

/* TbCloudRTE(...).execute() for a batch of profiles; HOST buffers, synchronous.
 * Replaces the triple loop + 4 x execute() of PyRTlib_processing.py:99-151 for one model. */
int mwrt_tb_batch(mwrt_context* ctx, const mwrt_model* model,
                  int64_t nprof, int32_t nlev,
                  const double* z_km, const double* p_hpa, const double* t_k, const double* rh_frac,
                  int32_t nf, const double* frq_ghz,
                  int32_t nang, const double* elev_deg,
                  double* tb_out, uint8_t* valid_out, const mwrt_tb_extras* extras);

/* Same, on DEVICE buffers (profiles, tb_out, valid_out and the extras already in HBM),
 * asynchronous on `stream` (a hipStream_t; NULL / MWRT_STREAM_LEGACY: see "Streams" above).
 * frq_ghz and elev_deg stay small host arrays.  This is the entry bench.py times. */
int mwrt_tb_batch_device(mwrt_context* ctx, const mwrt_model* model,
                         int64_t nprof, int32_t nlev,
                         const double* d_z_km, const double* d_p_hpa, const double* d_t_k,
                         const double* d_rh_frac,
                         int32_t nf, const double* frq_ghz,
                         int32_t nang, const double* elev_deg,
                         double* d_tb_out, uint8_t* d_valid_out, const mwrt_tb_extras* d_extras,
                         void* stream);

/* mwrt_tb_batch / mwrt_tb_batch_device with the opt-in physics of mwrt_tb_options (NULL = none). */
int mwrt_tb_batch_opt(mwrt_context* ctx, const mwrt_model* model,
                      int64_t nprof, int32_t nlev,
                      const double* z_km, const double* p_hpa, const double* t_k, const double* rh_frac,
                      int32_t nf, const double* frq_ghz, int32_t nang, const double* elev_deg,
                      double* tb_out, uint8_t* valid_out, const mwrt_tb_extras* extras,
                      const mwrt_tb_options* options);
int mwrt_tb_batch_opt_device(mwrt_context* ctx, const mwrt_model* model,
                             int64_t nprof, int32_t nlev,
                             const double* d_z_km, const double* d_p_hpa, const double* d_t_k,
                             const double* d_rh_frac,
                             int32_t nf, const double* frq_ghz, int32_t nang, const double* elev_deg,
                             double* d_tb_out, uint8_t* d_valid_out, const mwrt_tb_extras* d_extras,
                             const mwrt_tb_options* d_options, void* stream);

/* Several absorption models over the SAME profiles in one launch (and one host->device copy): what
 * the wrapper does four times per profile, R20/R24/R17/R98 (PyRTlib_processing.py:121-151).
 * nmodels <= 8; tb_out [nmodels][nprof][nang][nf], valid_out [nmodels][nprof]. */
int mwrt_tb_batch_multi(mwrt_context* ctx, int32_t nmodels, const mwrt_model* const* models,
                        int64_t nprof, int32_t nlev,
                        const double* z_km, const double* p_hpa, const double* t_k, const double* rh_frac,
                        int32_t nf, const double* frq_ghz, int32_t nang, const double* elev_deg,
                        double* tb_out, uint8_t* valid_out);
int mwrt_tb_batch_multi_device(mwrt_context* ctx, int32_t nmodels, const mwrt_model* const* models,
                               int64_t nprof, int32_t nlev,
                               const double* d_z_km, const double* d_p_hpa, const double* d_t_k,
                               const double* d_rh_frac,
                               int32_t nf, const double* frq_ghz, int32_t nang, const double* elev_deg,
                               double* d_tb_out, uint8_t* d_valid_out, void* stream);

/* RTEquation.clearsky_absorption for a batch: awet, adry [nprof][nf][nlev] in Np/km
 * (exposes kernel K1 alone, for parity tests and the roofline measurement). HOST buffers. */
int mwrt_absorption_batch(mwrt_context* ctx, const mwrt_model* model,
                          int64_t nprof, int32_t nlev,
                          const double* p_hpa, const double* t_k, const double* rh_frac,
                          int32_t nf, const double* frq_ghz,
                          double* awet_out, double* adry_out);
int mwrt_absorption_batch_device(mwrt_context* ctx, const mwrt_model* model,
                                 int64_t nprof, int32_t nlev,
                                 const double* d_p_hpa, const double* d_t_k, const double* d_rh_frac,
                                 int32_t nf, const double* frq_ghz,
                                 double* d_awet_out, double* d_adry_out, void* stream);

/* Frequencies per workgroup of the fused TB kernel: 0 = automatic (default), or 8 / 14 / 16.  Automatic: 14 for channel lists
 * that are a multiple of 14 (the HATPRO list: one workgroup per profile, every per-(level, line) quantity computed once), 16
 * otherwise, whatever the batch size -- so a profile's results do not depend on the batch it arrives in, bit for bit.
 * 8 is the latency setting for small batches: two workgroups per 14-channel profile (MI355X, seven elevations: one profile
 * 58 instead of 75 us, 256 profiles 61 instead of 76, 512 profiles 77 instead of 83; slower from ~600 profiles up).  The width
 * changes the grouping of the line sums, so results move by ~1e-13 relative between widths. */
int mwrt_set_chunk_width(mwrt_context* ctx, int width);

/* How a fine spectral grid is evaluated: 0 = automatic (windowed when the frequency list qualifies: >= 128 strictly
 * increasing frequencies whose 128-frequency windows each span <= 6 GHz, <= 505 levels, LDS permitting), 1 = always every
 * line at every frequency, 2 = windowed or MWRT_ERR_UNSUPPORTED.  Windowed: the lines >= 4 GHz beyond a window are summed
 * at 16 Chebyshev nodes of the window and interpolated (error <= 1e-10 of the line sum), the others are evaluated
 * directly; results agree with mode 1 to ~1e-10 relative.
 * Governs mwrt_absorption_batch[_device] and mwrt_layer_tau_batch_device (modes 0, 1, 2 as above), and the TB entry
 * points' automatic fine-grid path (windowed K1 -> layer optical depth in HBM -> RTE kernel): mode 1 switches that
 * path off (one fused kernel, every line at every frequency); modes 0 and 2 both leave it automatic -- a TB call is
 * never refused for its frequency list. */
int mwrt_set_absorption_mode(mwrt_context* ctx, int mode);

/* The second half of execute() on its own: layer optical depths (exponential_integration, zeroflg = True) +
 * downwelling Planck-space RTE (planck, bright) from absorption coefficients ALREADY in HBM, laid out as
 * mwrt_absorption_batch_device writes them (awet, adry [nprof][nf][nlev], Np/km).  Together the two calls are
 * the K1 -> alpha -> K2 two-kernel form of the fine-grid configuration (BASELINE configs[4]: alpha is 3.6 GB
 * per GPU, written once and read once); it is also the entry for callers who bring their own absorption.
 * `model` supplies the RTE constants only (t_cosmic, h, k).  DEVICE buffers; valid as for mwrt_tb_batch
 * (0 also for a NaN absorption coefficient). */
int mwrt_tb_from_absorption_device(mwrt_context* ctx, const mwrt_model* model,
                                   int64_t nprof, int32_t nlev,
                                   const double* d_z_km, const double* d_t_k,
                                   int32_t nf, const double* frq_ghz, int32_t nang, const double* elev_deg,
                                   const double* d_awet, const double* d_adry,
                                   double* d_tb_out, uint8_t* d_valid_out, void* stream);

/* The fine-grid two-kernel form with the LAYER OPTICAL DEPTH as the hand-over (what the TB entry points run
 * automatically on window-eligible frequency lists; BASELINE configs[4]: 1.8 GB per GPU written once, read once):
 *   mwrt_layer_tau_batch_device   clearsky_absorption + exponential_integration(zeroflg = True) on wet and dry,
 *                                 summed [EXT, reached from PyRTlib_processing.py:126]: zenith layer optical depth
 *                                 d_tau_out [nprof][nlev][tau_pitch] in Np (entry [.][0][.] = 0; 8 B per (profile,
 *                                 level, frequency), FREQUENCY fastest), d_valid_out [nprof] as for mwrt_tb_batch
 *                                 (a profile flagged 0 or 2 has NaN rows).  tau_pitch = doubles between consecutive
 *                                 levels: a multiple of 16, >= mwrt_layer_tau_pitch(nf) (= nf rounded up to 16);
 *                                 columns [nf, tau_pitch) are scratch.  nlev <= 1009.
 *   mwrt_tb_from_layer_tau_device RTEquation.planck (from_sat = False) + bright [EXT]: TBs [nprof][nang][nf] from such
 *                                 an array (tau_pitch >= nf), T [nprof][nlev] and d_valid [nprof] (1 = integrate,
 *                                 anything else = that profile's TBs are NaN).  `model` supplies h, k, t_cosmic.
 * DEVICE buffers, asynchronous on `stream`. */
int mwrt_layer_tau_pitch(int32_t nf);
int mwrt_layer_tau_batch_device(mwrt_context* ctx, const mwrt_model* model,
                                int64_t nprof, int32_t nlev,
                                const double* d_z_km, const double* d_p_hpa, const double* d_t_k, const double* d_rh_frac,
                                int32_t nf, const double* frq_ghz,
                                double* d_tau_out, int32_t tau_pitch, uint8_t* d_valid_out, void* stream);
int mwrt_tb_from_layer_tau_device(mwrt_context* ctx, const mwrt_model* model,
                                  int64_t nprof, int32_t nlev,
                                  const double* d_tau, int32_t tau_pitch, const double* d_t_k,
                                  int32_t nf, const double* frq_ghz, int32_t nang, const double* elev_deg,
                                  const uint8_t* d_valid, double* d_tb_out, void* stream);

/* K-matrix of the operator in ONE call: the block the reference parses out of RTTOV-gb's K run
 * (python_src/proc/RTTOV_gb_processing.py:286-300, :418-432: dTB/dT, dTB/dq per level and channel).  Partial derivatives
 * of every TB with respect to the LBL inputs of each level, [nprof][nang][nf][nlev]:
 *   dtb_dt   K/K      d TB / d T_i       at fixed vapour pressure e_i, pressure and heights
 *   dtb_de   K/hPa    d TB / d e_i       (e = rh * es(T), Goff-Gratch) at fixed T_i
 *   dtb_ddz  K/km     d TB / d (z_i - z_{i-1})   thickness of the layer below level i (entry 0 = 0)
 * The absorption of a level is a local function of (p, T, e): its derivatives are central differences of five
 * evaluations per level (T +- 0.01 K, e (1 +- 1e-4)); the layer rule (exponential_integration), the Planck-space recursion
 * and bright() are differentiated analytically (adjoint), so the cost is ~6 forward runs whatever nlev -- not the
 * 3 nlev + 1 forward runs of a brute-force K-matrix.  Clear sky, plane-parallel.  HOST buffers, synchronous; tb_out as
 * mwrt_tb_batch; valid_out as there (0 / 2: that profile's outputs are NaN). */
int mwrt_tb_jacobian_batch(mwrt_context* ctx, const mwrt_model* model,
                           int64_t nprof, int32_t nlev,
                           const double* z_km, const double* p_hpa, const double* t_k, const double* rh_frac,
                           int32_t nf, const double* frq_ghz, int32_t nang, const double* elev_deg,
                           double* tb_out, double* dtb_dt, double* dtb_de, double* dtb_ddz, uint8_t* valid_out);

/* Diagnostic: evaluates the kernels' own exp / log / division helpers (fexp, flog, fdiv, fdiv1) on
 * host arrays x[n], y_pos[n] (y > 0), so their accuracy can be checked against libm. */
int mwrt_selftest_math(mwrt_context* ctx, int32_t n, const double* x, const double* y_pos,
                       double* exp_x, double* log_y, double* x_div_y, double* x_div1_y);

/* Block until everything queued on the context's stream (or `stream`) has finished. */
int mwrt_synchronize(mwrt_context* ctx, void* stream);

/* Kernel timing with HIP events: mwrt_set_timing(ctx, 1) brackets every kernel launch with a
 * hipEvent pair recorded on the launch stream (ring of 512 pairs, no host synchronisation).
 * mwrt_timing_collect sums the device time of the launches since the last collect/enable and
 * returns how many there were; mwrt_last_kernel_ms reads the most recent one. */
int mwrt_set_timing(mwrt_context* ctx, int enabled);
int mwrt_timing_collect(mwrt_context* ctx, double* total_ms, int32_t* launches);
int mwrt_last_kernel_ms(mwrt_context* ctx, double* ms_out);

#ifdef __cplusplus
}
#endif
#endif /* MWRT_H */
