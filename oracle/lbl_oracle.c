/*
 * TEST INFRASTRUCTURE ONLY -- plain-C restatement of the LBL hot path (second oracle + CPU baseline).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load the library
 * built from this file.  The product never links it.
 *
 * PARITY STATUS: parity unpinned (see oracle/lbl_oracle.py header): pyrtlib -- where the
 * reference's arithmetic for python_src/proc/PyRTlib_processing.py:123-127 lives -- is not in
 * /root/reference nor in this image, and the reference holds no golden vectors for the path.
 * This file restates the same published algorithm a second, independent time, scalar and
 * loop-for-loop the way pyrtlib's own Python loops run (angle -> frequency -> level -> line,
 * absorption re-evaluated for every angle, exactly the cost structure of the reference), so it
 * doubles as the honest single-core CPU baseline ("port").
 *
 * Routines follow pyrtlib [EXT]: RTEquation.vapor, clearsky_absorption, H2OAbsModel.h2o_absorption,
 * O2AbsModel.o2_absorption, N2AbsModel.n2_absorption, exponential_integration, planck, bright; and, for
 * the opt-in physics, LiqAbsModel.liquid_water_absorption, RTEquation.cloudy_absorption,
 * refractivity, ray_tracing.
 */
#define _GNU_SOURCE
#include <complex.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define LBL_MAX_H2O 32
#define LBL_MAX_O2 64
#define LBL_MAX_X 64

/* Same field order as the product's mwrt_model_desc so a test can hand over the same bytes;
 * declared independently on purpose (oracle/ includes nothing from the product). */
typedef struct lbl_tables {
  int32_t n_h2o, n_o2, h2o_shift_mode, o2_mix_mode, o2_line1_dens, n2_fdep, n2_ptot, liq_mode;
  double h2o_reftcon, h2o_reftline, h2o_cf, h2o_xcf, h2o_cs, h2o_xcs, h2o_pvap_div, h2o_den_coef;
  double o2_x, o2_wb300, o2_pvap_div, o2_wv_factor, o2_nonres, o2_coef;
  double n2_l, n2_m, n2_n;
  double t_cosmic, planck_h, boltzmann_k;
  double h2o_fl[LBL_MAX_H2O], h2o_s1[LBL_MAX_H2O], h2o_b2[LBL_MAX_H2O], h2o_w0[LBL_MAX_H2O], h2o_x[LBL_MAX_H2O],
      h2o_w0s[LBL_MAX_H2O], h2o_xs[LBL_MAX_H2O], h2o_sh[LBL_MAX_H2O], h2o_xh[LBL_MAX_H2O], h2o_shs[LBL_MAX_H2O],
      h2o_xhs[LBL_MAX_H2O], h2o_aair[LBL_MAX_H2O], h2o_aself[LBL_MAX_H2O], h2o_w2[LBL_MAX_H2O],
      h2o_xw2[LBL_MAX_H2O], h2o_w2s[LBL_MAX_H2O], h2o_xw2s[LBL_MAX_H2O], h2o_d2[LBL_MAX_H2O], h2o_d2s[LBL_MAX_H2O];
  double o2_f[LBL_MAX_O2], o2_s300[LBL_MAX_O2], o2_be[LBL_MAX_O2], o2_w300[LBL_MAX_O2], o2_y0[LBL_MAX_O2],
      o2_y1[LBL_MAX_O2], o2_g0[LBL_MAX_O2], o2_g1[LBL_MAX_O2], o2_dnu0[LBL_MAX_O2], o2_dnu1[LBL_MAX_O2];
  /* extra trace species (ozone) -- opt-in, data supplied by the caller */
  int32_t n_x, x_reserved;
  double x_reft, x_qvib_t, x_mass, x_coef;
  double x_fl[LBL_MAX_X], x_s1[LBL_MAX_X], x_b[LBL_MAX_X], x_w[LBL_MAX_X], x_x[LBL_MAX_X];
} lbl_tables;

size_t lbl_tables_size(void) { return sizeof(lbl_tables); }

/* RTEquation.vapor [EXT]: Goff-Gratch over water */
static void fill_nan(int nout, double* tbtotal, double* tbatm, double* tmr, double* tauwet, double* taudry) {
  for (int o = 0; o < nout; ++o) {
    tbtotal[o] = NAN;
    if (tbatm) tbatm[o] = NAN;
    if (tmr) tmr[o] = NAN;
    if (tauwet) tauwet[o] = NAN;
    if (taudry) taudry[o] = NAN;
  }
}

static double vapor_e(double tk, double rh) {
  double y = 373.16 / tk;
  double es = -7.90298 * (y - 1.0) + 5.02808 * log10(y) - 1.3816e-07 * (pow(10.0, 11.344 * (1.0 - (1.0 / y))) - 1.0) +
              0.0081328 * (pow(10.0, -3.49149 * (y - 1.0)) - 1.0) + log10(1013.246);
  return rh * pow(10.0, es);
}

/* Rosenkranz DCERROR [EXT] (Hui, Armstrong & Wray 1978) */
static double complex dcerror(double x, double y) {
  static const double a[7] = {122.607931777104326, 214.382388694706425, 181.928533092181549, 93.155580458138441,
                              30.180142196210589,  5.912626209773153,   0.564189583562615};
  static const double b[7] = {122.607931773875350, 352.730625110963558, 457.334478783897737, 348.703917719495792,
                              170.354001821091472, 53.992906912940207,  10.479857114260399};
  double complex zh = fabs(y) - x * I;
  double complex asum = (((((a[6] * zh + a[5]) * zh + a[4]) * zh + a[3]) * zh + a[2]) * zh + a[1]) * zh + a[0];
  double complex bsum = ((((((zh + b[6]) * zh + b[5]) * zh + b[4]) * zh + b[3]) * zh + b[2]) * zh + b[1]) * zh + b[0];
  double complex w = asum / bsum;
  if (y >= 0.0) return w;
  double complex z = x + y * I;
  return 2.0 * cexp(-(z * z)) - conj(w);
}

/* H2OAbsModel.h2o_absorption [EXT]; returns npp + ncpp ("ppm" units) for one level */
static double h2o_ppm(const lbl_tables* m, double pdrykpa, double vx, double ekpa, double frq) {
  const double db2np = log(10.0) * 0.1;
  const double rvap = (0.01 * 8.314510) / 18.01528;
  const double factor = 0.182 * frq;
  double t = 300.0 / vx;
  double p = (pdrykpa + ekpa) * 10.0;
  double rho = ekpa * 10.0 / (rvap * t);
  double f = frq;
  if (rho <= 0.0) return 0.0;
  double pvap = (rho * t) / m->h2o_pvap_div;
  double pda = p - pvap;
  double den = m->h2o_den_coef * rho;
  double ti = m->h2o_reftcon / t;
  double con = (m->h2o_cf * pda * pow(ti, m->h2o_xcf) + m->h2o_cs * pvap * pow(ti, m->h2o_xcs)) * pvap * f * f;
  ti = m->h2o_reftline / t;
  double tiln = log(ti);
  double ti2 = m->h2o_shift_mode == 0 ? pow(ti, 2.5) : exp(2.5 * tiln);
  double sum = 0.0;
  for (int i = 0; i < m->n_h2o; ++i) {
    double width0 = m->h2o_w0[i] * pda * pow(ti, m->h2o_x[i]) + m->h2o_w0s[i] * pvap * pow(ti, m->h2o_xs[i]);
    double width2 = 0.0;
    if (m->h2o_w2[i] > 0.0)
      width2 = m->h2o_w2[i] * pda * pow(ti, m->h2o_xw2[i]) + m->h2o_w2s[i] * pvap * pow(ti, m->h2o_xw2s[i]);
    double delta2 = m->h2o_d2[i] * pda + m->h2o_d2s[i] * pvap;
    double shift = 0.0;
    if (m->h2o_shift_mode != 0) {
      double shiftf = m->h2o_sh[i] * pda * (1.0 - m->h2o_aair[i] * tiln) * pow(ti, m->h2o_xh[i]);
      double shifts = m->h2o_shs[i] * pvap * (1.0 - m->h2o_aself[i] * tiln) * pow(ti, m->h2o_xhs[i]);
      shift = shiftf + shifts;
    }
    double wsq = width0 * width0;
    double s = m->h2o_s1[i] * ti2 * exp(m->h2o_b2[i] * (1.0 - ti));
    double df[2] = {f - m->h2o_fl[i] - shift, f + m->h2o_fl[i] + shift};
    double base = width0 / (562500.0 + wsq);
    double res = 0.0;
    for (int j = 0; j < 2; ++j) {
      if (j == 0 && width2 > 0.0 && fabs(df[j]) < 10.0 * width0) {
        double complex dn = width2 - delta2 * I;
        double complex xc = ((width0 - 1.5 * width2) + (df[j] + 1.5 * delta2) * I) / dn;
        double complex xrt = csqrt(xc);
        double complex pxw = 1.77245385090551603 * xrt * dcerror(-cimag(xrt), creal(xrt));
        double complex sd = 2.0 * (1.0 - pxw) / dn;
        res += creal(sd) - base;
      } else if (fabs(df[j]) < 750.0) {
        res += width0 / (df[j] * df[j] + wsq) - base;
      }
    }
    sum += s * res * (f / m->h2o_fl[i]) * (f / m->h2o_fl[i]);
  }
  double npp = (3.183e-05 * den * sum / db2np) / factor;
  double ncpp = (con / db2np) / factor;
  return npp + ncpp;
}

/* N2AbsModel.n2_absorption [EXT] */
static double n2_abs(const lbl_tables* m, double t, double p, double f) {
  double th = 300.0 / t;
  double fdepen = m->n2_fdep ? 0.5 + 0.5 / (1.0 + (f / 450.0) * (f / 450.0)) : 1.0;
  return m->n2_n * (m->n2_l * fdepen * p * p * f * f * pow(th, m->n2_m));
}

/* O2AbsModel.o2_absorption [EXT]; returns npp + ncpp */
static double o2_ppm(const lbl_tables* m, double pdrykpa, double vx, double ekpa, double frq) {
  const double db2np = log(10.0) * 0.1;
  const double rvap = (0.01 * 8.314510) / 18.01528;
  const double factor = 0.182 * frq;
  double temp = 300.0 / vx;
  double pres = (pdrykpa + ekpa) * 10.0;
  double vapden = (ekpa * 10.0) / (rvap * temp);
  double freq = frq;
  double th = 300.0 / temp;
  double th1 = th - 1.0;
  double b = pow(th, m->o2_x);
  double preswv = vapden * temp / m->o2_pvap_div;
  double presda = pres - preswv;
  double den = 0.001 * (presda * b + m->o2_wv_factor * preswv * th);
  double dens = 0.001 * (presda + m->o2_wv_factor * preswv) * th;
  double dfnr = m->o2_wb300 * den;
  double pe2 = den * den;
  double nonres = m->o2_nonres * freq * freq * dfnr / (th * (freq * freq + dfnr * dfnr));
  double sum = nonres;
  for (int k = 0; k < m->n_o2; ++k) {
    double sf1, sf2, str;
    if (m->o2_mix_mode == 0) {
      double df = m->o2_w300[k] * ((k == 0 && m->o2_line1_dens) ? dens : den);
      double y = 0.001 * pres * b * (m->o2_y0[k] + m->o2_y1[k] * th1);
      str = m->o2_s300[k] * exp(-m->o2_be[k] * th1);
      sf1 = (df + (freq - m->o2_f[k]) * y) / ((freq - m->o2_f[k]) * (freq - m->o2_f[k]) + df * df);
      sf2 = (df - (freq + m->o2_f[k]) * y) / ((freq + m->o2_f[k]) * (freq + m->o2_f[k]) + df * df);
    } else {
      double y = den * (m->o2_y0[k] + m->o2_y1[k] * th1);
      double dnu = pe2 * (m->o2_dnu0[k] + m->o2_dnu1[k] * th1);
      double gfac = 1.0 + pe2 * (m->o2_g0[k] + m->o2_g1[k] * th1);
      double df = m->o2_w300[k] * den;
      str = m->o2_s300[k] * exp(-m->o2_be[k] * th1);
      double del1 = freq - m->o2_f[k] - dnu;
      double del2 = freq + m->o2_f[k] + dnu;
      double d1 = del1 * del1 + df * df;
      double d2 = del2 * del2 + df * df;
      sf1 = (df * gfac + del1 * y) / d1;
      sf2 = (df * gfac - del2 * y) / d2;
    }
    sum += str * (sf1 + sf2) * (freq / m->o2_f[k]) * (freq / m->o2_f[k]);
  }
  double o2abs = m->o2_coef * sum * presda * th * th * th;
  if (o2abs < 0.0) o2abs = 0.0;
  double ncpp = m->o2_coef * nonres * presda * th * th * th;
  double npp = (o2abs / db2np) / factor - (ncpp / db2np) / factor;
  ncpp = (ncpp / db2np) / factor;
  if (m->n2_ptot) ncpp += (n2_abs(m, temp, pres, freq) / db2np) / factor;
  return npp + ncpp;
}

/* RTEquation.clearsky_absorption [EXT] for one level */
static void clearsky_abs(const lbl_tables* m, double p, double tk, double e, double frq, double* awet, double* adry) {
  const double factor = 0.182 * frq;
  const double db2np = log(10.0) * 0.1;
  double v = 300.0 / tk;
  double ekpa = e / 10.0;
  double pdrykpa = p / 10.0 - ekpa;
  *awet = (factor * h2o_ppm(m, pdrykpa, v, ekpa, frq)) * db2np;
  double ao2 = (factor * o2_ppm(m, pdrykpa, v, ekpa, frq)) * db2np;
  double an2 = m->n2_ptot ? 0.0 : n2_abs(m, tk, pdrykpa * 10.0, frq);
  *adry = ao2 + an2;
}

/* RTEquation.exponential_integration [EXT]; returns -1 on negative input */
static int expint(int zeroflg, const double* x, const double* ds, int nl, double* xds, double* sxds) {
  double s = 0.0;
  xds[0] = 0.0;
  for (int i = 1; i < nl; ++i) {
    double xlayer;
    if (x[i - 1] < 0.0 || x[i] < 0.0) return -1;
    else if (fabs(x[i] - x[i - 1]) < 1e-09) xlayer = x[i];
    else if (x[i - 1] == 0.0 || x[i] == 0.0) xlayer = zeroflg ? (x[i] + x[i - 1]) * 0.5 : 0.0;
    else xlayer = (x[i] - x[i - 1]) / log(x[i] / x[i - 1]);
    xds[i] = xlayer * ds[i];
    s += xds[i];
  }
  *sxds = s;
  return 0;
}

/* LiqAbsModel.liquid_water_absorption [EXT] (Rosenkranz ABLIQ), Np/km; liq_mode as in lbl_oracle.py */
static double abliq(const lbl_tables* m, double water, double freq, double temp) {
  if (water <= 0.0) return 0.0;
  double complex eps;
  if (m->liq_mode == 0) {
    double theta1 = 1.0 - 300.0 / temp;
    double eps0 = 77.66 - 103.3 * theta1;
    double eps1 = 0.0671 * eps0;
    double eps2 = 3.52;
    double fp = (316.0 * theta1 + 146.4) * theta1 + 20.2;
    double fs = 39.8 * fp;
    eps = (eps0 - eps1) / (1.0 + I * (freq / fp)) + (eps1 - eps2) / (1.0 + I * (freq / fs)) + eps2;
  } else {
    double tc = temp - 273.15;
    double complex z = I * freq;
    double theta = 300.0 / temp;
    double eps0 = -43.7527 * pow(theta, 0.05) + 299.504 * pow(theta, 1.47) - 399.364 * pow(theta, 2.11) +
                  221.327 * pow(theta, 2.31);
    double delta = 80.69715 * exp(-tc / 226.45);
    double sd = 1164.023 * exp(-651.4728 / (tc + 133.07));
    double complex kappa = -delta * z / (sd + z);
    delta = 4.008724 * exp(-tc / 103.05);
    double hdelta = delta / 2.0;
    double f1 = 10.46012 + 0.1454962 * tc + 0.063267156 * tc * tc + 0.00093786645 * tc * tc * tc;
    double complex z1 = (-0.75 + I * 1.0) * f1;
    double complex z2 = -4500.0 + I * 2000.0;
    double complex cnorm = clog(z2 / z1);
    double complex chip = (hdelta * clog((z - z2) / (z - z1))) / cnorm;
    double complex chij = (hdelta * clog((z - conj(z2)) / (z - conj(z1)))) / conj(cnorm);
    double complex dchi = chip + chij - delta;
    eps = eps0 + (kappa + dchi);
  }
  double complex re = (eps - 1.0) / (eps + 2.0);
  return -0.06286 * cimag(re) * freq * water;
}

/* RTEquation.refractivity [EXT] (Thayer 1974): refractive index */
static double refindex(double p, double tk, double e) {
  double pa = p - e, tc = tk - 273.16, tk2 = tk * tk, tc2 = tc * tc;
  double rza = 1.0 + pa * (5.79e-07 * (1.0 + 0.52 / tk) - (0.00094611 * tc) / tk2);
  double rzw = 1.0 + 1650.0 * (e / (tk * tk2)) * (1.0 - 0.01317 * tc + 0.000175 * tc2 + 1.44e-06 * (tc2 * tc));
  double wetn = (64.79 * (e / tk) + 377600.0 * (e / tk2)) * rzw;
  double dryn = 77.6036 * (pa / tk) * rza;
  return 1.0 + (dryn + wetn) * 1e-06;
}

/* RTEquation.ray_tracing [EXT] (TBMODEL RAYTRAC); returns -1 when the ray is trapped (ducting) */
static int raytrace(int nl, const double* z, const double* n, double angle, double z0, double* ds) {
  const double re = 6370.949;
  ds[0] = 0.0;
  if ((angle >= 89 && angle <= 91) || (angle >= -91 && angle <= -89)) {
    for (int i = 1; i < nl; ++i) ds[i] = z[i] - z[i - 1];
    return 0;
  }
  double theta0 = angle * (M_PI / 180.0);
  double rs = re + z[0] + z0;
  double costh0 = cos(theta0), sina = sin(theta0 * 0.5);
  double a0 = 2.0 * (sina * sina);
  double phil = 0.0, taul = 0.0, rl = re + z[0] + z0, tanthl = tan(theta0);
  for (int i = 1; i < nl; ++i) {
    double r = re + z[i] + z0, refbar;
    if (n[i] == n[i - 1] || n[i] == 1.0 || n[i - 1] == 1.0) refbar = (n[i] + n[i - 1]) * 0.5;
    else refbar = 1.0 + (n[i - 1] - n[i]) / (log((n[i - 1] - 1.0) / (n[i] - 1.0)));
    double argdth = z[i] / rs - ((n[0] - n[i]) * costh0 / n[i]);
    double argth = 0.5 * (a0 + argdth) / r;
    if (argth <= 0) return -1;
    double sint = sqrt(r * argth);
    double theta = 2.0 * asin(sint), dtheta;
    if ((theta - 2.0 * theta0) <= 0.0) {
      double dendth = 2.0 * (sint + sina) * cos((theta + theta0) * 0.25);
      double sind4 = (0.5 * argdth - z[i] * argth) / dendth;
      dtheta = 4.0 * asin(sind4);
      theta = theta0 + dtheta;
    } else {
      dtheta = theta - theta0;
    }
    double tanth = tan(theta);
    double cthbar = ((1.0 / tanth) + (1.0 / tanthl)) * 0.5;
    double dtau = cthbar * (n[i - 1] - n[i]) / refbar;
    double tau = taul + dtau;
    double phi = dtheta + tau;
    double sh = sin((phi - phil) * 0.5);
    ds[i] = sqrt((z[i] - z[i - 1]) * (z[i] - z[i - 1]) + 4.0 * r * rl * (sh * sh));
    if (dtau != 0.0) {
      double dtaua = fabs(tau - taul);
      ds[i] = ds[i] * (dtaua / (2.0 * sin(dtaua * 0.5)));
    }
    phil = phi; taul = tau; rl = r; tanthl = tanth;
  }
  return 0;
}


/* O3AbsModel.o3_absorption [EXT, recalled from Rosenkranz's o3abs -- UNVERIFIED; the line list is data the caller
 * supplies]: Np/km at one level for number density o3n [molecules m-3]. */
static double o3_abs(const lbl_tables* m, double tk, double p, double frq, double o3n) {
  const double ti = m->x_reft / tk;
  const double qvinv = m->x_qvib_t > 0 ? 1.0 - exp(-m->x_qvib_t / tk) : 1.0;
  double sum = 0.0;
  for (int k = 0; k < m->n_x; ++k) {
    double widthc = m->x_w[k] * p * pow(ti, m->x_x[k]);
    double betad = 4.3e-07 * sqrt(tk / m->x_mass) * m->x_fl[k];
    double width = 0.5346 * widthc + sqrt(0.2166 * widthc * widthc + 0.6931 * betad * betad);
    double s = m->x_s1[k] * exp(m->x_b[k] * (1.0 - ti));
    double df1 = frq - m->x_fl[k], df2 = frq + m->x_fl[k];
    double shape = width / (df1 * df1 + width * width) + width / (df2 * df2 + width * width);
    double r = frq / m->x_fl[k];
    sum = sum + s * shape * (r * r);
  }
  return m->x_coef * o3n * qvinv * pow(ti, 2.5) * sum;
}

/*
 * TbCloudRTE(z,p,t,rh,frq,angles) + init_absmdl + satellite=False + execute() for ONE profile
 * (PyRTlib_processing.py:123-126).  Outputs [nang][nf] each (pyrtlib DataFrame row order).
 * Opt-in physics: denliq / denice [g m-3 per level, NULL = clear sky] (cloudy=True + init_cloudy),
 * ray_tracing != 0 (ray_tracing=True).
 * Returns 0 ok, 1 NaN input (outputs NaN), 2 negative absorption (pyrtlib raises ValueError), 3 ducting.
 * A NaN elevation blanks its own rows only (the wrapper's per-k check, :106, :117).
 */
static int tb_profile_core(const lbl_tables* m, int nl, const double* z, const double* p, const double* tk, const double* rh,
                           int nf, const double* frq, int nang, const double* ang,
                           const double* denliq, const double* denice, int ray_tracing, const double* o3n,
                           double* tbtotal, double* tbatm, double* tmr, double* tauwet, double* taudry,
                           double* tauliq, double* tauice) {
  const int nout = nf * nang;
  int bad = 0, allnan = 1;
  if (o3n) for (int i = 0; i < nl; ++i) bad |= isnan(o3n[i]);
  for (int i = 0; i < nl; ++i) bad |= isnan(z[i]) || isnan(p[i]) || isnan(tk[i]) || isnan(rh[i]);
  if (denliq) for (int i = 0; i < nl; ++i) bad |= isnan(denliq[i]);
  if (denice) for (int i = 0; i < nl; ++i) bad |= isnan(denice[i]);
  for (int j = 0; j < nf; ++j) bad |= isnan(frq[j]);
  for (int k = 0; k < nang; ++k) allnan &= isnan(ang[k]);
  if (bad || allnan) {
    fill_nan(nout, tbtotal, tbatm, tmr, tauwet, taudry);
    if (tauliq) for (int o = 0; o < nout; ++o) tauliq[o] = NAN;
    if (tauice) for (int o = 0; o < nout; ++o) tauice[o] = NAN;
    return 1;
  }
  const int cloudy = denliq != NULL || denice != NULL;
  double* buf = (double*)malloc(sizeof(double) * (size_t)nl * 14);
  double *e = buf, *zz = buf + nl, *ds = buf + 2 * nl, *awet = buf + 3 * nl, *adry = buf + 4 * nl,
         *pw = buf + 5 * nl, *pd = buf + 6 * nl, *boft = buf + 7 * nl, *tauprof = buf + 8 * nl,
         *aliq = buf + 9 * nl, *aice = buf + 10 * nl, *pl = buf + 11 * nl, *pi = buf + 12 * nl, *nref = buf + 13 * nl;
  for (int i = 0; i < nl; ++i) {
    e[i] = vapor_e(tk[i], rh[i]); zz[i] = z[i] - z[0]; nref[i] = refindex(p[i], tk[i], e[i]);
    pl[i] = 0.0; pi[i] = 0.0;
  }
  int rc = 0;
  for (int k = 0; k < nang && rc == 0; ++k) {
    if (isnan(ang[k])) {
      for (int j = 0; j < nf; ++j) {
        const int o = k * nf + j;
        tbtotal[o] = NAN;
        if (tbatm) tbatm[o] = NAN;
        if (tmr) tmr[o] = NAN;
        if (tauwet) tauwet[o] = NAN;
        if (taudry) taudry[o] = NAN;
        if (tauliq) tauliq[o] = NAN;
        if (tauice) tauice[o] = NAN;
      }
      continue;
    }
    if (ray_tracing) {
      if (raytrace(nl, zz, nref, ang[k], z[0], ds)) { rc = 3; break; }
    } else {
      double amass = 1.0 / sin(ang[k] * M_PI / 180.0);
      ds[0] = 0.0;
      for (int i = 1; i < nl; ++i) ds[i] = (zz[i] - zz[i - 1]) * amass;
    }
    for (int j = 0; j < nf; ++j) {
      for (int i = 0; i < nl; ++i) {
        clearsky_abs(m, p[i], tk[i], e[i], frq[j], &awet[i], &adry[i]);
        if (o3n) adry[i] = adry[i] + o3_abs(m, tk[i], p[i], frq[j], o3n[i]);    /* clearsky_absorption(..., o3n) [EXT] */
      }
      double sw, sd, sl = 0.0, si = 0.0;
      if (expint(1, awet, ds, nl, pw, &sw) || expint(1, adry, ds, nl, pd, &sd)) { rc = 2; break; }
      if (cloudy) {
        /* RTEquation.cloudy_absorption [EXT] */
        const double wave = (299792458.0 * 100.0) / (frq[j] * 1e9);
        const double db2np = log(10.0) * 0.1;
        for (int i = 0; i < nl; ++i) {
          aliq[i] = (denliq && denliq[i] > 0) ? abliq(m, denliq[i], frq[j], tk[i]) : 0.0;
          aice[i] = 0.0;
          if (denice && denice[i] > 0) { aice[i] = (8.18645 / wave) * denice[i] * 0.000959553; aice[i] = aice[i] * db2np; }
        }
        if (expint(0, aliq, ds, nl, pl, &sl) || expint(0, aice, ds, nl, pi, &si)) { rc = 2; break; }
      }
      /* RTEquation.planck [EXT], ground-based branch */
      double hvk = (frq[j] * 1e9) * m->planck_h / m->boltzmann_k;
      double boftatm = 0.0;
      tauprof[0] = 0.0;
      boft[0] = 1.0 / (exp(hvk / tk[0]) - 1.0);
      for (int i = 1; i < nl; ++i) {
        double taulay = pw[i] + pd[i] + pi[i] + pl[i];
        boft[i] = 1.0 / (exp(hvk / tk[i]) - 1.0);
        double boftlay = (boft[i - 1] + boft[i] * exp(-taulay)) / (1.0 + exp(-taulay));
        double batmlay = boftlay * exp(-tauprof[i - 1]) * (1.0 - exp(-taulay));
        boftatm += batmlay;
        tauprof[i] = tauprof[i - 1] + taulay;
      }
      double boftotl, boftmr;
      if (tauprof[nl - 1] < 125.0) {
        double boftbg = 1.0 / (exp(hvk / m->t_cosmic) - 1.0);
        boftotl = boftbg * exp(-tauprof[nl - 1]) + boftatm;
        boftmr = boftatm / (1.0 - exp(-tauprof[nl - 1]));
      } else {
        boftotl = boftatm; boftmr = boftatm;
      }
      const int o = k * nf + j;
      tbtotal[o] = hvk / log(1.0 + (1.0 / boftotl));     /* RTEquation.bright [EXT] */
      if (tbatm) tbatm[o] = hvk / log(1.0 + (1.0 / boftatm));
      if (tmr) tmr[o] = hvk / log(1.0 + (1.0 / boftmr));
      if (tauwet) tauwet[o] = sw;
      if (taudry) taudry[o] = sd;
      if (tauliq) tauliq[o] = sl;
      if (tauice) tauice[o] = si;
    }
  }
  free(buf);
  if (rc >= 2) {
    fill_nan(nout, tbtotal, tbatm, tmr, tauwet, taudry);
    if (tauliq) for (int o = 0; o < nout; ++o) tauliq[o] = NAN;
    if (tauice) for (int o = 0; o < nout; ++o) tauice[o] = NAN;
  }
  return rc;
}

int lbl_tb_profile_opt(const lbl_tables* m, int nl, const double* z, const double* p, const double* tk, const double* rh,
                       int nf, const double* frq, int nang, const double* ang,
                       const double* denliq, const double* denice, int ray_tracing,
                       double* tbtotal, double* tbatm, double* tmr, double* tauwet, double* taudry,
                       double* tauliq, double* tauice) {
  return tb_profile_core(m, nl, z, p, tk, rh, nf, frq, nang, ang, denliq, denice, ray_tracing, NULL, tbtotal, tbatm, tmr,
                         tauwet, taudry, tauliq, tauice);
}

/* ... with an ozone number-density profile o3n [molecules m-3] (TbCloudRTE(..., o3n=...)); needs m->n_x > 0 */
int lbl_tb_profile_o3(const lbl_tables* m, int nl, const double* z, const double* p, const double* tk, const double* rh,
                      int nf, const double* frq, int nang, const double* ang,
                      const double* denliq, const double* denice, int ray_tracing, const double* o3n,
                      double* tbtotal, double* tbatm, double* tmr, double* tauwet, double* taudry,
                      double* tauliq, double* tauice) {
  return tb_profile_core(m, nl, z, p, tk, rh, nf, frq, nang, ang, denliq, denice, ray_tracing, o3n, tbtotal, tbatm, tmr,
                         tauwet, taudry, tauliq, tauice);
}

int lbl_tb_profile(const lbl_tables* m, int nl, const double* z, const double* p, const double* tk, const double* rh,
                   int nf, const double* frq, int nang, const double* ang,
                   double* tbtotal, double* tbatm, double* tmr, double* tauwet, double* taudry) {
  return lbl_tb_profile_opt(m, nl, z, p, tk, rh, nf, frq, nang, ang, NULL, NULL, 0, tbtotal, tbatm, tmr, tauwet, taudry,
                            NULL, NULL);
}

/* Batch driver: profiles [nprof][nl]; tb [nprof][nang][nf]; OpenMP over profiles when built with it. */
int lbl_tb_batch(const lbl_tables* m, long nprof, int nl, const double* z, const double* p, const double* tk,
                 const double* rh, int nf, const double* frq, int nang, const double* ang,
                 double* tbtotal, unsigned char* valid, int nthreads) {
  (void)nthreads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 4) num_threads(nthreads > 0 ? nthreads : 1)
#endif
  for (long i = 0; i < nprof; ++i) {
    int rc = lbl_tb_profile(m, nl, z + i * nl, p + i * nl, tk + i * nl, rh + i * nl, nf, frq, nang, ang,
                            tbtotal + i * (long)nang * nf, NULL, NULL, NULL, NULL);
    valid[i] = rc == 0 ? 1 : (rc == 1 ? 0 : 2);
  }
  return 0;
}

/* awet/adry [nf][nl] for one profile */
void lbl_absorption_profile(const lbl_tables* m, int nl, const double* p, const double* tk, const double* rh,
                            int nf, const double* frq, double* awet, double* adry) {
  for (int j = 0; j < nf; ++j)
    for (int i = 0; i < nl; ++i)
      clearsky_abs(m, p[i], tk[i], vapor_e(tk[i], rh[i]), frq[j], &awet[j * nl + i], &adry[j * nl + i]);
}
