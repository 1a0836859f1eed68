/*
 * wfk_oracle.c -- CPU restatement of the reference sampler.  TEST INFRASTRUCTURE.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library, and only as the checker / reported baseline.  The product path
 * (waveforms_amd + libwfk_hip.so) never links, imports or calls it.
 *
 * It restates, scalar and in plain C, what the reference does with NumPy ufuncs:
 *   calc_parts  waveforms/_waveform.pyx:155-169   np.searchsorted slicing, clip
 *   _calc       waveforms/_waveform.pyx:134-152   sum of amp * product of factor**n
 *   _apply      waveforms/_waveform.pyx:130-131   f(x - shift, *args)
 *   primitives  waveforms/_waveform.pyx:290-371
 *   _fill_parts waveforms/waveform.py:524-527     out[start:stop] += part
 *   WaveVStack.__call__ waveforms/waveform.py:679-693  offset, x - shift, members
 *   predistort(ker=) waveforms/distortion.py:329-337   zero-padded 'same' FIR
 * over the flattened program of include/wfk.h.  Pinned against golden vectors
 * produced by running the real reference (oracle/make_golden.py ->
 * tests/golden/) in tests/test_oracle_golden.py.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off: one rounding per
 * operation, like NumPy's elementwise passes).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "wfk.h"

/* ---- grid ------------------------------------------------------------- */
static double grid_t(const wfk_grid* g, int64_t i) {
  if (g->has_last && i == g->n - 1) return g->last;
  volatile double m = (double)(i + g->i0) * g->step; /* two roundings, no FMA; i0: index of the slice's first sample in the full grid */
  return m + g->t0;
}

typedef struct {
  const wfk_grid* g;
  const double* t;
  int64_t n;
} tsrc;

static double ts_at(const tsrc* s, int64_t i) {
  return s->t ? s->t[i] : grid_t(s->g, i);
}

/* np.searchsorted(x - tshift, b, side='left'): first i with x[i]-tshift >= b */
static int64_t search_left(const tsrc* s, double tshift, double b) {
  int64_t lo = 0, hi = s->n;
  while (lo < hi) {
    int64_t mid = lo + (hi - lo) / 2;
    double x = ts_at(s, mid);
    if (tshift != 0.0) x = x - tshift;
    if (x < b) lo = mid + 1; else hi = mid;
  }
  return lo;
}

/* ---- primitives (evaluate at u = t - shift) ---------------------------- */
static double lin_at(double start, double stop, int64_t m, int64_t k) {
  /* np.linspace(start, stop, m)[k] */
  if (m == 1) return start;
  if (k == m - 1) return stop;
  double step = (stop - start) / (double)(m - 1);
  volatile double a = (double)k * step;
  return a + start;
}

static double np_interp(double x, double start, double stop, const double* fp,
                        int64_t m) {
  /* numpy/core/src/multiarray/compiled_base.c arr_interp semantics */
  if (isnan(x)) return x;
  double x0 = lin_at(start, stop, m, 0), xl = lin_at(start, stop, m, m - 1);
  if (x > xl) return fp[m - 1];
  if (x < x0) return fp[0];
  int64_t lo = 0, hi = m; /* find j: xp[j] <= x < xp[j+1] */
  while (hi - lo > 1) {
    int64_t mid = lo + (hi - lo) / 2;
    if (x >= lin_at(start, stop, m, mid)) lo = mid; else hi = mid;
  }
  int64_t j = lo;
  if (j == m - 1) return fp[j];
  double xj = lin_at(start, stop, m, j), xj1 = lin_at(start, stop, m, j + 1);
  if (xj == x) return fp[j];
  double slope = (fp[j + 1] - fp[j]) / (xj1 - xj);
  double r = slope * (x - xj) + fp[j];
  if (isnan(r)) {
    r = slope * (x - xj1) + fp[j + 1];
    if (isnan(r) && fp[j] == fp[j + 1]) r = fp[j];
  }
  return r;
}

static double mollifier(double u, double r, int d) {
  double x = u / r;
  double q = fabs(x) * fabs(x) - 1.0;
  if (d == 0) return q >= 0 ? 0.0 : exp(1.0 / q + 1.0);
  /* p = poly1d([-2, 0]); for n in 1..d-1:
   *   p = (x^4 - 2x^2 + 1) p' + (-4n x^3 + (4n-2) x) p      (pyx:365-368) */
  double p[64] = {0}, nx[64];
  int deg = 1;
  p[1] = -2.0; /* p[k] = coefficient of x^k */
  for (int n = 1; n < d; ++n) {
    memset(nx, 0, sizeof nx);
    for (int k = 1; k <= deg; ++k) { /* derivative term k*p[k] x^(k-1) */
      double c = k * p[k];
      nx[k - 1 + 4] += c;
      nx[k - 1 + 2] += -2.0 * c;
      nx[k - 1] += c;
    }
    for (int k = 0; k <= deg; ++k) {
      nx[k + 3] += -4.0 * n * p[k];
      nx[k + 1] += (4.0 * n - 2.0) * p[k];
    }
    deg += 3;
    memcpy(p, nx, sizeof p);
  }
  double px = 0.0;
  for (int k = deg; k >= 0; --k) px = px * x + p[k];
  double env = q >= 0 ? 0.0 : exp(1.0 / q + 1.0) / pow(-q, 2.0 * d);
  return env * px / pow(r, (double)d);
}

static double d_gaussian(double u, double s, int n) {
  double x = u / s;
  double h0 = 1.0, h1 = 2.0 * x, h = n == 0 ? h0 : h1;
  for (int k = 1; k < n; ++k) { /* physicists' Hermite */
    h = 2.0 * x * h1 - 2.0 * k * h0;
    h0 = h1;
    h1 = h;
  }
  return pow(-1.0, n) / pow(s, n) * h * exp(-(x * x));
}

static double drag(double t, const double* a) {
  double t0 = a[0], freq = a[1], width = a[2], delta = a[3], bf = a[4],
         phase = a[5];
  double o = M_PI / width;
  double sx = sin(o * (t - t0));
  double Ox = sx * sx;
  double wt = 2 * M_PI * (freq + delta) * t - (2 * M_PI * delta * t0 + phase);
  if (isnan(bf) || bf - delta == 0) return Ox * cos(wt);
  double b = 1 / M_PI / 2 / (bf - delta);
  double Oy = -b * o * sin(2 * o * (t - t0));
  return Ox * cos(wt) + Oy * sin(wt);
}

/* multi-notch DRAG (ids 16/17) from the compiled argument block (include/wfk.h);
 * reference semantics: waveforms/multy_drag.py:31-155 */
static double mdrag(double u, const double* a) {
  double t0 = a[0], freq = a[1], width = a[2], delta = a[3], phase = a[4], plateau = a[5],
         half = a[6];
  int m = (int)a[7], dq = (int)a[8];
  const double* px = a + 9;
  const double* py = px + (m + 1);
  const double* cst = py + (m + 1);
  double o = M_PI / width, mid1 = t0 + width / 2, mid2 = t0 + plateau + width / 2;
  double ox, oy;
  int rising = u <= mid1, falling = u >= mid2;
  if (rising || falling) {
    double tau = falling ? u - t0 - plateau : u - t0;
    double s = sin(o * tau), c = cos(o * tau), sp = 1.0;
    double ex = 0, ey = 0, dx = 0, dy = 0;
    for (int p = 0; p <= m; ++p) {
      if (p & 1) { dx += px[p] * sp; dy += py[p] * sp; }
      else { ex += px[p] * sp; ey += py[p] * sp; }
      sp *= s;
    }
    ox = ex + c * dx;
    oy = ey + c * dy;
  } else {
    ox = cst[0];
    oy = cst[1];
  }
  if (dq >= 0) {
    const double* q = cst + 2;
    if (u >= mid1 - half && u <= mid1) {
      double tau = u - t0 - width / 2, hx = 0, hy = 0;
      for (int i = 0; i <= dq; ++i) { hx = hx * tau + q[i]; hy = hy * tau + q[dq + 1 + i]; }
      ox = hx; oy = hy;
    }
    if (u >= mid2 && u <= mid2 + half) {
      double tau = u - t0 - plateau - width / 2, hx = 0, hy = 0;
      const double* qr = q + 2 * (dq + 1);
      for (int i = 0; i <= dq; ++i) { hx = hx * tau + qr[i]; hy = hy * tau + qr[dq + 1 + i]; }
      ox = hx; oy = hy;
    }
  }
  double wt = 2 * M_PI * (freq + delta) * u - (2 * M_PI * delta * t0 + phase);
  return ox * cos(wt) + oy * sin(wt);
}

static int prim(int type, double u, const double* a, int64_t na, double* out) {
  switch (type) {
    case WFK_LINEAR: *out = u; return 0;
    case WFK_GAUSSIAN: { double x = u / a[0]; *out = exp(-(x * x)); return 0; }
    case WFK_ERF: *out = erf(u / a[0]); return 0;
    case WFK_COS: *out = cos(a[0] * u); return 0;
    case WFK_SINC: {
      double x = a[0] * u;
      double y = M_PI * (x == 0 ? 1.0e-20 : x);
      *out = sin(y) / y;
      return 0;
    }
    case WFK_EXP: *out = exp(a[0] * u); return 0;
    case WFK_INTERP: *out = np_interp(u, a[0], a[1], a + 2, na - 2); return 0;
    case WFK_LINEARCHIRP:
      *out = sin(a[3] + 2 * M_PI * ((a[1] - a[0]) / (2 * a[2]) * (u * u) + a[0] * u));
      return 0;
    case WFK_EXPONENTIALCHIRP:
      *out = sin(a[2] + 2 * M_PI * a[0] * (exp(a[1] * u) - 1) / a[1]);
      return 0;
    case WFK_HYPERBOLICCHIRP:
      *out = sin(a[2] + 2 * M_PI * a[0] / a[1] * log(1 + a[1] * u));
      return 0;
    case WFK_COSH: *out = cosh(a[0] * u); return 0;
    case WFK_SINH: *out = sinh(a[0] * u); return 0;
    case WFK_DRAG: *out = drag(u, a); return 0;
    case WFK_MOLLIFIER: *out = mollifier(u, a[0], (int)a[1]); return 0;
    case WFK_D_GAUSSIAN: *out = d_gaussian(u, a[0], (int)a[1]); return 0;
    case WFK_DRAG_SIN: case WFK_DRAG_SINX: *out = mdrag(u, a); return 0;
    default: return WFK_EUNSUP;
  }
}

static double np_power(double v, double n) {
  /* value ** n on a float64 array (pyx:146): NumPy's fast paths then pow() */
  if (n == 2.0) return v * v;
  if (n == -1.0) return 1.0 / v;
  if (n == 0.5) return sqrt(v);
  if (n == 0.0) return 1.0;
  return pow(v, n);
}

static double clipd(double v, double lo, double hi) {
  /* np.clip = minimum(maximum(v, lo), hi), NaN propagates */
  if (isnan(v)) return v;
  if (v < lo) v = lo;
  if (v > hi) v = hi;
  return v;
}

static int eval(const wfk_program* P, const tsrc* ts, double* re, double* im,
                int64_t stride) {
  for (int32_t c = 0; c < P->n_channels; ++c) {
    double* ore = re + (int64_t)c * stride;
    double* oim = im ? im + (int64_t)c * stride : NULL;
    double tshift = P->ch_tshift[c];
    double lo = P->ch_clip_lo[c], hi = P->ch_clip_hi[c];
    for (int64_t i = 0; i < ts->n; ++i) {
      ore[i] = P->ch_offset[c];
      if (oim) oim[i] = 0.0;
    }
    for (int32_t m = P->ch_member_off[c]; m < P->ch_member_off[c + 1]; ++m) {
      int64_t start = 0;
      for (int32_t p = P->mb_piece_off[m]; p < P->mb_piece_off[m + 1]; ++p) {
        int64_t stop = search_left(ts, tshift, P->pc_bound[p]);
        int32_t t0 = P->pc_term_off[p], t1 = P->pc_term_off[p + 1];
        if (start < stop && t1 > t0) {
          for (int64_t i = start; i < stop; ++i) {
            double x = ts_at(ts, i);
            if (tshift != 0.0) x = x - tshift;
            double sre = 0.0, sim = 0.0;
            for (int32_t k = t0; k < t1; ++k) {
              double prod = 1.0;
              for (int32_t f = P->tm_factor_off[k]; f < P->tm_factor_off[k + 1]; ++f) {
                double v;
                int rc = 0;
                if (P->fc_type[f] == WFK_SAMPLED) {
                  /* caller-evaluated factor: args = (i0, values...), value of sample i is values[i - i0]
                     (include/wfk.h; the reference evaluates function_lib[id] on x[start:stop], pyx:130-131) */
                  const double* a = P->pool + P->fc_arg_off[f];
                  int64_t m = P->fc_arg_off[f + 1] - P->fc_arg_off[f] - 1, j = i - (int64_t)a[0];
                  if (m < 1) return WFK_EINVAL;
                  j = j < 0 ? 0 : (j >= m ? m - 1 : j);
                  v = a[1 + j];
                } else {
                  rc = prim(P->fc_type[f], x - P->fc_shift[f],
                            P->pool + P->fc_arg_off[f],
                            P->fc_arg_off[f + 1] - P->fc_arg_off[f], &v);
                }
                if (rc) return rc;
                double n = P->fc_power[f];
                prod = prod * (n == 1.0 ? v : np_power(v, n));
              }
              sre = sre + P->tm_amp_re[k] * prod;
              sim = sim + P->tm_amp_im[k] * prod;
            }
            if (oim) {
              /* np.clip of a complex part: NumPy orders complex numbers lexicographically (real part,
                 then imaginary) against the real bounds lo + 0j, hi + 0j; NaN propagates */
              if (sre < lo || (sre == lo && sim < 0.0)) { sre = lo; sim = 0.0; }
              if (sre > hi || (sre == hi && sim > 0.0)) { sre = hi; sim = 0.0; }
              ore[i] += sre;
              oim[i] += sim;
            } else {
              ore[i] += clipd(sre, lo, hi);
            }
          }
        }
        start = stop;
      }
    }
  }
  return 0;
}

int wfk_oracle_eval_grid(const wfk_program* P, const wfk_grid* g, double* re,
                         double* im, int64_t stride) {
  tsrc ts = {g, NULL, g->n};
  return eval(P, &ts, re, im, stride);
}

int wfk_oracle_eval_tlist(const wfk_program* P, const double* t, int64_t n,
                          double* re, double* im, int64_t stride) {
  tsrc ts = {NULL, t, n};
  return eval(P, &ts, re, im, stride);
}

/* np.searchsorted(x - tshift, bounds) for one member; returns #bounds */
int wfk_oracle_member_index(const wfk_program* P, const wfk_grid* g,
                            const double* t, int64_t n, int32_t member,
                            int64_t* idx) {
  tsrc ts = {g, t, g ? g->n : n};
  int32_t c = 0;
  while (c + 1 < P->n_channels && P->ch_member_off[c + 1] <= member) ++c;
  int32_t k = 0;
  for (int32_t p = P->mb_piece_off[member]; p < P->mb_piece_off[member + 1]; ++p)
    idx[k++] = search_left(&ts, P->ch_tshift[c], P->pc_bound[p]);
  return k;
}

void wfk_oracle_grid(const wfk_grid* g, double* t) {
  for (int64_t i = 0; i < g->n; ++i) t[i] = grid_t(g, i);
}

/* predistort(sig, ker=ker): hstack(0_N, sig, 0_N) (*) ker, 'full', cropped at
 * [N + K//2, 2N + K//2)  ==  out[i] = sum_k ker[k] * sig[i + K//2 - k]
 * (waveforms/distortion.py:329-333; time-domain restatement of fftconvolve) */
void wfk_oracle_fir(const double* sig, int64_t n, const double* ker, int32_t K,
                    double* out) {
  int64_t h = K / 2;
  for (int64_t i = 0; i < n; ++i) {
    long double acc = 0.0L;
    for (int32_t k = 0; k < K; ++k) {
      int64_t j = i + h - k;
      if (j >= 0 && j < n) acc += (long double)ker[k] * sig[j];
    }
    out[i] = (double)acc;
  }
}
