// wfk_short_dev.h -- device code of the short-piece tier shared by its two kernels: the sampler
// (wfk_short.hip) and the sampler fused into the FIR transform at AWG rates (wfk_fir_sampled.hip,
// fir_short).  One op record (WFK_SH_* in wfk_internal.h) evaluated over a lane's run of <= R
// contiguous samples of one piece: exact straight-line seeds, then recurrences that step by dt.
// Reference arithmetic: calc_parts / _calc (waveforms/_waveform.pyx:134-169).
#pragma once
#include <hip/hip_runtime.h>

#include <utility>

#include "wfk_internal.h"

namespace shdev {

template <int... K, typename F>
__device__ __forceinline__ void sh_for_impl(std::integer_sequence<int, K...>, F&& f) {
  (f(std::integral_constant<int, K>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void sh_for(F&& f) {
  sh_for_impl(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f));
}
#define SH_EACH(N, k) sh_for<N>([&](auto k##_) __attribute__((always_inline)) { constexpr int k = decltype(k##_)::value;
#define SH_END });

// polynomial coefficient pinned to an SGPR pair at its use (see wfk_fir_sampled.hip: left alone the
// compiler hoists all of them into VGPRs that stay live across the sample loops)
__device__ __forceinline__ double kc(double v) {
  asm volatile("" : "+s"(v));
  return v;
}

// sin(pi r), cos(pi r) for |r| <= 1/2: fold to z in [0, 1/4], Taylor in t = pi z (truncation < 5e-18)
__device__ __forceinline__ void sincospi_small(double r, double* sn, double* cs) {
  const double a = fabs(r);
  const bool swap = a > 0.25;
  const double z = swap ? 0.5 - a : a;                    // exact
  const double t = z * 3.141592653589793116 + z * 1.2246467991473532e-16;
  const double t2 = t * t;
  double ps = kc(-2.8114572543455206e-15);                    // -1/17!
  ps = fma(ps, t2, kc(7.6471637318198164e-13));
  ps = fma(ps, t2, kc(-1.6059043836821613e-10));
  ps = fma(ps, t2, kc(2.5052108385441720e-08));
  ps = fma(ps, t2, kc(-2.7557319223985893e-06));
  ps = fma(ps, t2, kc(1.9841269841269841e-04));
  ps = fma(ps, t2, kc(-8.3333333333333332e-03));
  ps = fma(ps, t2, kc(1.6666666666666666e-01));
  const double s = fma(-t * t2, ps, t);
  double pc = kc(1.5619206968586226e-16);                     //  1/18!
  pc = fma(pc, t2, kc(-4.7794773323873853e-14));
  pc = fma(pc, t2, kc(1.1470745597729725e-11));
  pc = fma(pc, t2, kc(-2.0876756987868099e-09));
  pc = fma(pc, t2, kc(2.7557319223985888e-07));
  pc = fma(pc, t2, kc(-2.4801587301587302e-05));
  pc = fma(pc, t2, kc(1.3888888888888889e-03));
  pc = fma(pc, t2, kc(-4.1666666666666664e-02));
  pc = fma(pc, t2, kc(0.5));
  const double c = fma(-t2, pc, 1.0);
  const double ss = swap ? c : s, cc = swap ? s : c;
  *sn = r < 0.0 ? -ss : ss;
  *cs = cc;
}

// exp(x), |x| < 700: Cody-Waite reduction, Taylor to r^14 on |r| <= ln2 / 2, ldexp
__device__ __forceinline__ double exp_small(double x) {
#pragma clang fp contract(off)   // the reduction's fma()s are explicit
  const double n = rint(x * 1.4426950408889634);
  double r = fma(-n, 6.93147180369123816490e-01, x);
  r = fma(-n, 1.90821492927058770002e-10, r);
  double p = kc(1.1470745597729725e-11);                      // 1/14!
  p = fma(p, r, kc(1.6059043836821613e-10));
  p = fma(p, r, kc(2.0876756987868099e-09));
  p = fma(p, r, kc(2.5052108385441720e-08));
  p = fma(p, r, kc(2.7557319223985888e-07));
  p = fma(p, r, kc(2.7557319223985893e-06));
  p = fma(p, r, kc(2.4801587301587302e-05));
  p = fma(p, r, kc(1.9841269841269841e-04));
  p = fma(p, r, kc(1.3888888888888889e-03));
  p = fma(p, r, kc(8.3333333333333332e-03));
  p = fma(p, r, kc(4.1666666666666664e-02));
  p = fma(p, r, kc(1.6666666666666666e-01));
  p = fma(p, r, kc(0.5));
  p = fma(p, r, kc(1.0));
  p = fma(p, r, kc(1.0));
  return ldexp(p, (int)n);
}

__device__ __forceinline__ double clip_np(double v, double lo, double hi) {
  v = v < lo ? lo : v;       // np.clip: NaN propagates
  v = v > hi ? hi : v;
  return v;
}

// one op record in registers (the first op of the NEXT unit is fetched while this unit stores)
struct OpRec {
  double2 a, b, c, d, e, f;    // doubles 0..11 of the record (WFK_SH_OP1)
};
// (plain loads: the 4-odd lanes of a piece, and the neighbouring unit's wave when a piece straddles units,
//  find the record's lines in the TCP / L2; non-temporal loads here cost 0.37 -> 0.51 ms on 2048 x 1e5)
__device__ __forceinline__ OpRec load_op(const double* p) {
  const double2* q = reinterpret_cast<const double2*>(p);
  OpRec r;
  r.a = q[0]; r.b = q[1]; r.c = q[2]; r.d = q[3]; r.e = q[4]; r.f = q[5];
  return r;
}
__device__ __forceinline__ int op_word(const OpRec& r) { return (int)__double2loint(r.a.x); }
__device__ __forceinline__ int op_ref(const OpRec& r) { return (int)__double2hiint(r.a.x); }

// one fused op over the lane's run: acc[k] += E_k (A(u_k) c_k + B(u_k) s_k), all state per lane.
// `kf`: samples between the record's reference sample and the lane's first one.
// SKIP (the sampler; not fir_short, whose 168-register budget the branch overflows): where no lane of the wave has an
// envelope on this op -- the tones of a multi-tone piece, cosine pulses -- the two inline exponentials of the envelope
// seeds are skipped.
template <int R, bool CUBIC, bool CPLX, bool SKIP = false>
__device__ __forceinline__ void short_op(const OpRec& o, const double* op, int w, double kf, double step,
                                         double (&acc)[R], double (&acci)[CPLX ? R : 1]) {
  const int env = (w >> 4) & 3;
  const double C1 = o.b.y, S1 = o.c.x, Hh = o.d.x, q = o.d.y;
  const double A0 = o.e.x, A1 = o.e.y, B0 = o.f.x, B1 = o.f.y;
  double A2 = 0.0, A3 = 0.0, B2 = 0.0, B3 = 0.0;
  if constexpr (CUBIC) {
    if ((w & 3) > 1) { A2 = op[12]; A3 = op[13]; B2 = op[14]; B3 = op[15]; }
  }
  // exact seeds at the lane's first sample, `kf` samples after the record's reference sample
  double c, s;
  {
    const double x = fma(kf, o.b.x, o.a.y);       // phase / pi
    const double n = rint(x);
    sincospi_small(x - n, &s, &c);
    if (((int)n) & 1) { c = -c; s = -s; }
  }
  double g = 1.0, r = 1.0;
  if (!SKIP || __any(env != 0)) {
    const double vv = fma(kf, Hh, o.c.y);
    const double ea = env == 1 ? -(vv * vv) : (env == 2 ? vv : 0.0);
    const double eb = env == 1 ? -Hh * (2.0 * vv + Hh) : (env == 2 ? Hh : 0.0);
    g = exp_small(ea);
    r = exp_small(eb);
  }
  double u = kf * step;
  double mr = 1.0, mi = 0.0;
  if constexpr (CPLX) {
    if (w & 8) { mr = 0.0; mi = 1.0; }
  }
  SH_EACH(R, k)
    double pa, pb;
    if constexpr (CUBIC) {
      pa = fma(fma(fma(A3, u, A2), u, A1), u, A0);
      pb = fma(fma(fma(B3, u, B2), u, B1), u, B0);
    } else {
      pa = fma(A1, u, A0);
      pb = fma(B1, u, B0);
    }
    const double val = fma(pa, c, pb * s);
    if constexpr (CPLX) {
      const double t = val * g;
      acc[k] = fma(mr, t, acc[k]);
      acci[k] = fma(mi, t, acci[k]);
    } else {
      acc[k] = fma(val, g, acc[k]);
    }
    if constexpr (k + 1 < R) {
      g *= r;
      r *= q;
      const double cn = fma(c, C1, -(s * S1));
      s = fma(s, C1, c * S1);
      c = cn;
      u += step;
    }
  SH_END
}

// short_op with the grid-rounding correction of the lean kernel's CORR builds (wfk_kernels.hip: corr_delta), for pulse
// trains milliseconds from t = 0: the reference evaluates at NumPy's rounded grid time x_k = fl(fl(j step) + t0) and rounds
// its phase, fl(w fl(x_k - shift)); W |t| ulp is 2e-10 rad at 300 MHz 1 ms out, and the recurrences stand for the ideal
// times.  Per sample: e_k = (x_k - x_ref) - (kf + k) dt, the rounding rho_k of ONE reference COS factor (w_m, s_m: the
// group's heaviest term's, host), d = W e_k - w_m eu + rho_k, and the phasor is turned by d to first order.  16-double
// record (word bit 12; degree <= 1): [12] w_m  [13] s_m  [14] W  [15] x_ref.  Channels without a pending shift only.
struct ShortCorr {
  double dj0;            // index of the lane's first sample in the caller's full grid, exact
  double step, t0, last, dlast;
};
template <int R, bool CPLX>
__device__ __forceinline__ void short_op_corr(const OpRec& o, const double* op, int w, double kf, const ShortCorr& cc,
                                              double (&acc)[R], double (&acci)[CPLX ? R : 1]) {
#pragma clang fp contract(off)
  const int env = (w >> 4) & 3;
  const double C1 = o.b.y, S1 = o.c.x, Hh = o.d.x, q = o.d.y;
  const double A0 = o.e.x, A1 = o.e.y, B0 = o.f.x, B1 = o.f.y;
  const double wm = op[12], sm = op[13], W = op[14], xref = op[15];
  double c, s;
  {
    const double x = __builtin_fma(kf, o.b.x, o.a.y);       // phase / pi
    const double n = rint(x);
    sincospi_small(x - n, &s, &c);
    if (((int)n) & 1) { c = -c; s = -s; }
  }
  double g = 1.0, r = 1.0;
  if (__any(env != 0)) {
    const double vv = __builtin_fma(kf, Hh, o.c.y);
    const double ea = env == 1 ? -(vv * vv) : (env == 2 ? vv : 0.0);
    const double eb = env == 1 ? -Hh * (2.0 * vv + Hh) : (env == 2 ? Hh : 0.0);
    g = exp_small(ea);
    r = exp_small(eb);
  }
  double u = kf * cc.step;
  double mr = 1.0, mi = 0.0;
  if constexpr (CPLX) {
    if (w & 8) { mr = 0.0; mi = 1.0; }
  }
  SH_EACH(R, k)
    // the reference's time of this sample, its distance from the ideal one, its rounded phase
    const double dj = cc.dj0 + (double)k;
    const double m = dj * cc.step;
    double t = m + cc.t0;
    if (dj == cc.dlast) t = cc.last;
    const double eps = (t - xref) - (kf + (double)k) * cc.step;
    const double um = t - sm;
    const double bb = um - t;
    const double eu = (t - (um - bb)) + (-sm - bb);
    const double pm = wm * um;
    const double rho = -__builtin_fma(wm, um, -pm);
    const double d = __builtin_fma(W, eps, __builtin_fma(-wm, eu, rho));
    const double ck = __builtin_fma(-d, s, c), sk = __builtin_fma(d, c, s);
    const double pa = __builtin_fma(A1, u, A0), pb = __builtin_fma(B1, u, B0);
    const double val = __builtin_fma(pa, ck, pb * sk);
    if constexpr (CPLX) {
      const double tv = val * g;
      acc[k] = __builtin_fma(mr, tv, acc[k]);
      acci[k] = __builtin_fma(mi, tv, acci[k]);
    } else {
      acc[k] = __builtin_fma(val, g, acc[k]);
    }
    if constexpr (k + 1 < R) {
      g *= r;
      r *= q;
      const double cn = __builtin_fma(c, C1, -(s * S1));
      s = __builtin_fma(s, C1, c * S1);
      c = cn;
      u += cc.step;
    }
  SH_END
}

// short_op for a BARE carrier (degree 0, no envelope: the tones of a multi-tone piece whose shared Gaussian is a closing
// op, the plateau of a flat top): acc[k] += A0 c_k + B0 s_k -- 6 instructions per sample instead of 12, no envelope seeds.
template <int R, bool CPLX>
__device__ __forceinline__ void short_op_bare(const OpRec& o, int w, double kf, double (&acc)[R], double (&acci)[CPLX ? R : 1]) {
  const double C1 = o.b.y, S1 = o.c.x, A0 = o.e.x, B0 = o.f.x;
  double c, s;
  {
    const double x = fma(kf, o.b.x, o.a.y);       // phase / pi
    const double n = rint(x);
    sincospi_small(x - n, &s, &c);
    if (((int)n) & 1) { c = -c; s = -s; }
  }
  SH_EACH(R, k)
    if constexpr (CPLX) {
      const double val = fma(A0, c, B0 * s);
      if (w & 8) acci[k] += val; else acc[k] += val;
    } else {
      acc[k] = fma(A0, c, fma(B0, s, acc[k]));
    }
    if constexpr (k + 1 < R) {
      const double cn = fma(c, C1, -(s * S1));
      s = fma(s, C1, c * S1);
      c = cn;
    }
  SH_END
}

// A chirp op (word bit 9; 16-double record): acc[k] += E_k (A(u_k) c_k + B(u_k) s_k) with the QUADRATIC phase
//   th(k) = th0 + k d1 + k^2 d2   (reference LINEARCHIRP, _waveform.pyx:323-324, times the carriers it is multiplied with).
// Along the lane z_{k+1} = z_k w_k, w_{k+1} = w_k v with v = exp(i 2 d2) from the record: the complex twin of the
// Gaussian recurrence, as in the lean kernel's chirp family; both phasors are seeded exactly at the lane's first sample.
template <int R, bool CPLX>
__device__ __forceinline__ void short_chirp(const OpRec& o, const double* op, int w, double kf, double step,
                                            double (&acc)[R], double (&acci)[CPLX ? R : 1]) {
  const int env = (w >> 4) & 3;
  const double vc = o.b.y, vs = o.c.x, Hh = o.d.x, q = o.d.y;
  const double A0 = o.e.x, A1 = o.e.y, B0 = o.f.x, B1 = o.f.y;
  const double d2 = op[12];                          // d2 / pi
  double c, s, wc, ws;
  {
    const double x = fma(kf, fma(kf, d2, o.b.x), o.a.y);          // phase / pi at the lane's first sample
    const double n = rint(x);
    sincospi_small(x - n, &s, &c);
    if (((int)n) & 1) { c = -c; s = -s; }
    const double y = fma(2.0 * kf + 1.0, d2, o.b.x);              // th(k + 1) - th(k), / pi
    const double m = rint(y);
    sincospi_small(y - m, &ws, &wc);
    if (((int)m) & 1) { wc = -wc; ws = -ws; }
  }
  const double vv = fma(kf, Hh, o.c.y);
  const double ea = env == 1 ? -(vv * vv) : (env == 2 ? vv : 0.0);
  const double eb = env == 1 ? -Hh * (2.0 * vv + Hh) : (env == 2 ? Hh : 0.0);
  double g = exp_small(ea), r = exp_small(eb);
  double u = kf * step;
  double mr = 1.0, mi = 0.0;
  if constexpr (CPLX) {
    if (w & 8) { mr = 0.0; mi = 1.0; }
  }
  SH_EACH(R, k)
    const double pa = fma(A1, u, A0), pb = fma(B1, u, B0);
    const double val = fma(pa, c, pb * s);
    if constexpr (CPLX) {
      const double t = val * g;
      acc[k] = fma(mr, t, acc[k]);
      acci[k] = fma(mi, t, acci[k]);
    } else {
      acc[k] = fma(val, g, acc[k]);
    }
    if constexpr (k + 1 < R) {
      g *= r;
      r *= q;
      const double cn = fma(c, wc, -(s * ws));
      s = fma(s, wc, c * ws);
      c = cn;
      const double wn = fma(wc, vc, -(ws * vs));
      ws = fma(ws, vc, wc * vs);
      wc = wn;
      u += step;
    }
  SH_END
}

// Closing op of a multi-tone piece (closing kind 1): the Gaussian the piece's tones share multiplies what they accumulated
// (g_k by the two-multiplier recurrence from exact seeds, as an op's own envelope)
template <int R, bool CPLX>
__device__ __forceinline__ void short_envmul(const OpRec& o, double kf, double (&acc)[R], double (&acci)[CPLX ? R : 1]) {
  const double Hh = o.d.x, q = o.d.y;
  const double vv = fma(kf, Hh, o.c.y);
  double g = exp_small(-(vv * vv)), r = exp_small(-Hh * (2.0 * vv + Hh));
  SH_EACH(R, k)
    acc[k] *= g;
    if constexpr (CPLX) acci[k] *= g;
    if constexpr (k + 1 < R) {
      g *= r;
      r *= q;
    }
  SH_END
}

// libm erf behind a call: inlined sixteen times into the closing op it would triple the kernel
static __device__ __attribute__((noinline)) double erf_call(double x) { return erf(x); }

// Closing op of a flat-top edge (square(width, edge): 0.5 +- 0.5 erf((t - s) / sigma), reference
// waveform.py:1096-1112): everything the piece's ops accumulated so far is multiplied by
// m0 + m1 erf(v_k), v_k = v0 + (koff + k) H.  At AWG rates an edge is a handful of samples and H = dt / sigma
// is of order 1, so erf is simply evaluated per sample (the lean kernel's Taylor-step form needs H <= 0.09).
template <int R, bool CPLX>
__device__ __forceinline__ void short_erfmul(const OpRec& o, double kf, double (&acc)[R], double (&acci)[CPLX ? R : 1]) {
  const double h = o.d.x, m0 = o.e.x, m1 = o.e.y;
  double v = fma(kf, h, o.c.y);
  SH_EACH(R, k)
    const double m = fma(m1, erf_call(v), m0);
    acc[k] *= m;
    if constexpr (CPLX) acci[k] *= m;
    v += h;
  SH_END
}

// Stateless closing multipliers (the op word's degree field: 2 | 3; records of 16 doubles): what the piece's ops
// accumulated is multiplied by the envelope they share.
//   2: a finite INTERP table on linspace knots (np.interp, reference _waveform.pyx:309-311) read as the continuous
//      piecewise-linear function it is: position in knot units q = q0 + (koff + k) dq clamped into [0, m - 1], value
//      f[floor q] + frac(q) (f[floor q + 1] - f[floor q]) from (value, difference) pairs -- one 16-byte gather per sample
//   3: mollifier(r): exp(1 / (x^2 - 1) + 1) inside |x| < 1, x = x0 + (koff + k) dx (reference _waveform.pyx:359-363)
template <int R, bool CPLX>
__device__ __forceinline__ void short_tabmul(const OpRec& o, const double* pool, double kf, double (&acc)[R],
                                             double (&acci)[CPLX ? R : 1]) {
  const double dq = o.d.x, qmax = o.e.x;
  const double2* tab = reinterpret_cast<const double2*>(pool) + (int64_t)o.e.y;
  double q = fma(kf, dq, o.c.y);
  // four samples' gathers in flight at a time (all sixteen would cost the kernel 48 more live registers: it sits
  // at its 168 for three waves per SIMD)
  constexpr int IB = R % 4 == 0 ? 4 : 1;
  SH_EACH(R / IB, kb)
    double fr[IB];
    double2 e[IB];
    SH_EACH(IB, kk)
      const double qc = fmin(fmax(q, 0.0), qmax);
      fr[kk] = __builtin_amdgcn_fract(qc);
      e[kk] = tab[(uint32_t)(int)qc];
      q += dq;
    SH_END
    __builtin_amdgcn_sched_barrier(0);
    SH_EACH(IB, kk)
      const double m = fma(fr[kk], e[kk].y, e[kk].x);
      acc[kb * IB + kk] *= m;
      if constexpr (CPLX) acci[kb * IB + kk] *= m;
    SH_END
    __builtin_amdgcn_sched_barrier(0);
  SH_END
}

template <int R, bool CPLX>
__device__ __forceinline__ void short_mollmul(const OpRec& o, double kf, double (&acc)[R], double (&acci)[CPLX ? R : 1]) {
  const double dx = o.d.x;
  double x = fma(kf, dx, o.c.y);
  SH_EACH(R, k)
    const double qv = fma(x, x, -1.0);
    const double qq = qv < -1e-300 ? qv : -1.0;                   // (outside the support: any harmless argument)
    double rc = __builtin_amdgcn_rcp(qq);
    rc = fma(fma(-qq, rc, 1.0), rc, rc);
    rc = fma(fma(-qq, rc, 1.0), rc, rc);
    const double v = exp_small(fmax(rc + 1.0, -740.0));            // (below: < 1e-321, the value is 0 to every bound)
    const double m = qv < 0.0 ? (rc + 1.0 < -740.0 ? 0.0 : v) : 0.0;
    acc[k] *= m;
    if constexpr (CPLX) acci[k] *= m;
    x += dx;
  SH_END
}

// libm log behind a call (the hyperbolic chirp multiplier: rare)
static __device__ __attribute__((noinline, unused)) double log_call(double x) { return log(x); }

// Chirp multipliers (closing kind with word bit 10; degree field 2: exponential, 3: hyperbolic; 16-double records): what
// the piece's ops accumulated is multiplied by sin(pi (ph0 + scale E_k)) with E_k = exp(a0 + (koff + k) da) -- advanced by
// its constant ratio from one exact seed per lane -- or E_k = log(l0 + (koff + k) dl).  Reference: EXPONENTIALCHIRP /
// HYPERBOLICCHIRP, waveforms/_waveform.pyx:326-332.  The phase is reduced per sample (rint) and the sine is the
// straight-line sincospi kernel: ~60 instructions per sample against ~2000 on the pointwise tier's term interpreter.
template <int R, bool CPLX>
__device__ __forceinline__ void short_xchirpmul(const OpRec& o, int w, double kf, double (&acc)[R], double (&acci)[CPLX ? R : 1]) {
  const double d = o.d.x, scale = o.e.x, ph0 = o.e.y;
  const bool hyp = (w & 1) != 0;
  double a = fma(kf, d, o.c.y);
  double e = hyp ? 0.0 : exp_small(a), rho = hyp ? 0.0 : exp_small(d);
  SH_EACH(R, k)
    double ev = e;
    if (hyp) ev = log_call(a > 1e-300 ? a : 1.0);
    const double x = fma(scale, ev, ph0);
    const double n = rint(x);
    double sn, cs;
    sincospi_small(x - n, &sn, &cs);
    const double m = (((int)n) & 1) ? -sn : sn;
    acc[k] *= m;
    if constexpr (CPLX) acci[k] *= m;
    e *= rho;
    a += d;
  SH_END
}

// Envelope x carrier in ONE op (word bit 7; bit 8: the envelope is a mollifier, else a table): acc[k] += F_k (A0 c_k + B0 s_k)
// -- a pulse as mixing(A * samplingPoints(...), freq, phase) makes it, in a record of 12 doubles like a Gaussian pulse's.
// The carrier's rotation of each batch runs while the batch's table gathers are in flight.
template <int R, bool CPLX>
__device__ __forceinline__ void short_cmul(const OpRec& o, const double* pool, int w, double kf, double (&acc)[R],
                                           double (&acci)[CPLX ? R : 1]) {
  const double C1 = o.b.y, S1 = o.c.x, A0 = o.e.x, B0 = o.f.x;
  double c = 1.0, s = 0.0;
  if (w & 4) {
    const double x = fma(kf, o.b.x, o.a.y);       // phase / pi
    const double n = rint(x);
    sincospi_small(x - n, &s, &c);
    if (((int)n) & 1) { c = -c; s = -s; }
  }
  double mr = 1.0, mi = 0.0;
  if constexpr (CPLX) {
    if (w & 8) { mr = 0.0; mi = 1.0; }
  }
  const double dq = o.d.x;
  double q = fma(kf, dq, o.c.y);
  if (w & 256) {
    SH_EACH(R, k)
      const double qv = fma(q, q, -1.0);
      const double qq = qv < -1e-300 ? qv : -1.0;                   // (outside the support: any harmless argument)
      double rc = __builtin_amdgcn_rcp(qq);
      rc = fma(fma(-qq, rc, 1.0), rc, rc);
      rc = fma(fma(-qq, rc, 1.0), rc, rc);
      const double v = exp_small(fmax(rc + 1.0, -740.0));
      const double m = qv < 0.0 ? (rc + 1.0 < -740.0 ? 0.0 : v) : 0.0;
      const double t = fma(A0, c, B0 * s) * m;
      if constexpr (CPLX) {
        acc[k] = fma(mr, t, acc[k]);
        acci[k] = fma(mi, t, acci[k]);
      } else {
        acc[k] += t;
      }
      if constexpr (k + 1 < R) {
        const double cn = fma(c, C1, -(s * S1));
        s = fma(s, C1, c * S1);
        c = cn;
        q += dq;
      }
    SH_END
    return;
  }
  const double qmax = o.d.y;
  const double2* tab = reinterpret_cast<const double2*>(pool) + (int64_t)o.e.y;
#ifndef WFK_SH_CMUL_IB
#define WFK_SH_CMUL_IB 4   // (8: 166 VGPRs + 2 spills, awg_interp 0.80 ms either way and the Gaussian pulses 0.39 -> 0.45 ms: the gathers are bound by the cache's tag rate, one line per lane, not by their latency)
#endif
  constexpr int IB = R % WFK_SH_CMUL_IB == 0 ? WFK_SH_CMUL_IB : 1;    // samples whose gathers are in flight together
  SH_EACH(R / IB, kb)
    double fr[IB];
    double2 e[IB];
    SH_EACH(IB, kk)
      const double qc = fmin(fmax(q, 0.0), qmax);
      fr[kk] = __builtin_amdgcn_fract(qc);
      e[kk] = tab[(uint32_t)(int)qc];
      q += dq;
    SH_END
    __builtin_amdgcn_sched_barrier(0);
    SH_EACH(IB, kk)
      const double t = fma(A0, c, B0 * s) * fma(fr[kk], e[kk].y, e[kk].x);
      {
        const double cn = fma(c, C1, -(s * S1));
        s = fma(s, C1, c * S1);
        c = cn;
      }
      if constexpr (CPLX) {
        acc[kb * IB + kk] = fma(mr, t, acc[kb * IB + kk]);
        acci[kb * IB + kk] = fma(mi, t, acci[kb * IB + kk]);
      } else {
        acc[kb * IB + kk] += t;
      }
    SH_END
    __builtin_amdgcn_sched_barrier(0);
  SH_END
}
// short_erfmul for the sampler: `len` = samples of the lane's run; an edge piece is 6-12 samples, and the erf call is
// skipped -- for the whole wave, once no lane is left inside its run -- for the samples behind it, which are never stored.
// (A function of its own: in fir_short the extra branches cost its 168-register budget up to 42 spills.)
template <int R, bool CPLX>
__device__ __forceinline__ void short_erfmul_run(const OpRec& o, double kf, int len, double (&acc)[R], double (&acci)[CPLX ? R : 1]) {
  const double h = o.d.x, m0 = o.e.x, m1 = o.e.y;
  double v = fma(kf, h, o.c.y);
  SH_EACH(R, k)
    double e = 0.0;
    if (k < len) e = erf_call(v);
    const double m = fma(m1, e, m0);
    acc[k] *= m;
    if constexpr (CPLX) acci[k] *= m;
    v += h;
  SH_END
}
// ---- float launches of the plain pulse train: packed fp32 arithmetic ------------------------------------------------
// Unpacked fp32 issues at the fp64 rate on this part, so the float launch of the short tier was issue-bound on the same
// ~37 fp64 instructions per sample as the double launch (0.31-0.35 of the 4 B/sample roof).  v_pk_fma_f32 / v_pk_mul_f32
// work on two floats per lane and instruction: a lane's run is evaluated as PAIRS of neighbouring samples (k, k + 1),
// the recurrences stepping by two samples -- rotation by (C2, S2) = (C1^2 - S1^2, 2 C1 S1), Gaussian pair
// g <- g R, R <- R q^4 with R_k = r_k r_(k+1) -- and the seeds (one sincospi, two exponentials per op and lane) are
// float too: the phase is reduced in fp64 (kf W dt + th0 to half-turns in [-1/2, 1/2]), the rest is a degree-9 / 8
// float polynomial and v_exp_f32.  Error budget: seeds 2e-7, eight recurrence steps 1e-6 -- against the 1e-3 contract of
// float outputs (tests: FP32_TOL = 5e-5 of peak).  6.5 packed instructions per op and sample instead of 14.
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void sincospi_f32(float r, float* sn, float* cs) {   // |r| <= 1/2
  const float a = fabsf(r);
  const bool swap = a > 0.25f;
  const float z = swap ? 0.5f - a : a;                      // exact
  const float t = z * 3.14159265358979f;
  const float t2 = t * t;
  float ps = 2.7557319e-06f;                                //  1/9!
  ps = fmaf(ps, t2, -1.9841270e-04f);
  ps = fmaf(ps, t2, 8.3333333e-03f);
  ps = fmaf(ps, t2, -1.6666667e-01f);
  const float s = fmaf(t * t2, ps, t);
  float pc = 2.4801587e-05f;                                //  1/8!
  pc = fmaf(pc, t2, -1.3888889e-03f);
  pc = fmaf(pc, t2, 4.1666667e-02f);
  pc = fmaf(pc, t2, -0.5f);
  const float c = fmaf(t2, pc, 1.0f);
  const float ss = swap ? c : s, cc = swap ? s : c;
  *sn = r < 0.0f ? -ss : ss;
  *cs = cc;
}

template <int R, bool CUBIC>
__device__ __forceinline__ void short_op_pk(const OpRec& o, const double* op, int w, double kf, double step,
                                            f32x2 (&acc)[R / 2]) {
  static_assert(R % 2 == 0, "pairs of samples");
  const int env = (w >> 4) & 3;
  const double C1 = o.b.y, S1 = o.c.x, Hh = o.d.x, q = o.d.y;
  const f32x2 A0 = (float)o.e.x, A1 = (float)o.e.y, B0 = (float)o.f.x, B1 = (float)o.f.y;
  f32x2 A2 = 0.0f, A3 = 0.0f, B2 = 0.0f, B3 = 0.0f;
  if constexpr (CUBIC) {
    if ((w & 3) > 1) { A2 = (float)op[12]; A3 = (float)op[13]; B2 = (float)op[14]; B3 = (float)op[15]; }
  }
  // seeds at the lane's first sample (phase reduced in fp64) and at the one behind it
  f32x2 c, s;
  {
    const double x = fma(kf, o.b.x, o.a.y);       // phase / pi
    const double n = rint(x);
    float s0, c0;
    sincospi_f32((float)(x - n), &s0, &c0);
    if (((int)n) & 1) { c0 = -c0; s0 = -s0; }
    const float c1 = (float)C1, s1 = (float)S1;
    c = f32x2{c0, c0 * c1 - s0 * s1};
    s = f32x2{s0, s0 * c1 + c0 * s1};
  }
  const f32x2 C2 = (float)(C1 * C1 - S1 * S1), S2 = (float)(2.0 * C1 * S1);
  f32x2 g = 1.0f, rr = 1.0f;
  f32x2 q4 = 1.0f;
  if (env != 0) {
    const double vv = fma(kf, Hh, o.c.y);
    const double ea = env == 1 ? -(vv * vv) : vv;
    const double eb = env == 1 ? -Hh * (2.0 * vv + Hh) : Hh;
    const float g0 = __builtin_amdgcn_exp2f((float)(ea * 1.4426950408889634));
    const float r0 = __builtin_amdgcn_exp2f((float)(eb * 1.4426950408889634));
    const float qf = (float)q, r1 = r0 * qf;
    g = f32x2{g0, g0 * r0};
    rr = f32x2{r0 * r0 * qf, r1 * r1 * qf};
    const float q2 = qf * qf;
    q4 = q2 * q2;
  }
  const float u0 = (float)(kf * step), dt = (float)step;
  f32x2 u = f32x2{u0, u0 + dt};
  const f32x2 du = 2.0f * dt;
  SH_EACH(R / 2, k)
    f32x2 pa, pb;
    if constexpr (CUBIC) {
      pa = ((A3 * u + A2) * u + A1) * u + A0;
      pb = ((B3 * u + B2) * u + B1) * u + B0;
    } else {
      pa = A1 * u + A0;
      pb = B1 * u + B0;
    }
    const f32x2 val = pa * c + pb * s;
    acc[k] = val * g + acc[k];
    if constexpr (k + 1 < R / 2) {
      g = g * rr;
      rr = rr * q4;
      const f32x2 cn = c * C2 - s * S2;
      s = s * C2 + c * S2;
      c = cn;
      u = u + du;
    }
  SH_END
}

}  // namespace shdev
