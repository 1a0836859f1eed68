// wfk_kernels.hip -- gfx950 (MI355X / CDNA4) sampler kernels.
//
// One kernel family evaluates
//     out[ch, j] = offset + clip( sum_k amp_k * prod_f f(t_j - shift_f)^n_f )
// for every channel and sample: the device replacement for the reference's
// calc_parts/_calc/_apply/_fill_parts passes (waveforms/_waveform.pyx:130-169,
// waveforms/waveform.py:524-527).  It is elementwise/transcendental work bounded by
// HBM *write* bandwidth (8 B/sample fp64): no contraction, hence no MFMA.
//
// Mapping (wave64, 256-thread workgroups):
//   workgroup -> `tiles_per_chunk` consecutive tiles of 256*NS samples of one channel
//   wave      -> 64*NS consecutive samples; lane l owns samples j0+l+64*k, k<NS, so every
//                store instruction of a wave writes 64 consecutive elements (coalesced)
//   piece     -> its parameter block is staged in LDS once per workgroup and reused for
//                every tile that stays inside the piece; all control flow on the program
//                (term / factor / mode) is wave-uniform (scalar branches)
// Uniform-grid fast paths keep the per-sample cost at a few FMAs instead of a libm call:
//   COS   : one exact sincos per lane per tile, then cos(th0+k*dth) = c0*C[k] - s0*S[k]
//           with the (C,S) table (k<NS) read from LDS as wave-wide broadcasts
//   GAUSS : g_{k+1} = g_k r_k, r_{k+1} = r_k q   (two exact exps per lane per tile)
//   EXP   : e_{k+1} = e_k rho          LINEAR: u_k = u_0 + k D
// Everything else (and tlist mode) evaluates the primitive with the device libm.
#include <hip/hip_runtime.h>

#include <string>
#include <type_traits>
#include <utility>

#include "wfk.h"
#include "wfk_internal.h"

namespace {

__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
// 64-bit values the compiler cannot prove wave-uniform (anything downstream of a 64-bit
// division, which only exists on the VALU): pin them to SGPRs.  What hangs off them then
// becomes scalar: s_load instead of global_load (lgkmcnt, so a wait for it does NOT also
// wait for every store in flight, which on gfx9 share vmcnt with the vector loads), and
// SGPR-base + 32-bit-offset addressing for the output stores.
__device__ __forceinline__ int64_t uni64(int64_t v) {
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
  const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)v >> 32));
  return (int64_t)(((uint64_t)hi << 32) | lo);
}
// Plan tables (pieces, channels) are never written by a kernel; reading them through the
// constant address space lets the backend keep using s_load after the kernel's first store
// (for plain global pointers it must assume the stores may have clobbered them).
#define WFK_CONST __attribute__((address_space(4)))
template <typename V>
__device__ __forceinline__ V cload(const void* base, int byte_off) {
  return *reinterpret_cast<const WFK_CONST V*>(reinterpret_cast<uintptr_t>(base) + byte_off);
}
__device__ __forceinline__ DevPiece load_piece(const DevPiece* p) {
  DevPiece P;
  P.start = cload<int64_t>(p, offsetof(DevPiece, start));
  P.stop = cload<int64_t>(p, offsetof(DevPiece, stop));
  P.par_off = cload<int64_t>(p, offsetof(DevPiece, par_off));
  P.n_blk = cload<int32_t>(p, offsetof(DevPiece, n_blk));
  P.flags = cload<int32_t>(p, offsetof(DevPiece, flags));
  P.first_len = cload<int32_t>(p, offsetof(DevPiece, first_len));
  P.pad = 0;
  return P;
}
template <typename P>
__device__ __forceinline__ P* uniptr(P* p) {
  // rebuilt as a GLOBAL pointer, or the integer round trip degrades the accesses to flat_*
  using G = __attribute__((address_space(1))) P*;
  return (P*)reinterpret_cast<G>(uni64(reinterpret_cast<int64_t>(p)));
}

// Compile-time unrolled loop.  Register arrays are only ever indexed with constants, so
// SROA splits them into independent scalars up front (a `#pragma unroll` loop indexes them
// dynamically until late, and the array then becomes ONE 32-register tuple that is copied
// wholesale at every control-flow merge).
template <int... K, typename F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, K...>, F&& f) {
  (f(std::integral_constant<int, K>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f));
}
#define WFK_EACH(N, k) static_for<N>([&](auto k##_) __attribute__((always_inline)) { constexpr int k = decltype(k##_)::value;
#define WFK_END });

// t[j] = fl(fl(j*step) + t0): NumPy's linspace / arange element formula, two roundings.
// SLICE (wfk_grid.i0 != 0: the plan is a time slice of a longer grid): the index is that of the caller's FULL grid.
// The general kernel is instantiated for both; whole grids run the code they always ran -- one more operation in
// here (a 64-bit add, a double add, anything) moves the register allocation of the complex double build of the direct
// tier, which sits at its 256 registers, from 37 to 700-840 spilled ones (0.80 -> 1.48 ms on tools/cplx_direct_bench.py).
template <bool SLICE = true>
__device__ __forceinline__ double grid_time(const KArgs& a, int64_t j) {
#pragma clang fp contract(off)
  // (int64 -> double is a multi-instruction sequence on this ISA, uint32 -> double is one: the
  //  per-sample callers of the direct tier feel it; both conversions are exact)
  double dj;
  // (slices: j and i0 are exact doubles and so is their sum (< 2^53): one conversion of a 32-bit index + one add,
  //  instead of a 64-bit add and the multi-instruction int64 conversion)
  if constexpr (SLICE) dj = a.n <= 0xffffffffLL ? (double)(uint32_t)j + (double)a.i0 : (double)(j + a.i0);
  else dj = a.n <= 0xffffffffLL ? (double)(uint32_t)j : (double)j;
  double m = dj * a.step;
  double t = m + a.t0;
  if (a.has_last && j == a.n - 1) t = a.last;
  return t;
}

// sin/cos of a large phase without the Payne-Hanek register cost of libm's sincos:
// theta/pi as an exact two-term product (1/pi split hi+lo), reduced with rint, then
// sincospi on |r| <= 1/2.  Phase error ~1 ulp of r, i.e. as accurate as libm.
__device__ __attribute__((noinline)) double2 sincos_phase(double theta) {
  // contraction OFF: under -ffp-contract=fast `xh - n` below becomes fma(theta, IPI_HI, -n), which
  // already contains the low half of the product that `xl` adds again -- a phase error of up to
  // half an ulp of theta/pi (3e-9 rad at theta = 3e7)
#pragma clang fp contract(off)
  const double IPI_HI = 0.31830988618379069, IPI_LO = -1.9678676675182486e-17;
  const double xh = theta * IPI_HI;
  const double xl = fma(theta, IPI_HI, -xh) + theta * IPI_LO;
  const double n = rint(xh);
  const double r = (xh - n) + xl;
  double ss, cc;
  sincospi(r, &ss, &cc);
  const bool odd = ((long long)n) & 1;
  return make_double2(odd ? -cc : cc, odd ? -ss : ss);  // (cos, sin)
}

// Out-of-line so that libm's polynomial constants are not hoisted into VGPRs that
// stay live across the whole sampling loop.
__device__ __attribute__((noinline)) double exp_seed(double x) { return exp(x); }

// (sin, cos)(pi x): libm's exact reduction, out of line like the other seed functions
__device__ __attribute__((noinline)) double2 sincospi_seed(double x) {
  double ss, cc;
  sincospi(x, &ss, &cc);
  return make_double2(cc, ss);
}

// 1 / y to double precision: the hardware estimate + two Newton steps (the IEEE division sequence is 2-3x longer)
__device__ __forceinline__ double rcp_nr(double y) {
  double r = __builtin_amdgcn_rcp(y);
  r = fma(fma(-y, r, 1.0), r, r);
  r = fma(fma(-y, r, 1.0), r, r);
  return r;
}

// exp(x) inline: x = n ln2 + r, |r| <= 0.347, exp(r) by its Taylor polynomial of degree 13 (next term 4e-18),
// scaled by v_ldexp_f64 (which also does the gradual underflow).  ~20 VALU instructions, no table, no call;
// <= 2 ulp against libm's <= 1 (the Gaussian and exponential envelopes of the pointwise ops).
__device__ __forceinline__ double exp_inline(double x) {
  const double n = rint(x * 1.4426950408889634);
  double r = fma(-n, 6.93147180369123816490e-01, x);
  r = fma(-n, 1.90821492927058770002e-10, r);
  double p = fma(r, 1.6059043836821613e-10, 2.08767569878681e-09);      // 1/13!, 1/12!
  p = fma(r, p, 2.505210838544172e-08);
  p = fma(r, p, 2.755731922398589e-07);
  p = fma(r, p, 2.7557319223985893e-06);
  p = fma(r, p, 2.48015873015873e-05);
  p = fma(r, p, 1.984126984126984e-04);
  p = fma(r, p, 1.3888888888888889e-03);
  p = fma(r, p, 8.333333333333333e-03);
  p = fma(r, p, 4.1666666666666664e-02);
  p = fma(r, p, 1.6666666666666666e-01);
  p = fma(r, p, 0.5);
  p = fma(r, p, 1.0);
  p = fma(r, p, 1.0);
  double e = ldexp(p, (int)n);
  e = x < -746.0 ? 0.0 : e;                       // (also x = -inf, where r is NaN)
  e = x > 710.0 ? __builtin_inf() : e;
  return e;
}

template <bool TLIST, bool SLICE = true>
__device__ __forceinline__ double time_at(const KArgs& a, int64_t j) {
  if (TLIST) {
    int64_t jj = j < a.n ? j : a.n - 1;
    return a.tlist[jj];
  }
  return grid_time<SLICE>(a, j);
}

// ---- direct primitives (device libm), evaluated at u = t - shift ----------------
// Formulas: reference waveforms/_waveform.pyx:290-371.
__device__ double np_linspace_at(double start, double stop, int64_t m, int64_t k) {
#pragma clang fp contract(off)
  if (m == 1) return start;
  if (k == m - 1) return stop;
  double step = (stop - start) / (double)(m - 1);
  double a = (double)k * step;
  return a + start;
}

// the same with the step (one IEEE division, done by the host) passed in: the division was most
// of the cost of a knot lookup
__device__ __forceinline__ double np_linspace_at_s(double start, double stop, int64_t m, int64_t k,
                                                   double step) {
#pragma clang fp contract(off)
  if (k == m - 1) return stop;
  double a = (double)k * step;
  return a + start;
}

__device__ __forceinline__ double np_linspace_at_s(double start, double stop, int m, int k, double step) {
#pragma clang fp contract(off)
  if (k == m - 1) return stop;
  double a = (double)k * step;
  return a + start;
}

__device__ double prim_interp(double x, double start, double stop, const double* fp, int64_t m) {
#pragma clang fp contract(off)
  if (isnan(x)) return x;
  double x0 = np_linspace_at(start, stop, m, 0), xl = np_linspace_at(start, stop, m, m - 1);
  if (x > xl) return fp[m - 1];
  if (x < x0) return fp[0];
  int64_t lo = 0, hi = m;
  while (hi - lo > 1) {
    int64_t mid = lo + (hi - lo) / 2;
    if (x >= np_linspace_at(start, stop, m, mid)) lo = mid; else hi = mid;
  }
  int64_t j = lo;
  if (j == m - 1) return fp[j];
  double xj = np_linspace_at(start, stop, m, j), xj1 = np_linspace_at(start, stop, m, j + 1);
  if (xj == x) return fp[j];
  double slope = (fp[j + 1] - fp[j]) / (xj1 - xj);
  double r = slope * (x - xj) + fp[j];
  if (isnan(r)) {
    r = slope * (x - xj1) + fp[j + 1];
    if (isnan(r) && fp[j] == fp[j + 1]) r = fp[j];
  }
  return r;
}

// The same np.interp result with the knot found from an O(1) guess (corrected against the
// exact knot abscissae, so the index is the one the binary search finds) and the slope read
// from the host's table.  Requires stop > start, m >= 2 (the host checks).
__device__ double prim_interp_fast(double x, double start, double stop, const double* fp,
                                   const double* sl, int64_t m, double inv_step, double step) {
#pragma clang fp contract(off)
  if (isnan(x)) return x;
  if (x > stop) return fp[m - 1];
  if (x < start) return fp[0];
  int64_t j = (int64_t)((x - start) * inv_step);
  j = j < 0 ? 0 : (j > m - 1 ? m - 1 : j);
  while (j > 0 && x < np_linspace_at_s(start, stop, m, j, step)) --j;
  while (j < m - 1 && x >= np_linspace_at_s(start, stop, m, j + 1, step)) ++j;
  if (j == m - 1) return fp[j];
  const double xj = np_linspace_at_s(start, stop, m, j, step);
  if (xj == x) return fp[j];
  const double slope = sl[j];
  double r = slope * (x - xj) + fp[j];
  if (isnan(r)) {
    r = slope * (x - np_linspace_at_s(start, stop, m, j + 1, step)) + fp[j + 1];
    if (isnan(r) && fp[j] == fp[j + 1]) r = fp[j];
  }
  return r;
}

__device__ double prim_drag(double t, const double* r) {
  const double t0 = r[3], freq = r[4], width = r[5], delta = r[6], bf = r[7], phase = r[8];
  const double PI = 3.141592653589793;
  double o = PI / width;
  double sx = sin(o * (t - t0));
  double Ox = sx * sx;
  double wt = 2 * PI * (freq + delta) * t - (2 * PI * delta * t0 + phase);
  if (isnan(bf) || bf - delta == 0) return Ox * cos(wt);
  double b = 1 / PI / 2 / (bf - delta);
  double Oy = -b * o * sin(2 * o * (t - t0));
  double s, c;
  sincos(wt, &s, &c);
  return Ox * c + Oy * s;
}

__device__ double prim_mollifier(double u, const double* r, const double* pool) {
  double x = u / r[3];
  double q = fabs(x) * fabs(x) - 1.0;
  int d = (int)r[4];
  if (d == 0) return q >= 0 ? 0.0 : exp(1.0 / q + 1.0);
  int deg = (int)r[5];
  const double* p = pool + (int64_t)r[6];
  double px = 0.0;
  for (int k = 0; k <= deg; ++k) px = px * x + p[k];
  double env = q >= 0 ? 0.0 : exp(1.0 / q + 1.0) / pow(-q, 2.0 * d);
  return env * px / r[7];
}

__device__ double prim_dgauss(double u, const double* r) {
  double x = u / r[3];
  int n = (int)r[4];
  double h0 = 1.0, h1 = 2.0 * x, h = n == 0 ? h0 : h1;
  for (int k = 1; k < n; ++k) {
    h = 2.0 * x * h1 - 2.0 * k * h0;
    h0 = h1;
    h1 = h;
  }
  return r[5] * h * exp(-(x * x));
}

// multi-notch DRAG (ids 16/17) from the compiled tables of waveforms_amd/multy_drag.py
// (reference semantics: waveforms/multy_drag.py:31-155, incl. np.piecewise's "last
// matching condition wins" at the region joints)
__device__ double prim_mdrag(double u, const double* a) {
  const double PI = 3.141592653589793;
  const double t0 = a[0], freq = a[1], width = a[2], delta = a[3], phase = a[4], plateau = a[5],
               half = a[6];
  const int m = (int)a[7], dq = (int)a[8];
  const double* px = a + 9;
  const double* py = px + (m + 1);
  const double* cst = py + (m + 1);
  const double o = PI / width;
  const double mid1 = t0 + width / 2, mid2 = t0 + plateau + width / 2;
  double ox, oy;
  const bool rising = u <= mid1, falling = u >= mid2;
  if (rising || falling) {
    const double tau = falling ? u - t0 - plateau : u - t0;
    double s, c;
    sincos(o * tau, &s, &c);
    double ex = 0, ey = 0, dx = 0, dy = 0, sp = 1.0;   // even / odd parts
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
      const double tau = u - t0 - width / 2;
      double hx = 0, hy = 0;
      for (int i = 0; i <= dq; ++i) { hx = hx * tau + q[i]; hy = hy * tau + q[dq + 1 + i]; }
      ox = hx; oy = hy;
    }
    if (u >= mid2 && u <= mid2 + half) {
      const double tau = u - t0 - plateau - width / 2;
      const double* qr = q + 2 * (dq + 1);
      double hx = 0, hy = 0;
      for (int i = 0; i <= dq; ++i) { hx = hx * tau + qr[i]; hy = hy * tau + qr[dq + 1 + i]; }
      ox = hx; oy = hy;
    }
  }
  const double wt = 2 * PI * (freq + delta) * u - (2 * PI * delta * t0 + phase);
  double sw, cw;
  sincos(wt, &sw, &cw);
  return ox * cw + oy * sw;
}

__device__ double prim_direct(int type, double u, const double* r, const double* pool) {
  const double PI = 3.141592653589793;
  switch (type) {
    case WFK_LINEAR: return u;
    case WFK_GAUSSIAN: { double x = u / r[3]; return exp(-(x * x)); }
    case WFK_ERF: return erf(u / r[3]);
    case WFK_COS: return cos(r[3] * u);
    case WFK_SINC: { double x = r[3] * u; double y = PI * (x == 0 ? 1.0e-20 : x); return sin(y) / y; }
    case WFK_EXP: return exp(r[3] * u);
    case WFK_INTERP:
      if (r[7] >= 0.0)
        return prim_interp_fast(u, r[3], r[4], pool + (int64_t)r[6], pool + (int64_t)r[7], (int64_t)r[5], r[8], r[9]);
      return prim_interp(u, r[3], r[4], pool + (int64_t)r[6], (int64_t)r[5]);
    case WFK_LINEARCHIRP:
      return sin(r[6] + 2 * PI * ((r[4] - r[3]) / (2 * r[5]) * (u * u) + r[3] * u));
    case WFK_EXPONENTIALCHIRP: return sin(r[5] + 2 * PI * r[3] * (exp(r[4] * u) - 1) / r[4]);
    case WFK_HYPERBOLICCHIRP: return sin(r[5] + 2 * PI * r[3] / r[4] * log(1 + r[4] * u));
    case WFK_COSH: return cosh(r[3] * u);
    case WFK_SINH: return sinh(r[3] * u);
    case WFK_DRAG: return prim_drag(u, r);
    case WFK_MOLLIFIER: return prim_mollifier(u, r, pool);
    case WFK_D_GAUSSIAN: return prim_dgauss(u, r);
    case WFK_DRAG_SIN: case WFK_DRAG_SINX: return prim_mdrag(u, pool + (int64_t)r[3]);
    default: return __builtin_nan("");
  }
}

__device__ double np_power(double v, double n) {
  if (n == 2.0) return v * v;
  if (n == -1.0) return 1.0 / v;
  if (n == 0.5) return sqrt(v);
  if (n == 0.0) return 1.0;
  return pow(v, n);
}

// ---- one factor over the wave tile: prod[k] *= f(t_k - shift)^power -------------
// blk: LDS parameter block, r: this factor's record inside it, j0: lane's first sample.
template <typename T, bool TLIST, bool DIRECT, int NS, bool SLICE>
__device__ __forceinline__ void apply_factor(const double* blk, const double* r, const KArgs& a,
                                             double tshift, int64_t j0, T (&prod)[NS],
                                             double* s_val) {
  const int mode = uni((int)r[0]);
  const double shift = r[2];
  if (DIRECT && mode >= WFK_M_REUSE) {
    // the previous term evaluated this very factor: its values are still in the value buffer
    const double* mine = s_val + threadIdx.x;
#pragma unroll
    for (int k = 0; k < NS; ++k) prod[k] *= (T)mine[k * WFK_WG];
    return;
  }
  if (!TLIST && mode >= 100) {
    // seed at the lane's first sample, evaluated exactly as the reference does
    double x = grid_time<SLICE>(a, j0);
    if (tshift != 0.0) x = x - tshift;
    const double u0 = x - shift;
    if (mode == WFK_M_COS_TAB) {
      const double2 cs0 = sincos_phase(r[3] * u0);
      const T c0 = (T)cs0.x, s0 = (T)cs0.y;
      const double2* tab = reinterpret_cast<const double2*>(blk + uni((int)r[9]));
#pragma unroll
      for (int k = 0; k < NS; ++k) {
        const double2 cs = tab[k];  // wave-wide LDS broadcast
        prod[k] *= c0 * (T)cs.x - s0 * (T)cs.y;
      }
    } else if (mode == WFK_M_GAUSS_REC) {
      const double v = u0 / r[3], Hh = r[4];
      const double gd = exp_seed(-(v * v)), rd = exp_seed(-Hh * (2.0 * v + Hh));
      if (sizeof(T) == 4 && uni((int)r[9]) == 0) {
        // float output but the state would leave float's exponent range: keep it in double
        double g = gd, rr = rd;
        const double q = r[5];
#pragma unroll
        for (int k = 0; k < NS; ++k) {
          prod[k] *= (T)g;
          g *= rr;
          rr *= q;
        }
      } else {
        T g = (T)gd, rr = (T)rd;
        const T q = (T)r[5];
#pragma unroll
        for (int k = 0; k < NS; ++k) {
          prod[k] *= g;
          g *= rr;
          rr *= q;
        }
      }
    } else if (mode == WFK_M_EXP_REC) {
      const double ed = exp_seed(r[3] * u0);
      if (sizeof(T) == 4 && uni((int)r[9]) == 0) {
        double e = ed;
        const double rho = r[4];
#pragma unroll
        for (int k = 0; k < NS; ++k) {
          prod[k] *= (T)e;
          e *= rho;
        }
      } else {
        T e = (T)ed;
        const T rho = (T)r[4];
#pragma unroll
        for (int k = 0; k < NS; ++k) {
          prod[k] *= e;
          e *= rho;
        }
      }
    } else if (mode == WFK_M_SINC_TAB) {
      // np.sinc(b u) = sin(pi x) / (pi x), x = b u (reference _waveform.pyx:303-305).  Seed: sincospi of the
      // lane's own x (rounded as the reference rounds it); sample k: the seed rotated by the table entry, the
      // argument th0 + k dth with the table's own dth -- numerator and denominator then belong to the same
      // argument to 1e-16 of IT, also where it passes through zero in mid-stride.  Below 1e-3 the quotient
      // is taken from its series (sin's absolute 1e-16 divided by a tiny argument would show).
      const double x0 = r[3] * u0;
      const double2 cs0 = sincospi_seed(x0);
      const double th0 = 3.141592653589793 * x0, dth = r[4];
      const double2* tab = reinterpret_cast<const double2*>(blk + uni((int)r[9]));
#pragma unroll
      for (int k = 0; k < NS; ++k) {
        const double2 cs = tab[k];  // wave-wide LDS broadcast
        const double sn = fma(cs0.y, cs.x, cs0.x * cs.y);
        const double y = fma((double)k, dth, th0);
        const double y2 = y * y;
        double v = sn * rcp_nr(y);
        const double ser = fma(y2, fma(y2, 8.3333333333333332e-03, -1.6666666666666666e-01), 1.0);
        v = y2 < 1e-6 ? ser : v;
        prod[k] *= (T)v;
      }
    } else if (mode == WFK_M_MOLL_REC) {
      const double ir = 1.0 / r[3], D = r[4];
#pragma unroll
      for (int k = 0; k < NS; ++k) {
        const double xx = fma((double)k, D, u0) * ir;
        const double q = fma(xx, xx, -1.0);
        const double qq = q < -1e-300 ? q : -1.0;                     // (outside the support: any harmless argument)
        const double v = exp_inline(rcp_nr(qq) + 1.0);
        prod[k] *= (T)(q < 0.0 ? v : 0.0);
        // (four samples interleaved at most: with all 16 exponentials in flight the complex double build of the
        //  direct tier, already at its 256 registers, spilled 700)
        if ((k & 3) == 3) __builtin_amdgcn_sched_barrier(0);
      }
    } else if (mode == WFK_M_INTERP_LIN) {
      // np.interp on linspace knots with a finite table (reference _waveform.pyx:309-311), as a continuous
      // piecewise-linear function: segment from the O(1) guess, x clamped into [start, stop] (the constant
      // continuation np.interp applies outside), 2 gathers per sample, all NS samples' loads in flight
      const double start = r[3], stop = r[4], inv_step = r[8], step = r[9];
      const int last = (int)r[5] - 2;                // last segment
      const double* fp = a.pool + (int64_t)r[6];
      const double* sl = a.pool + (int64_t)r[7];
      const double D = 64.0 * a.step;                 // (the general kernel's lane stride; x to an ulp or two: continuity again)
      // batches of 8 samples: all 16 gathers of a batch are issued before the first value is used (left to
      // itself the compiler waited for each sample's pair with vmcnt(0): one memory round trip per sample)
      constexpr int IB = NS % 8 == 0 ? 8 : 1;
#pragma unroll
      for (int k0 = 0; k0 < NS; k0 += IB) {
        double dd[IB], f0[IB], s0[IB];
#pragma unroll
        for (int kk = 0; kk < IB; ++kk) {
          const double x = fma((double)(k0 + kk), D, u0);   // (same box: 2.37 ms against 2.46 with the exact per-sample grid time)
          const double xc = fmin(fmax(x, start), stop);
          int j = (int)((xc - start) * inv_step);
          j = j > last ? last : j;
          const double d0 = xc - fma((double)j, step, start);
          dd[kk] = x != x ? x : d0;                           // np.interp(NaN) is NaN
          f0[kk] = fp[j];
          s0[kk] = sl[j];
        }
        __builtin_amdgcn_sched_barrier(0);                  // loads above, uses below
#pragma unroll
        for (int kk = 0; kk < IB; ++kk) prod[k0 + kk] *= (T)fma(s0[kk], dd[kk], f0[kk]);
      }
    } else if (mode == WFK_M_INTERP_GRID) {
      // np.interp (reference _waveform.pyx:309-311) at the exact per-sample grid times; same
      // operations as prim_interp_fast, restructured so that a batch of four samples has its
      // eight table loads in flight at once
#pragma clang fp contract(off)
      const double start = r[3], stop = r[4], inv_step = r[8], step = r[9];
      const int m = (int)r[5];                       // knots: < 2^31 (host-checked)
      const double* fp = a.pool + (int64_t)r[6];
      const double* sl = a.pool + (int64_t)r[7];
#pragma unroll
      for (int k0 = 0; k0 < NS; k0 += 4) {
        double xs[4], xj[4], f0[4], sv[4];
        int js[4];
        bool flat[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          if (k0 + kk < NS) {
            double x = grid_time<SLICE>(a, j0 + 64 * (int64_t)(k0 + kk));
            if (tshift != 0.0) x = x - tshift;
            x = x - shift;
            int j;
            bool fl = true;
            if (!(x <= stop)) j = m - 1;              // x > stop, or NaN (patched below)
            else if (x < start) j = 0;
            else {
              j = (int)((x - start) * inv_step);
              j = j < 0 ? 0 : (j > m - 1 ? m - 1 : j);
              while (j > 0 && x < np_linspace_at_s(start, stop, m, j, step)) --j;
              while (j < m - 1 && x >= np_linspace_at_s(start, stop, m, j + 1, step)) ++j;
              fl = j == m - 1;
            }
            xs[kk] = x; js[kk] = j;
            xj[kk] = np_linspace_at_s(start, stop, m, j, step);
            flat[kk] = fl || xj[kk] == x;
          }
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
          if (k0 + kk < NS) { f0[kk] = fp[js[kk]]; sv[kk] = sl[js[kk]]; }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          if (k0 + kk < NS) {
            double v = flat[kk] ? f0[kk] : sv[kk] * (xs[kk] - xj[kk]) + f0[kk];
            if (isnan(xs[kk])) v = xs[kk];
            else if (isnan(v))   // non-finite table values: the scalar routine's fallbacks
              v = prim_interp_fast(xs[kk], start, stop, fp, sl, m, inv_step, step);
            prod[k0 + kk] *= (T)v;
          }
        }
      }
    } else {  // WFK_M_LIN_REC
      const double D = r[3];
#pragma unroll
      for (int k = 0; k < NS; ++k) prod[k] *= (T)(u0 + (double)k * D);
    }
    return;
  }
  if (!DIRECT) return;  // fast-only build: the plan holds no direct factor
  // direct evaluation: a rolled loop (small code: prim_direct is one big switch over libm
  // calls) that parks the NS values of this thread in LDS, then an unrolled pass multiplies
  // them into the register array with compile-time indices.  (Rotating the register array
  // through the rolled loop instead cost ~30 moves per sample and factor, more than most
  // primitives.)  Every thread reads back only what it wrote itself: no barrier.
  const double pw = r[1];
  const bool has_pw = pw != 1.0;
  double* mine = s_val + threadIdx.x;
  if (mode == WFK_M_SAMPLED) {
    // caller-evaluated factor (WFK_SAMPLED; Python callables of function()/function_lib=): the
    // value of sample j is table[j - i0]; the index is clamped, so a lane beyond the piece (or
    // beyond n) reads a valid entry that the store phase then drops
    const double* tab = a.pool + (int64_t)r[3];
    const int64_t i0 = (int64_t)r[4], m = (int64_t)r[5];
#pragma unroll 1
    for (int k = 0; k < NS; ++k) {
      int64_t i = j0 + 64 * (int64_t)k - i0;
      i = i < 0 ? 0 : (i >= m ? m - 1 : i);
      double v = m > 0 ? tab[i] : __builtin_nan("");
      if (has_pw) v = np_power(v, pw);
      mine[k * WFK_WG] = v;
    }
  } else {
#pragma unroll 1
    for (int k = 0; k < NS; ++k) {
      double x = time_at<TLIST, SLICE>(a, j0 + 64 * (int64_t)k);
      if (tshift != 0.0) x = x - tshift;
      double v = prim_direct(mode, x - shift, r, a.pool);
      if (has_pw) v = np_power(v, pw);
      mine[k * WFK_WG] = v;
    }
  }
#pragma unroll
  for (int k = 0; k < NS; ++k) prod[k] *= (T)mine[k * WFK_WG];
}

// ---- fused carrier-envelope op over the wave tile (see WFK_FCE_* ) --------------
//   acc[k] += E_k * ( A(u_k) * cos(th_k) + B(u_k) * sin(th_k) )
// One phasor seed (exact sincos) and at most one Gaussian seed (two exps) per lane per
// tile; per sample 2..13 FMAs.  The phasor at sample k is seed * table[k] (no recurrence,
// no error growth); the (C,S) table entries are wave-wide LDS broadcasts.
struct FceSeeds { double c, s, g, r; };

// All libm work of one fused op in ONE out-of-line routine: the three polynomial chains
// (sincospi, exp, exp) are independent, so the scheduler interleaves them, and their
// constants are not live across the sampling loops.
__device__ __attribute__((noinline)) FceSeeds fce_seeds(double theta, double theta_lo, double ea,
                                                        double eb, int carrier, int env) {
#pragma clang fp contract(off)   // see sincos_phase(): the explicit fma()s below must stay the only ones
  FceSeeds o;
  o.c = 1.0; o.s = 0.0; o.g = 1.0; o.r = 1.0;
  if (carrier == 1) {
    // phase = theta + theta_lo (the low word is the rounding error of the product W * (x - s_ref),
    // passed by the kernels that correct per sample and therefore need the EXACT seed phase)
    const double IPI_HI = 0.31830988618379069, IPI_LO = -1.9678676675182486e-17;
    const double xh = theta * IPI_HI;
    const double xl = fma(theta, IPI_HI, -xh) + fma(theta, IPI_LO, theta_lo * IPI_HI);
    const double n = rint(xh);
    double ss, cc;
    sincospi((xh - n) + xl, &ss, &cc);
    const bool odd = ((long long)n) & 1;
    o.s = odd ? -ss : ss;
    o.c = odd ? -cc : cc;
  }
  if (carrier == 2) o.c = erf(theta);   // closing erf multiplier: the running value E = erf(v)
  if (env) {
    o.g = exp(ea);
    o.r = exp(eb);
  }
  return o;
}

// Rounding correction (CORR): the phasor of sample k stands for the IDEAL uniform time
// x_ref + (kbase + k) * D and the exact phase there; the reference evaluates at NumPy's rounded
// grid value x_k (e_k = x_k - ideal_k, about one ulp of |t|) and rounds its phase,
// fl(w * fl(x_k - shift)) (rho_k, up to half an ulp of the phase).  Far from t = 0 a fast carrier
// makes both visible, so the carrier is corrected per sample to first order:
// cos(th + d) = cos th - d sin th, d = W e_k + rho_k (wfk_compile.cpp: corr_safe).
struct CorrCtx {
  double dj0;      // this lane's first sample index of the tile, as a double (exact)
  double xref;     // time (minus tshift) at which the op state was seeded exactly
  double step, t0, tshift, last, dlast;   // grid (dlast = n - 1 when the last sample is overridden, else -1)
  int kbase;       // lane-strides (of 64 samples) between the seed sample and this tile's first
};

// phase deviation of sample K from the ideal phasor: W * e_K + rho_K (see CorrCtx)
template <int K>
__device__ __forceinline__ double corr_delta(const CorrCtx& cc, double dj0, double D, double W,
                                             double wm, double sm) {
#pragma clang fp contract(off)
  const double dj = dj0 + (double)(64 * K);      // exact integer arithmetic in double
  const double m = dj * cc.step;                 // t[j] = fl(fl(j*step) + t0): two roundings
  double t = m + cc.t0;
  if (dj == cc.dlast) t = cc.last;     // (dlast = -1 without an overridden last sample: never equal, dj >= 0)
  t = t - cc.tshift;                   // (unconditional: t - 0.0 == t bit for bit; the reference's `if shift != 0`
                                       //  guards an allocation, not a rounding -- 3 instructions per sample here)
  const double eps = (t - cc.xref) - (double)(cc.kbase + K) * D;
  const double um = t - sm;                      // the reference's fl(x - shift) ...
  const double bb = um - t;
  const double eu = (t - (um - bb)) + (-sm - bb);   // ... and what it rounded away (TwoSum: x - shift = um + eu)
  const double pm = wm * um;                     // the reference's rounded phase fl(w * fl(x - shift))
  const double rho = -__builtin_fma(wm, um, -pm);   // pm - wm*um, exactly
  return __builtin_fma(W, eps, __builtin_fma(-wm, eu, rho));
}

// ENV: 0 none, 1 Gaussian with state in T, 2 Gaussian with state in double (float
// output whose state would leave float's exponent range).  DEG: 0, 1, or 3 (= 2..3).
template <typename T, int NS, int DEG, bool CARRIER, int ENV, bool CORR = false>
__device__ __forceinline__ void fce_loop(const double2* tab, const double* r, FceSeeds& sd,
                                         double u0, T (&acc)[NS],   // u0 by value: made opaque below
                                         const CorrCtx* cc = nullptr) {
  using S = typename std::conditional<ENV == 2, double, T>::type;
  constexpr int SB = 4;  // sub-batch: bounds the live LDS-table / temporary registers
  // The specialised loops share sub-expressions (u_k, the table loads); without this
  // opaque barrier GVN hoists all of them above the variant dispatch and every variant
  // pays ~100 live VGPRs for it.
  asm volatile("" : "+v"(u0) : : "memory");
  double dj0 = 0.0, wm = 0.0, sm = 0.0;
  if constexpr (CORR) {
    wm = r[WFK_FCE_WM];
    sm = r[WFK_FCE_SM];
    dj0 = cc->dj0;    // opaque per variant as well: the per-sample times are the same in every CORR
    asm volatile("" : "+v"(dj0));   // variant, and hoisted above the dispatch they would stay live (32 VGPRs)
  }
  const T c0 = (T)sd.c, s0 = (T)sd.s;
  const T A0 = (T)r[WFK_FCE_A], A1 = (T)r[WFK_FCE_A + 1], B0 = (T)r[WFK_FCE_B],
          B1 = (T)r[WFK_FCE_B + 1];
  const T A2 = DEG > 1 ? (T)r[WFK_FCE_A + 2] : (T)0, A3 = DEG > 1 ? (T)r[WFK_FCE_A + 3] : (T)0;
  const T B2 = DEG > 1 ? (T)r[WFK_FCE_B + 2] : (T)0, B3 = DEG > 1 ? (T)r[WFK_FCE_B + 3] : (T)0;
  const T ac = A0 * c0, as = A0 * s0;  // DEG == 0
  // DEG >= 1 without the rounding correction: the seed phasor folds into the polynomials once per
  // tile -- A(u) ck + B(u) sk with ck = c0 C - s0 S, sk = s0 C + c0 S is C P(u) + S Q(u) with
  // P_i = A_i c0 + B_i s0, Q_i = B_i c0 - A_i s0: per sample two Horner chains and two products
  // instead of the polynomials, the phasor rotation AND the combination (degree 1: 4 instead of 8).
  const bool fold = CARRIER && !CORR && DEG > 0;
  const T P0 = fold ? A0 * c0 + B0 * s0 : (T)0, P1 = fold ? A1 * c0 + B1 * s0 : (T)0;
  const T Q0 = fold ? B0 * c0 - A0 * s0 : (T)0, Q1 = fold ? B1 * c0 - A1 * s0 : (T)0;
  const T P2 = fold && DEG > 1 ? A2 * c0 + B2 * s0 : (T)0, P3 = fold && DEG > 1 ? A3 * c0 + B3 * s0 : (T)0;
  const T Q2 = fold && DEG > 1 ? B2 * c0 - A2 * s0 : (T)0, Q3 = fold && DEG > 1 ? B3 * c0 - A3 * s0 : (T)0;
  const double D = r[WFK_FCE_D];
  S g = (S)sd.g, rr = (S)sd.r;
  const S q = (S)r[WFK_FCE_Q];
  T u = (T)u0;
  const T Dt = (T)D;
  // phasor table entries are wave-wide LDS broadcasts; batch kb+1 is fetched while batch
  // kb computes (one batch of look-ahead: 16 more registers instead of an lgkmcnt stall
  // at the head of every sub-batch)
  double2 tb[2][SB];
  if (CARRIER) {
    WFK_EACH(SB, kk) tb[0][kk] = tab[kk]; WFK_END
  }
  WFK_EACH(NS / SB, kb)
    if constexpr (CARRIER && (kb + 1) * SB < NS) {
      WFK_EACH(SB, kk) tb[(kb + 1) & 1][kk] = tab[(kb + 1) * SB + kk]; WFK_END
    }
    WFK_EACH(SB, kk)
      constexpr int k = kb * SB + kk;
      T val;
      if constexpr (DEG == 0 && CARRIER && !CORR && ENV == 0) {
        // a bare carrier (the tones of a multiplexed pulse under a shared envelope): two fused multiply-adds
        // straight into the accumulator instead of product, fma and add
        const double2 cs = tb[kb & 1][kk];
        acc[k] = __builtin_fma(ac, (T)cs.x, acc[k]);
        acc[k] = __builtin_fma(-as, (T)cs.y, acc[k]);
        return;
      }
      if (DEG == 0) {
        if (CARRIER) {
          const double2 cs = tb[kb & 1][kk];
          if constexpr (CORR) {
            const T we = (T)corr_delta<k>(*cc, dj0, D, r[WFK_FCE_W], wm, sm);
            const T ck = c0 * (T)cs.x - s0 * (T)cs.y;
            const T sk = s0 * (T)cs.x + c0 * (T)cs.y;
            val = A0 * (ck - we * sk);
          } else {
            val = ac * (T)cs.x - as * (T)cs.y;
          }
        } else {
          val = A0;
        }
      } else if constexpr (CARRIER && !CORR) {
        const double2 cs = tb[kb & 1][kk];
        T pp, qq;
        if (DEG == 1) {
          pp = P1 * u + P0;
          qq = Q1 * u + Q0;
        } else {
          pp = ((P3 * u + P2) * u + P1) * u + P0;
          qq = ((Q3 * u + Q2) * u + Q1) * u + Q0;
        }
        u += Dt;
        val = pp * (T)cs.x + qq * (T)cs.y;
      } else {
        T pa, pb;
        if (DEG == 1) {
          pa = A1 * u + A0;
          pb = B1 * u + B0;
        } else {
          pa = ((A3 * u + A2) * u + A1) * u + A0;
          pb = ((B3 * u + B2) * u + B1) * u + B0;
        }
        u += Dt;
        if (CARRIER) {
          const double2 cs = tb[kb & 1][kk];
          T ck = c0 * (T)cs.x - s0 * (T)cs.y;
          T sk = s0 * (T)cs.x + c0 * (T)cs.y;
          if constexpr (CORR) {
            const T we = (T)corr_delta<k>(*cc, dj0, D, r[WFK_FCE_W], wm, sm);
            const T c1 = ck - we * sk;
            sk = sk + we * ck;
            ck = c1;
          }
          val = pa * ck + pb * sk;
        } else {
          val = pa;
        }
      }
      if (ENV) {
        acc[k] += val * (T)g;
        g *= rr;
        rr *= q;
      } else {
        acc[k] += val;
      }
    WFK_END
    __builtin_amdgcn_sched_barrier(0);
  WFK_END
  // advance the per-lane state by one wave tile (used by the lean kernel, which carries
  // it to the next tile; dead code elsewhere): NS recurrence steps already happened for
  // the envelope, the phasor turns by table entry NS.
  if (ENV) {
    sd.g = (double)g;
    sd.r = (double)rr;
  }
  if (CARRIER) {
    const double2 e = tab[NS];
    const double c = sd.c, sn = sd.s;
    sd.c = c * e.x - sn * e.y;
    sd.s = sn * e.x + c * e.y;
  }
}

template <typename T, int NS, int DEG, bool CARRIER, bool CORR = false>
__device__ __forceinline__ void fce_env(const double2* tab, const double* r, FceSeeds& sd,
                                        double u0, int env, T (&acc)[NS], const CorrCtx* cc = nullptr) {
  if (env == 0) fce_loop<T, NS, DEG, CARRIER, 0, CORR>(tab, r, sd, u0, acc, cc);
  else if (sizeof(T) == 8 || env == 1) fce_loop<T, NS, DEG, CARRIER, 1, CORR>(tab, r, sd, u0, acc, cc);
  else fce_loop<T, NS, DEG, CARRIER, 2>(tab, r, sd, u0, acc);   // (float output: CORR never set)
}

// exact per-lane seeds of one fused op at sample time x (already minus tshift)
// `fl`: the op's packed word (WFK_FCE_DEG), read once by the caller
template <bool EXACT = false>
__device__ __forceinline__ FceSeeds fce_make_seeds(const double* r, double x, int fl) {
  double v = (x - r[WFK_FCE_SG]) / r[WFK_FCE_SIGMA];
  const double Hh = r[WFK_FCE_H];
  if ((fl & 0x33) == 0x31) {
    // closing erf multiplier: E = erf(v) at the sample, the Gaussian state half a stride further on
    // (the midpoint of the first step, see fce_erfmul)
    const double vm = v + 0.5 * Hh;
    return fce_seeds(v, 0.0, -(vm * vm), -Hh * (2.0 * vm + Hh), 2, 1);
  }
  if (fl & WFK_FCE_EXPENV) v = 0.0;   // (exponential envelope: the seed exponents are linear, see below)
  const double sref = r[WFK_FCE_SREF], W = r[WFK_FCE_W];
  const double d = x - sref;
  const double th = W * d;
  double lo = 0.0;
  if (EXACT) {   // W * (x - s_ref) to twice the working precision: TwoSum of the difference, TwoProd
    const double bb = d - x;
    const double ed = (x - (d - bb)) + (-sref - bb);
    lo = fma(W, ed, fma(W, d, -th));
  }
  if (fl & WFK_FCE_EXPENV)   // g = exp(alpha (x - ref)), ratio exp(alpha D) (q = 1): SIGMA = alpha, SG = ref, H = alpha D
    return fce_seeds(th, lo, r[WFK_FCE_SIGMA] * (x - r[WFK_FCE_SG]), Hh, (fl >> 2) & 1, (fl >> 4) & 3);
  return fce_seeds(th, lo, -(v * v), -Hh * (2.0 * v + Hh), (fl >> 2) & 1, (fl >> 4) & 3);
}

// run one fused op over the wave tile from the given state; `wide_env`: keep the
// Gaussian state in double even for float output
template <typename T, int NS, bool CORR = false>
__device__ __forceinline__ void fce_eval(const double* blk, const double* r, FceSeeds& sd,
                                         double x, bool wide_env, T (&acc)[NS], int fl,
                                         const CorrCtx* cc = nullptr) {
  const int deg = fl & 3;
  const int carrier = (fl >> 2) & 1;
  int env = (fl >> 4) & 3;
  if (env && sizeof(T) == 4 && (wide_env || ((fl >> 6) & 1) == 0)) env = 2;
  const double2* tab = reinterpret_cast<const double2*>(blk + WFK_FCE_TABOFF(fl));
  const double u0 = x - r[WFK_FCE_SLIN];
  if constexpr (CORR && sizeof(T) == 8) {
    if (carrier && ((fl >> 7) & 1)) {   // this carrier's phase feels the grid rounding
      if (deg == 0) fce_env<T, NS, 0, true, true>(tab, r, sd, u0, env, acc, cc);
      else if (deg == 1) fce_env<T, NS, 1, true, true>(tab, r, sd, u0, env, acc, cc);
      else fce_env<T, NS, 3, true, true>(tab, r, sd, u0, env, acc, cc);
      return;
    }
  }
  if (carrier) {
    if (deg == 0) fce_env<T, NS, 0, true>(tab, r, sd, u0, env, acc);
    else if (deg == 1) fce_env<T, NS, 1, true>(tab, r, sd, u0, env, acc);
    else fce_env<T, NS, 3, true>(tab, r, sd, u0, env, acc);
  } else {
    if (deg == 0) fce_env<T, NS, 0, false>(tab, r, sd, u0, env, acc);
    else fce_env<T, NS, 3, false>(tab, r, sd, u0, env, acc);
  }
}

// closing pseudo-op of a piece whose carriers share one Gaussian envelope (WFK_FCE_ENV == 3):
// acc (and the imaginary accumulators) *= g_k; the state advances by the NS recurrence steps
template <typename T, int NS, bool CPLX>
__device__ __forceinline__ void fce_envmul(const double* r, FceSeeds& sd, T (&acc)[NS],
                                           T (&acci)[CPLX ? NS : 1]) {
  double g = sd.g, rr = sd.r;
  const double q = r[WFK_FCE_Q];
  WFK_EACH(NS, k)
    acc[k] *= (T)g;
    if constexpr (CPLX) acci[k] *= (T)g;
    g *= rr;
    rr *= q;
  WFK_END
  sd.g = g;
  sd.r = rr;
}

// closing pseudo-op "erf edge" (env == 3, deg == 1): everything accumulated so far is multiplied by
//   M_k = m0 + m1 erf(v_k),  v_k = (x_k - s) / sigma,  v_{k+1} = v_k + h  (h = lane stride / sigma)
// -- the flat-top pulse's edge, 0.5 + 0.5 erf, over its carriers.  erf advances by its own integral:
//   erf(v + h) - erf(v) = 2/sqrt(pi) exp(-vm^2) * [h + H2(vm) h^3/24 + H4(vm) h^5/1920 + H6(vm) h^7/322560 + ...]
// about the MIDPOINT vm = v + h/2 (odd orders vanish; Hn = Hermite polynomials), the Gaussian at the
// midpoints from the same two-multiplier recurrence the envelopes use and the bracket as a cubic
// in w = vm^2 with host-made coefficients (B0..B3).  Nine fp64 operations per sample instead of a
// libm erf (~100); the host admits the op for |h| <= 0.09 only, where the first neglected term is
// below 8e-15 per step (the state is reseeded exactly every WFK_LEAN_RESEED tiles).
template <typename T, int NS, bool CPLX>
__device__ __forceinline__ void fce_erfmul(const double* r, FceSeeds& sd, double x, T (&acc)[NS],
                                           T (&acci)[CPLX ? NS : 1]) {
  double E = sd.c, g = sd.g, rr = sd.r;
  const double q = r[WFK_FCE_Q], h = r[WFK_FCE_H];
  const double m0 = r[WFK_FCE_A], m1 = r[WFK_FCE_A + 1];
  const double p0 = r[WFK_FCE_B], p1 = r[WFK_FCE_B + 1], p2 = r[WFK_FCE_B + 2], p3 = r[WFK_FCE_B + 3];
  double vm = (x - r[WFK_FCE_SG]) / r[WFK_FCE_SIGMA] + 0.5 * h;
  WFK_EACH(NS, k)
    const T m = (T)fma(m1, E, m0);
    acc[k] *= m;
    if constexpr (CPLX) acci[k] *= m;
    const double w = vm * vm;
    E = fma(g, fma(fma(fma(p3, w, p2), w, p1), w, p0), E);
    g *= rr;
    rr *= q;
    vm += h;
  WFK_END
  sd.c = E;
  sd.g = g;
  sd.r = rr;
}

// A run of n bare carriers (WFK_FCE_BANK: degree 0, no envelope, no correction; the tones of a multiplexed pulse whose
// shared envelope the closing op applies): acc[k] += A0 (c_k cos.. ) for every tone in ONE rolled loop -- per tone and
// tile two state reads, two products with the amplitude, 2 FMAs per sample against the broadcast phasor table and
// the phasor advance.  The per-op path costs ~70 VALU + ~50 scalar instructions of dispatch and set-up per op and
// tile on top of the same arithmetic (PMC, ten tones: 80 VALU + 30 SALU instructions per sample; rocprof r04 multitone).
template <typename T, int NS>
__device__ __forceinline__ void fce_bank(const double* blk, const double* gblk, const double* rec0, int n, double* s_st, int lane,
                                         T (&acc)[NS]) {
  constexpr int SB = 4;
#pragma unroll 1
  for (int i = 0; i < n; ++i) {
    const double* r = rec0 + i * WFK_FCE_REC;
    const int fl = uni(WFK_FCE_WORD(r));
#ifndef WFK_BANK_LDS
    // the phasor table through the SCALAR cache (the block's image in global memory): the entries arrive in SGPRs and
    // the FMAs take them as scalar operands -- 17 wave-wide LDS broadcasts per tone and tile less, and the 32 VGPRs of
    // their double buffer (ten tones, same box: 2.15 -> 1.76 ms; -DWFK_BANK_LDS: the LDS form)
    const WFK_CONST double* tabd = reinterpret_cast<const WFK_CONST double*>(reinterpret_cast<uintptr_t>(gblk + WFK_FCE_TABOFF(fl)));
#else
    const double* tabd = blk + WFK_FCE_TABOFF(fl);
#endif
    auto tab = [&](int k) __attribute__((always_inline)) { return make_double2(tabd[2 * k], tabd[2 * k + 1]); };
    double* const st = s_st + WFK_FCE_STOFF(fl) + lane;
    const double c = st[0], sn = st[64];
    const double A0 = r[WFK_FCE_A];
    const T ac = (T)(A0 * c), as = (T)(A0 * sn);
    double2 tb[2][SB];
    WFK_EACH(SB, kk) tb[0][kk] = tab(kk); WFK_END
    WFK_EACH(NS / SB, kb)
      if constexpr ((kb + 1) * SB < NS) {
        WFK_EACH(SB, kk) tb[(kb + 1) & 1][kk] = tab((kb + 1) * SB + kk); WFK_END
      }
      WFK_EACH(SB, kk)
        constexpr int k = kb * SB + kk;
        const double2 cs = tb[kb & 1][kk];
        acc[k] = __builtin_fma(ac, (T)cs.x, acc[k]);
        acc[k] = __builtin_fma(-as, (T)cs.y, acc[k]);
      WFK_END
    WFK_END
    const double2 e = tab(NS);
    st[0] = c * e.x - sn * e.y;
    st[64] = sn * e.x + c * e.y;
  }
}

// closing pseudo-ops WITHOUT state (env == 3, deg == 2 | 3; lean kernel family 3): everything accumulated so far is
// multiplied by F(u_k), u_k = x - s + k D -- the envelope the piece's carriers share, where F has no recurrence form.
__device__ __forceinline__ double udbl(double v) {   // a wave-uniform double read from LDS: pinned to an SGPR pair
  const uint64_t b = (uint64_t)__double_as_longlong(v);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)b), hi = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));
  return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}

// deg == 2: np.interp over linspace knots with a finite table (reference _waveform.pyx:309-311), read as the
// continuous piecewise-linear function it is, in knot units: q = (x - start) * (m - 1) / (stop - start) clamped
// into [0, m - 1] (np.interp's constant continuation), value = f[floor q] + frac(q) * (f[floor q + 1] - f[floor q]).
// q advances along the lane by additions of D (m - 1) / (stop - start).  The host packs (f_j, f_{j+1} - f_j) pairs:
// one 16-byte gather per sample, eight samples' gathers in flight; 8 VALU instructions per sample next to it.
// (Admission, wfk_compile.cpp: the rounding of q -- seed, sixteen additions -- times the largest step, and the
// grid rounding times the largest slope, both stay inside the jitter budget.)
template <typename T, int NS, bool CPLX>
__device__ __forceinline__ void fce_tabmul(const double* r, const KArgs& a, double x, T (&acc)[NS],
                                           T (&acci)[CPLX ? NS : 1]) {
  const double qmax = udbl(r[WFK_FCE_A + 1]), Dq = udbl(r[WFK_FCE_B]);
  double q = ((x - r[WFK_FCE_SLIN]) - r[WFK_FCE_A]) * r[WFK_FCE_A + 2];
  const double2* tab = uniptr(reinterpret_cast<const double2*>(a.pool) + uni64((int64_t)r[WFK_FCE_A + 3]));
  constexpr int IB = NS % 8 == 0 ? 8 : 1;
#pragma unroll
  for (int k0 = 0; k0 < NS; k0 += IB) {
    double fr[IB];
    double2 e[IB];
#pragma unroll
    for (int kk = 0; kk < IB; ++kk) {
      const double qc = fmin(fmax(q, 0.0), qmax);
      fr[kk] = __builtin_amdgcn_fract(qc);
      e[kk] = tab[(uint32_t)(int)qc];
      q += Dq;
    }
    __builtin_amdgcn_sched_barrier(0);   // gathers above, uses below
#pragma unroll
    for (int kk = 0; kk < IB; ++kk) {
      const T m = (T)fma(fr[kk], e[kk].y, e[kk].x);
      acc[k0 + kk] *= m;
      if constexpr (CPLX) acci[k0 + kk] *= m;
    }
  }
}

// deg == 3: mollifier(r) (d = 0): exp(1 / (x^2 - 1) + 1), x = u / r, inside |x| < 1, else 0 (reference
// _waveform.pyx:359-363); Newton reciprocal and the inline exponential, as WFK_M_MOLL_REC
template <typename T, int NS, bool CPLX>
__device__ __forceinline__ void fce_mollmul(const double* r, double x, T (&acc)[NS], T (&acci)[CPLX ? NS : 1]) {
  const double u0 = x - r[WFK_FCE_SLIN];
  const double D = udbl(r[WFK_FCE_D]), ir = udbl(r[WFK_FCE_A]);
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    const double xx = fma((double)k, D, u0) * ir;
    const double q = fma(xx, xx, -1.0);
    const double qq = q < -1e-300 ? q : -1.0;                     // (outside the support: any harmless argument)
    const double v = exp_inline(rcp_nr(qq) + 1.0);
    const T m = (T)(q < 0.0 ? v : 0.0);
    acc[k] *= m;
    if constexpr (CPLX) acci[k] *= m;
    if ((k & 3) == 3) __builtin_amdgcn_sched_barrier(0);
  }
}

// WFK_FCE_OWNMUL (family 4): a plain carrier op whose term carries an envelope of its own,
//   acc[k] += F(u_k) A0 cos(th_k)        (A0 alone without a carrier),
// F a finite INTERP table read as in fce_tabmul or a mollifier as in fce_mollmul.  Unlike the closing multipliers it
// touches nothing but its own term, so a piece may hold any number of them: overlapping pulses of different envelopes.
template <typename T, int NS>
__device__ __forceinline__ void fce_ownmul(const double* blk, const double* r, const KArgs& a, double x, double* st,
                                           int fl, T (&acc)[NS]) {
  const bool car = (fl & 4) != 0;
  double c = 1.0, sn = 0.0;
  if (car) { c = st[0]; sn = st[64]; }
  const double A0 = r[WFK_FCE_A];
  const double ac = A0 * c, as = A0 * sn;
  const double2* ptab = reinterpret_cast<const double2*>(blk + WFK_FCE_TABOFF(fl));
  if (uni((int)r[WFK_FCE_B + 3]) == 0) {
    const double qmax = udbl(r[WFK_FCE_A + 1]), Dq = udbl(r[WFK_FCE_B + 1]);
    double q = ((x - r[WFK_FCE_SLIN]) - r[WFK_FCE_B + 2]) * r[WFK_FCE_A + 2];
    const double2* tab = uniptr(reinterpret_cast<const double2*>(a.pool) + uni64((int64_t)r[WFK_FCE_A + 3]));
    constexpr int IB = NS % 8 == 0 ? 8 : 1;
#pragma unroll
    for (int k0 = 0; k0 < NS; k0 += IB) {
      double fr[IB];
      double2 e[IB];
#pragma unroll
      for (int kk = 0; kk < IB; ++kk) {
        const double qc = fmin(fmax(q, 0.0), qmax);
        fr[kk] = __builtin_amdgcn_fract(qc);
        e[kk] = tab[(uint32_t)(int)qc];
        q += Dq;
      }
      __builtin_amdgcn_sched_barrier(0);   // gathers above, uses below
#pragma unroll
      for (int kk = 0; kk < IB; ++kk) {
        double val = ac;
        if (car) {
          const double2 cs = ptab[k0 + kk];
          val = fma(ac, cs.x, -(as * cs.y));
        }
        acc[k0 + kk] += (T)(val * fma(fr[kk], e[kk].y, e[kk].x));
      }
    }
  } else {
    const double u0 = x - r[WFK_FCE_SLIN];
    const double D = udbl(r[WFK_FCE_D]), ir = udbl(r[WFK_FCE_A + 1]);
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      const double xx = fma((double)k, D, u0) * ir;
      const double qv = fma(xx, xx, -1.0);
      const double qq = qv < -1e-300 ? qv : -1.0;
      const double v = exp_inline(rcp_nr(qq) + 1.0);
      double val = ac;
      if (car) {
        const double2 cs = ptab[k];
        val = fma(ac, cs.x, -(as * cs.y));
      }
      acc[k] += (T)(qv < 0.0 ? val * v : 0.0);
      if ((k & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
  }
  if (car) {
    const double2 e = ptab[NS];
    st[0] = c * e.x - sn * e.y;
    st[64] = sn * e.x + c * e.y;
  }
}

// ---- fused chirp op (lean kernel, family 2): E_k (A(u_k) cos th_k + B(u_k) sin th_k) with a QUADRATIC
// phase, th(tau) = K tau^2 + W' tau + phi0 about tau = t' - tref (reference LINEARCHIRP,
// _waveform.pyx:323-324, times whatever carriers the term multiplies it with).  Along a lane (stride D)
//   z_{k+1} = z_k w_k,   w_{k+1} = w_k v,   v = exp(i 2 K D^2)  (constant, from the host)
// -- the complex twin of the Gaussian recurrence: 8 flops per sample instead of a libm sin.
struct ChirpSeeds { double c, s, wc, ws, g, r; };

__device__ __attribute__((noinline)) ChirpSeeds chirp_seeds(double th, double dth, double ea, double eb, int env) {
  ChirpSeeds o;
  const double2 z = sincos_phase(th), w = sincos_phase(dth);
  o.c = z.x; o.s = z.y; o.wc = w.x; o.ws = w.y;
  o.g = 1.0; o.r = 1.0;
  if (env) {
    o.g = exp(ea);
    o.r = exp(eb);
  }
  return o;
}

__device__ __forceinline__ ChirpSeeds chirp_make_seeds(const double* r, double x, int fl) {
  const double tau = x - r[WFK_FCE_SREF], K = r[WFK_FCE_WM], W = r[WFK_FCE_W], D = r[WFK_FCE_D];
  const double th = fma(K * tau, tau, fma(W, tau, r[WFK_FCE_SM]));
  const double dth = D * fma(K, 2.0 * tau + D, W);                 // th(tau + D) - th(tau)
  const double Hh = r[WFK_FCE_H];
  if (fl & WFK_FCE_EXPENV)
    return chirp_seeds(th, dth, r[WFK_FCE_SIGMA] * (x - r[WFK_FCE_SG]), Hh, (fl >> 4) & 3);
  const double v = (x - r[WFK_FCE_SG]) / r[WFK_FCE_SIGMA];
  return chirp_seeds(th, dth, -(v * v), -Hh * (2.0 * v + Hh), (fl >> 4) & 3);
}

// one chirp op over the wave tile; the state (z, w, g, r) is advanced by NS strides for the next tile
template <typename T, int NS, bool ENV>
__device__ __forceinline__ void chirp_loop(const double* r, ChirpSeeds& sd, double u0, T (&acc)[NS]) {
  asm volatile("" : "+v"(u0) : : "memory");      // (see fce_loop: keeps the variants' prologues apart)
  const double A0 = r[WFK_FCE_A], A1 = r[WFK_FCE_A + 1], A2 = r[WFK_FCE_A + 2], A3 = r[WFK_FCE_A + 3];
  const double B0 = r[WFK_FCE_B], B1 = r[WFK_FCE_B + 1], B2 = r[WFK_FCE_B + 2], B3 = r[WFK_FCE_B + 3];
  const double vc = r[WFK_FCE_TAB], vs = r[WFK_FCE_F32OK], D = r[WFK_FCE_D], q = r[WFK_FCE_Q];
  double c = sd.c, s = sd.s, wc = sd.wc, ws = sd.ws, g = sd.g, rr = sd.r, u = u0;
  WFK_EACH(NS, k)
    const double pa = fma(fma(fma(A3, u, A2), u, A1), u, A0);
    const double pb = fma(fma(fma(B3, u, B2), u, B1), u, B0);
    double val = fma(pa, c, pb * s);
    if constexpr (ENV) val *= g;
    acc[k] += (T)val;
    const double cn = fma(c, wc, -(s * ws));
    s = fma(s, wc, c * ws);
    c = cn;
    const double wn = fma(wc, vc, -(ws * vs));
    ws = fma(ws, vc, wc * vs);
    wc = wn;
    u += D;
    if constexpr (ENV) {
      g *= rr;
      rr *= q;
    }
  WFK_END
  sd.c = c; sd.s = s; sd.wc = wc; sd.ws = ws; sd.g = g; sd.r = rr;
}

template <typename T, int NS, bool SLICE>
__device__ __forceinline__ void apply_fce(const double* blk, const double* r, const KArgs& a,
                                          double tshift, int64_t j0, T (&acc)[NS]) {
  double x = grid_time<SLICE>(a, j0);
  if (tshift != 0.0) x = x - tshift;
  const int fl = uni(WFK_FCE_WORD(r));
  FceSeeds sd = fce_make_seeds(r, x, fl);
  fce_eval<T, NS>(blk, r, sd, x, false, acc, fl);
}

// ---- fused ops evaluated POINTWISE (tlist plans: arbitrary sorted sample times) ------------------
// The host's fusion pass turns the terms of a piece into groups  E(t') (A(u) cos th + B(u) sin th),
// th = W (t' - s_ref), u = t' - s_lin, E = 1 | exp(-((t' - s_g)/sigma)^2) | exp(alpha (t' - ref))  -- on a
// grid those advance by recurrences; on a time LIST every sample is evaluated from its own time, but still
// as ONE sincos + ONE exp per group instead of one libm call per factor (a DRAG pulse: 1 + 1 against 3 + 3).

// (cos, sin) of a phase |th| <= 1.6e6 (host-checked, WFK_FCE_TLSMALL): pi/2 in three parts, reduced with
// fma (n P1 is formed exactly inside the fma: one rounding per step, absolute error of r <= 1.2e-16),
// then the fdlibm kernels on |r| <= pi/4.  ~35 VALU instructions against ~130 of libm's sincos.
__device__ __forceinline__ void sincos_small(double th, double& c, double& s) {
  const double n = rint(th * 0.63661977236758138);
  double r = fma(-n, 1.57079632679489655800e+00, th);
  r = fma(-n, 6.12323399573676603587e-17, r);
  const double z = r * r;
  double ps = fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
  ps = fma(z, ps, 2.75573137070700676789e-06);
  ps = fma(z, ps, -1.98412698298579493134e-04);
  ps = fma(z, ps, 8.33333333332248946124e-03);
  ps = fma(z, ps, -1.66666666666666324348e-01);
  const double sr = fma(z * r, ps, r);
  double pc = fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
  pc = fma(z, pc, -2.75573143513906633035e-07);
  pc = fma(z, pc, 2.48015872894767294178e-05);
  pc = fma(z, pc, -1.38888888888741095749e-03);
  pc = fma(z, pc, 4.16666666666666019037e-02);
  const double cr = fma(z * z, pc, fma(z, -0.5, 1.0));
  const int q = (int)n;
  const double a = (q & 1) ? sr : cr, b = (q & 1) ? cr : sr;     // |cos|, |sin| of the quadrant
  c = ((q + 1) & 2) ? -a : a;
  s = (q & 2) ? -b : b;
}

// value of one group at time x (on the channel's own axis); everything wave-uniform comes in by value
// One group over the lane's NS samples, specialised by SHAPE -- polynomial degree <= 1 or cubic, carrier or not,
// no envelope / Gaussian / exponential -- so that the per-sample code is straight line: the eight samples'
// dependent chains (range reduction -> polynomials, exponent -> polynomial -> ldexp) are interleaved by the
// scheduler instead of sitting in eight separate basic blocks behind wave-uniform branches, and the group's
// coefficients are read from LDS once per group, not per sample.  ENV: 0 none, 1 Gaussian, 2 exponential.
template <typename T, int NS, bool CPLX, bool CUBIC, bool CARRIER, int ENV>
__device__ __forceinline__ void fce_point_shape(const double* r, const double (&x)[NS], T (&acc)[NS],
                                                T (&acci)[CPLX ? NS : 1], bool imag) {
  // (wave-uniform values read from LDS land in VGPRs: pinned to SGPR pairs, they leave the registers to the samples)
  auto ud = [](double v) __attribute__((always_inline)) {
    return __hiloint2double(uni(__double2hiint(v)), uni(__double2loint(v)));
  };
  const double A0 = ud(r[WFK_FCE_A]), A1 = ud(r[WFK_FCE_A + 1]), B0 = ud(r[WFK_FCE_B]), B1 = ud(r[WFK_FCE_B + 1]);
  const double A2 = CUBIC ? ud(r[WFK_FCE_A + 2]) : 0.0, A3 = CUBIC ? ud(r[WFK_FCE_A + 3]) : 0.0;
  const double B2 = CUBIC ? ud(r[WFK_FCE_B + 2]) : 0.0, B3 = CUBIC ? ud(r[WFK_FCE_B + 3]) : 0.0;
  const double W = ud(r[WFK_FCE_W]), sref = ud(r[WFK_FCE_SREF]), slin = ud(r[WFK_FCE_SLIN]);
  const double sg = ud(r[WFK_FCE_SG]), ea = ud(ENV == 1 ? r[WFK_FCE_H] : r[WFK_FCE_SIGMA]);   // 1 / sigma | alpha
  constexpr int PB = NS % 4 == 0 ? 4 : 1;    // samples interleaved at a time (2 / 4 / 8: the same 0.58 ms on 64 x 2e6; bounds the live values)
#pragma unroll
  for (int k0 = 0; k0 < NS; k0 += PB) {
  double val[PB];
#pragma unroll
  for (int kk = 0; kk < PB; ++kk) {
    const int k = k0 + kk;
    const double u = x[k] - slin;
    double A, Bq;
    if constexpr (CUBIC) {
      A = fma(fma(fma(A3, u, A2), u, A1), u, A0);
      Bq = fma(fma(fma(B3, u, B2), u, B1), u, B0);
    } else {
      A = fma(A1, u, A0);                        // (degree 0: A1 = B1 = 0 in the record)
      Bq = fma(B1, u, B0);
    }
    double v = A;
    if constexpr (CARRIER) {
      double c, sn;
      sincos_small(W * (x[k] - sref), c, sn);
      v = fma(Bq, sn, A * c);
    }
    if constexpr (ENV != 0) {
      double q = (x[k] - sg) * ea;               // Gaussian: (t' - s_g) / sigma; exponential: alpha (t' - ref)
      if constexpr (ENV == 1) q = -(q * q);
      v *= exp_inline(q);
    }
    val[kk] = v;
  }
#pragma unroll
  for (int kk = 0; kk < PB; ++kk) {
    const int k = k0 + kk;
    if constexpr (CPLX) {
      if (imag) acci[k] += (T)val[kk];
      else acc[k] += (T)val[kk];
    } else {
      acc[k] += (T)val[kk];
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  }
}

// (the pointwise-ops-only build of the general kernel; a time-list plan that also holds generic terms runs as two
// launches: this build takes the fully fused pieces, the build with the direct tier the others)
template <typename T, int NS, bool CPLX>
__device__ __forceinline__ void fce_point(const double* r, const double (&x)[NS], T (&acc)[NS],
                                          T (&acci)[CPLX ? NS : 1]) {
  const int fl = uni(WFK_FCE_WORD(r));
  const int deg = fl & 3, env = (fl >> 4) & 3;
  const bool carrier = (fl & 4) != 0, imag = (fl & 8) != 0;
  if (imag && !CPLX) return;                       // a real-output launch drops the imaginary groups
  if (env == 3) {
    // closing op of a piece whose carriers share one Gaussian: multiply what they accumulated by it
    const double sg = r[WFK_FCE_SG], isig = r[WFK_FCE_H];
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      const double v = (x[k] - sg) * isig;
      const T e = (T)exp_inline(-(v * v));
      acc[k] *= e;
      if constexpr (CPLX) acci[k] *= e;
    }
    return;
  }
  const int ev = env == 0 ? 0 : ((fl & WFK_FCE_EXPENV) ? 2 : 1);
  if (carrier && !(fl & WFK_FCE_TLSMALL)) {
    // a phase beyond 1.6e6 rad (never with the rounding budget of a time list unless s_ref lies far outside the
    // piece): the two-term 1/pi reduction, out of line, sample by sample
    const double W = r[WFK_FCE_W], sref = r[WFK_FCE_SREF], slin = r[WFK_FCE_SLIN], sg = r[WFK_FCE_SG];
    const double ea = ev == 1 ? r[WFK_FCE_H] : r[WFK_FCE_SIGMA];
#pragma unroll 1
    for (int k = 0; k < NS; ++k) {
      const double u = x[k] - slin;
      double A = r[WFK_FCE_A + 3], Bq = r[WFK_FCE_B + 3];
      for (int i = 2; i >= 0; --i) { A = fma(A, u, r[WFK_FCE_A + i]); Bq = fma(Bq, u, r[WFK_FCE_B + i]); }
      const double2 cs = sincos_phase(W * (x[k] - sref));
      double v = fma(Bq, cs.y, A * cs.x);
      if (ev) { double q = (x[k] - sg) * ea; if (ev == 1) q = -(q * q); v *= exp_inline(q); }
      // (NS is 1 or 8: a rolled loop over a register array needs compile-time indices)
      static_for<NS>([&](auto kk) __attribute__((always_inline)) {
        if (decltype(kk)::value == k) {
          if constexpr (CPLX) { if (imag) acci[decltype(kk)::value] += (T)v; else acc[decltype(kk)::value] += (T)v; }
          else acc[decltype(kk)::value] += (T)v;
        }
      });
    }
    return;
  }
  const bool cubic = deg > 1;
#define WFK_PT(CU, CA, EN) fce_point_shape<T, NS, CPLX, CU, CA, EN>(r, x, acc, acci, imag)
  if (!cubic) {
    if (carrier) { if (ev == 1) WFK_PT(false, true, 1); else if (ev == 2) WFK_PT(false, true, 2); else WFK_PT(false, true, 0); }
    else { if (ev == 1) WFK_PT(false, false, 1); else if (ev == 2) WFK_PT(false, false, 2); else WFK_PT(false, false, 0); }
  } else {
    if (carrier) { if (ev == 1) WFK_PT(true, true, 1); else if (ev == 2) WFK_PT(true, true, 2); else WFK_PT(true, true, 0); }
    else { if (ev == 1) WFK_PT(true, false, 1); else if (ev == 2) WFK_PT(true, false, 2); else WFK_PT(true, false, 0); }
  }
#undef WFK_PT
}

template <typename T> struct OutOps;
template <> struct OutOps<double> {
  using Real = double;
  using Cplx = double2;
};
template <> struct OutOps<float> {
  using Real = float;
  using Cplx = float2;
};

template <typename T>
__device__ __forceinline__ T clip_np(T v, T lo, T hi) {
  // np.clip: NaN propagates, otherwise min(max(v, lo), hi)
  v = v < lo ? lo : v;
  v = v > hi ? hi : v;
  return v;
}

// np.clip of a COMPLEX value against real bounds (reference _waveform.pyx:162 on a complex part):
// minimum(maximum(v, lo + 0j), hi + 0j) in NumPy's lexicographic order -- real part first, then the
// imaginary part -- so a value whose real part leaves [lo, hi] becomes the bare bound (imaginary part 0),
// and on the bound itself the sign of the imaginary part decides.  NaN propagates (comparisons false).
template <typename T>
__device__ __forceinline__ void clip_np_cplx(T& re, T& im, T lo, T hi) {
  if (re < lo || (re == lo && im < (T)0)) { re = lo; im = (T)0; }
  if (re > hi || (re == hi && im > (T)0)) { re = hi; im = (T)0; }
}

// write one wave tile of one piece: clip (evaluated pieces only), + offset, optional
// accumulate into `out`; lanes outside [P.start, P.stop) keep their hands off.
// `tr`/`tc` point at the tile's first sample (wave-uniform => scalar base + lane offset).
// PLAIN = the tile lies inside the piece, no clip, no accumulate: add + store per sample.
// NT: non-temporal stores.  A loss for the plain lean kernel (headline 3.45 -> 3.52 ms, C3 0.199 -> 0.207 on one
// box) but a gain for its CORR variant, which runs fewer waves per SIMD and keeps re-reading nothing the stream
// could evict (dense far-from-origin workload 0.93 -> 0.84 ms, same box): used there only.
// E: element type of the output (T, or float under double arithmetic: the "wide" builds of the general kernel)
template <typename T, bool CPLX, int NS, bool PLAIN, bool NT = false, typename E = T>
__device__ __forceinline__ void store_tile_impl(const KArgs& a, const DevChannel& C,
                                                const DevPiece& P, typename OutOps<E>::Real* tr,
                                                typename OutOps<E>::Cplx* tc, int64_t w0, int lane,
                                                const T (&acc)[NS],
                                                const T (&acci)[CPLX ? NS : 1]) {
  using OutC = typename OutOps<E>::Cplx;
  constexpr int WT = 64 * NS;
  const T base = (T)C.offset;
  const bool clip = !PLAIN && C.do_clip && (P.flags & WFK_PF_HAS_TERMS);
  const bool accum = !PLAIN && a.accumulate != 0;
  const T lo = (T)C.clip_lo, hi = (T)C.clip_hi;
  const int lo_l = (int)(P.start - w0 > 0 ? P.start - w0 : 0);         // tile-relative range
  const int hi_l = (int)(P.stop - w0 < WT ? P.stop - w0 : WT);
  WFK_EACH(NS, k)
    const int o = lane + 64 * k;
    if (PLAIN || (o >= lo_l && o < hi_l)) {
      T v = acc[k];
      T vi = (T)0;
      if constexpr (CPLX) vi = acci[k];
      if (clip) {
        if constexpr (CPLX) clip_np_cplx(v, vi, lo, hi);
        else v = clip_np(v, lo, hi);
      }
      v += base;
      if constexpr (CPLX) {
        if (accum) {
          const OutC old = tc[o];
          v += (T)old.x;
          vi += (T)old.y;
        }
        OutC w;
        w.x = (E)v;
        w.y = (E)vi;
        if constexpr (NT) {
          __builtin_nontemporal_store(w.x, &tc[o].x);
          __builtin_nontemporal_store(w.y, &tc[o].y);
        } else {
          tc[o] = w;
        }
      } else {
        if (accum) v += (T)tr[o];
        if constexpr (NT) __builtin_nontemporal_store((E)v, &tr[o]);
        else tr[o] = (E)v;
      }
    }
  WFK_END
}

template <typename T, bool CPLX, int NS, bool NT = false, typename E = T>
__device__ __forceinline__ void store_tile(const KArgs& a, const DevChannel& C, const DevPiece& P,
                                           typename OutOps<E>::Real* tr,
                                           typename OutOps<E>::Cplx* tc, int64_t w0, int lane,
                                           const T (&acc)[NS], const T (&acci)[CPLX ? NS : 1]) {
  const bool full = P.start <= w0 && P.stop >= w0 + 64 * NS;
  const bool clip = C.do_clip && (P.flags & WFK_PF_HAS_TERMS);
  if (full && !clip && !a.accumulate)
    store_tile_impl<T, CPLX, NS, true, NT, E>(a, C, P, tr, tc, w0, lane, acc, acci);
  else
    store_tile_impl<T, CPLX, NS, false, NT, E>(a, C, P, tr, tc, w0, lane, acc, acci);
}

// XCD-aware workgroup -> chunk map.  Workgroups are dealt round-robin to the 8 XCDs
// (workgroup b runs on XCD b % 8), so a plain chunk = b makes every XCD touch every region
// of the output at once.  Give XCD x the x-th contiguous eighth of the chunk list instead,
// walked in order: each XCD's L2 / TLB then sees one compact, linearly advancing write
// window (tools/attic/store_pattern6.hip: 5.86 -> 6.26 TB/s for this walk with stores only).
// Returns -1 for the padding workgroups of the rounded-up grid.
__device__ __forceinline__ int64_t xcd_chunk(const KArgs& a) {
  const int64_t b = blockIdx.x;
  const int64_t per = (a.n_chunks + 7) >> 3;
  const int64_t c = (b & 7) * per + (b >> 3);
  return c < a.n_chunks ? a.chunk_base + c : -1;
}

// ---- lean kernel: fully fused plans ----------------------------------------------------
// One wave per workgroup owns `tiles_per_chunk` CONSECUTIVE wave tiles of one channel.
// Per piece (<= WFK_LEAN_OPS fused ops, one parameter block) the per-lane op state
// (phasor c,s and Gaussian g,r) lives in LDS and is carried from tile to tile: advancing it
// costs 4 FMAs per op per tile; exact libm seeds are taken when a piece is entered and
// every WFK_LEAN_RESEED tiles, in a phase where no accumulator is live (so the libm call
// does not inflate the kernel's register allocation).  No barriers between waves at all.
// FAM: op families compiled into this instantiation -- 0: carrier / envelope ops only (every BASELINE
// config), 1: + the closing ops (erf edges, shared envelopes), 2: + chirps, 3: + the stateless closing multipliers
// (INTERP tables, mollifiers), 4: + carrier terms under envelopes of their own (several envelopes per piece: a family
// of its own so that the single-envelope shapes keep family 3's register allocation: 161 VGPRs against 167 + 2 spills,
// 1-2 % on bench.py direct_interp / direct_mollifier).  The host picks the smallest
// family a plan needs, so a shape added to one family cannot move the register allocation and code
// layout of the others (round 2 took the chirp op out again for exactly that: inlined into the one
// kernel it cost the multi-tone workloads 4-9 %).
template <typename T, bool CPLX, int NS, bool CORR, int FAM>
#ifndef WFK_LEAN_WAVES
#define WFK_LEAN_WAVES 3   // occupancy target (waves per SIMD) for the register allocator
#endif
#ifndef WFK_LEAN_WAVES_F32
#define WFK_LEAN_WAVES_F32 4   // fp32: 120 VGPRs, 10 KB LDS per wave -> 4 waves per SIMD (latency-bound kernel)
#endif
// (complex outputs carry two accumulator sets: one wave per SIMD fewer, no spills)
// (the CORR variant recomputes every sample's exact grid time next to the op: one wave fewer too)
__global__ void __launch_bounds__(64, (sizeof(T) == 4 ? WFK_LEAN_WAVES_F32 : WFK_LEAN_WAVES) - (CPLX ? 1 : 0) -
                                          (CORR && sizeof(T) == 8 && !CPLX ? 1 : 0))
wfk_sample_lean(const KArgs a) {
  // LDS is sized per plan (dynamic): the parameter block (a.lean_par doubles) followed by the
  // per-lane state of the plan's largest piece, (c, s, g, r) x 64 lanes per op.  Plans with up to
  // four ops per piece (all BASELINE configs) take 10 KB per wave; a ten-tone multiplexed pulse
  // takes 25 KB instead of falling back to the general kernel.
  extern __shared__ __attribute__((aligned(16))) double s_dyn[];
  double* const s_par = s_dyn;
  double* const s_st = s_dyn + a.lean_par;      // per op: [c | s][64] and / or [g | r][64] (WFK_FCE_STOFF)
  constexpr int WT = 64 * NS;
  using OutR = typename OutOps<T>::Real;
  using OutC = typename OutOps<T>::Cplx;

  const int lane = threadIdx.x;
  const int64_t chunk = xcd_chunk(a);
  if (chunk < 0) return;
  const int ch = uni((int)(chunk / a.chunks_per_ch));
  const int64_t cc = uni64(chunk - (int64_t)ch * a.chunks_per_ch);
  const DevChannel C = a.channels[ch];
  int p = uni(a.chunk_first[chunk]);
  DevPiece cur = load_piece(a.pieces + p);
  int64_t staged = -1;
  int state_piece = -1;       // piece whose op state sits in LDS ...
  int64_t state_w0 = -1;      // ... valid for the tile that starts here
  int since_seed = 0;
  int cs_age = 0;             // tiles since the pure-phasor ops of the piece in LDS were seeded exactly

  OutR* const outr = uniptr(reinterpret_cast<OutR*>(a.out) + (int64_t)ch * a.ch_stride);
  OutC* const outc = uniptr(reinterpret_cast<OutC*>(a.out) + (int64_t)ch * a.ch_stride);

  for (int tt = 0; tt < a.tiles_per_chunk; ++tt) {
    const int64_t w0 = uni64((cc * a.tiles_per_chunk + tt) * WT);
    if (w0 >= a.n) break;
    const int64_t w1 = w0 + WT < a.n ? w0 + WT : a.n;
    const int64_t j0 = w0 + lane;
    // the piece record stays in SGPRs from tile to tile: a tile inside the current piece
    // (the common case: pieces are ~100 tiles long) issues no table load at all
    while (p < C.piece_end - 1 && cur.stop <= w0) {
      p = uni(p + 1);
      cur = load_piece(a.pieces + p);
    }

    for (int q = p;;) {
      const DevPiece P = cur;
      if (P.start >= w1) break;
      T acc[NS], acci[CPLX ? NS : 1];
      WFK_EACH(NS, k) acc[k] = (T)0; WFK_END
      WFK_EACH(CPLX ? NS : 1, k) acci[k] = (T)0; WFK_END
      // mixed plans: pieces with generic terms belong to the general kernel's launch
      const bool mine = !a.mixed || P.n_blk == 0 || (P.flags & WFK_PF_LEAN);
      if (P.n_blk != 0 && mine) {
        if (P.par_off != staged) {
          __syncthreads();
          for (int i = lane; i < P.first_len; i += 64) s_par[i] = a.params[P.par_off + i];
          __syncthreads();
          staged = P.par_off;
        }
        const int nops = uni((int)s_par[1]);
        double x = grid_time(a, j0);
        if (C.tshift != 0.0) x = x - C.tshift;
        const bool valid = state_piece == q && state_w0 == w0;
        const bool carried = valid && since_seed < a.reseed;
        if (!carried) {
          // seed phase: libm, nothing else live.  A phasor that is only advanced once per tile (seed x table entry per
          // sample, one complex product per tile) drifts by ~1e-16 per TILE: on a periodic reseed -- there for the
          // per-sample envelope recurrences -- ops that carry nothing but a phasor are skipped WFK_LEAN_RESEED_CS
          // tiles long (ten tones per pulse: 10 of 11 seeds, ~9 VALU instructions per sample)
          bool seed_cs = true;
          if constexpr (!CORR) {
            seed_cs = !valid || cs_age >= WFK_LEAN_RESEED_CS;
          }
          if (seed_cs) cs_age = 0;
          for (int op = 0; op < nops; ++op) {
            const double* srec = s_par + WFK_BLK_HDR + op * WFK_FCE_REC;
            const int sfl = uni(WFK_FCE_WORD(srec));
            if constexpr (!CORR) {
              if (!seed_cs && (sfl & (WFK_FCE_HAS_CS | WFK_FCE_HAS_GR | WFK_FCE_CHIRP)) == WFK_FCE_HAS_CS) continue;
            }
            if constexpr (FAM >= 2) {
              if (sfl & WFK_FCE_CHIRP) {      // state: (c, s), the step phasor (wc, ws), then (g, r)
                const ChirpSeeds cd = chirp_make_seeds(srec, x, sfl);
                double* ct = s_st + WFK_FCE_STOFF(sfl) + lane;
                ct[0] = cd.c; ct[64] = cd.s; ct[128] = cd.wc; ct[192] = cd.ws;
                if (sfl & WFK_FCE_HAS_GR) { ct[256] = cd.g; ct[320] = cd.r; }
                continue;
              }
            }
            if constexpr (FAM >= 3) {
              if ((sfl & 0x32) == 0x32) continue;      // stateless closing multiplier: nothing to seed
            }
            const FceSeeds sd = fce_make_seeds<CORR>(srec, x, sfl);
            double* st = s_st + WFK_FCE_STOFF(sfl) + lane;
            if (sfl & WFK_FCE_HAS_CS) {
              st[0] = sd.c;
              st[64] = sd.s;
              st += 128;
            }
            if (sfl & WFK_FCE_HAS_GR) {
              st[0] = sd.g;
              st[64] = sd.r;
            }
          }
          since_seed = 0;
        }
        CorrCtx cc;
        if constexpr (CORR) {
          // the op state was seeded exactly `since_seed` tiles ago, at this lane's sample there
          const int64_t js = j0 - (int64_t)since_seed * WT;
          double xs = grid_time(a, js);
          if (C.tshift != 0.0) xs = xs - C.tshift;
          cc.dj0 = (double)(j0 + a.i0);    // index in the caller's FULL grid (wfk_grid.i0): the times are formed from it
          cc.xref = xs;
          cc.step = a.step; cc.t0 = a.t0; cc.tshift = C.tshift; cc.last = a.last;
          cc.dlast = a.has_last ? (double)(a.i0 + a.n - 1) : -1.0;
          cc.kbase = since_seed * NS;
        }
        for (int op = 0; op < nops; ++op) {
          FceSeeds sd;
          const double* rec = s_par + WFK_BLK_HDR + op * WFK_FCE_REC;
          // an op of the imaginary part (complex amplitudes) adds into acci; a real-output
          // launch of such a channel keeps the real part only, like WaveVStack's `.real`
          const int fl = uni(WFK_FCE_WORD(rec));        // packed op word: one 32-bit read for all flags
          // per-lane state: (c, s) and / or (g, r), 64 doubles each, only what the op has
          double* const st = s_st + WFK_FCE_STOFF(fl) + lane;
          if constexpr (FAM >= 2) {
            if (fl & WFK_FCE_CHIRP) {
              ChirpSeeds cd;
              cd.c = st[0]; cd.s = st[64]; cd.wc = st[128]; cd.ws = st[192];
              cd.g = 1.0; cd.r = 1.0;
              const bool cenv = (fl & WFK_FCE_HAS_GR) != 0;
              if (cenv) { cd.g = st[256]; cd.r = st[320]; }
              const double cu0 = x - rec[WFK_FCE_SLIN];
              if (fl & 8) {
                if constexpr (CPLX) {
                  if (cenv) chirp_loop<T, NS, true>(rec, cd, cu0, acci); else chirp_loop<T, NS, false>(rec, cd, cu0, acci);
                }
              } else {
                if (cenv) chirp_loop<T, NS, true>(rec, cd, cu0, acc); else chirp_loop<T, NS, false>(rec, cd, cu0, acc);
              }
              st[0] = cd.c; st[64] = cd.s; st[128] = cd.wc; st[192] = cd.ws;
              if (cenv) { st[256] = cd.g; st[320] = cd.r; }
              continue;
            }
          }
          if constexpr (FAM >= 1) {
            if (fl & WFK_FCE_BANK) {        // a run of bare carriers: one compact loop over its tones
              const int nb = uni((int)rec[WFK_FCE_SIGMA]);
              if (fl & 8) {
                if constexpr (CPLX) fce_bank<T, NS>(s_par, a.params + P.par_off, rec, nb, s_st, lane, acci);
              } else {
                fce_bank<T, NS>(s_par, a.params + P.par_off, rec, nb, s_st, lane, acc);
              }
              op += nb - 1;
              continue;
            }
          }
          if constexpr (FAM >= 4) {
            if (fl < 0) {                   // WFK_FCE_OWNMUL: a carrier term under an envelope of its own
              if (fl & 8) {
                if constexpr (CPLX) fce_ownmul<T, NS>(s_par, rec, a, x, st, fl, acci);
              } else {
                fce_ownmul<T, NS>(s_par, rec, a, x, st, fl, acc);
              }
              continue;
            }
          }
          if constexpr (FAM >= 3) {
            if ((fl & 0x32) == 0x32) {      // env == 3, deg >= 2: stateless closing multiplier
              if (fl & 1) fce_mollmul<T, NS, CPLX>(rec, x, acc, acci);
              else fce_tabmul<T, NS, CPLX>(rec, a, x, acc, acci);
              continue;
            }
          }
          double* const stg = st + ((fl & (WFK_FCE_HAS_CS | WFK_FCE_HAS_GR)) == (WFK_FCE_HAS_CS | WFK_FCE_HAS_GR) ? 128 : 0);
          // (all four loads issue at once whatever the op has: a pair it lacks reads the pair it has,
          //  and is never used or written back)
          sd.c = st[0];
          sd.s = st[64];
          sd.g = stg[0];
          sd.r = stg[64];
          if (FAM >= 1 && ((fl >> 4) & 3) == 3) {
            if constexpr (FAM >= 1) {
              if (fl & 3) fce_erfmul<T, NS, CPLX>(rec, sd, x, acc, acci);
              else fce_envmul<T, NS, CPLX>(rec, sd, acc, acci);
            }
          } else if (fl & 8) {
            if constexpr (CPLX) fce_eval<T, NS, CORR>(s_par, rec, sd, x, true, acci, fl, &cc);
          } else {
            fce_eval<T, NS, CORR>(s_par, rec, sd, x, true, acc, fl, &cc);
          }
          if (fl & WFK_FCE_HAS_CS) {
            st[0] = sd.c;
            st[64] = sd.s;
          }
          if (fl & WFK_FCE_HAS_GR) {
            stg[0] = sd.g;
            stg[64] = sd.r;
          }
        }
        state_piece = q;
        state_w0 = w0 + WT;
        ++since_seed;
        ++cs_age;
      }
      // tile base pinned to SGPRs: the stores become `global_store v_lane_off, data, s[base]
      // offset:k*512` (one address VGPR instead of a hoisted 64-bit pointer pair per store)
      if (mine) store_tile<T, CPLX, NS, CORR>(a, C, P, uniptr(outr + w0), uniptr(outc + w0), w0, lane, acc, acci);
      if (P.stop >= w1 || q + 1 >= C.piece_end) break;   // the tile ends inside this piece
      q = uni(q + 1);
      cur = load_piece(a.pieces + q);
      p = q;
    }
  }
}

#ifndef WFK_TL_WGS
#define WFK_TL_WGS 3
#endif
template <typename T, bool CPLX, bool TLIST, bool GENERIC, bool DIRECT, int NS>
// (the build with generic terms but no direct tier is capped at 256 VGPRs as well: left alone it took 261 + 5 AGPRs = one
//  wave per SIMD; same box, sinc / INTERP / mollifier pulses 4.79 / 3.04 / 2.18 ms uncapped, 3.25 / 2.34 / 1.37 capped)
// (the build with direct primitives inlines all of device libm's shapes: left alone it takes 280 VGPRs = ONE
// workgroup per CU, one wave per SIMD walking serial libm chains; capped at 256 it runs two: direct tier 1.8x.
// The tlist builds (no fused code, 8 samples per lane) fit three at 168: 4.80 -> 3.45 ms on 64 x 2e6 times.)
__global__ void __launch_bounds__(WFK_WG, TLIST ? WFK_TL_WGS : ((DIRECT || GENERIC) ? 2 : 1)) wfk_sample(const KArgs a);
// the body, shared with wfk_sample_wide: T arithmetic / accumulators, E elements of the output
template <typename T, typename E, bool CPLX, bool TLIST, bool GENERIC, bool DIRECT, int NS, bool SLICE>
__device__ __forceinline__ void wfk_sample_body(const KArgs& a) {
  __shared__ __attribute__((aligned(16))) double s_par_all[WFK_LDS_DOUBLES];
  __shared__ double s_val[DIRECT ? NS * WFK_WG : 1];   // direct-factor values (apply_factor)
  constexpr int WT = 64 * NS;
  constexpr int TILE = WFK_WG * NS;
  using OutR = typename OutOps<E>::Real;
  using OutC = typename OutOps<E>::Cplx;

  const int lane = threadIdx.x & 63;
  const int wave = uni(threadIdx.x >> 6);
  // (time-list builds with one sample per lane: wave-private staging of small blocks, see the block loop)
  constexpr bool WP = TLIST && NS == 1;
  const bool wp = WP && a.wavepriv != 0;
  double* const s_par = (WP && wp) ? s_par_all + wave * (WFK_LDS_DOUBLES / 4) : s_par_all;
  const int64_t chunk = xcd_chunk(a);
  if (chunk < 0) return;
  // indices and bases pinned to SGPRs, plan tables read with s_load: see uni64()/cload()
  const int ch = uni((int)(chunk / a.chunks_per_ch));
  const int64_t cc = uni64(chunk - (int64_t)ch * a.chunks_per_ch);
  const DevChannel C = a.channels[ch];
  int p = uni(a.chunk_first[chunk]);
  int64_t staged = -1;

  OutR* const outr = uniptr(reinterpret_cast<OutR*>(a.out) + (int64_t)ch * a.ch_stride);
  OutC* const outc = uniptr(reinterpret_cast<OutC*>(a.out) + (int64_t)ch * a.ch_stride);

  for (int tt = 0; tt < a.tiles_per_chunk; ++tt) {
    const int64_t g0 = uni64((cc * a.tiles_per_chunk + tt) * TILE);
    if (g0 >= a.n) break;
    const int64_t g1 = g0 + TILE < a.n ? g0 + TILE : a.n;
    const int64_t w0 = uni64(g0 + (int64_t)wave * WT);
    const int64_t j0 = w0 + lane;
    while (p < C.piece_end - 1 && cload<int64_t>(a.pieces + p, offsetof(DevPiece, stop)) <= g0)
      p = uni(p + 1);

    for (int q = p; q < C.piece_end; q = uni(q + 1)) {
      const DevPiece P = load_piece(a.pieces + q);
      if (P.start >= g1) break;
      // (mixed plans: the lean / short and the zero pieces were written by the other kernel's launch)
      if constexpr (TLIST && !GENERIC && !DIRECT) {
        // (mixed time-list plans: this build takes the fully fused and the zero pieces, the other build the rest)
        if (a.mixed && P.n_blk != 0 && !(P.flags & WFK_PF_LEAN)) continue;
      } else {
        if (a.mixed && (P.n_blk == 0 || (P.flags & (WFK_PF_LEAN | WFK_PF_SHORT)))) continue;
      }
      const bool active = w0 < a.n && P.start < w0 + WT && P.stop > w0;  // wave-uniform

      T acc[NS], acci[CPLX ? NS : 1];
#pragma unroll
      for (int k = 0; k < NS; ++k) acc[k] = (T)0;
      if (CPLX) {
#pragma unroll
        for (int k = 0; k < NS; ++k) acci[k] = (T)0;
      }

      // fused-only tlist plans: the lane's sample times on the channel's own axis, once per piece and tile
      // (the pointwise fused ops all read them; the build with the direct tier reloads them per op instead)
      double tx[(TLIST && !GENERIC && !DIRECT) ? NS : 1];
      if constexpr (TLIST && !GENERIC && !DIRECT) {
        if (active) {
#pragma unroll
          for (int i = 0; i < NS; ++i) {
            tx[i] = time_at<true>(a, j0 + 64 * (int64_t)i);
            if (C.tshift != 0.0) tx[i] = tx[i] - C.tshift;
          }
        }
      }

      int64_t off = P.par_off;
      int len = P.first_len;
      for (int b = 0; b < P.n_blk; ++b) {
        if (WP && wp) {
          // one-sample-per-lane builds on plans of small blocks: every wave stages for itself, and only the blocks of
          // pieces that reach into its own 64 samples -- no workgroup barrier (with short pieces the four waves of a
          // workgroup otherwise take turns: each piece is evaluated by one of them while three wait at its barriers)
          if (active && off != staged) {
            for (int i = lane; i < len; i += 64) s_par[i] = a.params[off + i];
            staged = off;
          }
        } else if (off != staged) {
          __syncthreads();  // every wave is done with the previous block
          for (int i = threadIdx.x; i < len; i += WFK_WG) s_par[i] = a.params[off + i];
          __syncthreads();
          staged = off;
        }
        if (active) {
          const int nops = uni((int)s_par[1]);
          int pos = WFK_BLK_HDR;
          for (int k = 0; k < nops; ++k) {
            const int kind = uni((int)s_par[pos]);
            if constexpr (TLIST && !GENERIC && !DIRECT) {
              if (kind == WFK_OP_FCE) {      // fused group, evaluated from each sample's own time
                fce_point<T, NS, CPLX>(s_par + pos, tx, acc, acci);
                pos += WFK_FCE_REC;
                continue;
              }
            }
            if (!TLIST && kind == WFK_OP_FCE) {
              const int fl = uni(WFK_FCE_WORD(s_par + pos));
              if (((fl >> 4) & 3) == 3) {
                double x = grid_time<SLICE>(a, j0);
                if (C.tshift != 0.0) x = x - C.tshift;
                FceSeeds sd = fce_make_seeds(s_par + pos, x, fl);
                if (fl & 3) fce_erfmul<T, NS, CPLX>(s_par + pos, sd, x, acc, acci);
                else fce_envmul<T, NS, CPLX>(s_par + pos, sd, acc, acci);
              } else if (fl & 8) {
                if constexpr (CPLX) apply_fce<T, NS, SLICE>(s_par, s_par + pos, a, C.tshift, j0, acci);
              } else {
                apply_fce<T, NS, SLICE>(s_par, s_par + pos, a, C.tshift, j0, acc);
              }
              pos += WFK_FCE_REC;
              continue;
            }
            if (!GENERIC) continue;  // fused-only build: the plan holds no generic term
            const T ar = (T)s_par[pos + 1], ai = (T)s_par[pos + 2];
            const int nf = uni((int)s_par[pos + 3]);
            pos += WFK_TERM_HDR;
            T prod[NS];
#pragma unroll
            for (int i = 0; i < NS; ++i) prod[i] = (T)1;
            for (int f = 0; f < nf; ++f) {
              apply_factor<T, TLIST, DIRECT, NS, SLICE>(s_par, s_par + pos, a, C.tshift, j0, prod, s_val);
              pos += WFK_FREC;
            }
#pragma unroll
            for (int i = 0; i < NS; ++i) acc[i] += ar * prod[i];
            if (CPLX) {
#pragma unroll
              for (int i = 0; i < NS; ++i) acci[i] += ai * prod[i];
            }
          }
        }
        off += len;
        if (b + 1 < P.n_blk) len = (int)a.params[off];
      }

      if (active)
        store_tile<T, CPLX, NS, false, E>(a, C, P, uniptr(outr + w0), uniptr(outc + w0), w0, lane, acc, acci);
    }
  }
}

template <typename T, bool CPLX, bool TLIST, bool GENERIC, bool DIRECT, int NS>
__global__ void __launch_bounds__(WFK_WG, TLIST ? WFK_TL_WGS : ((DIRECT || GENERIC) ? 2 : 1)) wfk_sample(const KArgs a) {
  wfk_sample_body<T, T, CPLX, TLIST, GENERIC, DIRECT, NS, false>(a);
}
// the same for a plan that is a time slice of a longer grid (wfk_grid.i0 != 0; grid plans only: a time list carries its times)
// (the complex builds with the direct tier sit on the 256-register edge and the slice offset pushed them over it -- 848 / 734
//  VGPRs in scratch: they run ONE workgroup per CU instead, i.e. one wave per SIMD with 512 registers per lane, and the
//  allocator's overflow lands in AGPRs, not in memory)
#ifndef WFK_SLICE_CPLX_WGS
#define WFK_SLICE_CPLX_WGS 1
#endif
#define WFK_SLICE_WGS(CPLX, GENERIC, DIRECT) ((CPLX) && (DIRECT) ? WFK_SLICE_CPLX_WGS : ((DIRECT) || (GENERIC)) ? 2 : 1)
template <typename T, bool CPLX, bool GENERIC, bool DIRECT, int NS>
__global__ void __launch_bounds__(WFK_WG, WFK_SLICE_WGS(CPLX, GENERIC, DIRECT)) wfk_sample_slice(const KArgs a) {
  wfk_sample_body<T, T, CPLX, false, GENERIC, DIRECT, NS, true>(a);
}

// Float (complex64) output under DOUBLE arithmetic: what a float launch of a plan with generic terms, or of a time list,
// runs.  These tiers are bound by instruction issue, where unpacked fp32 costs what fp64 costs, so float accumulators
// bought nothing -- and they lose the cancellation factor x 6e-8 where a piece's terms cancel (tools/fuzz_soak.py awg,
// seed 306405: terms 1e3 x the result, 9.2e-5 of the peak in float; 6e-8 here).  The fused tiers keep their float forms.
template <bool CPLX, bool TLIST, bool GENERIC, bool DIRECT, int NS>
__global__ void __launch_bounds__(WFK_WG, TLIST ? WFK_TL_WGS : ((DIRECT || GENERIC) ? 2 : 1)) wfk_sample_wide(const KArgs a) {
  wfk_sample_body<double, float, CPLX, TLIST, GENERIC, DIRECT, NS, false>(a);
}
template <bool CPLX, bool GENERIC, bool DIRECT, int NS>
__global__ void __launch_bounds__(WFK_WG, WFK_SLICE_WGS(CPLX, GENERIC, DIRECT)) wfk_sample_wide_slice(const KArgs a) {
  wfk_sample_body<double, float, CPLX, false, GENERIC, DIRECT, NS, true>(a);
}

template <typename T, bool CPLX, bool TLIST, int NS>
int launch(const KArgs& a, int64_t blocks, hipStream_t s, bool lean, bool generic, bool direct) {
  const dim3 g((unsigned)blocks), b(WFK_WG);
  if constexpr (!TLIST) {   // (the lean kernel exists for grid plans only)
    if (lean) {
      const size_t lds = (size_t)(a.lean_par + 128 * a.lean_ops) * sizeof(double);
      if constexpr (sizeof(T) == 8) {
        if (a.corr) {      // (corrected carriers and chirps never share a plan's lean pieces: family <= 1 here)
          if (a.lean_fam >= 1) hipLaunchKernelGGL((wfk_sample_lean<T, CPLX, NS, true, 1>), g, dim3(64), lds, s, a);
          else hipLaunchKernelGGL((wfk_sample_lean<T, CPLX, NS, true, 0>), g, dim3(64), lds, s, a);
          return hipGetLastError() == hipSuccess ? 0 : -1;
        }
      }
      if (a.lean_fam >= 4) hipLaunchKernelGGL((wfk_sample_lean<T, CPLX, NS, false, 4>), g, dim3(64), lds, s, a);
      else if (a.lean_fam == 3) hipLaunchKernelGGL((wfk_sample_lean<T, CPLX, NS, false, 3>), g, dim3(64), lds, s, a);
      else if (a.lean_fam == 2) hipLaunchKernelGGL((wfk_sample_lean<T, CPLX, NS, false, 2>), g, dim3(64), lds, s, a);
      else if (a.lean_fam == 1) hipLaunchKernelGGL((wfk_sample_lean<T, CPLX, NS, false, 1>), g, dim3(64), lds, s, a);
      else hipLaunchKernelGGL((wfk_sample_lean<T, CPLX, NS, false, 0>), g, dim3(64), lds, s, a);
      return hipGetLastError() == hipSuccess ? 0 : -1;
    }
  }
  if constexpr (!TLIST) {
    if (a.i0 != 0) {                     // a time slice of a longer grid: the builds that offset the sample index
      if constexpr (sizeof(T) == 4) {
        if (direct) { hipLaunchKernelGGL((wfk_sample_wide_slice<CPLX, true, true, NS>), g, b, 0, s, a); return hipGetLastError() == hipSuccess ? 0 : -1; }
        if (generic) { hipLaunchKernelGGL((wfk_sample_wide_slice<CPLX, true, false, NS>), g, b, 0, s, a); return hipGetLastError() == hipSuccess ? 0 : -1; }
      }
      if (direct) hipLaunchKernelGGL((wfk_sample_slice<T, CPLX, true, true, NS>), g, b, 0, s, a);
      else if (generic) hipLaunchKernelGGL((wfk_sample_slice<T, CPLX, true, false, NS>), g, b, 0, s, a);
      else hipLaunchKernelGGL((wfk_sample_slice<T, CPLX, false, false, NS>), g, b, 0, s, a);
      return hipGetLastError() == hipSuccess ? 0 : -1;
    }
  }
  if constexpr (sizeof(T) == 4) {        // float / complex64 outputs of these tiers: double arithmetic (wfk_sample_wide)
    if (TLIST && !generic && !direct) { hipLaunchKernelGGL((wfk_sample_wide<CPLX, TLIST, false, false, NS>), g, b, 0, s, a); return hipGetLastError() == hipSuccess ? 0 : -1; }
    if (TLIST || direct) { hipLaunchKernelGGL((wfk_sample_wide<CPLX, TLIST, true, true, NS>), g, b, 0, s, a); return hipGetLastError() == hipSuccess ? 0 : -1; }
    if (generic) { hipLaunchKernelGGL((wfk_sample_wide<CPLX, false, true, false, NS>), g, b, 0, s, a); return hipGetLastError() == hipSuccess ? 0 : -1; }
  }
  if (TLIST && !generic && !direct)      // every term of the plan fused: pointwise ops only, no libm shapes of the direct tier
    hipLaunchKernelGGL((wfk_sample<T, CPLX, TLIST, false, false, NS>), g, b, 0, s, a);
  else if (TLIST || direct)
    hipLaunchKernelGGL((wfk_sample<T, CPLX, TLIST, true, true, NS>), g, b, 0, s, a);
  else if (generic)
    hipLaunchKernelGGL((wfk_sample<T, CPLX, false, true, false, NS>), g, b, 0, s, a);
  else
    hipLaunchKernelGGL((wfk_sample<T, CPLX, false, false, false, NS>), g, b, 0, s, a);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace

int wfk_launch_sampler(const KArgs& a, int32_t n_channels, int out_kind, bool tlist, int ns, bool lean,
                       bool generic, bool direct, void* stream, std::string& err) {
  if (a.chunk_base < 0 || a.chunk_base + a.n_chunks > (int64_t)n_channels * a.chunks_per_ch) { err = "chunk range outside the plan"; return WFK_EINVAL; }
  const int64_t blocks = ((a.n_chunks + 7) >> 3) << 3;   // see xcd_chunk()
  if (blocks == 0) return WFK_OK;
  if (blocks > 0x7fffffffLL) { err = "grid too large"; return WFK_EINVAL; }
  hipStream_t s = (hipStream_t)stream;
  int rc;
  if (!tlist) {
    switch (out_kind) {
      case WFK_OUT_F64: rc = launch<double, false, false, WFK_NS_GRID>(a, blocks, s, lean, generic, direct); break;
      case WFK_OUT_F32: rc = launch<float, false, false, WFK_NS_GRID>(a, blocks, s, lean, generic, direct); break;
      case WFK_OUT_C128: rc = launch<double, true, false, WFK_NS_GRID>(a, blocks, s, lean, generic, direct); break;
      case WFK_OUT_C64: rc = launch<float, true, false, WFK_NS_GRID>(a, blocks, s, lean, generic, direct); break;
      default: err = "bad out_kind"; return WFK_EINVAL;
    }
  } else if (ns == WFK_NS_TLIST_SMALL) {
    switch (out_kind) {
      case WFK_OUT_F64: rc = launch<double, false, true, WFK_NS_TLIST_SMALL>(a, blocks, s, false, generic, direct); break;
      case WFK_OUT_F32: rc = launch<float, false, true, WFK_NS_TLIST_SMALL>(a, blocks, s, false, generic, direct); break;
      case WFK_OUT_C128: rc = launch<double, true, true, WFK_NS_TLIST_SMALL>(a, blocks, s, false, generic, direct); break;
      case WFK_OUT_C64: rc = launch<float, true, true, WFK_NS_TLIST_SMALL>(a, blocks, s, false, generic, direct); break;
      default: err = "bad out_kind"; return WFK_EINVAL;
    }
  } else {
    switch (out_kind) {
      case WFK_OUT_F64: rc = launch<double, false, true, WFK_NS_TLIST>(a, blocks, s, false, generic, direct); break;
      case WFK_OUT_F32: rc = launch<float, false, true, WFK_NS_TLIST>(a, blocks, s, false, generic, direct); break;
      case WFK_OUT_C128: rc = launch<double, true, true, WFK_NS_TLIST>(a, blocks, s, false, generic, direct); break;
      case WFK_OUT_C64: rc = launch<float, true, true, WFK_NS_TLIST>(a, blocks, s, false, generic, direct); break;
      default: err = "bad out_kind"; return WFK_EINVAL;
    }
  }
  if (rc) { err = std::string("kernel launch failed: ") + hipGetErrorString(hipGetLastError()); return WFK_EHIP; }
  return WFK_OK;
}
