// wfk_chain_dev.h -- the lean sampler's arithmetic for kernels that EVALUATE their input instead of loading it
// (fir_sampled: the FIR transform's windows, wfk_fir_sampled.hip; iir_sampled: the IIR scan's chunks, wfk_iir.hip).
// A thread owns one CHAIN of CL samples a fixed lane stride apart: one exact seed (sincospi + two exp) per fused
// carrier-envelope op and thread, then the phasor-table / Gaussian-recurrence arithmetic of the lean sampler kernel
// (wfk_kernels.hip) at that stride -- the plan is compiled for the geometry by wfk_compile_geom (wfk_compile.cpp).
// Reference semantics: Waveform.__call__ -> calc_parts / _calc (waveforms/_waveform.pyx:134-169).
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>
#include <utility>

#include "wfk_internal.h"

namespace {

template <int... K, typename F>
__device__ __forceinline__ void sfor_impl(std::integer_sequence<int, K...>, F&& f) {
  (f(std::integral_constant<int, K>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void sfor(F&& f) {
  sfor_impl(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f));
}
#define CH_EACH(N, k) sfor<N>([&](auto k##_) __attribute__((always_inline)) { constexpr int k = decltype(k##_)::value;
#define CH_END });

__device__ __forceinline__ int cuni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ int64_t cuni64(int64_t v) {     // a block-uniform 64-bit value, pinned to an SGPR pair
  return (int64_t)(((uint64_t)(uint32_t)cuni((int)(v >> 32)) << 32) | (uint32_t)cuni((int)v));
}

struct ChSeeds { double c, s, g, r; };

// ---- seeds: sin/cos(pi r) on |r| <= 1/2 and exp(x), straight-line --------------------------
// The libm routines cost ~300 VALU instructions per op and thread here (general argument
// reduction, special cases), a quarter of the kernel's time.  The arguments of the seeds are
// already reduced / bounded by the host's range checks, so two short polynomial kernels do:
//   sin(pi r), cos(pi r): fold to z in [0, 1/4], Taylor in t = pi z up to t^17 / t^18
//   exp(x), |x| < 700:    n = rint(x log2 e), Cody-Waite r = x - n ln2 (two words), Taylor to
//                         r^14 on |r| <= ln2/2, ldexp
// Truncation < 5e-18 relative for all three; rounding of the Horner chains ~1e-16.
// A polynomial coefficient pinned to an SGPR pair where it is used.  Left to itself the compiler
// hoists all ~35 coefficients of the seed kernels out of the piece / op loops into VGPRs and then
// SPILLS them there (the accumulators need the registers): 150 B of scratch per thread, 16 GB of
// extra HBM writes per launch of the C4 chain (rocprofv3 WRITE_SIZE 35 GB for 20 GB of output).
__device__ __forceinline__ double kc(double v) {
  asm volatile("" : "+s"(v));
  return v;
}

__device__ __forceinline__ void sincospi_small(double r, double* sn, double* cs) {
  const double a = fabs(r);
  const bool swap = a > 0.25;
  const double z = swap ? 0.5 - a : a;                    // exact
  const double t = z * 3.141592653589793116 + z * 1.2246467991473532e-16;
  const double t2 = t * t;
  double ps = kc(-2.8114572543455206e-15);                    // -1/17!
  ps = fma(ps, t2, kc(7.6471637318198164e-13));               //  1/15!
  ps = fma(ps, t2, kc(-1.6059043836821613e-10));              // -1/13!
  ps = fma(ps, t2, kc(2.5052108385441720e-08));               //  1/11!
  ps = fma(ps, t2, kc(-2.7557319223985893e-06));              // -1/9!
  ps = fma(ps, t2, kc(1.9841269841269841e-04));               //  1/7!
  ps = fma(ps, t2, kc(-8.3333333333333332e-03));              // -1/5!
  ps = fma(ps, t2, kc(1.6666666666666666e-01));               //  1/3!  (sign folded below)
  const double s = fma(-t * t2, ps, t);                   // t - t^3/6 + ...   (ps holds +1/6 - t^2/120 ...)
  double pc = kc(1.5619206968586226e-16);                     //  1/18!
  pc = fma(pc, t2, kc(-4.7794773323873853e-14));              // -1/16!
  pc = fma(pc, t2, kc(1.1470745597729725e-11));               //  1/14!
  pc = fma(pc, t2, kc(-2.0876756987868099e-09));              // -1/12!
  pc = fma(pc, t2, kc(2.7557319223985888e-07));               //  1/10!
  pc = fma(pc, t2, kc(-2.4801587301587302e-05));              // -1/8!
  pc = fma(pc, t2, kc(1.3888888888888889e-03));               //  1/6!
  pc = fma(pc, t2, kc(-4.1666666666666664e-02));              // -1/4!
  pc = fma(pc, t2, kc(0.5));                                  //  1/2! (sign folded below)
  const double c = fma(-t2, pc, 1.0);                     // 1 - t^2/2 + t^4/24 ...
  const double ss = swap ? c : s, cc = swap ? s : c;
  *sn = r < 0.0 ? -ss : ss;
  *cs = cc;
}

__device__ __forceinline__ double exp_small(double x) {
#pragma clang fp contract(off)   // the reduction's fma()s are explicit
  const double n = rint(x * 1.4426950408889634);
  double r = fma(-n, 6.93147180369123816490e-01, x);      // ln2 high word (low 11 bits zero: n * hi exact)
  r = fma(-n, 1.90821492927058770002e-10, r);             // ln2 low word
  double p = kc(1.1470745597729725e-11);                      // 1/14!
  p = fma(p, r, kc(1.6059043836821613e-10));                  // 1/13!
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

// exact per-thread seeds of one fused op (the arithmetic of fce_seeds in wfk_kernels.hip)
__device__ __forceinline__ ChSeeds chain_seeds(double theta, double ea, double eb, int carrier, int env) {
#pragma clang fp contract(off)   // the explicit fma()s must stay the only ones (see sincos_phase)
  ChSeeds o;
  o.c = 1.0; o.s = 0.0; o.g = 1.0; o.r = 1.0;
  if (carrier) {
    const double IPI_HI = 0.31830988618379069, IPI_LO = -1.9678676675182486e-17;
    const double xh = theta * IPI_HI;
    const double xl = fma(theta, IPI_HI, -xh) + theta * IPI_LO;
    const double n = rint(xh);
    double ss, cc;
    sincospi_small((xh - n) + xl, &ss, &cc);
    const bool odd = ((long long)n) & 1;
    o.s = odd ? -ss : ss;
    o.c = odd ? -cc : cc;
  }
  if (env) {
    o.g = exp_small(ea);
    o.r = exp_small(eb);
  }
  return o;
}

// one fused carrier-envelope op over the thread's chain:
//   acc[k] += E_k * (A(u_k) cos th_k + B(u_k) sin th_k),  k < CL, samples 256 apart
// ONE loop for every op shape (degree <= 3 Horner with zero high coefficients, a unit phasor
// table for ops without carrier, q = 1 without envelope): with 2 * CL accumulator registers the
// per-shape specialisations of the lean sampler kernel cost more in copies and spills at their
// control-flow merges than the few multiplies they save.  MASK: only the samples
// klo <= k < khi belong to the piece.  The Gaussian state stays in double (also for float output).
// KC: the chain position that is the NEXT pair's first sample (2 * HOPB): the op state there
// (phasor turned by table entry KC, Gaussian recurrence after KC steps) is returned in `nx`, so a
// workgroup walking consecutive pairs needs exact seeds for its first pair only.  KC == CL: the state after the whole chain
// (iir_sampled: a lane's run continues in the next round).
// SEL (with MASK): the mask as a per-sample select on the value instead of a branch around the sample's arithmetic (which
// the compiler turns into 32 precomputed exec masks in SGPR pairs: iir_sampled has no SGPRs to spare).
// INIT: the op's values are STORED into acc (the first op of a round whose every lane runs it), not added.
template <typename T, int CL, int KC, bool MASK, bool DEG1, bool SEL = false, bool INIT = false>
__device__ __forceinline__ void chain_loop(const double2* tab, const double* r, const ChSeeds& sd, double u0,
                                           double q, T (&acc)[CL], int klo, int khi, ChSeeds& nx) {
  const T c0 = (T)sd.c, s0 = (T)sd.s;
  const T A0 = (T)r[WFK_FCE_A], A1 = (T)r[WFK_FCE_A + 1], A2 = (T)r[WFK_FCE_A + 2], A3 = (T)r[WFK_FCE_A + 3];
  const T B0 = (T)r[WFK_FCE_B], B1 = (T)r[WFK_FCE_B + 1], B2 = (T)r[WFK_FCE_B + 2], B3 = (T)r[WFK_FCE_B + 3];
  // A(u) ck + B(u) sk with ck = c0 C - s0 S, sk = s0 C + c0 S  ==  C P(u) + S Q(u), P_i = A_i c0 + B_i s0,
  // Q_i = B_i c0 - A_i s0: for the degree-1 variant (every op of the BASELINE chain) four products per op
  // and thread replace four per sample.  (The cubic variant keeps the explicit rotation: folded, its
  // eight coefficients pushed fir_sampled<double,12> to 20 spilled VGPRs, 10.70 -> 10.83 ms.)
  const T P0 = DEG1 ? A0 * c0 + B0 * s0 : (T)0, P1 = DEG1 ? A1 * c0 + B1 * s0 : (T)0;
  const T Q0 = DEG1 ? B0 * c0 - A0 * s0 : (T)0, Q1 = DEG1 ? B1 * c0 - A1 * s0 : (T)0;
  double g = sd.g, rr = sd.r;
  T u = (T)u0;
  const T Dt = (T)r[WFK_FCE_D];
#ifndef WFK_FIRS_SB
#define WFK_FIRS_SB 1      // table entries fetched per sub-batch.  Same box, C4: 4 -> 9.70 ms (8 VGPRs spilled: +13 % HBM writes), 2 -> 9.66, 1 -> 9.58 (none)
#endif
  constexpr int SB = CL % WFK_FIRS_SB == 0 ? WFK_FIRS_SB : 2;   // sub-batch: bounds the live table entries / temporaries
  static_assert(CL % SB == 0, "chain length must be even");
  CH_EACH(CL / SB, kb)
    double2 tb[SB];
    CH_EACH(SB, kk) tb[kk] = tab[kb * SB + kk]; CH_END   // wave-wide LDS broadcasts (fetching a batch ahead
                                                          // was measured slower: 13.0 vs 11.9 ms, spills)
    CH_EACH(SB, kk)
      constexpr int k = kb * SB + kk;
      T val;
      if constexpr (DEG1) {
        // degree <= 1: the seed phasor is folded into the polynomials (C P(u) + S Q(u), see chain_loop's head)
        const T pp = P1 * u + P0, qq = Q1 * u + Q0;
        u += Dt;
        val = (pp * (T)tb[kk].x + qq * (T)tb[kk].y) * (T)g;
      } else {
        const T pa = ((A3 * u + A2) * u + A1) * u + A0;
        const T pb = ((B3 * u + B2) * u + B1) * u + B0;
        u += Dt;
        const T ck = c0 * (T)tb[kk].x - s0 * (T)tb[kk].y;
        const T sk = s0 * (T)tb[kk].x + c0 * (T)tb[kk].y;
        val = (pa * ck + pb * sk) * (T)g;
      }
      if constexpr (k == KC) {
        nx.g = g;
        nx.r = rr;
        nx.c = sd.c * tb[kk].x - sd.s * tb[kk].y;
        nx.s = sd.s * tb[kk].x + sd.c * tb[kk].y;
      }
      g *= rr;
      rr *= q;
      if constexpr (INIT) acc[k] = val;
      else if constexpr (MASK && SEL) acc[k] += (k >= klo && k < khi) ? val : (T)0;
      else if (!MASK || (k >= klo && k < khi)) acc[k] += val;
    CH_END
    __builtin_amdgcn_sched_barrier(0);
  CH_END
  if constexpr (KC == CL) {      // the state one whole chain further on (table entry CL = NS: the compiler's "advance by one tile" entry)
    const double2 te = tab[CL];
    nx.g = g;
    nx.r = rr;
    nx.c = sd.c * te.x - sd.s * te.y;
    nx.s = sd.s * te.x + sd.c * te.y;
  }
}

__device__ __forceinline__ ChSeeds chain_make_seeds(const double* r, double x, int fl) {
  const double v = (x - r[WFK_FCE_SG]) / r[WFK_FCE_SIGMA], Hh = r[WFK_FCE_H];
  if (fl & WFK_FCE_EXPENV)   // exponential envelope: g = exp(alpha (x - ref)), constant ratio exp(alpha D) (q = 1)
    return chain_seeds(r[WFK_FCE_W] * (x - r[WFK_FCE_SREF]), r[WFK_FCE_SIGMA] * (x - r[WFK_FCE_SG]), Hh,
                       (fl >> 2) & 1, true);
  return chain_seeds(r[WFK_FCE_W] * (x - r[WFK_FCE_SREF]), -(v * v), -Hh * (2.0 * v + Hh), (fl >> 2) & 1,
                     ((fl >> 4) & 3) != 0);
}

// closing pseudo-op (envelope shared by all carriers of the piece): acc[k] *= g_k inside the piece
template <typename T, int CL, int KC, bool SEL = false>
__device__ __forceinline__ void chain_envmul(const double* r, const ChSeeds& sd, T (&acc)[CL], int klo, int khi,
                                             ChSeeds& nx) {
  double g = sd.g, rr = sd.r;
  const double q = r[WFK_FCE_Q];
  nx = sd;
  CH_EACH(CL, k)
    if constexpr (k == KC) {
      nx.g = g;
      nx.r = rr;
    }
    if constexpr (SEL) acc[k] *= (k >= klo && k < khi) ? (T)g : (T)1;
    else if (k >= klo && k < khi) acc[k] *= (T)g;
    g *= rr;
    rr *= q;
  CH_END
  if constexpr (KC == CL) {
    nx.g = g;
    nx.r = rr;
  }
}

}  // namespace
