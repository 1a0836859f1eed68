// wfk_fir_sampled.hip -- the sampler fused into the FIR transform (BASELINE configs[3]).
//
// Reference chain:  predistort(wav(t), ker=ker)  =  Waveform.__call__ (waveforms/waveform.py:529-563)
// followed by the FIR branch of predistort (waveforms/distortion.py:329-337).  Unfused, the
// samples make a round trip through HBM (8 B written by the sampler, 8 * L/M read back by the
// FIR).  Here the FIR workgroup EVALUATES its input windows instead of loading them: algorithmic
// traffic of the chain is the 8 B/sample of the filtered output only.
//
// Geometry.  fir_fused (wfk_fir_fused.hip) packs two overlap-save windows of 4096 samples as
// z = x1 + i x2 and thread `tid` holds v[n1] = (x1[256 n1 + tid], x2[256 n1 + tid]), n1 < 16.
// With the hop between windows a multiple of 256 (HOPB * 256; 3072 for K <= 1025) the second
// window's samples are the first window's continued: x2[256 n1 + tid] = x1[256 (n1 + HOPB) + tid].
// So every thread evaluates ONE chain of CL = 16 + HOPB samples 256 apart, starting at the first
// window's sample `tid`: one exact seed (sincospi + two exp) per fused op and thread, then the
// same phasor-table / Gaussian-recurrence arithmetic as the lean sampler kernel, at lane stride
// 256 instead of 64 (the plan is compiled a second time for that geometry, wfk_compile_geom).
// No libm inside the sample loops, no cross-lane traffic, and the parameter block is staged in
// the LDS array the transform uses afterwards.
//
// Only fully fused ("lean") real-valued plans without clip run here; wfk_chain_launch() falls
// back to sampler -> workspace -> FIR for everything else.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include "wfk.h"
#include "wfk_fft4096.h"
#include "wfk_internal.h"

extern "C" void wfk_internal_set_error(const char* msg);
extern "C" void wfk_internal_fir_tables(const wfk_fir_plan* p, const void** kspec, const void** tw,
                                        int* fused, int* nseg, int* K, int* lead);

namespace {

struct ChainArgs {
  const DevChannel* channels;
  const DevPiece* pieces;
  const double* params;
  const int32_t* pair_first;   // [n_channels * npairs] first piece overlapping each pair's chain
  void* out;
  int64_t out_stride, n, npairs;
  const void* hspec;
  const void* tw;
  double t0, step, last;
  int32_t has_last, hop, K, lead;
};

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

// t[j] = fl(fl(j*step) + t0) (NumPy's linspace / arange element formula), continued linearly for
// the zero-padded samples j < 0 and j >= n of the first and last windows
__device__ __forceinline__ double chain_time(const ChainArgs& a, int64_t j) {
#pragma clang fp contract(off)
  const double m = (double)j * a.step;
  double t = m + a.t0;
  if (a.has_last && j == a.n - 1) t = a.last;
  return t;
}

struct ChSeeds { double c, s, g, r; };

// exact per-thread seeds of one fused op (same arithmetic as fce_seeds in wfk_kernels.hip)
__device__ __attribute__((noinline)) ChSeeds chain_seeds(double theta, double ea, double eb, int carrier,
                                                         int env) {
#pragma clang fp contract(off)   // the explicit fma()s must stay the only ones (see sincos_phase)
  ChSeeds o;
  o.c = 1.0; o.s = 0.0; o.g = 1.0; o.r = 1.0;
  if (carrier) {
    const double IPI_HI = 0.31830988618379069, IPI_LO = -1.9678676675182486e-17;
    const double xh = theta * IPI_HI;
    const double xl = fma(theta, IPI_HI, -xh) + theta * IPI_LO;
    const double n = rint(xh);
    double ss, cc;
    sincospi((xh - n) + xl, &ss, &cc);
    const bool odd = ((long long)n) & 1;
    o.s = odd ? -ss : ss;
    o.c = odd ? -cc : cc;
  }
  if (env) {
    o.g = exp(ea);
    o.r = exp(eb);
  }
  return o;
}

// one fused carrier-envelope op over the thread's chain:
//   acc[k] += E_k * (A(u_k) cos th_k + B(u_k) sin th_k),  k < CL, samples 256 apart
// DEG 0 / 1 / 3 (= 2..3); MASK: only the samples klo <= k < khi belong to the piece.
// The Gaussian state stays in double (also for float output).
template <typename T, int CL, int DEG, bool CARRIER, bool ENV, bool MASK>
__device__ __forceinline__ void chain_loop(const double2* tab, const double* r, const ChSeeds& sd, double u0,
                                           T (&acc)[CL], int klo, int khi) {
  asm volatile("" : "+v"(u0) : : "memory");   // keeps the variants' shared sub-expressions from being hoisted
  const T c0 = (T)sd.c, s0 = (T)sd.s;
  const T A0 = (T)r[WFK_FCE_A], A1 = (T)r[WFK_FCE_A + 1], B0 = (T)r[WFK_FCE_B], B1 = (T)r[WFK_FCE_B + 1];
  const T A2 = DEG > 1 ? (T)r[WFK_FCE_A + 2] : (T)0, A3 = DEG > 1 ? (T)r[WFK_FCE_A + 3] : (T)0;
  const T B2 = DEG > 1 ? (T)r[WFK_FCE_B + 2] : (T)0, B3 = DEG > 1 ? (T)r[WFK_FCE_B + 3] : (T)0;
  const T ac = A0 * c0, as = A0 * s0;
  double g = sd.g, rr = sd.r;
  const double q = r[WFK_FCE_Q];
  T u = (T)u0;
  const T Dt = (T)r[WFK_FCE_D];
  constexpr int SB = CL % 4 == 0 ? 4 : 2;   // sub-batch: bounds the live table entries / temporaries
  static_assert(CL % SB == 0, "chain length must be even");
  CH_EACH(CL / SB, kb)
    double2 tb[SB];
    if (CARRIER) {
      CH_EACH(SB, kk) tb[kk] = tab[kb * SB + kk]; CH_END   // wave-wide LDS broadcasts
    }
    CH_EACH(SB, kk)
      constexpr int k = kb * SB + kk;
      T val;
      if (DEG == 0) {
        val = CARRIER ? ac * (T)tb[kk].x - as * (T)tb[kk].y : A0;
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
          const T ck = c0 * (T)tb[kk].x - s0 * (T)tb[kk].y;
          const T sk = s0 * (T)tb[kk].x + c0 * (T)tb[kk].y;
          val = pa * ck + pb * sk;
        } else {
          val = pa;
        }
      }
      if (ENV) {
        val *= (T)g;
        g *= rr;
        rr *= q;
      }
      if (!MASK || (k >= klo && k < khi)) acc[k] += val;
    CH_END
    __builtin_amdgcn_sched_barrier(0);
  CH_END
}

template <typename T, int CL, bool MASK>
__device__ __forceinline__ void chain_op(const double* blk, const double* r, double x, int fl, T (&acc)[CL],
                                         int klo, int khi) {
  const int deg = fl & 3, carrier = (fl >> 2) & 1, env = (fl >> 4) & 3;
  const double v = (x - r[WFK_FCE_SG]) / r[WFK_FCE_SIGMA], Hh = r[WFK_FCE_H];
  const ChSeeds sd = chain_seeds(r[WFK_FCE_W] * (x - r[WFK_FCE_SREF]), -(v * v), -Hh * (2.0 * v + Hh), carrier,
                                 env);
  const double2* tab = reinterpret_cast<const double2*>(blk + (fl >> 8));
  const double u0 = x - r[WFK_FCE_SLIN];
#define CHAIN_CALL(DG, CA, EN) chain_loop<T, CL, DG, CA, EN, MASK>(tab, r, sd, u0, acc, klo, khi)
  if (carrier) {
    if (env) {
      if (deg == 0) CHAIN_CALL(0, true, true); else if (deg == 1) CHAIN_CALL(1, true, true); else CHAIN_CALL(3, true, true);
    } else {
      if (deg == 0) CHAIN_CALL(0, true, false); else if (deg == 1) CHAIN_CALL(1, true, false); else CHAIN_CALL(3, true, false);
    }
  } else {
    if (env) {
      if (deg == 0) CHAIN_CALL(0, false, true); else CHAIN_CALL(3, false, true);
    } else {
      if (deg == 0) CHAIN_CALL(0, false, false); else CHAIN_CALL(3, false, false);
    }
  }
#undef CHAIN_CALL
}

// closing pseudo-op (envelope shared by all carriers of the piece): acc[k] *= g_k inside the piece
template <typename T, int CL>
__device__ __forceinline__ void chain_envmul(const double* r, double x, T (&acc)[CL], int klo, int khi) {
  const double v = (x - r[WFK_FCE_SG]) / r[WFK_FCE_SIGMA], Hh = r[WFK_FCE_H];
  const ChSeeds sd = chain_seeds(0.0, -(v * v), -Hh * (2.0 * v + Hh), 0, 1);
  double g = sd.g, rr = sd.r;
  const double q = r[WFK_FCE_Q];
  CH_EACH(CL, k)
    if (k >= klo && k < khi) acc[k] *= (T)g;
    g *= rr;
    rr *= q;
  CH_END
}

#ifndef WFK_FIRS_WAVES
#define WFK_FIRS_WAVES 3
#endif
template <typename T, int HOPB>
__global__ void __launch_bounds__(256, WFK_FIRS_WAVES) fir_sampled(const ChainArgs a) {
  constexpr int CL = 16 + HOPB;     // chain length: both windows of the pair
  __shared__ __attribute__((aligned(16))) T lds[LDS_ELEMS];
  double* const s_par = reinterpret_cast<double*>(lds);   // parameter block, before the transform needs the array
  const int tid = threadIdx.x;
  const int64_t pair = blockIdx.x;
  const int ch = blockIdx.y;
  const int64_t s1 = 2 * pair * (int64_t)a.hop - a.lead;   // first sample of the first window
  const int64_t j0 = s1 + tid;
  const int64_t range_end = s1 + 256 * (int64_t)CL;
  const DevChannel C = a.channels[ch];

  // ---- sampling phase: the chain of this thread ------------------------------------------
  T acc[CL];
  CH_EACH(CL, k) acc[k] = (T)0; CH_END
  double x = chain_time(a, j0);
  if (C.tshift != 0.0) x = x - C.tshift;
  int q = cuni(a.pair_first[(int64_t)ch * a.npairs + pair]);
  for (; q < C.piece_end; q = cuni(q + 1)) {
    const DevPiece P = a.pieces[q];
    if (P.start >= range_end) break;
    if (P.n_blk == 0 || P.stop <= s1) continue;             // zero piece / entirely before the chain
    __syncthreads();                                         // every wave is done with the previous block
    for (int i = tid; i < P.first_len; i += 256) s_par[i] = a.params[P.par_off + i];
    __syncthreads();
    const int nops = cuni((int)s_par[1]);
    const bool full = P.start <= s1 && P.stop >= range_end;  // wave-uniform: the whole chain range is inside
    // samples k of this thread inside the piece: P.start <= j0 + 256 k < P.stop
    int klo = 0, khi = CL;
    if (!full) {
      const int64_t lo = P.start - j0, hi = P.stop - j0;     // k >= lo/256 (ceil), k < hi/256 (ceil)
      klo = lo <= 0 ? 0 : (int)((lo + 255) >> 8);
      khi = hi <= 0 ? 0 : (int)((hi + 255) >> 8);
      klo = klo > CL ? CL : klo;
      khi = khi > CL ? CL : khi;
    }
    for (int op = 0; op < nops; ++op) {
      const double* rec = s_par + WFK_BLK_HDR + op * WFK_FCE_REC;
      const int fl = cuni((int)rec[WFK_FCE_DEG]);
      if (((fl >> 4) & 3) == 3) chain_envmul<T, CL>(rec, x, acc, klo, khi);
      else if (full) chain_op<T, CL, false>(s_par, rec, x, fl, acc, 0, CL);
      else chain_op<T, CL, true>(s_par, rec, x, fl, acc, klo, khi);
    }
  }
  __syncthreads();   // parameter block no longer needed: the array becomes the FFT exchange buffer

  // ---- zero padding, channel offset, packing of the two windows ---------------------------
  const T base = (T)C.offset;
  CH_EACH(CL, k)
    const int64_t j = j0 + 256 * k;
    acc[k] = (j >= 0 && j < a.n) ? acc[k] + base : (T)0;
  CH_END
  cx<T> v[16];
  CH_EACH(16, n1)
    v[n1].x = acc[n1];
    v[n1].y = acc[n1 + HOPB];
  CH_END

  // ---- transform, multiply by the kernel spectrum, inverse transform (as fir_fused) ---------
  const cx<T>* hspec = static_cast<const cx<T>*>(a.hspec);
  const cx<T>* tw = static_cast<const cx<T>*>(a.tw);
  const cx<T> wa = tw[tid], wb = tw[16 * (tid & 15)];
  fft4096<false>(v, lds, wa, wb, tid);
#pragma unroll
  for (int k3 = 0; k3 < 16; ++k3) v[k3] = cmul(v[k3], hspec[tid + 256 * k3]);
  __builtin_amdgcn_s_setprio(2);
  fft4096<true>(v, lds, wa, wb, tid);
  T* orow = static_cast<T*>(a.out) + (int64_t)ch * a.out_stride;
  const int M = a.hop;
  const int64_t b1 = 2 * pair, b2 = b1 + 1;
#pragma unroll
  for (int q3 = 0; q3 < 16; ++q3) {
    const int r = tid + 256 * q3 - (a.K - 1);
    if (r >= 0 && r < M) {
      const int64_t d1 = b1 * M + r, d2 = b2 * M + r;
      if (d1 < a.n) orow[d1] = v[q3].x;
      if (d2 < a.n) orow[d2] = v[q3].y;
    }
  }
}

int chain_fail(int code, const std::string& m) {
  wfk_internal_set_error(m.c_str());
  return code;
}

}  // namespace

struct wfk_chain_plan {
  wfk_plan* sampler = nullptr;    // the plain sampler plan (fallback path; also serves queries)
  wfk_fir_plan* fir = nullptr;
  bool fused = false;
  int32_t kind = 0, n_channels = 0, hopb = 0;
  int64_t n = 0, npairs = 0;
  std::string why;                // why the chain is not fused (diagnostics)
  // fused path: device tables of the plan compiled for the window geometry
  void* d_tables = nullptr;
  DevChannel* d_channels = nullptr;
  DevPiece* d_pieces = nullptr;
  double* d_params = nullptr;
  int32_t* d_pair_first = nullptr;
  double t0 = 0, step = 0, last = 0;
  int32_t has_last = 0;
  // unfused path: the sampler's output
  void* workspace = nullptr;
};

extern "C" {

int wfk_chain_plan_destroy(wfk_chain_plan* p) {
  if (!p) return WFK_OK;
  if (p->d_tables || p->workspace) (void)hipDeviceSynchronize();
  (void)hipFree(p->d_tables);
  (void)hipFree(p->workspace);
  wfk_plan_destroy(p->sampler);
  wfk_fir_plan_destroy(p->fir);
  delete p;
  return WFK_OK;
}

int wfk_chain_plan_create(const wfk_program* prog, const wfk_grid* grid, const double* ker_host, int32_t K,
                          int kind, wfk_chain_plan** out) {
  if (!out) return chain_fail(WFK_EINVAL, "null out");
  *out = nullptr;
  if (!prog || !grid || !ker_host) return chain_fail(WFK_EINVAL, "null argument");
  if (kind != WFK_OUT_F64 && kind != WFK_OUT_F32) return chain_fail(WFK_EINVAL, "chain kind must be F64 or F32");
  wfk_chain_plan* p = new wfk_chain_plan();
  p->kind = kind;
  int rc = wfk_plan_create_grid(prog, grid, &p->sampler);
  if (rc) { wfk_chain_plan_destroy(p); return rc; }
  p->n = grid->n;
  p->n_channels = prog->n_channels;
  rc = wfk_fir_plan_create(ker_host, K, grid->n, std::max(1, prog->n_channels), kind, &p->fir);
  if (rc) { wfk_chain_plan_destroy(p); return rc; }
  p->t0 = grid->t0; p->step = grid->step; p->last = grid->last; p->has_last = grid->has_last;
  if (p->n == 0 || p->n_channels == 0) { *out = p; return WFK_OK; }

  // ---- can the sampler run inside the transform? ---------------------------------------
  const void *kspec = nullptr, *tw = nullptr;
  int fir_fused = 0, nseg = 0, Kf = 0, lead = 0;
  wfk_internal_fir_tables(p->fir, &kspec, &tw, &fir_fused, &nseg, &Kf, &lead);
  const char* off = getenv("WFK_CHAIN_UNFUSED");
  HostPlan H;
  std::string err;
  // hop = HOPB * 256 <= 4096 - K + 1: 3072 for K <= 1025, 2560 up to the 1537 taps of one transform
  p->hopb = K <= 1025 ? 12 : 10;
  if (off && off[0] == '1') p->why = "disabled by WFK_CHAIN_UNFUSED";
  else if (!fir_fused || nseg != 1) p->why = "FIR kernel longer than one on-chip transform";
  else if (prog->n_channels > 65535) p->why = "more than 65535 channels";
  else if (wfk_compile_geom(prog, grid, 256, 16 + p->hopb, H, err) != WFK_OK) p->why = "geometry compile: " + err;
  else if (!H.lean) p->why = "plan is not fully fused (generic / direct terms, or more than 10 ops per piece)";
  else {
    for (const DevChannel& c : H.channels)
      if (c.do_clip) p->why = "clip (min/max) on a channel";
    for (uint8_t cx_ : H.channel_complex)
      if (cx_) p->why = "complex-valued channel";
  }
  if (p->why.empty()) {
    const int64_t hop = 256 * (int64_t)p->hopb;
    const int64_t nblk = (p->n + hop - 1) / hop;
    p->npairs = (nblk + 1) / 2;
    std::vector<int32_t> pair_first((size_t)p->npairs * p->n_channels);
    for (int32_t c = 0; c < p->n_channels; ++c) {
      int32_t q = H.channels[c].piece_begin;
      for (int64_t pr = 0; pr < p->npairs; ++pr) {
        const int64_t s1 = 2 * pr * hop - lead;
        while (q < H.channels[c].piece_end - 1 && H.pieces[q].stop <= s1) ++q;
        pair_first[(size_t)c * p->npairs + pr] = q;
      }
    }
    auto al = [](size_t x) { return (x + 255) & ~size_t(255); };
    const size_t b_ch = H.channels.size() * sizeof(DevChannel), b_pc = H.pieces.size() * sizeof(DevPiece),
                 b_pa = H.params.size() * sizeof(double), b_pf = pair_first.size() * sizeof(int32_t);
    const size_t o_pc = al(b_ch), o_pa = al(o_pc + b_pc), o_pf = al(o_pa + b_pa), total = al(o_pf + b_pf) + 256;
    std::vector<char> stage(total, 0);
    std::memcpy(stage.data(), H.channels.data(), b_ch);
    std::memcpy(stage.data() + o_pc, H.pieces.data(), b_pc);
    std::memcpy(stage.data() + o_pa, H.params.data(), b_pa);
    std::memcpy(stage.data() + o_pf, pair_first.data(), b_pf);
    if (hipMalloc(&p->d_tables, total) != hipSuccess ||
        hipMemcpy(p->d_tables, stage.data(), total, hipMemcpyHostToDevice) != hipSuccess) {
      wfk_chain_plan_destroy(p);
      return chain_fail(WFK_ENOMEM, "chain table allocation failed");
    }
    char* base = static_cast<char*>(p->d_tables);
    p->d_channels = reinterpret_cast<DevChannel*>(base);
    p->d_pieces = reinterpret_cast<DevPiece*>(base + o_pc);
    p->d_params = reinterpret_cast<double*>(base + o_pa);
    p->d_pair_first = reinterpret_cast<int32_t*>(base + o_pf);
    p->fused = true;
  } else {
    // unfused: the samples go through a workspace owned by the plan (no allocation at launch)
    const size_t es = kind == WFK_OUT_F32 ? 4 : 8;
    if (hipMalloc(&p->workspace, (size_t)p->n_channels * (size_t)p->n * es) != hipSuccess) {
      wfk_chain_plan_destroy(p);
      return chain_fail(WFK_ENOMEM, "chain workspace allocation failed");
    }
  }
  *out = p;
  return WFK_OK;
}

int wfk_chain_is_fused(const wfk_chain_plan* p) { return p && p->fused ? 1 : 0; }

const char* wfk_chain_unfused_reason(const wfk_chain_plan* p) { return p ? p->why.c_str() : ""; }

int wfk_chain_launch(wfk_chain_plan* p, void* out_dev, int64_t out_stride, void* hip_stream) {
  if (!p) return chain_fail(WFK_EINVAL, "null plan");
  if (p->n == 0 || p->n_channels == 0) return WFK_OK;
  if (!out_dev) return chain_fail(WFK_EINVAL, "null output");
  if (out_stride < p->n) return chain_fail(WFK_EINVAL, "out_stride smaller than n");
  hipStream_t s = (hipStream_t)hip_stream;
  if (!p->fused) {
    int rc = wfk_plan_launch(p->sampler, p->workspace, p->n, p->kind, 0, hip_stream);
    if (rc) return rc;
    return wfk_fir_apply(p->fir, p->workspace, p->n, out_dev, out_stride, hip_stream);
  }
  const void *kspec = nullptr, *tw = nullptr;
  int fir_fused = 0, nseg = 0, K = 0, lead = 0;
  wfk_internal_fir_tables(p->fir, &kspec, &tw, &fir_fused, &nseg, &K, &lead);
  ChainArgs a{};
  a.channels = p->d_channels; a.pieces = p->d_pieces; a.params = p->d_params; a.pair_first = p->d_pair_first;
  a.out = out_dev; a.out_stride = out_stride; a.n = p->n; a.npairs = p->npairs;
  a.hspec = kspec; a.tw = tw;
  a.t0 = p->t0; a.step = p->step; a.last = p->last; a.has_last = p->has_last;
  a.hop = 256 * p->hopb; a.K = K; a.lead = lead;
  const dim3 grid((unsigned)p->npairs, (unsigned)p->n_channels);
  if (p->kind == WFK_OUT_F32) {
    if (p->hopb == 12) hipLaunchKernelGGL((fir_sampled<float, 12>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((fir_sampled<float, 10>), grid, dim3(256), 0, s, a);
  } else {
    if (p->hopb == 12) hipLaunchKernelGGL((fir_sampled<double, 12>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((fir_sampled<double, 10>), grid, dim3(256), 0, s, a);
  }
  if (hipGetLastError() != hipSuccess) return chain_fail(WFK_EHIP, "fused sampler->FIR kernel launch failed");
  return WFK_OK;
}

}  // extern "C"
