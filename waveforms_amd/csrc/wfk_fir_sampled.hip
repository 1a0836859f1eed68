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
#include "wfk_short_dev.h"
#include "wfk_chain_dev.h"

extern "C" void wfk_internal_set_error(const char* msg);
extern "C" void wfk_internal_fir_tables(const wfk_fir_plan* p, const void** kspec, const void** tw,
                                        int* fused, int* nseg, int* K, int* lead);
extern "C" void wfk_internal_plan_tables(const wfk_plan* p, const HostPlan** h, const double** d_params);
extern "C" int64_t wfk_internal_fir_krow(const wfk_fir_plan* p);
extern "C" int wfk_internal_plan_launch_foreign(wfk_plan* p, void* out_dev, int64_t ch_stride, int out_kind, void* hip_stream);

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
  int64_t hrow;                // per-row kernels: complex elements between the spectra of consecutive rows (else 0)
  int64_t i0;                  // wfk_grid.i0: sample j is sample i0 + j of the caller's full grid
};

// t[j] = fl(fl(j*step) + t0) (NumPy's linspace / arange element formula), continued linearly for
// the zero-padded samples j < 0 and j >= n of the first and last windows
__device__ __forceinline__ double chain_time(const ChainArgs& a, int64_t j) {
#pragma clang fp contract(off)
  const double m = (double)(j + a.i0) * a.step;
  double t = m + a.t0;
  if (a.has_last && j == a.n - 1) t = a.last;
  return t;
}

#ifndef WFK_FIRS_WAVES
#define WFK_FIRS_WAVES 3
#endif
#ifndef WFK_FIRS_PPW
#define WFK_FIRS_PPW 1      // consecutive pairs per workgroup; > 1 carries the op state from pair to pair (exact
                            // seeds for the first pair only).  Measured at 4: 20.7 ms vs 12.5 (the loop-carried
                            // values push the allocator into 120 spills), so every workgroup takes ONE pair
#endif
#define WFK_FIRS_NCAR 2     // ops of a piece whose state is carried in registers from pair to pair

template <typename T, int HOPB>
__global__ void __launch_bounds__(256, WFK_FIRS_WAVES) fir_sampled(const ChainArgs a) {
  constexpr int CL = 16 + HOPB;     // chain length: both windows of the pair
  constexpr int PPW = WFK_FIRS_PPW, NCAR = WFK_FIRS_NCAR;
  constexpr bool CARRY = PPW > 1;
  constexpr int KC = CARRY ? 2 * HOPB : -1;   // chain position of the next pair's first sample
  static_assert(KC < CL, "the chains of consecutive pairs must overlap");
  __shared__ __attribute__((aligned(16))) T lds[LDS_ELEMS];
  double* const s_par = reinterpret_cast<double*>(lds);   // parameter block, before the transform needs the array
  constexpr int LDS_DOUBLES = (int)(sizeof(T) * LDS_ELEMS / sizeof(double));
  double2* const unit_tab = reinterpret_cast<double2*>(s_par + LDS_DOUBLES - 2 * CL);
  const int tid = threadIdx.x;
  const int ch = blockIdx.y;
  const DevChannel C = a.channels[ch];
  const cx<T>* hspec = static_cast<const cx<T>*>(a.hspec) + (int64_t)ch * a.hrow;
  const cx<T>* tw = static_cast<const cx<T>*>(a.tw);
  T* const orow = static_cast<T*>(a.out) + (int64_t)ch * a.out_stride;
  const int M = a.hop;

  int car_piece = -1;               // piece whose op state at this pair's first sample sits in car[] (wave-uniform)
  ChSeeds car[NCAR];
  for (int i = 0; i < NCAR; ++i) { car[i].c = 1.0; car[i].s = 0.0; car[i].g = 1.0; car[i].r = 1.0; }

  for (int pp = 0; pp < PPW; ++pp) {
    const int64_t pair = (int64_t)blockIdx.x * PPW + pp;
    if (pair >= a.npairs) break;
    const int64_t s1 = 2 * pair * (int64_t)a.hop - a.lead;   // first sample of the first window
    const int64_t range_end = s1 + 256 * (int64_t)CL;   // (per-thread sample index: s1 + tid + 256 k; kept as
                                                         //  the 32-bit `tid`, a 64-bit copy would be spilled)

    // ---- sampling phase: the chain of this thread ----------------------------------------
    T acc[CL];
    CH_EACH(CL, k) acc[k] = (T)0; CH_END
    double x = chain_time(a, s1 + tid);
    if (C.tshift != 0.0) x = x - C.tshift;
    __syncthreads();                                         // the previous pair's transform is done with the array
    if (tid < CL) unit_tab[tid] = make_double2(1.0, 0.0);    // phasor table of ops without a carrier
    int next_car = -1;
    int q = cuni(a.pair_first[(int64_t)ch * a.npairs + pair]);
    for (; q < C.piece_end; q = cuni(q + 1)) {
      const DevPiece P = a.pieces[q];
      if (P.start >= range_end) break;
      if (P.n_blk == 0 || P.stop <= s1) continue;           // zero piece / entirely before the chain
      __syncthreads();                                       // every wave is done with the previous block
      for (int i = tid; i < P.first_len; i += 256) s_par[i] = a.params[P.par_off + i];
      __syncthreads();
      const int nops = cuni((int)s_par[1]);
      const bool full = P.start <= s1 && P.stop >= range_end;   // wave-uniform: the whole chain is inside
      const bool carried = CARRY && full && car_piece == q;
      // samples k of this thread inside the piece: P.start <= j0 + 256 k < P.stop
      int klo = 0, khi = CL;
      if (!full) {
        // k >= (P.start - j0)/256 (ceil), k < (P.stop - j0)/256 (ceil), j0 = s1 + tid; the scalar parts
        // are clamped to what matters for k in [0, CL) so that the per-thread part is 32-bit
        const int64_t lim = 256 * (int64_t)(CL + 2);
        const int64_t los = P.start - s1, his = P.stop - s1;
        const int lo = (int)(los < -lim ? -lim : (los > lim ? lim : los)) - tid;
        const int hi = (int)(his < -lim ? -lim : (his > lim ? lim : his)) - tid;
        klo = lo <= 0 ? 0 : (lo + 255) >> 8;
        khi = hi <= 0 ? 0 : (hi + 255) >> 8;
        klo = klo > CL ? CL : klo;
        khi = khi > CL ? CL : khi;
      }
      for (int op = 0; op < nops; ++op) {
        const double* rec = s_par + WFK_BLK_HDR + op * WFK_FCE_REC;
        const int fl = cuni(WFK_FCE_WORD(rec));
        const int env = (fl >> 4) & 3, carrier = (fl >> 2) & 1;
        ChSeeds sd, nx;
        if (carried && op == 0) sd = car[0];
        else if (carried && NCAR > 1 && op == 1) sd = car[NCAR > 1 ? 1 : 0];
        else sd = chain_make_seeds(rec, x, fl);              // exact (libm)
        if (env == 3) {
          chain_envmul<T, CL, KC>(rec, sd, acc, klo, khi, nx);
        } else {
          const double2* tab = carrier ? reinterpret_cast<const double2*>(s_par + WFK_FCE_TABOFF(fl)) : unit_tab;
          const double qq = env ? rec[WFK_FCE_Q] : 1.0;
          const double u0 = x - rec[WFK_FCE_SLIN];
          // (the common shapes -- whole chain inside the piece, polynomials of degree <= 1: a
          //  DRAG-corrected pulse -- get the short loop; everything else the general one)
          if (full && (fl & 3) <= 1) chain_loop<T, CL, KC, false, true>(tab, rec, sd, u0, qq, acc, 0, CL, nx);
          else if (full) chain_loop<T, CL, KC, false, false>(tab, rec, sd, u0, qq, acc, 0, CL, nx);
          else chain_loop<T, CL, KC, true, false>(tab, rec, sd, u0, qq, acc, klo, khi, nx);
        }
        if (CARRY && op == 0) car[0] = nx;
        if (CARRY && NCAR > 1 && op == 1) car[NCAR > 1 ? 1 : 0] = nx;
      }
      // the next pair can start from car[] if it lies in this piece as a whole, too
      if (full) next_car = q;
    }
    car_piece = next_car;
    __syncthreads();   // parameter block no longer needed: the array becomes the FFT exchange buffer

    // ---- zero padding, channel offset, packing of the two windows -------------------------
    // (interior pairs -- all but the first and the last few of a row -- need no range checks at all:
    //  a wave-uniform branch keeps ~100 compares and selects out of their way)
    const T base = (T)C.offset;
    const int64_t b1 = 2 * pair, b2 = b1 + 1;
    const bool interior = s1 >= 0 && range_end <= a.n && (b2 + 1) * (int64_t)M <= a.n;
    if (interior) {
      if (base != (T)0) {
        CH_EACH(CL, k) acc[k] += base; CH_END
      }
    } else {
      CH_EACH(CL, k)
        const int64_t j = s1 + tid + 256 * k;
        acc[k] = (j >= 0 && j < a.n) ? acc[k] + base : (T)0;
      CH_END
    }
    cx<T> v[16];
    CH_EACH(16, n1)
      v[n1].x = acc[n1];
      v[n1].y = acc[n1 + HOPB];
    CH_END

    // ---- transform, multiply by the kernel spectrum, inverse transform (as fir_fused) -------
    // (nothing of the transform -- twiddles, exchange addresses -- may be hoisted above the sampling
    //  phase: it would be spilled there, and 150 B of scratch per thread are 16 GB per launch)
    int tf = tid;                      // the transform's view of the thread index, opaque: everything
    asm volatile("" : "+v"(tf));       // derived from it is computed HERE, after the sampling phase
    const cx<T> wa = tw[tf], wb = tw[16 * (tf & 15)];
    fft4096<false>(v, lds, wa, wb, tf);
#pragma unroll
    // (the 16 L2 loads of the kernel spectrum cost 3.6 %: 9.76 vs 10.12 ms with constants in their place)
    for (int k3 = 0; k3 < 16; ++k3) v[k3] = cmul(v[k3], hspec[tf + 256 * k3]);
    fft4096<true>(v, lds, wa, wb, tf);
    if (interior) {
      T* const o1 = orow + b1 * M + (tf - (a.K - 1));       // wave-uniform base + lane offset
      T* const o2 = o1 + M;
#pragma unroll
      for (int q3 = 0; q3 < 16; ++q3) {
        const int r = tf + 256 * q3 - (a.K - 1);
        if (r >= 0 && r < M) {
          o1[256 * q3] = v[q3].x;
          o2[256 * q3] = v[q3].y;
        }
      }
    } else {
#pragma unroll
      for (int q3 = 0; q3 < 16; ++q3) {
        const int r = tf + 256 * q3 - (a.K - 1);
        if (r >= 0 && r < M) {
          const int64_t d1 = b1 * M + r, d2 = b2 * M + r;
          if (d1 < a.n) orow[d1] = v[q3].x;
          if (d2 < a.n) orow[d2] = v[q3].y;
        }
      }
    }
  }
}

// ---- the same chain at AWG sample rates: fir_short ------------------------------------------------
// At 1-5 GS/s a pulse is 20-200 samples: a chain of 28 samples 256 apart crosses 28 different pieces, and
// the plan is in the short geometry anyway (wfk_short.hip).  Here the window is sampled the short tier's
// way: a THREAD owns a run of <= 16 CONTIGUOUS samples of one piece (one "entry" of the window), seeds
// its ops exactly once from the piece's compact record and steps the recurrences by dt
// (wfk_short_dev.h); the runs land in the LDS array, from where every thread picks up the
// stride-256 samples the transform wants from it.  The array holds 4624 elements and a pair of windows is
// 7168 samples, so the pair is sampled in two halves of HR = CL / 2 rows (3584 samples, ~240 entries of
// the 256 threads' one round for 60-sample pulses), padded by one element per 16 (a run of 16 starts 17
// elements after its neighbour's: conflict-free writes, and the strided reads skip one bank per 16
// lanes).  Zero padding and the channel offset are a prefill of the half before its entries are written.
// The host cuts the pieces of the sampler's own short plan into per-half entry lists
// (wfk_chain_plan_create); the op records are the plan's, read where they are.
// (ShortWin, WFK_CW_ENTRY: wfk_internal.h; the tables are built by wfk_chain_windows, wfk_compile.cpp)
struct ChainShortArgs {
  const DevChannel* channels;
  const ShortWin* wins;       // [n_channels * npairs * 2]
  const uint32_t* entries;    // drec (16 bit) | offset in the half (12 bit) | length - 1 (4 bit)
  const double* recs;         // the sampler plan's parameter table (op records)
  const void* ws;             // mixed plans: rows holding the general kernel's pieces (elements of T), else null
  int64_t ws_stride;
  void* out;
  int64_t out_stride, n, npairs;
  const void* hspec;
  const void* tw;
  double step;
  int32_t hop, K, lead;
  int64_t hrow;               // per-row kernels: complex elements between the spectra of consecutive rows (else 0)
};

// runs of pieces without a short form (mixed plans), copied from the workspace into the half's LDS image.
// Out of line: inlined, the rare path cost the common one 390 static / 10 % dynamic VALU instructions.
template <typename T>
__device__ __attribute__((noinline)) void chain_copy_runs(T* lds, const T* wrow, const uint32_t* ents, int ccnt,
                                                          bool sw, int tid) {
  for (int cb = 0; cb < ccnt; cb += 256) {
    const int idx = cb + tid;
    if (idx < ccnt) {
      const uint32_t word = ents[idx];
      const int len = (int)(word >> 28) + 1, o = (int)((word >> 16) & 0xfff);
      T* const b0 = lds + (sw ? o + (o >> 4) : o);
      const int t = sw ? 16 - (o & 15) : 99;
      for (int k = 0; k < len; ++k) *((k >= t ? b0 + 1 : b0) + k) = wrow[o + k];
    }
  }
}

#ifndef WFK_FSH_PARK
#define WFK_FSH_PARK 7     // same box, 2048 x 1e5 at 2 GS/s: 0 -> 1.017 ms (8 VGPRs in scratch), 5 -> 0.985 (2), 6 -> 0.978 (2), 7 -> 0.965 (none; 45 KB of LDS per workgroup, three still fit a CU)
#endif
template <typename T, int HOPB>
__global__ void __launch_bounds__(256, WFK_FIRS_WAVES) fir_short(const ChainShortArgs a) {
  constexpr int CL = 16 + HOPB, HR = CL / 2, HALF = 256 * HR, R = WFK_SH_R;
  static_assert(CL % 2 == 0 && HALF + HALF / 16 <= LDS_ELEMS, "a half window must fit the transform's array");
  static_assert(HALF <= 4096, "entry offsets are 12 bit");
  // The first half's HR samples per thread wait in registers while the second half is sampled -- on top of that
  // phase's own 16 accumulators and op record.  The double kernel had 4 of them in scratch (8 VGPRs spilled:
  // WRITE_SIZE 1.30x the output).  They are parked in the array instead: the second half's image ends at
  // HALF + HALF / 16, the PARK rows behind it belong to nobody until the transform (one more row of LDS).
  constexpr int PARK = sizeof(T) == 8 ? WFK_FSH_PARK : 0, PARK_BASE = HALF + HALF / 16;
  constexpr int LDS_ALL = PARK_BASE + 256 * PARK > LDS_ELEMS ? PARK_BASE + 256 * PARK : LDS_ELEMS;
  __shared__ __attribute__((aligned(16))) T lds[LDS_ALL];
  const int tid = threadIdx.x;
  const int ch = blockIdx.y;
  const DevChannel C = a.channels[ch];
  const cx<T>* hspec = static_cast<const cx<T>*>(a.hspec) + (int64_t)ch * a.hrow;
  const cx<T>* tw = static_cast<const cx<T>*>(a.tw);
  T* const orow = static_cast<T*>(a.out) + (int64_t)ch * a.out_stride;
  const int M = a.hop;
  const int64_t pair = blockIdx.x;
  const int64_t s1 = 2 * pair * (int64_t)a.hop - a.lead;   // first sample of the first window
  const int64_t range_end = s1 + 256 * (int64_t)CL;
  const double base = C.offset;

  // both halves' window descriptors and first entry words up front: the dependent chain
  // descriptor -> entry -> op record is three memory round trips, two of them are taken here for
  // both halves at once
  const ShortWin* const wp0 = a.wins + ((int64_t)ch * a.npairs + pair) * 2;
  int64_t w_rec0[2], w_e0[2];
  int w_cnt[2], w_pad[2], w_ccnt[2];
  uint32_t w_first[2];
  CH_EACH(2, half)
    w_rec0[half] = cuni64(wp0[half].rec0);               // (block-uniform: SGPRs; as VGPR pairs they were 8 of the
    w_e0[half] = cuni64(wp0[half].e0);                   //  registers this kernel spilled)
    w_cnt[half] = cuni(wp0[half].cnt);
    w_pad[half] = cuni(wp0[half].pad);
    w_ccnt[half] = cuni(wp0[half].ccnt);
    w_first[half] = tid < w_cnt[half] ? a.entries[w_e0[half] + tid] : 0u;
  CH_END

  // (The third, the op record, is loaded where it is used.  Tried: half 1's record loaded a half ahead --
  // 24 more live VGPRs on top of 168, 45 spilled, 1.05 -> 1.32 ms; one dword of each record loaded here to
  // warm L2 -- 1.045 vs 1.052 ms, nothing.  Timing experiments on 2048 x 1e5 (tools/experiments/timing_switches.patch): without the
  // evaluation 0.73 ms, i.e. the sampling phases cost 0.32 ms against 0.23 ms of pure VALU issue for
  // their 37 instructions per sample: they are issue-bound like the transform, not latency-bound.)

  T x[CL];                                                 // the thread's samples: s1 + tid + 256 k
  CH_EACH(2, half)
    const int64_t h0 = s1 + (int64_t)HALF * half;
    __syncthreads();                                       // the previous half has been picked up
    // Layout of the half in the array, chosen by the host for the half's own entries: plain (sample i at
    // element i: conflict-free when the runs start an odd number of samples apart, 15 for 60-sample
    // pulses) or padded by one element per 16 (i + (i >> 4): runs of 16 then start 17 apart).  The
    // wrong one makes every write of a wave hit one bank pair (PMC: 27 % of the LDS pipe's time).
    const bool sw = w_pad[half] != 0;
    T* const mine = lds + (sw ? tid + (tid >> 4) : tid);   // the thread's own samples: rows 272 / 256 apart
    // (the two layouts as two copies of the row loops, under a block-uniform branch: the row offsets stay
    //  immediates of the LDS instructions -- with a variable row stride the kernel executed 10 % more VALU
    //  instructions, PMC 355.7 M -> 390.4 M per launch)
    // zero padding outside [0, n), the channel offset inside (skipped _zero pieces, gaps between entries)
    const bool inside = h0 >= 0 && h0 + HALF <= a.n;
    auto prefill = [&](auto rs_) __attribute__((always_inline)) {
      constexpr int rs = decltype(rs_)::value;
      if (inside) {
        CH_EACH(HR, k) mine[rs * k] = (T)base; CH_END
      } else {
        CH_EACH(HR, k)
          const int64_t j = h0 + 256 * k + tid;
          mine[rs * k] = (j >= 0 && j < a.n) ? (T)base : (T)0;
        CH_END
      }
    };
    if (sw) prefill(std::integral_constant<int, 272>{});
    else prefill(std::integral_constant<int, 256>{});
    __syncthreads();
    const int64_t rec0 = w_rec0[half], e0 = w_e0[half];
    const int cnt = w_cnt[half];
    for (int eb = 0; eb < cnt; eb += 256) {                // block-uniform trip count
      const int idx = eb + tid;
      bool live = idx < cnt;
      const uint32_t word = eb == 0 ? w_first[half] : (live ? a.entries[e0 + idx] : 0u);
      const int len = live ? (int)(word >> 28) + 1 : 0;
      const int o = (int)((word >> 16) & 0xfff);
      const double* op = a.recs + 2 * (rec0 + (int64_t)(word & 0xffff));
      shdev::OpRec rec = shdev::load_op(op);
      const double kf = (double)((int)(uint32_t)(h0 + o) - shdev::op_ref(rec));   // samples from the record's reference sample
      double acc[R], acci[1];
      CH_EACH(R, k) acc[k] = 0.0; CH_END
      acci[0] = 0.0;
      auto eval = [&](const shdev::OpRec& rc, const double* opp, bool lv) -> bool {   // -> another op follows
        const int w = shdev::op_word(rc);
        const bool closing = ((w >> 4) & 3) == 3;          // erf edge multiplier
        const bool mine = lv && !closing && !(w & 8);      // (real channels only: no op of an imaginary part)
        const bool cubic = __any(mine && (w & 3) > 1);
        if (mine) {
          if (cubic) shdev::short_op<R, true, false>(rc, opp, w, kf, a.step, acc, acci);
          else shdev::short_op<R, false, false>(rc, opp, w, kf, a.step, acc, acci);
        }
        if (__any(lv && closing)) {
          if (lv && closing) shdev::short_erfmul<R, false>(rc, kf, acc, acci);
        }
        return lv && !(w & WFK_SH_LAST);
      };
      live = eval(rec, op, live);
      while (__any(live)) {
        op += (shdev::op_word(rec) & 3) > 1 ? WFK_SH_OP3 : WFK_SH_OP1;
        if (live) rec = shdev::load_op(op);
        live = eval(rec, op, live);
      }
      if (C.do_clip) {
        CH_EACH(R, k) acc[k] = shdev::clip_np(acc[k], C.clip_lo, C.clip_hi); CH_END
      }
      // element o + k of the half sits at (o + k) + ((o + k) >> 4) = swz(o) + k + [k >= 16 - (o & 15)]
      T* const b0 = lds + (sw ? o + (o >> 4) : o);
      const int t = sw ? 16 - (o & 15) : 99;
      CH_EACH(R, k)
        if (k < len) *((k >= t ? b0 + 1 : b0) + k) = (T)(acc[k] + base);
      CH_END
    }
    // pieces the short tier cannot take (mixed plans): their samples -- offset and clip applied -- were
    // written to the workspace by the general kernel in the launch before this one; copy the runs in
    if (__builtin_expect(w_ccnt[half] != 0, 0))
      chain_copy_runs<T>(lds, static_cast<const T*>(a.ws) + (int64_t)ch * a.ws_stride + h0, a.entries + e0 + cnt,
                         w_ccnt[half], sw, tid);
    __syncthreads();
    // pick up the thread's stride-256 samples; the last PARK rows of the FIRST half go to the parking rows
    // (written and read back by the same thread: no barrier of their own)
    auto pickup = [&](auto rs_) __attribute__((always_inline)) {
      constexpr int rs = decltype(rs_)::value;
      CH_EACH(HR, k)
        if constexpr (half == 0 && k >= HR - PARK) lds[PARK_BASE + 256 * (k - (HR - PARK)) + tid] = mine[rs * k];
        else x[half * HR + k] = mine[rs * k];
      CH_END
    };
    if (sw) pickup(std::integral_constant<int, 272>{});
    else pickup(std::integral_constant<int, 256>{});
  CH_END
  CH_EACH(PARK, k) x[HR - PARK + k] = lds[PARK_BASE + 256 * k + tid]; CH_END
  __syncthreads();                                         // the array becomes the FFT exchange buffer

  cx<T> v[16];
  CH_EACH(16, n1)
    v[n1].x = x[n1];
    v[n1].y = x[n1 + HOPB];
  CH_END
  const int64_t b1 = 2 * pair, b2 = b1 + 1;
  const bool interior = s1 >= 0 && range_end <= a.n && (b2 + 1) * (int64_t)M <= a.n;

  // ---- transform, multiply by the kernel spectrum, inverse transform (as fir_sampled) ---------------
  int tf = tid;
  asm volatile("" : "+v"(tf));
  const cx<T> wa = tw[tf], wb = tw[16 * (tf & 15)];
  fft4096<false>(v, lds, wa, wb, tf);
#pragma unroll
  for (int k3 = 0; k3 < 16; ++k3) v[k3] = cmul(v[k3], hspec[tf + 256 * k3]);
  fft4096<true>(v, lds, wa, wb, tf);
  if (interior) {
    T* const o1 = orow + b1 * M + (tf - (a.K - 1));
    T* const o2 = o1 + M;
#pragma unroll
    for (int q3 = 0; q3 < 16; ++q3) {
      const int r = tf + 256 * q3 - (a.K - 1);
      if (r >= 0 && r < M) {
        o1[256 * q3] = v[q3].x;
        o2[256 * q3] = v[q3].y;
      }
    }
  } else {
#pragma unroll
    for (int q3 = 0; q3 < 16; ++q3) {
      const int r = tf + 256 * q3 - (a.K - 1);
      if (r >= 0 && r < M) {
        const int64_t d1 = b1 * M + r, d2 = b2 * M + r;
        if (d1 < a.n) orow[d1] = v[q3].x;
        if (d2 < a.n) orow[d2] = v[q3].y;
      }
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
  // fused path at AWG rates (fir_short): half-window entry lists over the sampler plan's own op records
  bool shortw = false, hybrid = false;   // hybrid: a mixed short plan -- the general kernel's pieces go through the workspace
  ShortWin* d_wins = nullptr;
  uint32_t* d_entries = nullptr;
  const double* d_recs = nullptr;
  int64_t table_bytes = 0;
  double t0 = 0, step = 0, last = 0;
  int32_t has_last = 0;
  int64_t i0 = 0;
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

static int chain_plan_create(const wfk_program* prog, const wfk_grid* grid, const double* ker_host, int32_t K,
                             int kind, wfk_chain_plan** out, bool per_row);

int wfk_chain_plan_create(const wfk_program* prog, const wfk_grid* grid, const double* ker_host, int32_t K,
                          int kind, wfk_chain_plan** out) {
  return chain_plan_create(prog, grid, ker_host, K, kind, out, false);
}

int wfk_chain_plan_create_rows(const wfk_program* prog, const wfk_grid* grid, const double* kers_host, int32_t K,
                               int kind, wfk_chain_plan** out) {
  return chain_plan_create(prog, grid, kers_host, K, kind, out, true);
}

static int chain_plan_create(const wfk_program* prog, const wfk_grid* grid, const double* ker_host, int32_t K,
                             int kind, wfk_chain_plan** out, bool per_row) {
  if (!out) return chain_fail(WFK_EINVAL, "null out");
  *out = nullptr;
  if (!prog || !grid || !ker_host) return chain_fail(WFK_EINVAL, "null argument");
  if (kind != WFK_OUT_F64 && kind != WFK_OUT_F32) return chain_fail(WFK_EINVAL, "chain kind must be F64 or F32");
  wfk_chain_plan* p = new wfk_chain_plan();
  p->kind = kind;
  wfk_internal_keep_mixed_short(true);       // (fir_short samples the short pieces itself: a mixed short plan stays one)
  int rc = wfk_plan_create_grid(prog, grid, &p->sampler);
  wfk_internal_keep_mixed_short(false);
  if (rc) { wfk_chain_plan_destroy(p); return rc; }
  {
    // pieces that close with a table / mollifier multiplier: fir_short does not evaluate those, the general kernel
    // writes them to the chain's workspace like the other pieces the short tier hands on
    const HostPlan* h0 = nullptr;
    const double* r0 = nullptr;
    wfk_internal_plan_tables(p->sampler, &h0, &r0);
    if (h0 && h0->shortp && h0->short_has_fmul) {
      wfk_plan_destroy(p->sampler);
      p->sampler = nullptr;
      wfk_internal_no_short_fmul(true);
      wfk_internal_keep_mixed_short(true);
      rc = wfk_plan_create_grid(prog, grid, &p->sampler);
      wfk_internal_keep_mixed_short(false);
      wfk_internal_no_short_fmul(false);
      if (rc) { wfk_chain_plan_destroy(p); return rc; }
    }
  }
  p->n = grid->n;
  p->n_channels = prog->n_channels;
  rc = per_row ? wfk_fir_plan_create_rows(ker_host, K, grid->n, std::max(1, prog->n_channels), kind, &p->fir)
               : wfk_fir_plan_create(ker_host, K, grid->n, std::max(1, prog->n_channels), kind, &p->fir);
  if (rc) { wfk_chain_plan_destroy(p); return rc; }
  p->t0 = grid->t0; p->step = grid->step; p->last = grid->last; p->has_last = grid->has_last; p->i0 = grid->i0;
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
  else if (!H.lean) p->why = "plan is not fully fused (generic / direct terms -- erf edges and exponential envelopes included, which this kernel does not evaluate -- or too many ops per piece)";
  else {
    for (const DevChannel& c : H.channels)
      if (c.do_clip) p->why = "clip (min/max) on a channel";
    for (uint8_t cx_ : H.channel_complex)
      if (cx_) p->why = "complex-valued channel";
  }
  if (!p->why.empty() && fir_fused && nseg == 1 && prog->n_channels <= 65535 && !(off && off[0] == '1')) {
    // ---- AWG rates: the sampler's plan is in the short geometry -> fir_short ------------------------
    const HostPlan* hs = nullptr;
    const double* d_recs = nullptr;
    wfk_internal_plan_tables(p->sampler, &hs, &d_recs);
    bool real = true;
    if (hs) for (uint8_t cx_ : hs->channel_complex) real = real && !cx_;
    if (hs && d_recs && hs->shortp && real) {
      const int64_t hop = 256 * (int64_t)p->hopb, HALF = 128 * (int64_t)(16 + p->hopb);
      const int64_t nblk = (p->n + hop - 1) / hop;
      p->npairs = (nblk + 1) / 2;
      std::vector<ShortWin> wins;
      std::vector<uint32_t> ents;
      std::string bad;
      (void)wfk_chain_windows(*hs, p->n, hop, lead, HALF, p->npairs, wins, ents, bad);
      if (bad.empty()) {
        if (ents.empty()) ents.push_back(0);
        auto al = [](size_t x) { return (x + 255) & ~size_t(255); };
        const size_t b_ch = hs->channels.size() * sizeof(DevChannel), b_w = wins.size() * sizeof(ShortWin),
                     b_e = ents.size() * sizeof(uint32_t);
        const size_t o_w = al(b_ch), o_e = al(o_w + b_w), total = al(o_e + b_e) + 256;
        if (hipMalloc(&p->d_tables, total) != hipSuccess ||
            hipMemcpy(p->d_tables, hs->channels.data(), b_ch, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(static_cast<char*>(p->d_tables) + o_w, wins.data(), b_w, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(static_cast<char*>(p->d_tables) + o_e, ents.data(), b_e, hipMemcpyHostToDevice) != hipSuccess) {
          wfk_chain_plan_destroy(p);
          return chain_fail(WFK_ENOMEM, "chain table allocation failed");
        }
        char* base = static_cast<char*>(p->d_tables);
        p->d_channels = reinterpret_cast<DevChannel*>(base);
        p->d_wins = reinterpret_cast<ShortWin*>(base + o_w);
        p->d_entries = reinterpret_cast<uint32_t*>(base + o_e);
        p->d_recs = d_recs;
        p->table_bytes = (int64_t)(b_ch + b_w + b_e + hs->params.size() * sizeof(double));
        if (hs->mixed) {
          // the general kernel's pieces (few, by construction: a plan dominated by them is not short) travel
          // through rows of a workspace, written sparsely in a launch of their own
          const size_t es = kind == WFK_OUT_F32 ? 4 : 8;
          if (hipMalloc(&p->workspace, (size_t)p->n_channels * (size_t)p->n * es) != hipSuccess) {
            wfk_chain_plan_destroy(p);
            return chain_fail(WFK_ENOMEM, "chain workspace allocation failed");
          }
          p->hybrid = true;
        }
        p->shortw = true;
        p->fused = true;
        p->why.clear();
        *out = p;
        return WFK_OK;
      }
      p->why += "; short geometry: " + bad;
    }
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
    p->table_bytes = (int64_t)total;
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

int64_t wfk_chain_table_bytes(const wfk_chain_plan* p) {
  if (!p) return chain_fail(WFK_EINVAL, "null plan");
  return p->fused ? p->table_bytes : wfk_plan_table_bytes(p->sampler);
}

const char* wfk_chain_kernel_name(const wfk_chain_plan* p) {
  if (!p) return "";
  static thread_local std::string name;
  const char* T = p->kind == WFK_OUT_F32 ? "float" : "double";
  if (p->fused) {
    name = std::string(p->shortw ? "fir_short<" : "fir_sampled<") + T + "," + std::to_string(p->hopb) + ">";
    if (p->hybrid) name = "wfk_sample<...> (pieces without a short form) + " + name;
  }
  else name = std::string(wfk_plan_kernel_name(p->sampler, p->kind)) + " + FIR";
  return name.c_str();
}

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
  if (p->shortw) {
    if (p->hybrid) {
      const int rc = wfk_internal_plan_launch_foreign(p->sampler, p->workspace, p->n, p->kind, hip_stream);
      if (rc) return rc;
    }
    ChainShortArgs a{};
    a.ws = p->workspace; a.ws_stride = p->n;
    a.channels = p->d_channels; a.wins = p->d_wins; a.entries = p->d_entries; a.recs = p->d_recs;
    a.out = out_dev; a.out_stride = out_stride; a.n = p->n; a.npairs = p->npairs;
    a.hspec = kspec; a.tw = tw; a.step = p->step; a.hrow = wfk_internal_fir_krow(p->fir);
    a.hop = 256 * p->hopb; a.K = K; a.lead = lead;
    const dim3 grid((unsigned)p->npairs, (unsigned)p->n_channels);
    if (p->kind == WFK_OUT_F32) {
      if (p->hopb == 12) hipLaunchKernelGGL((fir_short<float, 12>), grid, dim3(256), 0, s, a);
      else hipLaunchKernelGGL((fir_short<float, 10>), grid, dim3(256), 0, s, a);
    } else {
      if (p->hopb == 12) hipLaunchKernelGGL((fir_short<double, 12>), grid, dim3(256), 0, s, a);
      else hipLaunchKernelGGL((fir_short<double, 10>), grid, dim3(256), 0, s, a);
    }
    if (hipGetLastError() != hipSuccess) return chain_fail(WFK_EHIP, "fused sampler->FIR kernel launch failed");
    return WFK_OK;
  }
  ChainArgs a{};
  a.channels = p->d_channels; a.pieces = p->d_pieces; a.params = p->d_params; a.pair_first = p->d_pair_first;
  a.out = out_dev; a.out_stride = out_stride; a.n = p->n; a.npairs = p->npairs;
  a.hspec = kspec; a.tw = tw; a.hrow = wfk_internal_fir_krow(p->fir);
  a.t0 = p->t0; a.step = p->step; a.last = p->last; a.has_last = p->has_last; a.i0 = p->i0;
  a.hop = 256 * p->hopb; a.K = K; a.lead = lead;
  const dim3 grid((unsigned)((p->npairs + WFK_FIRS_PPW - 1) / WFK_FIRS_PPW), (unsigned)p->n_channels);
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
