// wfk_iir.hip -- IIR stage: scipy.signal.sosfilt / lfilter semantics
// applied along every row, as used by Waveform.sample(filters=(sos, initial))
// (reference: waveforms/waveform.py:193-203, 244-251) and predistort(filters=...)
// (waveforms/distortion.py:298-321).
//
// A filter is a cascade of direct-form-II-transposed sections (sosfilt: order-2 sections;
// lfilter: one section of order max(len(a),len(b))-1), per sample and section:
//     y = b0*x + z[0];   z[i] = b[i+1]*x - a[i+1]*y + z[i+1];   z[ord-1] = b[ord]*x - a[ord]*y
// The recurrence is sequential in time; it is linear, so rows are cut into blocks of LB
// samples and solved in three phases (state dimension D = sum of section orders <= 16):
//   A  iir_pass<false>: every block from ZERO state -> its local final state f_b; the wave
//                       (= one GROUP of 64 consecutive blocks) then scans them in registers,
//                       v_l = sum_{j<=l} T^(l-j) f_j  (Hillis-Steele, T^(2^k) tables), and
//                       writes the group-local prefixes v_l and the group total v_63
//   B  iir_scan       : true state at the START of every group from the group totals: the
//                       same scan one level up (U = T^64), one wave per row; 1/64 of the
//                       blocks, so its serial part no longer shows (it was 28% of the stage)
//   C  iir_pass<true> : every block again, from its true initial state
//                       S = v_{l-1} + T^l * (state at its group's start); writes y
// (T = LB-step transition matrix, computed on the host by running the homogeneous system.)
// HBM traffic 24 B/sample fp64 (x read twice, y written once) vs 16 algorithmic.
// Lanes own blocks; a wave moves 64x64-sample tiles through LDS (transposed) so that
// global loads/stores are 512-byte contiguous per instruction.  State is always double.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "wfk.h"
#include "wfk_chain_dev.h"
#include "wfk_internal.h"

#define IIR_MAXD 16
#define IIR_MAXSEC 8
#define IIR_LB 2048   // block length (samples)
#define IIR_SEG 16    // samples per block per LDS tile (64 blocks x 16 samples = 8.7 KB)

extern "C" void wfk_internal_set_error(const char* msg);

namespace {

struct IirCoef {
  int nsec;
  int ord[IIR_MAXSEC];
  int off[IIR_MAXSEC];          // state offset of each section
  double b[IIR_MAXSEC][IIR_MAXD + 1];
  double a[IIR_MAXSEC][IIR_MAXD + 1];
  int D;
};

__host__ __device__ inline double iir_step(const IirCoef& c, double x, double* z) {
  for (int s = 0; s < c.nsec; ++s) {
    double* zs = z + c.off[s];
    const int ord = c.ord[s];
    const double y = c.b[s][0] * x + (ord > 0 ? zs[0] : 0.0);
    for (int i = 0; i + 1 < ord; ++i) zs[i] = c.b[s][i + 1] * x - c.a[s][i + 1] * y + zs[i + 1];
    if (ord > 0) zs[ord - 1] = c.b[s][ord] * x - c.a[s][ord] * y;
    x = y;
  }
  return x;
}

// Compile-time shaped cascade (NSEC sections of order ORD each): every index is a
// constant, so the state stays in registers.  NSEC == 0 selects the generic runtime-shaped
// step (any mix of orders; state array then lives in scratch: correct, slower).
template <int NSEC, int ORD>
__device__ __forceinline__ double iir_step_t(const IirCoef& c, double x, double (&z)[IIR_MAXD]) {
  if constexpr (NSEC == 0) {
    return iir_step(c, x, z);
  } else {
#pragma unroll
    for (int s = 0; s < NSEC; ++s) {
      const double y = c.b[s][0] * x + z[s * ORD];
#pragma unroll
      for (int i = 0; i + 1 < ORD; ++i)
        z[s * ORD + i] = c.b[s][i + 1] * x - c.a[s][i + 1] * y + z[s * ORD + i + 1];
      z[s * ORD + ORD - 1] = c.b[s][ORD] * x - c.a[s][ORD] * y;
      x = y;
    }
    return x;
  }
}

// (sh + sl) += (th + tl) * x in double-double (TwoProd via fma, TwoSum)
__device__ __forceinline__ void dd_acc(double& sh, double& sl, double th, double tl, double x) {
  const double p = th * x;
  const double e = fma(th, x, -p) + tl * x;
  const double s = sh + p;
  const double bb = s - sh;
  sl += ((sh - (s - bb)) + (p - bb)) + e;
  sh = s;
}

// Inclusive scan over the 64 lanes of a wave: v_l <- sum_{j<=l} M^(l-j) v_j, with
// pw[k] = M^(2^k) as (hi, lo) pairs.  DD > 0: compile-time state dimension (registers).
// M has entries ~1e5 that cancel against each other when applied to a state (see the host
// note): the mat-vecs therefore run in double-double, so a block boundary perturbs the
// state no more than one step of the sequential filter does.
template <int DD>
__device__ __forceinline__ void wave_scan(double (&v)[IIR_MAXD], const double* __restrict__ pw,
                                          int D_rt, int lane) {
  const int D = DD > 0 ? DD : D_rt;
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const int d = 1 << k;
    double u[IIR_MAXD];
#pragma unroll
    for (int i = 0; i < (DD > 0 ? DD : IIR_MAXD); ++i) u[i] = i < D ? __shfl_up(v[i], d) : 0.0;
    if (lane >= d) {
      const double* M = pw + (int64_t)k * D * D * 2;
      double nv[IIR_MAXD];
#pragma unroll
      for (int i = 0; i < (DD > 0 ? DD : IIR_MAXD); ++i) {
        if (i < D) {
          double sh = v[i], sl = 0.0;
#pragma unroll
          for (int j = 0; j < (DD > 0 ? DD : IIR_MAXD); ++j)
            if (j < D) dd_acc(sh, sl, M[(i * D + j) * 2], M[(i * D + j) * 2 + 1], u[j]);
          nv[i] = sh + sl;
        }
      }
#pragma unroll
      for (int i = 0; i < (DD > 0 ? DD : IIR_MAXD); ++i)
        if (i < D) v[i] = nv[i];
    }
  }
}

// v += M * c in double-double; M = D x D (hi, lo) pairs
template <int DD>
__device__ __forceinline__ void dd_matvec_add(double (&v)[IIR_MAXD], const double* __restrict__ M,
                                              const double (&c)[IIR_MAXD], int D_rt) {
  const int D = DD > 0 ? DD : D_rt;
  double nv[IIR_MAXD];
#pragma unroll
  for (int i = 0; i < (DD > 0 ? DD : IIR_MAXD); ++i) {
    if (i < D) {
      double sh = v[i], sl = 0.0;
#pragma unroll
      for (int j = 0; j < (DD > 0 ? DD : IIR_MAXD); ++j)
        if (j < D) dd_acc(sh, sl, M[(i * D + j) * 2], M[(i * D + j) * 2 + 1], c[j]);
      nv[i] = sh + sl;
    }
  }
#pragma unroll
  for (int i = 0; i < (DD > 0 ? DD : IIR_MAXD); ++i)
    if (i < D) v[i] = nv[i];
}

// One wave per 64 consecutive blocks of one row.  WRITE=false: zero initial state, emit the
// final state of each block.  WRITE=true: initial state from `init`, write y (+post).
template <typename T, bool WRITE, int NSEC, int ORD>
__global__ void __launch_bounds__(64) iir_pass(const IirCoef c, const T* __restrict__ in,
                                               int64_t in_stride, T* __restrict__ out,
                                               int64_t out_stride, double* __restrict__ state,
                                               double* __restrict__ grp,
                                               const double* __restrict__ pw,
                                               const double* __restrict__ lanep,
                                               double* __restrict__ zf, int64_t n, int64_t nblk,
                                               double pre_sub, double post_add) {
  constexpr int DD = NSEC * ORD;   // compile-time state dimension (0: runtime-shaped)
  __shared__ T tile[64][IIR_SEG + 1];
  const int lane = threadIdx.x;
  const int64_t row = blockIdx.y;
  const int64_t blk0 = (int64_t)blockIdx.x * 64;
  const int64_t blk = blk0 + lane;
  const T* x = in + row * in_stride;
  T* y = WRITE ? out + row * out_stride : nullptr;
  const int D = c.D;
  double z[IIR_MAXD];
#pragma unroll
  for (int i = 0; i < IIR_MAXD; ++i) z[i] = 0.0;
  const int64_t ngrp = gridDim.x;
  if (WRITE && blk < nblk) {
    // true initial state: the state at the group's start (phase B) pushed through the l blocks
    // before this one, plus the group-local prefix of the previous block (phase A)
    double carry[IIR_MAXD];
#pragma unroll
    for (int i = 0; i < IIR_MAXD; ++i) carry[i] = i < D ? grp[(row * ngrp + blockIdx.x) * D + i] : 0.0;
    if (lane == 0) {
#pragma unroll
      for (int i = 0; i < IIR_MAXD; ++i) z[i] = carry[i];
    } else {
      const double* s = state + (row * nblk + blk - 1) * D;
#pragma unroll
      for (int i = 0; i < IIR_MAXD; ++i)
        if (i < D) z[i] = s[i];
      dd_matvec_add<DD>(z, lanep + (int64_t)(lane - 1) * D * D * 2, carry, D);
    }
  }
  const int nact = (int)(nblk - blk0 < 64 ? nblk - blk0 : 64);   // active blocks in this wave
  constexpr int RPI = 64 / IIR_SEG;      // tile rows (blocks) moved per wave instruction
  const int rsub = lane / IIR_SEG, col = lane % IIR_SEG;
  // software pipeline: the global loads of segment seg+1 are in flight while segment seg
  // is being filtered out of LDS
  constexpr int NSEG = IIR_LB / IIR_SEG, NLD = 64 / RPI;
  T pre[NLD];
  auto fetch = [&](int seg) {
#pragma unroll
    for (int q = 0; q < NLD; ++q) {
      const int r = q * RPI + rsub;
      const int64_t j = (blk0 + r) * IIR_LB + (int64_t)seg * IIR_SEG + col;
      pre[q] = (r < nact && j < n) ? x[j] : (T)0;
    }
  };
  fetch(0);
  for (int seg = 0; seg < NSEG; ++seg) {
#pragma unroll
    for (int q = 0; q < NLD; ++q) tile[q * RPI + rsub][col] = pre[q];
    if (seg + 1 < NSEG) fetch(seg + 1);
    __syncthreads();
    if (blk < nblk) {
      const int64_t base = blk * IIR_LB + seg * IIR_SEG;
      const int lim = (int)(n - base < IIR_SEG ? (n - base > 0 ? n - base : 0) : IIR_SEG);
      if (lim == IIR_SEG) {
#pragma unroll
        for (int i = 0; i < IIR_SEG; ++i) {
          const double v = iir_step_t<NSEC, ORD>(c, (double)tile[lane][i] - pre_sub, z);
          if (WRITE) tile[lane][i] = (T)(v + post_add);
        }
      } else {
        for (int i = 0; i < lim; ++i) {
          const double v = iir_step_t<NSEC, ORD>(c, (double)tile[lane][i] - pre_sub, z);
          if (WRITE) tile[lane][i] = (T)(v + post_add);
        }
      }
    }
    __syncthreads();
    if (WRITE) {
#pragma unroll 4
      for (int r0 = 0; r0 < 64; r0 += RPI) {
        const int r = r0 + rsub;
        const int64_t j = (blk0 + r) * IIR_LB + seg * IIR_SEG + col;
        if (r < nact && j < n) y[j] = tile[r][col];
      }
      __syncthreads();
    }
  }
  if (!WRITE) {
    // scan the 64 local final states of this group in registers (lanes past the last block
    // hold zeros and nothing downstream reads their results)
    wave_scan<DD>(z, pw, D, lane);
    if (blk < nblk) {
      double* s = state + (row * nblk + blk) * D;
#pragma unroll
      for (int i = 0; i < IIR_MAXD; ++i)
        if (i < D) s[i] = z[i];
    }
    if (lane == 63) {
      double* s = grp + (row * ngrp + blockIdx.x) * D;
#pragma unroll
      for (int i = 0; i < IIR_MAXD; ++i)
        if (i < D) s[i] = z[i];
    }
  } else if (blk == nblk - 1 && zf) {
    for (int i = 0; i < D; ++i) zf[row * D + i] = z[i];
  }
}

// Phase B: in place, items[row][g] (group totals) -> the true state at the START of group g.
// pw: [7][D][D][2] = U^(2^k) as (hi, lo) pairs, lanep: [64][D][D][2] = U^(l+1), U = T^64.
// zi: [row][D] or null.  One wave per row, 64 groups (= 4096 blocks = 8.4M samples) per step.
__global__ void __launch_bounds__(64) iir_scan(double* __restrict__ state, const double* __restrict__ pw,
                                               const double* __restrict__ lanep,
                                               const double* __restrict__ zi, int64_t nblk, int D) {
  const int lane = threadIdx.x;
  const int64_t row = blockIdx.x;
  double carry[IIR_MAXD];   // S at the start of the current group (wave-uniform)
  for (int i = 0; i < IIR_MAXD; ++i) carry[i] = (zi && i < D) ? zi[row * D + i] : 0.0;
  for (int64_t g0 = 0; g0 < nblk; g0 += 64) {
    const int64_t b = g0 + lane;
    double v[IIR_MAXD];
    for (int i = 0; i < IIR_MAXD; ++i) v[i] = 0.0;
    if (b < nblk)
      for (int i = 0; i < D; ++i) v[i] = state[(row * nblk + b) * D + i];
    // inclusive scan: v_l = sum_{j<=l} T^(l-j) f_j
    for (int k = 0; k < 6; ++k) {
      const int d = 1 << k;
      double u[IIR_MAXD];
      for (int i = 0; i < IIR_MAXD; ++i) u[i] = i < D ? __shfl_up(v[i], d) : 0.0;
      if (lane >= d) {
        const double* M = pw + (int64_t)k * D * D * 2;
        double nv[IIR_MAXD];
        for (int i = 0; i < D; ++i) {
          double sh = v[i], sl = 0.0;
          for (int j = 0; j < D; ++j) dd_acc(sh, sl, M[(i * D + j) * 2], M[(i * D + j) * 2 + 1], u[j]);
          nv[i] = sh + sl;
        }
        for (int i = 0; i < D; ++i) v[i] = nv[i];
      }
    }
    // S_b (state at END of block b) = v_l + T^(l+1) * carry
    {
      const double* M = lanep + (int64_t)lane * D * D * 2;
      double nv[IIR_MAXD];
      for (int i = 0; i < D; ++i) {
        double sh = v[i], sl = 0.0;
        for (int j = 0; j < D; ++j) dd_acc(sh, sl, M[(i * D + j) * 2], M[(i * D + j) * 2 + 1], carry[j]);
        nv[i] = sh + sl;
      }
      for (int i = 0; i < D; ++i) v[i] = nv[i];
    }
    // state at the START of block b = end state of block b-1 (lane 0: carry)
    double st[IIR_MAXD];
    for (int i = 0; i < IIR_MAXD; ++i) {
      const double up = i < D ? __shfl_up(v[i], 1) : 0.0;
      st[i] = lane == 0 ? carry[i] : up;
    }
    if (b < nblk)
      for (int i = 0; i < D; ++i) state[(row * nblk + b) * D + i] = st[i];
    for (int i = 0; i < IIR_MAXD; ++i) carry[i] = i < D ? __shfl(v[i], 63) : 0.0;
  }
}


// ---- single pass: chained scan with decoupled look-back (state dimension <= 4) -----------------
// The three-launch form reads x twice (24 B/sample for 16 algorithmic).  Here a wave owns a CHUNK of
// 64 lane-blocks of OP_LB = 32 samples and keeps them in registers between the two sweeps:
//   1. load the chunk (coalesced, transposed through LDS into lane blocks), sweep from ZERO state,
//      in-wave scan of the 64 block states (T1 = 32-step transition; T1^(2^k) tables) -> the
//      chunk AGGREGATE (state at its end from a zero state at its start), published with flag 1;
//   2. state at the chunk's start: look back over the preceding chunks of the row -- lane k reads
//      chunk c-1-k; the nearest one that has published its inclusive PREFIX (flag 2) closes the sum
//      S_in = sum_{k<m} U^k agg_{c-1-k} + U^m prefix_{c-1-m}   (U = T1^64, powers from a table),
//      evaluated by the 64 lanes in parallel (double-double mat-vecs) and reduced over the wave;
//   3. publish the prefix agg_c + U S_in (flag 2); sweep again from the true block states
//      v_{l-1} + T1^l S_in, write y (transposed back through LDS).
// Chunks are handed out by an atomic TICKET per row (workgroup b serves row b % rows and takes that
// row's next chunk): every predecessor a wave waits for holds a smaller ticket of the same row, i.e.
// is already running or done, so the waits cannot deadlock; they are bounded all the same (OP_SPIN polls, then NaN).
// Flags and states travel as device-scope atomics (no cache-wide flush): a thread's state stores
// are acknowledged (vmcnt 0) before its flag store is issued.
#ifndef OP_LB
#define OP_LB 32
#endif
#define OP_CHUNK (64 * OP_LB)
#ifndef OP_WAVES
#define OP_WAVES 2
#endif
#define OP_SPIN (1 << 22)
#define OP_WINDOWS 4            // look-back reach: 256 chunks of the row in flight

__device__ __forceinline__ void op_store(double* p, double v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double op_load(const double* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// in-wave scan and mat-vec in plain double (hi words only): for block transitions whose powers up to
// T1^32 have entries of order 1 (no cancellation to protect), a tenth of the double-double work
template <int DD>
__device__ __forceinline__ void wave_scan_plain(double (&v)[IIR_MAXD], const double* __restrict__ pw, int lane) {
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const int d = 1 << k;
    double u[DD];
#pragma unroll
    for (int i = 0; i < DD; ++i) u[i] = __shfl_up(v[i], d);
    if (lane >= d) {
      const double* M = pw + (int64_t)k * DD * DD * 2;
#pragma unroll
      for (int i = 0; i < DD; ++i) {
        double acc = v[i];
#pragma unroll
        for (int j = 0; j < DD; ++j) acc = fma(M[(i * DD + j) * 2], u[j], acc);
        v[i] = acc;
      }
    }
  }
}
template <int DD>
__device__ __forceinline__ void matvec_add_plain(double (&v)[IIR_MAXD], const double* __restrict__ M,
                                                 const double (&c)[IIR_MAXD]) {
#pragma unroll
  for (int i = 0; i < DD; ++i) {
    double acc = v[i];
#pragma unroll
    for (int j = 0; j < DD; ++j) acc = fma(M[(i * DD + j) * 2], c[j], acc);
    v[i] = acc;
  }
}

template <typename T, int NSEC, int ORD, bool PLAIN>
__global__ void __launch_bounds__(64, OP_WAVES) iir_onepass(const IirCoef c, const T* __restrict__ in, int64_t in_stride,
                                                  T* __restrict__ out, int64_t out_stride,
                                                  unsigned* __restrict__ status, double* __restrict__ aggbuf,
                                                  double* __restrict__ prefbuf, unsigned* __restrict__ ticket,
                                                  const double* __restrict__ pw1, const double* __restrict__ lanep1,
                                                  const double* __restrict__ lanepU, const double* __restrict__ zi,
                                                  double* __restrict__ zf, int64_t n, int64_t nchunks, int rows,
                                                  unsigned epoch, double pre_sub, double post_add, int persist,
                                                  unsigned* __restrict__ fault, int spin_limit) {
  constexpr int DD = NSEC * ORD;       // state dimension (<= 4)
  __shared__ T tile[64][OP_LB + 1];
  const int lane = threadIdx.x;
  // one ticket counter per row (64 B apart): a single counter for all rows serialises 3e5 atomics on
  // one address -- measured 5.2 ms for the whole kernel against 3.1 ms of the three-launch form
  const int row = (int)(blockIdx.x % (unsigned)rows);
  // persist != 0: the grid holds a bounded number of waves per row and each walks the ticket counter
  // until the row is used up (few long rows: otherwise hundreds of chunks of one row are in flight
  // and every look-back reads all of them); persist == 0: one chunk per workgroup
  for (;;) {
  unsigned t = 0;
  if (lane == 0) t = atomicAdd(ticket + 16 * row, 1u);
  t = (unsigned)__builtin_amdgcn_readfirstlane((int)t);
  const int64_t chunk = t;
  if (chunk >= nchunks) return;
  const T* x = in + (int64_t)row * in_stride + chunk * OP_CHUNK;
  T* y = out + (int64_t)row * out_stride + chunk * OP_CHUNK;
  const int64_t left = n - chunk * OP_CHUNK;                 // samples of this row from the chunk's start
  const int cnt = (int)(left - (int64_t)lane * OP_LB < 0 ? 0 : (left - (int64_t)lane * OP_LB > OP_LB ? OP_LB : left - (int64_t)lane * OP_LB));
  const bool whole = left >= OP_CHUNK;                       // wave-uniform

  // ---- load: sample j = i*64 + lane of the chunk belongs to block j / OP_LB, position j % OP_LB
  // (a whole chunk loads unconditionally: with the bounds test inside the loop every load sits in
  //  its own branch and is waited for before the next one is issued -- the whole kernel then runs
  //  at a fraction of the copy rate, the stream probe of round 2, HISTORY)
  if (whole) {
    T v[OP_LB];
#pragma unroll
    for (int i = 0; i < OP_LB; ++i) v[i] = x[i * 64 + lane];
#pragma unroll
    for (int i = 0; i < OP_LB; ++i) {
      const int j = i * 64 + lane;
      tile[j / OP_LB][j % OP_LB] = v[i];
    }
  } else {
#pragma unroll
    for (int i = 0; i < OP_LB; ++i) {
      const int64_t j = (int64_t)i * 64 + lane;
      tile[(int)(j / OP_LB)][(int)(j % OP_LB)] = j < left ? x[j] : (T)0;
    }
  }
  __syncthreads();
  double xr[OP_LB];
#pragma unroll
  for (int i = 0; i < OP_LB; ++i) xr[i] = (double)tile[lane][i] - pre_sub;
  __syncthreads();

  // ---- sweep 1: zero state -> local final state; scan over the 64 blocks
  double z[IIR_MAXD];
#pragma unroll
  for (int i = 0; i < IIR_MAXD; ++i) z[i] = 0.0;
  if (whole) {
#pragma unroll
    for (int i = 0; i < OP_LB; ++i) (void)iir_step_t<NSEC, ORD>(c, xr[i], z);
  } else {
#pragma unroll
    for (int i = 0; i < OP_LB; ++i)
      if (i < cnt) (void)iir_step_t<NSEC, ORD>(c, xr[i], z);
  }
  // (a block past the end of the row leaves its state alone: T1^0; the scan below still multiplies
  //  by T1 per block, which only matters AFTER the last sample -- nothing there is used)
  if (PLAIN) wave_scan_plain<DD>(z, pw1, lane);
  else wave_scan<DD>(z, pw1, DD, lane);                      // z = v_l (inclusive)
  double vprev[DD], agg[DD];
#pragma unroll
  for (int i = 0; i < DD; ++i) {
    const double up = __shfl_up(z[i], 1);
    vprev[i] = lane == 0 ? 0.0 : up;
    agg[i] = __shfl(z[i], 63);
  }
  const int64_t slot = ((int64_t)row * nchunks + chunk);
  const unsigned F_AGG = epoch * 4u + 1u, F_PRE = epoch * 4u + 2u;
  if (chunk > 0 && lane == 0) {
#pragma unroll
    for (int i = 0; i < DD; ++i) op_store(aggbuf + slot * DD + i, agg[i]);
    __builtin_amdgcn_s_waitcnt(0);                           // the state is in memory before the flag is
    __hip_atomic_store(status + slot, F_AGG, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }

  // ---- state at the chunk's start
  double sin_[IIR_MAXD];
#pragma unroll
  for (int i = 0; i < IIR_MAXD; ++i) sin_[i] = 0.0;
  bool poisoned = false;
  if (chunk == 0) {
#pragma unroll
    for (int i = 0; i < DD; ++i) sin_[i] = zi ? zi[(int64_t)row * DD + i] : 0.0;
  } else {
    // Look back window by window (64 chunks each, lane l of window w looks at chunk c - 1 - 64 w - l).
    // A window is closed by the nearest chunk in it that has published its PREFIX, once every chunk
    // nearer than that one has at least its aggregate; a window in which all 64 chunks have aggregates
    // but none a prefix yet (more than 64 chunks of the row in flight: few rows, long rows) is summed
    // whole and the walk goes on one window further back (its sum then passes through U^64 once per
    // window).  Nothing waits for a PARTICULAR neighbour's prefix -- the first version did and chained
    // the chunks in flight one behind the other.  Bounded: OP_SPIN polls, then NaN.
    double hop[IIR_MAXD];                                    // sum over the windows so far
#pragma unroll
    for (int i = 0; i < IIR_MAXD; ++i) hop[i] = 0.0;
    int spins = 0;
    for (int w = 0;; ++w) {
      const int64_t pc = chunk - 1 - 64 * (int64_t)w - lane;
      const unsigned* f = status + (int64_t)row * nchunks + (pc >= 0 ? pc : 0);
      unsigned st = 0;
      int kstop = -1;
      bool whole_window = false;
      for (; spins < spin_limit; ++spins) {
        if (pc >= 0 && !(st == F_PRE)) st = __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool pre = pc >= 0 && st == F_PRE;
        const bool agg = pc >= 0 && (st == F_AGG || st == F_PRE);
        const unsigned long long has_pre = __ballot(pre), has_agg = __ballot(agg);
        if (has_pre != 0ull) {
          const int k = __ffsll((long long)has_pre) - 1;     // nearest chunk with a prefix
          const unsigned long long need = k == 0 ? 0ull : (~0ull >> (64 - k));   // lanes 0 .. k-1
          if ((has_agg & need) == need) { kstop = k; break; }
        } else if (has_agg == ~0ull && w < OP_WINDOWS - 1) {
          whole_window = true;
          kstop = 64;                                        // every lane contributes an aggregate
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
      if (kstop < 0) { poisoned = true; kstop = 0; }
      double contrib[IIR_MAXD];
#pragma unroll
      for (int i = 0; i < IIR_MAXD; ++i) contrib[i] = 0.0;
      if (lane <= kstop && pc >= 0) {
        const double* src = (lane == kstop ? prefbuf : aggbuf) + ((int64_t)row * nchunks + pc) * DD;
        double v[IIR_MAXD];
#pragma unroll
        for (int i = 0; i < IIR_MAXD; ++i) v[i] = i < DD ? op_load(src + i) : 0.0;
        if (lane == 0) {
#pragma unroll
          for (int i = 0; i < DD; ++i) contrib[i] = v[i];   // U^0
        } else {
          dd_matvec_add<DD>(contrib, lanepU + (int64_t)(lane - 1) * DD * DD * 2, v, DD);   // U^lane
        }
      }
      double wsum[IIR_MAXD];
#pragma unroll
      for (int i = 0; i < IIR_MAXD; ++i) wsum[i] = 0.0;
#pragma unroll
      for (int i = 0; i < DD; ++i) {
        double sum = contrib[i];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off);
        wsum[i] = sum;
      }
      for (int q = 0; q < w; ++q) {                          // this window lies 64 w chunks back: U^(64 w)
        double nx[IIR_MAXD];
#pragma unroll
        for (int i = 0; i < IIR_MAXD; ++i) nx[i] = 0.0;
        dd_matvec_add<DD>(nx, lanepU + (int64_t)63 * DD * DD * 2, wsum, DD);
#pragma unroll
        for (int i = 0; i < IIR_MAXD; ++i) wsum[i] = nx[i];
      }
#pragma unroll
      for (int i = 0; i < DD; ++i) hop[i] += wsum[i];
      if (!whole_window) break;
    }
#pragma unroll
    for (int i = 0; i < DD; ++i) sin_[i] = hop[i];
  }
  poisoned = __any(poisoned);
  // a look-back that ran out of polls (a predecessor preempted or stalled on a shared GPU): the chunk's
  // outputs become NaN AND the plan's fault word -- host memory, mapped -- is raised, so the failure
  // reaches the caller as an error code (wfk_iir_status / the next wfk_iir_apply), never as silent NaNs
  if (poisoned && lane == 0) __hip_atomic_fetch_or(fault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);

  // ---- publish the inclusive prefix: agg + U * S_in
  if (lane == 0) {
    double so[IIR_MAXD];
#pragma unroll
    for (int i = 0; i < IIR_MAXD; ++i) so[i] = i < DD ? agg[i < DD ? i : 0] : 0.0;
    dd_matvec_add<DD>(so, lanepU, sin_, DD);                 // U^1 (plain double where PLAIN allows it: 9.82 -> 9.65 ms at 256 x 1e7, inside the noise; not kept)
    if (chunk + 1 < nchunks) {
#pragma unroll
      for (int i = 0; i < DD; ++i) op_store(prefbuf + slot * DD + i, so[i]);
      __builtin_amdgcn_s_waitcnt(0);
      __hip_atomic_store(status + slot, F_PRE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }

  // ---- sweep 2 from the true block states: v_{l-1} + T1^l * S_in
#pragma unroll
  for (int i = 0; i < IIR_MAXD; ++i) z[i] = i < DD ? vprev[i < DD ? i : 0] : 0.0;
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < DD; ++i) z[i] = sin_[i];
  } else if (PLAIN) {
    matvec_add_plain<DD>(z, lanep1 + (int64_t)(lane - 1) * DD * DD * 2, sin_);
  } else {
    dd_matvec_add<DD>(z, lanep1 + (int64_t)(lane - 1) * DD * DD * 2, sin_, DD);
  }
  const double bad = poisoned ? __builtin_nan("") : 0.0;
  if (whole) {
#pragma unroll
    for (int i = 0; i < OP_LB; ++i) tile[lane][i] = (T)(iir_step_t<NSEC, ORD>(c, xr[i], z) + post_add + bad);
  } else {
#pragma unroll
    for (int i = 0; i < OP_LB; ++i)
      if (i < cnt) tile[lane][i] = (T)(iir_step_t<NSEC, ORD>(c, xr[i], z) + post_add + bad);
  }
  // final state of the row: the block that holds its last sample
  if (zf && chunk == nchunks - 1) {
    const int64_t lastblk = (left - 1) / OP_LB;
    if (lane == lastblk)
      for (int i = 0; i < DD; ++i) zf[(int64_t)row * DD + i] = z[i];
  }
  __syncthreads();
  if (whole) {
#pragma unroll
    for (int i = 0; i < OP_LB; ++i) {
      const int j = i * 64 + lane;
      y[j] = tile[j / OP_LB][j % OP_LB];
    }
  } else {
#pragma unroll
    for (int i = 0; i < OP_LB; ++i) {
      const int64_t j = (int64_t)i * 64 + lane;
      if (j < left) y[j] = tile[(int)(j / OP_LB)][(int)(j % OP_LB)];
    }
  }
  if (!persist) return;
  __syncthreads();
  }
}

// ---- the sampler inside the single pass: iir_sampled ----------------------------------------------------------
// Reference chain: Waveform.sample(filters=(sos, initial)) = sosfilt(sos, wav(t) - initial) + initial
// (waveforms/waveform.py:190-203, chunked :244-251) and predistort(wav(t), filters) (waveforms/distortion.py:298-321).
// Unfused, the samples make a round trip through HBM: 8 B written by the sampler, 8 B read and 8 B written by the
// filter.  Here the wave that owns a chunk EVALUATES its input with fir_sampled's arithmetic (wfk_chain_dev.h: one
// exact seed per fused op and lane, phasor table / Gaussian recurrence stepping by dt, no libm; the plan compiled for
// that geometry by wfk_compile_geom(1, 32)) straight into the registers the sweep reads: no load, no input transpose.
//
// What iir_onepass pays per 2048-sample chunk -- ticket, flag polls, aggregate loads: dependent memory round trips
// that two waves per SIMD cannot hide (its 256 x 1e7 launch waits three quarters of its time) -- is paid here per chunk
// of 64 RUN samples: a lane owns a RUN of consecutive samples -- OPS_RUN = 128 where pass 1 is a dot product (PLAIN),
// OPS_RUN_SWEEP = 256 where it is the sweep (slow poles; double-double scans) -- taken in rounds of 32 with the sampler's
// op state (phasor, Gaussian pair) and the filter state carried from round to round.  A sample costs ~10 VALU
// instructions to evaluate, so the chunk is not kept between the two sweeps (that would be 64 KB per wave), it is
// evaluated TWICE:
//   pass 1  RUN / 32 rounds: evaluate 32, sweep on from the running state (zero at the run's start); then the in-wave
//           scan of the 64 run states (TL = RUN-step transition), aggregate, look-back (as iir_onepass);
//   pass 2  the rounds again from the true state v_(l-1) + TL^l S_in: evaluate, sweep, transpose through LDS, store.
// All parameter blocks of the pieces a chunk overlaps are staged in LDS once per chunk (the host admits a plan only
// if they fit).  Algorithmic traffic: the 8 (4) B/sample of the filtered output.
#ifndef OPS_RUN
#define OPS_RUN 128
#endif
#ifndef OPS_RUN_SWEEP
#define OPS_RUN_SWEEP 256     // ... of plans whose scan is the sweep / double-double form (iir_sampled<..., PLAIN = false>)
#endif
#ifndef OPS_WAVES
#define OPS_WAVES 2
#endif
#define OPS_PAR 768             // doubles of LDS for the parameter blocks of the pieces a chunk overlaps
#define OPS_NCAR 2              // ops of a piece whose state is carried in registers from round to round

struct IirSampArgs {
  const DevChannel* channels;
  const DevPiece* pieces;
  const double* params;
  const int32_t* chunk_first;   // [rows * nchunks] first piece overlapping each chunk of 64 * RUN samples
  int64_t nchunks;
  double t0, step, last;
  int32_t has_last, pad;
  int64_t n, i0;
};

// The pieces a chunk overlaps (zero pieces included), as the wave staged them: piece i covers samples
// [start[i], start[i + 1]) of the row and has its parameter block at s_par + off[i] (off < 0: a zero piece).
#define OPS_PMAX 16
struct OpsPieces {
  int32_t start[OPS_PMAX + 1];      // relative to the chunk's first sample, clamped to +-2^30 (all index arithmetic of a round is 32-bit)
  int32_t off[OPS_PMAX];
};

// One round of a lane's run: the OP_LB consecutive samples js .. js + 31 of the row -> acc (channel offset included;
// zeros past the end of the row).  FAST form: every live piece of the chunk has the same op shapes (the host / the
// staging pass checks the packed op words) -- the usual pulse train.  Each LANE then reads the record and the phasor
// table of ITS OWN piece (per-lane LDS addresses; all lanes of a wave sit in one or two pieces), so a chunk that
// crosses piece edges costs what an interior chunk costs.  A lane whose 32 samples lie inside one piece runs the
// unmasked loop and carries its op state (phasor, Gaussian pair) to the next round as long as it stays in that piece;
// the one or two lanes per edge whose segment STRADDLES it take a per-sample-masked pass of their own (entered only when
// some lane of the wave straddles).  Live pieces are >= 32 samples long (host check): a segment meets at most two.
template <typename T>
__device__ __forceinline__ void ops_round(const IirSampArgs& a, const DevChannel& C, const double* s_par,
                                          const OpsPieces& pd, int m, int nops, int shape_off, bool deg1,
                                          int js, double dbase, int nrel, bool first_round, int& prev_p,
                                          ChSeeds (&car)[OPS_NCAR], T (&acc)[OP_LB]) {
  // js: the segment's first sample RELATIVE to the chunk (32-bit); dbase = (double)(chunk's first sample + grid.i0), exact, so
  // that dbase + js is the exact double of the full-grid index (one conversion + one add instead of the int64 sequence);
  // nrel: samples of the row from the chunk's start (clamped)
  constexpr int CL = OP_LB;
  const double2* const unit_tab = reinterpret_cast<const double2*>(s_par + OPS_PAR);
  double x;
  {
#pragma clang fp contract(off)
    const double mm = (dbase + (double)js) * a.step;
    x = mm + a.t0;
    if (a.has_last && js == nrel - 1) x = a.last;
  }
  if (C.tshift != 0.0) x = x - C.tshift;
  // the lane's piece at its first and at its last sample
  int p0 = 0, p1 = 0;
  for (int i = 1; i < m; ++i) {
    const int st = pd.start[i];
    p0 += st <= js ? 1 : 0;
    p1 += st <= js + (CL - 1) ? 1 : 0;
  }
  const int o0 = pd.off[p0], o1 = pd.off[p1];
  const bool inside = p1 == p0;
  const bool main_on = inside && o0 >= 0;                      // the whole segment in one live piece
  const bool part0 = !inside && o0 >= 0, part1 = !inside && o1 >= 0;
  const double* const blk0 = s_par + (o0 >= 0 ? o0 : shape_off);
  const double* const blk1 = s_par + (o1 >= 0 ? o1 : shape_off);
  const bool keep = !first_round && main_on && prev_p == p0;   // the carried state continues this lane's run in this piece
  prev_p = main_on ? p0 : -1;
  const bool any_main = __any(main_on), any_fresh = __any(main_on && !keep), any_part = __any(!inside);
  CH_EACH(CL, k) acc[k] = (T)0; CH_END
  for (int op = 0; op < nops; ++op) {
    const int fl = cuni(WFK_FCE_WORD(s_par + shape_off + WFK_BLK_HDR + op * WFK_FCE_REC));   // the same in every live piece
    const int env = (fl >> 4) & 3, carrier = (fl >> 2) & 1;
    if (any_main) {
      const double* rec = blk0 + WFK_BLK_HDR + op * WFK_FCE_REC;
      const bool fresh = !keep || op >= OPS_NCAR;
      ChSeeds sd, nx;
      sd.c = 1.0; sd.s = 0.0; sd.g = 1.0; sd.r = 1.0;
      if (op == 0) sd = car[0];
      if (OPS_NCAR > 1 && op == 1) sd = car[OPS_NCAR > 1 ? 1 : 0];
      if (any_fresh || op >= OPS_NCAR) {
        const ChSeeds ex = chain_make_seeds(rec, x, fl);       // exact
        if (fresh) sd = ex;
      }
      nx = sd;
      if (main_on) {
        if (env == 3) {
          chain_envmul<T, CL, CL>(rec, sd, acc, 0, CL, nx);
        } else {
          const double2* tab = carrier ? reinterpret_cast<const double2*>(blk0 + WFK_FCE_TABOFF(fl)) : unit_tab;
          const double qq = env ? rec[WFK_FCE_Q] : 1.0;
          const double u0 = x - rec[WFK_FCE_SLIN];
          if (deg1) chain_loop<T, CL, CL, false, true>(tab, rec, sd, u0, qq, acc, 0, CL, nx);
          else chain_loop<T, CL, CL, false, false>(tab, rec, sd, u0, qq, acc, 0, CL, nx);
        }
      }
      if (op == 0) car[0] = nx;
      if (OPS_NCAR > 1 && op == 1) car[OPS_NCAR > 1 ? 1 : 0] = nx;
    }
    if (any_part) {
      // segments that straddle a piece edge: samples [0, e0) belong to piece p0, [e1, CL) to piece p1
      const int d0 = pd.start[p0 + 1] - js, d1 = pd.start[p1] - js;
      const int e0 = d0 > CL ? CL : d0, e1 = d1 < 0 ? 0 : d1;
      ChSeeds nx;
      if (part0) {
        const double* rec = blk0 + WFK_BLK_HDR + op * WFK_FCE_REC;
        const ChSeeds sd = chain_make_seeds(rec, x, fl);
        if (env == 3) {
          chain_envmul<T, CL, -1, true>(rec, sd, acc, 0, e0, nx);
        } else {
          const double2* tab = carrier ? reinterpret_cast<const double2*>(blk0 + WFK_FCE_TABOFF(fl)) : unit_tab;
          chain_loop<T, CL, -1, true, false, true>(tab, rec, sd, x - rec[WFK_FCE_SLIN], env ? rec[WFK_FCE_Q] : 1.0, acc, 0, e0, nx);
        }
      }
      if (part1) {
        const double* rec = blk1 + WFK_BLK_HDR + op * WFK_FCE_REC;
        const ChSeeds sd = chain_make_seeds(rec, x, fl);
        if (env == 3) {
          chain_envmul<T, CL, -1, true>(rec, sd, acc, e1, CL, nx);
        } else {
          const double2* tab = carrier ? reinterpret_cast<const double2*>(blk1 + WFK_FCE_TABOFF(fl)) : unit_tab;
          chain_loop<T, CL, -1, true, false, true>(tab, rec, sd, x - rec[WFK_FCE_SLIN], env ? rec[WFK_FCE_Q] : 1.0, acc, e1, CL, nx);
        }
      }
    }
  }
  // (the channel offset is added by the caller, folded with the filter's `- initial`: one add per sample)
}

template <typename T, int NSEC, int ORD, bool PLAIN>
__global__ void __launch_bounds__(64, OPS_WAVES) iir_sampled(const IirCoef c, const IirSampArgs sa,
                                                  T* __restrict__ out, int64_t out_stride,
                                                  unsigned* __restrict__ status, double* __restrict__ aggbuf,
                                                  double* __restrict__ prefbuf, unsigned* __restrict__ ticket,
                                                  const double* __restrict__ pwL, const double* __restrict__ lanepL,
                                                  const double* __restrict__ lanepU, const double* __restrict__ wdot,
                                                  const double* __restrict__ zi, double* __restrict__ zf, int64_t n,
                                                  int64_t nchunks, int rows, unsigned epoch, double pre_sub,
                                                  double post_add, int persist, unsigned* __restrict__ fault,
                                                  int spin_limit) {
  constexpr int DD = NSEC * ORD;       // state dimension (<= 4)
  // samples per lane run: the dot-product form (PLAIN) is best at 128 (its weights are LDS reads per sample); the sweep /
  // double-double form amortises its scans and look-back over runs twice as long (four first-order sections, 256 x 1e7,
  // same box: 7.87 ms at 128, 7.01 at 256, 6.9 at 512; two biquads 7.06 / 7.20 / 7.67)
  constexpr int RUN = PLAIN ? OPS_RUN : OPS_RUN_SWEEP;
  __shared__ __attribute__((aligned(16))) double s_par[OPS_PAR + 2 * (OP_LB + 1) + 2];
  __shared__ T tile[64][OP_LB / 2 + 1];
  __shared__ OpsPieces pd;
  // PLAIN: the run's end state from zero state as a DOT PRODUCT, f = sum_k W[k] x_k with W[k] = TL-step response to a
  // unit sample at position k (host, quad precision): DD independent fmas per sample instead of the sweep's 5 per
  // section in one dependent chain.  (Only where the transition powers have entries of order 1: no cancellation.)
  __shared__ __attribute__((aligned(16))) double s_w[PLAIN ? RUN * 4 : 4];
  const int lane = threadIdx.x;
  if (PLAIN) {
    for (int i = lane; i < RUN * 4; i += 64) s_w[i] = wdot[i];
  }
  const int row = (int)(blockIdx.x % (unsigned)rows);
  const DevChannel C = sa.channels[row];
  const double xoff = C.offset - pre_sub;                      // sample + channel offset - initial: what the filter sees
  if (lane <= OP_LB) reinterpret_cast<double2*>(s_par + OPS_PAR)[lane] = make_double2(1.0, 0.0);   // phasor table of ops without a carrier
  for (;;) {
  unsigned t = 0;
  if (lane == 0) t = atomicAdd(ticket + 16 * row, 1u);
  t = (unsigned)__builtin_amdgcn_readfirstlane((int)t);
  const int64_t chunk = t;
  if (chunk >= nchunks) return;
  const int64_t base = chunk * (64 * RUN), chunk_end = base + (64 * RUN);
  const int jrun = lane * RUN;                             // first sample of this lane's run, relative to the chunk
  const int64_t left = n - base;
  const int nrel = (int)(left > (1 << 30) ? (1 << 30) : left); // samples of the row from the chunk's start (clamped)
  const double dbase = (double)(base + sa.i0);                 // exact (< 2^53)
  T* y = out + (int64_t)row * out_stride + base;

  // ---- stage the pieces the chunk overlaps: descriptors (zero pieces included) and the parameter blocks of the live
  // ones, back to back (the host has checked that they fit, that there are <= OPS_PMAX, and that all live pieces of
  // the plan have the same op shapes)
  int m = 0, shape_off = 0;
  bool have_live = false;
  {
    __syncthreads();                                           // the previous chunk is done with the blocks
    int off = 0;
    bool& have = have_live;
    for (int q = cuni(sa.chunk_first[(int64_t)row * sa.nchunks + chunk]); q < C.piece_end && m < OPS_PMAX; q = cuni(q + 1)) {
      const DevPiece P = sa.pieces[q];
      if (P.start >= chunk_end) break;
      if (lane == 0) {
        const int64_t rs = P.start - base;
        pd.start[m] = m == 0 ? -(1 << 30) : (int32_t)(rs > (1 << 30) ? (1 << 30) : rs);
        pd.start[m + 1] = 1 << 30;
        pd.off[m] = P.n_blk == 0 ? -1 : off;
      }
      ++m;
      if (P.n_blk == 0) continue;
      for (int i = lane; i < P.first_len; i += 64) s_par[off + i] = sa.params[P.par_off + i];
      if (!have) { shape_off = off; have = true; }
      off = cuni(off + ((P.first_len + 1) & ~1));
    }
    __syncthreads();
  }
  const int nops = have_live ? cuni((int)s_par[shape_off + 1]) : 0;
  bool deg1 = true;
  for (int op = 0; op < nops; ++op) {
    const int fl = cuni(WFK_FCE_WORD(s_par + shape_off + WFK_BLK_HDR + op * WFK_FCE_REC));
    deg1 = deg1 && (fl & 3) <= 1 && ((fl >> 4) & 3) != 3;
  }
  int prev_p = -1;

  // ---- pass 1: the run from zero state, round by round
  ChSeeds car[OPS_NCAR];
#pragma unroll
  for (int i = 0; i < OPS_NCAR; ++i) { car[i].c = 1.0; car[i].s = 0.0; car[i].g = 1.0; car[i].r = 1.0; }
  double z[IIR_MAXD];
#pragma unroll
  for (int i = 0; i < IIR_MAXD; ++i) z[i] = 0.0;
#pragma unroll 1
  for (int r = 0; r < (RUN / OP_LB); ++r) {
    const int js = jrun + r * OP_LB;
    if (r * OP_LB >= nrel) break;                          // (wave-uniform: no lane has a sample in this round)
    T acc[OP_LB];
    ops_round<T>(sa, C, s_par, pd, m, nops, shape_off, deg1, js, dbase, nrel, r == 0, prev_p, car, acc);
    const int rest = nrel - js;
    if (PLAIN && chunk_end <= n) {                             // (wave-uniform: every run of the chunk is whole)
      const double* w = s_w + r * OP_LB * 4;
#pragma unroll
      for (int i = 0; i < OP_LB; ++i) {
        const double xv = (double)acc[i] + xoff;
#pragma unroll
        for (int j = 0; j < DD; ++j) z[j] = fma(w[i * 4 + j], xv, z[j]);
      }
    } else if (rest >= OP_LB) {
#pragma unroll
      for (int i = 0; i < OP_LB; ++i) (void)iir_step_t<NSEC, ORD>(c, (double)acc[i] + xoff, z);
    } else {
#pragma unroll
      for (int i = 0; i < OP_LB; ++i)
        if (i < rest) (void)iir_step_t<NSEC, ORD>(c, (double)acc[i] + xoff, z);
    }
  }
  // (a run past the end of the row leaves its state alone; the scan still multiplies by TL per run, which only matters
  //  AFTER the last sample -- nothing there is used)
  if (PLAIN) wave_scan_plain<DD>(z, pwL, lane);
  else wave_scan<DD>(z, pwL, DD, lane);                        // z = v_l (inclusive)
  double vprev[DD], agg[DD];
#pragma unroll
  for (int i = 0; i < DD; ++i) {
    const double up = __shfl_up(z[i], 1);
    vprev[i] = lane == 0 ? 0.0 : up;
    agg[i] = __shfl(z[i], 63);
  }
  const int64_t slot = ((int64_t)row * nchunks + chunk);
  const unsigned F_AGG = epoch * 4u + 1u, F_PRE = epoch * 4u + 2u;
  if (chunk > 0 && lane == 0) {
#pragma unroll
    for (int i = 0; i < DD; ++i) op_store(aggbuf + slot * DD + i, agg[i]);
    __builtin_amdgcn_s_waitcnt(0);                             // the state is in memory before the flag is
    __hip_atomic_store(status + slot, F_AGG, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }

  // ---- state at the chunk's start: decoupled look-back, window by window (see iir_onepass)
  double sin_[IIR_MAXD];
#pragma unroll
  for (int i = 0; i < IIR_MAXD; ++i) sin_[i] = 0.0;
  bool poisoned = false;
  if (chunk == 0) {
#pragma unroll
    for (int i = 0; i < DD; ++i) sin_[i] = zi ? zi[(int64_t)row * DD + i] : 0.0;
  } else {
    double hop[IIR_MAXD];
#pragma unroll
    for (int i = 0; i < IIR_MAXD; ++i) hop[i] = 0.0;
    int spins = 0;
    for (int w = 0;; ++w) {
      const int64_t pc = chunk - 1 - 64 * (int64_t)w - lane;
      const unsigned* f = status + (int64_t)row * nchunks + (pc >= 0 ? pc : 0);
      unsigned st = 0;
      int kstop = -1;
      bool whole_window = false;
      for (; spins < spin_limit; ++spins) {
        if (pc >= 0 && !(st == F_PRE)) st = __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool pre = pc >= 0 && st == F_PRE;
        const bool ag = pc >= 0 && (st == F_AGG || st == F_PRE);
        const unsigned long long has_pre = __ballot(pre), has_agg = __ballot(ag);
        if (has_pre != 0ull) {
          const int k = __ffsll((long long)has_pre) - 1;       // nearest chunk with a prefix
          const unsigned long long need = k == 0 ? 0ull : (~0ull >> (64 - k));   // lanes 0 .. k-1
          if ((has_agg & need) == need) { kstop = k; break; }
        } else if (has_agg == ~0ull && w < OP_WINDOWS - 1) {
          whole_window = true;
          kstop = 64;
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
      if (kstop < 0) { poisoned = true; kstop = 0; }
      double contrib[IIR_MAXD];
#pragma unroll
      for (int i = 0; i < IIR_MAXD; ++i) contrib[i] = 0.0;
      if (lane <= kstop && pc >= 0) {
        const double* src = (lane == kstop ? prefbuf : aggbuf) + ((int64_t)row * nchunks + pc) * DD;
        double v[IIR_MAXD];
#pragma unroll
        for (int i = 0; i < IIR_MAXD; ++i) v[i] = i < DD ? op_load(src + i) : 0.0;
        if (lane == 0) {
#pragma unroll
          for (int i = 0; i < DD; ++i) contrib[i] = v[i];     // U^0
        } else {
          dd_matvec_add<DD>(contrib, lanepU + (int64_t)(lane - 1) * DD * DD * 2, v, DD);   // U^lane
        }
      }
      double wsum[IIR_MAXD];
#pragma unroll
      for (int i = 0; i < IIR_MAXD; ++i) wsum[i] = 0.0;
#pragma unroll
      for (int i = 0; i < DD; ++i) {
        double sum = contrib[i];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) sum += __shfl_xor(sum, o);
        wsum[i] = sum;
      }
      for (int q = 0; q < w; ++q) {                            // this window lies 64 w chunks back: U^(64 w)
        double nx[IIR_MAXD];
#pragma unroll
        for (int i = 0; i < IIR_MAXD; ++i) nx[i] = 0.0;
        dd_matvec_add<DD>(nx, lanepU + (int64_t)63 * DD * DD * 2, wsum, DD);
#pragma unroll
        for (int i = 0; i < IIR_MAXD; ++i) wsum[i] = nx[i];
      }
#pragma unroll
      for (int i = 0; i < DD; ++i) hop[i] += wsum[i];
      if (!whole_window) break;
    }
#pragma unroll
    for (int i = 0; i < DD; ++i) sin_[i] = hop[i];
  }
  poisoned = __any(poisoned);
  if (poisoned && lane == 0) __hip_atomic_fetch_or(fault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);

  // ---- publish the inclusive prefix: agg + U * S_in
  if (lane == 0) {
    double so[IIR_MAXD];
#pragma unroll
    for (int i = 0; i < IIR_MAXD; ++i) so[i] = i < DD ? agg[i < DD ? i : 0] : 0.0;
    dd_matvec_add<DD>(so, lanepU, sin_, DD);
    if (chunk + 1 < nchunks) {
#pragma unroll
      for (int i = 0; i < DD; ++i) op_store(prefbuf + slot * DD + i, so[i]);
      __builtin_amdgcn_s_waitcnt(0);
      __hip_atomic_store(status + slot, F_PRE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }

  // ---- pass 2: the run again from its true state v_(l-1) + TL^l S_in: evaluate, sweep, store
#pragma unroll
  for (int i = 0; i < IIR_MAXD; ++i) z[i] = i < DD ? vprev[i < DD ? i : 0] : 0.0;
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < DD; ++i) z[i] = sin_[i];
  } else if (PLAIN) {
    matvec_add_plain<DD>(z, lanepL + (int64_t)(lane - 1) * DD * DD * 2, sin_);
  } else {
    dd_matvec_add<DD>(z, lanepL + (int64_t)(lane - 1) * DD * DD * 2, sin_, DD);
  }
  const double bad = poisoned ? __builtin_nan("") : 0.0;
#pragma unroll 1
  for (int r = 0; r < (RUN / OP_LB); ++r) {
    const int js = jrun + r * OP_LB;
    if (r * OP_LB >= nrel) break;                          // (wave-uniform)
    T acc[OP_LB];
    ops_round<T>(sa, C, s_par, pd, m, nops, shape_off, deg1, js, dbase, nrel, r == 0, prev_p, car, acc);
    const int rest = nrel - js;                                // samples of the row from this lane's segment on
    if (rest >= OP_LB) {
#pragma unroll
      for (int i = 0; i < OP_LB; ++i) acc[i] = (T)(iir_step_t<NSEC, ORD>(c, (double)acc[i] + xoff, z) + post_add + bad);
    } else {
#pragma unroll
      for (int i = 0; i < OP_LB; ++i)
        if (i < rest) acc[i] = (T)(iir_step_t<NSEC, ORD>(c, (double)acc[i] + xoff, z) + post_add + bad);
    }
    // final state of the row: the segment that holds its last sample
    if (zf && rest > 0 && rest <= OP_LB)
      for (int i = 0; i < DD; ++i) zf[(int64_t)row * DD + i] = z[i];
    // Store, in two halves of 16 samples per lane through the LDS tile: tile row l then holds samples
    // l * RUN + 32 r + 16 h + [0, 16) of the chunk -- one 128-byte line; a wave instruction stores four rows.
    // (The store phase's view of the lane index is opaque: left to itself the compiler computes the LDS addresses and
    // global offsets of this phase once, above the chunk loop, and spills them.)
    int lt = lane;
    asm volatile("" : "+v"(lt));
    constexpr int HS = OP_LB / 2, RPI = 64 / HS;               // samples per half; tile rows per wave instruction
    const int col = lt % HS, rsub = lt / HS;                   // rows RPI i + rsub, column col
    const T* tp = &tile[0][0] + rsub * (HS + 1) + col;
    CH_EACH(2, h)
      __syncthreads();                                         // the previous half's stores have read the tile
      CH_EACH(HS, i) tile[lane][i] = acc[h * HS + i]; CH_END
      __syncthreads();
      T* ys = y + (int64_t)rsub * RUN + r * OP_LB + h * HS + col;
      if (chunk_end <= n) {
#pragma unroll
        for (int i = 0; i < 64 / RPI; ++i) ys[(int64_t)i * RPI * RUN] = tp[i * RPI * (HS + 1)];
      } else {
#pragma unroll
        for (int i = 0; i < 64 / RPI; ++i)
          if ((RPI * i + rsub) * RUN + r * OP_LB + h * HS + col < nrel) ys[(int64_t)i * RPI * RUN] = tp[i * RPI * (HS + 1)];
      }
    CH_END
  }
  if (!persist) return;
  }
}

// state dimension 0 (every section is a bare gain): y = g (x - pre) + post
template <typename T>
__global__ void __launch_bounds__(256) iir_scale(const T* __restrict__ in, int64_t in_stride,
                                                 T* __restrict__ out, int64_t out_stride, int64_t n,
                                                 double g, double pre, double post) {
  const int64_t row = blockIdx.y;
  for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < n; j += (int64_t)gridDim.x * 256)
    out[row * out_stride + j] = (T)(g * ((double)in[row * in_stride + j] - pre) + post);
}

int iir_fail(int code, const std::string& m) {
  wfk_internal_set_error(m.c_str());
  return code;
}

// Host-side transition matrices in quad precision.  In direct-form coordinates T is wildly
// non-normal for clustered poles (entries ~1e5 while every eigenvalue is < 1): a T computed
// by a double-precision recurrence is off by ~1e-9 relative, which the scan would amplify
// into 1e-6 output errors.  Computed in __float128 and rounded once, T*S is as accurate as
// the sequential filter itself.
typedef __float128 quad;

void quad_step(const IirCoef& c, quad x, quad* z) {
  for (int s = 0; s < c.nsec; ++s) {
    quad* zs = z + c.off[s];
    const int ord = c.ord[s];
    const quad y = (quad)c.b[s][0] * x + (ord > 0 ? zs[0] : (quad)0);
    for (int i = 0; i + 1 < ord; ++i)
      zs[i] = (quad)c.b[s][i + 1] * x - (quad)c.a[s][i + 1] * y + zs[i + 1];
    if (ord > 0) zs[ord - 1] = (quad)c.b[s][ord] * x - (quad)c.a[s][ord] * y;
    x = y;
  }
}

void qmatmul(const std::vector<quad>& A, const std::vector<quad>& B, std::vector<quad>& C, int D) {
  C.assign((size_t)D * D, (quad)0);
  for (int i = 0; i < D; ++i)
    for (int k = 0; k < D; ++k)
      for (int j = 0; j < D; ++j) C[(size_t)i * D + j] += A[(size_t)i * D + k] * B[(size_t)k * D + j];
}

}  // namespace

struct wfk_iir_plan {
  IirCoef c;
  int64_t n = 0, nblk = 0;
  int32_t batch = 0, kind = 0;
  int64_t ngrp = 0;          // groups of 64 blocks per row
  double* state = nullptr;   // [batch][nblk][D]  group-local prefixes
  double* grp = nullptr;     // [batch][ngrp][D]  group totals -> group start states
  double* pw = nullptr;      // [7][D][D][2]   T^(2^k)
  double* lanep = nullptr;   // [64][D][D][2]  T^(l+1)
  double* pw2 = nullptr;     // the same for U = T^64
  double* lanep2 = nullptr;
  // A cascade of more than 4 biquads has no register-resident kernel (its 10+ state values
  // would go to scratch: 34 ms instead of 3.3 on 64 x 1e7).  It runs as consecutive passes of
  // <= 4 biquads each, in place on `out`, every pass with its own slice of zi / zf.
  std::vector<wfk_iir_plan*> parts;
  std::vector<int> part_off;     // state offset of each part
  double* zi_tmp = nullptr;      // [batch][D_part] repacked state slices
  double* zf_tmp = nullptr;
  double gain = 1.0;             // D == 0: y = gain * (x - pre) + post
  // single-pass form (biquad cascades with <= 4 states): chunk flags / aggregates / prefixes, ticket,
  // and the tables of the 32-step block transition T1 and of U = T1^64
  bool onepass = false;
  bool op_plain = false;         // in-wave scan in plain double (entries of T1^1..T1^64 of order 1)
  int64_t op_chunks = 0;
  unsigned epoch = 0;
  unsigned* op_status = nullptr;
  unsigned* op_ticket = nullptr;
  double* op_agg = nullptr;
  double* op_pref = nullptr;
  double* op_pw1 = nullptr;
  double* op_lanep1 = nullptr;
  double* op_lanepU = nullptr;
  // iir_sampled: a lane owns a run of op_run samples: TL = T1^(op_run / OP_LB), UL = TL^64
  double* op_pwL = nullptr;          // TL^(2^k)
  double* op_lanepL = nullptr;       // TL^(l+1)
  double* op_lanepUL = nullptr;      // UL^(l+1)
  double* op_wdot = nullptr;         // [op_run][4]: end state of a run from zero state per unit sample at position k (dot-product form)
  int op_run = 0;                    // samples per lane run of iir_sampled: OPS_RUN (dot-product form) | OPS_RUN_SWEEP
  bool op_plainL = false;
  unsigned* op_fault = nullptr;      // host memory, mapped: raised by a chunk whose look-back timed out
  unsigned* op_fault_dev = nullptr;  // ... its device address
};

extern "C" {

int wfk_iir_plan_destroy(wfk_iir_plan* p) {
  if (!p) return WFK_OK;
  for (wfk_iir_plan* q : p->parts) wfk_iir_plan_destroy(q);
  (void)hipFree(p->zi_tmp);
  (void)hipFree(p->zf_tmp);
  (void)hipFree(p->op_status);
  (void)hipFree(p->op_ticket);
  (void)hipFree(p->op_agg);
  (void)hipFree(p->op_pref);
  (void)hipFree(p->op_pw1);
  (void)hipFree(p->op_lanep1);
  (void)hipFree(p->op_lanepU);
  (void)hipFree(p->op_pwL);
  (void)hipFree(p->op_lanepL);
  (void)hipFree(p->op_lanepUL);
  (void)hipFree(p->op_wdot);
  if (p->op_fault) (void)hipHostFree(p->op_fault);
  (void)hipFree(p->state);
  (void)hipFree(p->grp);
  (void)hipFree(p->pw);
  (void)hipFree(p->lanep);
  (void)hipFree(p->pw2);
  (void)hipFree(p->lanep2);
  delete p;
  return WFK_OK;
}

int wfk_iir_plan_create(int32_t n_sections, const int32_t* orders, const double* b, const double* a,
                        int64_t n, int32_t batch, int kind, wfk_iir_plan** out) {
  if (!out) return iir_fail(WFK_EINVAL, "null out");
  *out = nullptr;
  if (n_sections < 1 || n_sections > 4096 || !orders || !b || !a || n < 0 || batch < 1 ||
      batch > 65535)
    return iir_fail(WFK_EINVAL, "bad IIR arguments");
  if (kind != WFK_OUT_F64 && kind != WFK_OUT_F32) return iir_fail(WFK_EINVAL, "IIR kind must be F64 or F32");
  wfk_iir_plan* p = new wfk_iir_plan();
  IirCoef& c = p->c;
  std::memset(&c, 0, sizeof c);
  {
    // A cascade is a composition of sections, so any cut of the section list into consecutive
    // runs is the same filter (scipy's sosfilt/lfilter state layout is section by section, so
    // zi/zf slice the same way).  One kernel pass takes <= IIR_MAXSEC sections with a total
    // state dimension <= IIR_MAXD; longer cascades run as consecutive passes.  Only a SINGLE
    // section of order > IIR_MAXD has no device form.
    int Dtot = 0;
    bool single_too_big = false;
    for (int s = 0; s < n_sections; ++s) {
      if (orders[s] < 0) { delete p; return iir_fail(WFK_EINVAL, "negative section order"); }
      single_too_big = single_too_big || orders[s] > IIR_MAXD;
      Dtot += orders[s];
    }
    if (single_too_big) {
      delete p;
      return iir_fail(WFK_EUNSUP, "a single IIR section of order > 16 (factor it into a cascade)");
    }
    // Register-resident kernels exist for runs of EQUAL order: up to four first-order sections or
    // four biquads, single sections of order 3..8; anything else (mixed orders in one pass, two
    // sections of order 3, ...) would take the runtime-shaped kernel with its state in scratch
    // (34-68 ms per pass on 64 x 1e7 against ~3).  So the cascade is cut into such runs.
    // three or four biquads: one three-launch pass (64 x 1e7: 3.1-3.2 ms; as two single passes 5.7-6.3),
    // except on small problems, where the three launches cost 0.4-0.6 ms whatever the size (64 x 1e5,
    // four biquads: 0.62 ms against 0.12 ms as two single passes)
    int biq_cap = (double)batch * (double)n <= 3e7 ? 2 : 4;
    if (const char* e = getenv("WFK_IIR_BIQ_CAP")) { const int v = atoi(e); if (v >= 1 && v <= 4) biq_cap = v; }
    auto run_cap = [biq_cap](int order) { return order == 1 ? 4 : (order == 2 ? biq_cap : 1); };
    bool one_run = n_sections <= 1;
    if (n_sections > 1) {
      one_run = n_sections <= run_cap(orders[0]);
      for (int s = 1; s < n_sections; ++s) one_run = one_run && orders[s] == orders[0];
    }
    const bool split = n_sections > IIR_MAXSEC || Dtot > IIR_MAXD || !one_run;
    if (split) {
      int ndev = 0;
      if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        (void)hipGetLastError();
        delete p;
        return iir_fail(WFK_EHIP, "no HIP device visible");
      }
      c.nsec = 0; c.D = Dtot;
      p->n = n; p->batch = batch; p->kind = kind;
      if (n == 0) { *out = p; return WFK_OK; }
      int s0 = 0, pos0 = 0, doff = 0;
      while (s0 < n_sections) {
        // greedy run of equal orders, as long as a register-resident kernel takes it
        int cnt = 0, dd = 0, pp = 0;
        while (s0 + cnt < n_sections) {
          const int o = orders[s0 + cnt];
          if (cnt > 0 && (o != orders[s0] || cnt >= run_cap(o))) break;
          if (dd + o > IIR_MAXD) break;
          dd += o; pp += o + 1; ++cnt;
        }
        wfk_iir_plan* q = nullptr;
        int rc = wfk_iir_plan_create(cnt, orders + s0, b + pos0, a + pos0, n, batch, kind, &q);
        if (rc) { wfk_iir_plan_destroy(p); return rc; }
        p->parts.push_back(q);
        p->part_off.push_back(doff);
        s0 += cnt; pos0 += pp; doff += dd;
      }
      if (hipMalloc(&p->zi_tmp, (size_t)batch * IIR_MAXD * 8) != hipSuccess ||
          hipMalloc(&p->zf_tmp, (size_t)batch * IIR_MAXD * 8) != hipSuccess) {
        wfk_iir_plan_destroy(p);
        return iir_fail(WFK_ENOMEM, "IIR buffer allocation failed");
      }
      *out = p;
      return WFK_OK;
    }
  }
  c.nsec = n_sections;
  int D = 0, pos = 0;
  for (int s = 0; s < n_sections; ++s) {
    const int ord = orders[s];
    if (ord < 0 || ord > IIR_MAXD || D + ord > IIR_MAXD) {
      delete p;
      return iir_fail(WFK_EINVAL, "IIR order too large (total state dimension <= 16)");
    }
    const double a0 = a[pos];
    if (!(a0 != 0.0) || !std::isfinite(a0)) {
      delete p;
      return iir_fail(WFK_EINVAL, "a[0] must be finite and non-zero");
    }
    c.ord[s] = ord;
    c.off[s] = D;
    for (int i = 0; i <= ord; ++i) {   // scipy normalises by a[0]
      c.b[s][i] = b[pos + i] / a0;
      c.a[s][i] = a[pos + i] / a0;
    }
    pos += ord + 1;
    D += ord;
  }
  c.D = D;
  p->n = n; p->batch = batch; p->kind = kind;
  p->nblk = n > 0 ? (n + IIR_LB - 1) / IIR_LB : 0;
  p->ngrp = (p->nblk + 63) / 64;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
    (void)hipGetLastError();
    delete p;
    return iir_fail(WFK_EHIP, "no HIP device visible");
  }
  if (n == 0 || D == 0) {
    // order-0 sections only: a pure gain (lfilter with len(a) == len(b) == 1 per section)
    p->gain = 1.0;
    for (int s = 0; s < n_sections; ++s) p->gain *= c.b[s][0];
    *out = p;
    return WFK_OK;
  }
  // LB-step transition matrix T: column i = homogeneous response to unit state e_i
  std::vector<quad> T((size_t)D * D);
  for (int i = 0; i < D; ++i) {
    quad z[IIR_MAXD];
    for (int r = 0; r < IIR_MAXD; ++r) z[r] = 0;
    z[i] = 1;
    for (int k = 0; k < IIR_LB; ++k) quad_step(c, (quad)0, z);
    for (int r = 0; r < D; ++r) T[(size_t)r * D + i] = z[r];
  }
  auto put = [](std::vector<double>& dst, size_t at, quad q) {   // quad -> (hi, lo)
    const double hi = (double)q;
    dst[at] = hi;
    dst[at + 1] = (double)(q - (quad)hi);
  };
  // tables of a base matrix B: pw[k] = B^(2^k) (k < 7), lanep[l] = B^(l+1) (l < 64); returns B^64
  auto tables = [&](const std::vector<quad>& B, std::vector<double>& pw, std::vector<double>& lanep) {
    pw.assign((size_t)7 * D * D * 2, 0.0);
    lanep.assign((size_t)64 * D * D * 2, 0.0);
    std::vector<quad> cur = B, nxt;
    for (int k = 0; k < 7; ++k) {
      for (size_t e = 0; e < cur.size(); ++e) put(pw, ((size_t)k * D * D + e) * 2, cur[e]);
      qmatmul(cur, cur, nxt, D);
      cur = nxt;
    }
    cur = B;
    std::vector<quad> last;
    for (int l = 0; l < 64; ++l) {
      for (size_t e = 0; e < cur.size(); ++e) put(lanep, ((size_t)l * D * D + e) * 2, cur[e]);
      last = cur;
      qmatmul(cur, B, nxt, D);
      cur = nxt;
    }
    return last;
  };
  std::vector<double> pw, lanep, pw2, lanep2;
  const std::vector<quad> U = tables(T, pw, lanep);   // U = T^64: one group
  tables(U, pw2, lanep2);
  auto upload = [](double** dst, const std::vector<double>& src) {
    return hipMalloc(dst, src.size() * 8) == hipSuccess &&
           hipMemcpy(*dst, src.data(), src.size() * 8, hipMemcpyHostToDevice) == hipSuccess;
  };
  bool ok = hipMalloc(&p->state, (size_t)batch * p->nblk * D * 8) == hipSuccess;
  ok = ok && hipMalloc(&p->grp, (size_t)batch * p->ngrp * D * 8) == hipSuccess;
  ok = ok && upload(&p->pw, pw) && upload(&p->lanep, lanep) && upload(&p->pw2, pw2) && upload(&p->lanep2, lanep2);
  if (!ok) {
    wfk_iir_plan_destroy(p);
    return iir_fail(WFK_ENOMEM, "IIR buffer allocation failed");
  }
  // single-pass form: one or two biquads (state dimension <= 4), rows long enough to chain
  {
    // shapes: one or two biquads, or up to four FIRST-order sections -- the cascade of exponential
    // corrections a flux-line predistortion is usually made of (state dimension <= 4)
    // (and single sections of order 3 or 4: a combined lfilter of three or four first-order filters)
    bool biq = n_sections >= 1 && orders[0] >= 1 && n_sections * orders[0] <= 4;
    for (int s2 = 0; s2 < n_sections; ++s2) biq = biq && orders[s2] == orders[0];
    // The default for these shapes since its loads are issued in one batch and the look-back no
    // longer chains on the nearest prefix (tools/iir_sweep.py, fp64, 1 / 2 biquads, three-launch vs
    // single pass): 1024 x 1e6 4.65 / 4.71 vs 3.08 / 3.44 ms (66 / 60 % of the HBM peak on 16 B/sample),
    // 256 x 1e7 11.7 / 11.5 vs 8.6 / 9.9, 64 x 1e7 3.0 / 3.2 vs 2.8 / 3.1, 64 x 1e5 0.39 / 0.46 vs
    // 0.04 / 0.05 (one launch instead of three).  WFK_IIR_ONEPASS=0 keeps the three-launch form.
    const char* on = getenv("WFK_IIR_ONEPASS");
    // Between 8 and 64 LONG rows the three-launch form is still ahead (round-2 depth probe, 2 biquads,
    // fp64, three-launch vs single pass: 8 x 1e7 0.61 vs 0.75 ms, 16 x 1e7 0.80 vs 1.22, 32 x 1e7 1.55 vs
    // 1.84, 48 x 1e7 2.29 vs 2.48; but 16 x 1e6 0.49 vs 0.12, 4 x 1e7 0.54 vs 0.38, 1 x 1e7 0.51 vs 0.15):
    // 2304 / rows chunks of a row are in flight there and every look-back reads all of them.
    // WFK_IIR_ONEPASS=1 / 0 force one form or the other.
    const bool band = batch >= 8 && batch < 64 && (double)batch * (double)n >= 6e7;
    const bool want = on ? on[0] != '0' : !band;
    if (biq && n >= 4 * OP_CHUNK && want) {
      std::vector<quad> T1((size_t)D * D);
      for (int i = 0; i < D; ++i) {
        quad z[IIR_MAXD];
        for (int r = 0; r < IIR_MAXD; ++r) z[r] = 0;
        z[i] = 1;
        for (int k = 0; k < OP_LB; ++k) quad_step(c, (quad)0, z);
        for (int r = 0; r < D; ++r) T1[(size_t)r * D + i] = z[r];
      }
      std::vector<double> pw1, lanep1, pwU, lanepU;
      const std::vector<quad> U1 = tables(T1, pw1, lanep1);    // U1 = T1^64: one chunk
      tables(U1, pwU, lanepU);                                  // U1^(l+1), l < 64: the look-back window
      std::vector<double> pwL, lanepL, pwUL, lanepUL;           // the same for the long runs of iir_sampled
      // (runs of OPS_RUN samples; where their transition powers rule out the dot-product form, of OPS_RUN_SWEEP)
      p->op_run = OPS_RUN;
      for (int attempt = 0; attempt < 2; ++attempt) {
        pwL.clear(); lanepL.clear(); pwUL.clear(); lanepUL.clear();
        std::vector<quad> TL = T1, nxt;
        for (int k = 1; k < p->op_run / OP_LB; ++k) { qmatmul(TL, T1, nxt, D); TL = nxt; }
        const std::vector<quad> UL = tables(TL, pwL, lanepL);
        tables(UL, pwUL, lanepUL);
        double tm = 0.0;
        for (size_t e = 0; e < lanepL.size(); e += 2) tm = std::max(tm, std::fabs(lanepL[e]));
        const char* dde = getenv("WFK_IIR_DD");
        p->op_plainL = attempt == 0 && tm < 16.0 && !(dde && dde[0] == '1');
        if (p->op_plainL || p->op_run == OPS_RUN_SWEEP) break;
        p->op_run = OPS_RUN_SWEEP;
      }
      std::vector<double> wdot((size_t)p->op_run * 4, 0.0);
      for (int k = 0; k < p->op_run && p->op_plainL; ++k) {
        quad z[IIR_MAXD];
        for (int r = 0; r < IIR_MAXD; ++r) z[r] = 0;
        for (int t = k; t < p->op_run; ++t) quad_step(c, t == k ? (quad)1 : (quad)0, z);
        for (int r = 0; r < D && r < 4; ++r) wdot[(size_t)k * 4 + r] = (double)z[r];
      }
      double tmax = 0.0;                                        // largest entry of T1^1 .. T1^64 (hi words)
      for (size_t e = 0; e < lanep1.size(); e += 2) tmax = std::max(tmax, std::fabs(lanep1[e]));
      const char* ddenv = getenv("WFK_IIR_DD");
      p->op_plain = tmax < 16.0 && !(ddenv && ddenv[0] == '1');
      p->op_chunks = (n + OP_CHUNK - 1) / OP_CHUNK;
      const size_t slots = (size_t)batch * (size_t)p->op_chunks;
      bool ok1 = hipMalloc(&p->op_status, slots * 4) == hipSuccess &&
                 hipMemset(p->op_status, 0, slots * 4) == hipSuccess &&
                 hipMalloc(&p->op_ticket, (size_t)batch * 64) == hipSuccess &&
                 hipMalloc(&p->op_agg, slots * D * 8) == hipSuccess &&
                 hipMalloc(&p->op_pref, slots * D * 8) == hipSuccess && upload(&p->op_pw1, pw1) &&
                 upload(&p->op_lanep1, lanep1) && upload(&p->op_lanepU, lanepU) && upload(&p->op_pwL, pwL) &&
                 upload(&p->op_lanepL, lanepL) && upload(&p->op_lanepUL, lanepUL) && upload(&p->op_wdot, wdot);
      ok1 = ok1 && hipHostMalloc((void**)&p->op_fault, 64, hipHostMallocMapped) == hipSuccess &&
            hipHostGetDevicePointer((void**)&p->op_fault_dev, p->op_fault, 0) == hipSuccess;
      if (!ok1) {
        wfk_iir_plan_destroy(p);
        return iir_fail(WFK_ENOMEM, "IIR single-pass buffer allocation failed");
      }
      *p->op_fault = 0;
      // ONE section of order 3 / 4 whose transition powers grow past 1e3 (clustered poles: butter(4, 0.022) as a single
      // (b, a), 3.2e3): the block start states this form re-injects every 32 samples as doubles cost it a digit against
      // the three-launch form (iirchain_soak seed 12713, against a long-double recursion: 1.1e-10 vs 2.4e-11 of peak;
      // SciPy 1.7e-11) -- such sections take that form unless WFK_IIR_ONEPASS=1 insists
      p->onepass = (on && on[0] == '1') || !(orders[0] >= 3 && tmax > 1e3);
    }
  }
  *out = p;
  return WFK_OK;
}

int wfk_iir_state_dim(const wfk_iir_plan* p) { return p ? p->c.D : WFK_EINVAL; }

}  // extern "C"

template <typename T, int NSEC, int ORD>
static void iir_launch_t(wfk_iir_plan* p, const void* in, int64_t is, void* out, int64_t os,
                         const double* zi, double* zf, double initial, double post, hipStream_t s) {
  const dim3 g((unsigned)p->ngrp, (unsigned)p->batch);
  hipLaunchKernelGGL((iir_pass<T, false, NSEC, ORD>), g, dim3(64), 0, s, p->c, (const T*)in, is,
                     (T*)nullptr, (int64_t)0, p->state, p->grp, p->pw, p->lanep, (double*)nullptr,
                     p->n, p->nblk, initial, 0.0);
  hipLaunchKernelGGL(iir_scan, dim3((unsigned)p->batch), dim3(64), 0, s, p->grp, p->pw2, p->lanep2,
                     zi, p->ngrp, p->c.D);
  hipLaunchKernelGGL((iir_pass<T, true, NSEC, ORD>), g, dim3(64), 0, s, p->c, (const T*)in, is,
                     (T*)out, os, p->state, p->grp, p->pw, p->lanep, zf, p->n, p->nblk, initial,
                     post);
}

template <typename T>
static void iir_launch(wfk_iir_plan* p, const void* in, int64_t is, void* out, int64_t os,
                       const double* zi, double* zf, double initial, double post, hipStream_t s) {
  const IirCoef& c = p->c;
  bool uniform = true;
  for (int k = 1; k < c.nsec; ++k) uniform = uniform && c.ord[k] == c.ord[0];
  const int ns = c.nsec, od = c.ord[0];
#define IIR_CASE(NS, OD) if (uniform && ns == NS && od == OD) return iir_launch_t<T, NS, OD>(p, in, is, out, os, zi, zf, initial, post, s)
  IIR_CASE(1, 2); IIR_CASE(2, 2); IIR_CASE(3, 2); IIR_CASE(4, 2);      // sosfilt cascades
  IIR_CASE(1, 1); IIR_CASE(1, 3); IIR_CASE(1, 4); IIR_CASE(1, 5); IIR_CASE(1, 6);  // lfilter
  IIR_CASE(1, 7); IIR_CASE(1, 8);   // predistort(filters=) combines its sections into ONE polynomial
#undef IIR_CASE
  iir_launch_t<T, 0, 0>(p, in, is, out, os, zi, zf, initial, post, s);
}

extern "C" {

// y = F(x - pre) + post   (the public entry has pre == post == `initial`)
// `src` != null: the first single-pass stage EVALUATES its input from the waveform program (iir_sampled) instead of
// reading `in_dev`; the caller has checked that this stage is in the single-pass form.
static int iir_apply_impl(wfk_iir_plan* p, const void* in_dev, int64_t in_stride, void* out_dev,
                          int64_t out_stride, const double* zi_dev, double* zf_dev, double initial,
                          double post, void* hip_stream, const IirSampArgs* src = nullptr) {
  if (!p) return iir_fail(WFK_EINVAL, "null plan");
  if (p->n == 0) return WFK_OK;
  if (src) { in_dev = out_dev; in_stride = out_stride; }     // (never read)
  if (!in_dev || !out_dev) return iir_fail(WFK_EINVAL, "null buffer");
  if (in_stride < p->n || out_stride < p->n) return iir_fail(WFK_EINVAL, "stride smaller than n");
  hipStream_t s = (hipStream_t)hip_stream;
  if (p->c.D == 0) {
    const dim3 g((unsigned)std::min<int64_t>((p->n + 255) / 256, 4096), (unsigned)p->batch);
    if (p->kind == WFK_OUT_F32)
      hipLaunchKernelGGL(iir_scale<float>, g, dim3(256), 0, s, (const float*)in_dev, in_stride,
                         (float*)out_dev, out_stride, p->n, p->gain, initial, post);
    else
      hipLaunchKernelGGL(iir_scale<double>, g, dim3(256), 0, s, (const double*)in_dev, in_stride,
                         (double*)out_dev, out_stride, p->n, p->gain, initial, post);
    if (hipGetLastError() != hipSuccess) return iir_fail(WFK_EHIP, "IIR kernel launch failed");
    return WFK_OK;
  }
  if (!p->parts.empty()) {
    const size_t D = (size_t)p->c.D;
    for (size_t i = 0; i < p->parts.size(); ++i) {
      wfk_iir_plan* q = p->parts[i];
      const size_t Di = (size_t)q->c.D, off = (size_t)p->part_off[i];
      if (zi_dev && Di > 0 &&
          hipMemcpy2DAsync(p->zi_tmp, Di * 8, zi_dev + off, D * 8, Di * 8, (size_t)p->batch,
                           hipMemcpyDeviceToDevice, s) != hipSuccess)
        return iir_fail(WFK_EHIP, "IIR state repack failed");
      // the DC offset comes off before the first pass and goes back on after the last one only:
      // between passes the signal can be 1e-7 of the offset (a 16th-order Butterworth after its
      // first four sections), which an fp32 buffer holding signal + offset would wipe out
      const bool first = i == 0, last = i + 1 == p->parts.size();
      int rc = iir_apply_impl(q, first ? in_dev : out_dev, first ? in_stride : out_stride, out_dev,
                              out_stride, zi_dev ? p->zi_tmp : nullptr, zf_dev ? p->zf_tmp : nullptr,
                              first ? initial : 0.0, last ? post : 0.0, hip_stream, first ? src : nullptr);
      if (rc == WFK_ETIMEOUT)   // a part reported an earlier stall: no part of this plan may wait on a chain again
        for (wfk_iir_plan* r : p->parts) r->onepass = false;
      if (rc) return rc;
      if (zf_dev && Di > 0 &&
          hipMemcpy2DAsync(zf_dev + off, D * 8, p->zf_tmp, Di * 8, Di * 8, (size_t)p->batch,
                           hipMemcpyDeviceToDevice, s) != hipSuccess)
        return iir_fail(WFK_EHIP, "IIR state repack failed");
    }
    return WFK_OK;
  }
  if (p->onepass && *(volatile unsigned*)p->op_fault != 0) {
    // an earlier launch of this plan timed out in a look-back (its outputs hold NaN): say so now, and
    // serve this plan in the three-launch form from here on (no chained waits, cannot time out)
    *(volatile unsigned*)p->op_fault = 0;
    p->onepass = false;
    return iir_fail(WFK_ETIMEOUT, "IIR single pass: a look-back timed out in an EARLIER launch of this plan (a stalled or "
                                  "preempted predecessor chunk); its outputs hold NaN. The plan now runs in the three-launch form: launch again");
  }
  if (src && !p->onepass) return iir_fail(WFK_EINVAL, "sampled source on a stage that is not in the single-pass form");
  if (p->onepass) {
    int spin_limit = OP_SPIN;
    if (const char* e = getenv("WFK_IIR_SPIN")) spin_limit = atoi(e);   // (tests: force the timeout)
    // tickets restart at 0; the flags of earlier launches are told apart by the epoch
    if (hipMemsetAsync(p->op_ticket, 0, (size_t)p->batch * 64, s) != hipSuccess)
      return iir_fail(WFK_EHIP, "IIR ticket reset failed");
    const unsigned epoch = ++p->epoch;
    // Few long rows: with one chunk per workgroup 2304 / rows chunks of a row are in flight, and a
    // look-back reads every one of them.  Below OP_DEPTH_ROWS rows the grid is op_depth persistent waves
    // per row instead (WFK_IIR_OP_DEPTH: experiments).
    const int64_t ops_chunk = 64 * (int64_t)p->op_run;
    const int64_t nchunks = src ? (p->n + ops_chunk - 1) / ops_chunk : p->op_chunks;   // (iir_sampled: long chunks)
    unsigned total = (unsigned)(nchunks * p->batch);
    int persist = 0;
    {
      // from 64 rows on: about two waves per SIMD's worth of persistent waves (same box, 2 biquads, one
      // chunk per workgroup vs this: 128 x 1e7 5.86 vs 5.41 ms, 256 x 1e7 10.8 vs 9.6, 512 x 1e6 2.15 vs
      // 2.03, 1024 x 1e6 3.82 vs 3.74); fewer rows run faster with everything in flight
      int depth = 0;
      // (float rows under two biquads are the exception: 256 x 1e7 7.4 vs 8.4 ms)
      if (p->batch >= 64 && p->batch <= 1152 && !(p->kind == WFK_OUT_F32 && p->c.nsec == 2)) {
        depth = (int)((4608 + p->batch - 1) / p->batch);
        depth = depth < 4 ? 4 : (depth > 36 ? 36 : depth);
      }
      if (const char* e = getenv("WFK_IIR_OP_DEPTH")) depth = atoi(e);
      if (depth > 0 && (int64_t)depth < nchunks) {
        total = (unsigned)(depth * p->batch);
        persist = 1;
      }
    }
#define OP_LAUNCH_K(KERNEL, PL, FIRST, TABS, NCH, TT, NS, OR)                                                     \
    if (PL)                                                                                                   \
    hipLaunchKernelGGL((KERNEL<TT, NS, OR, true>), dim3(total), dim3(64), 0, s, p->c, FIRST,                   \
                       (TT*)out_dev, out_stride, p->op_status, p->op_agg, p->op_pref, p->op_ticket,              \
                       TABS, zi_dev, zf_dev, p->n, NCH, (int)p->batch, epoch,                                     \
                       initial, post, persist, p->op_fault_dev, spin_limit);                                            \
    else                                                                                                       \
    hipLaunchKernelGGL((KERNEL<TT, NS, OR, false>), dim3(total), dim3(64), 0, s, p->c, FIRST,                  \
                       (TT*)out_dev, out_stride, p->op_status, p->op_agg, p->op_pref, p->op_ticket,              \
                       TABS, zi_dev, zf_dev, p->n, NCH, (int)p->batch, epoch,                                     \
                       initial, post, persist, p->op_fault_dev, spin_limit)
#define OP_COMMA ,
#define OP_LAUNCH(TT, NS, OR)                                                                                     \
    if (src) { OP_LAUNCH_K(iir_sampled, p->op_plainL, *src, p->op_pwL OP_COMMA p->op_lanepL OP_COMMA p->op_lanepUL OP_COMMA p->op_wdot, nchunks, TT, NS, OR); } \
    else { OP_LAUNCH_K(iir_onepass, p->op_plain, (const TT*)in_dev OP_COMMA in_stride, p->op_pw1 OP_COMMA p->op_lanep1 OP_COMMA p->op_lanepU, p->op_chunks, TT, NS, OR); }
#ifdef OPS_ONLY_22     /* A/B builds (tools/iirchain_ablate.sh): one shape, a third of the compile time */
#define OP_SHAPES(TT)  do { OP_LAUNCH(TT, 2, 2); } while (0)
#else
#define OP_SHAPES(TT)                                                                              \
    do {                                                                                           \
      const int ns_ = p->c.nsec, or_ = p->c.ord[0];                                                \
      if (or_ == 2) { if (ns_ == 1) { OP_LAUNCH(TT, 1, 2); } else { OP_LAUNCH(TT, 2, 2); } }       \
      else if (or_ == 3) { OP_LAUNCH(TT, 1, 3); }                                                  \
      else if (or_ == 4) { OP_LAUNCH(TT, 1, 4); }                                                  \
      else if (ns_ == 1) { OP_LAUNCH(TT, 1, 1); }                                                  \
      else if (ns_ == 2) { OP_LAUNCH(TT, 2, 1); }                                                  \
      else if (ns_ == 3) { OP_LAUNCH(TT, 3, 1); }                                                  \
      else { OP_LAUNCH(TT, 4, 1); }                                                                \
    } while (0)
#endif
#ifdef OPS_ONLY_22
    OP_SHAPES(double);
#else
    if (p->kind == WFK_OUT_F32) OP_SHAPES(float); else OP_SHAPES(double);
#endif
#undef OP_SHAPES
#undef OP_LAUNCH
#undef OP_LAUNCH_K
#undef OP_COMMA
    if (hipGetLastError() != hipSuccess) return iir_fail(WFK_EHIP, "IIR kernel launch failed");
    return WFK_OK;
  }
  if (p->kind == WFK_OUT_F32) iir_launch<float>(p, in_dev, in_stride, out_dev, out_stride, zi_dev, zf_dev, initial, post, s);
  else iir_launch<double>(p, in_dev, in_stride, out_dev, out_stride, zi_dev, zf_dev, initial, post, s);
  if (hipGetLastError() != hipSuccess) return iir_fail(WFK_EHIP, "IIR kernel launch failed");
  return WFK_OK;
}

// Synchronise `hip_stream` and report whether a single-pass launch of this plan (or of one of its
// passes) ran into a look-back timeout since the last check.  On a fault the plan switches to the
// three-launch form, so the caller can simply launch again (not in place: the input is gone then).
extern "C" int wfk_iir_status(wfk_iir_plan* p, void* hip_stream) {
  if (!p) return iir_fail(WFK_EINVAL, "null plan");
  if (hipStreamSynchronize((hipStream_t)hip_stream) != hipSuccess) return iir_fail(WFK_EHIP, "stream synchronisation failed");

  bool fault = false;
  auto look = [&](wfk_iir_plan* q) {
    if (q->op_fault && *(volatile unsigned*)q->op_fault != 0) {
      *(volatile unsigned*)q->op_fault = 0;
      q->onepass = false;
      fault = true;
    }
  };
  look(p);
  for (wfk_iir_plan* q : p->parts) look(q);
  if (fault) {
    // one chunk chain stalled: the retry must not be able to time out in ANOTHER part either
    p->onepass = false;
    for (wfk_iir_plan* q : p->parts) q->onepass = false;
  }
  if (fault)
    return iir_fail(WFK_ETIMEOUT, "IIR single pass: a look-back timed out (a stalled or preempted predecessor chunk); the outputs "
                                  "of that launch hold NaN. The plan now runs in the three-launch form: launch again");
  return WFK_OK;
}

extern "C" int wfk_iir_apply(wfk_iir_plan* p, const void* in_dev, int64_t in_stride, void* out_dev,
                             int64_t out_stride, const double* zi_dev, double* zf_dev,
                             double initial, void* hip_stream) {
  return iir_apply_impl(p, in_dev, in_stride, out_dev, out_stride, zi_dev, zf_dev, initial, initial,
                        hip_stream);
}

}  // extern "C"

// ---- sampler -> IIR (-> FIR) chain --------------------------------------------------------------------------
// Reference: Waveform.sample(filters=(sos, initial)) (waveforms/waveform.py:190-203,244-251) and
// predistort(wav(t), filters, ker) (waveforms/distortion.py:298-337): sampler -> sosfilt / lfilter -> FIR.
extern "C" void wfk_internal_plan_tables(const wfk_plan* p, const HostPlan** h, const double** d_params);

struct wfk_chain_iir_plan {
  wfk_plan* sampler = nullptr;     // the plain sampler plan: the unfused path, queries
  wfk_iir_plan* iir = nullptr;
  wfk_fir_plan* fir = nullptr;     // optional third stage
  bool fused = false;
  std::string why;                 // why the sampler does not run inside the IIR pass
  int32_t kind = 0, n_channels = 0;
  int64_t n = 0;
  void* d_tables = nullptr;        // fused path: the plan compiled for the chunk geometry (a lane owns 32 consecutive samples)
  IirSampArgs sa{};
  int64_t table_bytes = 0;
  void* workspace = nullptr;       // FIR stage: the filtered rows between the IIR pass and the FIR
};

static wfk_iir_plan* chain_first_stage(wfk_chain_iir_plan* p) {
  return p->iir->parts.empty() ? p->iir : p->iir->parts[0];
}

extern "C" {

int wfk_chain_iir_plan_destroy(wfk_chain_iir_plan* p) {
  if (!p) return WFK_OK;
  if (p->d_tables || p->workspace) (void)hipDeviceSynchronize();
  (void)hipFree(p->d_tables);
  (void)hipFree(p->workspace);
  wfk_plan_destroy(p->sampler);
  wfk_iir_plan_destroy(p->iir);
  wfk_fir_plan_destroy(p->fir);
  delete p;
  return WFK_OK;
}

int wfk_chain_iir_plan_create(const wfk_program* prog, const wfk_grid* grid, int32_t n_sections,
                              const int32_t* orders, const double* b, const double* a, const double* ker_host,
                              int32_t K, int32_t ker_per_row, int kind, wfk_chain_iir_plan** out) {
  if (!out) return iir_fail(WFK_EINVAL, "null out");
  *out = nullptr;
  if (!prog || !grid) return iir_fail(WFK_EINVAL, "null argument");
  if (kind != WFK_OUT_F64 && kind != WFK_OUT_F32) return iir_fail(WFK_EINVAL, "chain kind must be F64 or F32");
  if (ker_host && K < 1) return iir_fail(WFK_EINVAL, "empty FIR kernel");
  wfk_chain_iir_plan* p = nullptr;
  try {
    p = new wfk_chain_iir_plan();
    p->kind = kind;
    p->n = grid->n;
    p->n_channels = prog->n_channels;
    int rc = wfk_plan_create_grid(prog, grid, &p->sampler);
    if (!rc) rc = wfk_iir_plan_create(n_sections, orders, b, a, grid->n, std::max(1, prog->n_channels), kind, &p->iir);
    if (!rc && ker_host)
      rc = ker_per_row ? wfk_fir_plan_create_rows(ker_host, K, grid->n, std::max(1, prog->n_channels), kind, &p->fir)
                       : wfk_fir_plan_create(ker_host, K, grid->n, std::max(1, prog->n_channels), kind, &p->fir);
    if (rc) { wfk_chain_iir_plan_destroy(p); return rc; }
    if (p->n == 0 || p->n_channels == 0) { *out = p; return WFK_OK; }
    const size_t es = kind == WFK_OUT_F32 ? 4 : 8;
    if (p->fir && hipMalloc(&p->workspace, (size_t)p->n_channels * (size_t)p->n * es) != hipSuccess) {
      wfk_chain_iir_plan_destroy(p);
      return iir_fail(WFK_ENOMEM, "chain workspace allocation failed");
    }
    // ---- can the sampler run inside the first IIR pass? ----------------------------------------------
    wfk_iir_plan* first = chain_first_stage(p);
    const char* off = getenv("WFK_CHAIN_UNFUSED");
    HostPlan H;
    std::string err;
    const int par_cap = OPS_PAR;     // doubles of LDS the kernel stages a piece's parameter block in
    const int64_t ops_chunk = 64 * (int64_t)(first->op_run > 0 ? first->op_run : OPS_RUN);      // samples per chunk of the fused scan
    if (off && off[0] == '1') p->why = "disabled by WFK_CHAIN_UNFUSED";
    else if (!first->onepass) p->why = "the first IIR pass is not in the single-pass form (state dimension > 4, mixed orders, a short or a mid-sized batch of long rows)";
    else if (p->n < 4 * ops_chunk) p->why = "rows shorter than four chunks of the fused scan";
    else if (wfk_compile_geom(prog, grid, 1, OP_LB, H, err) != WFK_OK) p->why = "geometry compile: " + err;
    else if (!H.lean) p->why = "plan is not fully fused (generic / direct terms, closing multipliers, or too many ops per piece)";
    else {
      for (const DevChannel& c : H.channels)
        if (c.do_clip) p->why = "clip (min/max) on a channel";
      for (uint8_t cx_ : H.channel_complex)
        if (cx_) p->why = "complex-valued channel";
    }
    // The kernel stages the pieces a chunk overlaps in LDS (descriptors + the blocks of the live ones) and lets every
    // lane read the record of its own piece: the blocks must fit, a chunk may overlap <= OPS_PMAX pieces, live pieces
    // are >= 32 samples long (a lane's 32-sample segment then meets at most two), and the live pieces of a CHUNK have
    // the same op shapes (op count, packed op words: the usual pulse train).  Anything else stays on sampler -> IIR.
    const int64_t nch = (p->n + ops_chunk - 1) / ops_chunk;
    std::vector<int32_t> chunk_first;
    auto same_shape = [&](const DevPiece& x, const DevPiece& y) {
      const double* bx = H.params.data() + x.par_off;
      const double* by = H.params.data() + y.par_off;
      bool same = bx[1] == by[1];
      for (int op = 0; same && op < (int)bx[1]; ++op)
        same = std::memcmp(bx + WFK_BLK_HDR + op * WFK_FCE_REC + WFK_FCE_DEG, by + WFK_BLK_HDR + op * WFK_FCE_REC + WFK_FCE_DEG,
                           sizeof(double)) == 0;
      return same;
    };
    if (p->why.empty())
      for (const DevPiece& pc : H.pieces) {
        if (pc.n_blk == 0) continue;
        if (pc.stop - pc.start < OP_LB) { p->why = "a piece shorter than 32 samples"; break; }
        if (pc.n_blk != 1) { p->why = "a piece of several parameter blocks"; break; }
      }
    if (p->why.empty()) {
      chunk_first.resize((size_t)nch * p->n_channels);
      for (int32_t c = 0; c < p->n_channels && p->why.empty(); ++c) {
        int32_t q = H.channels[c].piece_begin;
        for (int64_t k = 0; k < nch; ++k) {
          const int64_t s1 = k * ops_chunk, s2 = s1 + ops_chunk;
          while (q < H.channels[c].piece_end - 1 && H.pieces[q].stop <= s1) ++q;
          chunk_first[(size_t)c * nch + k] = q;
          int64_t need = 0, cnt = 0;
          int32_t first_live = -1;
          for (int32_t j = q; j < H.channels[c].piece_end && H.pieces[j].start < s2; ++j) {
            ++cnt;
            if (H.pieces[j].n_blk == 0) continue;
            need += (H.pieces[j].first_len + 1) & ~1;
            if (first_live < 0) first_live = j;
            else if (!same_shape(H.pieces[first_live], H.pieces[j]))
              p->why = "pieces of different op shapes in one chunk (the fused scan reads every lane's own piece record with one op loop)";
          }
          if (!p->why.empty()) break;
          if (need > par_cap) { p->why = "the parameter blocks of the pieces of one chunk do not fit its LDS buffer"; break; }
          if (cnt > OPS_PMAX) { p->why = "more than 16 pieces in one chunk of the fused scan (8192 | 16384 samples)"; break; }
        }
      }
    }
    if (p->why.empty()) {
      auto al = [](size_t x) { return (x + 255) & ~size_t(255); };
      const size_t b_ch = H.channels.size() * sizeof(DevChannel), b_pc = H.pieces.size() * sizeof(DevPiece),
                   b_pa = H.params.size() * sizeof(double), b_cf = chunk_first.size() * sizeof(int32_t);
      const size_t o_pc = al(b_ch), o_pa = al(o_pc + b_pc), o_cf = al(o_pa + b_pa), total = al(o_cf + b_cf) + 256;
      std::vector<char> stage(total, 0);
      std::memcpy(stage.data(), H.channels.data(), b_ch);
      std::memcpy(stage.data() + o_pc, H.pieces.data(), b_pc);
      std::memcpy(stage.data() + o_pa, H.params.data(), b_pa);
      std::memcpy(stage.data() + o_cf, chunk_first.data(), b_cf);
      if (hipMalloc(&p->d_tables, total) != hipSuccess ||
          hipMemcpy(p->d_tables, stage.data(), total, hipMemcpyHostToDevice) != hipSuccess) {
        wfk_chain_iir_plan_destroy(p);
        return iir_fail(WFK_ENOMEM, "chain table allocation failed");
      }
      char* base = static_cast<char*>(p->d_tables);
      IirSampArgs& sa = p->sa;
      sa.channels = reinterpret_cast<const DevChannel*>(base);
      sa.pieces = reinterpret_cast<const DevPiece*>(base + o_pc);
      sa.params = reinterpret_cast<const double*>(base + o_pa);
      sa.chunk_first = reinterpret_cast<const int32_t*>(base + o_cf);
      sa.nchunks = nch;
      sa.t0 = grid->t0; sa.step = grid->step; sa.last = grid->last; sa.has_last = grid->has_last;
      sa.n = grid->n; sa.i0 = grid->i0;
      p->table_bytes = (int64_t)total;
      p->fused = true;
    }
    *out = p;
    return WFK_OK;
  } catch (const std::bad_alloc&) {
    if (p) wfk_chain_iir_plan_destroy(p);
    return iir_fail(WFK_ENOMEM, "out of host memory while building the chain plan");
  }
}

/* 1 while the next launch evaluates the samples inside the IIR pass (a look-back timeout switches the plan to the
 * unfused path for good) */
int wfk_chain_iir_is_fused(const wfk_chain_iir_plan* p) {
  return p && p->fused && chain_first_stage(const_cast<wfk_chain_iir_plan*>(p))->onepass ? 1 : 0;
}

const char* wfk_chain_iir_unfused_reason(const wfk_chain_iir_plan* p) {
  if (!p) return "";
  if (p->fused && !chain_first_stage(const_cast<wfk_chain_iir_plan*>(p))->onepass)
    return "a look-back of the single pass timed out earlier: the plan runs sampler -> IIR in three launches now";
  return p->why.c_str();
}

int wfk_chain_iir_state_dim(const wfk_chain_iir_plan* p) { return p ? wfk_iir_state_dim(p->iir) : WFK_EINVAL; }

int64_t wfk_chain_iir_table_bytes(const wfk_chain_iir_plan* p) {
  if (!p) return iir_fail(WFK_EINVAL, "null plan");
  return p->fused ? p->table_bytes : wfk_plan_table_bytes(p->sampler);
}

const char* wfk_chain_iir_kernel_name(const wfk_chain_iir_plan* p) {
  if (!p) return "";
  static thread_local std::string name;
  const char* T = p->kind == WFK_OUT_F32 ? "float" : "double";
  if (wfk_chain_iir_is_fused(p)) {
    const wfk_iir_plan* f = chain_first_stage(const_cast<wfk_chain_iir_plan*>(p));
    name = std::string("iir_sampled<") + T + "," + std::to_string(f->c.nsec) + "," + std::to_string(f->c.ord[0]) + "," +
           (f->op_plain ? "true" : "false") + ">";
    if (!p->iir->parts.empty() && p->iir->parts.size() > 1) name += " + IIR passes";
  } else {
    name = std::string(wfk_plan_kernel_name(p->sampler, p->kind)) + " + IIR";
  }
  if (p->fir) name += " + FIR";
  return name.c_str();
}

int wfk_chain_iir_launch(wfk_chain_iir_plan* p, void* out_dev, int64_t out_stride, const double* zi_dev,
                         double* zf_dev, double initial, void* hip_stream) {
  if (!p) return iir_fail(WFK_EINVAL, "null plan");
  if (p->n == 0 || p->n_channels == 0) return WFK_OK;
  if (!out_dev) return iir_fail(WFK_EINVAL, "null output");
  if (out_stride < p->n) return iir_fail(WFK_EINVAL, "out_stride smaller than n");
  void* mid = p->fir ? p->workspace : out_dev;
  const int64_t mid_stride = p->fir ? p->n : out_stride;
  int rc;
  if (wfk_chain_iir_is_fused(p)) {
    rc = iir_apply_impl(p->iir, nullptr, 0, mid, mid_stride, zi_dev, zf_dev, initial, initial, hip_stream, &p->sa);
  } else {
    rc = wfk_plan_launch(p->sampler, mid, mid_stride, p->kind, 0, hip_stream);
    if (!rc) rc = iir_apply_impl(p->iir, mid, mid_stride, mid, mid_stride, zi_dev, zf_dev, initial, initial, hip_stream);
  }
  if (rc) return rc;
  if (p->fir) return wfk_fir_apply(p->fir, p->workspace, p->n, out_dev, out_stride, hip_stream);
  return WFK_OK;
}

/* as wfk_iir_status: synchronises, WFK_ETIMEOUT if a look-back of a launch since the last check timed out (outputs
 * hold NaN); launching again then takes the three-launch form behind the plain sampler */
int wfk_chain_iir_status(wfk_chain_iir_plan* p, void* hip_stream) {
  if (!p) return iir_fail(WFK_EINVAL, "null plan");
  return wfk_iir_status(p->iir, hip_stream);
}

}  // extern "C"
