// wfk_short.hip -- gfx950 sampler for SHORT pieces: the regime Waveform.sample() is used in.
//
// Reference path: calc_parts / _calc / _fill_parts (waveforms/_waveform.pyx:134-169,
// waveforms/waveform.py:524-527) on the grid of Waveform.sample (waveform.py:190:
// np.arange(start, stop, 1 / sample_rate)) at AWG rates: 1-5 GS/s, pulses of 20-40 ns, i.e.
// pieces of 20-200 samples.  The lean kernel (wfk_kernels.hip) strides a lane by 64 samples and
// carries op state from tile to tile; with pieces shorter than its 1024-sample wave tile every
// piece entry costs a libm seed per lane for 1-5 samples of use, and the Gaussian recurrence at
// stride 64 dt is not even admissible (64 dt / sigma > 2).
//
// Geometry here (see WFK_SH_* in wfk_internal.h; the host builds the tables, wfk_compile.cpp):
//   lane  -> one SEGMENT = <= R CONSECUTIVE samples of ONE piece.  The recurrences step by dt
//            (H = dt / sigma ~ 0.08: always admissible); one exact seed per (lane, op) -- short
//            straight-line sin/cos(pi r) and exp kernels, no libm call -- serves the whole run.
//            Lanes of a wave sit in DIFFERENT pieces: every op parameter is per lane, read from
//            the piece's compact record (16 doubles for a Gaussian + DRAG pulse).
//   wave  -> one UNIT = <= 64 segments plus the zero stretches between them, a contiguous range
//            of <= WFK_SH_LCAP samples of one channel.  Results go through LDS (lane l writes its
//            run at its offset; plain or stride-17 padded layout, chosen per unit by the host for
//            the fewest bank conflicts), then the wave stores the
//            range row by row: 64 consecutive samples per store instruction, rows aligned to
//            128-B lines.  Long zero stretches are pure-fill units (no slots, no LDS).
//   workgroup = one wave, walking `units_per_chunk` consecutive units; chunks are dealt to the
//            XCDs in contiguous eighths (same map as the lean kernel).
// All arithmetic is fp64 whatever the output type (the fp32 VALU rate of this part is the fp64
// rate unless packed); T only sets the width of the LDS staging and of the stores.
// Bound: HBM writes (8 / 4 / 16 B per sample) + the record, slot and unit tables (96 B per op of a
// piece, 128 B for cubics; 4 B per segment; 64 B per unit: +24 % at 60 samples per piece).
// No contraction => no MFMA.
#include <hip/hip_runtime.h>

#include <string>
#include <type_traits>
#include <utility>

#include "wfk.h"
#include "wfk_internal.h"
#include "wfk_short_dev.h"

namespace {

[[maybe_unused]] __device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ int64_t uni64(int64_t v) {
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
  const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)v >> 32));
  return (int64_t)(((uint64_t)hi << 32) | lo);
}
#define WFK_CONST __attribute__((address_space(4)))
template <typename V>
__device__ __forceinline__ V cload(const void* base, int64_t byte_off) {
  return *reinterpret_cast<const WFK_CONST V*>(reinterpret_cast<uintptr_t>(base) + byte_off);
}
template <typename P>
__device__ __forceinline__ P* uniptr(P* p) {
  using G = __attribute__((address_space(1))) P*;
  return (P*)reinterpret_cast<G>(uni64(reinterpret_cast<int64_t>(p)));
}

using namespace shdev;

template <typename T> struct ShOut;
template <> struct ShOut<double> { using Cplx = double2; };
template <> struct ShOut<float> { using Cplx = float2; };

__device__ __forceinline__ int swz(int i) { return i + (i >> 4); }   // lane stride 16 -> 17 elements

#ifndef WFK_SH_WAVES
#define WFK_SH_WAVES 3
#endif
struct UnitDesc {
  int64_t j0, rec0;
  int ch, ns, slot0, nslots, gaps, do_clip;
  double offset, clip_lo, clip_hi;
};
__device__ __forceinline__ UnitDesc load_unit(const ShortUnit* up) {
  UnitDesc u;
  u.j0 = cload<int64_t>(up, offsetof(ShortUnit, j0));
  u.ch = cload<int32_t>(up, offsetof(ShortUnit, ch));
  u.ns = cload<int32_t>(up, offsetof(ShortUnit, n_samples));
  u.slot0 = cload<int32_t>(up, offsetof(ShortUnit, slot0));
  u.nslots = cload<int32_t>(up, offsetof(ShortUnit, n_slots));
  u.gaps = cload<int32_t>(up, offsetof(ShortUnit, gaps));
  u.do_clip = cload<int32_t>(up, offsetof(ShortUnit, do_clip));
  u.offset = cload<double>(up, offsetof(ShortUnit, offset));
  u.clip_lo = cload<double>(up, offsetof(ShortUnit, clip_lo));
  u.clip_hi = cload<double>(up, offsetof(ShortUnit, clip_hi));
  u.rec0 = cload<int64_t>(up, offsetof(ShortUnit, rec0));
  return u;
}

// Raw buffer resources: the hardware range check does the masking.  A lane whose byte offset is
// >= num_records (or "negative": huge as unsigned) loads 0 / stores nothing, so masked row stores and
// the slot load of a partly filled unit are straight-line code without exec-mask branches.
// Cache policy of the row stores: non-temporal (nt).  The output is a write-once stream several times the
// size of L2 + Infinity Cache; written through L2 as ordinary lines it evicts the record and slot tables the
// neighbouring waves are about to read.  Same box, 2048 x 1e5 fp64: back-to-back pulses 0.414 -> 0.400 ms,
// 30 % duty (table lines re-used across the fill units) 0.444 -> 0.358 ms; sc0 / sc1 instead: no gain.
#ifndef WFK_SH_STORE_AUX
#define WFK_SH_STORE_AUX 2
#endif
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
template <typename E>
__device__ __forceinline__ void buf_store(const E& v, __amdgpu_buffer_rsrc_t r, int byte_off) {
  if constexpr (sizeof(E) == 4) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, byte_off, 0, WFK_SH_STORE_AUX);
  else if constexpr (sizeof(E) == 8) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), r, byte_off, 0, WFK_SH_STORE_AUX);
  else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, byte_off, 0, WFK_SH_STORE_AUX);
}
template <typename E>
__device__ __forceinline__ E buf_load(__amdgpu_buffer_rsrc_t r, int byte_off) {
  if constexpr (sizeof(E) == 4) return __builtin_bit_cast(E, __builtin_amdgcn_raw_buffer_load_b32(r, byte_off, 0, 0));
  else if constexpr (sizeof(E) == 8) return __builtin_bit_cast(E, __builtin_amdgcn_raw_buffer_load_b64(r, byte_off, 0, 0));
  else return __builtin_bit_cast(E, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0));
}

template <typename E, bool CPLX>
__device__ __forceinline__ void add_old(E& v, const E& old) {
  if constexpr (CPLX) { v.x += old.x; v.y += old.y; } else { v += old; }
}

// elements of the staging array: the longest unit (WFK_SH_LCAP samples + 15 of row alignment = 16 rows
// of 64) in the 17-per-16 swizzle, plus the overrun of a last partial row
constexpr int kStage = 68 * 16 + 16;

// Software pipeline over the units of a chunk.  Every stage of the chain unit -> slot -> op record
// is a dependent memory round trip of 1-2 us, as long as the evaluation of a whole unit, and on this
// ISA loads and stores share ONE in-order counter (vmcnt): waiting for a load also waits for every
// store issued before it.  So:
//   * at the top of unit u the wave holds desc(u), desc(u+1), the decoded slot of u and the first op
//     record of u (issued BEFORE the stores of u-1: the wait for it leaves those stores in flight);
//   * it issues desc(u+2) and slot(u+1), evaluates u, decodes slot(u+1), issues the first record of
//     u+1, then stages and stores u;
//   * a unit is stored by a FIXED sequence of 16 masked row stores (units span <= 16 rows, the host
//     sees to that): with a variable row loop the compiler cannot count the stores behind a load and
//     falls back to vmcnt(0) at the loop's back edge -- every wave then idles until its own stores
//     have been acknowledged (measured: 6 us per unit).
// FAM: the op family the plan needs (HostPlan::short_fam) -- 0: carrier-envelope ops and nothing else (the plain pulse train:
// the body of round 3); 1: + the closing ops of flat-top edges (erf) and multi-tone pieces (shared Gaussian), linear chirps,
// envelope seeds skipped where no lane has an envelope; 2: + table / mollifier envelopes (closing multipliers, own-term ops); 3: family 0
// in packed fp32 (real float launches); 4: + exponential / hyperbolic chirp multipliers; 6: family 0 + carriers with the
// grid-rounding correction (pulse trains milliseconds from t = 0).  Separate instantiations, so that a shape added to one family cannot move the code generation of the others (the
// chirp / cmul / shared-envelope ops of round 4, inlined into the one body, cost the plain pulse train 23 % more VALU
// instructions and 113-120 spilled SGPRs).
#ifndef WFK_SH_WAVES6
#define WFK_SH_WAVES6 2       // family 6 (corrected carriers): 219 registers and no spill at two waves per SIMD against 168 + 46 spilled
#endif                        // at three -- rows 1 ms from t = 0, same box: 1.06 -> 0.835 ms
template <typename T, bool CPLX, bool ACC, int R, int FAM>
__global__ void __launch_bounds__(64, CPLX ? 2 : (FAM == 6 ? WFK_SH_WAVES6 : WFK_SH_WAVES)) wfk_sample_short(const SArgs a) {
  using E = typename std::conditional<CPLX, typename ShOut<T>::Cplx, T>::type;
  // float launches of the plain pulse train (family 3 = family 0 in packed fp32 arithmetic, wfk_short_dev.h: short_op_pk)
  constexpr bool PK = FAM == 3;
  // what a family holds beside the carrier-envelope ops: F1 closing ops of flat tops / multi-tone pieces and linear chirps,
  // F2 table / mollifier envelopes, F4 exponential / hyperbolic chirp multipliers (libm log behind a call: a family of
  // its own, so that the table envelopes keep their register allocation)
  constexpr bool F1 = FAM == 1 || FAM == 2 || FAM == 4, F2 = FAM == 2 || FAM == 4, F4 = FAM == 4;
  // family 6 = family 0 + carriers with the grid-rounding correction (pulse trains milliseconds from t = 0: short_op_corr)
  constexpr bool F6 = FAM == 6;
  static_assert(!PK || (std::is_same<T, float>::value && !CPLX), "packed arithmetic: real float launches only");
  __shared__ __attribute__((aligned(16))) E s_out[kStage];
  const int lane = threadIdx.x;
  // XCD-aware chunk map (workgroup b runs on XCD b % 8): XCD x walks the x-th contiguous eighth
  const int64_t b = blockIdx.x;
  const int64_t per = (a.n_chunks + 7) >> 3;
  const int64_t lchunk = (b & 7) * per + (b >> 3);
  if (lchunk >= a.n_chunks) return;
  const int64_t chunk = a.chunk_base + lchunk;
  const int64_t u0 = chunk * a.units_per_chunk;
  const int64_t u1 = u0 + a.units_per_chunk < a.n_units ? u0 + a.units_per_chunk : a.n_units;
  const int64_t ulast = a.n_units - 1;

  auto slot_of = [&](const UnitDesc& d) -> uint32_t {     // lanes >= nslots read 0: no segment
    return buf_load<uint32_t>(make_rsrc(a.slots + d.slot0, (unsigned)d.nslots * 4u), lane * 4);
  };
  // (len, o, first sample's index, record) of this lane's segment
  struct Seg { int len, o, j; const double* rec; };
  auto decode = [&](uint32_t slot, const UnitDesc& d) -> Seg {
    Seg g;
    g.len = (slot >> 31) ? (int)((slot >> 26) & 15) + 1 : 0;
    g.o = (int)((slot >> 16) & 0x3ff);
    g.j = (int)(uint32_t)d.j0 + g.o;
    g.rec = a.recs + 2 * (d.rec0 + (int64_t)(slot & 0xffff));
    return g;
  };

  // The pipeline fills INSIDE the loop: iteration u0 - 1 works on an empty unit (no slots, zero samples:
  // the range check drops its stores).  With a separate prologue the compiler merges two histories at the
  // loop header and waits for the record as if only the prologue's single load followed it (vmcnt(1):
  // every store of the previous unit), instead of leaving the 16 stores and the slot load in flight.
  UnitDesc cur{};
  UnitDesc nxt = load_unit(a.units + u0);
  Seg seg{0, 0, 0, a.recs};
  OpRec first{};

  for (int64_t ui = u0 - 1; ui < u1; ++ui) {
    // two units ahead: descriptor; one ahead: slot words (none past the end of the chunk)
    const UnitDesc nn = load_unit(a.units + (ui + 2 <= ulast ? ui + 2 : ulast));
    if (ui + 1 >= u1) { nxt.nslots = 0; nxt.ns = 0; }
    const uint32_t nslot = slot_of(nxt);
    E* const orow = uniptr(reinterpret_cast<E*>(a.out) + (int64_t)cur.ch * a.ch_stride + cur.j0);
    const int head = (int)(cur.j0 & 15);     // rows start on 16-sample boundaries (whole 128-B lines for fp64)
    const int ns = cur.ns;
    E fillv;
    if constexpr (CPLX) { fillv.x = (T)cur.offset; fillv.y = (T)0; } else { fillv = (T)cur.offset; }
    const int l0 = lane - head;                 // this lane's sample in row 0 (negative: before the unit)
    // Staging layout, chosen per unit by the host for the unit's own segment offsets: plain (sample i at
    // element i: conflict-free when the runs start an odd number of samples apart, e.g. 15), or padded by
    // one element per 16 (i + (i >> 4): runs of 16 then start 17 apart).  Either way row r, lane x sits at
    // rs * r + f(x), one per-lane base and a uniform row stride.
    const bool sw = (cur.gaps & 2) != 0;
    const int rs = sw ? 68 : 64;
    const int s0 = sw ? l0 + (l0 >> 4) : l0;    // (arithmetic shift: exact for the negative l0 of row 0 too)

    double acc[PK ? 1 : R], acci[CPLX ? R : 1];
    f32x2 accp[PK ? R / 2 : 1];
    SH_EACH(PK ? 1 : R, k) acc[k] = 0.0; SH_END
    SH_EACH(CPLX ? R : 1, k) acci[k] = 0.0; SH_END
    SH_EACH(PK ? R / 2 : 1, k) accp[k] = 0.0f; SH_END

    if (cur.nslots != 0) {
      const double kf = (double)(seg.j - op_ref(first));   // samples from the record's reference sample
      const double* op = seg.rec;
      OpRec rec = first;
      bool live = seg.len > 0;
      // The first op is evaluated in straight-line code, further ops (multi-tone pieces) in a loop:
      // a loop whose body reads registers loaded before it gets a vmcnt(0) in its preheader from the
      // compiler (SIInsertWaitcnts' preheader flush), which here would wait for the slot load just
      // issued and for every store of the previous unit.
      auto eval = [&](const OpRec& rc, const double* opp, bool lv) -> bool {   // -> another op follows
        // (the unused halves of the record stay "live" up to here: registers that die at the load are
        //  handed out again as temporaries while the load is still in flight, and writing them means
        //  waiting for it -- a vmcnt(3) right behind the prefetch)
        asm volatile("" : : "v"(rc.a.x));
        const int w = op_word(rc);
        const bool closing = ((w >> 4) & 3) == 3;       // closing multiplier (erf edge, table, mollifier)
        const bool mine = lv && !closing && (CPLX || !(w & 8));   // op of the imaginary part: a real launch keeps .real
        const bool chirp = F1 && (w & 512) != 0;  // quadratic phase (16-double record, polynomials of degree <= 1)
        const bool cubic = __any(mine && !chirp && (w & 3) > 1 && !(F6 && (w & 4096)));
        if constexpr (PK) {
          if (mine) {
            if (cubic) short_op_pk<R, true>(rc, opp, w, kf, a.step, accp);
            else short_op_pk<R, false>(rc, opp, w, kf, a.step, accp);
          }
          return lv && !(w & WFK_SH_LAST);
        } else {
        if constexpr (F1) {
          if (__any(mine && chirp)) {
            if (mine && chirp) short_chirp<R, CPLX>(rc, opp, w, kf, a.step, acc, acci);
          }
        }
        // bare carriers (degree 0, no envelope): the tones of a multi-tone piece, plateaus -- half the instructions per sample
        const bool bare = F1 && !chirp && (w & 0x33) == 0 && (w & 4) != 0;
        if constexpr (F1) {
          if (__any(mine && bare)) {
            if (mine && bare) short_op_bare<R, CPLX>(rc, w, kf, acc, acci);
          }
        }
        bool corr = false;
        if constexpr (F6) {
          corr = (w & 4096) != 0;
          if (__any(mine && corr)) {
            if (mine && corr) {
              ShortCorr cc;
              cc.dj0 = a.di0 + (double)seg.j;
              cc.step = a.step; cc.t0 = a.t0; cc.last = a.last; cc.dlast = a.dlast;
              short_op_corr<R, CPLX>(rc, opp, w, kf, cc, acc, acci);
            }
          }
        }
        if (mine && !chirp && !bare && !corr) {
#ifndef WFK_SH_SKIP0
#define WFK_SH_SKIP0 0
#endif
          if (cubic) short_op<R, true, CPLX, (F1 || WFK_SH_SKIP0)>(rc, opp, w, kf, a.step, acc, acci);
          else short_op<R, false, CPLX, (F1 || WFK_SH_SKIP0)>(rc, opp, w, kf, a.step, acc, acci);
        }
        if constexpr (F1)
        if (__any(lv && closing)) {
          const bool own = F2 && (w & 128) != 0;      // envelope x carrier in one op: adds its own term
          if constexpr (F2) {
            if (__any(lv && closing && own)) {
              if (lv && closing && own && (CPLX || !(w & 8))) short_cmul<R, CPLX>(rc, a.pool, w, kf, acc, acci);
            }
          }
          const bool xch = F4 && (w & 1024) != 0;    // chirp multiplier (exponential / hyperbolic)
          if constexpr (F4) {
            if (__any(lv && closing && xch)) {
              if (lv && closing && xch) short_xchirpmul<R, CPLX>(rc, w, kf, acc, acci);
            }
          }
          const int kind = (own || xch) ? -1 : (w & 3);      // 0: erf edge; 1: shared Gaussian; 2: INTERP table, 3: mollifier (stateless multipliers)
          if constexpr (F1) {
            if (__any(lv && closing && kind == 1)) {
              if (lv && closing && kind == 1) short_envmul<R, CPLX>(rc, kf, acc, acci);
            }
          }
          if (__any(lv && closing && kind == 0)) {
            if (lv && closing && kind == 0) short_erfmul_run<R, CPLX>(rc, kf, seg.len, acc, acci);
          }
          if constexpr (F2) {
            if (__any(lv && closing && kind == 2)) {
              if (lv && closing && kind == 2) short_tabmul<R, CPLX>(rc, a.pool, kf, acc, acci);
            }
            if (__any(lv && closing && kind == 3)) {
              if (lv && closing && kind == 3) short_mollmul<R, CPLX>(rc, kf, acc, acci);
            }
          }
        }
        return lv && !(w & WFK_SH_LAST);
        }
      };
      live = eval(rec, op, live);
      while (__any(live)) {
        op += (op_word(rec) & 3) > 1 ? WFK_SH_OP3 : WFK_SH_OP1;
        if (live) rec = load_op(op);
        live = eval(rec, op, live);
      }
    }

    // the next unit's segment and first op record: in flight while this unit is staged and stored
    const Seg nseg = decode(nslot, nxt);
    first = load_op(nseg.rec);

    if (cur.nslots != 0) {
      __syncthreads();            // the previous unit's store phase is done with the staging array
      if (cur.gaps & 1) {
        const int f0 = sw ? swz(lane) : lane;
        SH_EACH(16, r)
          if (64 * r < ns) s_out[f0 + rs * r] = fillv;
        SH_END
        __syncthreads();
      }
      // clip (evaluated pieces only: every slot is one), + offset
      if (cur.do_clip) {
        if constexpr (CPLX) {
          // np.clip of complex values: lexicographic against the real bounds (see clip_np_cplx in wfk_kernels.hip)
          SH_EACH(R, k)
            if (acc[k] < cur.clip_lo || (acc[k] == cur.clip_lo && acci[k] < 0.0)) { acc[k] = cur.clip_lo; acci[k] = 0.0; }
            if (acc[k] > cur.clip_hi || (acc[k] == cur.clip_hi && acci[k] > 0.0)) { acc[k] = cur.clip_hi; acci[k] = 0.0; }
          SH_END
        } else if constexpr (PK) {
          // (the double path clips the fp64 sum and rounds once; here the float sum is clipped against the bounds --
          //  np.clip semantics, NaN propagates)
          SH_EACH(R, k)
            float v = accp[k / 2][k % 2];
            v = v < (float)cur.clip_lo ? (float)cur.clip_lo : v;
            v = v > (float)cur.clip_hi ? (float)cur.clip_hi : v;
            accp[k / 2][k % 2] = v;
          SH_END
        } else {
          SH_EACH(R, k) acc[k] = clip_np(acc[k], cur.clip_lo, cur.clip_hi); SH_END
        }
      }
      {
        // padded layout: element o + k lands at swz(o) + k + [k >= 16 - (o & 15)]: two bases, immediate offsets
        const int o = seg.o, len = seg.len;
        const int t = sw ? 16 - (o & 15) : 99;
        E* const b0 = s_out + (sw ? swz(o) : o);
        SH_EACH(R, k)
          if (k < len) {                      // (a masked LDS write needs no wait: the branch is cheap)
            E* const at = (k >= t ? b0 + 1 : b0) + k;
            double v;
            if constexpr (PK) v = (double)(accp[k / 2][k % 2] + (float)cur.offset);
            else v = acc[k] + cur.offset;
            if constexpr (CPLX) {
              E e;
              e.x = (T)v;
              e.y = (T)acci[k];
              *at = e;
            } else {
              *at = (T)v;
            }
          }
        SH_END
      }
      __syncthreads();
    }

    // store the unit's range: 64 consecutive samples per instruction, 16 masked rows in two batches of
    // eight.  The LDS reads are unconditional (row 0's index clamped: lanes before the unit read element
    // 0 and never store it); a pure-fill unit (a zero stretch: skipped _zero pieces, no clip,
    // waveforms/_waveform.pyx:160-163) stores `offset`.
    const __amdgpu_buffer_rsrc_t ores = make_rsrc(orow, (unsigned)ns * (unsigned)sizeof(E));
    const int b0off = l0 * (int)sizeof(E);                  // row 0: "negative" for the lanes before the unit
    const int b1off = (64 + l0) * (int)sizeof(E);           // rows >= 1: never negative
    SH_EACH(2, hb)
      E v[8];
      SH_EACH(8, rr)
        constexpr int r = hb * 8 + rr;
        v[rr] = fillv;
        if (cur.nslots != 0) v[rr] = s_out[r == 0 ? max(s0, 0) : s0 + rs * r];
      SH_END
      if constexpr (ACC) {
        SH_EACH(8, rr)
          constexpr int r = hb * 8 + rr;
          add_old<E, CPLX>(v[rr], buf_load<E>(ores, r == 0 ? b0off : b1off + 64 * (r - 1) * (int)sizeof(E)));
        SH_END
      }
      SH_EACH(8, rr)
        constexpr int r = hb * 8 + rr;
        buf_store<E>(v[rr], ores, r == 0 ? b0off : b1off + 64 * (r - 1) * (int)sizeof(E));
      SH_END
    SH_END

    cur = nxt;
    nxt = nn;
    seg = nseg;
  }
}

template <typename T, bool CPLX>
int launch_short(const SArgs& a, hipStream_t s) {
  const int64_t blocks = ((a.n_chunks + 7) >> 3) << 3;
  if (blocks == 0) return 0;
  if (blocks > 0x7fffffffLL) return -2;
  if (a.lds_samples > WFK_SH_LCAP) return -3;
#define SH_LAUNCH(ACCV, FAMV) hipLaunchKernelGGL((wfk_sample_short<T, CPLX, ACCV, WFK_SH_R, FAMV>), dim3((unsigned)blocks), dim3(64), 0, s, a)
  // (real float launches of family 0 run its packed-fp32 build, "family 3")
  constexpr bool kPk = std::is_same<T, float>::value && !CPLX;
  if (a.accumulate) {
    if (a.fam <= 0) { if constexpr (kPk) { if (a.pk) SH_LAUNCH(true, 3); else SH_LAUNCH(true, 0); } else SH_LAUNCH(true, 0); }
    else if (a.fam == 1) SH_LAUNCH(true, 1); else if (a.fam == 2) SH_LAUNCH(true, 2); else if (a.fam == 6) SH_LAUNCH(true, 6); else SH_LAUNCH(true, 4);
  } else {
    if (a.fam <= 0) { if constexpr (kPk) { if (a.pk) SH_LAUNCH(false, 3); else SH_LAUNCH(false, 0); } else SH_LAUNCH(false, 0); }
    else if (a.fam == 1) SH_LAUNCH(false, 1); else if (a.fam == 2) SH_LAUNCH(false, 2); else if (a.fam == 6) SH_LAUNCH(false, 6); else SH_LAUNCH(false, 4);
  }
#undef SH_LAUNCH
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace

int wfk_launch_short(const SArgs& a, int out_kind, void* stream, std::string& err) {
  hipStream_t s = (hipStream_t)stream;
  int rc;
  switch (out_kind) {
    case WFK_OUT_F64: rc = launch_short<double, false>(a, s); break;
    case WFK_OUT_F32: rc = launch_short<float, false>(a, s); break;
    case WFK_OUT_C128: rc = launch_short<double, true>(a, s); break;
    case WFK_OUT_C64: rc = launch_short<float, true>(a, s); break;
    default: err = "bad out_kind"; return WFK_EINVAL;
  }
  if (rc == -2) { err = "grid too large"; return WFK_EINVAL; }
  if (rc == -3) { err = "short plan: unit longer than the staging array"; return WFK_EINVAL; }
  if (rc) { err = std::string("short kernel launch failed: ") + hipGetErrorString(hipGetLastError()); return WFK_EHIP; }
  return WFK_OK;
}
