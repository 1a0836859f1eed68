// wfk_internal.h -- device-side program layout shared by the host compiler
// (wfk_compile.cpp) and the gfx950 kernels (wfk_kernels.hip).
//
// Data layout in HBM (all tables are tiny next to the output stream):
//   DevChannel[n_channels]    offset / tshift / clip + piece range
//   DevPiece[n_pieces]        DISJOINT sample ranges covering [0, n) per channel,
//                             merged over WaveVStack members on the host
//   double params[]           per-piece "parameter blocks" (see below); staged
//                             into LDS by the workgroup that evaluates the piece
//   double pool[]             variable-length args (INTERP points, MOLLIFIER
//                             polynomial), read per lane
//   int32 chunk_first[]       first piece overlapping each workgroup chunk
//
// Parameter block (doubles; integers stored exactly as doubles):
//   [0] block length (doubles, this header included)   [1] n_ops
//   per op:  kind, then
//     kind 1 (generic term): amp_re, amp_im, n_factors, then per factor WFK_FREC
//                            doubles { mode, power, shift, a0..a5, aux }
//     kind 2 (fused carrier-envelope group, WFK_FCE_REC doubles): see WFK_FCE_*
//   then phasor tables referenced by offset inside the block (even => 16-B aligned)
#pragma once
#include <stdint.h>

#define WFK_WG 256            // threads per workgroup (4 wave64)
#define WFK_NS_GRID 16        // samples per lane per wave tile, grid mode
#define WFK_NS_TLIST 8        // samples per lane per wave tile, tlist mode
#define WFK_NS_TLIST_SMALL 1  // ... when n < WFK_TLIST_SMALL_N: a 10 001-point call (the README case)
                              // then spreads over 40 workgroups instead of 5 (device libm per sample:
                              // the launch is latency-bound, not byte-bound)
#define WFK_TLIST_SMALL_N (1 << 20)
#define WFK_LDS_DOUBLES 2048  // LDS parameter buffer (16 KiB)
#define WFK_FREC 10           // doubles per factor record
#define WFK_BLK_HDR 2
#define WFK_TERM_HDR 4        // kind, amp_re, amp_im, n_factors
#define WFK_OP_TERM 1
#define WFK_OP_FCE 2

// Fused carrier-envelope op: the host rewrites every term of the shape
//   amp * LINEAR^p * [GAUSSIAN] * COS*COS*...      (real amp, p <= 3)
// by product-to-sum into single carriers and merges all terms of a piece that share
// (envelope, carrier frequency W) into   E(t) * ( A(u) cos(th) + B(u) sin(th) ),
//   th = W (t - s_ref),  u = t - s_lin,  A/B polynomials of degree <= 3,
//   E = 1 or exp(-((t - s_g)/sigma)^2).
// Device cost: ONE phasor seed + (optionally) one Gaussian seed per lane per tile.
#define WFK_FCE_REC 22
#define WFK_FCE_W 1
#define WFK_FCE_SREF 2
#define WFK_FCE_SLIN 3
#define WFK_FCE_DEG 4         // PACKED op word (one LDS read instead of five dependent ones per op and tile):
                              // deg (bits 0-1) | carrier << 2 | imag << 3 | env << 4 (2 bits) | f32ok << 6 |
                              // corr << 7 (per-sample grid-rounding correction of the carrier) | table offset << 8
                              // (11 bits) | state offset << 19 (6 bits, units of 128 doubles of the lean kernel's
                              // per-lane state) | has (c, s) state << 25 | has (g, r) state << 26 | exp envelope << 27
#define WFK_FCE_PACK(deg, carrier, imag, env, f32ok) ((deg) | ((carrier) << 2) | ((imag) << 3) | ((env) << 4) | ((f32ok) << 6))
// The word travels as a 32-bit INTEGER in the low half of its slot (the host converts when it closes a
// block): the device reads it with one 32-bit load and decodes it on the scalar unit -- as a double it
// had to go through v_cvt_i32_f64, and every flag test after it became a VALU compare + vcc branch.
#define WFK_FCE_WORD(rec) (reinterpret_cast<const int*>(rec)[2 * WFK_FCE_DEG])
#define WFK_FCE_TABOFF(fl) (((fl) >> 8) & 0x7FF)
#define WFK_FCE_STOFF(fl) ((((fl) >> 19) & 63) * 128)
#define WFK_FCE_HAS_CS (1 << 25)
#define WFK_FCE_HAS_GR (1 << 26)
#define WFK_FCE_CHIRP (1 << 28)    // quadratic phase: W = W' (frequency at SREF = tref), WM = K, SM = phase at tref,
                                   // TAB / F32OK = (cos, sin)(2 K D^2); state: (c, s), the step phasor (wc, ws), then (g, r)
#define WFK_FCE_TLSMALL (1 << 29)   // tlist plans (ops evaluated pointwise): |W (t' - s_ref)| <= 1.6e6 over the piece -> cheap phase reduction
#define WFK_FCE_OWNMUL (1u << 31)   // lean family 4: a plain carrier op (degree 0) whose term is multiplied by an envelope of its own --
                                    // acc += F(u_k) A0 cos(th_k), F a table (B+3 = 0) or a mollifier (B+3 = 1): envelope shift in SLIN,
                                    // table: B+2 start, A+1 m - 1, A+2 (m - 1) / (stop - start), A+3 first entry, B+1 lane stride in knots;
                                    // mollifier: A+1 = 1 / r.  Any number per piece (overlapping pulses of different envelopes).
#define WFK_FCE_BANK (1 << 30)      // first of a run of bare carriers (degree 0, no envelope, no correction, same part of the output):
                                    // SIGMA holds the run's length; the lean kernel (family >= 1) evaluates the run in one compact loop
#define WFK_FCE_EXPENV (1 << 27)   // the envelope is exp(alpha (t - ref)): SIGMA = alpha, SG = ref, H = alpha * D, Q = 1
#define WFK_FCE_A 5           // A0..A3
#define WFK_FCE_B 9           // B0..B3
#define WFK_FCE_WM 13         // corr ops: |w| of the reference COS factor whose rounded phase is mimicked
#define WFK_FCE_SIGMA 14
#define WFK_FCE_SG 15
#define WFK_FCE_H 16
#define WFK_FCE_Q 17
#define WFK_FCE_F32OK 18
#define WFK_FCE_TAB 19
#define WFK_FCE_D 20
#define WFK_FCE_SM 21         // corr ops: shift of that factor

// factor evaluation modes (record slot 0).  1..15 = direct evaluation of that
// primitive with device libm; >=100 = uniform-grid fast paths (power == 1).
#define WFK_M_LIN_REC 101     // u_n = u_0 + n*D                       a0 = D = 64*step
#define WFK_M_GAUSS_REC 102   // g_{n+1}=g_n r_n, r_{n+1}=r_n q        a0 = sigma, a1 = H, a2 = q
#define WFK_M_COS_TAB 104     // cos(th0 + n*dth) = c0*C[n] - s0*S[n]  a0 = w, aux = table
#define WFK_M_EXP_REC 106     // e_{n+1} = e_n * rho                   a0 = alpha, a1 = rho
#define WFK_M_SAMPLED 20      // caller-evaluated factor (WFK_SAMPLED): a0 = pool offset of the values, a1 = i0, a2 = m
#define WFK_M_REUSE 1000      // added to a direct factor's type: same record as the previous direct factor of
                              // the block -> its values are still in the LDS value buffer
#define WFK_M_MOLL_REC 111    // mollifier(width) (d = 0): exp(1 / ((u/r)^2 - 1) + 1) inside |u| < r, inline reciprocal and exponential   a0 = r
#define WFK_M_SINC_TAB 110    // sinc(b u) = sin(pi b u) / (pi b u): sin by the phasor table (as COS_TAB), the argument advanced by the
                              // SAME per-stride phase, one reciprocal per sample   a0 = b, a1 = pi b D (the table's phase step), aux = table
#define WFK_M_INTERP_LIN 109  // the same table (finite values) read as a continuous piecewise-linear function: knot index from
                              // the O(1) guess alone, x clamped to [start, stop]; same record layout as INTERP_GRID
#define WFK_M_INTERP_GRID 108 // np.interp on linspace knots, exact per-sample times, knot/slope loads
                              // batched 4 samples deep: a0=start a1=stop a2=m a3=pool(fp) a4=pool(slopes) a5=1/step

// "lean" plans: every piece is one block of <= WFK_LEAN_OPS fused ops and nothing else.
// They run on the wave-per-workgroup kernel that carries per-lane op state across tiles.
#ifndef WFK_LEAN_OPS
#define WFK_LEAN_OPS 15       // upper limit; LDS is reserved for what the plan actually has: 1 KB per op for a
                              // phasor (c, s), 1 KB for a Gaussian / erf state (g, r).  Same-box A/B of 11 vs
                              // 17 (64 x 1e7, Gaussian pulses under N tones): 12 tones 4.0 vs 3.4 ms, 14 tones
                              // 4.5 vs 4.2, 16 tones 5.0 vs 5.3 -- the general kernel wins from 16 on
#endif
// doubles of LDS parameter buffer per wave: header + 4 ops x (22 record + 34 table) = 232.
// With the 8 KB of per-lane op state that makes 10 KB per wave = 16 waves per CU, which the
// fp32 kernel (<= 128 VGPRs) uses: 2.11 -> 1.95 ms on 256 x 1e7 fp32 against 512 doubles.
#ifndef WFK_LEAN_PAR
#define WFK_LEAN_PAR 896      // upper limit (15 ops x 56 doubles + header), reserved per plan as needed
#endif
#define WFK_CHAIN_PAR 2048    // sampler inside the FIR transform: largest parameter block (doubles); it is staged
                              // in the transform's exchange array (18.5 KB in the float kernel)
#ifndef WFK_LEAN_RESEED
#define WFK_LEAN_RESEED 8     // exact libm reseed every this many tiles (fp64 outputs: drift 9e-13 over 128 steps)
#endif
#define WFK_LEAN_RESEED_F32 32 // ... float outputs
#ifndef WFK_LEAN_RESEED_CS
#define WFK_LEAN_RESEED_CS 64  // ops that carry a phasor only (advanced once per TILE, not per sample): exact reseed every this many tiles
#endif
#define WFK_LEAN_TPC 5         // tiles per chunk of a double lean launch (upper limit; WFK_TPC overrides)
#define WFK_LEAN_TPC_F32 20    // ... whose lean launches take up to this many tiles per chunk (8 for double), as
                              // long as ~8 chunks per resident wave remain (total tiles / 24576: C3 gets 10);
                              // same box, cap 8 / 12 / 16 / 20 / 24 / 32: fp32 256 x 1e7 1.99 / 1.91 / 1.88 / 1.88 /
                              // 1.91 / 1.87 ms, C3 0.211 / 0.207 / 0.201 / 0.197 / 0.199 / 0.201 ms

#define WFK_PF_HAS_TERMS 1    // piece is "evaluated": clip applies (pyx:161-163)
#define WFK_PF_LEAN 2         // piece is one block of <= WFK_LEAN_OPS fused ops (lean kernel can take it)

// ---- "short" plans: pieces of tens to hundreds of samples (AWG sample rates) ------------------
// The reference's users sample at 1-5 GS/s (Waveform.sample, waveforms/waveform.py:173-207): a
// 20 ns pulse is 20-100 samples, far below the lean kernel's wave tile of 1024 in which a lane
// strides 64 samples.  The short tier turns the geometry round (wfk_short.hip):
//   * a LANE owns one SEGMENT: <= WFK_SH_R CONSECUTIVE samples of ONE piece, so the Gaussian /
//     phasor recurrences step by dt (H = dt / sigma is always admissible) and the one exact seed
//     per (lane, op) is amortised over the lane's run;
//   * a wave owns one UNIT: <= 64 segments (+ the zero gaps between them) covering a contiguous
//     sample range of <= WFK_SH_LCAP samples; results are transposed through LDS so that the wave
//     still writes whole 128-B lines;
//   * pieces carry COMPACT op records (16 or 20 doubles per op instead of 22 + a 34-double
//     phasor table): at 60 samples per piece the tables are no longer negligible traffic.
// Host tables: ShortUnit[] (one per wave unit), uint32 slots[] (one per segment), records in
// `params`.  At 60 samples per piece the tables are real traffic (output 480 B per piece), so they
// are kept small: 96 B per op, 4 B per segment.  Record of a piece (or of each <= WFK_SH_SUB-sample
// stretch of a long piece): its ops back to back, 16-B aligned, WFK_SH_OP1 doubles each
// (WFK_SH_OP3 when deg > 1):
//     [0] low half: deg | carrier << 2 | imag << 3 | env << 4 (0 none, 1 Gaussian, 2 exp; 3: the op is the
//         closing erf multiplier of a flat-top edge: [5] v0, [6] H as for a Gaussian, [8] m0, [9] m1) | last op << 6
//         high half: the record's reference sample (index in the channel): a lane's segment starts
//         koff = j0 + o - ref samples after it
//     [1] th0/pi at the reference sample, reduced to [-1, 1]    [2] W dt / pi
//     [3] cos(W dt)  [4] sin(W dt)
//     [5] Gaussian: v0 = (x_ref - s_g) / sigma; exp: alpha (x_ref - ref)
//     [6] Gaussian: H = dt / sigma; exp: alpha dt        [7] q = exp(-2 H^2) (exp / none: 1)
//     [8] A0 [9] A1 [10] B0 [11] B1   ([12] A2 [13] A3 [14] B2 [15] B3): polynomials in u = t - x_ref
// (Taking cos / sin(W dt) from [2] on the device instead -- 80-B records -- was measured: the extra
//  sin/cos kernel per lane and op cost more than the 16 B saved: 0.415 -> 0.446 ms on the AWG workload.)
#define WFK_SH_R 16           // samples per lane segment
#define WFK_SH_LCAP 1008      // samples a unit may span: with <= 15 samples of row alignment that is <= 16 rows of 64
                              // (the kernel stores a unit with a FIXED sequence of 16 masked row stores)
#define WFK_SH_SUB 4096       // samples per record of a long piece (phase = th0 + koff dth: koff stays small)
#define WFK_SH_FILL 1008      // samples per pure-fill unit (long zero stretches: no slots, no LDS; same 16 rows)
#define WFK_SH_ERFTAB 512     // flat-top edges of at most this many samples carry their multiplier values behind the record's ops
#define WFK_SH_OP1 12
#define WFK_SH_OP3 16
#define WFK_SH_LAST 64        // op word: last op of its record
#define WFK_PF_SHORT 4        // piece carries compact records (par_off, n_blk = #records, first_len = doubles each)
// slot word: record offset from the unit's first record in 16-B units (16 bits) | sample offset in the
//            unit << 16 (10 bits) | (segment length - 1) << 26 (4 bits) | valid << 31
#define WFK_SH_SLOT(drec16, o, len) \
  ((uint32_t)(drec16) | ((uint32_t)(o) << 16) | ((uint32_t)((len) - 1) << 26) | 0x80000000u)
#define WFK_SH_DREC_MAX 0xffff

struct ShortUnit {            // 64 B: everything a wave needs about its unit in one scalar load
  int64_t j0;                 // first sample of the unit in its channel
  int32_t ch;
  int32_t n_samples;          // samples covered (<= WFK_SH_LCAP)
  int32_t slot0;              // first slot
  int32_t n_slots;            // 0: pure fill (`offset` everywhere)
  int32_t gaps;               // bit 0: the slots do not cover the range: LDS is pre-filled with `offset`; bit 1: the
                              // staging array is padded by one element per 16 (segments that start 16 apart)
  int32_t do_clip;            // channel constants, copied here: no dependent second load
  double offset, clip_lo, clip_hi;
  int64_t rec0;               // first record of the unit, in 16-B units of `params`
};

struct DevPiece {
  int64_t start, stop;        // sample range [start, stop)
  int64_t par_off;            // first parameter block (doubles into params[])
  int32_t n_blk;              // consecutive blocks (0: zero piece)
  int32_t flags;
  int32_t first_len;          // length of the first block
  int32_t pad;
};

struct DevChannel {
  double offset, tshift, clip_lo, clip_hi;
  int32_t piece_begin, piece_end;
  int32_t do_clip, pad;
};

struct SArgs {                // short-tier launch (wfk_short.hip)
  const DevChannel* channels;
  const ShortUnit* units;
  const uint32_t* slots;
  const double* recs;         // == the plan's `params`
  void* out;
  int64_t ch_stride;          // elements
  int64_t n_units;
  int64_t n_chunks;           // workgroups with work; the grid is rounded up to a multiple of 8
  int64_t chunk_base;         // first chunk of this launch (a launch may cover a sub-range: wfk_plan_run_host's pipeline)
  int32_t units_per_chunk;
  int32_t accumulate;
  int32_t lds_samples;        // largest n_samples of a unit with slots
  int32_t fam;                // op family (HostPlan::short_fam): the smallest instantiation that holds the plan's ops
  double step;
  const double* pool;         // INTERP tables of the closing multipliers ((value, difference) pairs)
  int32_t pk, pad2;           // real float launches of family 0: the packed-fp32 build (WFK_SH_NO_PK=1: the double-arithmetic one)
  double t0, last, dlast, di0;   // family 6 (corrected carriers): the grid's t0, its overridden last sample and that sample's index
                                 // in the caller's full grid (-1: none), the slice offset i0 -- as doubles
};

struct KArgs {
  const DevChannel* channels;
  const DevPiece* pieces;
  const double* params;
  const double* pool;
  const int32_t* chunk_first;  // [n_channels * chunks_per_ch]
  const double* tlist;         // tlist mode only
  void* out;
  int64_t ch_stride;           // elements
  int64_t n;                   // samples per channel
  int64_t chunks_per_ch;
  int64_t n_chunks;            // chunks of this launch (all: n_channels * chunks_per_ch); the grid is rounded up to 8 * ceil(n_chunks / 8)
  int64_t chunk_base;          // first chunk of this launch (sub-range launches: wfk_plan_run_host's pipeline)
  int32_t tiles_per_chunk;     // workgroup tiles per workgroup
  int32_t accumulate;
  double t0, step, last;
  int32_t has_last, pad;
  int32_t lean_par, lean_ops;  // lean kernel: doubles of parameter block / units (128 doubles) of op state to reserve in LDS
  int32_t corr;                // plan holds carriers that need the grid-rounding correction (lean kernel variant)
  int32_t reseed;              // lean kernel: tiles between exact (libm) reseeds of the carried op state
  int32_t lean_fam;            // lean kernel family (HostPlan::lean_fam)
  int32_t wavepriv;            // one-sample-per-lane time-list builds: every block fits a wave's quarter of the parameter buffer
  int32_t mixed;               // mixed plan: the lean kernel skips the pieces without WFK_PF_LEAN, the general
                               // kernel skips the lean and the zero pieces (two launches, one output)
  int64_t i0;                  // wfk_grid.i0: sample j of the plan is sample i0 + j of the caller's full grid
};

#ifdef __cplusplus
#include <string>
#include <vector>
struct wfk_program;
struct wfk_grid;

struct HostPlan {
  bool tlist = false;
  int32_t n_channels = 0;
  int64_t n = 0;
  double t0 = 0, step = 0, last = 0;
  int32_t has_last = 0;
  int64_t i0 = 0;              // wfk_grid.i0: index of the plan's first sample in the caller's full grid
  int32_t ns = 0, tile = 0, tiles_per_chunk = 1;
  int64_t chunks_per_ch = 0;
  std::vector<DevChannel> channels;
  std::vector<DevPiece> pieces;
  std::vector<double> params, pool;
  std::vector<int32_t> chunk_first;
  std::vector<std::vector<int64_t>> member_idx;
  std::vector<uint8_t> channel_complex;
  int32_t n_fast = 0, n_direct = 0, n_fused = 0, n_generic = 0;
  int32_t n_corr = 0;          // fused ops carrying the grid-rounding correction
  bool lean = false;           // wave-per-workgroup fused kernel (see WFK_LEAN_*)
  int32_t lean_fam = 0;        // instantiation of that kernel the plan needs: 0 plain ops, 1 + closing ops (erf
                               // edges, shared envelopes), 2 + chirps -- a shape added to one family cannot move
                               // the code generation of the others
  bool mixed = false;          // lean pieces go to the lean kernel, the rest to the general kernel
  int32_t lean_tile = 0, lean_tiles_per_chunk = 1;   // mixed: the lean launch's own chunking
  int64_t lean_chunks_per_ch = 0;
  std::vector<int32_t> lean_chunk_first;
  // float outputs of the lean kernel: longer chunks and exact reseeds every WFK_LEAN_RESEED_F32 tiles (the
  // carried state is double whatever the output: its drift over 512 steps, 1e-11, is far below float's
  // resolution).  Same-box A/B on 256 x 1e7 fp32: 1.99 -> 1.88 ms, C3 0.211 -> 0.197 ms.
  int32_t f32_tiles_per_chunk = 0;
  int64_t f32_chunks_per_ch = 0;
  std::vector<int32_t> f32_chunk_first;
  int32_t lean_par = 0, lean_ops = 0;   // largest parameter block (doubles, rounded) / most state units (128 doubles) of a piece
  // short tier (WFK_SH_*): `params` then holds the compact records
  bool shortp = false;
  int32_t max_block_len = 0;       // longest parameter block of the plan (doubles)
  double foreign_frac = 0.0;       // short plans: fraction of the evaluated samples in pieces handed to the general kernel
  double mean_piece_len = 0.0;     // grid plans: mean length of the live member pieces in samples (the short-tier decision)
  bool short_gave_up = false;      // grid plan: the pieces are of AWG-rate length but the short tier could not take most of them
  bool grid_as_tlist = false;      // ... and the plan was therefore compiled on the grid's sample times as a time list (wfk_api.cpp)
  bool short_has_fmul = false;     // some short piece holds an op wfk_sample_short evaluates and fir_short does not (table / mollifier multipliers, chirps)
  int32_t short_fam = 0;           // instantiation of wfk_sample_short the plan needs: 0 carrier-envelope ops only, 1 + erf edges, chirps and
                                   // shared Gaussians, 2 + table / mollifier envelopes (closing multipliers, own-term ops), 4 + exponential / hyperbolic chirp
                                   // multipliers (3 is family 0 in packed fp32, picked at launch)
  bool short_corr = false;         // some short op carries the grid-rounding correction (family 6: wfk_short_dev.h short_op_corr)
  bool short_needs_corr = false;   // some carrier wanted the grid-rounding correction, which only the lean kernel has
  bool pool_real = false;          // `pool` holds tables the parameter blocks point into (INTERP / mollifier / SAMPLED)
  std::vector<ShortUnit> s_units;
  std::vector<uint32_t> s_slots;
  int32_t s_lds_samples = 0, s_units_per_chunk = 1;
};

// host compiler: flattened program + time axis -> device tables.  Returns 0 or a
// negative WFK_E* code with a message in err.
int wfk_compile(const wfk_program* prog, const wfk_grid* grid, const double* tlist,
                int64_t n, HostPlan& out, std::string& err);

int wfk_compile_geom(const wfk_program* prog, const wfk_grid* grid, int lane_stride, int ns,
                     HostPlan& out, std::string& err);
#define WFK_RETRY_STD 1      // (not an error: compile again another way)
// big batches: contiguous channel blocks compiled on `nthreads` host threads, plans concatenated; WFK_RETRY_STD when
// the plan is not of a shape this takes (compile it in one piece then)
int wfk_compile_blocks(const wfk_program* prog, const wfk_grid* grid, int nthreads, HostPlan& out, std::string& err);

// Sampler fused into the FIR transform at AWG rates (fir_short, wfk_fir_sampled.hip): the pieces of a pure
// short plan cut into per-half-window entry lists.  Window pair `pr` of a row starts at sample
// 2 pr hop - lead and is sampled in two halves of `half` samples; an entry is a run of <= WFK_SH_R
// contiguous samples of one piece inside one half.
struct ShortWin {        // one half of one pair of windows of one channel
  int64_t rec0;          // base of the records its entries refer to (units of 16 B)
  int64_t e0;            // first entry: `cnt` runs to evaluate, then `ccnt` runs to copy
  int32_t cnt, pad;      // pad: LDS layout of the half (1: one spare element per 16)
  int32_t ccnt, pad2;    // runs of pieces the short tier cannot take (mixed plans): the general kernel has
                         // written their samples to the chain's workspace, the half copies them in
};
#define WFK_PLAN_FOREIGN_ONLY 0x80000000u   // wfk_plan_launch flag (internal): a mixed short plan launches only its general-kernel part
#define WFK_CW_ENTRY(drec, o, len) ((uint32_t)(drec) | ((uint32_t)(o) << 16) | ((uint32_t)((len) - 1) << 28))
// -> wins[(c * npairs + pr) * 2 + h], entries; returns 0, or WFK_EINVAL with the reason in err
int wfk_chain_windows(const HostPlan& H, int64_t n, int64_t hop, int64_t lead, int64_t half, int64_t npairs,
                      std::vector<ShortWin>& wins, std::vector<uint32_t>& entries, std::string& err);

void wfk_internal_keep_mixed_short(bool on);                          // this thread's next compiles keep mixed short plans (the FIR chain's sampler)
void wfk_internal_tlist_ns(int ns);                                   // samples per lane of this thread's next time-list compiles (0: by size)
void wfk_internal_grid_times(const wfk_grid* g, double* out);         // out[g->n]: the grid's sample times, as NumPy forms them
// this thread's next plan compiles keep table / mollifier multipliers out of short pieces (the FIR chain's sampler plan)
void wfk_internal_no_short_fmul(bool on);

// kernels (wfk_kernels.hip)
int wfk_launch_sampler(const KArgs& a, int32_t n_channels, int out_kind, bool tlist, int ns,
                       bool lean, bool generic, bool direct, void* stream, std::string& err);
// (a.corr selects the lean kernel variant with the per-sample grid-rounding correction)
int wfk_launch_short(const SArgs& a, int out_kind, void* stream, std::string& err);
#endif
