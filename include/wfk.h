/*
 * wfk.h -- C-ABI of the MI355X waveform-sampling engine (libwfk_hip.so).
 *
 * This is the drop-in boundary for ONE hot path of feihoo87/waveforms:
 *   piecewise sum-of-products expression tree  ->  dense sample array
 * i.e. what the reference does inside
 *   calc_parts()            waveforms/_waveform.pyx:155-169   (+ _calc :134-152, _apply :130-131)
 *   Waveform._fill_parts()  waveforms/waveform.py:524-527
 *   Waveform.__call__()     waveforms/waveform.py:529-563
 *   WaveVStack.__call__()   waveforms/waveform.py:679-693
 *   Waveform.sample()       waveforms/waveform.py:173-207    (np.arange grid)
 *   predistort(ker=...)     waveforms/distortion.py:323-337  (FIR branch)
 *
 * Plain C: pointers and sizes only, no torch / HIP types.  Every function
 * returns 0 on success or a negative WFK_E* code; wfk_last_error() returns a
 * thread-local message.  The caller owns every host buffer and every opaque
 * handle (explicit *_destroy); the library owns device tables inside a plan.
 * wfk_plan_launch() and wfk_fir_apply() do no allocation and no host sync.
 */
#ifndef WFK_H
#define WFK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WFK_ABI_VERSION 2   /* 2: wfk_grid.i0 (time slices of a grid) */

/* error codes */
#define WFK_OK 0
#define WFK_EINVAL (-1)   /* malformed program / argument                      */
#define WFK_EUNSUP (-2)   /* primitive id without a device implementation      */
#define WFK_EHIP   (-3)   /* HIP runtime / rocFFT failure                      */
#define WFK_ENOMEM (-4)
#define WFK_ETIMEOUT (-5)  /* a chained device-side wait ran out of polls (IIR single pass)   */

/* Primitive ids == the reference registry (waveforms/_waveform.pyx:374-388).
 * args (in `pool`, at fc_arg_off) per id, evaluated at u = t - shift:
 *   1 LINEAR            ()                      u
 *   2 GAUSSIAN          (std_sq2)               exp(-(u/std_sq2)^2)                 pyx:294-295
 *   3 ERF               (std_sq2)               erf(u/std_sq2)                      pyx:303-304
 *   4 COS               (w)                     cos(w*u)                            pyx:307-308
 *   5 SINC              (bw)                    np.sinc(bw*u)                       pyx:311-312
 *   6 EXP               (alpha)                 exp(alpha*u)                        pyx:315-316
 *   7 INTERP            (start, stop, p0..pm-1) np.interp(u, linspace(start,stop,m), p)   pyx:319-320
 *   8 LINEARCHIRP       (f0, f1, T, phi0)       pyx:323-324
 *   9 EXPONENTIALCHIRP  (f0, alpha, phi0)       pyx:327-328
 *  10 HYPERBOLICCHIRP   (f0, k, phi0)           pyx:331-332
 *  11 COSH (w)  12 SINH (w)                     pyx:335-340
 *  13 DRAG              (t0, freq, width, delta, block_freq|NaN=None, phase)        pyx:343-356
 *  14 MOLLIFIER         (r, d)                  pyx:359-371
 *  15 D_GAUSSIAN        (std_sq2, n)            pyx:298-300
 *  16 DRAG_SIN / 17 DRAG_SINX  multi-notch DRAG (waveforms/multy_drag.py:158-213), args in
 *     COMPILED form (the per-pulse matrices folded into coefficient tables on the host,
 *     waveforms_amd/multy_drag.py:device_args):
 *       (t0, freq, width, delta, phase, plateau, tab_half_width, m, dq,
 *        Px[0..m], Py[0..m], Cx, Cy [, QLx, QLy, QRx, QRy (dq+1 each, highest power first)])
 * 1000 SAMPLED (i0, v[0..m-1]): the factor's value at sample index j is v[j - i0] (j - i0
 *     clamped into [0, m)); `shift` is ignored.  This is how a primitive that only the CALLER
 *     can evaluate reaches the device: the caller evaluates it once per distinct factor on the
 *     plan's own time axis, over the samples of the piece (i0 = first sample of the piece), exactly
 *     where the reference calls function_lib[id](x[start:stop] - shift, *args)
 *     (_apply, waveforms/_waveform.pyx:130-131) -- Python callables registered with function() /
 *     registerBaseFunc (waveform.py:1470-1478, _waveform.pyx:264-271) and function_lib= overrides
 *     (waveform.py:178,535,679).
 * Any other id -> WFK_EUNSUP.                                                     */
enum {
  WFK_LINEAR = 1, WFK_GAUSSIAN = 2, WFK_ERF = 3, WFK_COS = 4, WFK_SINC = 5,
  WFK_EXP = 6, WFK_INTERP = 7, WFK_LINEARCHIRP = 8, WFK_EXPONENTIALCHIRP = 9,
  WFK_HYPERBOLICCHIRP = 10, WFK_COSH = 11, WFK_SINH = 12, WFK_DRAG = 13,
  WFK_MOLLIFIER = 14, WFK_D_GAUSSIAN = 15, WFK_DRAG_SIN = 16, WFK_DRAG_SINX = 17,
  WFK_SAMPLED = 1000
};

/*
 * Flattened expression forest (struct-of-arrays, depth-first in the order of
 * the reference's own flattener Waveform._tolist, waveforms/waveform.py:259-276):
 *
 *   channel  = one output row.  A `Waveform` is a channel with ONE member and
 *              its min/max as clip; a `WaveVStack` is a channel with one member
 *              per wlist entry, `offset`, `shift`, and clip = -inf/+inf.
 *   member   = (bounds, seq): pieces in ascending bound order, last bound +inf.
 *              piece i is live for bound[i-1] <= t - tshift < bound[i]
 *              (np.searchsorted side='left', _waveform.pyx:156).
 *   piece    = sum of terms;  term = amp * prod_j factor_j(t - tshift - shift_j)^power_j
 */
typedef struct wfk_program {
  int32_t n_channels, n_members, n_pieces, n_terms, n_factors;
  int64_t n_pool;
  const int32_t* ch_member_off;  /* [n_channels+1]                               */
  const double*  ch_offset;      /* [n_channels]  WaveVStack.offset (real part)  */
  const double*  ch_tshift;      /* [n_channels]  WaveVStack.shift               */
  const double*  ch_clip_lo;     /* [n_channels]  Waveform.min                   */
  const double*  ch_clip_hi;     /* [n_channels]  Waveform.max                   */
  const int32_t* mb_piece_off;   /* [n_members+1]                                */
  const double*  pc_bound;       /* [n_pieces]    upper bound in seconds         */
  const int32_t* pc_term_off;    /* [n_pieces+1]                                 */
  const double*  tm_amp_re;      /* [n_terms]                                    */
  const double*  tm_amp_im;      /* [n_terms]                                    */
  const int32_t* tm_factor_off;  /* [n_terms+1]                                  */
  const int32_t* fc_type;        /* [n_factors]   primitive id                   */
  const double*  fc_power;       /* [n_factors]                                  */
  const double*  fc_shift;       /* [n_factors]                                  */
  const int64_t* fc_arg_off;     /* [n_factors+1] into pool                      */
  const double*  pool;           /* [n_pool]                                     */
} wfk_program;

/* Uniform time grid, bit-identical to NumPy's:
 *   t[i] = fl(fl(i*step) + t0)           np.linspace / np.arange element formula
 *   t[n-1] = last  if has_last            np.linspace(endpoint=True) override
 * (replaces the `x` argument of Waveform.__call__ and the
 *  np.arange of Waveform.sample, waveforms/waveform.py:190,232).               */
typedef struct wfk_grid {
  double t0, step;
  int64_t n;
  int32_t has_last;
  double last;
  int64_t i0;      /* index of this grid's first sample in the caller's FULL grid: t[i] = fl(fl((i0 + i)*step) + t0).
                    * 0 for a whole grid; > 0 for a time slice of one (time-axis sharding, chunked sampling):
                    * the slice's samples and piece indices are then bit-identical to the same samples of the
                    * whole grid.  has_last refers to the slice's own last sample (set it only where the slice
                    * ends with the full grid's overridden last sample).                                     */
} wfk_grid;

enum { WFK_OUT_F64 = 0, WFK_OUT_F32 = 1, WFK_OUT_C128 = 2, WFK_OUT_C64 = 3 };

/* launch flags */
#define WFK_ACCUMULATE 1u   /* out += samples  (Waveform.__call__(accumulate=True)) */

typedef struct wfk_plan wfk_plan;
typedef struct wfk_fir_plan wfk_fir_plan;

typedef struct wfk_plan_info {
  int32_t n_channels;
  int64_t n;             /* samples per channel                                  */
  int32_t tile;          /* samples per workgroup tile                           */
  int64_t n_tiles;       /* workgroups per launch                                */
  int32_t n_pieces;      /* merged device pieces                                 */
  int64_t param_doubles; /* device parameter stream length                      */
  int32_t n_fast;        /* factor uses on a recurrence / table fast path        */
  int32_t n_direct;      /* factor uses evaluated with device libm               */
  int32_t n_fused;       /* terms absorbed into fused carrier-envelope ops       */
  int32_t n_generic;     /* terms evaluated factor by factor                     */
} wfk_plan_info;

int         wfk_abi_version(void);
const char* wfk_last_error(void);
int         wfk_device_count(int* count);
int         wfk_set_device(int ordinal);

/* -- sampler plans ------------------------------------------------------- */
/* grid mode: replaces calc_parts(bounds, seq, x=grid, ...) for every channel   */
int wfk_plan_create_grid(const wfk_program* prog, const wfk_grid* grid,
                         wfk_plan** out);
/* tlist mode: x is an explicit sorted float64 array on the host (it is
 * uploaded); replaces Waveform.__call__(x) for arbitrary x                     */
int wfk_plan_create_tlist(const wfk_program* prog, const double* t_host,
                          int64_t n, wfk_plan** out);
/* 1 and *grid filled if the sorted host array t_host[0..n) is BIT-IDENTICAL to a uniform NumPy
 * grid (np.linspace with or without endpoint, np.arange: every element is compared), else 0.
 * Lets Waveform.__call__(x) (waveform.py:529-563) compile grid mode for the usual x.     */
int wfk_grid_detect(const double* t_host, int64_t n, wfk_grid* grid);
/* The same for an x that is SEVERAL such grids back to back (np.concatenate of the chunks of a chunked job,
 * waveforms/waveform.py:232; a sequence sampled at two rates): number of runs found -- starts[k] = index of
 * run k's first sample, grids[k] its grid, every element verified -- or 0 (a run shorter than min_len,
 * more than max_runs runs, anything that is not exactly a grid).  The caller then samples run by run in grid
 * mode instead of handing the whole x over as a time list.                                                */
int wfk_grid_detect_runs(const double* t_host, int64_t n, int64_t min_len, int32_t max_runs,
                         int64_t* starts, wfk_grid* grids);
int wfk_plan_destroy(wfk_plan* plan);
int wfk_plan_get_info(const wfk_plan* plan, wfk_plan_info* info);
/* np.searchsorted(x - tshift, bounds) of one member (integer parity probe);
 * writes min(cap, #bounds) values, returns #bounds                            */
int wfk_plan_member_index(const wfk_plan* plan, int32_t member, int64_t* idx,
                          int32_t cap);
/* 1 if a live piece of `channel` has a complex amplitude (dtype detection of
 * calc_parts, _waveform.pyx:164-166)                                          */
int wfk_plan_channel_is_complex(const wfk_plan* plan, int32_t channel);
/* Name of the device kernel a launch of this plan with `out_kind` selects (the symbol
 * rocprofv3 --kernel-trace shows, template arguments included), for reports.           */
const char* wfk_plan_kernel_name(const wfk_plan* plan, int out_kind);
/* Bytes of device tables a launch of this plan reads (channels, pieces / units, parameter records,
 * slots, pool): at AWG sample rates (tens of samples per piece) they are a real share of the HBM
 * traffic next to the output stream, and bench.py reports them beside the algorithmic bytes.      */
int64_t wfk_plan_table_bytes(const wfk_plan* plan);
/* Evaluate every channel into out_dev[ch*ch_stride + i] (elements of out_kind).
 * Asynchronous on `hip_stream` (a hipStream_t, or NULL for the null stream).   */
int wfk_plan_launch(wfk_plan* plan, void* out_dev, int64_t ch_stride,
                    int out_kind, uint32_t flags, void* hip_stream);
/* launch into an internal device buffer, copy to out_host, synchronise         */
int wfk_plan_run_host(wfk_plan* plan, void* out_host, int64_t ch_stride,
                      int out_kind);

/* -- FIR stage: out[i] = sum_k ker[k] * sig[i + K/2 - k], zero padded ------- */
/* replaces predistort(sig, ker=ker), waveforms/distortion.py:329-337          */
int wfk_fir_plan_create(const double* ker_host, int32_t K, int64_t n,
                        int32_t batch, int kind /* WFK_OUT_F64|F32 */,
                        wfk_fir_plan** out);
/* the same with one kernel PER ROW: kers_host[row * K + k] (every AWG line has its own predistortion kernel;
 * upstream the call is per signal anyway).  On-chip transform only: K <= 6148, batch <= 65535.          */
int wfk_fir_plan_create_rows(const double* kers_host, int32_t K, int64_t n,
                             int32_t batch, int kind /* WFK_OUT_F64|F32 */,
                             wfk_fir_plan** out);
int wfk_fir_apply(wfk_fir_plan* plan, const void* in_dev, int64_t in_stride,
                  void* out_dev, int64_t out_stride, void* hip_stream);
int wfk_fir_plan_destroy(wfk_fir_plan* plan);

/* -- sampler -> FIR chain (BASELINE configs[3]) --------------------------------- */
/* out = predistort(wav(t), ker=ker): Waveform.__call__ (waveforms/waveform.py:529-563) followed by
 * the FIR branch of predistort (waveforms/distortion.py:329-337), for every channel of `prog` on
 * `grid`.  When every piece of the program is fully fused (carrier-envelope ops only, no clip, real
 * amplitudes) and the kernel fits one on-chip transform (K <= 1537) the FIR workgroups EVALUATE
 * their input windows instead of loading them: the samples never touch HBM and the chain moves the
 * 8 (4) B/sample of the filtered output only.  At AWG sample rates (plans in the short geometry) the
 * windows are sampled the short tier's way (fir_short); pieces without a short form travel through a
 * sparsely written workspace.  Otherwise the plan owns a workspace and runs
 * sampler -> workspace -> FIR (wfk_chain_is_fused() == 0, wfk_chain_unfused_reason() says why).
 * wfk_chain_launch() allocates nothing and does not synchronise.                              */
typedef struct wfk_chain_plan wfk_chain_plan;
int wfk_chain_plan_create(const wfk_program* prog, const wfk_grid* grid, const double* ker_host,
                          int32_t K, int kind /* WFK_OUT_F64|F32 */, wfk_chain_plan** out);
/* one kernel per channel: kers_host[channel * K + k] */
int wfk_chain_plan_create_rows(const wfk_program* prog, const wfk_grid* grid, const double* kers_host,
                               int32_t K, int kind /* WFK_OUT_F64|F32 */, wfk_chain_plan** out);
int wfk_chain_is_fused(const wfk_chain_plan* plan);
/* "fir_sampled<T,HOPB>" (stride-256 chains, fine grids), "fir_short<T,HOPB>" (contiguous lane runs, AWG
 * rates) or the two kernels of the unfused path; the string lives until the next call on this thread */
const char* wfk_chain_kernel_name(const wfk_chain_plan* plan);
/* bytes of device tables one launch reads besides the kernel spectrum (records, window entries) */
int64_t wfk_chain_table_bytes(const wfk_chain_plan* plan);
const char* wfk_chain_unfused_reason(const wfk_chain_plan* plan);
int wfk_chain_launch(wfk_chain_plan* plan, void* out_dev, int64_t out_stride, void* hip_stream);
int wfk_chain_plan_destroy(wfk_chain_plan* plan);

/* -- IIR stage ---------------------------------------- */
/* y = cascade of direct-form-II-transposed sections along each row, i.e.
 *   scipy.signal.sosfilt (n_sections sections of order 2)  -- Waveform.sample(filters=),
 *                                                             waveforms/waveform.py:193-203,244-251
 *   scipy.signal.lfilter (ONE section of order max(len(a),len(b))-1) -- predistort(filters=),
 *                                                             waveforms/distortion.py:298-321
 * orders[s] = order of section s; b and a hold the sections back to back, orders[s]+1
 * coefficients each (a[0] of a section need not be 1).  State layout == scipy's zi:
 * sections back to back, orders[s] values each (total wfk_iir_state_dim()).
 * apply: out = F(in - initial) + initial; zi_dev/zf_dev: optional [batch][state_dim]
 * device arrays (initial state in, final state out; NULL = zeros / not wanted).
 * Execution form, chosen per plan: equal-order cascades of state dimension <= 4 (one or two biquads, up to
 * four first-order sections, one section of order 3 or 4) run as ONE kernel that reads x once (chained scan with decoupled
 * look-back); other shapes as a block scan in three launches; cascades of mixed orders or beyond
 * those sizes as consecutive passes.  A plan owns scratch state for its launches: use one plan per
 * concurrent stream.  In place (out_dev == in_dev) is allowed in every form.           */
typedef struct wfk_iir_plan wfk_iir_plan;
int wfk_iir_plan_create(int32_t n_sections, const int32_t* orders, const double* b,
                        const double* a, int64_t n, int32_t batch,
                        int kind /* WFK_OUT_F64|F32 */, wfk_iir_plan** out);
int wfk_iir_state_dim(const wfk_iir_plan* plan);
int wfk_iir_apply(wfk_iir_plan* plan, const void* in_dev, int64_t in_stride, void* out_dev,
                  int64_t out_stride, const double* zi_dev, double* zf_dev, double initial,
                  void* hip_stream);
/* Synchronises `hip_stream`; WFK_ETIMEOUT if a single-pass launch of this plan since the last check ran
 * out of look-back polls (its outputs then hold NaN -- never silently: the same condition also fails the
 * NEXT wfk_iir_apply of the plan).  The plan switches to the three-launch form, so launching again works. */
int wfk_iir_status(wfk_iir_plan* plan, void* hip_stream);
int wfk_iir_plan_destroy(wfk_iir_plan* plan);

/* -- sampler -> IIR (-> FIR) chain ------------------------------------------------------------ */
/* out = F(wav(t) - initial) + initial for every channel of `prog` on `grid`, F the cascade of wfk_iir_plan_create
 * (same section layout, same zi / zf state layout, [n_channels][state_dim]) -- Waveform.sample(filters=(sos, initial))
 * (waveforms/waveform.py:190-203, chunked with carried zi :244-251); with `ker_host` != NULL followed by the FIR of
 * wfk_fir_plan_create (ker_per_row != 0: kers_host[channel * K + k]) -- predistort(wav(t), filters, ker)
 * (waveforms/distortion.py:298-337).  Everything stays on the device.  When the program is fully fused (carrier-envelope
 * ops only, real amplitudes, no clip) and the cascade's first pass is in the single-pass form (state dimension <= 4),
 * the wave that owns a 2048-sample chunk of the scan EVALUATES its input instead of loading it (iir_sampled): the
 * unfiltered samples never touch HBM and the pass moves the 8 (4) B/sample of its output only.  Otherwise
 * sampler -> (in place) IIR.  The FIR stage reads the filtered rows from a workspace the plan owns.
 * wfk_chain_iir_launch() allocates nothing and does not synchronise; status / timeout semantics as wfk_iir_status
 * (after a timeout the plan runs unfused in the three-launch form).                                            */
typedef struct wfk_chain_iir_plan wfk_chain_iir_plan;
int wfk_chain_iir_plan_create(const wfk_program* prog, const wfk_grid* grid, int32_t n_sections,
                              const int32_t* orders, const double* b, const double* a,
                              const double* ker_host /* or NULL */, int32_t K, int32_t ker_per_row,
                              int kind /* WFK_OUT_F64|F32 */, wfk_chain_iir_plan** out);
int wfk_chain_iir_is_fused(const wfk_chain_iir_plan* plan);
const char* wfk_chain_iir_unfused_reason(const wfk_chain_iir_plan* plan);
/* "iir_sampled<T,NSEC,ORD,PLAIN>" (+ later passes / FIR) or the kernels of the unfused path */
const char* wfk_chain_iir_kernel_name(const wfk_chain_iir_plan* plan);
int64_t wfk_chain_iir_table_bytes(const wfk_chain_iir_plan* plan);
int wfk_chain_iir_state_dim(const wfk_chain_iir_plan* plan);
int wfk_chain_iir_launch(wfk_chain_iir_plan* plan, void* out_dev, int64_t out_stride, const double* zi_dev,
                         double* zf_dev, double initial, void* hip_stream);
int wfk_chain_iir_status(wfk_chain_iir_plan* plan, void* hip_stream);
int wfk_chain_iir_plan_destroy(wfk_chain_iir_plan* plan);

/* -- whole-signal transfer function ------------------- */
/* out = irfft(rfft(in) * H) per row; rows contiguous (stride n); H_dev = n/2+1 complex128
 * bins on the device (f_k = k*fs/n).  Replaces the scipy.fftpack calls of
 * reflection / correct_reflection (waveforms/distortion.py:208-223).                  */
typedef struct wfk_spectral_plan wfk_spectral_plan;
int wfk_spectral_plan_create(int64_t n, int32_t batch, int kind, wfk_spectral_plan** out);
int wfk_spectral_apply(wfk_spectral_plan* plan, const void* in_dev, void* out_dev,
                       const void* H_dev, void* hip_stream);
int wfk_spectral_plan_destroy(wfk_spectral_plan* plan);

/* -- pinned host blocks for results ------------------------------------------------------- */
/* Page-locked host memory from a per-process cache (power-of-two blocks, parked on free).  A result
 * buffer taken from here costs no page faults, takes the D2H DMA directly and lets wfk_plan_run_host
 * overlap the copy of one part of a big single-channel result with the kernel of the next -- what the
 * drop-in calls Waveform.__call__ / Waveform.sample (waveforms/waveform.py:529-563, 173-207) need at
 * 1e7 points.  wfk_plan_run_host accepts any host pointer; only blocks from here get the pipeline.  */
int wfk_host_alloc(void** host_ptr, size_t bytes);
int wfk_host_free(void* host_ptr);

/* 1 if every one of the n 64-bit floats at `host` is finite, 0 if one is NaN or +-inf (a few host threads;
 * ~1 ms for 80 MB).  `Waveform.__call__(x, out=buf)` zeroes the caller's array with `out *= 0`
 * (waveforms/waveform.py:551), which KEEPS NaN / inf: the drop-in call asks this before it lets
 * wfk_plan_run_host overwrite `buf`, and follows the reference's two passes when the answer is 0.   */
int wfk_host_all_finite(const double* host, int64_t n);

/* -- device memory helpers for FFI callers without a HIP binding ---------- */
int wfk_malloc(void** dev_ptr, size_t bytes);
int wfk_free(void* dev_ptr);
int wfk_memcpy_h2d(void* dst_dev, const void* src_host, size_t bytes);
int wfk_memcpy_d2h(void* dst_host, const void* src_dev, size_t bytes);
int wfk_memset(void* dst_dev, int byte, size_t bytes);
int wfk_stream_sync(void* hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* WFK_H */
