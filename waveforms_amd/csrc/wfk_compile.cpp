// wfk_compile.cpp -- host compiler: flattened expression forest (include/wfk.h)
// + time axis -> device tables (wfk_internal.h).  Pure host C++, no HIP calls.
//
// What it decides, all of it O(#pieces), never per sample:
//  * integer piece membership: np.searchsorted(x - tshift, bounds, 'left') per
//    member (reference: waveforms/_waveform.pyx:156), evaluated on the EXACT grid
//    formula t[i] = fl(fl(i*step)+t0) so indices are bit-identical to NumPy's
//  * WaveVStack members (waveforms/waveform.py:690-692) merged into disjoint
//    pieces per channel, so each output sample is written exactly once
//  * per factor: uniform-grid fast path (phasor table / recurrences) or direct
//    device-libm evaluation
//  * parameter blocks sized for the LDS staging buffer
//
// Compiled with -ffp-contract=off: the grid formula must round twice.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <map>
#include <string>
#include <thread>
#include <vector>

#include "wfk.h"
#include "wfk_internal.h"

namespace {

struct TimeAxis {
  const wfk_grid* g;
  const double* t;
  int64_t n;
  double at(int64_t i) const {
    if (t) return t[i];
    if (g->has_last && i == g->n - 1) return g->last;
    volatile double m = (double)(i + g->i0) * g->step;
    return m + g->t0;
  }
  // first i with fl(x[i] - tshift) >= b
  int64_t search_left(double tshift, double b) const {
    int64_t lo = 0, hi = n;
    while (lo < hi) {
      int64_t mid = lo + (hi - lo) / 2;
      double x = at(mid);
      if (tshift != 0.0) x = x - tshift;
      if (x < b) lo = mid + 1; else hi = mid;
    }
    return lo;
  }
};

int argc_of(int type) {
  switch (type) {
    case WFK_LINEAR: return 0;
    case WFK_GAUSSIAN: case WFK_ERF: case WFK_COS: case WFK_SINC: case WFK_EXP:
    case WFK_COSH: case WFK_SINH: return 1;
    case WFK_MOLLIFIER: case WFK_D_GAUSSIAN: return 2;
    case WFK_EXPONENTIALCHIRP: case WFK_HYPERBOLICCHIRP: return 3;
    case WFK_LINEARCHIRP: return 4;
    case WFK_DRAG: return 6;
    case WFK_INTERP: return -1;
    case WFK_DRAG_SIN: case WFK_DRAG_SINX: return -3;
    case WFK_SAMPLED: return -4;
    default: return -2;
  }
}

// MOLLIFIER d>0 polynomial, highest degree first (reference: pyx:365-368)
std::vector<double> mollifier_poly(int d) {
  std::vector<double> p = {0.0, -2.0};  // p[k] = coeff of x^k
  for (int n = 1; n < d; ++n) {
    std::vector<double> nx(p.size() + 3, 0.0);
    for (size_t k = 1; k < p.size(); ++k) {
      double c = (double)k * p[k];
      nx[k - 1 + 4] += c;
      nx[k - 1 + 2] += -2.0 * c;
      nx[k - 1] += c;
    }
    for (size_t k = 0; k < p.size(); ++k) {
      nx[k + 3] += -4.0 * n * p[k];
      nx[k + 1] += (4.0 * n - 2.0) * p[k];
    }
    p.swap(nx);
  }
  std::reverse(p.begin(), p.end());
  return p;
}

// one fused carrier-envelope group (see WFK_FCE_* in wfk_internal.h)
struct FceGroup {
  double W = 0, sref = 0;
  long double psi_ref = 0;
  bool has_env = false, env32 = false, has_lin = false;
  bool has_exp = false;         // the envelope is exp(sigma * (t - sg)) (sigma = rate, sg = reference time), not a Gaussian
  double sigma = 0, sg = 0, slin = 0;
  long double A[4] = {0, 0, 0, 0}, B[4] = {0, 0, 0, 0};
  int deg = 0, nterms = 0;
  bool imag = false;            // the group adds to the IMAGINARY part of the output
  bool envmul = false;          // pseudo-op: multiply the accumulators by the shared Gaussian envelope
  bool erfmul = false;          // pseudo-op: multiply them by m0 + m1 erf((t - sg) / sigma) (flat-top edge)
  double m0 = 0, m1 = 1;
  int bank = 0;                 // first of a run of `bank` bare carriers (WFK_FCE_BANK)
  int fmul = 0;                 // pseudo-op: multiply them by a stateless function of t - slin: 2 = a finite INTERP table read as a
                                // continuous piecewise-linear function, 3 = mollifier(r) (lean kernel family 3 only)
  int32_t fmul_f = -1;          // ... the program factor it stands for
  int own_kind = 0;             // lean kernel: this plain carrier group carries an envelope of its own (2 table, 3 mollifier) ...
  int32_t own_f = -1;           // ... the program factor of it (WFK_FCE_OWNMUL)
  bool fmul_own = false;        // short tier: the multiplier belongs to the ONE group in front of it (envelope x carrier as one op)
  long double K = 0;            // chirp: the phase is K u^2 + W u - psi_ref, u = t' - corg (W, psi_ref as for a plain carrier)
  long double corg = 0;         // chirp: the origin its phase polynomial is expanded about (the chirp's own shift: about t' = 0
                                // the three terms are 1e10 rad each 3 ms from t = 0 and cancel to 80-bit rounding, 5e-9 rad)
  long double Wl = 0;           // chirp: W before its rounding to double (|W| ~ 2 K |shift|: 2^-53 of it times t' shows in the phase)
  bool chirp = false;
  double tref = 0;              // chirp: reference time inside the piece (the device works in t' - tref)
  bool corr = false;            // carrier needs the per-sample rounding correction (WFK_FCE_PACK bit 7)
  double wm = 0, sm = 0;        // corr: the reference COS factor (w, shift) whose rounded phase fl(w*fl(x-shift)) is mimicked
  double tl_thmax = INFINITY;   // time-list plans: largest |W (t' - s_ref)| over the piece (cheap phase reduction below 1.6e6)
};

struct BlockBuilder {
  std::vector<double> body;     // after the 2-double header
  std::vector<double> tables;   // appended behind the records
  std::vector<std::pair<size_t, int>> table_refs;  // (index of aux slot in body, table id)
  std::vector<size_t> fce_ats;                      // body index of every fused-op record (packed word fix-up)
  std::map<double, int> table_of_w;               // dedupe COS tables by dphase
  int n_terms = 0;                                // ops in this block
  int state_units = 0;                            // lean kernel: per-lane state handed out so far (128 doubles each)
  size_t size() const { return WFK_BLK_HDR + body.size() + tables.size() + 1; }
};

}  // namespace

static thread_local bool g_no_short_corr = false;   // second attempt at a short plan whose corrected carriers met ops family 6 does not hold
static thread_local bool g_no_short_fmul = false;   // the FIR chain's sampler plan: fir_short has no table / mollifier multipliers, such pieces stay with the general kernel
void wfk_internal_no_short_fmul(bool on) { g_no_short_fmul = on; }
static thread_local bool g_no_chirp = false;   // second compile of a plan that mixes corrected carriers and chirps
static thread_local bool g_keep_mixed_short = false;   // the FIR chain's sampler plan: fir_short fuses the short pieces, the rest is copied in
void wfk_internal_keep_mixed_short(bool on) { g_keep_mixed_short = on; }
static thread_local int g_tlist_ns = 0;        // samples per lane of this thread's next time-list compiles (0: by size)
// wfk_compile_blocks: this thread compiles a BLOCK of the channels of a bigger job -- the program's channel arrays are
// views into the job's (ch_member_off does not start at 0), the job was validated once by the caller, and the
// launch geometry (tiles per chunk) is decided on the job's channel count, so that the blocks' tables concatenate
static thread_local int32_t g_block_total_channels = 0;
void wfk_internal_tlist_ns(int ns) { g_tlist_ns = ns; }
// the sample times of a grid, as NumPy forms them (this file is built -ffp-contract=off)
void wfk_internal_grid_times(const wfk_grid* g, double* out) {
  const TimeAxis ax{g, nullptr, g->n};
  for (int64_t i = 0; i < g->n; ++i) out[i] = ax.at(i);
}

// want_short: -1 = decide from the mean live piece length (grid plans), 0 = never.  Returns
// WFK_RETRY_STD when the short geometry was chosen but some piece cannot run in it.
// a vector of at most N elements on the stack (the fusion pass runs per term: heap traffic there was a third of a compile)
template <class T, int N>
struct SmallVec {
  alignas(T) unsigned char raw[N * sizeof(T)];     // (no element is constructed until it is pushed)
  int n = 0;
  SmallVec() {}
  SmallVec(const SmallVec& o) : n(o.n) { std::memcpy(raw, o.raw, (size_t)o.n * sizeof(T)); }
  SmallVec& operator=(const SmallVec& o) { n = o.n; std::memcpy(raw, o.raw, (size_t)o.n * sizeof(T)); return *this; }
  T* data() { return reinterpret_cast<T*>(raw); }
  const T* data() const { return reinterpret_cast<const T*>(raw); }
  bool push_back(const T& x) { if (n >= N) return false; data()[n++] = x; return true; }
  size_t size() const { return (size_t)n; }
  bool empty() const { return n == 0; }
  T* begin() { return data(); }
  T* end() { return data() + n; }
  const T* begin() const { return data(); }
  const T* end() const { return data() + n; }
  T& operator[](size_t i) { return data()[i]; }
  const T& operator[](size_t i) const { return data()[i]; }
  void swap(SmallVec& o) { SmallVec t(*this); *this = o; o = t; }
};

// WFK_TIMING=1: where a compile spends its time, phase by phase, on stderr (tools/plan_build_bench.py)
struct PhaseTimer {
  bool on;
  std::chrono::steady_clock::time_point t0;
  PhaseTimer() : on(std::getenv("WFK_TIMING") != nullptr), t0(std::chrono::steady_clock::now()) {}
  void mark(const char* what) {
    if (!on) return;
    const auto t1 = std::chrono::steady_clock::now();
    std::fprintf(stderr, "wfk_compile %-14s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
    t0 = t1;
  }
  ~PhaseTimer() { mark("rest"); }
};

static int compile_impl(const wfk_program* P, const wfk_grid* grid, const double* tlist,
                        int64_t n_tlist, HostPlan& H, std::string& err, bool allow_corr,
                        int lane_stride = 64, int ns_override = 0, int want_short = 0);

// Grid plan for another evaluation geometry: lanes `lane_stride` samples apart, `ns` samples per
// lane (the sampler fused into the FIR transform walks a window with stride 256, wfk_fir_sampled.hip).
// Only the piece / parameter tables are meaningful in the result; H.lean tells whether every piece
// is one block of fused ops (the only form that kernel evaluates).
int wfk_compile_geom(const wfk_program* P, const wfk_grid* grid, int lane_stride, int ns, HostPlan& H,
                     std::string& err) {
  return compile_impl(P, grid, nullptr, 0, H, err, false, lane_stride, ns);
}

int wfk_compile(const wfk_program* P, const wfk_grid* grid, const double* tlist,
                int64_t n_tlist, HostPlan& H, std::string& err) {
  // Carriers whose phase is sensitive to NumPy's grid rounding (far from t = 0) stay fused with a
  // first-order per-sample correction, which only the lean kernel implements.  A plan that turns
  // out not to be lean is compiled again with such carriers on the exact (libm) path.
  // Plans whose live pieces are short (AWG sample rates: tens to hundreds of samples per pulse) are
  // compiled for the contiguous-lane geometry of wfk_short.hip first; a piece that tier cannot take
  // (generic terms, erf edges, corrected carriers) sends the whole plan back to the standard tiers.
  int rc = compile_impl(P, grid, tlist, n_tlist, H, err, true, 64, 0, -1);
  if (rc == WFK_OK && H.shortp && H.short_corr && H.short_fam != 6) {
    // corrected carriers next to closing ops / chirps / tables: no family of the short tier holds both -- the attempt
    // again without the tier's correction (those carriers' pieces then go the way they went before family 6)
    g_no_short_corr = true;
    H = HostPlan();
    rc = compile_impl(P, grid, tlist, n_tlist, H, err, true, 64, 0, -1);
    g_no_short_corr = false;
  }
  const bool gave_up = rc == WFK_RETRY_STD;
  const double mean_len = H.mean_piece_len;      // (of the first compile: the later ones do not take the short-tier decision)
  if (rc == WFK_RETRY_STD) rc = compile_impl(P, grid, tlist, n_tlist, H, err, true);
  else if (rc == WFK_OK && H.shortp && H.short_needs_corr) {
    // far from t = 0 fast carriers need the per-sample rounding correction, which only the lean kernel
    // has: where the standard tiers can run the plan lean (pieces long enough for its recurrences) they
    // win; otherwise the short tier keeps what it can take and libm serves those carriers either way
    HostPlan S;
    std::string e2;
    if (compile_impl(P, grid, tlist, n_tlist, S, e2, true) == WFK_OK && (S.lean || S.mixed)) H = std::move(S);
  }
  if (rc == WFK_OK && !H.shortp && H.n_corr > 0 && !H.lean && !H.mixed) rc = compile_impl(P, grid, tlist, n_tlist, H, err, false);
  if (rc == WFK_OK && !H.shortp && H.n_corr > 0 && H.lean_fam >= 2) {
    // corrected carriers (far from t = 0) and fused chirps in one plan: the lean kernel is not instantiated
    // for that combination; the chirps take the general path
    g_no_chirp = true;
    rc = compile_impl(P, grid, tlist, n_tlist, H, err, true);
    if (rc == WFK_OK && H.n_corr > 0 && !H.lean && !H.mixed) rc = compile_impl(P, grid, tlist, n_tlist, H, err, false);
    g_no_chirp = false;
  }
  // pieces of AWG-rate length that the short tier could not take: the standard tiers walk every piece over whole wave
  // tiles of 1024 samples (wfk_api.cpp: such grid plans are evaluated pointwise instead)
  // (not where the standard compile came out lean: chirp pulses stay on the lean kernel's chirp family, measured
  //  12.6 ms against 17.6 pointwise on 2048 x 1e5 at 2 GS/s)
  //  12.6 ms against 17.6 pointwise on 2048 x 1e5 at 2 GS/s; and only for pieces well below a wave tile: from a few
  //  hundred samples per piece on, the general kernel's per-factor fast paths cost less than pointwise libm)
  if (rc == WFK_OK) H.short_gave_up = gave_up && grid != nullptr && !H.lean && !H.mixed && mean_len > 0.0 && mean_len < 192.0;
  // A short plan that hands more than a few per cent of its samples on: on 2e8 samples the general kernel's launch over those
  // pieces costs ~1.25 ms per per cent (every piece over a wave tile) next to 0.4 ms for the short pieces; evaluated pointwise
  // the fused pieces take ~4.5 ms and the rest 0.34 ms per per cent (rocprofv3 --stats, one exponential chirp in ten
  // pulses: 12.9 -> 7.9 ms) -- break-even near 4.5 %
  if (rc == WFK_OK && grid && H.shortp && H.mixed && H.foreign_frac >= 0.05 && mean_len > 0.0 && mean_len < 192.0 &&
      !g_keep_mixed_short && !std::getenv("WFK_KEEP_MIXED_SHORT"))
    H.short_gave_up = true;
  return rc;
}

static int compile_impl(const wfk_program* P, const wfk_grid* grid, const double* tlist,
                        int64_t n_tlist, HostPlan& H, std::string& err, bool allow_corr,
                        int lane_stride, int ns_override, int want_short) {
  PhaseTimer ptimer;
  if (!P || (!grid && !tlist && n_tlist != 0)) { err = "null program or time axis"; return WFK_EINVAL; }
  if (P->n_channels < 0 || P->n_members < 0) { err = "negative counts"; return WFK_EINVAL; }
  TimeAxis ax{grid, tlist, grid ? grid->n : n_tlist};
  if (ax.n < 0) { err = "negative sample count"; return WFK_EINVAL; }
  H = HostPlan();
  H.tlist = grid == nullptr;
  H.n_channels = P->n_channels;
  H.n = ax.n;
  if (grid) {
    H.t0 = grid->t0; H.step = grid->step; H.last = grid->last; H.has_last = grid->has_last; H.i0 = grid->i0;
    if (grid->i0 < 0) { err = "negative grid.i0"; return WFK_EINVAL; }
    if (ax.n > 1 && !(grid->step > 0)) { err = "grid step must be positive"; return WFK_EINVAL; }
  }
  H.ns = H.tlist ? (g_tlist_ns > 0 ? g_tlist_ns : (ax.n < WFK_TLIST_SMALL_N ? WFK_NS_TLIST_SMALL : WFK_NS_TLIST)) : WFK_NS_GRID;
  if (ns_override > 0 && !H.tlist) H.ns = ns_override;
  H.tile = WFK_WG * H.ns;
  int NS = H.ns;
  double dstride = grid ? (double)lane_stride * grid->step : 0.0;  // time between a lane's samples
  bool shortm = false;          // the plan is compiled for the short tier (decided after the piece search)
  bool cur_short = false;       // ... and the piece being built uses its contiguous-lane geometry
  const int lean_par_cap = ns_override > 0 ? WFK_CHAIN_PAR : WFK_LEAN_PAR;
  // (experiment switches, read once per compile: a getenv per piece is a scan of the environment per piece)
  const bool env_no_sinc_tab = std::getenv("WFK_NO_SINC_TAB") != nullptr;
  const bool env_no_moll_rec = std::getenv("WFK_NO_MOLL_REC") != nullptr;
  const bool env_no_interp_grid = std::getenv("WFK_NO_INTERP_GRID") != nullptr;
  const bool env_no_interp_lin = std::getenv("WFK_NO_INTERP_LIN") != nullptr;
  const bool env_no_short_cmul = std::getenv("WFK_NO_SHORT_CMUL") != nullptr;
  const bool env_no_short_chirp = std::getenv("WFK_NO_SHORT_CHIRP") != nullptr;
  const bool env_no_short_multi = std::getenv("WFK_NO_SHORT_MULTI") != nullptr;
  const bool env_no_lean_multi = std::getenv("WFK_NO_LEAN_MULTI") != nullptr;
  const bool env_no_short_envmul = std::getenv("WFK_NO_SHORT_ENVMUL") != nullptr;
  const bool env_no_bank = std::getenv("WFK_NO_BANK") != nullptr;
  const char* const env_tlsmall_limit = std::getenv("WFK_TLSMALL_LIMIT");
  const bool env_no_short_xchirp = std::getenv("WFK_NO_SHORT_XCHIRP") != nullptr;
  const bool env_no_short_erftab = std::getenv("WFK_NO_SHORT_ERFTAB") != nullptr;
  const bool env_no_short_corr = std::getenv("WFK_NO_SHORT_CORR") != nullptr || g_no_short_corr;
  // validation / A-B switch: evaluate every factor with device libm even on a grid
  const char* nofast_env = std::getenv("WFK_DISABLE_FAST");
  const bool nofast = nofast_env && nofast_env[0] == '1';

  if (g_block_total_channels == 0) {
  // ---- validate structure ----------------------------------------------------
  auto offsets_ok = [](const auto* off, int64_t count, int64_t total) {
    if (off[0] != 0 || off[count] != total) return false;
    for (int64_t i = 0; i < count; ++i)
      if (off[i] > off[i + 1]) return false;
    return true;
  };
  if (P->n_pieces < 0 || P->n_terms < 0 || P->n_factors < 0 || P->n_pool < 0 ||
      !offsets_ok(P->ch_member_off, P->n_channels, P->n_members) ||
      !offsets_ok(P->mb_piece_off, P->n_members, P->n_pieces) ||
      !offsets_ok(P->pc_term_off, P->n_pieces, P->n_terms) ||
      !offsets_ok(P->tm_factor_off, P->n_terms, P->n_factors) ||
      !offsets_ok(P->fc_arg_off, P->n_factors, P->n_pool)) {
    err = "offset arrays must start at 0, be non-decreasing and end at the element counts";
    return WFK_EINVAL;
  }
  for (int32_t c = 0; c < P->n_channels; ++c) {
    if (P->ch_member_off[c] > P->ch_member_off[c + 1]) { err = "ch_member_off not monotone"; return WFK_EINVAL; }
    if (std::isnan(P->ch_clip_lo[c]) || std::isnan(P->ch_clip_hi[c])) { err = "NaN clip"; return WFK_EINVAL; }
  }
  for (int32_t m = 0; m < P->n_members; ++m) {
    int32_t a = P->mb_piece_off[m], b = P->mb_piece_off[m + 1];
    if (b <= a) { err = "member without pieces"; return WFK_EINVAL; }
    if (!(P->pc_bound[b - 1] == INFINITY)) { err = "last bound of a member must be +inf"; return WFK_EINVAL; }
    for (int32_t p = a + 1; p < b; ++p)
      if (!(P->pc_bound[p - 1] <= P->pc_bound[p])) { err = "bounds not ascending"; return WFK_EINVAL; }
  }
  for (int32_t f = 0; f < P->n_factors; ++f) {
    int want = argc_of(P->fc_type[f]);
    int64_t have = P->fc_arg_off[f + 1] - P->fc_arg_off[f];
    if (want == -2) {
      err = "primitive id " + std::to_string(P->fc_type[f]) + " has no device implementation";
      return WFK_EUNSUP;
    }
    if (want == -3) {   // compiled multi-notch DRAG: header + tables, lengths must agree
      const double* a = P->pool + P->fc_arg_off[f];
      bool ok = have >= 9;
      if (ok) {
        const double m = a[7], dq = a[8];
        ok = m >= 0 && m <= 64 && m == std::floor(m) && dq >= -1 && dq <= 256 && dq == std::floor(dq) &&
             have == 9 + 2 * ((int64_t)m + 1) + 2 + (dq >= 0 ? 4 * ((int64_t)dq + 1) : 0) &&
             (P->fc_type[f] == WFK_DRAG_SINX) == (dq >= 0);
      }
      if (!ok) { err = "malformed compiled DRAG_SIN/DRAG_SINX argument block"; return WFK_EINVAL; }
      continue;
    }
    if (want == -4) {   // caller-evaluated factor: (i0, values...)
      const double i0 = have >= 1 ? P->pool[P->fc_arg_off[f]] : NAN;
      if (!(have >= 1) || !(i0 >= 0) || i0 != std::floor(i0) || i0 > 9.0e15) {
        err = "malformed SAMPLED factor: needs (i0 >= 0, values...)"; return WFK_EINVAL;
      }
      continue;
    }
    if ((want >= 0 && have != want) || (want == -1 && have < 3)) {
      err = "wrong argument count for primitive id " + std::to_string(P->fc_type[f]);
      return WFK_EINVAL;
    }
    const double* a = P->pool + P->fc_arg_off[f];
    if ((P->fc_type[f] == WFK_MOLLIFIER && (a[1] < 0 || a[1] > 12 || a[1] != std::floor(a[1]))) ||
        (P->fc_type[f] == WFK_D_GAUSSIAN && (a[1] < 0 || a[1] > 64 || a[1] != std::floor(a[1])))) {
      err = "derivative order out of range"; return WFK_EINVAL;
    }
  }

  }
  ptimer.mark("validate");
  // ---- searchsorted per member ----------------------------------------------
  H.member_idx.resize(P->n_members);
  for (int32_t c = 0; c < P->n_channels; ++c)
    for (int32_t m = P->ch_member_off[c]; m < P->ch_member_off[c + 1]; ++m) {
      auto& idx = H.member_idx[m];
      for (int32_t p = P->mb_piece_off[m]; p < P->mb_piece_off[m + 1]; ++p)
        idx.push_back(ax.search_left(P->ch_tshift[c], P->pc_bound[p]));
    }

  ptimer.mark("searchsorted");
  // ---- geometry: short tier? ---------------------------------------------------------------------
  // Mean length of the live member pieces (samples).  Below WFK_SH_MAXLEN the plan is compiled for
  // the contiguous-lane geometry: lane stride = one sample, WFK_SH_R samples per lane.
  // (the tier's records and the fir_short window entries hold sample indices as 32-bit words: a row of
  // 2^31 samples or more stays on the standard tiers, which index in 64 bits)
  if (want_short != 0 && grid && !H.tlist && !nofast && ns_override == 0 && ax.n > 0 && ax.n < ((int64_t)1 << 31)) {
    const char* e = std::getenv("WFK_SHORT");          // 0: never, 1: whatever the piece length
    const int mode = e ? std::atoi(e) : -1;
    int64_t live = 0, live_samples = 0;
    for (int32_t m = 0; m < P->n_members; ++m) {
      const auto& idx = H.member_idx[m];
      int64_t prev = 0;
      for (size_t k = 0; k < idx.size(); ++k) {
        const int32_t p = P->mb_piece_off[m] + (int32_t)k;
        if (idx[k] > prev && P->pc_term_off[p + 1] > P->pc_term_off[p]) { ++live; live_samples += idx[k] - prev; }
        prev = std::max(prev, idx[k]);
      }
    }
    int64_t maxlen = 1536;                              // tools/short_crossover.py
    if (const char* m = std::getenv("WFK_SHORT_MAXLEN")) maxlen = std::atoll(m);
    if (mode != 0 && live > 0 && (mode == 1 || live_samples < maxlen * live)) {
      shortm = true;
    }
    H.mean_piece_len = live > 0 ? (double)live_samples / (double)live : 0.0;
  }
  // geometry of the piece being built: lanes one sample apart (short tier) or `lane_stride` apart
  auto set_geom = [&](bool sh) {
    cur_short = sh;
    NS = sh ? WFK_SH_R : H.ns;
    dstride = grid ? (sh ? 1.0 : (double)lane_stride) * grid->step : 0.0;
  };
  set_geom(shortm);

  // phasor table (C[k], S[k]) = (cos, sin)(k * dphase), k < NS, shared per block
  auto table_for = [&](BlockBuilder& B, double dphase) -> int {
    auto it = B.table_of_w.find(dphase);
    if (it != B.table_of_w.end()) return it->second;
    int table = (int)(B.tables.size() / (2 * (NS + 1)));
    B.table_of_w[dphase] = table;
    // entry NS advances a carried phasor by one tile.  One long-double sincos, then the
    // rotation recurrence in long double (64-bit mantissa: after NS = 16 steps the error is
    // ~1e-18, below half an ulp of the stored doubles) instead of 2 (NS + 1) libm calls per
    // table -- those were most of the plan-creation time of a 100-pulse channel.
    const long double c1 = cosl((long double)dphase), s1 = sinl((long double)dphase);
    long double c = 1.0L, sn = 0.0L;
    for (int k = 0; k <= NS; ++k) {
      B.tables.push_back((double)c);
      B.tables.push_back((double)sn);
      const long double cn = c * c1 - sn * s1;
      sn = sn * c1 + c * s1;
      c = cn;
    }
    return table;
  };

  // Gaussian recurrence validity over the samples [s0, s1) (+ NS strides of overhang)
  auto gauss_range = [&](double sigma, double shift, double tshift, int64_t s0, int64_t s1,
                         bool& f64_ok, bool& f32_ok) {
    double ua = (ax.at(s0) - tshift) - shift, ub = (ax.at(s1 - 1) - tshift) - shift;
    double umax = std::max(std::fabs(ua), std::fabs(ub));
    double Hs = dstride / std::fabs(sigma);
    double vext = umax / std::fabs(sigma) + NS * Hs;
    f64_ok = std::isfinite(sigma) && sigma != 0.0 && Hs <= 2.0 && vext <= 26.0;
    f32_ok = f64_ok && vext <= 8.0 && 2.0 * vext * Hs + Hs * Hs <= 80.0;
  };

  // The uniform-grid fast paths evaluate sample j0 + 64 k at t(j0) + k * 64 * step, the reference
  // at NumPy's rounded grid value fl(fl(i*step) + t0).  The two differ by up to ~one ulp of
  // |i*step| + |t_i|; a carrier turns that into a phase difference W * dt_jitter, a Gaussian into
  // (u/sigma^2) * dt_jitter.  Negligible for the BASELINE configs (1e-12), but a waveform sampled
  // 16 ms away from t = 0 with a 300 MHz carrier sees 3e-9.  Where the bound exceeds
  // WFK_JITTER_TOL the factor keeps the exact per-sample time (device libm) instead.
  constexpr double WFK_JITTER_TOL = 2.5e-10;
  // ... shared by the TERMS that meet in a piece: a factor may use the bound over the number of terms (a pulse as mixing()
  // makes it: 3; overlapping pulses, vstacks: more), so that the terms' errors -- grid jitter, and the phase rounding a
  // corrected group does not mimic for the terms that merely joined it -- cannot add up past the contract.  Near t = 0
  // the jitter is 1e-6 of any budget; tools/fuzz_soak.py awgfar / far -- the random scripts moved 10 us .. 10 ms from
  // t = 0 -- had 2 % of their cases at 1.0-2.5e-9 of peak with the flat bound (240-360 terms per channel, every
  // corrected group at 1-3e-10), 1 % with the bound over n / 3, none of 6000 above 7.4e-10 with this one.  The price is
  // paid where it applies: `also.far` (3-term pieces 1 ms out) 0.84 -> 0.95 ms -- more terms found groups of their own.
  double jtol = WFK_JITTER_TOL;
  auto grid_jitter = [&](int64_t s0, int64_t s1) -> double {
    if (s1 <= s0) return 0.0;
    if (!grid) {
      // time list: every sample is evaluated AT its own time; what separates a fused group from the
      // reference there is the rounding of the reference's own phase fl(w fl(t' - shift)) and of the group's
      // W (t' - s_ref): about an ulp of the phase, i.e. of |t'| times the rate
      const double m = std::max(std::fabs(ax.at(s0)), std::fabs(ax.at(s1 - 1)));
      return 2.4e-16 * m;
    }
    const double m = std::max(std::fabs(ax.at(s0)), std::fabs(ax.at(s1 - 1)));
    return 1.2e-16 * (m + std::fabs((double)(grid->i0 + s1 - 1) * grid->step));
  };
  auto rate_safe = [&](double rate, int64_t s0, int64_t s1) -> bool {   // |d value / dt| <= rate
    return std::fabs(rate) * grid_jitter(s0, s1) <= WFK_JITTER_TOL;
  };
  // (carriers: every term brings a phase of its own, their errors add -- the shared budget; an envelope is evaluated
  //  once for the terms under it: the flat bound above)
  auto rate_safe_n = [&](double rate, int64_t s0, int64_t s1) -> bool {
    return std::fabs(rate) * grid_jitter(s0, s1) <= jtol;
  };

  // A carrier beyond that bound can still run fused.  Two things separate the reference from the
  // ideal phasor there: (i) it evaluates AT NumPy's rounded grid time x_k, e_k = x_k - ideal_k
  // (about an ulp of |t|), and (ii) its phase is the ROUNDED product fl(w * fl(x_k - shift)),
  // rho_k = that minus the exact product (up to half an ulp of the phase: 2e-9 rad at 3e7 rad).
  // Both are reproduced per sample, to first order: cos(th + d) = cos th - d sin th with
  // d_k = W e_k + rho_k, rho_k recomputed for ONE reference COS factor per group (w_m, s_m): the
  // factor of the group's heaviest term with the largest phase.  Terms whose own factor differs
  // may join only if their weight times the phase noise stays inside the budget; what is left
  // after the correction is d^2 / 2.  (fl(x - s_m) is not exact when the carrier is referenced to
  // t = 0, as mixing()'s is: what the subtraction rounds away is recovered with a TwoSum.)
  const char* nocorr_env = std::getenv("WFK_DISABLE_CORR");
  // (short tier: family 6 evaluates corrected carriers of degree <= 1 on channels without a pending shift)
  const bool corr_enabled = allow_corr && (!shortm || !env_no_short_corr) && !H.tlist && !(nocorr_env && nocorr_env[0] == '1');
  bool piece_corr_ok = true;   // cleared for the second attempt at a piece that turned out not to be lean
  const char* nochirp_env = std::getenv("WFK_DISABLE_CHIRP");
  const bool chirp_base = !H.tlist && ns_override == 0 && !g_no_chirp && !(nochirp_env && nochirp_env[0] == '1');
  bool piece_chirp_ok = true;  // likewise: the fused chirp op exists in the lean kernel only
  bool piece_fuse_ok = true;   // time lists: cleared for the second attempt at a piece that kept a generic term (see below)
  bool chirp_ok = false;
  auto corr_safe = [&](double rate, int64_t s0, int64_t s1) -> bool {
    const double x = std::fabs(rate) * grid_jitter(s0, s1);
    return corr_enabled && piece_corr_ok && std::isfinite(x) && 2.0 * x * x <= 1e-11;   // (d <= ~2 W e)
  };

  // ---- factor record emission -------------------------------------------------
  // The reference evaluates every distinct factor of a piece once (_calc's cache,
  // _waveform.pyx:135-147); the generic tier evaluates term by term.  For the expensive case --
  // a direct (libm) factor shared by consecutive terms, e.g. the erf edge of a flat-top pulse
  // under ten multiplexed carriers -- the values the previous term left in the workgroup's LDS
  // value buffer are reused: the repeated record is marked (type + WFK_M_REUSE) when its
  // signature equals the last direct factor emitted into the same block.
  std::map<int64_t, int64_t> sampled_at; // SAMPLED tables already copied: program pool offset -> H.pool offset
  std::vector<double> last_direct;      // record of the last direct factor of the current block
  auto emit_factor = [&](BlockBuilder& B, int32_t f, double tshift, int64_t s0, int64_t s1) {
    const int type = P->fc_type[f];
    const double pw = P->fc_power[f], shift = P->fc_shift[f];
    const double* a = P->pool + P->fc_arg_off[f];
    const int64_t na = P->fc_arg_off[f + 1] - P->fc_arg_off[f];
    double rec[WFK_FREC] = {(double)type, pw, shift, 0, 0, 0, 0, 0, 0, 0};
    int table = -1;
    bool fast = false;
    if (!H.tlist && !nofast && pw == 1.0 && s1 > s0) {
      // time range of the piece on this channel's shifted axis
      double ua = (ax.at(s0) - tshift) - shift, ub = (ax.at(s1 - 1) - tshift) - shift;
      double umax = std::max(std::fabs(ua), std::fabs(ub));
      if (type == WFK_LINEAR && (umax > 0.0 ? rate_safe(1.0 / umax, s0, s1) : grid_jitter(s0, s1) == 0.0)) {
        // (u itself: the jitter must be negligible against the largest |u| of the piece)
        rec[0] = WFK_M_LIN_REC; rec[3] = dstride; fast = true;
      } else if (type == WFK_COS && std::isfinite(a[0]) && rate_safe_n(a[0], s0, s1)) {
        rec[0] = WFK_M_COS_TAB; rec[3] = a[0]; fast = true;
        table = table_for(B, a[0] * dstride);
      } else if (type == WFK_GAUSSIAN) {
        // exp(-((u+k D)/s)^2): v=u/s, H=D/s.  A lane's seed may sit up to NS strides
        // outside the piece, so the range check includes that overhang.
        bool ok64, ok32;
        gauss_range(a[0], shift, tshift, s0, s1, ok64, ok32);
        if (ok64 && rate_safe(0.86 / a[0], s0, s1)) {   // exp(-676) ~ 1e-294: normal fp64
          double Hh = dstride / a[0];
          rec[0] = WFK_M_GAUSS_REC; rec[3] = a[0]; rec[4] = Hh; rec[5] = std::exp(-2.0 * Hh * Hh);
          // fp32 state is safe only while g and r stay inside float's exponent range
          rec[9] = ok32 ? 1.0 : 0.0;
          fast = true;
        }
      } else if (type == WFK_SINC && std::isfinite(a[0]) && a[0] != 0.0 && rate_safe(3.141592653589793 * a[0], s0, s1) &&
                 !env_no_sinc_tab) {
        // sin(pi b u) from the phasor table, the argument pi b u advanced by the very same phase step (so the two
        // stay consistent where the argument passes through zero), one reciprocal per sample instead of libm's
        // sin + a division (reference _waveform.pyx:303-305: np.sinc)
        const double dphase = 3.141592653589793 * a[0] * dstride;
        rec[0] = WFK_M_SINC_TAB; rec[3] = a[0]; rec[4] = dphase; fast = true;
        table = table_for(B, dphase);
      } else if (type == WFK_MOLLIFIER && a[1] == 0.0 && std::isfinite(a[0]) && a[0] > 0.0 && rate_safe(4.0 / a[0], s0, s1) &&
                 !env_no_moll_rec) {
        // exp(1 / (x^2 - 1) + 1), x = u / r (reference _waveform.pyx:359-363): the exponent is <= 0 inside the
        // support; Newton reciprocal + inline exponential instead of an IEEE division and libm's exp
        rec[0] = WFK_M_MOLL_REC; rec[3] = a[0]; rec[4] = dstride; fast = true;
      } else if (type == WFK_EXP && std::isfinite(a[0]) && rate_safe(a[0], s0, s1)) {
        double ext = std::fabs(a[0]) * (umax + dstride * NS);
        if (ext <= 600.0) {
          rec[0] = WFK_M_EXP_REC; rec[3] = a[0]; rec[4] = std::exp(a[0] * dstride);
          rec[9] = ext <= 80.0 ? 1.0 : 0.0;
          fast = true;
        }
      }
    }
    bool interp_lin = false;
    if (!fast) {
      switch (type) {
        case WFK_INTERP: {
          rec[3] = a[0]; rec[4] = a[1]; rec[5] = (double)(na - 2); rec[6] = (double)H.pool.size();
          H.pool.insert(H.pool.end(), a + 2, a + na);
          // np.interp's per-sample slope (fp[j+1]-fp[j]) / (xp[j+1]-xp[j]) once per knot, in the
          // same IEEE double operations (this file is built -ffp-contract=off), and 1/step for an
          // O(1) knot index: the device then skips the binary search and the division
          const int64_t m = na - 2;
          const double start = a[0], stop = a[1];
          rec[7] = -1.0;
          if (m >= 2 && stop > start && std::isfinite(start) && std::isfinite(stop)) {
            const double step = (stop - start) / (double)(m - 1);
            auto X = [&](int64_t k) {
              if (k == m - 1) return stop;
              volatile double q = (double)k * step;
              return q + start;
            };
            rec[7] = (double)H.pool.size();
            rec[8] = 1.0 / step;
            rec[9] = step;
            const double* fp = a + 2;
            for (int64_t j = 0; j + 1 < m; ++j) {
              volatile double dy = fp[j + 1] - fp[j], dx = X(j + 1) - X(j);
              H.pool.push_back(dy / dx);
            }
            H.pool.push_back(0.0);
            // grid mode: same arithmetic, but the knot/slope loads (L2 latency) of four samples
            // are in flight together instead of one dependent load pair per sample
            if (!H.tlist && !nofast && pw == 1.0 && m < (int64_t(1) << 31) && !env_no_interp_grid) {
              rec[0] = WFK_M_INTERP_GRID;
              // A finite table is a CONTINUOUS piecewise-linear function: next to a knot the two adjoining
              // segments agree to rounding, so the knot index may come straight from the O(1) guess, without
              // the comparisons against the exact knot abscissae that make up most of the exact lookup (an
              // error of one segment there costs |slope difference| x a few ulp of x).  Admitted while
              // max |slope| x (rounding of x) stays inside the jitter budget; counted as a fast factor: the plan
              // then runs the build of the general kernel without the direct tier.
              double smax = 0.0;
              bool finite = true;
              for (int64_t j = 0; j < m && finite; ++j) finite = std::isfinite(fp[j]);
              for (int64_t j = 0; j + 1 < m && finite; ++j) {
                const double sl = H.pool[(size_t)rec[7] + (size_t)j];
                finite = std::isfinite(sl);
                smax = std::max(smax, std::fabs(sl));
              }
              if (finite && s1 > s0 && rate_safe(4.0 * smax, s0, s1) && !env_no_interp_lin) {
                rec[0] = WFK_M_INTERP_LIN;
                interp_lin = true;
              }
            }
          }
          break;
        }
        case WFK_MOLLIFIER: {
          int d = (int)a[1];
          rec[3] = a[0]; rec[4] = d;
          if (d > 0) {
            std::vector<double> poly = mollifier_poly(d);
            rec[5] = (double)(poly.size() - 1); rec[6] = (double)H.pool.size();
            rec[7] = std::pow(a[0], (double)d);
            H.pool.insert(H.pool.end(), poly.begin(), poly.end());
          }
          break;
        }
        case WFK_D_GAUSSIAN:
          rec[3] = a[0]; rec[4] = a[1];
          rec[5] = std::pow(-1.0, a[1]) / std::pow(a[0], a[1]);
          break;
        case WFK_DRAG_SIN: case WFK_DRAG_SINX:
          rec[3] = (double)H.pool.size();
          H.pool.insert(H.pool.end(), a, a + na);
          break;
        case WFK_SAMPLED: {
          // one copy per table, however many device pieces the member piece was cut into
          auto it = sampled_at.find(P->fc_arg_off[f]);
          if (it == sampled_at.end()) {
            it = sampled_at.emplace(P->fc_arg_off[f], (int64_t)H.pool.size()).first;
            H.pool.insert(H.pool.end(), a + 1, a + na);
            if (na == 1) H.pool.push_back(NAN);    // empty table: never indexed, keep the offset valid
          }
          rec[0] = WFK_M_SAMPLED; rec[3] = (double)it->second; rec[4] = a[0]; rec[5] = (double)(na - 1);
          break;
        }
        default:
          for (int64_t k = 0; k < na && k < 6; ++k) rec[3 + k] = a[k];
      }
      if (interp_lin) ++H.n_fast; else ++H.n_direct;
      std::vector<double> sig(rec, rec + WFK_FREC);
      const bool pooled = type == WFK_INTERP || type == WFK_MOLLIFIER || type == WFK_DRAG_SIN ||
                          type == WFK_DRAG_SINX || type == WFK_SAMPLED;             // records point into the pool: never equal
      if (!pooled && rec[0] < 100.0 && sig == last_direct) rec[0] += WFK_M_REUSE;
      else last_direct = pooled || rec[0] >= 100.0 ? std::vector<double>() : sig;
    } else {
      ++H.n_fast;
    }
    size_t at = B.body.size();
    B.body.insert(B.body.end(), rec, rec + WFK_FREC);
    if (table >= 0) B.table_refs.emplace_back(at + WFK_FREC - 1, table);
  };

  auto flush_block = [&](BlockBuilder& B) -> int32_t {
    last_direct.clear();   // the value buffer is only trusted within one block
    // [len, n_terms, body..., pad?, tables...]
    size_t tab_off = WFK_BLK_HDR + B.body.size();
    if (tab_off & 1) ++tab_off;  // 16-byte aligned tables
    for (auto& r : B.table_refs) B.body[r.first] = (double)(tab_off + (size_t)r.second * 2 * (NS + 1));
    for (size_t at : B.fce_ats) {   // the table offset also travels in the packed op word ...
      if (!(((int64_t)B.body[at + WFK_FCE_DEG]) & WFK_FCE_CHIRP))      // (a chirp has no table: the slot holds its constant phasor)
        B.body[at + WFK_FCE_DEG] += 256.0 * B.body[at + WFK_FCE_TAB];
      // ... which the device reads as a 32-bit integer (WFK_FCE_WORD): low half of the slot
      const uint64_t word = (uint32_t)(int64_t)B.body[at + WFK_FCE_DEG];
      std::memcpy(&B.body[at + WFK_FCE_DEG], &word, sizeof word);
    }
    size_t len = tab_off + B.tables.size();
    H.max_block_len = std::max<int32_t>(H.max_block_len, (int32_t)len);
    H.params.push_back((double)len);
    H.params.push_back((double)B.n_terms);
    H.params.insert(H.params.end(), B.body.begin(), B.body.end());
    if ((WFK_BLK_HDR + B.body.size()) & 1) H.params.push_back(0.0);
    H.params.insert(H.params.end(), B.tables.begin(), B.tables.end());
    if (H.params.size() & 1) { H.params.push_back(0.0); }  // keep every block start even
    B = BlockBuilder();
    return (int32_t)len;
  };

  // ---- fusion: terms -> carrier-envelope groups ---------------------------------
  const char* nolean_env = std::getenv("WFK_DISABLE_LEAN");
  const bool nolean = nolean_env && nolean_env[0] == '1';
  const char* nofuse_env = std::getenv("WFK_DISABLE_FUSE");
  // (time-list plans fuse too: their ops are evaluated pointwise -- one sincos + one exp per group and
  //  sample instead of a libm call per factor; WFK_DISABLE_TLFUSE=1: every factor on device libm, as before)
  const char* notlfuse_env = std::getenv("WFK_DISABLE_TLFUSE");
  const bool can_fuse = (!H.tlist || !(notlfuse_env && notlfuse_env[0] == '1')) && !nofast &&
                        !(nofuse_env && nofuse_env[0] == '1');

  const char* noexp_env = std::getenv("WFK_DISABLE_EXPFUSE");
  const bool expfuse = !(noexp_env && noexp_env[0] == '1');
  const char* noerf_env = std::getenv("WFK_DISABLE_ERFMUL");
  const bool erfmod_base = can_fuse && !H.tlist && ns_override == 0 && !(noerf_env && noerf_env[0] == '1');
  // stateless closing multipliers (INTERP tables, mollifiers): ops of the lean kernel's family 3 ONLY, so they
  // exist where every piece flagged lean is certain to run on that kernel (lean or mixed plans in the standard
  // geometry, no corrected carriers: that instantiation has families 0 / 1 only)
  const char* nofmul_env = std::getenv("WFK_DISABLE_FMUL");
  const char* nomix_env0 = std::getenv("WFK_DISABLE_MIXED");
  const bool fmul_base = erfmod_base && !nolean && !g_no_chirp && !(nofmul_env && nofmul_env[0] == '1') &&
                         !(nomix_env0 && nomix_env0[0] == '1');
  bool piece_fmul_ok = true;   // cleared for the second attempt at a piece that turned out not to be lean
  bool plan_has_bank = false;  // some piece holds a run of bare carriers (tone loop): longer chunks pay there

  std::vector<FceGroup> fuse_staged;
  auto fuse_term = [&](std::vector<FceGroup>& groups, int32_t k, double tshift, int64_t s0,
                       int64_t s1, int32_t skip = -1) -> bool {   // skip: a factor handled by the caller
    // a complex amplitude a + ib contributes a * (...) to the real part and b * (...) to the
    // imaginary part of the output: two real-coefficient contributions, the second into groups
    // marked `imag` (their op accumulates into the imaginary accumulators)
    const int32_t f0 = P->tm_factor_off[k], f1 = P->tm_factor_off[k + 1];
    int p = 0, ncos = 0;
    // the term's polynomial factor in u = t' - slin: LINEAR powers shift it up, a Gaussian derivative
    // (D_GAUSSIAN) multiplies it by its Hermite polynomial
    long double tp[4] = {1.0L, 0.0L, 0.0L, 0.0L};
    auto poly_times = [&](const long double (&c)[4]) -> bool {   // tp *= c(u), degree <= 3
      long double out[7] = {0, 0, 0, 0, 0, 0, 0};
      for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) out[i + j] += tp[i] * c[j];
      for (int i = 4; i < 7; ++i)
        if (out[i] != 0.0L) return false;
      for (int i = 0; i < 4; ++i) tp[i] = out[i];
      return true;
    };
    bool has_lin = false, has_env = false, env32 = false;
    double slin = 0, sigma = 0, sg = 0;
    double first_cos_shift = 0;
    // c * cos(K u^2 + W u - Psi),  u = t' - o,  t' = t - tshift; o = 0 for every plain carrier (K == 0), a chirp's own
    // shift for a chirp -- and for whatever it has been multiplied with
    struct Car { long double c, W, Psi, K = 0.0L, o = 0.0L; };
    auto rebase = [](Car r, long double o_new) -> Car {     // the same carrier expanded about another origin
      const long double d = o_new - r.o;                    // t' - r.o = (t' - o_new) + d
      return Car{r.c, r.W + 2 * r.K * d, r.Psi - r.K * d * d - r.W * d, r.K, o_new};
    };
    struct ExpV { long double c, a, b; };    // c * exp(a t' + b): EXP factors, COSH / SINH as two of them
    SmallVec<ExpV, 8> evs;
    evs.push_back({1.0L, 0.0L, 0.0L});
    bool has_expf = false;
    struct CosF { double w, sh, thmax; };    // reference COS factors of the term (|w|, shift, largest |phase|)
    SmallVec<CosF, 40> cosf;
    bool has_drag = false;
    const long double PI = 3.141592653589793238462643383279502884L;
    SmallVec<Car, 32> cars;                  // empty: no carrier factor seen yet
    // multiply the running carrier sum by another sum of carriers:
    //   cos a cos b = (cos(a+b) + cos(a-b)) / 2
    auto times = [&](std::initializer_list<Car> f) -> bool {
      if (cars.empty()) { for (const Car& r : f) cars.push_back(r); return true; }
      if (cars.size() * f.size() * 2 > 32) return false;
      SmallVec<Car, 32> nx;
      for (const Car& q0 : cars)
        for (const Car& r0 : f) {
          Car q = q0, r = r0;
          if (q.o != r.o) {                  // (never for plain carriers: both at 0)
            if (q.K != 0.0L) r = rebase(r, q.o);
            else q = rebase(q, r.o);
          }
          nx.push_back({q.c * r.c / 2, q.W + r.W, q.Psi + r.Psi, q.K + r.K, q.o});
          nx.push_back({q.c * r.c / 2, q.W - r.W, q.Psi - r.Psi, q.K - r.K, q.o});
        }
      cars.swap(nx);
      return true;
    };
    for (int32_t f = f0; f < f1; ++f) {
      if (f == skip) continue;
      const double pw = P->fc_power[f], sh = P->fc_shift[f];
      const double* a = P->pool + P->fc_arg_off[f];
      switch (P->fc_type[f]) {
        case WFK_LINEAR: {
          if (!(pw == 1.0 || pw == 2.0 || pw == 3.0)) return false;
          if (has_lin && sh != slin) return false;
          const double ua = (ax.at(s0) - tshift) - sh, ub = (ax.at(s1 - 1) - tshift) - sh;
          const double um = std::max(std::fabs(ua), std::fabs(ub));
          if (!(um > 0.0) || !rate_safe(pw / um, s0, s1)) return false;   // grid jitter vs |u|
          has_lin = true; slin = sh; p += (int)pw;
          long double mono[4] = {0, 0, 0, 0};
          mono[(int)pw] = 1.0L;
          if (!poly_times(mono)) return false;
          break;
        }
        case WFK_D_GAUSSIAN: {
          // (-1/s)^n H_n(u/s) exp(-(u/s)^2), n <= 3: a polynomial times the Gaussian envelope
          // (reference _waveform.pyx:298-300; what gaussian(width, d=n) and second derivatives make)
          const double sgm = a[0];
          const int n = (int)a[1];
          if (pw != 1.0 || has_env || n < 0 || n > 3 || !std::isfinite(sgm) || sgm == 0.0) return false;
          if (has_lin && sh != slin) return false;
          bool ok64;
          gauss_range(sgm, sh, tshift, s0, s1, ok64, env32);
          if (!ok64 || !rate_safe(2.0 * (n + 1) / sgm, s0, s1)) return false;
          has_env = true; sigma = sgm; sg = sh;
          if (n > 0) {
            const long double s2 = (long double)sgm * sgm;
            long double h[4] = {0, 0, 0, 0};
            if (n == 1) { h[1] = -2.0L / s2; }
            else if (n == 2) { h[0] = -2.0L / s2; h[2] = 4.0L / (s2 * s2); }
            else { h[1] = 12.0L / (s2 * s2); h[3] = -8.0L / (s2 * s2 * s2); }
            if (!has_lin) { has_lin = true; slin = sh; }
            p += n;
            if (!poly_times(h)) return false;
          }
          break;
        }
        case WFK_GAUSSIAN: {
          // exp(-(u/s)^2)^p = exp(-(u / (s / sqrt p))^2): a power of a Gaussian is a narrower (p > 1) or wider one
          // (the reference takes np.power of the value: the same number to (u/s)^2 p x 2^-52 <= 2e-13 relative)
          if (!(pw > 0.0) || !std::isfinite(pw) || has_env) return false;
          const double sg_eff = pw == 1.0 ? a[0] : a[0] / std::sqrt(pw);
          if (!std::isfinite(sg_eff) || sg_eff == 0.0) return false;
          bool ok64;
          gauss_range(sg_eff, sh, tshift, s0, s1, ok64, env32);
          if (!ok64 || !rate_safe(0.86 / sg_eff, s0, s1)) return false;
          has_env = true; sigma = sg_eff; sg = sh;
          break;
        }
        case WFK_COS: {
          // cos^2, cos^3: the factor taken two / three times by product-to-sum (no rounding correction then: the
          // reference rounds the phase ONCE and raises the cosine, which the corrected form does not mimic)
          if (!(pw == 1.0 || pw == 2.0 || pw == 3.0) || !std::isfinite(a[0]) || !std::isfinite(sh)) return false;
          if (pw != 1.0) has_drag = true;
          for (int rep = 0; rep < (int)pw; ++rep) {
            if (ncos == 0) first_cos_shift = sh;
            ++ncos;
            if (!times({{1.0L, (long double)a[0], (long double)a[0] * sh}})) return false;
            const double ua = (ax.at(s0) - tshift) - sh, ub = (ax.at(s1 - 1) - tshift) - sh;
            cosf.push_back({std::fabs(a[0]), sh, std::fabs(a[0]) * std::max(std::fabs(ua), std::fabs(ub))});
          }
          break;
        }
        case WFK_EXP: {
          // exp(alpha u)^p = exp(p alpha t' - p alpha shift)
          if (!expfuse || !std::isfinite(a[0]) || !std::isfinite(pw) || !std::isfinite(sh)) return false;
          for (ExpV& e : evs) { e.a += (long double)pw * a[0]; e.b -= (long double)pw * a[0] * sh; }
          has_expf = true;
          break;
        }
        case WFK_COSH: case WFK_SINH: {
          // (e^{w u} +- e^{-w u}) / 2
          if (!expfuse || pw != 1.0 || !std::isfinite(a[0]) || !std::isfinite(sh) || evs.size() > 2) return false;
          const long double sgn = P->fc_type[f] == WFK_COSH ? 1.0L : -1.0L;
          SmallVec<ExpV, 8> nx;
          for (const ExpV& e : evs) {
            nx.push_back({e.c / 2, e.a + a[0], e.b - (long double)a[0] * sh});
            nx.push_back({sgn * e.c / 2, e.a - a[0], e.b + (long double)a[0] * sh});
          }
          evs.swap(nx);
          has_expf = true;
          break;
        }
        case WFK_LINEARCHIRP: {
          // sin(phi0 + 2 pi (a u^2 + f0 u)), a = (f1 - f0) / (2 T), u = t' - shift (reference
          // _waveform.pyx:323-324) = cos(K t'^2 + W t' - Psi): a carrier whose phase is quadratic.  Fused
          // where the piece can run on the lean kernel (its chirp family: the phasor advances by a phasor
          // that itself advances by a constant -- the complex twin of the Gaussian recurrence).
          if (!chirp_ok || pw != 1.0 || !std::isfinite(sh)) return false;
          for (int i = 0; i < 4; ++i)
            if (!std::isfinite(a[i])) return false;
          if (a[2] == 0.0) return false;
          const long double Kq = 2 * PI * ((long double)a[1] - a[0]) / (2 * (long double)a[2]);
          const long double W0 = 2 * PI * (long double)a[0];
          ++ncos;
          ++ncos;      // (never a single COS factor: the group's phase reference is the term's own)
          if (!times({{1.0L, W0, -((long double)a[3] - PI / 2), Kq, (long double)sh}})) return false;
          break;
        }
        case WFK_DRAG: {
          // sin^2(o tau) cos(wt) + Oy sin(wt),  tau = u - t0,  Oy = -b o sin(2 o tau)
          //   = cos(wt)/2 + (-1/4 + b o/2) cos(wt + 2 o tau) + (-1/4 - b o/2) cos(wt - 2 o tau)
          // (reference: _waveform.pyx:343-356): three plain carriers, no libm per sample
          const double t0 = a[0], freq = a[1], width = a[2], delta = a[3], bf = a[4], phase = a[5];
          if (pw != 1.0 || !(width != 0.0) || !std::isfinite(sh)) return false;
          for (int i = 0; i < 6; ++i)
            if (i != 4 && !std::isfinite(a[i])) return false;
          const long double o = PI / width;
          const long double W = 2 * PI * ((long double)freq + delta);
          const long double Psi0 = W * sh + (2 * PI * (long double)delta * t0 + phase);
          const long double Om = 2 * o, Psi1 = Om * ((long double)sh + t0);
          long double bo = 0.0L;
          if (!std::isnan(bf) && bf - delta != 0.0) bo = o / (2 * PI * ((long double)bf - delta));
          ncos += 2;   // reference shift is derived from Psi/W
          has_drag = true;
          if (!times({{0.5L, W, Psi0},
                      {-0.25L + bo / 2, W + Om, Psi0 + Psi1},
                      {-0.25L - bo / 2, W - Om, Psi0 - Psi1}}))
            return false;
          break;
        }
        default:
          return false;
      }
    }
    if (p > 3 || !std::isfinite(P->tm_amp_re[k]) || !std::isfinite(P->tm_amp_im[k])) return false;
    if (cars.empty()) cars.push_back({1.0L, 0.0L, 0.0L});
    const double cs[1] = {first_cos_shift};
    // stage the contributions; commit only if every carrier finds/creates a group
    std::vector<FceGroup>& staged = fuse_staged;       // (scratch kept across terms: no allocation here)
    staged = groups;
    bool any_corr = false;
    const double tpa = ax.at(s0) - tshift, tpb = ax.at(s1 - 1) - tshift;   // the piece on the channel's own axis
    for (const Car& q : cars) {
      if (q.K != 0.0L) {
        // a chirp: largest instantaneous frequency over the piece; no rounding correction for it
        const double wmax = (double)std::max(fabsl(q.W + 2 * q.K * (tpa - q.o)), fabsl(q.W + 2 * q.K * (tpb - q.o)));
        if (!std::isfinite(wmax) || !rate_safe_n(wmax, s0, s1)) return false;
        continue;
      }
      if (!rate_safe_n((double)q.W, s0, s1)) {
        if (shortm && (!cur_short || tshift != 0.0 || p > 1 || env_no_short_corr || !corr_enabled)) {
          // (only the lean kernel corrects these: a shifted channel, a polynomial of degree > 1 -- and a piece of a short
          //  plan that is built again for the GENERAL kernel, which has no correction: 3e-9 on such a piece, found by the
          //  far fuzz)
          H.short_needs_corr = true;
          return false;
        }
        if (!corr_safe((double)q.W, s0, s1)) { if (shortm) H.short_needs_corr = true; return false; }
        any_corr = true;
      }
    }
    // the term's weight and the reference factor whose rounding it mimics (see corr_safe)
    double wm = 0, sm = 0, weight = 0;
    if (any_corr) {
      if (has_drag || cosf.empty()) return false;      // (the DRAG primitive rounds its phase differently)
      size_t im = 0;
      for (size_t i = 1; i < cosf.size(); ++i)
        if (cosf[i].thmax > cosf[im].thmax) im = i;
      wm = cosf[im].w; sm = cosf[im].sh;
      weight = std::fabs(P->tm_amp_re[k]) + std::fabs(P->tm_amp_im[k]);
      if (has_lin) {
        const double ua = (ax.at(s0) - tshift) - slin, ub = (ax.at(s1 - 1) - tshift) - slin;
        weight *= std::pow(std::max(std::fabs(ua), std::fabs(ub)), p);
      }
      for (size_t i = 0; i < cosf.size(); ++i)         // the other factors' own phase noise must not matter
        if (i != im && 2.3e-16 * cosf[i].thmax * weight > jtol) return false;
    }
    // Exponential factors: under a Gaussian they only move its centre and scale it,
    //   exp(-((t'-sg)/s)^2 + a t' + b) = exp(b + a sg + a^2 s^2 / 4) exp(-((t' - sg - a s^2 / 2)/s)^2);
    // alone they are an envelope of their own, exp(a (t' - ref)) with the reference time in the piece
    // (state g = exp(a (x - ref)), constant ratio exp(a D): the Gaussian recurrence with q = 1).
    struct EnvV { long double amp; bool has_env, env32, has_exp; double sigma, sg; };
    SmallVec<EnvV, 8> envs;
    for (const ExpV& e : evs) {
      EnvV v{e.c, has_env, env32, false, sigma, sg};
      if (has_expf && e.a != 0.0L) {
        if (has_env) {
          const long double lg = e.b + e.a * sg + e.a * e.a * (long double)sigma * sigma / 4;
          const double sg2 = (double)((long double)sg + e.a * (long double)sigma * sigma / 2);
          bool ok64, ok32;
          gauss_range(sigma, sg2, tshift, s0, s1, ok64, ok32);
          if (!(fabsl(lg) <= 600.0L) || !std::isfinite(sg2) || !ok64) return false;
          v.amp *= expl(lg); v.sg = sg2; v.env32 = ok32;
        } else {
          const double ref = 0.5 * (ax.at(s0) + ax.at(s1 - 1)) - tshift;
          const double half = 0.5 * std::fabs(ax.at(s1 - 1) - ax.at(s0)) + (NS + 1) * dstride;   // + a tile of overhang
          const long double lg = e.a * ref + e.b;
          const double ar = (double)e.a;
          if (!std::isfinite(ref) || !(fabsl(lg) <= 600.0L) || !(std::fabs(ar) * half <= 600.0) || !rate_safe(ar, s0, s1))
            return false;
          {
            // an exponential of amplitude >> 1 (coshPulse with a small eps is 23 cos - 22 cos cosh, cancelling to O(1))
            // turns the envelope's share of the grid jitter into that many times the value error: 1.5e-9 of peak 1 ms
            // out (tools/fuzz_soak.py awgfar) -- the envelope's rate is weighed with the term's amplitude there
            double wamp = (std::fabs(P->tm_amp_re[k]) + std::fabs(P->tm_amp_im[k])) * (double)fabsl(e.c * expl(lg));
            if (has_lin) {
              const double ua = (ax.at(s0) - tshift) - slin, ub = (ax.at(s1 - 1) - tshift) - slin;
              wamp *= std::pow(std::max(std::fabs(ua), std::fabs(ub)), p);
            }
            wamp *= std::exp(std::fabs(ar) * 0.5 * std::fabs(ax.at(s1 - 1) - ax.at(s0)));      // its largest value over the piece
            if (std::isfinite(wamp) && wamp > 1.0 && !rate_safe_n(2.0 * ar * wamp, s0, s1)) return false;   // (2: seed and sample both rounded)
          }
          v.amp *= expl(lg); v.has_env = true; v.has_exp = true; v.sigma = ar; v.sg = ref;
          v.env32 = std::fabs(ar) * half <= 80.0;
        }
      } else if (has_expf) {
        if (!(fabsl(e.b) <= 600.0L)) return false;
        v.amp *= expl(e.b);                                  // (rates cancelled: a constant)
      }
      envs.push_back(v);
    }
    for (const EnvV& ev : envs) {
    const bool has_env = ev.has_env, env32 = ev.env32, has_exp = ev.has_exp;   // (shadow the term-level ones)
    const double sigma = ev.sigma, sg = ev.sg;
    for (int part = 0; part < 2; ++part) {
    const long double amp_part = (part == 0 ? (long double)P->tm_amp_re[k] : (long double)P->tm_amp_im[k]) * ev.amp;
    if (amp_part == 0.0L) continue;                 // nothing in this part
    const bool imag = part == 1;
    for (Car q : cars) {
      q.c *= amp_part;
      if (q.W < 0 || (q.W == 0 && q.K < 0)) { q.W = -q.W; q.Psi = -q.Psi; q.K = -q.K; }
      const double W = (double)q.W;
      const bool is_chirp = q.K != 0.0L;
      // (a sum/difference frequency is rounded to double: relative error <= 2^-53, the
      //  same class as the reference's own rounding of w*t)
      FceGroup* G = nullptr;
      double thm = 0;
      for (const CosF& cf : cosf) thm = std::max(thm, cf.thmax);
      const bool heavy = 2.3e-16 * thm * weight > jtol;   // its own phase rounding must be mimicked
      for (FceGroup& g : staged)
        if (g.W == W && (double)g.K == (double)q.K && g.corg == q.o && g.imag == imag && g.has_env == has_env && g.has_exp == has_exp &&
            (!has_env || (g.sigma == sigma && g.sg == sg))) {
          // a corrected group mimics ONE reference factor: a heavy term with another factor founds
          // its own group (same carrier, own op) instead of joining
          if (any_corr && heavy && (g.wm != wm || g.sm != sm)) continue;
          G = &g;
          break;
        }
      if (!G) {
        staged.emplace_back();
        G = &staged.back();
        G->W = W; G->has_env = has_env; G->has_exp = has_exp; G->sigma = sigma; G->sg = sg; G->env32 = env32;
        G->imag = imag;
        G->K = q.K; G->chirp = is_chirp; G->Wl = q.W; G->corg = q.o;
        G->corr = !is_chirp && W != 0.0 && !rate_safe_n(W, s0, s1);
        G->wm = wm; G->sm = sm;
        G->sref = W == 0.0 ? 0.0 : (ncos == 1 ? cs[0] : (double)(q.Psi / q.W));
        G->psi_ref = (long double)W * G->sref;
        if (is_chirp) {      // (the founding term's own phase; reference time = middle of the piece)
          G->sref = 0.0; G->psi_ref = q.Psi;
          G->tref = 0.5 * (tpa + tpb);
        }
        if (!is_chirp && W != 0.0 && ncos != 1) {
          // A sum / difference frequency is rounded to double: (W_exact - W) * (x - s_ref) must stay
          // small, so a corrected carrier takes its reference time INSIDE the piece (with s_ref =
          // Psi / W next to t = 0 the rounding of W is multiplied by |x|: 5e-9 rad at 100 s) -- and so does every
          // other such carrier once that product is more than 2e-12 rad (2e-10 rad per group 1 ms out at 300 MHz:
          // the random AWG-rate scripts moved 1-10 ms from t = 0 had grid AND time-list launches at 1-2.5e-9 of peak,
          // tools/fuzz_soak.py awgfar; plans near t = 0 keep their records bit for bit)
          const double mid = 0.5 * (ax.at(s0) + ax.at(s1 - 1)) - tshift;
          if (std::isfinite(mid) && (G->corr || 1.2e-16 * std::fabs(W) * std::fabs(mid - G->sref) > 2e-12)) {
            G->sref = mid;
            G->psi_ref = q.W * (long double)mid;       // q.W: the exact sum (long double)
          }
        }
      }
      if (has_env && !env32) G->env32 = false;
      long double ca, cb;
      if (W == 0.0 && !is_chirp) { ca = q.Psi == 0.0L ? q.c : q.c * cosl(q.Psi); cb = 0.0L; }
      else {
        const long double delta = G->psi_ref - q.Psi;   // cos(th_ref + delta)
        if (delta == 0.0L) { ca = q.c; cb = 0.0L; }     // (the term that founded the group)
        else {
          long double sd, cd;
          sincosl(delta, &sd, &cd);
          ca = q.c * cd;
          cb = -q.c * sd;
        }
      }
      // multiply by u_term^p with u_term = u_group + d
      long double d = 0.0L;
      if (p > 0) {
        // the group's polynomial variable is centred on the piece (u = t' - slin with slin the
        // mid time of [s0, s1)), not on the first term's own origin: |u| stays <= half the piece
        // span, which keeps the float evaluation of A(u), B(u) free of the cancellation a far
        // origin brings (poly() terms have theirs at t = 0)
        if (!G->has_lin) {
          G->has_lin = true;
          G->slin = 0.5 * (ax.at(s0) + ax.at(s1 - 1)) - tshift;
          if (!std::isfinite(G->slin)) G->slin = slin;
        }
        d = (long double)G->slin - slin;
      }
      static const int binom[4][4] = {{1, 0, 0, 0}, {1, 1, 0, 0}, {1, 2, 1, 0}, {1, 3, 3, 1}};
      const long double dpow[4] = {1.0L, d, d * d, d * d * d};
      for (int m = 0; m <= p; ++m) {           // every monomial tp[m] (u_group + d)^m of the term's polynomial
        if (tp[m] == 0.0L) continue;
        for (int i = 0; i <= m; ++i) {
          long double f = tp[m] * binom[m][i] * dpow[m - i];
          G->A[i] += ca * f;
          G->B[i] += cb * f;
        }
      }
      if (p > G->deg) G->deg = p;
      ++G->nterms;
    }
    }
    }
    groups.swap(staged);
    return true;
  };

  // (f_j, f_{j+1} - f_j) pairs of an INTERP factor's table in the pool, in 16-byte entries; one copy per distinct
  // table of the plan (a gate set reuses a few pulse shapes many times), found again by content
  std::multimap<uint64_t, int64_t> fmul_tables;
  auto fmul_table = [&](int32_t f) -> int64_t {
    const double* fa = P->pool + P->fc_arg_off[f];
    const int64_t m = P->fc_arg_off[f + 1] - P->fc_arg_off[f] - 2;
    const double* fp = fa + 2;
    uint64_t h = 1469598103934665603ull ^ (uint64_t)m;
    for (int64_t j = 0; j < m; ++j) {
      uint64_t b;
      std::memcpy(&b, &fp[j], 8);
      h = (h ^ b) * 1099511628211ull;
    }
    auto range = fmul_tables.equal_range(h);
    for (auto it = range.first; it != range.second; ++it) {
      const size_t at = (size_t)it->second * 2;
      if (at + 2 * (size_t)(m + 1) > H.pool.size()) continue;          // (rolled back with its piece)
      bool same = true;
      for (int64_t j = 0; j < m && same; ++j) {
        const double dj = j + 1 < m ? fp[j + 1] - fp[j] : 0.0;
        same = std::memcmp(&H.pool[at + 2 * (size_t)j], &fp[j], 8) == 0 && std::memcmp(&H.pool[at + 2 * (size_t)j + 1], &dj, 8) == 0;
      }
      same = same && H.pool[at + 2 * (size_t)m + 1] == 0.0 && std::memcmp(&H.pool[at + 2 * (size_t)m], &fp[m - 1], 8) == 0;
      if (same) return it->second;
    }
    if (H.pool.size() & 1) H.pool.push_back(0.0);
    const int64_t at = (int64_t)(H.pool.size() / 2);
    for (int64_t j = 0; j < m; ++j) {
      H.pool.push_back(fp[j]);
      H.pool.push_back(j + 1 < m ? fp[j + 1] - fp[j] : 0.0);
    }
    H.pool.push_back(fp[m - 1]); H.pool.push_back(0.0);
    fmul_tables.emplace(h, at);
    return at;
  };

  auto emit_group = [&](BlockBuilder& B, FceGroup& G) {
    double rec[WFK_FCE_REC] = {0};
    long double A0 = G.A[0], B0 = G.B[0];
    double sref = G.sref;
    int deg = G.deg;
    if (G.chirp) {
      if (B0 != 0.0L && deg == 0) deg = 1;   // (a chirp keeps A and B: its loop has no degree-0 form with a folded shift)
    } else if (G.deg == 0 && G.W != 0.0 && G.corr) {
      // (corrected carriers keep A and B: the folded shift s_ref + phi/W would be ROUNDED to a
      //  double next to |s_ref| -- half an ulp of a time 100 s from the origin is 7e-15 s, 3e-9 rad
      //  under a 60 kHz carrier.  The degree-0 loop has no B term: such an op runs as degree 1.)
      if (B0 != 0.0L) deg = 1;
    } else if (G.deg == 0 && G.W != 0.0 && B0 != 0.0L) {
      // A cos(th) + B sin(th) = R cos(th - phi): fold B into the reference shift -- unless the folded shift, ROUNDED to a
      // double next to |s_ref|, would cost more than 1e-12 of the group's amplitude in phase (W ulp(s_ref) / 2: 8e-11 rad
      // at 1 ms under 120 MHz): the op then runs as degree 1 with A and B, like a corrected carrier.  (A group with
      // B == 0 is left alone: folding a NEGATIVE A as phi = pi moved its reference by pi / W for nothing -- 7e-12 of
      // relative error per ms from t = 0 on every term with a negative amplitude, 3e-9 on coshPulse's 23 cos - 22 cos cosh:
      // the far-from-zero fuzz's last class.)
      long double R = hypotl(A0, B0), phi = atan2l(B0, A0);
      const double folded = (double)((long double)G.sref + phi / (long double)G.W);
      const double ulp = std::nextafter(std::fabs(folded), INFINITY) - std::fabs(folded);
      if (G.bank > 0 || G.own_kind || 0.5 * ulp * std::fabs(G.W) * std::max(1.0, (double)R) <= 1e-12) {   // (tone loops and own-term ops are degree 0 by construction)
        sref = folded;
        A0 = R; B0 = 0.0L;
      } else {
        deg = 1;
      }
    }
    if (G.erfmul) {
      // closing erf multiplier (wfk_kernels.hip: fce_erfmul): step series about the midpoint,
      //   2/sqrt(pi) [h + H2 h^3/24 + H4 h^5/1920 + H6 h^7/322560] as a cubic in w = vm^2
      const long double h = (long double)dstride / G.sigma;
      const long double c = 2.0L * h / sqrtl(3.141592653589793238462643383279502884L);
      const long double a2 = h * h / 24, a4 = h * h * h * h / 1920, a6 = h * h * h * h * h * h / 322560;
      rec[0] = WFK_OP_FCE;
      rec[WFK_FCE_DEG] = (double)WFK_FCE_PACK(1, 0, 0, 3, 0);
      rec[WFK_FCE_A] = G.m0; rec[WFK_FCE_A + 1] = G.m1;
      rec[WFK_FCE_B] = (double)(c * (1 - 2 * a2 + 12 * a4 - 120 * a6));
      rec[WFK_FCE_B + 1] = (double)(c * (4 * a2 - 48 * a4 + 720 * a6));
      rec[WFK_FCE_B + 2] = (double)(c * (16 * a4 - 480 * a6));
      rec[WFK_FCE_B + 3] = (double)(c * 64 * a6);
      rec[WFK_FCE_SIGMA] = G.sigma; rec[WFK_FCE_SG] = G.sg;
      rec[WFK_FCE_H] = (double)h; rec[WFK_FCE_Q] = (double)expl(-2.0L * h * h);
      rec[WFK_FCE_D] = dstride;
      rec[WFK_FCE_DEG] += (double)((B.state_units << 19) | WFK_FCE_HAS_CS | WFK_FCE_HAS_GR);
      B.state_units += 2;
      const size_t at = B.body.size();
      B.body.insert(B.body.end(), rec, rec + WFK_FCE_REC);
      B.fce_ats.push_back(at);
      return;
    }
    if (G.fmul) {
      // stateless closing multiplier (wfk_kernels.hip: fce_tabmul / fce_mollmul): no per-lane state, no seed
      const int32_t f = G.fmul_f;
      const double* fa = P->pool + P->fc_arg_off[f];
      rec[0] = WFK_OP_FCE;
      rec[WFK_FCE_DEG] = (double)WFK_FCE_PACK(G.fmul, 0, 0, 3, 0);
      rec[WFK_FCE_SLIN] = P->fc_shift[f];
      rec[WFK_FCE_D] = dstride;
      if (G.fmul == 2) {
        // np.interp on linspace knots (reference _waveform.pyx:309-311) as value + fraction * difference: the
        // table travels as (f_j, f_{j+1} - f_j) pairs -- ONE 16-byte gather per sample -- with a closing
        // (f_{m-1}, 0) entry for x == stop (and one more, should the index guess round up there)
        const int64_t m = P->fc_arg_off[f + 1] - P->fc_arg_off[f] - 2;
        const double start = fa[0], stop = fa[1];
        rec[WFK_FCE_A] = start; rec[WFK_FCE_A + 1] = (double)(m - 1);
        rec[WFK_FCE_A + 2] = (double)(m - 1) / (stop - start);
        rec[WFK_FCE_B] = dstride * rec[WFK_FCE_A + 2];             // the lane stride in knot units
        rec[WFK_FCE_A + 3] = (double)fmul_table(f);                // in 16-byte entries
      } else {
        rec[WFK_FCE_A] = 1.0 / fa[0];
      }
      const size_t at = B.body.size();
      B.body.insert(B.body.end(), rec, rec + WFK_FCE_REC);
      B.fce_ats.push_back(at);
      return;
    }
    rec[0] = WFK_OP_FCE;
    rec[WFK_FCE_W] = G.W;
    rec[WFK_FCE_SREF] = sref;
    rec[WFK_FCE_SLIN] = G.has_lin ? G.slin : 0.0;
    rec[WFK_FCE_DEG] = (double)(WFK_FCE_PACK(deg, (G.W != 0.0 || G.chirp) ? 1 : 0, G.imag ? 1 : 0,
                                             G.envmul ? 3 : (G.has_env ? 1 : 0), G.env32 ? 1 : 0) |
                                (G.corr ? 128 : 0));
    if (G.corr) ++H.n_corr;
    rec[WFK_FCE_A] = (double)A0;
    rec[WFK_FCE_B] = (double)B0;
    for (int i = 1; i < 4; ++i) { rec[WFK_FCE_A + i] = (double)G.A[i]; rec[WFK_FCE_B + i] = (double)G.B[i]; }
    rec[WFK_FCE_WM] = G.wm;     // (slots 13 / 21 carried redundant copies of the packed word before)
    rec[WFK_FCE_SM] = G.sm;
    if (G.has_exp) {
      // exp(alpha (t - ref)): the seeds are exp(alpha (x - ref)) and the constant ratio exp(alpha D)
      rec[WFK_FCE_DEG] += (double)WFK_FCE_EXPENV;
      rec[WFK_FCE_SIGMA] = G.sigma; rec[WFK_FCE_SG] = G.sg;
      rec[WFK_FCE_H] = G.sigma * dstride; rec[WFK_FCE_Q] = 1.0;
      rec[WFK_FCE_F32OK] = G.env32 ? 1.0 : 0.0;
    } else if (G.has_env) {
      double Hh = dstride / G.sigma;
      rec[WFK_FCE_SIGMA] = G.sigma; rec[WFK_FCE_SG] = G.sg;
      rec[WFK_FCE_H] = Hh; rec[WFK_FCE_Q] = std::exp(-2.0 * Hh * Hh);
      rec[WFK_FCE_F32OK] = G.env32 ? 1.0 : 0.0;
    }
    rec[WFK_FCE_D] = dstride;
    if (G.bank > 0) {
      rec[WFK_FCE_DEG] += (double)WFK_FCE_BANK;
      rec[WFK_FCE_SIGMA] = (double)G.bank;
    }
    if (G.own_kind) {
      // (a plain carrier -- degree 0, B folded into the phase -- or a constant: the slots of the higher coefficients are free)
      const int32_t f = G.own_f;
      const double* fa = P->pool + P->fc_arg_off[f];
      rec[WFK_FCE_DEG] += (double)WFK_FCE_OWNMUL;
      rec[WFK_FCE_SLIN] = P->fc_shift[f];
      if (G.own_kind == 2) {
        const int64_t m = P->fc_arg_off[f + 1] - P->fc_arg_off[f] - 2;
        rec[WFK_FCE_B + 3] = 0.0;
        rec[WFK_FCE_B + 2] = fa[0];
        rec[WFK_FCE_A + 1] = (double)(m - 1);
        rec[WFK_FCE_A + 2] = (double)(m - 1) / (fa[1] - fa[0]);
        rec[WFK_FCE_A + 3] = (double)fmul_table(f);
        rec[WFK_FCE_B + 1] = dstride * rec[WFK_FCE_A + 2];
      } else {
        rec[WFK_FCE_B + 3] = 1.0;
        rec[WFK_FCE_A + 1] = 1.0 / fa[0];
      }
    }
    if (H.tlist) {
      // (pointwise evaluation: no lane stride; the H slot carries 1 / sigma -- the Gaussian's argument is formed
      //  by a multiplication, one rounding off the reference's division: 2 v^2 ulp <= 1.5e-13 relative at |v| = 26)
      if (G.has_env && !G.has_exp) rec[WFK_FCE_H] = 1.0 / G.sigma;
      double lim = 1.6e6;       // (tests: WFK_TLSMALL_LIMIT=0 sends every carrier through the two-term 1/pi reduction)
      if (env_tlsmall_limit) lim = std::atof(env_tlsmall_limit);
      if (G.tl_thmax <= lim) rec[WFK_FCE_DEG] += (double)WFK_FCE_TLSMALL;
    }
    if (G.chirp) {
      // phase about tref:  K tau^2 + W' tau + phi0,  tau = t' - tref;  the per-stride rotation advances by
      // the constant phasor exp(i 2 K D^2)
      const long double PI2 = 6.283185307179586476925286766559005768L;
      const long double tr = G.tref - G.corg;           // the reference time, from the origin of the phase polynomial
      rec[WFK_FCE_W] = (double)(G.Wl + 2 * G.K * tr);
      rec[WFK_FCE_SREF] = G.tref;
      rec[WFK_FCE_WM] = (double)G.K;
      rec[WFK_FCE_SM] = (double)remainderl(G.K * tr * tr + G.Wl * tr - G.psi_ref, PI2);
      const long double d2 = 2 * G.K * (long double)dstride * (long double)dstride;
      rec[WFK_FCE_TAB] = (double)cosl(d2);
      rec[WFK_FCE_F32OK] = (double)sinl(d2);
      rec[WFK_FCE_DEG] += (double)WFK_FCE_CHIRP;
    }
    {
      // per-lane state of the lean kernel: a phasor (c, s) and / or a Gaussian (g, r), 1 KB each (a chirp:
      // the phasor, the phasor it advances by, then the envelope)
      const bool cs = G.W != 0.0 || G.chirp, gr = G.has_env || G.envmul;
      const int need = (cs ? 1 : 0) + (G.chirp ? 1 : 0) + (gr ? 1 : 0);
      if (B.state_units + need <= 63) {
        rec[WFK_FCE_DEG] += (double)((B.state_units << 19) | (cs ? WFK_FCE_HAS_CS : 0) | (gr ? WFK_FCE_HAS_GR : 0));
        B.state_units += need;
      } else {
        B.state_units = 1000;   // (a block this large is never lean)
      }
    }
    size_t at = B.body.size();
    B.body.insert(B.body.end(), rec, rec + WFK_FCE_REC);
    B.fce_ats.push_back(at);
    if (G.W != 0.0 && !G.chirp && !H.tlist) B.table_refs.emplace_back(at + WFK_FCE_TAB, table_for(B, G.W * dstride));
  };

  // Flat-top edges at AWG rates (short tier): m0 + m1 erf(v_k) sampled HERE, once per distinct edge, as a table of the
  // pool in fmul_table's layout (value, step to the next) -- the envelope of an own-term op (short_cmul) that is read at
  // whole knots.  A libm erf per sample on the device was two thirds of what a flat-top pulse train cost (150
  // instructions behind a call, per sample of an edge), and the closing op a record of its own behind a dependent load.
  // Two edges share a table when (sigma, m0, m1, length) agree and their v at the first sample differ by no more than
  // the rounding noise of the reference's own t - shift (1.5 ulp of the edge's largest |t|, over sigma) and 2e-11:
  // pulses that sit on the sample grid alike.  Every other edge gets a table of its own (16 B per sample).
  struct ErfTab { double sigma, m0, m1; long double v0; int64_t len, at; double first, last; };
  std::vector<ErfTab> erf_tabs;
  auto erf_table = [&](const FceGroup& E, double tshift, int64_t s0, int64_t s1) -> int64_t {
    const int64_t len = s1 - s0;
    auto xat = [&](int64_t k) { double x = ax.at(k); if (tshift != 0.0) x = x - tshift; return x; };
    const double xa = xat(s0), xb = xat(s1 - 1);
    const long double v0 = ((long double)xa - (long double)E.sg) / (long double)E.sigma;
    const double tmax = std::max(std::max(std::fabs(xa), std::fabs(xb)), std::fabs(E.sg));
    const double tol = std::min(1.5 * (std::nextafter(tmax, INFINITY) - tmax) / std::fabs(E.sigma), 2e-11);
    int scanned = 0;
    for (auto it = erf_tabs.rbegin(); it != erf_tabs.rend() && scanned < 64; ++it, ++scanned) {
      if (it->sigma != E.sigma || it->m0 != E.m0 || it->m1 != E.m1 || it->len != len || fabsl(it->v0 - v0) > (long double)tol) continue;
      const size_t a2 = 2 * (size_t)it->at;
      if (a2 + 2 * (size_t)len > H.pool.size()) continue;                   // (rolled back with its piece)
      if (std::memcmp(&H.pool[a2], &it->first, 8) != 0 || std::memcmp(&H.pool[a2 + 2 * (size_t)(len - 1)], &it->last, 8) != 0) continue;
      return it->at;
    }
    if (H.pool.size() & 1) H.pool.push_back(0.0);
    const int64_t at = (int64_t)(H.pool.size() / 2);
    double prev = 0.0;
    for (int64_t k = s0; k < s1; ++k) {
      const double val = (double)((long double)E.m0 + (long double)E.m1 * erfl(((long double)xat(k) - (long double)E.sg) / (long double)E.sigma));
      if (k > s0) H.pool.push_back(val - prev);
      H.pool.push_back(val);
      prev = val;
    }
    H.pool.push_back(0.0);
    erf_tabs.push_back(ErfTab{E.sigma, E.m0, E.m1, v0, len, at, H.pool[2 * (size_t)at], prev});
    return at;
  };

  // ---- short tier: compact op records (WFK_SH_*), referenced to the first sample of each stretch ----
  // The group stands for  E(t') (A(u) cos th + B(u) sin th),  th = W t' - psi_ref,  u = t' - s_lin.
  // Everything a lane needs to seed the op `koff` samples after the reference sample x_ref:
  //   th/pi = th0p + koff dthp,  u' = koff dt (polynomials re-centred on x_ref),
  //   Gaussian v = v0 + koff H  (exp: alpha t'' = v0 + koff H)
  auto emit_short_piece = [&](const std::vector<FceGroup>& groups, double tshift, int64_t s0, int64_t s1,
                              int32_t& n_rec) -> int32_t {
    const long double PIl = 3.141592653589793238462643383279502884L;
    // ONE carrier (or none) under a table / mollifier envelope -- a pulse as mixing() makes it -- is a single
    // 12-double record: the op adds envelope x carrier itself (word bit 7; bit 8: mollifier) instead of a carrier
    // op followed by the closing multiplier in a record of its own (a dependent load per piece)
    const bool one = groups.size() >= 2 && (groups[1].fmul == 2 || groups[1].fmul == 3) && groups[0].deg == 0 && !groups[0].has_env && !groups[0].has_exp &&
                     !groups[0].erfmul && !groups[0].envmul && !groups[0].chirp && !groups[0].fmul && !env_no_short_cmul;
    // (several envelopes in one piece: the host marked every (group, multiplier) pair -- FceGroup::fmul_own)
    auto own_pair = [&](size_t gi) { return gi + 1 < groups.size() && groups[gi + 1].fmul && (groups[gi + 1].fmul_own || (one && gi == 0)); };
    int32_t rec_len = 0;
    for (size_t gi = 0; gi < groups.size(); ++gi) {
      if (gi > 0 && own_pair(gi - 1)) continue;
      rec_len += (groups[gi].deg > 1 || groups[gi].fmul || groups[gi].chirp || groups[gi].corr) ? WFK_SH_OP3 : WFK_SH_OP1;
    }
    // one carrier (or a constant) under a flat-top edge of at most WFK_SH_ERFTAB samples: ONE own-term op over the edge's
    // sampled table (erf_table above), read at whole knots -- position k, step 1 (not for the FIR chain's sampler plan:
    // fir_short evaluates the erf closing op only)
    if (groups.size() == 2 && groups[1].erfmul && s1 - s0 <= WFK_SH_ERFTAB && !g_no_short_fmul && !env_no_short_erftab &&
        groups[0].deg == 0 && !groups[0].has_env && !groups[0].has_exp && !groups[0].erfmul && !groups[0].envmul && !groups[0].chirp &&
        !groups[0].fmul && !groups[0].corr) {
      const FceGroup& G = groups[0];
      double x = ax.at(s0);
      if (tshift != 0.0) x = x - tshift;
      const long double x0 = x;
      const size_t at = H.params.size();
      H.params.resize(at + (size_t)WFK_SH_OP1, 0.0);
      double* o = H.params.data() + at;
      const uint64_t word = (uint64_t)(uint32_t)(((G.W != 0.0 ? 1 : 0) << 2) | ((G.imag ? 1 : 0) << 3) | (3 << 4) | WFK_SH_LAST | 128) |
                            ((uint64_t)(uint32_t)s0 << 32);
      std::memcpy(&o[0], &word, sizeof word);
      if (G.W != 0.0) {
        const long double th0 = (long double)G.W * x0 - G.psi_ref;
        o[1] = (double)remainderl(th0 / PIl, 2.0L);
        const long double dth = (long double)G.W * (long double)grid->step;
        o[2] = (double)(dth / PIl);
        o[3] = (double)cosl(dth);
        o[4] = (double)sinl(dth);
      } else {
        o[3] = 1.0;
      }
      o[5] = 0.0;                      // knot of the reference sample
      o[6] = 1.0;                      // knots per sample
      o[7] = (double)(s1 - s0 - 1);
      o[8] = (double)G.A[0];
      o[9] = (double)erf_table(groups[1], tshift, s0, s1);       // (H.pool may move: o points into H.params)
      o[10] = (double)G.B[0];
      H.short_has_fmul = true;
      H.short_fam = std::max(H.short_fam, 2);
      n_rec = 1;
      return WFK_SH_OP1;
    }
    n_rec = 0;
    for (int64_t r0 = s0; r0 < s1; r0 += WFK_SH_SUB, ++n_rec) {
      double x = ax.at(r0);
      if (tshift != 0.0) x = x - tshift;                 // fl(x - shift), as the reference forms it
      const long double x0 = x;
      const size_t at = H.params.size();
      H.params.resize(at + (size_t)rec_len, 0.0);
      double* o = H.params.data() + at;
      for (size_t gi = 0; gi < groups.size(); ++gi) {
        const FceGroup& G = groups[gi];
        if (gi > 0 && own_pair(gi - 1)) continue;           // (folded into the record of the group in front of it, below)
        if (G.fmul) {
          // stateless closing multiplier (wfk_short_dev.h: short_tabmul / short_mollmul); the word's degree field
          // (2 | 3) names the kind, so the record is stepped over as a 16-double one.  [5] position at the
          // reference sample and [6] its step per sample, in knot units (table) or in units of r (mollifier);
          // table: [8] m - 1, [9] the table's first entry in the pool (16-byte entries)
          const int32_t f = G.fmul_f;
          const double* fa = P->pool + P->fc_arg_off[f];
          // (chirp multipliers: degree field 2 | 3 as well -- the record is stepped over as a 16-double one -- and bit 10)
          const uint64_t word = (uint64_t)(uint32_t)((G.fmul >= 4 ? (G.fmul - 2) | 1024 : G.fmul) | (3 << 4) | (&G == &groups.back() ? WFK_SH_LAST : 0)) | ((uint64_t)(uint32_t)r0 << 32);
          std::memcpy(&o[0], &word, sizeof word);
          const long double u0 = x0 - (long double)P->fc_shift[f];
          if (G.fmul == 4) {
            // phase / pi = ph0 + scale exp(a_k),  a_k = alpha u0 + (koff + k) alpha dt
            const long double al = fa[1];
            o[5] = (double)(al * u0);
            o[6] = (double)(al * (long double)grid->step);
            o[8] = (double)(2.0L * (long double)fa[0] / al);
            o[9] = (double)remainderl(((long double)fa[2] - 2.0L * PIl * (long double)fa[0] / al) / PIl, 2.0L);
          } else if (G.fmul == 5) {
            // phase / pi = ph0 + scale log(l_k),  l_k = 1 + k u0 + (koff + k) k dt
            const long double kk = fa[1];
            o[5] = (double)(1.0L + kk * u0);
            o[6] = (double)(kk * (long double)grid->step);
            o[8] = (double)(2.0L * (long double)fa[0] / kk);
            o[9] = (double)remainderl((long double)fa[2] / PIl, 2.0L);
          } else if (G.fmul == 2) {
            const int64_t m = P->fc_arg_off[f + 1] - P->fc_arg_off[f] - 2;
            const long double inv = (long double)(m - 1) / ((long double)fa[1] - (long double)fa[0]);
            o[5] = (double)((u0 - (long double)fa[0]) * inv);
            o[6] = (double)((long double)grid->step * inv);
            o[8] = (double)(m - 1);
            o[9] = (double)fmul_table(f);
          } else {
            o[5] = (double)(u0 / (long double)fa[0]);
            o[6] = (double)((long double)grid->step / (long double)fa[0]);
          }
          H.short_has_fmul = true;
          H.short_fam = std::max(H.short_fam, G.fmul >= 4 ? 4 : 2);
          o += WFK_SH_OP3;
          continue;
        }
        if (G.envmul) {
          // closing op of a multi-tone piece: everything accumulated so far *= the Gaussian the tones share
          // (word: closing kind 1; [5] v0, [6] H, [7] q as for an op's own Gaussian)
          const uint64_t word = (uint64_t)(uint32_t)(1 | (3 << 4) | (&G == &groups.back() ? WFK_SH_LAST : 0)) | ((uint64_t)(uint32_t)r0 << 32);
          std::memcpy(&o[0], &word, sizeof word);
          const long double Hh = (long double)grid->step / G.sigma;
          o[5] = (double)((x0 - (long double)G.sg) / G.sigma);
          o[6] = (double)Hh;
          o[7] = (double)expl(-2.0L * Hh * Hh);
          H.short_has_fmul = true;      // (an op fir_short does not evaluate)
          H.short_fam = std::max(H.short_fam, 1);
          o += WFK_SH_OP1;
          continue;
        }
        if (G.erfmul) {
          // closing op of a flat-top edge: everything accumulated so far *= m0 + m1 erf(v), v = v0 + koff H
          const uint64_t word = (uint64_t)(uint32_t)((3 << 4) | (&G == &groups.back() ? WFK_SH_LAST : 0)) | ((uint64_t)(uint32_t)r0 << 32);
          std::memcpy(&o[0], &word, sizeof word);
          o[5] = (double)((x0 - (long double)G.sg) / G.sigma);
          o[6] = (double)((long double)grid->step / G.sigma);
          o[8] = G.m0; o[9] = G.m1;
          H.short_fam = std::max(H.short_fam, 1);
          o += WFK_SH_OP1;
          continue;
        }
        const int env = G.has_exp ? 2 : (G.has_env ? 1 : 0);
        // (a chirp: degree field 2 -- a 16-double record -- and bit 9; its polynomials are of degree <= 1)
        // (a corrected carrier: degree field 2 as well -- a 16-double record --, bit 12, [12..15] w_m, s_m, W, x_ref)
        const uint64_t word = (uint64_t)(uint32_t)(((G.chirp || G.corr) ? 2 : (G.deg & 3)) | (((G.W != 0.0 || G.chirp) ? 1 : 0) << 2) | ((G.imag ? 1 : 0) << 3) | (env << 4) |
                                                   (&G == &groups.back() ? WFK_SH_LAST : 0) | (G.chirp ? 512 : 0) | (G.corr ? 4096 : 0)) |
                              ((uint64_t)(uint32_t)r0 << 32);
        std::memcpy(&o[0], &word, sizeof word);
        if (G.chirp) {
          // phase K t'^2 + W t' - psi_ref at sample k after the reference sample x0:
          //   th0 + k d1 + k^2 d2,  th0 = K x0^2 + W x0 - psi_ref,  d1 = (2 K x0 + W) dt,  d2 = K dt^2
          // [1] th0 / pi (reduced), [2] d1 / pi, [12] d2 / pi, [3] / [4] (cos, sin)(2 d2): the constant the step phasor advances by
          const long double dt = (long double)grid->step;
          const long double xo = x0 - G.corg;           // from the origin of the phase polynomial (the chirp's own shift)
          const long double th0 = G.K * xo * xo + G.Wl * xo - G.psi_ref;
          const long double d1 = (2 * G.K * xo + G.Wl) * dt, d2 = G.K * dt * dt;
          o[1] = (double)remainderl(th0 / PIl, 2.0L);
          o[2] = (double)(d1 / PIl);
          o[12] = (double)(d2 / PIl);
          o[3] = (double)cosl(2 * d2);
          o[4] = (double)sinl(2 * d2);
          H.short_has_fmul = true;      // (an op fir_short does not evaluate: the chain's sampler plan keeps such pieces off the short tier)
          H.short_fam = std::max(H.short_fam, 1);
        } else if (G.W != 0.0) {
          // (a corrected carrier is referenced to a time inside its piece with psi_ref = W_exact s_ref: its phase is
          //  W (x - s_ref) with the ROUNDED W the kernel rotates by -- W x0 - psi_ref would put the rounding of W times
          //  |t| into the seed: 1.3e-8 rad at t = -122 s, the far golden case 31)
          const long double th0 = G.corr ? (long double)G.W * (x0 - (long double)G.sref) : (long double)G.W * x0 - G.psi_ref;
          o[1] = (double)remainderl(th0 / PIl, 2.0L);
          const long double dth = (long double)G.W * (long double)grid->step;
          o[2] = (double)(dth / PIl);
          o[3] = (double)cosl(dth);
          o[4] = (double)sinl(dth);
        } else {
          o[3] = 1.0;
        }
        o[7] = 1.0;
        if (env == 1) {
          const long double Hh = (long double)grid->step / G.sigma;
          o[5] = (double)((x0 - (long double)G.sg) / G.sigma);
          o[6] = (double)Hh;
          o[7] = (double)expl(-2.0L * Hh * Hh);
        } else if (env == 2) {
          o[5] = (double)((long double)G.sigma * (x0 - (long double)G.sg));
          o[6] = (double)((long double)G.sigma * (long double)grid->step);
        }
        // A(u), B(u) about u0 = x_ref - s_lin:  P(u0 + w) = sum_i w^i sum_{m >= i} C(m, i) P_m u0^(m - i)
        const long double u0 = G.has_lin ? x0 - (long double)G.slin : 0.0L;
        static const int binom[4][4] = {{1, 0, 0, 0}, {1, 1, 0, 0}, {1, 2, 1, 0}, {1, 3, 3, 1}};
        long double Ar[4] = {0, 0, 0, 0}, Br[4] = {0, 0, 0, 0};
        const long double upow[4] = {1.0L, u0, u0 * u0, u0 * u0 * u0};
        for (int m = 0; m <= 3; ++m)
          for (int i = 0; i <= m; ++i) {
            const long double f = binom[m][i] * upow[m - i];
            Ar[i] += G.A[m] * f;
            Br[i] += G.B[m] * f;
          }
        o[8] = (double)Ar[0]; o[9] = (double)Ar[1]; o[10] = (double)Br[0]; o[11] = (double)Br[1];
        if (own_pair(gi)) {
          // envelope x carrier in one op (wfk_short_dev.h: short_cmul): the carrier fields as above, [5] / [6] the
          // envelope's position at the reference sample and its step per sample (knot units | units of r),
          // table: [7] m - 1, [9] first entry in the pool (16-byte entries)
          const FceGroup& E = groups[gi + 1];
          const int32_t f = E.fmul_f;
          const double* fa = P->pool + P->fc_arg_off[f];
          const uint64_t w2 = (uint64_t)(uint32_t)(((G.W != 0.0 ? 1 : 0) << 2) | ((G.imag ? 1 : 0) << 3) | (3 << 4) |
                                                   (gi + 2 == groups.size() ? WFK_SH_LAST : 0) | 128 | (E.fmul == 3 ? 256 : 0)) |
                              ((uint64_t)(uint32_t)r0 << 32);
          std::memcpy(&o[0], &w2, sizeof w2);
          const long double ue = x0 - (long double)P->fc_shift[f];
          if (E.fmul == 2) {
            const int64_t m = P->fc_arg_off[f + 1] - P->fc_arg_off[f] - 2;
            const long double inv = (long double)(m - 1) / ((long double)fa[1] - (long double)fa[0]);
            o[5] = (double)((ue - (long double)fa[0]) * inv);
            o[6] = (double)((long double)grid->step * inv);
            o[7] = (double)(m - 1);
            o[9] = (double)fmul_table(f);
          } else {
            o[5] = (double)(ue / (long double)fa[0]);
            o[6] = (double)((long double)grid->step / (long double)fa[0]);
          }
          H.short_has_fmul = true;
          H.short_fam = std::max(H.short_fam, 2);
          o += WFK_SH_OP1;
          continue;
        }
        if (G.corr) {
          o[12] = G.wm; o[13] = G.sm; o[14] = G.W; o[15] = x;
          H.short_corr = true;
          o += WFK_SH_OP3;
        } else if (G.chirp) {
          o += WFK_SH_OP3;
        } else if (G.deg > 1) {
          o[12] = (double)Ar[2]; o[13] = (double)Ar[3]; o[14] = (double)Br[2]; o[15] = (double)Br[3];
          o += WFK_SH_OP3;
        } else {
          o += WFK_SH_OP1;
        }
      }
    }
    return rec_len;
  };

  ptimer.mark("setup");
  // ---- merge members into disjoint device pieces -----------------------------
  bool lean_ok = can_fuse;
  int64_t n_lean_pieces = 0, n_short_pieces = 0, n_foreign_pieces = 0, n_short_samples = 0, n_foreign_samples = 0;
  H.channels.resize(P->n_channels);
  H.channel_complex.assign(P->n_channels, 0);
  for (int32_t c = 0; c < P->n_channels; ++c) {
    DevChannel& C = H.channels[c];
    C.offset = P->ch_offset[c]; C.tshift = P->ch_tshift[c];
    C.clip_lo = P->ch_clip_lo[c]; C.clip_hi = P->ch_clip_hi[c];
    C.do_clip = (C.clip_lo != -INFINITY || C.clip_hi != INFINITY) ? 1 : 0;
    C.pad = 0;
    C.piece_begin = (int32_t)H.pieces.size();
    const int32_t m0 = P->ch_member_off[c], m1 = P->ch_member_off[c + 1];
    std::vector<int64_t> cuts = {0, ax.n};
    for (int32_t m = m0; m < m1; ++m)
      cuts.insert(cuts.end(), H.member_idx[m].begin(), H.member_idx[m].end());
    std::sort(cuts.begin(), cuts.end());
    cuts.erase(std::unique(cuts.begin(), cuts.end()), cuts.end());
    std::vector<int32_t> cur(m1 - m0, 0);  // per member: current piece (relative)
    // (scratch of the piece loop, kept across pieces: a 2 GS/s channel is thousands of pieces, and their heap traffic
    //  was a third of a compile)
    std::vector<int32_t> live, generic, order;
    std::vector<FceGroup> groups, mgroups;
    for (size_t ci = 0; ci + 1 < cuts.size(); ++ci) {
      const int64_t s0 = cuts[ci], s1 = cuts[ci + 1];
      if (s0 < 0 || s1 > ax.n || s0 >= s1) continue;
      // collect live member pieces
      live.clear();
      for (int32_t m = m0; m < m1; ++m) {
        const auto& idx = H.member_idx[m];
        int32_t& k = cur[m - m0];
        while (k < (int32_t)idx.size() - 1 && idx[k] <= s0) ++k;
        if (idx[k] <= s0) continue;  // past the last bound cannot happen (idx.back()==n)
        int32_t p = P->mb_piece_off[m] + k;
        if (P->pc_term_off[p + 1] > P->pc_term_off[p]) live.push_back(p);
      }
      {
        int64_t nt = 0;
        for (int32_t p : live) nt += P->pc_term_off[p + 1] - P->pc_term_off[p];
        jtol = WFK_JITTER_TOL / std::max(1.0, (double)nt);
      }
      DevPiece D{};
      D.start = s0; D.stop = s1; D.n_blk = 0; D.flags = 0; D.first_len = 0; D.pad = 0;
      if (H.params.size() & 1) H.params.push_back(0.0);
      D.par_off = (int64_t)H.params.size();
      // A piece is built at most twice: the corrected (far-from-origin) carriers exist in the lean
      // kernel only, so a piece that turns out NOT to be lean is rebuilt with them on the exact path.
      struct Snap { size_t params, pool; int32_t nf, nd, nu, ng, nc; std::map<int64_t, int64_t> sampled; };
      const Snap snap{H.params.size(), H.pool.size(), H.n_fast, H.n_direct, H.n_fused, H.n_generic, H.n_corr, sampled_at};
      const DevPiece D0 = D;
      bool piece_lean = false;
      for (int attempt = 0; attempt < 2 && !live.empty(); ++attempt) {
        const int32_t corr_before = H.n_corr;
        // (a short plan: in its short pieces only -- the pieces that tier hands on run on the general kernel, which has no chirp op)
        chirp_ok = chirp_base && piece_chirp_ok && can_fuse && (!shortm || (cur_short && !g_no_short_fmul && !env_no_short_chirp));
        D.flags |= WFK_PF_HAS_TERMS;
        BlockBuilder B;
        auto room_for = [&](size_t need) -> int {
          if (WFK_BLK_HDR + need + 2 > WFK_LDS_DOUBLES) return -1;
          if (B.n_terms > 0 && B.size() + need > WFK_LDS_DOUBLES) {
            int32_t len = flush_block(B);
            if (D.n_blk == 0) D.first_len = len;
            ++D.n_blk;
          }
          return 0;
        };
        // pass 1: fuse eligible terms into carrier-envelope groups
        groups.clear();
        generic.clear();
        order.clear();
        for (int32_t p : live)
          for (int32_t k = P->pc_term_off[p]; k < P->pc_term_off[p + 1]; ++k) order.push_back(k);
        if (can_fuse && corr_enabled) {
          // where a carrier needs the rounding correction the HEAVIEST term must found the group
          // (its COS factor is the one mimicked): visit the terms by descending amplitude there
          bool any = false;
          for (int32_t k : order)
            for (int32_t f = P->tm_factor_off[k]; f < P->tm_factor_off[k + 1] && !any; ++f)
              any = P->fc_type[f] == WFK_COS && !rate_safe_n(P->pool[P->fc_arg_off[f]], s0, s1);
          if (any) {
            auto weight_of = [&](int32_t k) {   // |amp| * max |LINEAR factors| over the piece
              double wgt = std::hypot(P->tm_amp_re[k], P->tm_amp_im[k]);
              for (int32_t f = P->tm_factor_off[k]; f < P->tm_factor_off[k + 1]; ++f)
                if (P->fc_type[f] == WFK_LINEAR) {
                  const double ua = (ax.at(s0) - C.tshift) - P->fc_shift[f], ub = (ax.at(s1 - 1) - C.tshift) - P->fc_shift[f];
                  wgt *= std::pow(std::max(std::fabs(ua), std::fabs(ub)), P->fc_power[f]);
                }
              return wgt;
            };
            std::stable_sort(order.begin(), order.end(),
                             [&](int32_t x, int32_t y) { return weight_of(x) > weight_of(y); });
          }
        }
        // Terms with ONE erf factor (the edge of a flat-top pulse: square(width, edge), 0.5 + 0.5 erf)
        // whose other factors fuse: the rest goes into a second group list, and a closing op multiplies
        // what those groups accumulated by the erf (advanced by its own Taylor step, see emit_group);
        // the unmodulated groups follow it.  One erf (sigma, shift) per piece.
        mgroups.clear();
        bool mod_on = false;
        double mod_sigma = 0, mod_shift = 0;
        int mod_kind = 0;             // 1: erf edge, 2: INTERP table, 3: mollifier (the closing multiplier of this piece)
        int32_t mod_f = -1;           // the first term's factor of that kind (later terms must carry an equal one)
        // (a short plan: in its short pieces only -- the pieces that tier hands on go to the general kernel)
        const bool fmul_now = fmul_base && piece_fmul_ok && (!shortm || (cur_short && !g_no_short_fmul)) && std::isfinite(ax.at(s0)) && std::isfinite(ax.at(s1 - 1));
        // a term's ONE factor of a kind the closing multipliers take (power 1); -1: none, or not admissible here
        // Short pieces may hold SEVERAL envelopes (overlapping pulses of different shapes: crosstalk-compensated channels),
        // as long as each stands over one plain carrier: every such term is an own-term op, acc += F (A cos + B sin).
        struct XMod { int kind; int32_t f; std::vector<FceGroup> g; };
        std::vector<XMod> xmods;      // the envelopes after the first one
        const bool multi_ok = !(cur_short ? env_no_short_multi : env_no_lean_multi);
        auto same_mod = [&](int32_t a_, int32_t b_) {
          const int64_t na = P->fc_arg_off[a_ + 1] - P->fc_arg_off[a_];
          if (P->fc_type[a_] != P->fc_type[b_] || P->fc_shift[a_] != P->fc_shift[b_] || na != P->fc_arg_off[b_ + 1] - P->fc_arg_off[b_]) return false;
          return P->fc_arg_off[a_] == P->fc_arg_off[b_] ||
                 std::memcmp(P->pool + P->fc_arg_off[a_], P->pool + P->fc_arg_off[b_], (size_t)na * sizeof(double)) == 0;
        };
        int fslot = 0;                // fmul_factor_of: 0 the piece's first envelope, i + 1: xmods[i], xmods.size() + 1: a new one
        auto fmul_factor_of = [&](int32_t k, int& kind_out) -> int32_t {
          if (!fmul_now || mod_kind == 1) return -1;
          fslot = 0;
          int32_t at = -1;
          for (int32_t f = P->tm_factor_off[k]; f < P->tm_factor_off[k + 1]; ++f)
            if (P->fc_type[f] == WFK_INTERP || P->fc_type[f] == WFK_MOLLIFIER ||
                (cur_short && !env_no_short_xchirp && (P->fc_type[f] == WFK_EXPONENTIALCHIRP || P->fc_type[f] == WFK_HYPERBOLICCHIRP))) {
              if (at >= 0 || P->fc_power[f] != 1.0) return -1;
              at = f;
            } else if (P->fc_type[f] == WFK_ERF) return -1;
          if (at < 0) return -1;
          const double* fa = P->pool + P->fc_arg_off[at];
          const int64_t na = P->fc_arg_off[at + 1] - P->fc_arg_off[at];
          const double sh = P->fc_shift[at];
          if (!std::isfinite(sh)) return -1;
          const int kind = P->fc_type[at] == WFK_INTERP ? 2 : (P->fc_type[at] == WFK_MOLLIFIER ? 3 :
                           (P->fc_type[at] == WFK_EXPONENTIALCHIRP ? 4 : 5));
          if (kind >= 4) {
            // Exponential / hyperbolic chirps at AWG rates (short pieces only): sin(phi0 + 2 pi f0 (exp(alpha u) - 1) / alpha) and
            // sin(phi0 + 2 pi f0 / k log(1 + k u)) (reference _waveform.pyx:326-332) have no recurrence form -- the phase does not
            // advance by a constant -- but as a MULTIPLIER of what the piece's other factors fuse to they cost one inline
            // exp step + one sine per sample (wfk_short_dev.h: short_xchirpmul) instead of a trip through the term interpreter
            // with device libm (one sample per lane: 10.6 ms for 2048 x 1e5 at 2 GS/s).  One such factor per piece.
            if (mod_f >= 0) { if (same_mod(at, mod_f)) { kind_out = kind; return at; } return -1; }
            for (int i = 0; i < 3; ++i)
              if (!std::isfinite(fa[i])) return -1;
            const double ua = (ax.at(s0) - C.tshift) - sh, ub = (ax.at(s1 - 1) - C.tshift) - sh;
            double wmax;
            if (kind == 4) {
              if (fa[1] == 0.0 || !(std::fabs(fa[1]) * std::max(std::fabs(ua), std::fabs(ub)) <= 600.0)) return -1;
              wmax = 6.283185307179586 * std::fabs(fa[0]) * std::exp(std::max(fa[1] * ua, fa[1] * ub));
            } else {
              const double la = 1.0 + fa[1] * ua, lb = 1.0 + fa[1] * ub;
              if (fa[1] == 0.0 || !(la > 1e-6) || !(lb > 1e-6)) return -1;
              wmax = 6.283185307179586 * std::fabs(fa[0]) / std::min(la, lb);
            }
            // (the phase itself is evaluated in double: its size times 2^-52 must stay inside the budget too)
            const double phmax = kind == 4 ? 6.283185307179586 * std::fabs(fa[0] / fa[1]) * (1.0 + std::exp(std::max(fa[1] * ua, fa[1] * ub)))
                                           : 6.283185307179586 * std::fabs(fa[0] / fa[1]) * std::max(std::fabs(std::log(1.0 + fa[1] * ua)), std::fabs(std::log(1.0 + fa[1] * ub)));
            if (!std::isfinite(wmax) || !rate_safe_n(wmax, s0, s1) || !(4.5e-16 * (phmax + std::fabs(fa[2])) <= jtol)) return -1;
            kind_out = kind;
            return at;
          }
          if (mod_f >= 0) {
            // one multiplier per piece: the same primitive, arguments and shift as the first term's
            if (same_mod(at, mod_f)) { kind_out = kind; return at; }
            if (!multi_ok) return -1;
            for (size_t i = 0; i < xmods.size(); ++i)
              if (same_mod(at, xmods[i].f)) { fslot = (int)i + 1; kind_out = kind; return at; }
            fslot = (int)xmods.size() + 1;      // a new envelope: admitted below
          }
          if (kind == 2) {
            // a finite table on increasing linspace knots, slopes inside the grid-rounding budget (as WFK_M_INTERP_LIN)
            const int64_t m = na - 2;
            const double start = fa[0], stop = fa[1];
            if (m < 2 || m > (int64_t(1) << 24) || !(stop > start) || !std::isfinite(start) || !std::isfinite(stop)) return -1;
            const double inv = (double)(m - 1) / (stop - start);
            if (!std::isfinite(inv)) return -1;
            double dmax = 0.0;
            for (int64_t j = 0; j < m; ++j) {
              if (!std::isfinite(fa[2 + j])) return -1;
              if (j + 1 < m) dmax = std::max(dmax, std::fabs(fa[3 + j] - fa[2 + j]));
            }
            if (!std::isfinite(dmax) || !rate_safe(4.0 * dmax * inv, s0, s1)) return -1;
            // the position in knot units, q = (x - start) * inv advanced by a tile's worth of additions: its
            // rounding (<= 2.5e-15 of the index range) times the largest step
            if (dmax * (double)m * 2.5e-15 > WFK_JITTER_TOL) return -1;
          } else {
            if (!(fa[1] == 0.0) || !std::isfinite(fa[0]) || !(fa[0] > 0.0) || !rate_safe(4.0 / fa[0], s0, s1)) return -1;
          }
          kind_out = kind;
          return at;
        };
        auto erf_factor_of = [&](int32_t k, double& sg_out, double& sh_out) -> int32_t {
          if (!erfmod_base || mod_kind >= 2) return -1;
          int32_t at = -1;
          for (int32_t f = P->tm_factor_off[k]; f < P->tm_factor_off[k + 1]; ++f)
            if (P->fc_type[f] == WFK_ERF) {
              if (at >= 0 || P->fc_power[f] != 1.0) return -1;
              at = f;
            }
          if (at < 0) return -1;
          const double sg = P->pool[P->fc_arg_off[at]], sh = P->fc_shift[at];
          if (!std::isfinite(sg) || sg == 0.0 || !std::isfinite(sh)) return -1;
          if (!rate_safe(1.13 / sg, s0, s1)) return -1;
          if (!cur_short) {
            // lean kernel: the erf advances by its own Taylor step along the lane stride (fine grids only);
            // the short tier evaluates erf itself at every sample of the (short) edge piece
            if (!(std::fabs(dstride / sg) <= 0.09)) return -1;
            bool ok64, ok32;
            gauss_range(sg, sh, C.tshift, s0, s1, ok64, ok32);
            if (!ok64) return -1;
          }
          if (mod_on && (mod_sigma != sg || mod_shift != sh)) return -1;
          sg_out = sg; sh_out = sh;
          return at;
        };
        for (int32_t k : order) {
          if (P->tm_amp_im[k] != 0.0) H.channel_complex[c] = 1;
          double esg = 0, esh = 0;
          const bool fuse_now = can_fuse && piece_fuse_ok;
          const int32_t fe = fuse_now ? erf_factor_of(k, esg, esh) : -1;
          int fkind = 0;
          const int32_t ff = fuse_now && fe < 0 ? fmul_factor_of(k, fkind) : -1;
          if (fe >= 0 && fuse_term(mgroups, k, C.tshift, s0, s1, fe)) {
            mod_on = true; mod_sigma = esg; mod_shift = esh; mod_kind = 1;
            ++H.n_fused;
          } else if (ff >= 0 && fslot == 0 && fuse_term(mgroups, k, C.tshift, s0, s1, ff)) {
            mod_on = true; mod_kind = fkind;
            if (mod_f < 0) mod_f = ff;
            ++H.n_fused;
          } else if (ff >= 0 && fslot >= 1 && fslot <= (int)xmods.size() && fuse_term(xmods[(size_t)fslot - 1].g, k, C.tshift, s0, s1, ff)) {
            ++H.n_fused;
          } else if (ff >= 0 && fslot == (int)xmods.size() + 1 && [&] {
                       XMod x{fkind, ff, {}};
                       if (!fuse_term(x.g, k, C.tshift, s0, s1, ff)) return false;
                       xmods.push_back(std::move(x));
                       return true;
                     }()) {
            ++H.n_fused;
          } else if (fuse_now && fuse_term(groups, k, C.tshift, s0, s1)) ++H.n_fused;
          else generic.push_back(k);
        }
        if (H.tlist && !generic.empty() && !groups.empty() && attempt == 0) {
          // Time lists evaluate fused ops pointwise in a build of the general kernel that holds nothing else
          // (with every libm shape of the direct tier next to them the kernel spills: measured 1.2 -> 2.3 ms on
          // flat-top pulses).  A piece is therefore either fused completely or not at all; a plan with both
          // kinds runs as two launches over disjoint pieces (H.mixed).
          H.params.resize(snap.params); H.pool.resize(snap.pool);
          H.n_fast = snap.nf; H.n_direct = snap.nd; H.n_fused = snap.nu; H.n_generic = snap.ng; H.n_corr = snap.nc;
          sampled_at = snap.sampled;
          D = D0;
          piece_fuse_ok = false;
          continue;
        }
        bool multi_bad = false;
        if (mod_on && mod_kind >= 2 && !xmods.empty()) {
          // several envelopes: each over ONE plain group (degree 0, no envelope of its own), emitted as own-term ops
          auto simple = [](const std::vector<FceGroup>& g) {
            return g.size() == 1 && g[0].deg == 0 && !g[0].has_env && !g[0].has_exp && !g[0].chirp && !g[0].corr;
          };
          multi_bad = !simple(mgroups);
          for (const XMod& x : xmods) multi_bad = multi_bad || !simple(x.g);
          std::vector<FceGroup> all;
          if (!multi_bad) {
            FceGroup E;
            E.fmul = mod_kind; E.fmul_f = mod_f; E.fmul_own = true;
            all.push_back(mgroups[0]); all.push_back(E);
            for (const XMod& x : xmods) {
              FceGroup Ex;
              Ex.fmul = x.kind; Ex.fmul_f = x.f; Ex.fmul_own = true;
              all.push_back(x.g[0]); all.push_back(Ex);
            }
            all.insert(all.end(), groups.begin(), groups.end());
            groups.swap(all);
          }
        } else if (mod_on && mod_kind >= 2) {
          // out = F S1 + S0: the groups of the modulated terms, the closing multiplier, then the rest
          FceGroup E;
          E.fmul = mod_kind; E.fmul_f = mod_f;
          mgroups.push_back(E);
          mgroups.insert(mgroups.end(), groups.begin(), groups.end());
          groups.swap(mgroups);
        } else if (mod_on) {
          // out = S0 + erf S1.  Where every group of S1 has a twin in S0 with the same coefficients
          // times ONE ratio rho (0.5 cos + 0.5 erf cos), the twins go: out = (rho + erf) S1 + rest.
          FceGroup E;
          E.erfmul = true; E.sigma = mod_sigma; E.sg = mod_shift; E.m0 = 0.0; E.m1 = 1.0;
          std::vector<int> twin(mgroups.size(), -1);
          bool all = !groups.empty();
          long double rho = 0;
          for (size_t i = 0; i < mgroups.size() && all; ++i) {
            const FceGroup& m = mgroups[i];
            int big = -1;                       // the largest coefficient of the modulated group
            long double bigv = 0;
            for (int j = 0; j < 8; ++j) {
              const long double v = fabsl(j < 4 ? m.A[j] : m.B[j - 4]);
              if (v > bigv) { bigv = v; big = j; }
            }
            if (big < 0) { all = false; break; }
            for (size_t j = 0; j < groups.size() && twin[i] < 0; ++j) {
              const FceGroup& g = groups[j];
              if (std::find(twin.begin(), twin.end(), (int)j) != twin.end()) continue;
              if (g.W != m.W || g.imag != m.imag || g.has_env != m.has_env || g.has_exp != m.has_exp || g.sigma != m.sigma || g.sg != m.sg ||
                  g.sref != m.sref || g.psi_ref != m.psi_ref || g.has_lin != m.has_lin || g.slin != m.slin ||
                  g.deg != m.deg || g.corr != m.corr || g.wm != m.wm || g.sm != m.sm)
                continue;
              const long double r = (big < 4 ? g.A[big] : g.B[big - 4]) / (big < 4 ? m.A[big] : m.B[big - 4]);
              bool same = true;
              for (int q = 0; q < 8 && same; ++q) {
                const long double gv = q < 4 ? g.A[q] : g.B[q - 4], mv = q < 4 ? m.A[q] : m.B[q - 4];
                same = fabsl(gv - r * mv) <= 1e-14L * fabsl(r) * bigv;
              }
              if (same && (i == 0 || fabsl(r - rho) <= 1e-14L * fabsl(rho))) {
                if (i == 0) rho = r;
                twin[i] = (int)j;
              }
            }
            all = twin[i] >= 0;
          }
          if (all && std::isfinite((double)rho)) {
            std::vector<bool> gone(groups.size(), false);
            for (int j : twin) gone[(size_t)j] = true;
            std::vector<FceGroup> rest;
            for (size_t j = 0; j < groups.size(); ++j)
              if (!gone[j]) rest.push_back(groups[j]);
            groups.swap(rest);
            E.m0 = (double)rho;
          }
          mgroups.push_back(E);
          mgroups.insert(mgroups.end(), groups.begin(), groups.end());
          groups.swap(mgroups);
        }
        // When every carrier of the piece sits under the SAME Gaussian (a frequency-multiplexed
        // pulse), the envelope is factored out: the ops run without envelope and one closing
        // pseudo-op multiplies the accumulators by it -- 2 instead of 5 FMAs per sample and tone.
        // (From four carriers on: the extra op costs a pair of pieces what it saves them.)
        // (short pieces too: a tone costs a phasor seed and 7 instructions per sample there instead of the seeds of its own
        //  Gaussian and 13; not for the FIR chain's sampler plan, whose fir_short does not know the closing op)
        if (groups.size() >= 4 && !mod_on && (!cur_short || (!g_no_short_fmul && !env_no_short_envmul))) {
          bool shared = true, e32 = true;
          for (const FceGroup& g : groups) {
            shared = shared && g.has_env && !g.has_exp && g.sigma == groups[0].sigma && g.sg == groups[0].sg;
            e32 = e32 && g.env32;
          }
          if (shared) {
            FceGroup E;
            E.envmul = true; E.has_env = true; E.sigma = groups[0].sigma; E.sg = groups[0].sg; E.env32 = e32;
            for (FceGroup& g : groups) g.has_env = false;
            groups.push_back(E);
          }
        }
        // Runs of >= 4 bare carriers (the tones of a multiplexed pulse once their envelope is factored out, CW
        // tones): marked for the lean kernel's compact tone loop -- per tone and tile two state reads, the products
        // with the amplitude, the phasor advance; no per-op dispatch (wfk_kernels.hip: fce_bank)
        bool piece_has_bank = false;
        if (!cur_short && !H.tlist && ns_override == 0 && !env_no_bank) {
          auto bare = [](const FceGroup& g) {
            return g.W != 0.0 && !g.chirp && g.deg == 0 && !g.has_env && !g.has_exp && !g.envmul && !g.erfmul && !g.fmul && !g.corr;
          };
          for (size_t i = 0; i < groups.size();) {
            size_t j = i;
            while (j < groups.size() && bare(groups[j]) && groups[j].imag == groups[i].imag) ++j;
            if (j - i >= 4) { groups[i].bank = (int)(j - i); piece_has_bank = true; }
            i = j > i ? j : i + 1;
          }
        }
        if (cur_short) {
          // compact records of the short tier (WFK_SH_*): one per <= WFK_SH_SUB samples of the piece.  A
          // piece that tier cannot take (generic terms: erf edges, chirps, ...) is built again for the
          // general kernel, in the standard geometry: the plan then runs as two launches (mixed).
          bool ok = generic.empty() && !groups.empty() && groups.size() <= 255 && !multi_bad;
          for (const FceGroup& G : groups)
            ok = ok && !(G.corr && (G.deg > 1 || G.chirp || G.fmul || G.erfmul || G.envmul || C.tshift != 0.0)) && !(G.chirp && G.deg > 1);
          if (!ok) {
            H.params.resize(snap.params); H.pool.resize(snap.pool);
            H.n_fast = snap.nf; H.n_direct = snap.nd; H.n_fused = snap.nu; H.n_generic = snap.ng; H.n_corr = snap.nc;
            sampled_at = snap.sampled;
            D = D0;
            set_geom(false);
            ++n_foreign_pieces;
            n_foreign_samples += s1 - s0;
            continue;
          }
          const int32_t first = emit_short_piece(groups, C.tshift, s0, s1, D.n_blk);
          D.first_len = first;
          D.flags |= WFK_PF_SHORT;
          ++n_short_pieces;
          n_short_samples += s1 - s0;
          break;
        }
        piece_lean = generic.empty() && !groups.empty() && groups.size() <= WFK_LEAN_OPS;
        bool piece_has_chirp = false, piece_has_fmul = false;
        int piece_fam = 0;          // lean kernel family the piece needs: 0 plain ops, 1 closing ops, 2 chirps, 3 stateless multipliers
        for (const FceGroup& G : groups) {
          piece_has_chirp = piece_has_chirp || G.chirp;
          piece_has_fmul = piece_has_fmul || G.fmul != 0;     // (own multipliers are still separate entries here: fmul set)
          piece_fam = std::max(piece_fam, G.fmul ? 3 : (G.chirp ? 2 : ((G.erfmul || G.envmul) ? 1 : 0)));
        }
        for (size_t gi = 0; gi + 1 < groups.size(); ++gi)
          if (!cur_short && groups[gi + 1].fmul && groups[gi + 1].fmul_own) piece_fam = 4;     // own-term ops: family 4
        if (piece_has_bank) piece_fam = std::max(piece_fam, 1);
        plan_has_bank = plan_has_bank || piece_has_bank;
        const int32_t piece_ops = (int32_t)groups.size();
        int32_t piece_units = 0;
        if (!cur_short) {
          // (lean pieces: a (group, own multiplier) pair is ONE op -- the multiplier's parameters ride in the group's record)
          std::vector<FceGroup> folded;
          for (size_t gi = 0; gi < groups.size(); ++gi) {
            if (gi + 1 < groups.size() && groups[gi + 1].fmul && groups[gi + 1].fmul_own) {
              FceGroup g = groups[gi];
              g.own_kind = groups[gi + 1].fmul; g.own_f = groups[gi + 1].fmul_f;
              folded.push_back(g);
              ++gi;
            } else {
              folded.push_back(groups[gi]);
            }
          }
          groups.swap(folded);
        }
        for (FceGroup& G : groups) {
          if (room_for(WFK_FCE_REC + 2 * (NS + 1)) < 0) { err = "LDS parameter buffer too small"; return WFK_EINVAL; }
          if (H.tlist) {
            const double ta = ax.at(s0) - C.tshift, tb = ax.at(s1 - 1) - C.tshift;
            G.tl_thmax = std::fabs(G.W) * std::max(std::fabs(ta - G.sref), std::fabs(tb - G.sref)) + 4.0;
          }
          emit_group(B, G);
          ++B.n_terms;
          piece_units = B.state_units;
        }
        // pass 2: whatever is left, factor by factor
        for (int32_t k : generic) {
          const int32_t f0 = P->tm_factor_off[k], f1 = P->tm_factor_off[k + 1];
          // conservative size of this term: header + records + one table per COS
          size_t need = WFK_TERM_HDR + (size_t)(f1 - f0) * (WFK_FREC + 2 * (NS + 1));
          if (room_for(need) < 0) {
            err = "a single term with " + std::to_string(f1 - f0) + " factors exceeds the LDS parameter buffer";
            return WFK_EINVAL;
          }
          B.body.push_back((double)WFK_OP_TERM);
          B.body.push_back(P->tm_amp_re[k]);
          B.body.push_back(P->tm_amp_im[k]);
          B.body.push_back((double)(f1 - f0));
          for (int32_t f = f0; f < f1; ++f) emit_factor(B, f, C.tshift, s0, s1);
          ++B.n_terms;
          ++H.n_generic;
        }
        int32_t len = flush_block(B);
        if (D.n_blk == 0) D.first_len = len;
        ++D.n_blk;
        piece_lean = piece_lean && D.n_blk == 1 && len <= lean_par_cap && piece_units <= 63;
        (void)piece_ops;
        if ((!piece_lean && (H.n_corr > corr_before || piece_has_chirp || piece_has_fmul) && attempt == 0) ||
            ((multi_bad || (piece_has_fmul && H.n_corr > corr_before)) && attempt == 0)) {
          // roll back and build the piece again without corrected carriers / fused chirps (lean kernel only)
          H.params.resize(snap.params); H.pool.resize(snap.pool);
          H.n_fast = snap.nf; H.n_direct = snap.nd; H.n_fused = snap.nu; H.n_generic = snap.ng; H.n_corr = snap.nc;
          sampled_at = snap.sampled;
          D = D0;
          piece_corr_ok = false;
          piece_chirp_ok = false;
          piece_fmul_ok = false;
          continue;
        }
        if (H.tlist) {
          // time lists: "lean" = the piece consists of fused ops only (any number of blocks): pointwise build
          if (generic.empty() && !groups.empty()) { D.flags |= WFK_PF_LEAN; ++n_lean_pieces; }
          else lean_ok = false;
        } else if (piece_lean && !shortm) {
          H.lean_fam = std::max(H.lean_fam, piece_fam);   // (a short plan has no lean launch: its other pieces all go to the general kernel)
          D.flags |= WFK_PF_LEAN;
          ++n_lean_pieces;
          H.lean_ops = std::max<int32_t>(H.lean_ops, piece_units);
          H.lean_par = std::max<int32_t>(H.lean_par, len);
        } else {
          lean_ok = false;
        }
        break;
      }
      piece_corr_ok = true;
      piece_chirp_ok = true;
      piece_fmul_ok = true;
      piece_fuse_ok = true;
      if (shortm) set_geom(true);
      // fuse adjacent zero pieces
      if (D.n_blk == 0 && (int32_t)H.pieces.size() > C.piece_begin &&
          H.pieces.back().n_blk == 0 && H.pieces.back().stop == s0) {
        H.pieces.back().stop = s1;
      } else {
        H.pieces.push_back(D);
      }
    }
    C.piece_end = (int32_t)H.pieces.size();
  }

  ptimer.mark("pieces+fusion");
  // ---- workgroup chunking ------------------------------------------------------
  // general kernel: workgroup = 4 waves, tile = 256*NS samples, chunk = tiles_per_chunk tiles
  // lean kernel   : workgroup = 1 wave,  tile = 64*NS samples (a wave owns a contiguous span)
  H.lean = lean_ok && !nolean && H.n_fused > 0 && !H.tlist;     // (time lists: fused ops run pointwise in the general kernel)
  // mixed plans: some pieces are lean, some are not (the erf edges of a flat-top pulse next to its
  // multi-tone plateau).  Two launches over the same output: the lean kernel takes the lean and the
  // zero pieces, the general kernel the rest -- every sample is still written exactly once.
  const char* nomix_env = std::getenv("WFK_DISABLE_MIXED");
  H.mixed = !H.lean && !H.tlist && can_fuse && !nolean && n_lean_pieces > 0 && ns_override == 0 &&
            !(nomix_env && nomix_env[0] == '1');
  // time lists: fully fused pieces on the pointwise build, the others on the build with the direct tier
  // (same chunking for both launches); all of one kind: a single launch of that build
  const bool tl_mixed = H.tlist && n_lean_pieces > 0 && !lean_ok;
  if (tl_mixed) H.mixed = true;
  H.lean_par = std::max(256, (H.lean_par + 63) / 64 * 64);      // >= the 2 KB every plan had so far
  const int32_t lean_units_max = H.lean_ops;                     // most state units of a lean piece (phasors + envelopes)
  H.lean_ops = std::max(8, H.lean_ops);                          // likewise: 8 units = 8 KB of state
  auto chunking = [&](bool lean_geom, int32_t& tile, int32_t& tiles_per_chunk, int64_t& chunks_per_ch,
                      std::vector<int32_t>& chunk_first, int lean_cap = WFK_LEAN_TPC, int64_t lean_div = 2048) {
    // general kernel: workgroup = 4 waves, tile = 256*NS samples, chunk = tiles_per_chunk tiles
    // lean kernel   : workgroup = 1 wave,  tile = 64*NS samples (a wave owns a contiguous span)
    tile = (lean_geom ? 64 : WFK_WG) * H.ns;
    const int64_t tiles_per_ch = (ax.n + tile - 1) / tile;
    const int64_t total_tiles = tiles_per_ch * (g_block_total_channels > 0 ? g_block_total_channels : P->n_channels);
    // lean: one wave per workgroup; ~2-3k workgroups already fill 256 CUs x 12 waves.  Longer
    // chunks amortise the exact seeds, shorter ones keep the set of regions being written at
    // any moment compact, which is what the HBM write rate depends on:
    // measured best at 8 tiles (= one seed per chunk) on the headline config while a degree-1 op cost
    // 12 instructions per sample, at 5 since the phasor fold (8: same box 3.19 / 6: 3.11 / 5: 3.07 / 4: 3.16 ms);
    // 4 on C2.
    // Pieces of many ops (multi-tone pulses): a chunk's first tile seeds EVERY op exactly (libm: ~200 instructions
    // each), which at 5 tiles per chunk is as much work as the ops' own arithmetic -- longer chunks there
    // (ten tones per pulse, same box: 5 / 10 / 20 / 40 tiles 2.16 / 1.95 / 1.87 / 1.84 ms; four tones 1.34 / 1.27 / 1.22)
    if (lean_geom && lean_cap == WFK_LEAN_TPC && lean_units_max >= 5 && plan_has_bank) lean_cap = 20;
    const int64_t tpc = total_tiles / (lean_geom ? lean_div : 8192);
    tiles_per_chunk = (int32_t)std::min<int64_t>(lean_geom ? lean_cap : 16, std::max<int64_t>(1, tpc));
    if (const char* e = std::getenv("WFK_TPC")) {   // tuning override
      int v = std::atoi(e);
      if (v >= 1 && v <= 64) tiles_per_chunk = v;
    }
    chunks_per_ch = (tiles_per_ch + tiles_per_chunk - 1) / tiles_per_chunk;
    chunk_first.assign((size_t)(chunks_per_ch * P->n_channels), 0);
    const int64_t chunk_samples = (int64_t)tiles_per_chunk * tile;
    for (int32_t c = 0; c < P->n_channels; ++c) {
      int32_t p = H.channels[c].piece_begin;
      for (int64_t k = 0; k < chunks_per_ch; ++k) {
        int64_t g0 = k * chunk_samples;
        while (p < H.channels[c].piece_end - 1 && H.pieces[p].stop <= g0) ++p;
        chunk_first[(size_t)(c * chunks_per_ch + k)] = p;
      }
    }
  };
  ptimer.mark("chunking");
  // ---- short tier: wave units and lane slots ------------------------------------
  if (shortm) {
    // Mostly pieces the short tier cannot take (e.g. carriers that need the lean kernel's grid-rounding
    // correction far from t = 0): the standard tiers serve the whole plan better
    if (n_foreign_samples > n_short_samples || (n_short_pieces == 0 && n_foreign_pieces > 0)) return WFK_RETRY_STD;
    H.shortp = true;
    H.lean = false;
    // corrected carriers live in family 6 = family 0 + the correction: with closing ops, chirps or tables in the same plan
    // the carriers go back to the tiers that have both
    if (H.short_corr && H.short_fam != 0) H.short_needs_corr = true;
    if (H.short_corr && H.short_fam == 0) H.short_fam = 6;
    H.mixed = n_foreign_pieces > 0;       // foreign pieces: a second launch of the general kernel
    H.foreign_frac = n_short_samples > 0 ? (double)n_foreign_samples / (double)(n_short_samples + n_foreign_samples) : 0.0;
    H.tile = 64 * WFK_SH_R;
    for (int32_t c = 0; c < P->n_channels; ++c) {
      ShortUnit U{};
      auto fresh = [&](int64_t j0) {
        U = ShortUnit{};
        U.ch = c; U.j0 = j0; U.slot0 = (int32_t)H.s_slots.size(); U.rec0 = -1;
        U.offset = H.channels[c].offset; U.clip_lo = H.channels[c].clip_lo; U.clip_hi = H.channels[c].clip_hi;
        U.do_clip = H.channels[c].do_clip;
      };
      fresh(0);
      auto close = [&](int64_t next_j0) {
        if (U.n_samples > 0) {
          if (U.n_slots > 0) H.s_lds_samples = std::max(H.s_lds_samples, U.n_samples);
          if (U.rec0 < 0) U.rec0 = 0;
          if (U.n_slots > 0) {
            // LDS staging layout of this unit (wfk_short.hip): plain or padded by one element per 16,
            // whichever spreads the lanes' first elements over more of the 16 bank pairs
            int plain[16] = {0}, padded[16] = {0}, wp = 0, wq = 0;
            for (int32_t k = 0; k < U.n_slots; ++k) {
              const int o = (int)((H.s_slots[(size_t)U.slot0 + k] >> 16) & 0x3ff);
              wp = std::max(wp, ++plain[o & 15]);
              wq = std::max(wq, ++padded[(o + (o >> 4)) & 15]);
            }
            if (wq < wp) U.gaps |= 2;
          }
          H.s_units.push_back(U);
        }
        fresh(next_j0);
      };
      for (int32_t pi = H.channels[c].piece_begin; pi < H.channels[c].piece_end; ++pi) {
        const DevPiece& D = H.pieces[pi];
        if (D.n_blk != 0 && !(D.flags & WFK_PF_SHORT)) {   // the general kernel's piece: no unit covers it
          close(D.stop);
          continue;
        }
        if (D.n_blk == 0) {
          // zero stretch: rides in the current unit's range while it fits, long ones as pure-fill units
          int64_t z0 = D.start, left = D.stop - D.start;
          if (U.n_slots > 0) {
            const int64_t take = std::min<int64_t>(left, WFK_SH_LCAP - U.n_samples);
            U.n_samples += (int32_t)take; U.gaps |= 1;
            z0 += take; left -= take;
            if (left > 0) close(z0);
          }
          while (left >= WFK_SH_LCAP / 2) {
            const int64_t take = std::min<int64_t>(left, WFK_SH_FILL);
            if (U.n_samples > 0) close(z0);
            U.n_samples = (int32_t)take;
            z0 += take; left -= take;
            close(z0);
          }
          if (left > 0) { U.n_samples += (int32_t)left; U.gaps |= 1; }
          continue;
        }
        int32_t rec = 0;
        for (int64_t r0 = D.start; r0 < D.stop; r0 += WFK_SH_SUB, ++rec) {
          const int64_t len = std::min<int64_t>(WFK_SH_SUB, D.stop - r0);
          const int64_t nseg = (len + WFK_SH_R - 1) / WFK_SH_R, base = len / nseg, rem = len % nseg;
          const int64_t rec16 = (D.par_off + (int64_t)rec * D.first_len) / 2;
          int64_t k0 = 0;
          for (int64_t sgi = 0; sgi < nseg; ++sgi) {
            const int64_t sl = base + (sgi < rem ? 1 : 0);
            if (U.n_slots == 64 || U.n_samples + sl > WFK_SH_LCAP ||
                (U.rec0 >= 0 && rec16 - U.rec0 > WFK_SH_DREC_MAX)) close(r0 + k0);
            if (U.rec0 < 0) U.rec0 = rec16;
            H.s_slots.push_back(WFK_SH_SLOT(rec16 - U.rec0, U.n_samples, sl));
            ++U.n_slots;
            U.n_samples += (int32_t)sl;
            k0 += sl;
          }
        }
      }
      close(ax.n);
    }
    if (H.s_slots.empty()) H.s_slots.push_back(0);
    // workgroup = one wave walking `units_per_chunk` consecutive units
    const int64_t nu = (int64_t)H.s_units.size();
    H.s_units_per_chunk = (int32_t)std::min<int64_t>(6, std::max<int64_t>(1, nu / 8192));
    if (const char* e = std::getenv("WFK_SH_UPC")) {   // tuning override
      const int v = std::atoi(e);
      if (v >= 1 && v <= 64) H.s_units_per_chunk = v;
    }
    H.chunks_per_ch = 0;
    H.pool_real = !H.pool.empty();
    if (H.pool.empty()) H.pool.push_back(0.0);
    H.params.resize(H.params.size() + 16, 0.0);   // (the kernel reads one op record past the last real one)
    if (H.mixed) {
      set_geom(false);
      H.ns = WFK_NS_GRID;
      chunking(false, H.tile, H.tiles_per_chunk, H.chunks_per_ch, H.chunk_first);
    }
    return WFK_OK;
  }

  chunking(H.lean, H.tile, H.tiles_per_chunk, H.chunks_per_ch, H.chunk_first);
  if (H.mixed && !H.tlist) chunking(true, H.lean_tile, H.lean_tiles_per_chunk, H.lean_chunks_per_ch, H.lean_chunk_first);
  if ((H.lean || H.mixed) && ns_override == 0 && !H.tlist) {
    // the lean launch's chunking for float outputs (longer chunks, see HostPlan)
    int32_t tile32 = 0;
    int cap32 = WFK_LEAN_TPC_F32;
    if (const char* e = std::getenv("WFK_TPC_F32")) {   // tuning override
      const int v = std::atoi(e);
      if (v >= 1 && v <= 64) cap32 = v;
    }
    // (beyond the double table's 8 tiles only where at least ~8 chunks per resident wave remain: C3's
    //  250 k tiles run best at 8-10 tiles per chunk -- 0.201 ms against 0.219 at 20 -- the 2.5 M tiles of
    //  256 x 1e7 at 16-20)
    chunking(true, tile32, H.f32_tiles_per_chunk, H.f32_chunks_per_ch, H.f32_chunk_first, cap32, 24576);
    if (H.f32_tiles_per_chunk <= (H.mixed ? H.lean_tiles_per_chunk : H.tiles_per_chunk)) {
      H.f32_chunk_first.clear();      // nothing to gain: the double table serves
      H.f32_tiles_per_chunk = 0;
    }
  }
  H.pool_real = !H.pool.empty();
  if (H.pool.empty()) H.pool.push_back(0.0);
  if (H.params.empty()) H.params.push_back(0.0);
  return WFK_OK;
}


// ---- big batches: channel blocks compiled on host threads ---------------------------------------
// The compile is O(#pieces) of scalar work with long-double seeds per op (2 us per piece): a fresh AWG sequence of
// 2048 rows x 1668 pulses is 7 s on one core.  Channels are independent (reference: one Waveform per channel,
// waveforms/waveform.py:529-563), so the job is cut into contiguous channel blocks, every block is compiled by
// wfk_compile on its own thread into its own HostPlan, and the plans are concatenated (indices rebased).  Taken for the
// two bulk shapes -- pure short-tier plans (their tables move with them) and pure lean plans without pool tables; anything
// else (mixed tiers, lean plans with INTERP / mollifier / SAMPLED tables, blocks that chose different tiers) returns WFK_RETRY_STD and the caller compiles in one piece.
int wfk_compile_blocks(const wfk_program* P, const wfk_grid* grid, int nthreads, HostPlan& H, std::string& err) {
  if (!P || !grid || nthreads < 2 || P->n_channels < 2 * nthreads) return WFK_RETRY_STD;
  {
    // validate the whole program once, here (a block skips it): a grid of zero points is enough for that
    HostPlan V;
    wfk_grid g0 = *grid;
    g0.n = 0; g0.has_last = 0;
    const int rc = compile_impl(P, &g0, nullptr, 0, V, err, true);
    if (rc != WFK_OK && rc != WFK_RETRY_STD) return rc;
  }
  const int K = nthreads;
  std::vector<HostPlan> parts((size_t)K);
  std::vector<std::string> errs((size_t)K);
  std::vector<int> rcs((size_t)K, WFK_OK);
  std::vector<int32_t> first((size_t)K + 1);
  for (int k = 0; k <= K; ++k) first[k] = (int32_t)((int64_t)P->n_channels * k / K);
  const bool t_fmul = g_no_short_fmul, t_mixed = g_keep_mixed_short, t_chirp = g_no_chirp;
  auto work = [&](int k) {
    g_no_short_fmul = t_fmul; g_keep_mixed_short = t_mixed; g_no_chirp = t_chirp;
    g_block_total_channels = P->n_channels;
    wfk_program Q = *P;
    const int32_t a = first[k];
    Q.n_channels = first[k + 1] - a;
    Q.ch_member_off += a; Q.ch_offset += a; Q.ch_tshift += a; Q.ch_clip_lo += a; Q.ch_clip_hi += a;
    try {
      rcs[k] = wfk_compile(&Q, grid, nullptr, 0, parts[k], errs[k]);
    } catch (...) {
      rcs[k] = WFK_ENOMEM;
      errs[k] = "out of host memory while compiling a channel block";
    }
    g_block_total_channels = 0;
  };
  {
    std::vector<std::thread> th;
    int started = 1;
    try {
      for (; started < K; ++started) th.emplace_back(work, started);
    } catch (...) {
    }
    work(0);
    for (auto& t : th) t.join();
    for (int k = started; k < K; ++k) work(k);               // (threads that could not be had)
  }
  for (int k = 0; k < K; ++k)
    if (rcs[k] != WFK_OK) { err = errs[k]; return rcs[k] == WFK_ENOMEM ? WFK_ENOMEM : WFK_RETRY_STD; }
  const HostPlan& A = parts[0];
  if (A.tlist || A.mixed || A.grid_as_tlist || !(A.shortp || A.lean)) return WFK_RETRY_STD;
  for (const HostPlan& B : parts)
    if (B.shortp != A.shortp || B.lean != A.lean || B.mixed || (B.pool_real && !B.shortp) || B.short_gave_up || B.ns != A.ns || B.tile != A.tile ||
        B.tiles_per_chunk != A.tiles_per_chunk || B.chunks_per_ch != A.chunks_per_ch ||
        B.f32_tiles_per_chunk != A.f32_tiles_per_chunk || B.f32_chunks_per_ch != A.f32_chunks_per_ch ||
        (B.n_corr > 0) != (A.n_corr > 0))
      return WFK_RETRY_STD;
  {
    // corrected carriers (family 6) in one block, closing ops / tables (families 1, 2, 4) in another: no build has both
    bool any_corr = false, any_other = false;
    for (const HostPlan& B : parts) {
      any_corr = any_corr || B.short_corr;
      any_other = any_other || (B.shortp && B.short_fam != 0 && B.short_fam != 6);
    }
    if (any_corr && any_other) return WFK_RETRY_STD;
  }
  H = HostPlan();
  H.tlist = false; H.n_channels = P->n_channels; H.n = A.n;
  H.t0 = A.t0; H.step = A.step; H.last = A.last; H.has_last = A.has_last; H.i0 = A.i0;
  H.ns = A.ns; H.tile = A.tile; H.tiles_per_chunk = A.tiles_per_chunk; H.chunks_per_ch = A.chunks_per_ch;
  H.lean = A.lean; H.shortp = A.shortp; H.mixed = false;
  H.f32_tiles_per_chunk = A.f32_tiles_per_chunk; H.f32_chunks_per_ch = A.f32_chunks_per_ch;
  H.lean_tile = A.lean_tile; H.lean_tiles_per_chunk = A.lean_tiles_per_chunk; H.lean_chunks_per_ch = A.lean_chunks_per_ch;
  H.member_idx.resize((size_t)P->n_members);
  double len_sum = 0.0, frac_sum = 0.0;
  for (int k = 0; k < K; ++k) {
    HostPlan& B = parts[k];
    if (H.params.size() & 1) H.params.push_back(0.0);
    const int64_t po = (int64_t)H.params.size();             // even: records keep their 16-byte alignment
    const int32_t pc = (int32_t)H.pieces.size(), so = (int32_t)H.s_slots.size();
    for (DevChannel c : B.channels) { c.piece_begin += pc; c.piece_end += pc; H.channels.push_back(c); }
    if (B.pool_real) {
      // short plans with tables (INTERP envelopes, sampled flat-top edges): the block's pool goes behind the others', and
      // the ops that name a table -- own-term ops over a table, closing table multipliers: [9], in 16-byte entries -- move with it
      if (H.pool.size() & 1) H.pool.push_back(0.0);
      const double tb = (double)(H.pool.size() / 2);
      H.pool.insert(H.pool.end(), B.pool.begin(), B.pool.end());
      for (const DevPiece& d : B.pieces) {
        if (d.n_blk <= 0 || !(d.flags & WFK_PF_SHORT)) continue;
        for (int32_t r = 0; r < d.n_blk; ++r) {
          double* o = B.params.data() + d.par_off + (int64_t)r * d.first_len;
          double* const end = o + d.first_len;
          while (o < end) {
            uint64_t word;
            std::memcpy(&word, o, sizeof word);
            const uint32_t w = (uint32_t)word;
            if (((w >> 4) & 3) == 3 && ((w & 128) ? !(w & 256) : (!(w & 1024) && (w & 3) == 2))) o[9] += tb;
            if (w & WFK_SH_LAST) break;
            o += (w & 3) > 1 ? WFK_SH_OP3 : WFK_SH_OP1;
          }
        }
      }
    }
    for (DevPiece d : B.pieces) { if (d.n_blk > 0) d.par_off += po; else d.par_off = po; H.pieces.push_back(d); }
    H.params.insert(H.params.end(), B.params.begin(), B.params.end());
    for (int32_t v : B.chunk_first) H.chunk_first.push_back(v + pc);
    for (int32_t v : B.f32_chunk_first) H.f32_chunk_first.push_back(v + pc);
    H.channel_complex.insert(H.channel_complex.end(), B.channel_complex.begin(), B.channel_complex.end());
    for (ShortUnit u : B.s_units) { u.ch += first[k]; u.slot0 += so; u.rec0 += po / 2; H.s_units.push_back(u); }
    H.s_slots.insert(H.s_slots.end(), B.s_slots.begin(), B.s_slots.end());
    for (size_t m = 0; m < B.member_idx.size(); ++m)
      if (!B.member_idx[m].empty()) H.member_idx[m] = std::move(B.member_idx[m]);
    H.n_fast += B.n_fast; H.n_direct += B.n_direct; H.n_fused += B.n_fused; H.n_generic += B.n_generic; H.n_corr += B.n_corr;
    H.lean_fam = std::max(H.lean_fam, B.lean_fam);
    H.lean_par = std::max(H.lean_par, B.lean_par); H.lean_ops = std::max(H.lean_ops, B.lean_ops);
    H.max_block_len = std::max(H.max_block_len, B.max_block_len);
    H.s_lds_samples = std::max(H.s_lds_samples, B.s_lds_samples);
    H.short_has_fmul = H.short_has_fmul || B.short_has_fmul;
    H.short_fam = std::max(H.short_fam, B.short_fam);
    H.short_needs_corr = H.short_needs_corr || B.short_needs_corr;
    H.short_corr = H.short_corr || B.short_corr;
    len_sum += B.mean_piece_len * (double)B.n_channels; frac_sum += B.foreign_frac * (double)B.n_channels;
  }
  H.mean_piece_len = len_sum / (double)P->n_channels;
  H.foreign_frac = frac_sum / (double)P->n_channels;
  H.pool_real = !H.pool.empty();
  if (H.pool.empty()) H.pool.assign(1, 0.0);
  if (H.shortp) {
    const int64_t nu = (int64_t)H.s_units.size();
    H.s_units_per_chunk = (int32_t)std::min<int64_t>(6, std::max<int64_t>(1, nu / 8192));
    if (const char* e = std::getenv("WFK_SH_UPC")) {
      const int v = std::atoi(e);
      if (v >= 1 && v <= 64) H.s_units_per_chunk = v;
    }
    if (H.s_slots.empty()) H.s_slots.push_back(0);
  }
  return WFK_OK;
}

// ---- uniform-grid detection -----------------------------------------------------------------
// Waveform.__call__(x) takes any sorted array (reference waveform.py:529-563), but scripts
// almost always pass np.linspace / np.arange output.  If x is BIT-IDENTICAL to the grid formula
// t[i] = fl(fl(i*step) + t0) (optionally with the last element overridden, np.linspace
// endpoint=True) the plan can be compiled in grid mode -- fused ops instead of one libm call per
// factor and sample, no upload of x -- and the device regenerates exactly the caller's times.
// The check is exact (every element is compared); anything else stays in tlist mode.
namespace {

struct GridProbe {
  const double* t;
  int64_t n;
  double t0;
  // (this file is built with -ffp-contract=off: the product rounds before the sum, as NumPy's does)
  double at(int64_t i, double step) const { return (double)i * step + t0; }
  // first index in [a, b) where the formula differs from t, or -1; *sign = formula - t there
  int64_t first_mismatch(double step, int64_t a, int64_t b, int* sign) const {
    for (int64_t c = a; c < b; c += 1024) {       // branch-free blocks (vectorisable), then locate
      const int64_t e = std::min(b, c + 1024);
      int bad = 0;
      const double dc = (double)c;                // c + k is exact in double (indices < 2^53)
      const double* tc = t + c;
      const int len = (int)(e - c);
      for (int k = 0; k < len; ++k) bad |= ((dc + (double)k) * step + t0) != tc[k];
      if (!bad) continue;
      for (int64_t i = c; i < e; ++i) {
        const double g = at(i, step);
        if (g != t[i]) {
          if (sign) *sign = g < t[i] ? -1 : 1;   // (a NaN in t lands in +1: never bracketed)
          return i;
        }
      }
    }
    return -1;
  }
  int64_t verify(double step, int64_t count, int* sign) const {
    if (count < (int64_t(1) << 20)) return first_mismatch(step, 0, count, sign);
    constexpr int nt = 8;
    std::atomic<int64_t> worst(INT64_MAX);
    int signs[nt] = {0};
    int64_t at_[nt];
    for (int k = 0; k < nt; ++k) at_[k] = -1;
    std::vector<std::thread> th;
    auto work = [&](int k) {
        const int64_t a = count * k / nt, b = count * (k + 1) / nt;
        // blocks of 64k so that a thread stops early once an earlier mismatch is known
        for (int64_t s0 = a; s0 < b && s0 < worst.load(std::memory_order_relaxed); s0 += 65536) {
          const int64_t m = first_mismatch(step, s0, std::min(b, s0 + 65536), &signs[k]);
          if (m >= 0) {
            at_[k] = m;
            int64_t w = worst.load();
            while (m < w && !worst.compare_exchange_weak(w, m)) {}
            break;
          }
        }
    };
    // (no exception may cross the C boundary: if threads cannot be had the slices run here)
    int started = 0;
    try {
      for (; started < nt - 1; ++started) th.emplace_back(work, started);
    } catch (...) {
    }
    for (int k = started; k < nt; ++k) work(k);
    for (auto& x : th) x.join();
    for (int k = 0; k < nt; ++k)
      if (at_[k] >= 0) { if (sign) *sign = signs[k]; return at_[k]; }   // lowest thread = lowest index
    return -1;
  }
};

}  // namespace

extern "C" int wfk_grid_detect(const double* t, int64_t n, wfk_grid* out) {
  if (!t || !out || n < 16) return 0;
  GridProbe P{t, n, t[0]};
  if (!(t[n - 1] > t[0]) || !std::isfinite(t[0]) || !std::isfinite(t[n - 1])) return 0;
  auto accept = [&](double step, int has_last) {
    out->t0 = t[0]; out->step = step; out->n = n; out->has_last = has_last; out->i0 = 0;
    out->last = has_last ? t[n - 1] : 0.0;
    return 1;
  };
  // np.linspace(a, b, n) (endpoint=True): step = (b - a) / (n - 1), last element := b
  {
    const double step = (t[n - 1] - t[0]) / (double)(n - 1);
    if (step > 0 && P.first_mismatch(step, 1, 16, nullptr) < 0 && P.verify(step, n - 1, nullptr) < 0)
      return accept(step, 0 + 1);
  }
  // np.arange(start, stop, d): t[1] - t[0] is the element step itself
  {
    const double step = t[1] - t[0];
    if (step > 0 && P.first_mismatch(step, 1, 16, nullptr) < 0 && P.verify(step, n, nullptr) < 0)
      return accept(step, 0);
  }
  // np.linspace(..., endpoint=False) and anything else of the same form: the step is not one of
  // the two above, but fl(fl(i*s) + t0) is monotone in s, so it is found by bisection over the
  // doubles between two brackets, on a subset of indices that grows with every counter-example.
  std::vector<int64_t> probe;
  for (int64_t i = n - 1; i >= 1; i = i * 7 / 8 - (i < 8 ? 1 : 0)) probe.push_back(i);
  auto classify = [&](double s) {   // -1: too small, +1: too big, 0: fits the subset, 2: contradiction
    bool lo = false, hi = false;
    for (int64_t i : probe) {
      const double g = P.at(i, s);
      lo = lo || g < t[i];
      hi = hi || g > t[i];
    }
    return lo && hi ? 2 : (lo ? -1 : (hi ? 1 : 0));
  };
  const double est = (t[n - 1] - t[0]) / (double)(n - 1);
  const double slack = 8.0 * (std::fabs(t[0]) + std::fabs(t[n - 1])) * 2.3e-16 / (double)(n - 1) + est * 1e-15;
  double lo = est - slack, hi = est + slack;
  if (!(lo > 0) || classify(lo) != -1 || classify(hi) != 1) return 0;
  for (int iter = 0; iter < 400; ++iter) {
    const double mid = lo + (hi - lo) / 2;
    if (!(mid > lo && mid < hi)) return 0;            // adjacent doubles: no step reproduces x
    const int c = classify(mid);
    if (c == 2) return 0;
    if (c < 0) { lo = mid; continue; }
    if (c > 0) { hi = mid; continue; }
    int sign = 0;
    const int64_t bad = P.verify(mid, n, &sign);
    if (bad < 0) return accept(mid, 0);
    probe.push_back(bad);                             // the counter-example narrows the bracket
    if (sign < 0) lo = mid; else hi = mid;
  }
  return 0;
}

// A sorted x that is several NumPy grids back to back -- np.concatenate of the chunks of a chunked job
// (waveforms/waveform.py:232: one np.linspace(..., endpoint=False) per chunk), a sequence sampled at two
// rates -- splits into runs that each pass wfk_grid_detect.  Run boundaries are where the spacing changes by
// more than the rounding of the grid formula can explain; every run is then verified element by element.
// -> number of runs (starts[k], grids[k]; the last run ends at n), 0 if x is not such a concatenation
//    (a run shorter than `min_len`, more than `max_runs` runs, one element off anywhere).
extern "C" int wfk_grid_detect_runs(const double* t, int64_t n, int64_t min_len, int32_t max_runs,
                                    int64_t* starts, wfk_grid* grids) {
  if (!t || !starts || !grids || n < 32 || max_runs < 1) return 0;
  if (min_len < 16) min_len = 16;
  std::vector<int64_t> cuts = {0};
  const double scale = 8.0 * 2.3e-16 * (std::max(std::fabs(t[0]), std::fabs(t[n - 1])) + std::fabs(t[n - 1] - t[0]));
  double prev = t[1] - t[0];
  int64_t run0 = 0;
  for (int64_t i = 1; i + 1 < n; ++i) {
    const double d = t[i + 1] - t[i];
    // within a grid the spacing wobbles by a few ulp of |t| (two roundings per element); a new grid starts
    // with a jump or with another step
    // (a few ulp of the largest |t| or |i * step| of ANY run: near t = 0 a grid's own wobble is that of
    //  its fl(i * step), not of the tiny t)
    const double tol = scale + 1e-12 * std::fabs(prev);
    if (!(std::fabs(d - prev) <= tol)) {
      // t[i + 1] opens a new run (the odd spacing is the gap between the runs) -- unless the run so far is
      // a single gap itself, i.e. two breaks in a row
      if (i + 1 - run0 < min_len) return 0;
      cuts.push_back(i + 1);
      if ((int64_t)cuts.size() > max_runs) return 0;
      run0 = i + 1;
      if (i + 2 < n) { prev = t[i + 2] - t[i + 1]; ++i; } else break;
    } else {
      prev = d;
    }
  }
  if (n - run0 < min_len) return 0;
  for (size_t k = 0; k < cuts.size(); ++k) {
    const int64_t a = cuts[k], b = k + 1 < cuts.size() ? cuts[k + 1] : n;
    if (!wfk_grid_detect(t + a, b - a, &grids[k])) return 0;
    starts[k] = a;
  }
  return (int)cuts.size();
}

// ---- sampler fused into the FIR transform at AWG rates: half-window entry lists ----------------------
// (reference chain: waveform.py:190-192 -> distortion.py:329-337; consumer: fir_short, wfk_fir_sampled.hip)
int wfk_chain_windows(const HostPlan& H, int64_t n, int64_t hop, int64_t lead, int64_t half_len, int64_t npairs,
                      std::vector<ShortWin>& wins, std::vector<uint32_t>& ents, std::string& bad) {
  bad.clear();
  if (!H.shortp) { bad = "not a short plan"; return WFK_EINVAL; }
  if (H.short_has_fmul) { bad = "table / mollifier envelopes (closing multipliers the fused chain does not evaluate)"; return WFK_EINVAL; }
  if (H.short_corr) { bad = "carriers with the grid-rounding correction (family 6 of the short tier: fir_short does not evaluate them)"; return WFK_EINVAL; }
  if (half_len <= 0 || half_len > 4096 || hop <= 0 || npairs < 0) { bad = "window geometry"; return WFK_EINVAL; }
  const int32_t nch = (int32_t)H.channels.size();
  wins.assign((size_t)npairs * nch * 2, ShortWin{});
  ents.clear();
  ents.reserve((size_t)(n / 12) * nch);
  for (int32_t c = 0; c < nch && bad.empty(); ++c) {
    const int32_t pe = H.channels[c].piece_end;
    int32_t q = H.channels[c].piece_begin;
    for (int64_t pr = 0; pr < npairs && bad.empty(); ++pr) {
      const int64_t s1 = 2 * pr * hop - lead;
      while (q < pe - 1 && H.pieces[q].stop <= s1) ++q;
      for (int hf = 0; hf < 2; ++hf) {
        const int64_t h0 = s1 + half_len * hf;
        const int64_t w0 = std::max<int64_t>(h0, 0), w1 = std::min<int64_t>(h0 + half_len, n);
        ShortWin W{};
        W.rec0 = -1;
        W.e0 = (int64_t)ents.size();
        for (int32_t qq = q; qq < pe && H.pieces[qq].start < w1; ++qq) {
          const DevPiece& D = H.pieces[qq];
          if (D.n_blk == 0 || D.stop <= w0) continue;          // zero stretch (the prefill) / before the half
          if (!(D.flags & WFK_PF_SHORT)) continue;             // the general kernel's piece: copied, below
          const int64_t a0 = std::max(D.start, w0), b0 = std::min(D.stop, w1);
          for (int64_t m = (a0 - D.start) / WFK_SH_SUB; D.start + m * WFK_SH_SUB < b0; ++m) {
            const int64_t r0 = D.start + m * WFK_SH_SUB, r1 = std::min<int64_t>(r0 + WFK_SH_SUB, D.stop);
            const int64_t aa = std::max(a0, r0), bb = std::min(b0, r1), len = bb - aa;
            if (len <= 0) continue;
            const int64_t rec16 = (D.par_off + m * (int64_t)D.first_len) / 2;
            if (W.rec0 < 0) W.rec0 = rec16;
            if (rec16 - W.rec0 > 0xffff) { bad = "op records of one window span more than 1 MB"; break; }
            const int64_t nsg = (len + WFK_SH_R - 1) / WFK_SH_R, bl = len / nsg, rem = len % nsg;
            int64_t k0 = 0;
            for (int64_t sg = 0; sg < nsg; ++sg) {
              const int64_t sl = bl + (sg < rem ? 1 : 0);
              ents.push_back(WFK_CW_ENTRY(rec16 - W.rec0, aa + k0 - h0, sl));
              k0 += sl;
            }
          }
          if (!bad.empty()) break;
        }
        if (W.rec0 < 0) W.rec0 = 0;
        W.cnt = (int32_t)((int64_t)ents.size() - W.e0);
        {
          // LDS layout of the half: per wave of 64 entries, the fullest of the 16 bank pairs the runs'
          // first elements fall into, plain vs padded
          int64_t cp = 0, cq = 0;
          for (int64_t e = W.e0; e < (int64_t)ents.size(); e += 64) {
            int plain[16] = {0}, padded[16] = {0}, wp = 0, wq = 0;
            for (int64_t k = e; k < std::min<int64_t>(e + 64, (int64_t)ents.size()); ++k) {
              const int o = (int)((ents[(size_t)k] >> 16) & 0xfff);
              wp = std::max(wp, ++plain[o & 15]);
              wq = std::max(wq, ++padded[(o + (o >> 4)) & 15]);
            }
            cp += wp; cq += wq;
          }
          W.pad = cq < cp ? 1 : 0;
        }
        if (H.mixed) {
          for (int32_t qq = q; qq < pe && H.pieces[qq].start < w1; ++qq) {
            const DevPiece& D = H.pieces[qq];
            if (D.n_blk == 0 || D.stop <= w0 || (D.flags & WFK_PF_SHORT)) continue;
            const int64_t a0 = std::max(D.start, w0), b0 = std::min(D.stop, w1);
            for (int64_t at = a0; at < b0; at += WFK_SH_R)
              ents.push_back(WFK_CW_ENTRY(0, at - h0, std::min<int64_t>(WFK_SH_R, b0 - at)));
          }
          W.ccnt = (int32_t)((int64_t)ents.size() - W.e0 - W.cnt);
        }
        wins[((size_t)c * npairs + pr) * 2 + hf] = W;
      }
    }
  }
  return bad.empty() ? WFK_OK : WFK_EINVAL;
}

// diagnostics (tools/): piece statistics of a plan compiled for another geometry -- out[0] pieces, out[1] live pieces,
// out[2] shortest / out[3] longest live piece (samples), out[4] lean, out[5] largest first_len
extern "C" int wfk_internal_geom_stats(const wfk_program* P, const wfk_grid* grid, int lane_stride, int ns, int64_t* out) {
  HostPlan H;
  std::string err;
  const int rc = wfk_compile_geom(P, grid, lane_stride, ns, H, err);
  if (rc) return rc;
  int64_t live = 0, lo = INT64_MAX, hi = 0, fl = 0;
  for (const DevPiece& p : H.pieces)
    if (p.n_blk > 0) {
      ++live;
      lo = std::min(lo, p.stop - p.start);
      hi = std::max(hi, p.stop - p.start);
      fl = std::max<int64_t>(fl, p.first_len);
    }
  out[0] = (int64_t)H.pieces.size(); out[1] = live; out[2] = live ? lo : 0; out[3] = hi; out[4] = H.lean; out[5] = fl;
  return 0;
}
