// wfk_api.cpp -- the extern "C" boundary of include/wfk.h (sampler part + helpers).
//
// A plan = host compile (wfk_compile.cpp) + one upload of the small device tables.
// wfk_plan_launch() only fills a KArgs struct and launches: no allocation, no sync.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <map>
#include <mutex>
#include <thread>
#include <new>
#include <string>
#include <vector>

#include "wfk.h"
#include "wfk_internal.h"

static thread_local std::string g_err;

static int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}

#define HIP_TRY(expr)                                                              \
  do {                                                                             \
    hipError_t e_ = (expr);                                                        \
    if (e_ != hipSuccess)                                                          \
      return fail(WFK_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_));   \
  } while (0)

// ---- device block cache ------------------------------------------------------------------
// The drop-in calls (`wav(t)`: plan -> launch -> copy back -> destroy, reference
// waveform.py:529-563) are dominated by hipMalloc/hipFree when the waveform is small: seven
// allocations and frees per call cost more than the kernel.  Plans therefore take ONE block
// for all their tables (and one for the run_host output) from a per-device cache of
// power-of-two blocks, and give it back on destroy.  Cached bytes are capped; big blocks
// (the 20 GB outputs belong to the caller anyway) go straight to hipMalloc/hipFree.
namespace {

struct DevBlockCache {
  static constexpr size_t kMinBlock = 4096;
  static constexpr size_t kMaxCachedBlock = size_t(64) << 20;
  static constexpr size_t kMaxCachedTotal = size_t(512) << 20;
  std::mutex mu;
  std::map<std::pair<int, size_t>, std::vector<void*>> free_;   // (device, size) -> blocks
  std::map<void*, std::pair<size_t, int>> handed_;              // wfk_malloc blocks -> (capacity, device)
  size_t cached = 0;

  static size_t round_up(size_t bytes) {
    size_t b = kMinBlock;
    while (b < bytes) b <<= 1;
    return b;
  }
  // blocks are keyed by the device that was current when they were allocated (`*dev_out`)
  hipError_t get(size_t bytes, void** out, size_t* cap, int* dev_out) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    *dev_out = dev;
    const size_t b = bytes > kMaxCachedBlock ? bytes : round_up(bytes);
    if (b <= kMaxCachedBlock) {
      std::lock_guard<std::mutex> g(mu);
      auto it = free_.find({dev, b});
      if (it != free_.end() && !it->second.empty()) {
        *out = it->second.back();
        it->second.pop_back();
        cached -= b;
        *cap = b;
        return hipSuccess;
      }
    }
    *cap = b;
    hipError_t e = hipMalloc(out, b);
    if (e != hipSuccess) {
      // out of device memory with blocks parked here: give them back and try once more
      (void)hipGetLastError();
      trim();
      e = hipMalloc(out, b);
    }
    return e;
  }
  void trim() {
    std::lock_guard<std::mutex> g(mu);
    for (auto& kv : free_)
      for (void* p : kv.second) (void)hipFree(p);
    free_.clear();
    cached = 0;
  }
  void put(void* ptr, size_t cap, int dev) {
    if (!ptr) return;
    if (cap <= kMaxCachedBlock) {
      std::lock_guard<std::mutex> g(mu);
      if (cached + cap <= kMaxCachedTotal) {
        free_[{dev, cap}].push_back(ptr);
        cached += cap;
        return;
      }
    }
    (void)hipFree(ptr);
  }
};

DevBlockCache& dev_cache() {
  static DevBlockCache* c = new DevBlockCache();   // leaked on purpose: no teardown-order races
  return *c;
}

}  // namespace

// Makes `dev` the current HIP device for a scope (plans and blocks stay on the device they
// were created on, whatever the caller's current device is when it comes back to them).
struct DeviceGuard {
  int prev = -1;
  bool switched = false;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) == hipSuccess && prev != dev) switched = hipSetDevice(dev) == hipSuccess;
  }
  ~DeviceGuard() {
    if (switched) (void)hipSetDevice(prev);
  }
};

struct wfk_plan {
  HostPlan h;
  bool on_device = false;
  void* d_tables = nullptr;    // ONE block: channels | pieces | params | pool | chunk_first | tlist
  size_t tables_cap = 0;
  DevChannel* d_channels = nullptr;
  DevPiece* d_pieces = nullptr;
  double* d_params = nullptr;
  double* d_pool = nullptr;
  int32_t* d_chunk_first = nullptr;
  int32_t* d_lean_chunk_first = nullptr;   // mixed plans: chunk table of the lean launch
  int32_t* d_f32_chunk_first = nullptr;    // lean launch with float output: its own (longer) chunks
  double* d_tlist = nullptr;
  ShortUnit* d_units = nullptr;            // short plans (wfk_short.hip): wave units and lane slots
  uint32_t* d_slots = nullptr;
  void* d_scratch = nullptr;   // wfk_plan_run_host output buffer
  size_t scratch_bytes = 0;
  size_t scratch_cap = 0;
  int dev = 0;                 // device the blocks live on
  bool async_launch = false;   // launched on a caller stream since the last host-side sync
};

static size_t align256(size_t x) { return (x + 255) & ~size_t(255); }

static int plan_upload(wfk_plan* p, const double* tlist) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
    (void)hipGetLastError();
    p->on_device = false;  // host-only plan: index/info queries work, launch fails loudly
    return WFK_OK;
  }
  const HostPlan& h = p->h;
  const size_t b_ch = h.channels.size() * sizeof(DevChannel);
  const size_t b_pc = h.shortp && !h.mixed ? 0 : h.pieces.size() * sizeof(DevPiece);   // (the short kernel walks units, not pieces)
  const size_t b_pa = h.params.size() * sizeof(double);
  const size_t b_po = h.pool.size() * sizeof(double);
  const size_t b_cf = h.chunk_first.size() * sizeof(int32_t);
  const size_t b_lf = h.lean_chunk_first.size() * sizeof(int32_t);
  const size_t b_ff = h.f32_chunk_first.size() * sizeof(int32_t);
  const size_t b_un = h.s_units.size() * sizeof(ShortUnit), b_sl = h.s_slots.size() * sizeof(uint32_t);
  const size_t o_ch = 0, o_pc = align256(o_ch + b_ch), o_pa = align256(o_pc + b_pc),
               o_po = align256(o_pa + b_pa), o_cf = align256(o_po + b_po),
               o_lf = align256(o_cf + b_cf), o_ff = align256(o_lf + b_lf), o_un = align256(o_ff + b_ff),
               o_sl = align256(o_un + b_un), o_tl = align256(o_sl + b_sl);
  const size_t b_tl = tlist ? (size_t)h.n * sizeof(double) : 0;
  const size_t total = align256(o_tl + b_tl) + 256;
  HIP_TRY(dev_cache().get(total, &p->d_tables, &p->tables_cap, &p->dev));
  char* base = static_cast<char*>(p->d_tables);
  p->d_channels = reinterpret_cast<DevChannel*>(base + o_ch);
  p->d_pieces = reinterpret_cast<DevPiece*>(base + o_pc);
  p->d_params = reinterpret_cast<double*>(base + o_pa);
  p->d_pool = reinterpret_cast<double*>(base + o_po);
  p->d_chunk_first = reinterpret_cast<int32_t*>(base + o_cf);
  p->d_lean_chunk_first = reinterpret_cast<int32_t*>(base + o_lf);
  p->d_f32_chunk_first = reinterpret_cast<int32_t*>(base + o_ff);
  p->d_units = reinterpret_cast<ShortUnit*>(base + o_un);
  p->d_slots = reinterpret_cast<uint32_t*>(base + o_sl);
  p->d_tlist = tlist ? reinterpret_cast<double*>(base + o_tl) : nullptr;
  // the small tables travel in ONE copy; the time axis (as large as the output) on its own
  std::vector<char> stage(o_tl);
  if (b_ch) std::memcpy(stage.data() + o_ch, h.channels.data(), b_ch);
  if (b_pc) std::memcpy(stage.data() + o_pc, h.pieces.data(), b_pc);
  if (b_pa) std::memcpy(stage.data() + o_pa, h.params.data(), b_pa);
  if (b_po) std::memcpy(stage.data() + o_po, h.pool.data(), b_po);
  if (b_cf) std::memcpy(stage.data() + o_cf, h.chunk_first.data(), b_cf);
  if (b_lf) std::memcpy(stage.data() + o_lf, h.lean_chunk_first.data(), b_lf);
  if (b_ff) std::memcpy(stage.data() + o_ff, h.f32_chunk_first.data(), b_ff);
  if (b_un) std::memcpy(stage.data() + o_un, h.s_units.data(), b_un);
  if (b_sl) std::memcpy(stage.data() + o_sl, h.s_slots.data(), b_sl);
  if (o_tl) HIP_TRY(hipMemcpy(base, stage.data(), o_tl, hipMemcpyHostToDevice));
  if (b_tl) HIP_TRY(hipMemcpy(base + o_tl, tlist, b_tl, hipMemcpyHostToDevice));
  p->on_device = true;
  return WFK_OK;
}

extern "C" {

// shared with wfk_fir.hip (not part of the public header)
void wfk_internal_set_error(const char* msg) { g_err = msg ? msg : ""; }

int wfk_abi_version(void) { return WFK_ABI_VERSION; }

const char* wfk_last_error(void) { return g_err.c_str(); }

int wfk_device_count(int* count) {
  if (!count) return fail(WFK_EINVAL, "null count");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    *count = 0;
    return fail(WFK_EHIP, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
  }
  *count = n;
  return WFK_OK;
}

int wfk_set_device(int ordinal) {
  HIP_TRY(hipSetDevice(ordinal));
  return WFK_OK;
}

namespace {
// resets the compiler's thread-local "samples per lane of the time-list tier" on every way out of a scope
struct TlistNsGuard {
  explicit TlistNsGuard(int ns) { wfk_internal_tlist_ns(ns); }
  ~TlistNsGuard() { wfk_internal_tlist_ns(0); }
};
}  // namespace

static int plan_create_impl(const wfk_program* prog, const wfk_grid* grid, const double* tlist,
                            int64_t n, wfk_plan** out) {
  wfk_plan* p = new (std::nothrow) wfk_plan();
  if (!p) return fail(WFK_ENOMEM, "out of host memory");
  std::string err;
  int rc = WFK_RETRY_STD;
  try {
    // big grid batches (a fresh AWG sequence: thousands of rows x thousands of pulses): channel blocks on host threads
    if (grid && prog && prog->n_channels >= 32 && prog->n_pieces >= 8192) {
      int nt = (int)std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency()));
      if (const char* e = std::getenv("WFK_COMPILE_THREADS")) nt = std::atoi(e);
      nt = std::min(nt, prog->n_channels / 8);
      if (nt > 1) rc = wfk_compile_blocks(prog, grid, nt, p->h, err);
    }
    if (rc == WFK_RETRY_STD) rc = wfk_compile(prog, grid, tlist, n, p->h, err);
  } catch (...) {
    delete p;
    throw;
  }
  if (rc) {
    delete p;
    return fail(rc, err);
  }
  // A grid whose pieces are tens to hundreds of samples long (AWG rates) with shapes the short tier does not take
  // (libm shapes, chirps, sinc, derivatives of mollifiers ...): the standard tiers would evaluate EVERY piece over whole
  // wave tiles of 1024 samples -- 30 to 300 times the work.  The time-list tier evaluates sample by sample (its fused
  // groups pointwise, the rest by device libm at the exact NumPy times): the plan is compiled on the grid's own sample
  // times, one sample per lane.  (8 bytes per grid point of device memory, shared by all channels.)
  std::vector<double> grid_t;
  const char* nopw = std::getenv("WFK_NO_POINTWISE_GRID");
  if (grid && p->h.short_gave_up && grid->n > 0 && grid->n <= ((int64_t)1 << 24) && !(nopw && nopw[0] == '1')) {
    // (an allocation failure in here -- up to 128 MB of times plus a second compile -- keeps the grid plan
    // that is already compiled)
    try {
      grid_t.resize((size_t)grid->n);
      wfk_internal_grid_times(grid, grid_t.data());
      HostPlan h2;
      std::string err2;
      TlistNsGuard ns_guard(WFK_NS_TLIST_SMALL);
      const int rc2 = wfk_compile(prog, nullptr, grid_t.data(), grid->n, h2, err2);
      if (rc2 == WFK_OK) {
        h2.grid_as_tlist = true;
        h2.t0 = p->h.t0; h2.step = p->h.step; h2.last = p->h.last;     // introspection: the grid it stands for
        h2.has_last = p->h.has_last; h2.i0 = p->h.i0;
        p->h = std::move(h2);
        tlist = grid_t.data();
      }
    } catch (const std::bad_alloc&) {
      grid_t.clear();
    }
  }
  try {
    rc = plan_upload(p, tlist);
  } catch (...) {
    wfk_plan_destroy(p);
    throw;
  }
  if (rc) {
    wfk_plan_destroy(p);
    return rc;
  }
  *out = p;
  return WFK_OK;
}

// No exception crosses the C boundary: the compiler and the upload allocate host vectors.
static int plan_create(const wfk_program* prog, const wfk_grid* grid, const double* tlist,
                       int64_t n, wfk_plan** out) {
  if (!out) return fail(WFK_EINVAL, "null out");
  *out = nullptr;
  try {
    return plan_create_impl(prog, grid, tlist, n, out);
  } catch (const std::bad_alloc&) {
    return fail(WFK_ENOMEM, "out of host memory while compiling the plan");
  } catch (const std::exception& e) {
    return fail(WFK_EINVAL, std::string("plan creation failed: ") + e.what());
  }
}

int wfk_plan_create_grid(const wfk_program* prog, const wfk_grid* grid, wfk_plan** out) {
  if (!grid) return fail(WFK_EINVAL, "null grid");
  return plan_create(prog, grid, nullptr, 0, out);
}

int wfk_plan_create_tlist(const wfk_program* prog, const double* t_host, int64_t n,
                          wfk_plan** out) {
  if (!t_host && n > 0) return fail(WFK_EINVAL, "null t_host");
  static const double dummy = 0.0;
  return plan_create(prog, nullptr, t_host ? t_host : &dummy, n, out);
}

int wfk_plan_destroy(wfk_plan* p) {
  if (!p) return WFK_OK;
  if (p->d_tables || p->d_scratch) {
    // blocks go back to the cache and may be handed to the next plan at once: work launched
    // on a caller stream must have drained first (hipFree used to imply that)
    DeviceGuard guard(p->dev);                       // synchronise the OWNING device
    if (p->async_launch) (void)hipDeviceSynchronize();
    dev_cache().put(p->d_tables, p->tables_cap, p->dev);
    dev_cache().put(p->d_scratch, p->scratch_cap, p->dev);
  }
  delete p;
  return WFK_OK;
}

int wfk_plan_get_info(const wfk_plan* p, wfk_plan_info* info) {
  if (!p || !info) return fail(WFK_EINVAL, "null argument");
  info->n_channels = p->h.n_channels;
  info->n = p->h.n;
  info->tile = p->h.tile;
  info->n_tiles = p->h.shortp ? (int64_t)p->h.s_units.size() : p->h.chunks_per_ch * p->h.n_channels;
  info->n_pieces = (int32_t)p->h.pieces.size();
  info->param_doubles = (int64_t)p->h.params.size();
  info->n_fast = p->h.n_fast;
  info->n_direct = p->h.n_direct;
  info->n_fused = p->h.n_fused;
  info->n_generic = p->h.n_generic;
  return WFK_OK;
}

int wfk_plan_member_index(const wfk_plan* p, int32_t member, int64_t* idx, int32_t cap) {
  if (!p || member < 0 || member >= (int32_t)p->h.member_idx.size())
    return fail(WFK_EINVAL, "bad member");
  const auto& v = p->h.member_idx[member];
  for (int32_t i = 0; i < cap && i < (int32_t)v.size(); ++i) idx[i] = v[i];
  return (int)v.size();
}

int wfk_plan_channel_is_complex(const wfk_plan* p, int32_t channel) {
  if (!p || channel < 0 || channel >= p->h.n_channels) return fail(WFK_EINVAL, "bad channel");
  return p->h.channel_complex[channel];
}

// the compiled tables of a plan, for the fused sampler -> FIR chain at AWG rates (wfk_fir_sampled.hip builds
// its window tables from the short plan's pieces and reads the plan's own op records on the device)
void wfk_internal_plan_tables(const wfk_plan* p, const HostPlan** h, const double** d_params) {
  *h = p ? &p->h : nullptr;
  *d_params = p && p->on_device ? p->d_params : nullptr;
}

int64_t wfk_plan_table_bytes(const wfk_plan* p) {
  if (!p) return fail(WFK_EINVAL, "null plan");
  const HostPlan& h = p->h;
  size_t b = h.channels.size() * sizeof(DevChannel) + h.params.size() * sizeof(double) + h.pool.size() * sizeof(double);
  if (h.shortp) b += h.s_units.size() * sizeof(ShortUnit) + h.s_slots.size() * sizeof(uint32_t);
  if (!h.shortp || h.mixed) b += h.pieces.size() * sizeof(DevPiece) + (h.chunk_first.size() + h.lean_chunk_first.size()) * sizeof(int32_t);
  return (int64_t)b;
}

const char* wfk_plan_kernel_name(const wfk_plan* p, int out_kind) {
  if (!p || out_kind < 0 || out_kind > 3) return "";
  static thread_local std::string name;
  const char* T = (out_kind == WFK_OUT_F32 || out_kind == WFK_OUT_C64) ? "float" : "double";
  const char* cplx = (out_kind == WFK_OUT_C128 || out_kind == WFK_OUT_C64) ? "true" : "false";
  const HostPlan& h = p->h;
  const bool f32 = out_kind == WFK_OUT_F32 || out_kind == WFK_OUT_C64;
  // the general kernel's symbol: float outputs of plans with generic terms and of time lists run the builds with
  // double arithmetic (wfk_sample_wide)
  auto general = [&](bool tl, bool g, bool d, int ns) {
    const std::string tail = std::string(cplx) + "," + (tl ? "true" : "false") + "," + (g ? "true" : "false") + "," +
                             (d ? "true" : "false") + "," + std::to_string(ns) + ">";
    return f32 && (tl || g || d) ? "wfk_sample_wide<" + tail : std::string("wfk_sample<") + T + "," + tail;
  };
  if (h.shortp) {
    name = std::string("wfk_sample_short<") + T + "," + cplx + ",false," + std::to_string(WFK_SH_R) + "," +
           std::to_string(h.short_fam == 0 && out_kind == WFK_OUT_F32 && !(std::getenv("WFK_SH_NO_PK") && std::getenv("WFK_SH_NO_PK")[0] == '1') ? 3 : h.short_fam) + ">";
    if (h.mixed)      // pieces the short tier cannot take: a second launch of the general kernel
      name += " + " + general(false, h.n_direct > 0 || h.n_generic > 0, h.n_direct > 0, h.ns);
    return name.c_str();
  }
  const std::string lean_name = std::string("wfk_sample_lean<") + T + "," + cplx + "," + std::to_string(h.ns) +
      (h.n_corr > 0 && out_kind != WFK_OUT_F32 && out_kind != WFK_OUT_C64
           ? std::string(",true,") + std::to_string(h.lean_fam >= 1 ? 1 : 0) + ">"
           : std::string(",false,") + std::to_string(h.lean_fam) + ">");
  if (!h.tlist && h.lean) {
    name = lean_name;
  } else {
    // (a time-list plan whose every term fused runs the pointwise-ops-only build; any generic term brings
    //  the build with the direct tier, as before)
    const bool tl_full = h.tlist && (h.n_direct > 0 || h.n_generic > 0);
    const bool direct = tl_full || (!h.tlist && h.n_direct > 0), generic = direct || (!h.tlist && h.n_generic > 0);
    name = general(h.tlist, generic, direct, h.ns);
    if (h.mixed && h.tlist) name = general(true, false, false, h.ns) + " + " + name;
    else if (h.mixed) name = lean_name + " + " + name;   // two launches over disjoint pieces
  }
  return name.c_str();
}

// Launch the plan, or part `part` of `nparts` of it (single-channel plans only: the part's chunks are
// a contiguous sample range, returned in *s_lo / *s_hi).  nparts == 1: everything.
static int plan_launch_part(wfk_plan* p, void* out_dev, int64_t ch_stride, int out_kind, uint32_t flags,
                            void* hip_stream, int part, int nparts, int64_t* s_lo, int64_t* s_hi) {
  if (!p || (!out_dev && p->h.n > 0)) return fail(WFK_EINVAL, "null plan or output");
  if (!p->on_device)
    return fail(WFK_EHIP, "plan has no device tables (no HIP device was visible at plan creation)");
  if (ch_stride < p->h.n) return fail(WFK_EINVAL, "ch_stride smaller than samples per channel");
  if (s_lo) { *s_lo = 0; *s_hi = p->h.n; }
  if (p->h.n == 0 || p->h.n_channels == 0) return WFK_OK;
  if (nparts > 1 && (p->h.n_channels != 1 || p->h.mixed)) return fail(WFK_EINVAL, "partial launch of a multi-channel or mixed plan");
  auto sub = [&](int64_t total, int64_t& base, int64_t& count) {
    base = total * part / nparts;
    count = total * (part + 1) / nparts - base;
  };
  if (p->h.shortp) {
    SArgs sa{};
    sa.channels = p->d_channels;
    sa.units = p->d_units;
    sa.slots = p->d_slots;
    sa.recs = p->d_params;
    sa.out = out_dev;
    sa.ch_stride = ch_stride;
    sa.n_units = (int64_t)p->h.s_units.size();
    sa.units_per_chunk = p->h.s_units_per_chunk;
    const int64_t all = (sa.n_units + sa.units_per_chunk - 1) / sa.units_per_chunk;
    sub(all, sa.chunk_base, sa.n_chunks);
    sa.accumulate = (flags & WFK_ACCUMULATE) ? 1 : 0;
    sa.lds_samples = p->h.s_lds_samples;
    sa.fam = p->h.short_fam;
    sa.t0 = p->h.t0; sa.last = p->h.last;
    sa.dlast = p->h.has_last ? (double)(p->h.i0 + p->h.n - 1) : -1.0;
    sa.di0 = (double)p->h.i0;
    { const char* e = std::getenv("WFK_SH_NO_PK"); sa.pk = (e && e[0] == '1') ? 0 : 1; }
    sa.step = p->h.step;
    sa.pool = p->d_pool;
    if (s_lo && nparts > 1) {
      const int64_t u0 = sa.chunk_base * sa.units_per_chunk, u1 = (sa.chunk_base + sa.n_chunks) * sa.units_per_chunk;
      *s_lo = u0 < sa.n_units ? p->h.s_units[(size_t)u0].j0 : p->h.n;
      *s_hi = u1 < sa.n_units ? p->h.s_units[(size_t)u1].j0 : p->h.n;
    }
    if (hip_stream) p->async_launch = true;
    std::string serr;
    const bool foreign_only = (flags & WFK_PLAN_FOREIGN_ONLY) && p->h.mixed;   // (the chain at AWG rates samples the short pieces itself)
    const int src = sa.n_chunks > 0 && !foreign_only ? wfk_launch_short(sa, out_kind, hip_stream, serr) : WFK_OK;
    if (src) return fail(src, serr);
    if (!p->h.mixed) return WFK_OK;
  }
  KArgs a{};
  a.channels = p->d_channels;
  a.pieces = p->d_pieces;
  a.params = p->d_params;
  a.pool = p->d_pool;
  a.chunk_first = p->d_chunk_first;
  a.tlist = p->d_tlist;
  a.out = out_dev;
  a.ch_stride = ch_stride;
  a.n = p->h.n;
  a.chunks_per_ch = p->h.chunks_per_ch;
  a.n_chunks = p->h.chunks_per_ch * p->h.n_channels;
  a.tiles_per_chunk = p->h.tiles_per_chunk;
  a.accumulate = (flags & WFK_ACCUMULATE) ? 1 : 0;
  a.t0 = p->h.t0;
  a.i0 = p->h.i0;
  a.step = p->h.step;
  a.last = p->h.last;
  a.has_last = p->h.has_last;
  a.lean_par = p->h.lean_par;
  a.lean_ops = p->h.lean_ops;
  a.corr = p->h.n_corr > 0 ? 1 : 0;
  a.lean_fam = p->h.lean_fam;
  a.wavepriv = (p->h.tlist && p->h.ns == WFK_NS_TLIST_SMALL && p->h.max_block_len <= WFK_LDS_DOUBLES / 4 &&
                !std::getenv("WFK_NO_WAVEPRIV")) ? 1 : 0;
  a.reseed = WFK_LEAN_RESEED;
  // float outputs of the lean launch: longer chunks, rarer exact reseeds (HostPlan::f32_*)
  const bool f32_lean = (out_kind == WFK_OUT_F32 || out_kind == WFK_OUT_C64) && p->h.f32_tiles_per_chunk > 0;
  auto use_f32_chunks = [&](KArgs& k) {
    k.chunk_first = p->d_f32_chunk_first;
    k.chunks_per_ch = p->h.f32_chunks_per_ch;
    k.n_chunks = p->h.f32_chunks_per_ch * p->h.n_channels;
    k.tiles_per_chunk = p->h.f32_tiles_per_chunk;
    k.reseed = WFK_LEAN_RESEED_F32;
  };
  if (f32_lean && p->h.lean) use_f32_chunks(a);
  if (hip_stream) p->async_launch = true;
  std::string err;
  int rc = WFK_OK;
  if (p->h.shortp) {
    a.mixed = 1;      // the short launch above wrote its pieces and the zero stretches; now the rest
  } else if (p->h.mixed && p->h.tlist) {
    // time list: the fully fused and the zero pieces on the pointwise-ops build (same chunking) ...
    KArgs l = a;
    l.mixed = 1;
    rc = wfk_launch_sampler(l, p->h.n_channels, out_kind, true, p->h.ns, false, false, false, hip_stream, err);
    if (rc) return fail(rc, err);
    a.mixed = 1;   // ... then the pieces with generic terms on the build with the direct tier
  } else if (p->h.mixed) {
    // lean and zero pieces first (own chunking: one wave per workgroup) ...
    KArgs l = a;
    l.mixed = 1;
    l.chunk_first = p->d_lean_chunk_first;
    l.chunks_per_ch = p->h.lean_chunks_per_ch;
    l.n_chunks = p->h.lean_chunks_per_ch * p->h.n_channels;
    l.tiles_per_chunk = p->h.lean_tiles_per_chunk;
    if (f32_lean) use_f32_chunks(l);
    rc = wfk_launch_sampler(l, p->h.n_channels, out_kind, false, p->h.ns, true, false, false, hip_stream, err);
    if (rc) return fail(rc, err);
    a.mixed = 1;   // ... then the pieces with generic terms
  }
  if (nparts > 1) {
    const int64_t all = a.n_chunks;
    sub(all, a.chunk_base, a.n_chunks);
    const int64_t tile = (int64_t)(p->h.lean ? 64 : WFK_WG) * p->h.ns, span = tile * a.tiles_per_chunk;
    if (s_lo) {
      *s_lo = std::min<int64_t>(p->h.n, a.chunk_base * span);
      *s_hi = std::min<int64_t>(p->h.n, (a.chunk_base + a.n_chunks) * span);
    }
    if (a.n_chunks == 0) return WFK_OK;
  }
  rc = wfk_launch_sampler(a, p->h.n_channels, out_kind, p->h.tlist, p->h.ns, p->h.lean,
                          p->h.n_generic > 0 || (p->h.tlist && p->h.n_direct > 0),   // (time lists: either generic + direct or neither)
                          p->h.n_direct > 0 || (p->h.tlist && p->h.n_generic > 0),
                          hip_stream, err);
  return rc ? fail(rc, err) : WFK_OK;
}

int wfk_plan_launch(wfk_plan* p, void* out_dev, int64_t ch_stride, int out_kind, uint32_t flags,
                    void* hip_stream) {
  // (public flags only: WFK_PLAN_FOREIGN_ONLY belongs to wfk_internal_plan_launch_foreign)
  return plan_launch_part(p, out_dev, ch_stride, out_kind, flags & WFK_ACCUMULATE, hip_stream, 0, 1, nullptr, nullptr);
}

// the general-kernel part of a mixed short plan alone (the chain at AWG rates samples the short pieces itself
// and takes these through its workspace: wfk_fir_sampled.hip)
int wfk_internal_plan_launch_foreign(wfk_plan* p, void* out_dev, int64_t ch_stride, int out_kind, void* hip_stream) {
  return plan_launch_part(p, out_dev, ch_stride, out_kind, WFK_PLAN_FOREIGN_ONLY, hip_stream, 0, 1, nullptr, nullptr);
}

static size_t elem_size(int kind) {
  switch (kind) {
    case WFK_OUT_F64: return 8;
    case WFK_OUT_F32: return 4;
    case WFK_OUT_C128: return 16;
    case WFK_OUT_C64: return 8;
    default: return 0;
  }
}

// ---- pinned host blocks ---------------------------------------------------------------------
// Results of the drop-in calls (Waveform.__call__ / sample, reference waveform.py:529-563) are NumPy
// arrays of up to 80 MB.  A fresh pageable array costs a page fault per 4 KB while the copy lands in
// it (3 ms per 80 MB) and as much again when it is freed; a pinned block from this cache costs
// neither, takes the DMA directly, and lets the copy of one part overlap the kernel of the next.
extern "C++" {
namespace {
struct HostBlockCache {
  static constexpr size_t kMinBlock = size_t(1) << 16;
  static constexpr size_t kMaxCachedTotal = size_t(1) << 30;     // parked blocks
  static constexpr size_t kMaxLive = size_t(6) << 30;            // handed out + parked
  std::mutex mu;
  std::map<size_t, std::vector<void*>> free_;
  std::map<void*, size_t> handed_;
  size_t cached = 0, live = 0;
};
HostBlockCache& host_cache() {
  static HostBlockCache* c = new HostBlockCache();
  return *c;
}
}  // namespace
}  // extern "C++"

int wfk_host_alloc(void** host_ptr, size_t bytes) {
  if (!host_ptr) return fail(WFK_EINVAL, "null host_ptr");
  HostBlockCache& hc = host_cache();
  size_t b = HostBlockCache::kMinBlock;
  while (b < bytes) b <<= 1;
  if (bytes > (size_t(1) << 28)) b = (bytes + (size_t(1) << 24) - 1) & ~((size_t(1) << 24) - 1);   // big ones: 16 MB steps
  {
    std::lock_guard<std::mutex> g(hc.mu);
    auto it = hc.free_.find(b);
    if (it != hc.free_.end() && !it->second.empty()) {
      *host_ptr = it->second.back();
      it->second.pop_back();
      hc.cached -= b;
      hc.handed_[*host_ptr] = b;
      return WFK_OK;
    }
    if (hc.live + b > HostBlockCache::kMaxLive) return fail(WFK_ENOMEM, "pinned host memory budget exhausted");
  }
  void* ptr = nullptr;
  if (hipHostMalloc(&ptr, b, hipHostMallocDefault) != hipSuccess) {
    (void)hipGetLastError();
    return fail(WFK_ENOMEM, "hipHostMalloc failed");
  }
  std::lock_guard<std::mutex> g(hc.mu);
  hc.live += b;
  hc.handed_[ptr] = b;
  *host_ptr = ptr;
  return WFK_OK;
}

int wfk_host_free(void* host_ptr) {
  if (!host_ptr) return WFK_OK;
  HostBlockCache& hc = host_cache();
  size_t b = 0;
  {
    std::lock_guard<std::mutex> g(hc.mu);
    auto it = hc.handed_.find(host_ptr);
    if (it == hc.handed_.end()) return fail(WFK_EINVAL, "wfk_host_free: not a wfk_host_alloc block");
    b = it->second;
    hc.handed_.erase(it);
    if (hc.cached + b <= HostBlockCache::kMaxCachedTotal) {
      hc.free_[b].push_back(host_ptr);
      hc.cached += b;
      return WFK_OK;
    }
    hc.live -= b;
  }
  (void)hipHostFree(host_ptr);
  return WFK_OK;
}

extern "C++" {
namespace {
// two streams and a few events per device for wfk_plan_run_host's kernel / copy pipeline (created once)
struct HostPipe {
  std::mutex mu;
  hipStream_t compute = nullptr, copy = nullptr;
  hipEvent_t ev[8] = {};
  bool ok = false, tried = false;
};
HostPipe& host_pipe(int dev) {
  static std::mutex m;
  static std::map<int, HostPipe*> pipes;
  std::lock_guard<std::mutex> g(m);
  HostPipe*& p = pipes[dev];
  if (!p) p = new HostPipe();
  return *p;
}
}  // namespace
}  // extern "C++"

int wfk_plan_run_host(wfk_plan* p, void* out_host, int64_t ch_stride, int out_kind) {
  if (!p) return fail(WFK_EINVAL, "null plan");
  size_t es = elem_size(out_kind);
  if (!es) return fail(WFK_EINVAL, "bad out_kind");
  if (p->h.n == 0 || p->h.n_channels == 0) return WFK_OK;
  if (!p->on_device)
    return fail(WFK_EHIP, "plan has no device tables (no HIP device was visible at plan creation)");
  size_t bytes = (size_t)p->h.n_channels * (size_t)p->h.n * es;
  DeviceGuard guard(p->dev);   // scratch, launch and copy all happen on the plan's own device
  if (bytes > p->scratch_bytes) {
    dev_cache().put(p->d_scratch, p->scratch_cap, p->dev);
    p->d_scratch = nullptr;
    p->scratch_bytes = 0;
    int sdev = 0;
    HIP_TRY(dev_cache().get(bytes, &p->d_scratch, &p->scratch_cap, &sdev));
    p->scratch_bytes = bytes;
  }
  // Pipeline (single-channel plans into PINNED memory, wfk_host_alloc blocks: the big drop-in calls): the
  // output is evaluated in kParts launches on one stream and copied part by part on another, so all but
  // the first part's kernel time hides behind the copies.
  constexpr int kParts = 4;
  bool pinned = false;
  {
    std::lock_guard<std::mutex> g(host_cache().mu);
    auto it = host_cache().handed_.upper_bound(out_host);
    if (it != host_cache().handed_.begin()) {
      --it;
      pinned = (const char*)out_host >= (const char*)it->first &&
               (const char*)out_host + bytes <= (const char*)it->first + it->second;
    }
  }
  if (pinned && p->h.n_channels == 1 && !p->h.mixed && bytes >= (size_t(8) << 20)) {
    HostPipe& hp = host_pipe(p->dev);
    std::lock_guard<std::mutex> g(hp.mu);
    if (!hp.tried) {
      hp.tried = true;
      hp.ok = hipStreamCreateWithFlags(&hp.compute, hipStreamNonBlocking) == hipSuccess &&
              hipStreamCreateWithFlags(&hp.copy, hipStreamNonBlocking) == hipSuccess;
      for (int k = 0; k < kParts && hp.ok; ++k)
        hp.ok = hipEventCreateWithFlags(&hp.ev[k], hipEventDisableTiming) == hipSuccess;
      if (!hp.ok) (void)hipGetLastError();
    }
    if (hp.ok) {
      for (int k = 0; k < kParts; ++k) {
        int64_t lo = 0, hi = 0;
        int rc = plan_launch_part(p, p->d_scratch, p->h.n, out_kind, 0, hp.compute, k, kParts, &lo, &hi);
        if (rc) { (void)hipStreamSynchronize(hp.compute); (void)hipStreamSynchronize(hp.copy); return rc; }
        HIP_TRY(hipEventRecord(hp.ev[k], hp.compute));
        HIP_TRY(hipStreamWaitEvent(hp.copy, hp.ev[k], 0));
        if (hi > lo)
          HIP_TRY(hipMemcpyAsync((char*)out_host + (size_t)lo * es, (const char*)p->d_scratch + (size_t)lo * es,
                                 (size_t)(hi - lo) * es, hipMemcpyDeviceToHost, hp.copy));
      }
      HIP_TRY(hipStreamSynchronize(hp.copy));
      HIP_TRY(hipStreamSynchronize(hp.compute));
      p->async_launch = false;
      return WFK_OK;
    }
  }
  int rc = wfk_plan_launch(p, p->d_scratch, p->h.n, out_kind, 0, nullptr);
  if (rc) return rc;
  if (ch_stride == p->h.n) {
    HIP_TRY(hipMemcpy(out_host, p->d_scratch, bytes, hipMemcpyDeviceToHost));
  } else {
    HIP_TRY(hipMemcpy2D(out_host, (size_t)ch_stride * es, p->d_scratch, (size_t)p->h.n * es,
                        (size_t)p->h.n * es, (size_t)p->h.n_channels, hipMemcpyDeviceToHost));
  }
  return WFK_OK;
}

int wfk_host_all_finite(const double* host, int64_t n) {
  if (!host || n <= 0) return 1;
  // finite <=> exponent field != all ones; OR-reduce "exponent is all ones" over slices on a few threads
  auto scan = [host](int64_t a, int64_t b) -> bool {
    const uint64_t* p = reinterpret_cast<const uint64_t*>(host);
    for (int64_t i = a; i < b;) {
      const int64_t e = std::min(b, i + 4096);
      uint64_t bad = 0;
      for (; i < e; ++i) bad |= (uint64_t)(((p[i] >> 52) & 0x7ff) == 0x7ff);
      if (bad) return false;
    }
    return true;
  };
  const int nt = n < (int64_t(1) << 20) ? 1 : (int)std::min<int64_t>(8, std::max(1u, std::thread::hardware_concurrency()));
  if (nt == 1) return scan(0, n) ? 1 : 0;
  std::vector<std::thread> th;
  std::vector<char> ok((size_t)nt, 1);
  for (int k = 0; k < nt; ++k)
    th.emplace_back([&, k] { ok[(size_t)k] = scan(n * k / nt, n * (k + 1) / nt) ? 1 : 0; });
  for (auto& t : th) t.join();
  for (char c : ok) if (!c) return 0;
  return 1;
}

int wfk_malloc(void** dev_ptr, size_t bytes) {
  if (!dev_ptr) return fail(WFK_EINVAL, "null dev_ptr");
  size_t cap = 0;
  int dev = 0;
  HIP_TRY(dev_cache().get(bytes ? bytes : 1, dev_ptr, &cap, &dev));
  std::lock_guard<std::mutex> g(dev_cache().mu);
  dev_cache().handed_[*dev_ptr] = {cap, dev};
  return WFK_OK;
}

int wfk_free(void* dev_ptr) {
  if (!dev_ptr) return WFK_OK;
  size_t cap = 0;
  int dev = 0;
  {
    std::lock_guard<std::mutex> g(dev_cache().mu);
    auto it = dev_cache().handed_.find(dev_ptr);
    if (it == dev_cache().handed_.end()) return fail(WFK_EINVAL, "wfk_free: not a wfk_malloc block");
    cap = it->second.first;
    dev = it->second.second;
    dev_cache().handed_.erase(it);
  }
  // like hipFree, freeing waits for the block's OWN device: the block may be reused at once
  {
    DeviceGuard guard(dev);
    HIP_TRY(hipDeviceSynchronize());
  }
  dev_cache().put(dev_ptr, cap, dev);
  return WFK_OK;
}

int wfk_memcpy_h2d(void* dst_dev, const void* src_host, size_t bytes) {
  HIP_TRY(hipMemcpy(dst_dev, src_host, bytes, hipMemcpyHostToDevice));
  return WFK_OK;
}

int wfk_memcpy_d2h(void* dst_host, const void* src_dev, size_t bytes) {
  HIP_TRY(hipMemcpy(dst_host, src_dev, bytes, hipMemcpyDeviceToHost));
  return WFK_OK;
}

int wfk_memset(void* dst_dev, int byte, size_t bytes) {
  HIP_TRY(hipMemset(dst_dev, byte, bytes));
  return WFK_OK;
}

int wfk_stream_sync(void* hip_stream) {
  HIP_TRY(hipStreamSynchronize((hipStream_t)hip_stream));
  return WFK_OK;
}

}  // extern "C"
