// wfk_api.cpp -- the extern "C" boundary of include/wfk.h (sampler part + helpers).
//
// A plan = host compile (wfk_compile.cpp) + one upload of the small device tables.
// wfk_plan_launch() only fills a KArgs struct and launches: no allocation, no sync.
#include <hip/hip_runtime.h>

#include <cstring>
#include <new>
#include <string>

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

struct wfk_plan {
  HostPlan h;
  bool on_device = false;
  DevChannel* d_channels = nullptr;
  DevPiece* d_pieces = nullptr;
  double* d_params = nullptr;
  double* d_pool = nullptr;
  int32_t* d_chunk_first = nullptr;
  double* d_tlist = nullptr;
  void* d_scratch = nullptr;   // wfk_plan_run_host output buffer
  size_t scratch_bytes = 0;
};

template <typename T>
static int upload(T** dst, const T* src, size_t count) {
  size_t bytes = (count ? count : 1) * sizeof(T);
  HIP_TRY(hipMalloc((void**)dst, bytes));
  if (count) HIP_TRY(hipMemcpy(*dst, src, count * sizeof(T), hipMemcpyHostToDevice));
  return WFK_OK;
}

static int plan_upload(wfk_plan* p, const double* tlist) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
    (void)hipGetLastError();
    p->on_device = false;  // host-only plan: index/info queries work, launch fails loudly
    return WFK_OK;
  }
  int rc;
  if ((rc = upload(&p->d_channels, p->h.channels.data(), p->h.channels.size()))) return rc;
  if ((rc = upload(&p->d_pieces, p->h.pieces.data(), p->h.pieces.size()))) return rc;
  if ((rc = upload(&p->d_params, p->h.params.data(), p->h.params.size()))) return rc;
  if ((rc = upload(&p->d_pool, p->h.pool.data(), p->h.pool.size()))) return rc;
  if ((rc = upload(&p->d_chunk_first, p->h.chunk_first.data(), p->h.chunk_first.size()))) return rc;
  if (tlist && (rc = upload(&p->d_tlist, tlist, (size_t)p->h.n))) return rc;
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

static int plan_create(const wfk_program* prog, const wfk_grid* grid, const double* tlist,
                       int64_t n, wfk_plan** out) {
  if (!out) return fail(WFK_EINVAL, "null out");
  *out = nullptr;
  wfk_plan* p = new (std::nothrow) wfk_plan();
  if (!p) return fail(WFK_ENOMEM, "out of host memory");
  std::string err;
  int rc = wfk_compile(prog, grid, tlist, n, p->h, err);
  if (rc) {
    delete p;
    return fail(rc, err);
  }
  rc = plan_upload(p, tlist);
  if (rc) {
    wfk_plan_destroy(p);
    return rc;
  }
  *out = p;
  return WFK_OK;
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
  if (p->on_device || p->d_channels) {
    (void)hipFree(p->d_channels);
    (void)hipFree(p->d_pieces);
    (void)hipFree(p->d_params);
    (void)hipFree(p->d_pool);
    (void)hipFree(p->d_chunk_first);
    (void)hipFree(p->d_tlist);
    (void)hipFree(p->d_scratch);
  }
  delete p;
  return WFK_OK;
}

int wfk_plan_get_info(const wfk_plan* p, wfk_plan_info* info) {
  if (!p || !info) return fail(WFK_EINVAL, "null argument");
  info->n_channels = p->h.n_channels;
  info->n = p->h.n;
  info->tile = p->h.tile;
  info->n_tiles = p->h.chunks_per_ch * p->h.n_channels;
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

int wfk_plan_launch(wfk_plan* p, void* out_dev, int64_t ch_stride, int out_kind, uint32_t flags,
                    void* hip_stream) {
  if (!p || (!out_dev && p->h.n > 0)) return fail(WFK_EINVAL, "null plan or output");
  if (!p->on_device)
    return fail(WFK_EHIP, "plan has no device tables (no HIP device was visible at plan creation)");
  if (ch_stride < p->h.n) return fail(WFK_EINVAL, "ch_stride smaller than samples per channel");
  if (p->h.n == 0 || p->h.n_channels == 0) return WFK_OK;
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
  a.step = p->h.step;
  a.last = p->h.last;
  a.has_last = p->h.has_last;
  std::string err;
  int rc = wfk_launch_sampler(a, p->h.n_channels, out_kind, p->h.tlist, p->h.lean,
                              p->h.n_generic > 0,
                              p->h.n_direct > 0,
                              hip_stream, err);
  return rc ? fail(rc, err) : WFK_OK;
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

int wfk_plan_run_host(wfk_plan* p, void* out_host, int64_t ch_stride, int out_kind) {
  if (!p) return fail(WFK_EINVAL, "null plan");
  size_t es = elem_size(out_kind);
  if (!es) return fail(WFK_EINVAL, "bad out_kind");
  if (p->h.n == 0 || p->h.n_channels == 0) return WFK_OK;
  if (!p->on_device)
    return fail(WFK_EHIP, "plan has no device tables (no HIP device was visible at plan creation)");
  size_t bytes = (size_t)p->h.n_channels * (size_t)p->h.n * es;
  if (bytes > p->scratch_bytes) {
    (void)hipFree(p->d_scratch);
    p->d_scratch = nullptr;
    p->scratch_bytes = 0;
    HIP_TRY(hipMalloc(&p->d_scratch, bytes));
    p->scratch_bytes = bytes;
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

int wfk_malloc(void** dev_ptr, size_t bytes) {
  if (!dev_ptr) return fail(WFK_EINVAL, "null dev_ptr");
  HIP_TRY(hipMalloc(dev_ptr, bytes ? bytes : 1));
  return WFK_OK;
}

int wfk_free(void* dev_ptr) {
  HIP_TRY(hipFree(dev_ptr));
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
