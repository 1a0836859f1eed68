// wfk_fir.hip -- FIR stage of the hot path:  predistort(sig, ker=ker)
// (reference: waveforms/distortion.py:329-337 -- zero-pad to [0_N, sig, 0_N],
//  scipy.signal.fftconvolve(..., 'full'), crop [N + K//2, 2N + K//2)), i.e.
//      out[i] = sum_k ker[k] * sig[i + K//2 - k]      with sig = 0 outside [0, N)
//
// The reference takes ONE real FFT of >= 3N + K - 1 points per channel.  Here: batched
// overlap-save on rocFFT.  Per chunk of channels:
//   gather   : overlapping length-L windows (hop M = L - K + 1) of the signal, zero padded,
//              window b of a row starts at sample b*M - (K-1) + K//2          [HIP kernel]
//   R2C FFT  : rocFFT, batch = channels * blocks
//   multiply : by the precomputed kernel spectrum (1/L folded in)             [HIP kernel]
//   C2R FFT  : rocFFT
//   scatter  : valid part [K-1, L) of every block -> out[b*M + r]            [HIP kernel]
// All buffers and rocFFT plans are created in wfk_fir_plan_create; wfk_fir_apply only
// enqueues work on the caller's stream.
#include <hip/hip_runtime.h>
#include <rocfft/rocfft.h>

#include <algorithm>
#include <cmath>
#include <complex>
#include <mutex>
#include <string>
#include <vector>

#include "wfk.h"

namespace {


template <typename T>
__global__ void __launch_bounds__(256) fir_gather(const T* __restrict__ in, int64_t in_stride,
                                                  T* __restrict__ win, int64_t n, int L, int M,
                                                  int lead, int64_t nblk) {
  // grid: x = L/256 segments, y = block, z = channel (within chunk)
  const int64_t b = blockIdx.y, ch = blockIdx.z;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= L) return;
  const int64_t src = b * M + i - lead;
  T v = (T)0;
  if (src >= 0 && src < n) v = in[ch * in_stride + src];
  win[(ch * nblk + b) * L + i] = v;
}

template <typename C>
__global__ void __launch_bounds__(256) fir_multiply(C* __restrict__ spec, const C* __restrict__ ks,
                                                    int nf, int64_t total) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int f = (int)(idx % nf);
  const C a = spec[idx], k = ks[f];
  C r;
  r.x = a.x * k.x - a.y * k.y;
  r.y = a.x * k.y + a.y * k.x;
  spec[idx] = r;
}

template <typename T>
__global__ void __launch_bounds__(256) fir_scatter(const T* __restrict__ win, T* __restrict__ out,
                                                   int64_t out_stride, int64_t n, int L, int M,
                                                   int K, int64_t nblk) {
  const int64_t b = blockIdx.y, ch = blockIdx.z;
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= M) return;
  const int64_t dst = b * M + r;
  if (dst >= n) return;
  out[ch * out_stride + dst] = win[(ch * nblk + b) * L + (K - 1) + r];
}

std::once_flag g_rocfft_once;

}  // namespace

// fused single-kernel path for short kernels (wfk_fir_fused.hip)
extern "C" int wfk_internal_fir_fused_launch(int kind, const void* in, int64_t in_stride, void* out,
                                             int64_t out_stride, const void* hspec, const void* tw,
                                             int64_t n, int M, int K, int lead, int64_t nblk,
                                             int32_t batch, int accumulate, void* stream, int64_t hspec_row_stride);
extern "C" int wfk_internal_fir_fused_len(void);

struct wfk_fir_plan {
  bool fused = false;
  int32_t nseg = 1, Kseg = 0;  // fused: the kernel is cut into nseg segments of Kseg taps, one
                               // pass of the fused kernel each (passes after the first accumulate)
  void* tw = nullptr;          // fused: exp(-2 pi i j / L), j < 256
  int32_t K = 0, batch = 0, kind = 0, L = 0, M = 0, lead = 0, chunk = 0;
  int64_t n = 0, nblk = 0;
  rocfft_plan fwd = nullptr, inv = nullptr, fwd_tail = nullptr, inv_tail = nullptr;
  int32_t tail = 0;  // channels in the last, smaller chunk (0: none)
  rocfft_execution_info info = nullptr;
  void* work = nullptr;
  void* win = nullptr;
  void* spec = nullptr;
  void* kspec = nullptr;
  int64_t krow = 0;            // per-row kernels (wfk_fir_plan_create_rows): complex elements between the
                               // spectra of consecutive rows; 0: one kernel for every row
};

extern "C" void wfk_internal_set_error(const char* msg);  // wfk_api.cpp (thread-local)

static int fir_fail(int code, const std::string& msg) {
  wfk_internal_set_error(msg.c_str());
  return code;
}

static int make_plans(wfk_fir_plan* p, int32_t channels, rocfft_plan* fwd, rocfft_plan* inv,
                      size_t* work_bytes) {
  const rocfft_precision prec =
      p->kind == WFK_OUT_F32 ? rocfft_precision_single : rocfft_precision_double;
  const size_t len[1] = {(size_t)p->L};
  const size_t nb = (size_t)channels * (size_t)p->nblk;
  const size_t nf = (size_t)p->L / 2 + 1;
  rocfft_plan_description df = nullptr, di = nullptr;
  const size_t one[1] = {1};
  if (rocfft_plan_description_create(&df) != rocfft_status_success) return -1;
  if (rocfft_plan_description_set_data_layout(df, rocfft_array_type_real,
                                              rocfft_array_type_hermitian_interleaved, nullptr,
                                              nullptr, 1, one, (size_t)p->L, 1, one, nf) !=
      rocfft_status_success)
    return -1;
  if (rocfft_plan_create(fwd, rocfft_placement_notinplace, rocfft_transform_type_real_forward, prec,
                         1, len, nb, df) != rocfft_status_success)
    return -1;
  rocfft_plan_description_destroy(df);
  if (rocfft_plan_description_create(&di) != rocfft_status_success) return -1;
  if (rocfft_plan_description_set_data_layout(di, rocfft_array_type_hermitian_interleaved,
                                              rocfft_array_type_real, nullptr, nullptr, 1, one, nf,
                                              1, one, (size_t)p->L) != rocfft_status_success)
    return -1;
  if (rocfft_plan_create(inv, rocfft_placement_notinplace, rocfft_transform_type_real_inverse, prec,
                         1, len, nb, di) != rocfft_status_success)
    return -1;
  rocfft_plan_description_destroy(di);
  size_t a = 0, b = 0;
  rocfft_plan_get_work_buffer_size(*fwd, &a);
  rocfft_plan_get_work_buffer_size(*inv, &b);
  *work_bytes = std::max(*work_bytes, std::max(a, b));
  return 0;
}

template <typename T, typename C>
static int fir_run(wfk_fir_plan* p, const void* in_dev, int64_t in_stride, void* out_dev,
                   int64_t out_stride, hipStream_t s) {
  const int L = p->L, M = p->M;
  const int64_t nf = L / 2 + 1;
  if (rocfft_execution_info_set_stream(p->info, s) != rocfft_status_success)
    return fir_fail(WFK_EHIP, "rocfft set_stream failed");
  for (int32_t c0 = 0; c0 < p->batch; c0 += p->chunk) {
    const int32_t nc = std::min(p->chunk, p->batch - c0);
    const bool tail = nc != p->chunk;
    const T* in = static_cast<const T*>(in_dev) + (int64_t)c0 * in_stride;
    T* out = static_cast<T*>(out_dev) + (int64_t)c0 * out_stride;
    T* win = static_cast<T*>(p->win);
    C* spec = static_cast<C*>(p->spec);
    dim3 gg((L + 255) / 256, (unsigned)p->nblk, (unsigned)nc);
    hipLaunchKernelGGL(fir_gather<T>, gg, dim3(256), 0, s, in, in_stride, win, p->n, L, M, p->lead,
                       p->nblk);
    void* ib[1] = {win};
    void* ob[1] = {spec};
    if (rocfft_execute(tail ? p->fwd_tail : p->fwd, ib, ob, p->info) != rocfft_status_success)
      return fir_fail(WFK_EHIP, "rocfft forward execute failed");
    const int64_t total = (int64_t)nc * p->nblk * nf;
    hipLaunchKernelGGL(fir_multiply<C>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, spec,
                       static_cast<const C*>(p->kspec), (int)nf, total);
    void* ib2[1] = {spec};
    void* ob2[1] = {win};
    if (rocfft_execute(tail ? p->inv_tail : p->inv, ib2, ob2, p->info) != rocfft_status_success)
      return fir_fail(WFK_EHIP, "rocfft inverse execute failed");
    dim3 gs((M + 255) / 256, (unsigned)p->nblk, (unsigned)nc);
    hipLaunchKernelGGL(fir_scatter<T>, gs, dim3(256), 0, s, win, out, out_stride, p->n, L, M, p->K,
                       p->nblk);
  }
  if (hipGetLastError() != hipSuccess) return fir_fail(WFK_EHIP, "FIR kernel launch failed");
  return WFK_OK;
}

// used by the sampler -> FIR chain (wfk_fir_sampled.hip): the kernel spectrum and twiddles of a
// plan that runs as ONE pass of the on-chip transform
extern "C" void wfk_internal_fir_tables(const wfk_fir_plan* p, const void** kspec, const void** tw,
                                        int* fused, int* nseg, int* K, int* lead) {
  *kspec = p->kspec; *tw = p->tw; *fused = p->fused ? 1 : 0; *nseg = p->nseg; *K = p->K;
  *lead = (p->K - 1) - p->K / 2;
}

extern "C" int64_t wfk_internal_fir_krow(const wfk_fir_plan* p) { return p->krow; }

extern "C" {

int wfk_fir_plan_destroy(wfk_fir_plan* p) {
  if (!p) return WFK_OK;
  if (p->fwd) rocfft_plan_destroy(p->fwd);
  if (p->inv) rocfft_plan_destroy(p->inv);
  if (p->fwd_tail) rocfft_plan_destroy(p->fwd_tail);
  if (p->inv_tail) rocfft_plan_destroy(p->inv_tail);
  if (p->info) rocfft_execution_info_destroy(p->info);
  (void)hipFree(p->work);
  (void)hipFree(p->win);
  (void)hipFree(p->spec);
  (void)hipFree(p->kspec);
  (void)hipFree(p->tw);
  delete p;
  return WFK_OK;
}

static int fir_plan_create(const double* ker_host, int32_t K, int64_t n, int32_t batch, int kind,
                           wfk_fir_plan** out, bool per_row);

int wfk_fir_plan_create(const double* ker_host, int32_t K, int64_t n, int32_t batch, int kind,
                        wfk_fir_plan** out) {
  return fir_plan_create(ker_host, K, n, batch, kind, out, false);
}

int wfk_fir_plan_create_rows(const double* kers_host, int32_t K, int64_t n, int32_t batch, int kind,
                             wfk_fir_plan** out) {
  return fir_plan_create(kers_host, K, n, batch, kind, out, true);
}

static int fir_plan_create(const double* ker_host, int32_t K, int64_t n, int32_t batch, int kind,
                           wfk_fir_plan** out, bool per_row) {
  if (!out) return fir_fail(WFK_EINVAL, "null out");
  *out = nullptr;
  if (!ker_host || K < 1 || n < 0 || batch < 1) return fir_fail(WFK_EINVAL, "bad FIR arguments");
  if (kind != WFK_OUT_F64 && kind != WFK_OUT_F32) return fir_fail(WFK_EINVAL, "FIR kind must be F64 or F32");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
    (void)hipGetLastError();
    return fir_fail(WFK_EHIP, "no HIP device visible");
  }
  std::call_once(g_rocfft_once, [] { rocfft_setup(); });
  wfk_fir_plan* p = new wfk_fir_plan();
  p->K = K; p->n = n; p->batch = batch; p->kind = kind;
  const int FL = wfk_internal_fir_fused_len();
  const char* force = getenv("WFK_FIR_ROCFFT");
  // The fused kernel takes kernels up to FL - 2559 = 1537 taps (hop >= 2560 = 5/8 of the
  // transform).  Longer kernels are cut into up to WFK_FIR_MAXSEG equal segments,
  //   out[i] = sum_j sum_k' ker[j*Kseg + k'] sig[(i + K//2 - j*Kseg) - k'],
  // i.e. one pass per segment with its own input offset, accumulating: 2-4 x 9.4 ms (+ the
  // re-read of `out`) against ~66 ms of the rocFFT pipeline on 256 x 1e7 fp64.
  const int KMAX = FL - 2559, WFK_FIR_MAXSEG = 4;
  p->nseg = (K + KMAX - 1) / KMAX;
  p->Kseg = (K + p->nseg - 1) / p->nseg;
  p->fused = p->nseg <= WFK_FIR_MAXSEG && batch <= 65535 && !(force && force[0] == '1');
  if (!p->fused) { p->nseg = 1; p->Kseg = K; }
  if (per_row && !p->fused) {
    wfk_fir_plan_destroy(p);
    return fir_fail(WFK_EUNSUP, "per-row FIR kernels need the on-chip transform (K <= 6148, batch <= 65535, WFK_FIR_ROCFFT unset)");
  }
  int L = 1024;
  while (L < 8 * K) L *= 2;               // hop M = L - K + 1 >= 7/8 L
  if (const char* e = getenv("WFK_FIR_L")) { int v = atoi(e); if (v >= 2 * K && (v & (v - 1)) == 0) L = v; }
  if (p->fused) L = FL;
  const int Kt = p->fused ? p->Kseg : K;             // taps per transform
  p->L = L; p->M = L - Kt + 1; p->lead = (Kt - 1) - K / 2;   // fused: + j*Kseg for segment j
  p->nblk = n > 0 ? (n + p->M - 1) / p->M : 0;
  if (n == 0) { *out = p; return WFK_OK; }
  const size_t es = kind == WFK_OUT_F32 ? 4 : 8;
  const size_t nf = p->fused ? (size_t)L : (size_t)L / 2 + 1;   // fused: full complex spectrum
  if (!p->fused) {
  const double per_ch = (double)p->nblk * ((double)L * es + (double)nf * 2 * es);
  int64_t chunk = (int64_t)(3.0e9 / per_ch);
  chunk = std::max<int64_t>(1, std::min<int64_t>(chunk, batch));
  p->chunk = (int32_t)chunk;
  p->tail = batch % p->chunk;
  size_t work_bytes = 0;
  if (make_plans(p, p->chunk, &p->fwd, &p->inv, &work_bytes) ||
      (p->tail && make_plans(p, p->tail, &p->fwd_tail, &p->inv_tail, &work_bytes))) {
    wfk_fir_plan_destroy(p);
    return fir_fail(WFK_EHIP, "rocFFT plan creation failed");
  }
  bool ok = rocfft_execution_info_create(&p->info) == rocfft_status_success;
  ok = ok && hipMalloc(&p->win, (size_t)p->chunk * p->nblk * L * es) == hipSuccess;
  ok = ok && hipMalloc(&p->spec, (size_t)p->chunk * p->nblk * nf * 2 * es) == hipSuccess;
  if (ok && work_bytes) {
    ok = hipMalloc(&p->work, work_bytes) == hipSuccess &&
         rocfft_execution_info_set_work_buffer(p->info, p->work, work_bytes) == rocfft_status_success;
  }
  if (!ok) {
    wfk_fir_plan_destroy(p);
    return fir_fail(WFK_ENOMEM, "FIR buffer allocation failed");
  }
  }
  const size_t nrow = per_row ? (size_t)batch : 1;
  p->krow = per_row ? (int64_t)((size_t)p->nseg * nf) : 0;
  if (hipMalloc(&p->kspec, nrow * (size_t)p->nseg * nf * 2 * es) != hipSuccess ||
      (p->fused && hipMalloc(&p->tw, 256 * 2 * es) != hipSuccess)) {
    wfk_fir_plan_destroy(p);
    return fir_fail(WFK_ENOMEM, "FIR buffer allocation failed");
  }
  // kernel spectrum on the host (K*L/2 flops, once): DFT of ker zero-padded to L, times 1/L
  std::vector<std::complex<long double>> tw(L);
  for (int i = 0; i < L; ++i) {
    long double th = -2.0L * 3.141592653589793238462643383279502884L * i / L;
    tw[i] = {cosl(th), sinl(th)};
  }
  std::vector<double> ks64(2 * nf * p->nseg * nrow);
  std::vector<float> ks32(2 * nf * p->nseg * nrow);
  if (!per_row) {
  for (int sg = 0; sg < p->nseg; ++sg) {
    const int k0 = sg * p->Kseg, k1 = std::min(K, k0 + (p->fused ? p->Kseg : K));
    for (size_t f = 0; f < nf; ++f) {
      std::complex<long double> acc = 0;
      for (int k = k0; k < k1; ++k)
        acc += (long double)ker_host[k] * tw[(size_t)((f * (size_t)(k - k0)) % L)];
      acc /= (long double)L;
      const size_t at = 2 * (sg * nf + f);
      ks64[at] = (double)acc.real(); ks64[at + 1] = (double)acc.imag();
      ks32[at] = (float)acc.real(); ks32[at + 1] = (float)acc.imag();
    }
  }
  } else {
    // one spectrum per row: a radix-2 transform in long double (K L / 2 products per kernel, as above, would
    // be seconds for a few thousand rows); bit-reversed input, L = 4096 here (fused path only)
    int lg = 0;
    while ((1 << lg) < L) ++lg;
    std::vector<std::complex<long double>> z((size_t)L);
    for (size_t r = 0; r < nrow; ++r)
      for (int sg = 0; sg < p->nseg; ++sg) {
        const int k0 = sg * p->Kseg, k1 = std::min(K, k0 + p->Kseg);
        for (int i = 0; i < L; ++i) {
          int rev = 0;
          for (int b = 0; b < lg; ++b) rev |= ((i >> b) & 1) << (lg - 1 - b);
          z[(size_t)rev] = i < k1 - k0 ? (long double)ker_host[r * (size_t)K + k0 + i] : 0.0L;
        }
        for (int half = 1; half < L; half <<= 1) {
          const int stepw = L / (2 * half);
          for (int i0 = 0; i0 < L; i0 += 2 * half)
            for (int j = 0; j < half; ++j) {
              const std::complex<long double> t = z[(size_t)(i0 + j + half)] * tw[(size_t)(j * stepw)];
              z[(size_t)(i0 + j + half)] = z[(size_t)(i0 + j)] - t;
              z[(size_t)(i0 + j)] += t;
            }
        }
        for (size_t f = 0; f < nf; ++f) {
          const std::complex<long double> acc = z[f] / (long double)L;
          const size_t at = 2 * ((r * p->nseg + sg) * nf + f);
          ks64[at] = (double)acc.real(); ks64[at + 1] = (double)acc.imag();
          ks32[at] = (float)acc.real(); ks32[at + 1] = (float)acc.imag();
        }
      }
  }
  hipError_t e = kind == WFK_OUT_F32
                     ? hipMemcpy(p->kspec, ks32.data(), ks32.size() * 4, hipMemcpyHostToDevice)
                     : hipMemcpy(p->kspec, ks64.data(), ks64.size() * 8, hipMemcpyHostToDevice);
  if (e == hipSuccess && p->fused) {
    std::vector<double> t64(512);
    std::vector<float> t32(512);
    for (int j = 0; j < 256; ++j) {
      t64[2 * j] = (double)tw[j].real(); t64[2 * j + 1] = (double)tw[j].imag();
      t32[2 * j] = (float)tw[j].real(); t32[2 * j + 1] = (float)tw[j].imag();
    }
    e = kind == WFK_OUT_F32 ? hipMemcpy(p->tw, t32.data(), 256 * 8, hipMemcpyHostToDevice)
                            : hipMemcpy(p->tw, t64.data(), 256 * 16, hipMemcpyHostToDevice);
  }
  if (e != hipSuccess) {
    wfk_fir_plan_destroy(p);
    return fir_fail(WFK_EHIP, "kernel spectrum upload failed");
  }
  *out = p;
  return WFK_OK;
}

int wfk_fir_apply(wfk_fir_plan* p, const void* in_dev, int64_t in_stride, void* out_dev,
                  int64_t out_stride, void* hip_stream) {
  if (!p) return fir_fail(WFK_EINVAL, "null plan");
  if (p->n == 0) return WFK_OK;
  if (!in_dev || !out_dev) return fir_fail(WFK_EINVAL, "null buffer");
  if (in_stride < p->n || out_stride < p->n) return fir_fail(WFK_EINVAL, "stride smaller than n");
  hipStream_t s = (hipStream_t)hip_stream;
  if (p->fused) {
    const size_t seg_bytes = (size_t)p->L * 2 * (p->kind == WFK_OUT_F32 ? 4 : 8);
    for (int sg = 0; sg < p->nseg; ++sg)
      if (wfk_internal_fir_fused_launch(p->kind, in_dev, in_stride, out_dev, out_stride,
                                        (const char*)p->kspec + sg * seg_bytes, p->tw, p->n, p->M,
                                        p->Kseg, p->lead + sg * p->Kseg, p->nblk, p->batch, sg > 0, s, p->krow))
        return fir_fail(WFK_EHIP, "fused FIR kernel launch failed");
    return WFK_OK;
  }
  if (p->nblk > 65535) return fir_fail(WFK_EINVAL, "signal too long for one rocFFT FIR plan (blocks > 65535)");
  if (p->kind == WFK_OUT_F32) return fir_run<float, float2>(p, in_dev, in_stride, out_dev, out_stride, s);
  return fir_run<double, double2>(p, in_dev, in_stride, out_dev, out_stride, s);
}

}  // extern "C"
