// wfk_spectral.hip -- whole-signal transfer-function stage:
//   out = irfft( rfft(sig) * H ),  H given per frequency bin k = 0 .. n/2  (f_k = k * fs / n)
// which is what the reference's FFT-domain operations do with scipy.fftpack on the host:
//   reflection(sig, A, tau, fs)          = ifft(fft(sig) * H_refl).real   distortion.py:208-210
//   correct_reflection(sig, A, tau, fs)  = ifft(fft(sig) / H_refl).real   distortion.py:213-223
// (H conjugate-symmetric => the real transforms are exact).  Batched rocFFT R2C / C2R of
// arbitrary length n plus one hand-written multiply kernel (1/n folded in).
#include <hip/hip_runtime.h>
#include <rocfft/rocfft.h>

#include <mutex>
#include <string>

#include "wfk.h"

extern "C" void wfk_internal_set_error(const char* msg);

namespace {
std::once_flag g_once;

template <typename C>
__global__ void __launch_bounds__(256) spec_mul(C* __restrict__ spec, const double2* __restrict__ H,
                                                int64_t nf, int64_t total, double scale) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const double2 h = H[idx % nf];
  const C a = spec[idx];
  C r;
  r.x = (decltype(r.x))((a.x * h.x - a.y * h.y) * scale);
  r.y = (decltype(r.y))((a.x * h.y + a.y * h.x) * scale);
  spec[idx] = r;
}

int sfail(int code, const std::string& m) {
  wfk_internal_set_error(m.c_str());
  return code;
}
}  // namespace

struct wfk_spectral_plan {
  int64_t n = 0, nf = 0;
  int32_t batch = 0, kind = 0;
  rocfft_plan fwd = nullptr, inv = nullptr;
  rocfft_execution_info info = nullptr;
  void* work = nullptr;
  void* spec = nullptr;
  void* tmp = nullptr;    // C2R may overwrite its input; R2C input is copied here when in == out
};

extern "C" {

int wfk_spectral_plan_destroy(wfk_spectral_plan* p) {
  if (!p) return WFK_OK;
  if (p->fwd) rocfft_plan_destroy(p->fwd);
  if (p->inv) rocfft_plan_destroy(p->inv);
  if (p->info) rocfft_execution_info_destroy(p->info);
  (void)hipFree(p->work);
  (void)hipFree(p->spec);
  (void)hipFree(p->tmp);
  delete p;
  return WFK_OK;
}

int wfk_spectral_plan_create(int64_t n, int32_t batch, int kind, wfk_spectral_plan** out) {
  if (!out) return sfail(WFK_EINVAL, "null out");
  *out = nullptr;
  if (n < 1 || batch < 1) return sfail(WFK_EINVAL, "bad spectral plan arguments");
  if (kind != WFK_OUT_F64 && kind != WFK_OUT_F32) return sfail(WFK_EINVAL, "kind must be F64 or F32");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
    (void)hipGetLastError();
    return sfail(WFK_EHIP, "no HIP device visible");
  }
  std::call_once(g_once, [] { rocfft_setup(); });
  wfk_spectral_plan* p = new wfk_spectral_plan();
  p->n = n; p->nf = n / 2 + 1; p->batch = batch; p->kind = kind;
  const rocfft_precision prec = kind == WFK_OUT_F32 ? rocfft_precision_single : rocfft_precision_double;
  const size_t es = kind == WFK_OUT_F32 ? 4 : 8;
  const size_t len[1] = {(size_t)n};
  bool ok = rocfft_plan_create(&p->fwd, rocfft_placement_notinplace, rocfft_transform_type_real_forward,
                               prec, 1, len, (size_t)batch, nullptr) == rocfft_status_success;
  ok = ok && rocfft_plan_create(&p->inv, rocfft_placement_notinplace, rocfft_transform_type_real_inverse,
                                prec, 1, len, (size_t)batch, nullptr) == rocfft_status_success;
  size_t wa = 0, wb = 0;
  if (ok) {
    rocfft_plan_get_work_buffer_size(p->fwd, &wa);
    rocfft_plan_get_work_buffer_size(p->inv, &wb);
    const size_t wbytes = wa > wb ? wa : wb;
    ok = rocfft_execution_info_create(&p->info) == rocfft_status_success;
    if (ok && wbytes)
      ok = hipMalloc(&p->work, wbytes) == hipSuccess &&
           rocfft_execution_info_set_work_buffer(p->info, p->work, wbytes) == rocfft_status_success;
    ok = ok && hipMalloc(&p->spec, (size_t)batch * p->nf * 2 * es) == hipSuccess;
    ok = ok && hipMalloc(&p->tmp, (size_t)batch * n * es) == hipSuccess;
  }
  if (!ok) {
    wfk_spectral_plan_destroy(p);
    return sfail(WFK_EHIP, "rocFFT plan / buffer creation failed");
  }
  *out = p;
  return WFK_OK;
}

/* rows are CONTIGUOUS (stride n); H_dev: n/2+1 complex128 values on the device */
int wfk_spectral_apply(wfk_spectral_plan* p, const void* in_dev, void* out_dev, const void* H_dev,
                       void* hip_stream) {
  if (!p || !in_dev || !out_dev || !H_dev) return sfail(WFK_EINVAL, "null argument");
  hipStream_t s = (hipStream_t)hip_stream;
  const size_t es = p->kind == WFK_OUT_F32 ? 4 : 8;
  if (rocfft_execution_info_set_stream(p->info, s) != rocfft_status_success)
    return sfail(WFK_EHIP, "rocfft set_stream failed");
  // rocFFT may overwrite the input of an out-of-place real transform: work on a copy
  if (hipMemcpyAsync(p->tmp, in_dev, (size_t)p->batch * p->n * es, hipMemcpyDeviceToDevice, s) != hipSuccess)
    return sfail(WFK_EHIP, "copy failed");
  void* ib[1] = {p->tmp};
  void* ob[1] = {p->spec};
  if (rocfft_execute(p->fwd, ib, ob, p->info) != rocfft_status_success)
    return sfail(WFK_EHIP, "rocfft forward failed");
  const int64_t total = (int64_t)p->batch * p->nf;
  const unsigned blocks = (unsigned)((total + 255) / 256);
  if (p->kind == WFK_OUT_F32)
    hipLaunchKernelGGL(spec_mul<float2>, dim3(blocks), dim3(256), 0, s, (float2*)p->spec,
                       (const double2*)H_dev, p->nf, total, 1.0 / (double)p->n);
  else
    hipLaunchKernelGGL(spec_mul<double2>, dim3(blocks), dim3(256), 0, s, (double2*)p->spec,
                       (const double2*)H_dev, p->nf, total, 1.0 / (double)p->n);
  void* ib2[1] = {p->spec};
  void* ob2[1] = {out_dev};
  if (rocfft_execute(p->inv, ib2, ob2, p->info) != rocfft_status_success)
    return sfail(WFK_EHIP, "rocfft inverse failed");
  if (hipGetLastError() != hipSuccess) return sfail(WFK_EHIP, "spectral kernel launch failed");
  return WFK_OK;
}

}  // extern "C"
