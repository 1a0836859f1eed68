// wfk_fir_fused.hip -- fused overlap-save FIR for short kernels (K <= 1537).
//
// One 256-thread workgroup convolves TWO consecutive overlap-save windows of one row:
// the windows are packed as z = x1 + i*x2; because the FIR kernel is real,
// conv(z, h) = conv(x1, h) + i*conv(x2, h), so ONE complex FFT of length L = 4096, one
// pointwise multiply by the (full, complex) kernel spectrum, and one inverse FFT give both
// blocks with no real-FFT split/merge pass.  The whole transform lives on chip:
//   4096 = 16 x 16 x 16: three radix-16 passes, each thread holds 16 complex points in
//   registers; two LDS exchanges per transform (layouts padded against bank conflicts).
// HBM traffic per output sample: read 8*L/M + write 8 bytes (L/M = 1.33 for K = 1024)
// instead of the ~120 B/sample of the rocFFT pipeline in wfk_fir.hip.
//
// Index maps (N = 4096, forward):  input  v[n1] = x[256*n1 + tid]
//                                  output v[k3] = X[tid + 256*k3]      (natural order)
// so the output of the forward transform is exactly the input layout of the inverse.
#include <hip/hip_runtime.h>

#include "wfk.h"
#include "wfk_fft4096.h"

namespace {

template <typename T>
#ifndef WFK_FIR_WAVES
#define WFK_FIR_WAVES 3   // 3 workgroups per CU (138 VGPRs fp64); 4 spills and measured slower
#endif
__global__ void __launch_bounds__(256, WFK_FIR_WAVES) fir_fused(const T* __restrict__ in, int64_t in_stride,
                                                 T* __restrict__ out, int64_t out_stride,
                                                 const cx<T>* __restrict__ hspec,
                                                 const cx<T>* __restrict__ tw, int64_t n, int M,
                                                 int K, int lead, int accumulate, int64_t hrow) {
  __shared__ __attribute__((aligned(16))) T lds[LDS_ELEMS];
  const int tid = threadIdx.x;
  const int64_t pair = blockIdx.x, ch = blockIdx.y;
  hspec += ch * hrow;                 // per-row kernels (wfk_fir_plan_create_rows): this row's spectrum; else hrow = 0
  const T* row = in + ch * in_stride;
  T* orow = out + ch * out_stride;
  const int64_t b1 = 2 * pair, b2 = b1 + 1;
  const int64_t s1 = b1 * M - lead, s2 = b2 * M - lead;  // window starts (sample index)

  cx<T> v[16];
  // wave priority by phase (measured +2.7 %): a starting workgroup gets its 32 window loads
  // out ahead of the arithmetic of its neighbours, and one on the way out (inverse transform)
  // goes ahead of one on the way in, so slots free sooner
  __builtin_amdgcn_s_setprio(3);
  // interior pairs (all but the first and the last of a row): both windows and both output blocks lie
  // inside [0, n) -- no range checks, scalar base + lane offset + immediate addressing
  const bool interior = s1 >= 0 && s2 + FL <= n && (b2 + 1) * (int64_t)M <= n && !accumulate;
  if (interior) {
    const T* p1 = row + s1 + tid;
    const T* p2 = row + s2 + tid;
#pragma unroll
    for (int n1 = 0; n1 < 16; ++n1) {
      v[n1].x = p1[256 * n1];
      v[n1].y = p2[256 * n1];
    }
  } else {
#pragma unroll
    for (int n1 = 0; n1 < 16; ++n1) {
      const int i = 256 * n1 + tid;
      const int64_t j1 = s1 + i, j2 = s2 + i;
      v[n1].x = (j1 >= 0 && j1 < n) ? row[j1] : (T)0;
      v[n1].y = (j2 >= 0 && j2 < n) ? row[j2] : (T)0;
    }
  }
  __builtin_amdgcn_s_setprio(0);
  // base twiddles loaded up front with the window (one wait), not between the passes
  const cx<T> wa = tw[tid], wb = tw[16 * (tid & 15)];
  fft4096<false>(v, lds, wa, wb, tid);
#pragma unroll
  for (int k3 = 0; k3 < 16; ++k3) v[k3] = cmul(v[k3], hspec[tid + 256 * k3]);
  __builtin_amdgcn_s_setprio(2);
  fft4096<true>(v, lds, wa, wb, tid);
  if (interior) {
    T* const o1 = orow + b1 * M + (tid - (K - 1));
    T* const o2 = o1 + M;
#pragma unroll
    for (int q3 = 0; q3 < 16; ++q3) {
      const int r = tid + 256 * q3 - (K - 1);
      if (r >= 0 && r < M) {
        o1[256 * q3] = v[q3].x;
        o2[256 * q3] = v[q3].y;
      }
    }
    return;
  }
#pragma unroll
  for (int q3 = 0; q3 < 16; ++q3) {
    const int r = tid + 256 * q3 - (K - 1);
    if (r >= 0 && r < M) {
      const int64_t d1 = b1 * M + r, d2 = b2 * M + r;
      // accumulate: a later segment of a kernel longer than one transform allows (wfk_fir.hip)
      if (d1 < n) orow[d1] = accumulate ? orow[d1] + v[q3].x : v[q3].x;
      if (d2 < n) orow[d2] = accumulate ? orow[d2] + v[q3].y : v[q3].y;
    }
  }
}

}  // namespace

// launched from wfk_fir.hip
extern "C" int wfk_internal_fir_fused_launch(int kind, const void* in, int64_t in_stride, void* out,
                                             int64_t out_stride, const void* hspec, const void* tw,
                                             int64_t n, int M, int K, int lead, int64_t nblk,
                                             int32_t batch, int accumulate, void* stream, int64_t hspec_row_stride) {
  const dim3 grid((unsigned)((nblk + 1) / 2), (unsigned)batch);
  hipStream_t s = (hipStream_t)stream;
  if (kind == WFK_OUT_F32)
    hipLaunchKernelGGL(fir_fused<float>, grid, dim3(256), 0, s, (const float*)in, in_stride,
                       (float*)out, out_stride, (const cx<float>*)hspec, (const cx<float>*)tw, n, M,
                       K, lead, accumulate, hspec_row_stride);
  else
    hipLaunchKernelGGL(fir_fused<double>, grid, dim3(256), 0, s, (const double*)in, in_stride,
                       (double*)out, out_stride, (const cx<double>*)hspec, (const cx<double>*)tw, n,
                       M, K, lead, accumulate, hspec_row_stride);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

extern "C" int wfk_internal_fir_fused_len(void) { return FL; }
