// wfk_fft4096.h -- 4096-point complex FFT held on chip by one 256-thread workgroup
// (shared by the FIR kernels wfk_fir_fused.hip and wfk_fir_sampled.hip).
//   4096 = 16 x 16 x 16: three radix-16 passes, each thread holds 16 complex points in
//   registers; two LDS exchanges per transform (layouts padded against bank conflicts).
// Index maps (N = 4096, forward):  input  v[n1] = x[256*n1 + tid]
//                                  output v[k3] = X[tid + 256*k3]      (natural order)
// so the output of the forward transform is exactly the input layout of the inverse.
#pragma once
#include <hip/hip_runtime.h>

namespace {

constexpr int FL = 4096;        // transform length
// LDS exchanges move real and imaginary parts in two rounds through ONE array of scalars
// (half the footprint of a complex image: 37 KB fp64, so 3-4 workgroups fit per CU instead
// of 2).  Row strides are padded for the 8-byte bank mapping: RA = 272 (== 16 mod 32) for the
// first exchange, RB = 289 (== 1 mod 32) for the second.
constexpr int RA = 272;
constexpr int RB = 289;
constexpr int LDS_ELEMS = 16 * RB;

template <typename T>
struct cx {
  T x, y;
};
template <typename T>
__device__ __forceinline__ cx<T> operator+(cx<T> a, cx<T> b) { return {a.x + b.x, a.y + b.y}; }
template <typename T>
__device__ __forceinline__ cx<T> operator-(cx<T> a, cx<T> b) { return {a.x - b.x, a.y - b.y}; }
template <typename T>
__device__ __forceinline__ cx<T> cmul(cx<T> a, cx<T> b) {
  return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x};
}
// multiply by -i (forward) or +i (inverse)
template <bool INV, typename T>
__device__ __forceinline__ cx<T> rot(cx<T> a) {
  return INV ? cx<T>{-a.y, a.x} : cx<T>{a.y, -a.x};
}

template <bool INV, typename T>
__device__ __forceinline__ void dft4(cx<T>& a, cx<T>& b, cx<T>& c, cx<T>& d) {
  const cx<T> s0 = a + c, s1 = a - c, s2 = b + d, s3 = rot<INV>(b - d);
  a = s0 + s2;
  b = s1 + s3;
  c = s0 - s2;
  d = s1 - s3;
}

// 16-point DFT in registers, natural order in and out: v[k] = sum_n v[n] W16^{+-nk}
template <bool INV, typename T>
__device__ __forceinline__ void dft16(cx<T> (&v)[16]) {
  // n = 4*n1 + n2: DFT4 over n1 for every n2
#pragma unroll
  for (int n2 = 0; n2 < 4; ++n2) dft4<INV>(v[n2], v[4 + n2], v[8 + n2], v[12 + n2]);
  // now v[4*k1 + n2]; twiddle W16^{n2*k1}
  const T c1 = (T)0.92387953251128673848, s1 = (T)0.38268343236508977173;  // cos/sin(pi/8)
  const T h = (T)0.70710678118654752440;
  const T sg = INV ? (T)1 : (T)-1;  // forward: exp(-i..)
  const cx<T> w1{c1, sg * s1}, w2{h, sg * h}, w3{s1, sg * c1}, w6{-h, sg * h}, w9{-c1, -sg * s1};
  v[4 * 1 + 1] = cmul(v[4 * 1 + 1], w1);
  v[4 * 1 + 2] = cmul(v[4 * 1 + 2], w2);
  v[4 * 1 + 3] = cmul(v[4 * 1 + 3], w3);
  v[4 * 2 + 1] = cmul(v[4 * 2 + 1], w2);
  v[4 * 2 + 2] = rot<INV>(v[4 * 2 + 2]);  // W16^4 = -+i
  v[4 * 2 + 3] = cmul(v[4 * 2 + 3], w6);
  v[4 * 3 + 1] = cmul(v[4 * 3 + 1], w3);
  v[4 * 3 + 2] = cmul(v[4 * 3 + 2], w6);
  v[4 * 3 + 3] = cmul(v[4 * 3 + 3], w9);
  // DFT4 over n2 for every k1: result A[k1 + 4*k2] sits at v[4*k1 + k2]
#pragma unroll
  for (int k1 = 0; k1 < 4; ++k1) dft4<INV>(v[4 * k1], v[4 * k1 + 1], v[4 * k1 + 2], v[4 * k1 + 3]);
  // transpose the 4x4 register tile to natural order
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = a + 1; b < 4; ++b) {
      const cx<T> t = v[4 * a + b];
      v[4 * a + b] = v[4 * b + a];
      v[4 * b + a] = t;
    }
}

// v[k] *= w^k, k = 1..15 (w = base twiddle)
template <typename T>
__device__ __forceinline__ void twiddle16(cx<T> (&v)[16], cx<T> w) {
  const cx<T> w2 = cmul(w, w), w4 = cmul(w2, w2), w8 = cmul(w4, w4);
  const cx<T> w3 = cmul(w2, w), w5 = cmul(w4, w), w6 = cmul(w4, w2), w7 = cmul(w4, w3);
  v[1] = cmul(v[1], w);
  v[2] = cmul(v[2], w2);
  v[3] = cmul(v[3], w3);
  v[4] = cmul(v[4], w4);
  v[5] = cmul(v[5], w5);
  v[6] = cmul(v[6], w6);
  v[7] = cmul(v[7], w7);
  v[8] = cmul(v[8], w8);
  v[9] = cmul(v[9], cmul(w8, w));
  v[10] = cmul(v[10], cmul(w8, w2));
  v[11] = cmul(v[11], cmul(w8, w3));
  v[12] = cmul(v[12], cmul(w8, w4));
  v[13] = cmul(v[13], cmul(w8, w5));
  v[14] = cmul(v[14], cmul(w8, w6));
  v[15] = cmul(v[15], cmul(w8, w7));
}

// in: v[n1] = x[256*n1 + tid]; out: v[k3] = X[tid + 256*k3].  tw[j] = exp(-2 pi i j/4096), j<256
template <bool INV, typename T>
__device__ __forceinline__ void fft4096(cx<T> (&v)[16], T* lds, cx<T> wa, cx<T> wb, int tid) {
  // pass 1: DFT16 over n1, twiddle W_4096^{tid*k1}
  dft16<INV>(v);
  {
    if (INV) wa.y = -wa.y;
    twiddle16(v, wa);
  }
  // exchange 1: E1[k1][m = tid]  ->  thread (k1 = tid>>4, n3 = tid&15) reads m = 16*n2 + n3
  {
    const int k1 = tid >> 4, n3 = tid & 15;
#pragma unroll
    for (int k = 0; k < 16; ++k) lds[k * RA + tid] = v[k].x;
    __syncthreads();
#pragma unroll
    for (int n2 = 0; n2 < 16; ++n2) v[n2].x = lds[k1 * RA + 16 * n2 + n3];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) lds[k * RA + tid] = v[k].y;
    __syncthreads();
#pragma unroll
    for (int n2 = 0; n2 < 16; ++n2) v[n2].y = lds[k1 * RA + 16 * n2 + n3];
    __syncthreads();
    // pass 2: DFT16 over n2, twiddle W_256^{n3*k2}
    dft16<INV>(v);
    if (INV) wb.y = -wb.y;
    twiddle16(v, wb);
    // exchange 2: E2[k1][n3][k2] -> thread (k1' = tid&15, k2' = tid>>4) reads over n3
    const int q1 = tid & 15, q2 = tid >> 4;
#pragma unroll
    for (int k2 = 0; k2 < 16; ++k2) lds[k1 * RB + n3 * 17 + k2] = v[k2].x;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j].x = lds[q1 * RB + j * 17 + q2];
    __syncthreads();
#pragma unroll
    for (int k2 = 0; k2 < 16; ++k2) lds[k1 * RB + n3 * 17 + k2] = v[k2].y;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j].y = lds[q1 * RB + j * 17 + q2];
    __syncthreads();
  }
  // pass 3: DFT16 over n3 -> k3
  dft16<INV>(v);
}


}  // namespace
