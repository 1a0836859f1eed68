// Sixth sweep: is the many-streams penalty a TLB effect (page-granular) or a DRAM-row effect?
//  A. one-shot 4 KB WGs, visit order scrambled only WITHIN blocks of B bytes, blocks in linear order
//  B. the lean sampler's chunk walk (1 wave/WG, TPC x 8 KB tiles) with an XCD-aware chunk map:
//     WG b runs on XCD b%8; give XCD x the x-th eighth of the chunks, walked linearly.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void __launch_bounds__(256) k_scr(double* out, long nwg, long per_block, double v) {
  long w = blockIdx.x;
  const long blk = w / per_block, i = w % per_block;
  w = blk * per_block + (i * 7919L) % per_block;   // per_block is a power of two, 7919 odd: a permutation
  if (w >= nwg) return;
  double2* p = reinterpret_cast<double2*>(out + w * 512) + threadIdx.x;
  *p = make_double2(v, v);
}

template <bool XCD>
__global__ void __launch_bounds__(64) k_chunk(double* out, long n_tiles, int tpc, long n_chunks, double v) {
  extern __shared__ double pad[];
  if (v == 123.0) pad[threadIdx.x] = v;
  long c = blockIdx.x;
  if (XCD) { const long per = (n_chunks + 7) / 8; c = (c & 7) * per + (c >> 3); if ((blockIdx.x >> 3) >= per || c >= n_chunks) return; }
  const long t0 = c * tpc;
  for (int tt = 0; tt < tpc; ++tt) {
    const long t = t0 + tt;
    if (t >= n_tiles) break;
    double* p = out + t * 1024 + threadIdx.x;
#pragma unroll
    for (int k = 0; k < 16; ++k) p[64 * k] = v + k;
  }
}

template <typename F>
static float timeit(F launch) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 2; ++i) launch();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  const int R = 8;
  for (int i = 0; i < R; ++i) launch();
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  return ms / R;
}

int main() {
  const long n = 256L * 10000000L;
  const long nwg = n / 512;
  double* out;
  CK(hipMalloc(&out, n * 8));
  for (long B : {4096L, 65536L, 1L << 20, 2L << 20, 8L << 20, 64L << 20, 1L << 30}) {
    const long per = B / 4096;
    const long g = (nwg + per - 1) / per * per;
    float ms = timeit([&] { hipLaunchKernelGGL(k_scr, dim3((unsigned)g), dim3(256), 0, 0, out, nwg, per, 1.0); });
    printf("one-shot, scrambled within %8ld KB blocks   %7.3f ms  %5.2f TB/s\n", B / 1024, ms, n * 8.0 / ms * 1e-9);
  }
  CK(hipFuncSetAttribute((const void*)k_chunk<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CK(hipFuncSetAttribute((const void*)k_chunk<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  const long n_tiles = n / 1024;
  for (int tpc : {1, 2, 4, 8, 32}) {
    for (int wpc : {4, 8, 12, 32}) {
      const unsigned lds = wpc >= 32 ? 0 : ((160 * 1024 / wpc) & ~255u);
      const long n_chunks = (n_tiles + tpc - 1) / tpc;
      const unsigned g = (unsigned)(((n_chunks + 7) / 8) * 8);
      float a = timeit([&] { hipLaunchKernelGGL(k_chunk<false>, dim3((unsigned)n_chunks), dim3(64), lds, 0, out, n_tiles, tpc, n_chunks, 1.0); });
      float b = timeit([&] { hipLaunchKernelGGL(k_chunk<true>, dim3(g), dim3(64), lds, 0, out, n_tiles, tpc, n_chunks, 1.0); });
      printf("chunk walk tpc=%2d w/CU=%2d   plain %6.3f ms %5.2f TB/s   xcd-aware %6.3f ms %5.2f TB/s\n", tpc, wpc, a,
             n * 8.0 / a * 1e-9, b, n * 8.0 / b * 1e-9);
    }
  }
  return 0;
}
