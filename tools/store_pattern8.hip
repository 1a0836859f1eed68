// Eighth sweep: what decides whether the contiguous-eighth XCD map is fast for a buffer?
// One large allocation; vary (a) the byte offset of the output inside it, (b) the spacing
// between the 8 XCD streams (region x starts at x*spacing; spacing >= its share).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void __launch_bounds__(64) k_chunk(double* out, int tpc, long per, long spacing_tiles, double v) {
  extern __shared__ double pad[];
  if (v == 123.0) pad[threadIdx.x] = v;
  const long b = blockIdx.x, x = b & 7, k = b >> 3;
  if (k >= per) return;
  const long t0 = x * spacing_tiles + k * tpc;
  for (int tt = 0; tt < tpc; ++tt) {
    double* p = out + (t0 + tt) * 1024 + threadIdx.x;
#pragma unroll
    for (int i = 0; i < 16; ++i) p[64 * i] = v + i;
  }
}

template <typename F>
static float timeit(F launch) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 2; ++i) launch();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  const int R = 6;
  for (int i = 0; i < R; ++i) launch();
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  return ms / R;
}

int main() {
  const long n = 256L * 10000000L;        // doubles written per launch
  const int tpc = 8;
  const long n_chunks = n / 1024 / tpc;   // 312500
  const long per = n_chunks / 8;          // 39062 chunks per XCD
  char* big;
  const size_t BIG = 40UL << 30;
  CK(hipMalloc(&big, BIG));
  printf("allocation at %p\n", (void*)big);
  CK(hipFuncSetAttribute((const void*)k_chunk, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  const unsigned lds = (160 * 1024 / 12) & ~255u;
  const unsigned g = (unsigned)(per * 8);
  const double bytes = (double)per * 8 * tpc * 8192;
  printf("-- offset sweep, streams packed (spacing = share = %.3f GB)\n", per * tpc * 8192.0 * 1e-9);
  for (size_t off : {0UL, 2UL << 20, 64UL << 20, 256UL << 20, 512UL << 20, 1UL << 30, 3UL << 29, 2UL << 30, 4UL << 30, 8UL << 30, 16UL << 30}) {
    float ms = timeit([&] { hipLaunchKernelGGL(k_chunk, dim3(g), dim3(64), lds, 0, (double*)(big + off), tpc, per, per * tpc, 1.0); });
    printf("offset %6zu MB   %6.3f ms  %5.2f TB/s\n", off >> 20, ms, bytes / ms * 1e-9);
  }
  printf("-- spacing sweep at offset 0\n");
  for (double gb : {2.560, 2.5625, 2.625, 2.6875, 2.75, 3.0, 3.25, 3.5, 4.0, 4.5}) {
    const long spacing_tiles = (long)(gb * (1UL << 30) / 8192);
    if ((size_t)(7 * spacing_tiles + per * tpc) * 8192 > BIG) continue;
    if (spacing_tiles < per * tpc) continue;
    float ms = timeit([&] { hipLaunchKernelGGL(k_chunk, dim3(g), dim3(64), lds, 0, (double*)big, tpc, per, spacing_tiles, 1.0); });
    printf("spacing %7.4f GiB   %6.3f ms  %5.2f TB/s\n", gb, ms, bytes / ms * 1e-9);
  }
  return 0;
}
