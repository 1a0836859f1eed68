// Ninth sweep: stagger the 8 XCD streams.  XCD x walks its contiguous eighth cyclically,
// starting `x * stagger` chunks in:  chunk = x*per + (k + x*stagger) % per.  If the slow
// buffers are the ones where the 8 streams alias in the memory channels/banks, a stagger that
// de-synchronises their low address bits should make every buffer fast.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void __launch_bounds__(64) k_chunk(double* out, long n_tiles, int tpc, long n_chunks, long stagger, double v) {
  extern __shared__ double pad[];
  if (v == 123.0) pad[threadIdx.x] = v;
  const long b = blockIdx.x, x = b & 7, k = b >> 3;
  const long per = (n_chunks + 7) / 8;
  if (k >= per) return;
  const long c = x * per + (k + x * stagger) % per;
  if (c >= n_chunks) return;
  const long t0 = c * tpc;
  for (int tt = 0; tt < tpc; ++tt) {
    const long t = t0 + tt;
    if (t >= n_tiles) break;
    double* p = out + t * 1024 + threadIdx.x;
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
  const int R = 5;
  for (int i = 0; i < R; ++i) launch();
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  return ms / R;
}

int main() {
  const long n = 256L * 10000000L;
  const long n_tiles = n / 1024;
  const int NB = 6;
  double* bufs[NB];
  for (auto& p : bufs) CK(hipMalloc(&p, n * 8));
  CK(hipFuncSetAttribute((const void*)k_chunk, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  const int wpc = 12, tpc = 8;
  const unsigned lds = (160 * 1024 / wpc) & ~255u;
  const long n_chunks = (n_tiles + tpc - 1) / tpc;
  const unsigned g = (unsigned)(((n_chunks + 7) / 8) * 8);
  for (long st : {0L, 1L, 3L, 16L, 17L, 32L, 33L, 129L, 517L, 1031L, 4099L, 4883L}) {
    float t[NB];
    for (int i = 0; i < NB; ++i)
      t[i] = timeit([&] { hipLaunchKernelGGL(k_chunk, dim3(g), dim3(64), lds, 0, bufs[i], n_tiles, tpc, n_chunks, st, 1.0); });
    printf("stagger %5ld chunks (%8.2f MB):", st, st * tpc * 8192.0 / 1048576);
    for (int i = 0; i < NB; ++i) printf(" %6.3f", t[i]);
    printf("   worst %.3f\n", *std::max_element(t, t + NB));
  }
  return 0;
}
