// Seventh sweep: XCD-aware chunk walk with a group size.  Workgroup b (XCD b&7, k = b>>3)
// writes chunk  c = G*(8*(k/G) + (b&7)) + k%G : every XCD writes runs of G consecutive chunks,
// the 8 XCDs together advance through one compact window of 8*G chunks.  G = "inf" is the
// contiguous-eighth map.  Each configuration is timed on 4 separately allocated buffers,
// because the result depends on where the buffer landed.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void __launch_bounds__(64) k_chunk(double* out, long n_tiles, int tpc, long n_chunks, long G, double v) {
  extern __shared__ double pad[];
  if (v == 123.0) pad[threadIdx.x] = v;
  const long b = blockIdx.x, x = b & 7, k = b >> 3;
  long c;
  if (G < 0) c = b;                                                    // plain
  else if (G == 0) { const long per = (n_chunks + 7) / 8; c = x * per + k; if (k >= per) return; }
  else c = G * (8 * (k / G) + x) + k % G;
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
  const int R = 6;
  for (int i = 0; i < R; ++i) launch();
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  return ms / R;
}

int main() {
  const long n = 256L * 10000000L;
  const long n_tiles = n / 1024;
  double* bufs[4];
  for (auto& p : bufs) CK(hipMalloc(&p, n * 8));
  CK(hipFuncSetAttribute((const void*)k_chunk, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  auto run = [&](const char* nm, int tpc, long G, int wpc) {
    const unsigned lds = wpc >= 32 ? 0 : ((160 * 1024 / wpc) & ~255u);
    const long n_chunks = (n_tiles + tpc - 1) / tpc;
    const unsigned g = (unsigned)(((n_chunks + 7) / 8) * 8);
    float t[4];
    for (int i = 0; i < 4; ++i)
      t[i] = timeit([&] { hipLaunchKernelGGL(k_chunk, dim3(g), dim3(64), lds, 0, bufs[i], n_tiles, tpc, n_chunks, G, 1.0); });
    printf("%-34s %6.3f %6.3f %6.3f %6.3f ms\n", nm, t[0], t[1], t[2], t[3]);
  };
  for (int i = 0; i < 4; ++i) printf("buf%d %p  ", i, (void*)bufs[i]);
  printf("\n");
  run("plain tpc=8 12w/CU", 8, -1, 12);
  run("eighths tpc=8 12w/CU", 8, 0, 12);
  run("eighths tpc=8 8w/CU", 8, 0, 8);
  run("eighths tpc=8 4w/CU", 8, 0, 4);
  run("eighths tpc=4 12w/CU", 4, 0, 12);
  run("eighths tpc=2 12w/CU", 2, 0, 12);
  run("eighths tpc=1 12w/CU", 1, 0, 12);
  run("eighths tpc=1 8w/CU", 1, 0, 8);
  run("eighths tpc=1 4w/CU", 1, 0, 4);
  run("eighths tpc=2 4w/CU", 2, 0, 4);
  run("plain tpc=1 4w/CU", 1, -1, 4);
  run("plain tpc=1 32w/CU", 1, -1, 32);
  return 0;
}
